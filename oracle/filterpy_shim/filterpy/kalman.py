"""KalmanFilter stand-in (see package docstring).  Only the members the
reference touches are provided: dim_x/dim_z ctor, x, P, Q, R, F, H, predict(F=,Q=),
update(z, R=)."""
import numpy as np
from numpy import dot, eye, zeros, isscalar


class KalmanFilter(object):
    def __init__(self, dim_x, dim_z, dim_u=0):
        if dim_x < 1 or dim_z < 1:
            raise ValueError("dim_x and dim_z must be >= 1")
        self.dim_x, self.dim_z, self.dim_u = dim_x, dim_z, dim_u
        self.x = zeros((dim_x, 1))
        self.P = eye(dim_x)
        self.Q = eye(dim_x)
        self.B = None
        self.F = eye(dim_x)
        self.H = zeros((dim_z, dim_x))
        self.R = eye(dim_z)
        self._alpha_sq = 1.0
        self.M = zeros((dim_x, dim_z))
        self.z = np.array([[None] * dim_z]).T
        self.K = zeros((dim_x, dim_z))
        self.y = zeros((dim_z, 1))
        self.S = zeros((dim_z, dim_z))
        self.SI = zeros((dim_z, dim_z))
        self._I = eye(dim_x)
        self.x_prior, self.P_prior = self.x.copy(), self.P.copy()
        self.x_post, self.P_post = self.x.copy(), self.P.copy()
        self.inv = np.linalg.inv

    def predict(self, u=None, B=None, F=None, Q=None):
        if B is None:
            B = self.B
        if F is None:
            F = self.F
        if Q is None:
            Q = self.Q
        elif isscalar(Q):
            Q = eye(self.dim_x) * Q
        if B is not None and u is not None:
            self.x = dot(F, self.x) + dot(B, u)
        else:
            self.x = dot(F, self.x)
        self.P = self._alpha_sq * dot(dot(F, self.P), F.T) + Q
        self.x_prior, self.P_prior = self.x.copy(), self.P.copy()

    def update(self, z, R=None, H=None):
        if z is None:
            self.z = np.array([[None] * self.dim_z]).T
            self.x_post, self.P_post = self.x.copy(), self.P.copy()
            self.y = zeros((self.dim_z, 1))
            return
        if R is None:
            R = self.R
        elif isscalar(R):
            R = eye(self.dim_z) * R
        if H is None:
            z = _reshape_z(z, self.dim_z, self.x.ndim)
            H = self.H
        self.y = z - dot(H, self.x)
        PHT = dot(self.P, H.T)
        self.S = dot(H, PHT) + R
        self.SI = self.inv(self.S)
        self.K = dot(PHT, self.SI)
        self.x = self.x + dot(self.K, self.y)
        I_KH = self._I - dot(self.K, H)
        self.P = dot(dot(I_KH, self.P), I_KH.T) + dot(dot(self.K, R), self.K.T)
        self.z = np.array(z, copy=True)
        self.x_post, self.P_post = self.x.copy(), self.P.copy()


def _reshape_z(z, dim_z, ndim):
    z = np.atleast_2d(z)
    if z.shape[1] == dim_z:
        z = z.T
    if z.shape != (dim_z, 1):
        raise ValueError("z must be convertible to shape ({}, 1)".format(dim_z))
    if ndim == 1:
        z = z[:, 0]
    if ndim == 0:
        z = z[0, 0]
    return z
