"""Minimal stand-in for filterpy==1.4.5 (absent from this image; pinned at
/root/reference/requirements.txt).  TEST INFRASTRUCTURE ONLY: it exists so that
the read-only reference (`/root/reference/src/{constants,Tracking}.py`) can be
imported in the build container to generate golden vectors.  It restates the
published filterpy 1.4.5 formulas used by the reference call sites
(Tracking.py:5,74-97,381-384,393; constants.py:2,210-215,239-243):

  predict:  x = F x ;  P = alpha^2 F P F^T + Q
  update:   y = z - H x ; S = H P H^T + R ; K = P H^T S^-1 ;
            x = x + K y ; P = (I-KH) P (I-KH)^T + K R K^T      (Joseph form)
  Q_discrete_white_noise(dim=3, dt, var)

"parity unpinned": filterpy itself is not available, so this shim cannot be
checked against the real package here.
"""
__version__ = "1.4.5-shim"
