"""fp64 numpy restatement of the MARS CNN forward pass -- the keypoint oracle.

TEST INFRASTRUCTURE ONLY.  Restates `define_CNN_3D` / `define_CNN`
(/root/reference/src/train.py:71-106, 33-68) at inference with Keras 2.15
semantics (keras/tensorflow are absent from the image and no trained weights
exist in the reference repo -> "parity unpinned"; weights are seeded random):

  Conv3D/Conv2D: channels-last, kernel (kd,kh,kw,in,out) / (kh,kw,in,out), stride 1,
                 padding "same" (zero pad 1), bias, ReLU
  Dropout:       identity at inference
  BatchNormalization(momentum=.95): y = gamma*(x-mean)/sqrt(var+1e-3)+beta on the last axis
  Flatten:       row-major over (d,h,w,c)
  Dense:         x @ W + b, W (in,out)

Weight dict keys: conv1_w conv1_b conv2_w conv2_b bn1_gamma bn1_beta bn1_mean bn1_var
dense1_w dense1_b bn2_gamma bn2_beta bn2_mean bn2_var dense2_w dense2_b.
"""
import numpy as np

BN_EPS = 1e-3


def _conv_same(x, w, b):
    """x (B, *spatial, Cin) ; w (*k, Cin, Cout) with k = 3 in every spatial dim."""
    nd = w.ndim - 2
    pad = [(0, 0)] + [(1, 1)] * nd + [(0, 0)]
    xp = np.pad(x, pad)
    out = np.zeros(x.shape[:-1] + (w.shape[-1],), dtype=np.float64)
    sp = x.shape[1:-1]
    for off in np.ndindex(*([3] * nd)):
        sl = (slice(None),) + tuple(slice(o, o + s) for o, s in zip(off, sp)) + (slice(None),)
        out += np.tensordot(xp[sl], w[off], axes=([-1], [0]))
    return out + b


def _bn(x, g, be, m, v):
    return g * (x - m) / np.sqrt(v + BN_EPS) + be


_W64 = {}   # id(weight dict) -> its float64 copy (9.5 M parameters: converting them per call dominated small batches)


def _weights64(w):
    ent = _W64.get(id(w))
    if ent is None or ent[0] is not w:
        if len(_W64) > 8:
            _W64.clear()
        ent = (w, {k: np.asarray(v, dtype=np.float64) for k, v in w.items()})
        _W64[id(w)] = ent
    return ent[1]


def _blas_threads():
    """The BLAS pool sized to the CPUs this process may really use (a container on a 256-thread host under a 16-CPU quota:
    256 BLAS threads only throttle each other)."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        import contextlib
        return contextlib.nullcontext()
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            n = min(n, max(1, int(float(a) / float(b) + 0.5)))
    except (AttributeError, OSError, ValueError):
        pass
    return threadpool_limits(limits=max(1, n))


def mars_forward_np(w, x):
    """x: (B,3,8,8,5) or (B,8,8,5) -> (B,57), float64."""
    with _blas_threads():
        return _forward(_weights64(w), x)


def _forward(w, x):
    h = np.asarray(x, dtype=np.float64)
    h = np.maximum(_conv_same(h, w["conv1_w"], w["conv1_b"]), 0.0)
    h = np.maximum(_conv_same(h, w["conv2_w"], w["conv2_b"]), 0.0)
    h = _bn(h, w["bn1_gamma"], w["bn1_beta"], w["bn1_mean"], w["bn1_var"])
    h = h.reshape(h.shape[0], -1)
    h = np.maximum(h @ w["dense1_w"] + w["dense1_b"], 0.0)
    h = _bn(h, w["bn2_gamma"], w["bn2_beta"], w["bn2_mean"], w["bn2_var"])
    return h @ w["dense2_w"] + w["dense2_b"]
