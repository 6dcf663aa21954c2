"""ctypes front-end of oracle/ops_ref.c (numpy in, numpy out) -- TEST INFRASTRUCTURE ONLY.

Each function restates one reference operator; see the header of ops_ref.c for the
reference file:line each follows.  Parity status: unpinned by the reference's own tests
(it has none); cross-pinned against oracle/closed_form.py and tests/golden/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_ops.so")
_lib = None
_fp = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile ops_ref.c with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "ops_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_build/liboracle_ops.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def correlation_out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2):
    oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib().oracle_correlation_out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2,
                                       ctypes.byref(oc), ctypes.byref(oh), ctypes.byref(ow))
    return oc.value, oh.value, ow.value


def correlation_fwd(in1, in2, pad_size, kernel_size, max_displacement, stride1, stride2):
    in1, p1 = _f32(in1)
    in2, p2 = _f32(in2)
    N, C, H, W = in1.shape
    oc, oh, ow = correlation_out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    out = np.zeros((N, oc, oh, ow), np.float32)
    rc = lib().oracle_correlation_fwd(p1, p2, out.ctypes.data_as(_fp), N, C, H, W, pad_size, kernel_size,
                                      max_displacement, stride1, stride2)
    assert rc == 0, rc
    return out


def correlation_bwd(in1, in2, gout, pad_size, kernel_size, max_displacement, stride1, stride2):
    in1, p1 = _f32(in1)
    in2, p2 = _f32(in2)
    gout, pg = _f32(gout)
    N, C, H, W = in1.shape
    g1 = np.zeros_like(in1)
    g2 = np.zeros_like(in2)
    rc = lib().oracle_correlation_bwd(p1, p2, pg, g1.ctypes.data_as(_fp), g2.ctypes.data_as(_fp), N, C, H, W,
                                      pad_size, kernel_size, max_displacement, stride1, stride2)
    assert rc == 0, rc
    return g1, g2


def resample2d_fwd(img, flow):
    img, pi = _f32(img)
    flow, pf = _f32(flow)
    N, C, H, W = img.shape
    assert flow.shape == (N, 2, H, W)
    out = np.zeros_like(img)
    lib().oracle_resample2d_fwd(pi, pf, out.ctypes.data_as(_fp), N, C, H, W)
    return out


def resample2d_bwd(img, flow, gout):
    img, pi = _f32(img)
    flow, pf = _f32(flow)
    gout, pg = _f32(gout)
    N, C, H, W = img.shape
    gi = np.zeros_like(img)
    gf = np.zeros_like(flow)
    lib().oracle_resample2d_bwd(pi, pf, pg, gi.ctypes.data_as(_fp), gf.ctypes.data_as(_fp), N, C, H, W)
    return gi, gf


def channelnorm_fwd(x):
    x, px = _f32(x)
    N, C, H, W = x.shape
    out = np.zeros((N, 1, H, W), np.float32)
    lib().oracle_channelnorm_fwd(px, out.ctypes.data_as(_fp), N, C, H, W)
    return out


def channelnorm_bwd(x, out, gout):
    x, px = _f32(x)
    out, po = _f32(out)
    gout, pg = _f32(gout)
    N, C, H, W = x.shape
    gin = np.zeros_like(x)
    lib().oracle_channelnorm_bwd(px, po, pg, gin.ctypes.data_as(_fp), N, C, H, W)
    return gin
