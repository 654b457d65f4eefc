"""ORACLE -- test infrastructure.  ctypes wrapper of oracle/ref_ops.c (torch-free C restatement of the ops)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libref_ops.so")


def _lib():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "ref_ops.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return ctypes.CDLL(_SO)


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def gated_conv(x, w, b, stride=1, rate=1, act="elu", upsample=False):
    x, w, b = (np.ascontiguousarray(a, np.float32) for a in (x, w, b))
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    pad = rate * (k - 1) // 2
    Hu, Wu = (2 * H, 2 * W) if upsample else (H, W)
    Ho = (Hu + 2 * pad - rate * (k - 1) - 1) // stride + 1
    Wo = (Wu + 2 * pad - rate * (k - 1) - 1) // stride + 1
    raw = act is None or Cout == 3
    y = np.empty((B, Cout if raw else Cout // 2, Ho, Wo), np.float32)
    _lib().ref_gated_conv(_fp(x), _fp(w), _fp(b), _fp(y), B, Cin, H, W, Cout, k, stride, rate,
                          {"elu": 0, "relu": 1, None: 2}[act], int(upsample))
    return y


def attention(x, mask_full):
    x, mask_full = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(mask_full, np.float32)
    B, C, h, w = x.shape
    hs, ws = (h - 4) // 2 + 1, (w - 4) // 2 + 1
    out = np.empty_like(x)
    sim = np.empty((B, hs * ws, hs, ws), np.float32)
    _lib().ref_attention(_fp(x), _fp(mask_full), _fp(out), _fp(sim), B, C, h, w)
    return out, sim
