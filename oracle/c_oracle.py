"""ctypes front end of oracle/_build/liboracle_c.so.  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "liboracle_c.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(_PATH)
        _lib.score_fma.restype = None
        _lib.score_fma.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return _lib


def score_fma(a, bt):
    """a [M,K] x bt [N,K]^T with a k-ordered float32 fmaf chain."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    bt = np.ascontiguousarray(bt, dtype=np.float32)
    M, K = a.shape
    N = bt.shape[0]
    assert bt.shape[1] == K
    c = np.empty((M, N), dtype=np.float32)
    _load().score_fma(a.ctypes.data, bt.ctypes.data, c.ctypes.data, M, N, K)
    return c
