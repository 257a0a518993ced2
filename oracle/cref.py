"""ctypes loader for oracle/libvmref.so (the plain-C half of the CPU ORACLE; test infrastructure only)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_DT = {"f16": 0, "bf16": 1, "f32": 2}


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return os.path.join(_HERE, "libvmref.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvmref.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.vmref_cosine_topk.restype = ctypes.c_int
        _LIB.vmref_cosine_matrix.restype = ctypes.c_int
    return _LIB


def _as_bits(x: np.ndarray, dtype: str) -> np.ndarray:
    """Accept raw uint16 bit patterns (f16/bf16) or float32 (dtype='f32')."""
    if dtype == "f32":
        return np.ascontiguousarray(x, dtype=np.float32)
    if x.dtype == np.float16:
        assert dtype == "f16"
        return np.ascontiguousarray(x).view(np.uint16)
    assert x.dtype == np.uint16, x.dtype
    return np.ascontiguousarray(x)


def cosine_topk(queries, memory, k, dtype="f16", score_mode=0, min_score=None):
    """-> (rows[Q,k] int64 -1-padded, scores[Q,k] fp64) with the reference's stable descending order."""
    q, m = _as_bits(queries, dtype), _as_bits(memory, dtype)
    Q, D = q.shape
    M = m.shape[0]
    rows = np.empty((Q, k), np.int64)
    scores = np.empty((Q, k), np.float64)
    rc = lib().vmref_cosine_topk(
        q.ctypes.data_as(ctypes.c_void_p), m.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(_DT[dtype]),
        ctypes.c_int(Q), ctypes.c_int64(M), ctypes.c_int(D), ctypes.c_int(k), ctypes.c_int(score_mode),
        ctypes.c_int(0 if min_score is None else 1), ctypes.c_double(0.0 if min_score is None else min_score),
        rows.ctypes.data_as(ctypes.c_void_p), scores.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return rows, scores


def cosine_matrix(queries, memory, dtype="f16"):
    q, m = _as_bits(queries, dtype), _as_bits(memory, dtype)
    Q, D = q.shape
    M = m.shape[0]
    out = np.empty((Q, M), np.float64)
    rc = lib().vmref_cosine_matrix(
        q.ctypes.data_as(ctypes.c_void_p), m.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(_DT[dtype]),
        ctypes.c_int(Q), ctypes.c_int64(M), ctypes.c_int(D), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return out
