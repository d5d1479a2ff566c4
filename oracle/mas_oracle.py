"""ORACLE (test infrastructure, not product): MAS entry points.

`b_mas` mirrors the reference signature `b_mas(b_attn_map, in_lens, out_lens) -> int16 [B,M,L]`
(/root/reference/tts/modules/aligner/mas.py:29-35) but does not mutate its input.  It runs the C restatement
(`mas_oracle.c`); `b_mas_numpy` is an independent row-vectorised numpy restatement used to cross-check the C
code on small cases.
"""
from __future__ import annotations

import ctypes

import numpy as np

from .build_oracle import build

_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_b_mas.restype = None
        _lib.oracle_b_mas.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 3
    return _lib


def b_mas(b_attn_map: np.ndarray, in_lens: np.ndarray, out_lens: np.ndarray, return_path: bool = False):
    x = np.ascontiguousarray(b_attn_map, dtype=np.float32)
    il = np.ascontiguousarray(in_lens, dtype=np.int64)
    ol = np.ascontiguousarray(out_lens, dtype=np.int64)
    B, M, L = x.shape
    assert (il <= L).all() and (ol <= M).all() and (il >= 1).all() and (ol >= 1).all()
    out = np.empty((B, M, L), dtype=np.int16)
    path = np.empty((B, M), dtype=np.int16)
    _load().oracle_b_mas(x.ctypes.data, il.ctypes.data, ol.ctypes.data, out.ctypes.data, path.ctypes.data, B, M, L)
    return (out, path) if return_path else out


def mas_numpy_one(lp: np.ndarray) -> np.ndarray:
    """One utterance, [n, m] fp32 -> path [n] (text index per mel row).  Row-vectorised DP (mas.py:11-24)."""
    n, m = lp.shape
    q = np.empty((n, m), dtype=np.float32)
    q[0, 0] = lp[0, 0]
    q[0, 1:] = -np.inf
    diag = np.zeros((n, m), dtype=bool)
    for i in range(1, n):
        q[i, 0] = q[i - 1, 0] + lp[i, 0]
        d, s = q[i - 1, :-1], q[i - 1, 1:]
        diag[i, 1:] = d >= s
        q[i, 1:] = lp[i, 1:] + np.where(diag[i, 1:], d, s)
    path = np.empty(n, dtype=np.int16)
    j = m - 1
    for i in range(n - 1, -1, -1):
        path[i] = j
        if i > 0 and diag[i, j]:
            j -= 1
    return path


def b_mas_numpy(b_attn_map: np.ndarray, in_lens, out_lens) -> np.ndarray:
    out = np.zeros(b_attn_map.shape, dtype=np.int16)
    for b in range(b_attn_map.shape[0]):
        n, m = int(out_lens[b]), int(in_lens[b])
        p = mas_numpy_one(np.asarray(b_attn_map[b, :n, :m], dtype=np.float32))
        out[b, np.arange(n), p] = 1
    return out
