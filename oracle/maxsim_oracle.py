"""ctypes front-end of oracle/maxsim_oracle.c (TEST INFRASTRUCTURE ONLY).

The C file is the restatement of the reference arithmetic
(colpali_engine/utils/processing_utils.py:170-186,
 colpali_engine/loss/late_interaction_losses.py:297-298); this module only
marshals numpy arrays into it and packs ragged python inputs into the
(blob, offsets, clamp0) layout described there.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmaxsim_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle in place (gcc is part of the image)."""
    src = os.path.join(_HERE, "maxsim_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        f32p = ctypes.POINTER(ctypes.c_float)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        common = [f32p, ctypes.c_int, ctypes.c_int, f32p, i32p, u8p, ctypes.c_int, ctypes.c_int, f32p]
        for name in ("oracle_maxsim_f32", "oracle_maxsim_bf16ref"):
            fn = getattr(lib, name)
            fn.argtypes = common
            fn.restype = ctypes.c_int
        lib.oracle_maxsim_argmax_f32.argtypes = common + [i32p]
        lib.oracle_maxsim_argmax_f32.restype = ctypes.c_int
        lib.oracle_num_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def num_threads() -> int:
    return int(_load().oracle_num_threads())


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    """uint16 bfloat16 bit patterns -> float32 (exact)."""
    return (bits.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bfloat16 bit patterns, round-to-nearest-even (torch semantics)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def pack_docs(docs: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """list of [L_i, dim] float32 arrays -> (blob [sum L, dim], offsets int32 [n+1])."""
    lens = [int(d.shape[0]) for d in docs]
    off = np.zeros(len(docs) + 1, dtype=np.int32)
    np.cumsum(lens, out=off[1:])
    dim = docs[0].shape[1]
    blob = np.zeros((max(int(off[-1]), 1), dim), dtype=np.float32)
    for d, o in zip(docs, off[:-1]):
        blob[o : o + d.shape[0]] = d
    return blob, off


def block_clamp0(lens: Sequence[int], batch_size: int) -> np.ndarray:
    """Which documents the reference zero-pads inside their passage block.

    processing_utils.py:175-178: passages j..j+batch_size-1 are padded with
    zero rows to the longest passage of that block, so a passage shorter than
    its block's maximum gains similarity-0 candidates in every per-token max.
    """
    lens = np.asarray(lens, dtype=np.int64)
    out = np.zeros(len(lens), dtype=np.uint8)
    for j in range(0, len(lens), batch_size):
        blk = lens[j : j + batch_size]
        out[j : j + batch_size] = (blk < blk.max()).astype(np.uint8)
    return out


def pad_queries(qs: Sequence[np.ndarray]) -> np.ndarray:
    """list of [L_i, dim] -> [n, Lmax, dim] zero padded.

    processing_utils.py:172 pads per 128-query block; zero query rows add
    exactly 0 to a score (every similarity is 0, so is the max), therefore
    padding all queries to the global maximum gives identical scores.
    """
    lmax = max(int(q.shape[0]) for q in qs)
    out = np.zeros((len(qs), lmax, qs[0].shape[1]), dtype=np.float32)
    for i, q in enumerate(qs):
        out[i, : q.shape[0]] = q
    return out


def _run(fn_name: str, Q: np.ndarray, blob: np.ndarray, off: np.ndarray,
         clamp0: Optional[np.ndarray], want_argmax: bool = False):
    lib = _load()
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    off = np.ascontiguousarray(off, dtype=np.int32)
    n_q, Lq, dim = Q.shape
    n_d = off.shape[0] - 1
    assert blob.shape[1] == dim
    scores = np.empty((n_q, n_d), dtype=np.float32)
    f32p = ctypes.POINTER(ctypes.c_float)
    c0 = None
    if clamp0 is not None:
        clamp0 = np.ascontiguousarray(clamp0, dtype=np.uint8)
        c0 = clamp0.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    args = [Q.ctypes.data_as(f32p), n_q, Lq, blob.ctypes.data_as(f32p),
            off.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), c0, n_d, dim,
            scores.ctypes.data_as(f32p)]
    if want_argmax:
        am = np.empty((n_q, n_d, Lq), dtype=np.int32)
        rc = lib.oracle_maxsim_argmax_f32(*args, am.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        assert rc == 0
        return scores, am
    rc = getattr(lib, fn_name)(*args)
    assert rc == 0
    return scores


def maxsim_f32(Q, blob, off, clamp0=None) -> np.ndarray:
    """Truth tier: fp32 inputs, see oracle_maxsim_f32 in the C file."""
    return _run("oracle_maxsim_f32", Q, blob, off, clamp0)


def maxsim_bf16ref(Q, blob, off, clamp0=None) -> np.ndarray:
    """Literal tier: bf16-input CPU semantics of the reference."""
    return _run("oracle_maxsim_bf16ref", Q, blob, off, clamp0)


def maxsim_argmax_f32(Q, blob, off, clamp0=None):
    return _run("", Q, blob, off, clamp0, want_argmax=True)


def score_multi_vector(qs: List[np.ndarray], ps: List[np.ndarray], batch_size: int = 128,
                       mode: str = "f32") -> np.ndarray:
    """Whole-function restatement of processing_utils.py:132-187 on float32 arrays.

    `mode="f32"` is the truth tier, `mode="bf16ref"` the literal tier (inputs
    must then hold bf16-representable values).
    """
    if len(qs) == 0:
        raise ValueError("No queries provided")
    if len(ps) == 0:
        raise ValueError("No passages provided")
    Q = pad_queries(qs)
    blob, off = pack_docs(ps)
    clamp0 = block_clamp0([p.shape[0] for p in ps], batch_size)
    fn = maxsim_f32 if mode == "f32" else maxsim_bf16ref
    return fn(Q, blob, off, clamp0)
