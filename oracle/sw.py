"""Pairwise-aligner oracle bindings and the synthetic pair generator.  TEST INFRASTRUCTURE ONLY.

oracle      oracle/sw_ref.c (restatement of apps/fastr/SWPairwiseAlignment.cpp:128-399)
reference   oracle/_ref/libfastr_ref.so when present: the reference's own SWPairwiseAlignment class,
            compiled from /root/reference by oracle/build_ref.sh
"""
import ctypes
import os

import numpy as np

from . import ref as _ref
from .fastr import REFERENCE_LIB, _Rng

_vp = ctypes.c_void_p
_dbl = ctypes.c_double
# apps/fastr/IndelRealigner.cpp:165-168
DEFAULT_WEIGHTS = (30.0, -10.0, -10.0, -2.0)
MAX_OPS = 64


def _bytes(x):
    return np.ascontiguousarray(np.frombuffer(x.encode() if isinstance(x, str) else bytes(x), dtype=np.uint8))


def _call(fn, ref, read, weights, max_ops):
    a, b = _bytes(ref), _bytes(read)
    off = ctypes.c_int(0)
    ty = np.zeros(max_ops, dtype=np.uint8)
    ln = np.zeros(max_ops, dtype=np.int32)
    n = fn(a.ctypes.data, a.size, b.ctypes.data, b.size, *[float(w) for w in weights], ctypes.byref(off),
           ty.ctypes.data, ln.ctypes.data, max_ops)
    if n < 0 or n > max_ops:
        raise RuntimeError("aligner returned %d operations" % n)
    return off.value, "".join("%d%s" % (int(ln[k]), chr(int(ty[k]))) for k in range(n))


_SIG = [_vp, ctypes.c_int, _vp, ctypes.c_int, _dbl, _dbl, _dbl, _dbl, ctypes.POINTER(ctypes.c_int), _vp, _vp, ctypes.c_int]


def align(ref, read, weights=DEFAULT_WEIGHTS, max_ops=MAX_OPS):
    """(alignment offset of the read w.r.t. the reference window, CIGAR string) from the oracle."""
    L = _ref.lib()
    L.ctc_ref_sw_align.restype = ctypes.c_int
    L.ctc_ref_sw_align.argtypes = _SIG
    return _call(L.ctc_ref_sw_align, ref, read, weights, max_ops)


_reference = None


def reference_available():
    return os.path.exists(REFERENCE_LIB)


def reference_align(ref, read, weights=DEFAULT_WEIGHTS, max_ops=MAX_OPS):
    """Same from the reference's own class."""
    global _reference
    if _reference is None:
        _reference = ctypes.CDLL(REFERENCE_LIB)
        _reference.fastr_ref_sw_align.restype = ctypes.c_int
        _reference.fastr_ref_sw_align.argtypes = _SIG
    return _call(_reference.fastr_ref_sw_align, ref, read, weights, max_ops)


def align_batch(batch, weights=DEFAULT_WEIGHTS, max_ops=MAX_OPS, threads=0):
    """Oracle over a whole batch (layout of make_pairs); returns offsets, n_ops, op_type, op_len."""
    L = _ref.lib()
    L.ctc_ref_sw_align_batch.restype = ctypes.c_long
    L.ctc_ref_sw_align_batch.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_int, _vp, ctypes.c_int, _dbl, _dbl, _dbl, _dbl,
                                         _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int]
    n = int(batch["read_len"].size)
    offsets = np.zeros(n, dtype=np.int32)
    n_ops = np.zeros(n, dtype=np.int32)
    ty = np.zeros((n, max_ops), dtype=np.uint8)
    ln = np.zeros((n, max_ops), dtype=np.int32)
    bad = L.ctc_ref_sw_align_batch(batch["ref"].ctypes.data, batch["ref_off"].ctypes.data, batch["ref_len"].ctypes.data,
                                   batch["reads"].ctypes.data, int(batch["read_stride"]), batch["read_len"].ctypes.data,
                                   n, *[float(w) for w in weights], offsets.ctypes.data, n_ops.ctypes.data,
                                   ty.ctypes.data, ln.ctypes.data, max_ops, threads)
    if bad:
        raise RuntimeError("oracle failed on %d pairs" % bad)
    return offsets, n_ops, ty, ln


def cigar_strings(n_ops, ty, ln):
    return ["".join("%d%s" % (int(ln[p, k]), chr(int(ty[p, k]))) for k in range(int(n_ops[p]))) for p in range(len(n_ops))]


def make_pairs(seed=0x5117, n_windows=32, reads_per_window=(8, 40), ref_len=(200, 1200), read_len=(60, 160),
               mismatch=0.02, indel=(1, 12)):
    """Synthetic realignment input: reference windows of random ACGT, reads cut from them with
    a few substitutions and (two reads in three) one planted insertion or deletion; some reads hang
    over either end of their window.  Every read is aligned against its own window."""
    rng = _Rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    refs, ref_off, ref_lens, reads, read_lens = [], [], [], [], []
    pos = 0
    for _ in range(n_windows):
        L = rng.between(ref_len[0], ref_len[1])
        w = acgt[[rng.between(0, 3) for _ in range(L)]]
        refs.append(w)
        for _ in range(rng.between(reads_per_window[0], reads_per_window[1])):
            rl = rng.between(read_len[0], min(read_len[1], L - 20))
            start = rng.between(-10, L - rl + 10)
            body = []
            for x in range(start, start + rl):
                body.append(int(w[x]) if 0 <= x < L else int(acgt[rng.between(0, 3)]))
            kind = rng.between(0, 2)
            size = rng.between(indel[0], indel[1])
            at = rng.between(10, rl - 10)
            if kind == 1:      # insertion into the read
                body[at:at] = [int(acgt[rng.between(0, 3)]) for _ in range(size)]
            elif kind == 2:    # deletion from the read
                del body[at:at + size]
            body = body[:read_len[1] + indel[1]]
            for k in range(len(body)):
                if rng.between(0, 9999) < int(mismatch * 10000):
                    body[k] = int(acgt[rng.between(0, 3)])
            reads.append(np.array(body, dtype=np.uint8))
            read_lens.append(len(body))
            ref_off.append(pos)
            ref_lens.append(L)
        pos += L
    stride = max(read_lens)
    flat = np.zeros((len(reads), stride), dtype=np.uint8)
    for i, r in enumerate(reads):
        flat[i, :r.size] = r
    return {"ref": np.ascontiguousarray(np.concatenate(refs)), "ref_off": np.array(ref_off, dtype=np.int64),
            "ref_len": np.array(ref_lens, dtype=np.int32), "reads": flat, "read_stride": stride,
            "read_len": np.array(read_lens, dtype=np.int32)}
