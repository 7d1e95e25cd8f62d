"""FASTR scoring oracle bindings and the synthetic batch generator.  TEST INFRASTRUCTURE ONLY.

oracle          oracle/fastr_ref.c (restatement of apps/fastr/IndelRealigner.cpp:264-295,502-542)
reference       oracle/_ref/libfastr_ref.so when present: the reference's own functions, compiled
                from /root/reference by oracle/build_ref.sh
"""
import ctypes
import os

import numpy as np

from . import ref as _ref

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_LIB = os.path.join(_HERE, "_ref", "libfastr_ref.so")
_vp = ctypes.c_void_p


def _oracle():
    L = _ref.lib()
    L.ctc_ref_fastr_find_best_offset.restype = None
    L.ctc_ref_fastr_find_best_offset.argtypes = [_vp, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int,
                                                 ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.ctc_ref_fastr_best_offsets.restype = ctypes.c_long
    L.ctc_ref_fastr_best_offsets.argtypes = [_vp, _vp, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, _vp, _vp,
                                             _vp, _vp, ctypes.c_int]
    return L


_reference = None


def reference_available():
    return os.path.exists(REFERENCE_LIB)


def reference_find_best_offset(ref, read, quals, original_alignment):
    """(best_index, best_score) from the reference's own findBestOffset."""
    global _reference
    if _reference is None:
        _reference = ctypes.CDLL(REFERENCE_LIB)
        _reference.fastr_ref_find_best_offset.argtypes = [_vp, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int,
                                                          ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    i, s = ctypes.c_int(), ctypes.c_int()
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    read = np.ascontiguousarray(read, dtype=np.uint8)
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    _reference.fastr_ref_find_best_offset(ref.ctypes.data, len(ref), read.ctypes.data, quals.ctypes.data, len(read),
                                          int(original_alignment), ctypes.byref(i), ctypes.byref(s))
    return i.value, s.value


def find_best_offset(ref, read, quals, original_alignment):
    i, s = ctypes.c_int(), ctypes.c_int()
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    read = np.ascontiguousarray(read, dtype=np.uint8)
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    _oracle().ctc_ref_fastr_find_best_offset(ref.ctypes.data, len(ref), read.ctypes.data, quals.ctypes.data,
                                             len(read), int(original_alignment), ctypes.byref(i), ctypes.byref(s))
    return i.value, s.value


def best_offsets(batch, threads=0):
    """(best_score, best_index) arrays for a whole batch (dict from make_batch)."""
    n = int(np.dot(batch["n_consensus"], batch["alt_reads"]))
    score = np.zeros(n, dtype=np.int32)
    index = np.zeros(n, dtype=np.int32)
    got = _oracle().ctc_ref_fastr_best_offsets(
        batch["ref"].ctypes.data, batch["read_seq"].ctypes.data, batch["quals"].ctypes.data, int(batch["read_stride"]),
        batch["read_len"].ctypes.data, batch["original_alignment"].ctypes.data, len(batch["n_consensus"]),
        batch["n_consensus"].ctypes.data, batch["alt_reads"].ctypes.data, batch["ref_lens"].ctypes.data,
        score.ctypes.data, index.ctypes.data, int(threads))
    assert got == n
    return score, index


class _Rng:
    """Integer stream from splitmix64(seed + counter): same on every platform."""

    def __init__(self, seed):
        self.seed, self.n = seed, 0

    def below(self, k):
        v = _ref.splitmix64(self.seed + self.n) % k
        self.n += 1
        return int(v)

    def between(self, lo, hi):
        return lo + self.below(hi - lo + 1)


def make_batch(seed=0xFA57, n_regions=32, max_consensus=8, reads=(20, 200), read_len=100, ref_len=(200, 1200),
               mismatch_pct=2, read_stride=None, with_n=True):
    """Synthetic scoring batch in the reference's flattened layout: per region one random ACGT
    reference window; consensus 0 is the window, the others carry one planted indel; reads are
    copies of a random consensus at a random offset with `mismatch_pct` % substitutions, quals in
    2..40, original alignment within +-5 of the true offset."""
    rng = _Rng(seed)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    read_stride = read_stride or read_len
    refs, ref_lens, n_cons, alt_reads = [], [], [], []
    read_seq, quals, read_lens, origs = [], [], [], []
    for _ in range(n_regions):
        L = rng.between(*ref_len)
        window = bases[np.array([rng.below(4) for _ in range(L)], dtype=np.int64)]
        if with_n and L > 50:
            window = window.copy()
            window[rng.below(L)] = ord("N")
        cons = [window]
        for _c in range(rng.between(1, max_consensus) - 1):
            at = rng.between(10, L - 20)
            size = rng.between(1, 8)
            if rng.below(2):
                ins = bases[np.array([rng.below(4) for _ in range(size)], dtype=np.int64)]
                cons.append(np.concatenate([window[:at], ins, window[at:]]))
            else:
                cons.append(np.concatenate([window[:at], window[at + size:]]))
        nr = rng.between(*reads)
        for _r in range(nr):
            src = cons[rng.below(len(cons))]
            rl = min(read_len, len(src))
            off = rng.between(0, len(src) - rl)
            seq = src[off:off + rl].copy()
            for i in range(rl):
                if rng.below(100) < mismatch_pct:
                    seq[i] = bases[rng.below(4)]
            q = np.array([rng.between(2, 40) for _ in range(rl)], dtype=np.uint8)
            slot_s = np.zeros(read_stride, dtype=np.uint8)
            slot_q = np.zeros(read_stride, dtype=np.uint8)
            slot_s[:rl], slot_q[:rl] = seq, q
            read_seq.append(slot_s)
            quals.append(slot_q)
            read_lens.append(rl)
            shortest = min(len(c) for c in cons)
            origs.append(max(0, min(off + rng.between(-5, 5), shortest - rl)))
        refs.extend(cons)
        ref_lens.extend(len(c) for c in cons)
        n_cons.append(len(cons))
        alt_reads.append(nr)
    return {
        "ref": np.ascontiguousarray(np.concatenate(refs)), "ref_lens": np.array(ref_lens, dtype=np.int32),
        "n_consensus": np.array(n_cons, dtype=np.int32), "alt_reads": np.array(alt_reads, dtype=np.int32),
        "read_seq": np.ascontiguousarray(np.concatenate(read_seq)), "quals": np.ascontiguousarray(np.concatenate(quals)),
        "read_len": np.array(read_lens, dtype=np.int32), "original_alignment": np.array(origs, dtype=np.int32),
        "read_stride": read_stride,
    }


def pair_iter(batch):
    """Yields (ref_bytes, read_bytes, quals, original_alignment) for every (read, consensus) pair in result order."""
    ref_off = read_off = cons = 0
    rs = batch["read_stride"]
    for t in range(len(batch["n_consensus"])):
        for _k in range(batch["n_consensus"][t]):
            L = batch["ref_lens"][cons]
            r = batch["ref"][ref_off:ref_off + L]
            for b in range(batch["alt_reads"][t]):
                g = read_off + b
                n = batch["read_len"][g]
                yield r, batch["read_seq"][g * rs:g * rs + n], batch["quals"][g * rs:g * rs + n], \
                    int(batch["original_alignment"][g])
            ref_off += L
            cons += 1
        read_off += batch["alt_reads"][t]
