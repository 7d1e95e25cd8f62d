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


def reference_regions(regions, stride):
    """Builds every region out of the reference's own BamAlignment / AlignedRead / ReadBin / Consensus /
    ConsensusHashTable objects, flattens them with the reference's getters in clean_gpu_prep's order
    (apps/fastr/IndelRealigner.cpp:714-752) and scores every (consensus, read) pair with the reference's
    findBestOffset (:857-880).  `regions`: list of dicts with
        bin_reads   [(position, soft_clip, bases str, phred list)]   reads that only widen the bin
        alt_reads   [(position, soft_clip, bases str, phred list)]   the reads that are scored
        candidates  [str]                                             consensus strings, duplicates allowed
    Returns the flattened batch (the layout ctc_fastr_best_offsets takes) plus best_index / best_score."""
    L = ctypes.CDLL(REFERENCE_LIB)
    L.fastr_ref_regions.restype = ctypes.c_int
    L.fastr_ref_regions.argtypes = [ctypes.c_int] + [_vp] * 10 + [ctypes.c_int] + [_vp] * 11
    n_bin, n_alt, pos, length, clip, n_cand, cand_len = [], [], [], [], [], [], []
    bases, phred, cand = bytearray(), bytearray(), bytearray()
    for r in regions:
        reads = list(r["bin_reads"]) + list(r["alt_reads"])
        n_bin.append(len(reads))
        n_alt.append(len(r["alt_reads"]))
        for p, c, b, q in reads:
            pos.append(p)
            length.append(len(b))
            clip.append(c)
            bases += b.encode()
            phred += bytes(q)
        n_cand.append(len(r["candidates"]))
        for c in r["candidates"]:
            cand_len.append(len(c))
            cand += c.encode()
    i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)  # noqa: E731
    u8 = lambda v: np.frombuffer(bytes(v) + b"\0", dtype=np.uint8).copy()  # noqa: E731
    n_bin, n_alt, pos, length, clip, n_cand, cand_len = map(i32, (n_bin, n_alt, pos, length, clip, n_cand, cand_len))
    bases, phred, cand = u8(bases), u8(phred), u8(cand)
    total_reads = int(n_alt.sum())
    total_cons = int(n_cand.sum())
    max_pairs = int((n_alt.astype(np.int64) * n_cand).sum())
    read_seq = np.zeros(max(total_reads, 1) * stride, np.uint8)
    quals = np.zeros(max(total_reads, 1) * stride, np.uint8)
    read_len = np.zeros(max(total_reads, 1), np.int32)
    orig = np.zeros(max(total_reads, 1), np.int32)
    n_consensus = np.zeros(len(regions), np.int32)
    alt_reads = np.zeros(len(regions), np.int32)
    ref_lens = np.zeros(max(total_cons, 1), np.int32)
    ref = np.zeros(max(int(cand_len.sum()), 1), np.uint8)
    totals = np.zeros(4, np.int32)
    best_index = np.zeros(max(max_pairs, 1), np.int32)
    best_score = np.zeros(max(max_pairs, 1), np.int32)
    d = lambda a: a.ctypes.data  # noqa: E731
    rc = L.fastr_ref_regions(len(regions), d(n_bin), d(n_alt), d(pos), d(length), d(clip), d(bases), d(phred),
                             d(n_cand), d(cand_len), d(cand), stride, d(read_seq), d(quals), d(read_len), d(orig),
                             d(n_consensus), d(alt_reads), d(ref_lens), d(ref), d(totals), d(best_index),
                             d(best_score))
    if rc != 0:
        raise RuntimeError("fastr_ref_regions: a read is longer than the stride")
    t_reads, t_ref, t_score, t_cons = (int(x) for x in totals)
    return {"ref": ref[:t_ref], "ref_lens": ref_lens[:t_cons], "n_consensus": n_consensus, "alt_reads": alt_reads,
            "read_seq": read_seq[:t_reads * stride], "quals": quals[:t_reads * stride], "read_len": read_len[:t_reads],
            "original_alignment": orig[:t_reads], "read_stride": stride,
            "best_index": best_index[:t_score], "best_score": best_score[:t_score]}
