"""Host side of the FASTR scoring path: mirrors what clean_gpu_prep / clean_gpu do in the
reference (apps/fastr/IndelRealigner.cpp:671-829) on top of ctc_fastr_best_offsets
(include/ctc_fastr.h).  Device memory is held in torch tensors; no torch operator computes
any part of the result."""
import ctypes

import numpy as np

from . import _lib
from .host import _ptr, _stream, check

_bound = False


def _fn():
    global _bound
    L = _lib.load()
    if not _bound:
        vp = ctypes.c_void_p
        L.ctc_fastr_best_offsets.restype = ctypes.c_int
        L.ctc_fastr_best_offsets.argtypes = [vp, vp, vp, ctypes.c_int, vp, vp, ctypes.c_int, vp, vp, vp, vp, vp, vp]
        _bound = True
    return L.ctc_fastr_best_offsets


class DeviceBatch:
    """A scoring batch resident in HBM.  `batch` uses the reference's flattened layout:
    ref, read_seq, quals (uint8), read_len, original_alignment (int32, one per read),
    n_consensus, alt_reads (int32, one per region), ref_lens (int32, one per consensus), read_stride."""

    @staticmethod
    def validate(batch):
        """The kernel follows the counts and lengths of the batch without bounds checks (it only clamps read
        lengths and original alignments): a batch whose arrays do not hold what its counts promise is
        refused here, before anything is uploaded."""
        n_cons = np.asarray(batch["n_consensus"], dtype=np.int64)
        alt = np.asarray(batch["alt_reads"], dtype=np.int64)
        ref_lens = np.asarray(batch["ref_lens"], dtype=np.int64)
        read_len = np.asarray(batch["read_len"], dtype=np.int64)
        stride = int(batch["read_stride"])
        if stride <= 0:
            raise ValueError("read_stride must be positive")
        if n_cons.shape != alt.shape or (n_cons < 0).any() or (alt < 0).any():
            raise ValueError("n_consensus and alt_reads: one non-negative count per region each")
        n_reads = int(alt.sum())
        if ref_lens.size != int(n_cons.sum()) or (ref_lens < 0).any():
            raise ValueError("ref_lens: one non-negative length per consensus (%d expected, %d given)" % (int(n_cons.sum()), ref_lens.size))
        if read_len.size != n_reads or np.asarray(batch["original_alignment"]).size != n_reads:
            raise ValueError("read_len and original_alignment: one entry per read (%d reads)" % n_reads)
        if ((read_len < 0) | (read_len > stride)).any():
            raise ValueError("a read length lies outside 0..read_stride")
        for name, need in (("ref", int(ref_lens.sum())), ("read_seq", n_reads * stride), ("quals", n_reads * stride)):
            a = np.asarray(batch[name])
            if a.dtype != np.uint8 or a.size < need:
                raise ValueError("%s: %d bytes (uint8) needed, %d %s given" % (name, need, a.size, a.dtype))

    def __init__(self, batch, device):
        import torch
        self.validate(batch)
        self.device = device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        self.ref = up(batch["ref"])
        self.read_seq = up(batch["read_seq"])
        self.quals = up(batch["quals"])
        self.read_len = up(batch["read_len"].astype(np.int32))
        self.original_alignment = up(batch["original_alignment"].astype(np.int32))
        self.n_consensus = np.ascontiguousarray(batch["n_consensus"], dtype=np.int32)
        self.alt_reads = np.ascontiguousarray(batch["alt_reads"], dtype=np.int32)
        self.ref_lens = np.ascontiguousarray(batch["ref_lens"], dtype=np.int32)
        self.read_stride = int(batch["read_stride"])
        self.n_pairs = int(np.dot(self.n_consensus.astype(np.int64), self.alt_reads.astype(np.int64)))
        self.best_score = torch.empty(max(self.n_pairs, 1), dtype=torch.int32, device=device)
        self.best_index = torch.empty(max(self.n_pairs, 1), dtype=torch.int32, device=device)

    def run(self, stream=None):
        """Enqueue the scoring of the whole batch; results land in best_score / best_index."""
        check(_fn()(_ptr(self.ref), _ptr(self.read_seq), _ptr(self.quals), self.read_stride, _ptr(self.read_len),
                    _ptr(self.original_alignment), len(self.n_consensus), self.n_consensus.ctypes.data,
                    self.alt_reads.ctypes.data, self.ref_lens.ctypes.data, _ptr(self.best_score),
                    _ptr(self.best_index), _stream(stream)))
        return self

    def results(self):
        return (self.best_score[:self.n_pairs].cpu().numpy(), self.best_index[:self.n_pairs].cpu().numpy())


def best_offsets(batch, device="cuda:0"):
    """(best_score, best_index) numpy arrays for a host batch: upload, score on the GPU, download."""
    return DeviceBatch(batch, device).run().results()


def _consensus_key(text):
    """The key the reference's consensus table deduplicates on: the 31-polynomial string hash in signed
    32-bit arithmetic over signed bytes (apps/fastr/Consensus.cpp:83-94) pushed through the table's bit
    mixer (apps/fastr/ConsensusHashTable.cpp:148-153).  Two strings with one key count as one consensus
    (:225-237 compares the keys only)."""
    h = 1
    for c in text:
        h = (31 * h + (c - 256 if c > 127 else c)) & 0xFFFFFFFF
    h ^= (h >> 20) ^ (h >> 12)
    h ^= (h >> 7) ^ (h >> 4)
    return h


def flatten_regions(regions, read_stride, reference_padding=0):
    """Flattens realignment regions into the batch ctc_fastr_best_offsets takes, as clean_gpu_prep does
    (apps/fastr/IndelRealigner.cpp:714-752).  Each region is a dict:
        bin_reads   [(position, soft_clip, bases, phred)]  reads of the bin that are not scored
        alt_reads   [(position, soft_clip, bases, phred)]  the reads that are scored, in order
        candidates  [bases]                                 consensus strings in the order they were found
    The bin starts at the FIRST read's position (apps/fastr/ReadBin.cpp:82-84), moved left by
    `reference_padding` (not below 0) once the bin's reference was fetched (:99-107; the reference pads
    by 30).  A read contributes its bases and qualities after the leading soft clip, which is what
    AlignedRead::getUnclippedBases keeps (apps/fastr/AlignedRead.cpp:61-124).  Candidates whose table
    keys collide are dropped after the first (see _consensus_key); the rest keep insertion order
    (ConsensusHashTable.cpp:113-119)."""
    enc = lambda b: b.encode() if isinstance(b, str) else bytes(b)  # noqa: E731
    seq, quals, read_len, orig, n_consensus, alt_reads, ref_lens, ref = [], [], [], [], [], [], [], bytearray()
    for region in regions:
        reads = list(region["bin_reads"]) + list(region["alt_reads"])
        if not reads:
            raise ValueError("a region needs at least one read to have a start")
        leftmost = max(reads[0][0] - reference_padding, 0) if reference_padding else reads[0][0]
        alt_reads.append(len(region["alt_reads"]))
        for position, clip, bases, phred in region["alt_reads"]:
            b, q = enc(bases)[clip:], bytes(phred)[clip:]
            if len(b) > read_stride:
                raise ValueError("read of %d bases does not fit the stride %d" % (len(b), read_stride))
            row = np.zeros(read_stride, np.uint8)
            row[:len(b)] = np.frombuffer(b, np.uint8)
            seq.append(row)
            row = np.zeros(read_stride, np.uint8)
            row[:len(q)] = np.frombuffer(q, np.uint8)
            quals.append(row)
            read_len.append(len(b))
            orig.append(position - leftmost)
        seen, kept = set(), 0
        for c in region["candidates"]:
            c = enc(c)
            key = _consensus_key(c)
            if key in seen:
                continue
            seen.add(key)
            ref += c
            ref_lens.append(len(c))
            kept += 1
        n_consensus.append(kept)
    cat = lambda rows: np.concatenate(rows) if rows else np.zeros(0, np.uint8)  # noqa: E731
    return {"ref": np.frombuffer(bytes(ref), np.uint8).copy(), "ref_lens": np.array(ref_lens, np.int32),
            "n_consensus": np.array(n_consensus, np.int32), "alt_reads": np.array(alt_reads, np.int32),
            "read_seq": cat(seq), "quals": cat(quals), "read_len": np.array(read_len, np.int32),
            "original_alignment": np.array(orig, np.int32), "read_stride": int(read_stride)}


# ---- pairwise alignment (apps/fastr/SWPairwiseAlignment.cpp) ----------------------------------
DEFAULT_SW_WEIGHTS = (30.0, -10.0, -10.0, -2.0)  # apps/fastr/IndelRealigner.cpp:165-168
_sw_bound = False


def _sw_fns():
    global _sw_bound
    L = _lib.load()
    if not _sw_bound:
        vp, dbl, ci = ctypes.c_void_p, ctypes.c_double, ctypes.c_int
        L.ctc_fastr_sw_workspace_bytes.restype = ctypes.c_size_t
        L.ctc_fastr_sw_workspace_bytes.argtypes = [ci, ci, ci]
        L.ctc_fastr_sw_align.restype = ci
        L.ctc_fastr_sw_align.argtypes = [vp, vp, vp, vp, ci, vp, ci, dbl, dbl, dbl, dbl, ci, ci, vp, vp, vp, vp, ci,
                                         vp, ctypes.c_size_t, vp]
        _sw_bound = True
    return L.ctc_fastr_sw_workspace_bytes, L.ctc_fastr_sw_align


class DeviceAlignmentBatch:
    """Reads and their reference windows resident in HBM.  `batch`: ref (uint8, windows back to
    back), ref_off (int64) and ref_len (int32) per pair, reads (uint8 [pairs][read_stride]),
    read_len (int32), read_stride."""

    @staticmethod
    def validate(batch):
        """Every window must lie inside `ref` and every read inside its slot: the kernel checks lengths against
        the batch maxima only."""
        ref_off = np.asarray(batch["ref_off"], dtype=np.int64)
        ref_len = np.asarray(batch["ref_len"], dtype=np.int64)
        read_len = np.asarray(batch["read_len"], dtype=np.int64)
        stride = int(batch["read_stride"])
        n = read_len.size
        if stride <= 0 or ref_off.size != n or ref_len.size != n:
            raise ValueError("ref_off, ref_len and read_len: one entry per pair; read_stride positive")
        ref, reads = np.asarray(batch["ref"]), np.asarray(batch["reads"])
        if ref.dtype != np.uint8 or reads.dtype != np.uint8 or reads.size < n * stride:
            raise ValueError("ref and reads must be uint8, reads holding read_stride bytes per pair")
        if n and ((ref_off < 0) | (ref_len < 0) | (ref_off + ref_len > ref.size)).any():
            raise ValueError("a reference window lies outside ref")
        if n and ((read_len < 0) | (read_len > stride)).any():
            raise ValueError("a read length lies outside 0..read_stride")
        if n and ref_len.max() > 32767:
            raise ValueError("reference windows are limited to 32767 bases")

    def __init__(self, batch, device, max_ops=64, workspace_pairs=None):
        import torch
        self.validate(batch)
        self.device = device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        self.ref = up(batch["ref"])
        self.ref_off = up(batch["ref_off"].astype(np.int64))
        self.ref_len = up(batch["ref_len"].astype(np.int32))
        self.reads = up(batch["reads"])
        self.read_len = up(batch["read_len"].astype(np.int32))
        self.read_stride = int(batch["read_stride"])
        self.n_pairs = int(batch["read_len"].size)
        self.max_ref_len = int(batch["ref_len"].max()) if self.n_pairs else 1
        self.max_read_len = int(batch["read_len"].max()) if self.n_pairs else 1
        self.max_ops = max_ops
        n = max(self.n_pairs, 1)
        self.offset = torch.empty(n, dtype=torch.int32, device=device)
        self.n_ops = torch.empty(n, dtype=torch.int32, device=device)
        self.op_type = torch.empty(n * max_ops, dtype=torch.uint8, device=device)
        self.op_len = torch.empty(n * max_ops, dtype=torch.int32, device=device)
        ws_fn, _ = _sw_fns()
        pairs = self.n_pairs if workspace_pairs is None else min(workspace_pairs, self.n_pairs)
        self.workspace_bytes = int(ws_fn(max(pairs, 1), self.max_ref_len, self.max_read_len))
        self.workspace = torch.empty(max(self.workspace_bytes, 1), dtype=torch.uint8, device=device)

    def run(self, weights=DEFAULT_SW_WEIGHTS, stream=None):
        _, fn = _sw_fns()
        check(fn(_ptr(self.ref), _ptr(self.ref_off), _ptr(self.ref_len), _ptr(self.reads), self.read_stride,
                 _ptr(self.read_len), self.n_pairs, *[float(w) for w in weights], self.max_ref_len, self.max_read_len,
                 _ptr(self.offset), _ptr(self.n_ops), _ptr(self.op_type), _ptr(self.op_len), self.max_ops,
                 _ptr(self.workspace), self.workspace_bytes, _stream(stream)))
        return self

    def results(self):
        n = self.n_pairs
        return (self.offset[:n].cpu().numpy(), self.n_ops[:n].cpu().numpy(),
                self.op_type[:n * self.max_ops].cpu().numpy().reshape(n, self.max_ops),
                self.op_len[:n * self.max_ops].cpu().numpy().reshape(n, self.max_ops))


def sw_align(batch, device="cuda:0", weights=DEFAULT_SW_WEIGHTS, max_ops=64):
    """(offset, n_ops, op_type, op_len) numpy arrays for a host batch of (window, read) pairs."""
    return DeviceAlignmentBatch(batch, device, max_ops).run(weights).results()
