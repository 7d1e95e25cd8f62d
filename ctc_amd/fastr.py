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

    def __init__(self, batch, device):
        import torch
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
