"""Pairwise aligner (apps/fastr/SWPairwiseAlignment.cpp): oracle against vectors produced by the
reference's own class, and the HIP kernel against the oracle (bit-exact: offsets and CIGARs)."""
import json
import os
import re

import numpy as np
import pytest

from oracle import sw as osw

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "sw_golden.json")


def _golden():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


def test_oracle_matches_reference_vectors():
    cases = _golden()
    assert len(cases) > 60
    for c in cases:
        assert osw.align(c["ref"], c["read"], c["weights"]) == (c["offset"], c["cigar"]), (c["ref"][:30], c["read"][:30])


def test_cigar_accounts_for_every_read_base():
    for c in _golden():
        ops = [(int(n), t) for n, t in re.findall(r"(\d+)([MIDS])", c["cigar"])]
        assert sum(n for n, t in ops if t in "MIS") == len(c["read"])
        # (a zero-length "0M" can close the CIGAR when the back-track starts in a gap: the reference
        #  emits its initial empty match segment, SWPairwiseAlignment.cpp:318,361-365)


@pytest.mark.skipif(not osw.reference_available(), reason="reference library not built (oracle/build_ref.sh)")
def test_oracle_matches_reference_live_on_a_fresh_batch():
    b = osw.make_pairs(seed=0xBEEF, n_windows=4, reads_per_window=(5, 8), ref_len=(120, 500))
    for p in range(b["read_len"].size):
        ref = bytes(b["ref"][b["ref_off"][p]:b["ref_off"][p] + b["ref_len"][p]])
        read = bytes(b["reads"][p, :b["read_len"][p]])
        assert osw.align(ref, read) == osw.reference_align(ref, read)


def test_batch_form_equals_single_calls():
    b = osw.make_pairs(seed=7, n_windows=3, reads_per_window=(4, 6), ref_len=(100, 300))
    off, n_ops, ty, ln = osw.align_batch(b, threads=2)
    strings = osw.cigar_strings(n_ops, ty, ln)
    for p in range(b["read_len"].size):
        ref = bytes(b["ref"][b["ref_off"][p]:b["ref_off"][p] + b["ref_len"][p]])
        read = bytes(b["reads"][p, :b["read_len"][p]])
        assert (int(off[p]), strings[p]) == osw.align(ref, read)


# ---- HIP kernel through the C ABI ------------------------------------------------------------
def _compare(batch, weights=osw.DEFAULT_WEIGHTS, **kw):
    import torch  # noqa: F401
    from ctc_amd import fastr
    d = fastr.DeviceAlignmentBatch(batch, "cuda:0", **kw).run(weights)
    off, n_ops, ty, ln = d.results()
    w_off, w_n, w_ty, w_ln = osw.align_batch(batch, weights, threads=8)
    assert np.array_equal(n_ops, w_n)
    assert np.array_equal(off, w_off)
    assert osw.cigar_strings(n_ops, ty, ln) == osw.cigar_strings(w_n, w_ty, w_ln)
    return d


def _batch_from_cases(cases):
    refs, ref_off, ref_len, read_len = [], [], [], []
    pos = 0
    for c in cases:
        r = np.frombuffer(c["ref"].encode(), dtype=np.uint8)
        refs.append(r)
        ref_off.append(pos)
        ref_len.append(r.size)
        read_len.append(len(c["read"]))
        pos += r.size
    stride = max(read_len)
    reads = np.zeros((len(cases), stride), dtype=np.uint8)
    for i, c in enumerate(cases):
        reads[i, :len(c["read"])] = np.frombuffer(c["read"].encode(), dtype=np.uint8)
    return {"ref": np.concatenate(refs), "ref_off": np.array(ref_off, dtype=np.int64),
            "ref_len": np.array(ref_len, dtype=np.int32), "reads": reads, "read_stride": stride,
            "read_len": np.array(read_len, dtype=np.int32)}


@pytest.mark.gpu
def test_gpu_reproduces_reference_vectors(cuda):
    from ctc_amd import fastr
    for weights in (osw.DEFAULT_WEIGHTS, (1.0, -1.0 / 3.0, -1.0 - 1.0 / 3.0, -1.0 / 3.0)):
        cases = [c for c in _golden() if tuple(c["weights"]) == tuple(weights)]
        off, n_ops, ty, ln = fastr.sw_align(_batch_from_cases(cases), weights=weights)
        strings = osw.cigar_strings(n_ops, ty, ln)
        for c, o, s in zip(cases, off, strings):
            assert (int(o), s) == (c["offset"], c["cigar"]), (c["ref"][:30], c["read"][:30])


@pytest.mark.gpu
def test_gpu_batch_equals_oracle(cuda):
    _compare(osw.make_pairs(seed=0x5117, n_windows=16))


@pytest.mark.gpu
def test_gpu_long_reads_and_small_workspace(cuda):
    """Reads of several hundred bases (several 64-row blocks AND long columns), and a workspace
    that holds 5 pairs so the call runs as many launches."""
    b = osw.make_pairs(seed=99, n_windows=6, reads_per_window=(5, 9), ref_len=(500, 900), read_len=(200, 420))
    _compare(b, workspace_pairs=5)


@pytest.mark.gpu
def test_gpu_reports_overflow_and_bad_lengths(cuda):
    from ctc_amd import fastr
    b = osw.make_pairs(seed=3, n_windows=2, reads_per_window=(4, 4), ref_len=(200, 300))
    d = fastr.DeviceAlignmentBatch(b, "cuda:0", max_ops=1)
    off, n_ops, ty, ln = d.run().results()
    w_off, w_n, _, _ = osw.align_batch(b)
    assert np.array_equal(n_ops, w_n)        # the true counts, also where they exceed max_ops
    assert np.array_equal(off, w_off)
    d2 = fastr.DeviceAlignmentBatch(b, "cuda:0")
    d2.max_read_len = int(b["read_len"].min())   # some reads now exceed the declared bound
    n2 = d2.run().results()[1]
    assert set(np.unique(n2[b["read_len"] > d2.max_read_len])) == {-1}
    assert np.array_equal(n2[b["read_len"] <= d2.max_read_len], w_n[b["read_len"] <= d2.max_read_len])
