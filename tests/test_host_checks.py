"""The Python mirror checks what it can see before it hands raw pointers to the C ABI: the kernels trust
their offset tables, so a buffer shorter than the plan addresses (or of another element size) would be read or
written past its end on the device.  CTC_TC_Q_SPANS reports what a plan addresses."""
import pytest
import torch

import ctc_amd
from ctc_amd import _lib

SPEC = "aeim,ebmj->abij"
EXT = dict(a=4, b=5, i=6, j=7, e=3, m=2)


def test_spans_of_packed_and_strided_plans_and_slabs():
    p = ctc_amd.Plan(SPEC, EXT, "f64")
    assert p.spans() == (4 * 3 * 6 * 2, 3 * 5 * 2 * 7, 4 * 5 * 6 * 7)
    # spans are 1 + the largest offset of the tables
    gemm_a = 1 if p.summary()["swapped"] else 0  # the GEMM's A operand is the caller's B when the roles are exchanged
    assert p.table(_lib.Q_ROW_A).max() + p.table(_lib.Q_K_A).max() + 1 == p.spans()[gemm_a]
    assert p.table(_lib.Q_COL_B).max() + p.table(_lib.Q_K_B).max() + 1 == p.spans()[1 - gemm_a]
    assert p.table(_lib.Q_ROW_C).max() + p.table(_lib.Q_COL_C).max() + 1 == p.spans()[2]
    # a padded fastest mode: the span follows the strides, not the element count
    q = ctc_amd.Plan("ak,kb->ab", dict(a=3, b=4, k=5), "f32", strides_a=[8, 1], strides_b=[6, 1], strides_c=[4, 1])
    assert q.spans() == (2 * 8 + 5, 4 * 6 + 4, 12)
    # a slab of a mode that is not the slowest keeps the full strides
    s, oa, ob, oc = p.shard("i", 1, 2)
    assert (oa, ob, oc) == (3 * 2, 0, 3 * 7)
    assert s.spans() == (4 * 3 * 6 * 2 - 3 * 2, 210, 840 - 3 * 7)
    # scalar result
    assert ctc_amd.Plan("k,k->", dict(k=9), "f64").spans() == (9, 9, 1)


@pytest.mark.parametrize("method", ["execute", "execute_noalloc", "time", "time_each"])
def test_short_or_mistyped_buffers_are_refused_before_any_launch(method):
    p = ctc_amd.Plan(SPEC, EXT, "f64")
    na, nb, nc = p.spans()
    ok = lambda n: torch.zeros(n, dtype=torch.float64)  # noqa: E731  (host tensors: the size check comes first)
    call = getattr(p, method)
    with pytest.raises(ValueError, match="operand A holds %d elements" % (na - 1)):
        call(ok(na - 1), ok(nb), ok(nc))
    with pytest.raises(ValueError, match="operand B holds"):
        call(ok(na), ok(nb)[1:], ok(nc))
    with pytest.raises(ValueError, match="operand C holds 0 elements"):
        call(ok(na), ok(nb), ok(nc)[nc:])
    with pytest.raises(ValueError, match="4-byte elements"):
        call(ok(na), torch.zeros(nb, dtype=torch.float32), ok(nc))
    with pytest.raises(ValueError, match="operand C is None"):
        call(ok(na), ok(nb), None)
    # large enough: the next check is that the operands live in device memory
    with pytest.raises(ValueError, match="device memory"):
        call(ok(na), ok(nb), ok(nc))
    # room is counted to the end of the storage, from the view's first element
    big = ok(na + 10)
    with pytest.raises(ValueError, match="device memory"):
        call(big[10:], ok(nb), ok(nc))
    with pytest.raises(ValueError, match="operand A holds %d" % (na - 1)):
        call(big[11:], ok(nb), ok(nc))


def test_permute_and_fill_check_their_buffers():
    x = torch.zeros(2, 3, 4, dtype=torch.float32)
    with pytest.raises(ValueError, match="result must be packed"):
        ctc_amd.permute(x, (2, 0, 1), out=torch.zeros(23, dtype=torch.float32))
    with pytest.raises(ValueError, match="result must be packed"):
        ctc_amd.permute(x, (2, 0, 1), out=torch.zeros(24, dtype=torch.float64))
    with pytest.raises(ValueError, match="packed"):
        ctc_amd.fill(torch.zeros(4, 4, dtype=torch.float64)[:, :2], 1)


def test_sharded_execute_checks_sizes():
    from ctc_amd.sharded import ShardedContraction
    ext = dict(a=4, b=4, i=4, j=4, e=4, m=4)
    sc = ShardedContraction(SPEC, ext, "f64", "e", 0, 2, reduce="reduce_scatter", nchunks=2)
    z = lambda n: torch.zeros(n, dtype=torch.float64)  # noqa: E731
    with pytest.raises(ValueError, match="operand A holds"):
        sc.execute(z(sc.local_numel("a") - 1), z(sc.local_numel("b")), z(sc.out_numel()), scratch=z(sc.workspace_numel()))
    with pytest.raises(ValueError, match="operand C holds"):
        sc.execute(z(sc.local_numel("a")), z(sc.local_numel("b")), z(sc.out_numel() - 1), scratch=z(sc.workspace_numel()))
    with pytest.raises(ValueError, match="scratch holds"):
        sc.execute(z(sc.local_numel("a")), z(sc.local_numel("b")), z(sc.out_numel()), scratch=z(sc.workspace_numel() - 1))


def _scoring_batch():
    import numpy as np
    stride = 8
    return {"ref": np.zeros(30, np.uint8), "ref_lens": np.array([10, 12, 8], np.int32), "n_consensus": np.array([2, 1], np.int32),
            "alt_reads": np.array([2, 3], np.int32), "read_seq": np.zeros(5 * stride, np.uint8), "quals": np.zeros(5 * stride, np.uint8),
            "read_len": np.array([8, 7, 8, 3, 0], np.int32), "original_alignment": np.zeros(5, np.int32), "read_stride": stride}


def test_scoring_batches_are_validated_before_upload():
    import numpy as np
    from ctc_amd.fastr import DeviceBatch
    DeviceBatch.validate(_scoring_batch())
    for key, value, msg in (("ref", np.zeros(29, np.uint8), "ref: 30 bytes"),
                            ("ref_lens", np.array([10, 12], np.int32), "one non-negative length per consensus"),
                            ("ref_lens", np.array([10, -1, 8], np.int32), "one non-negative length per consensus"),
                            ("alt_reads", np.array([2, -3], np.int32), "non-negative count"),
                            ("read_len", np.array([8, 7, 8, 3], np.int32), "one entry per read"),
                            ("read_len", np.array([8, 7, 9, 3, 0], np.int32), "outside 0..read_stride"),
                            ("quals", np.zeros(39, np.uint8), "quals: 40 bytes"),
                            ("read_seq", np.zeros(40, np.int8), "read_seq: 40 bytes"),
                            ("read_stride", 0, "read_stride must be positive")):
        b = _scoring_batch()
        b[key] = value
        with pytest.raises(ValueError, match=msg):
            DeviceBatch(b, "cpu")


def test_alignment_batches_are_validated_before_upload():
    import numpy as np
    from ctc_amd.fastr import DeviceAlignmentBatch

    def batch():
        return {"ref": np.zeros(100, np.uint8), "ref_off": np.array([0, 40, 60], np.int64), "ref_len": np.array([40, 20, 40], np.int32),
                "reads": np.zeros(3 * 16, np.uint8), "read_len": np.array([16, 10, 0], np.int32), "read_stride": 16}
    DeviceAlignmentBatch.validate(batch())
    for key, value, msg in (("ref_off", np.array([0, 40, 61], np.int64), "window lies outside ref"),
                            ("ref_off", np.array([0, -1, 60], np.int64), "window lies outside ref"),
                            ("ref_len", np.array([40, 20], np.int32), "one entry per pair"),
                            ("read_len", np.array([17, 10, 0], np.int32), "outside 0..read_stride"),
                            ("reads", np.zeros(47, np.uint8), "read_stride bytes per pair")):
        b = batch()
        b[key] = value
        with pytest.raises(ValueError, match=msg):
            DeviceAlignmentBatch(b, "cpu")


def test_gemm_plan_is_the_plan_or_the_sub_plan_behind_the_copies():
    even = ctc_amd.Plan(SPEC, dict(a=48, b=48, i=48, j=48, e=48, m=48), "f64")
    assert even.gemm_plan() is even
    odd = ctc_amd.Plan(SPEC, dict(a=49, b=49, i=49, j=49, e=49, m=49), "f64")
    assert odd.summary()["transposes"] == "padded_copies"
    sub = odd.gemm_plan()
    s = sub.summary()
    assert sub is not odd and (s["M"], s["N"], s["K"]) == (2401, 2401, 2408), s   # K = 49 * 49 rounded up to 64 bytes
    assert (s["layout_a"], s["layout_b"], s["transposes"]) == ("k_vec", "k_vec", "fused"), s
    assert sub.spans()[2] == odd.spans()[2]   # C is the caller's
    assert odd.gemm_plan().summary() == s      # built once, handed out again
    # a batch mode as the unit-stride mode of all three: explicit transposes, GEMM on packed operands
    ttgt = ctc_amd.Plan("akz,kbz->abz", dict(a=64, b=64, k=64, z=8), "f64")
    assert ttgt.summary()["transposes"] == "explicit"
    packed = ttgt.gemm_plan().summary()
    assert packed["transposes"] == "fused" and packed["layout_a"] != "scalar", packed
