"""ctc_tc_execute_noalloc under HIP graph capture (one of the callers it exists for): captured once, replayed,
bit-equal to ctc_tc_execute."""
import pytest
import torch

import ctc_amd

pytestmark = pytest.mark.gpu

SPEC = "aeim,ebmj->abij"


@pytest.mark.parametrize("n,dtype", [(48, "f64"), (24, "f64"), (48, "f32")])
def test_noalloc_replays_from_a_captured_graph(cuda, n, dtype):
    tdt = {"f64": torch.float64, "f32": torch.float32}[dtype]
    ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
    A = torch.empty(n ** 4, dtype=tdt, device=cuda)
    B = torch.empty(n ** 4, dtype=tdt, device=cuda)
    ctc_amd.fill(A, 0xC7C0A)
    ctc_amd.fill(B, 0xC7C0B)
    plan = ctc_amd.Plan(SPEC, ext, dtype)
    plan.prepare()
    assert plan.schedule()["workspace_bytes"] > 0  # the launch uses the scratch buffer and a fix-up pass
    want = torch.empty(n ** 4, dtype=tdt, device=cuda)
    plan.execute(A, B, want)
    got = torch.full((n ** 4,), float("nan"), dtype=tdt, device=cuda)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()  # the fills above ran on the default stream: the side stream must not overtake them
    with torch.cuda.stream(side):
        plan.execute_noalloc(A, B, got)  # once outside capture: every lazily cached query has been made
        side.synchronize()
        assert torch.equal(got, want)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            plan.execute_noalloc(A, B, got)
    for _ in range(3):
        got.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    # new operand values through the same graph: the graph holds pointers, not data
    ctc_amd.fill(A, 0xABCDE)
    plan.execute(A, B, want)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
