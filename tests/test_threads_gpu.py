"""One plan used from several host threads on several streams at once (the header's promise: a plan is immutable
after creation and may be used from several threads).  The launches share the plan's split-K workspace, which
the library orders through an event; the first executions race for the upload of the plan's tables.  Every
result must equal the single-threaded one bit for bit."""
import threading

import pytest
import torch

import ctc_amd

pytestmark = pytest.mark.gpu

SPEC = "aeim,ebmj->abij"


def _operands(n, dtype, dev):
    A = torch.empty(n ** 4, dtype=dtype, device=dev)
    B = torch.empty(n ** 4, dtype=dtype, device=dev)
    ctc_amd.fill(A, 0xC7C0A)
    ctc_amd.fill(B, 0xC7C0B)
    return A, B


@pytest.mark.parametrize("prepared,n", [(False, 48), (True, 48), (False, 49), (True, 49)])
def test_one_plan_many_threads_and_streams(cuda, prepared, n):
    threads, reps = 6, 8
    # extent 48 (2304^3): the last round of tiles is cut along K, so every launch uses the plan's workspace;
    # extent 49: odd, every execution copies A and B into the plan's padded workspace first (serialised per plan)
    A, B = _operands(n, torch.float64, cuda)
    ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
    ref_plan = ctc_amd.Plan(SPEC, ext, "f64")
    ref_plan.prepare()
    if n % 2:
        assert ref_plan.summary()["transposes"] == "padded_copies"
    else:
        assert ref_plan.schedule()["workspace_bytes"] > 0
    want = torch.empty(n ** 4, dtype=torch.float64, device=cuda)
    ref_plan.execute(A, B, want)
    torch.cuda.synchronize()

    plan = ctc_amd.Plan(SPEC, ext, "f64")  # a fresh plan: unprepared, its first executions race for the table upload
    if prepared:
        plan.prepare()
    outs = [torch.full((n ** 4,), float("nan"), dtype=torch.float64, device=cuda) for _ in range(threads)]
    torch.cuda.synchronize()  # the fills ran on the default stream: the workers' streams must not overtake them
    gate = threading.Barrier(threads)
    errors = []

    def worker(i):
        try:
            torch.cuda.set_device(cuda)
            s = torch.cuda.Stream(device=cuda)
            gate.wait(timeout=60)
            for _ in range(reps):
                plan.execute(A, B, outs[i], stream=s)
            s.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts), "a worker did not finish"
    assert not errors, errors
    torch.cuda.synchronize()
    for i, out in enumerate(outs):
        assert torch.equal(out, want), "thread %d" % i


def test_different_plans_from_different_threads(cuda, oracle):
    """Plans of different shapes and element types created and executed concurrently: the per-kernel occupancy
    caches, the permutation-table cache and the launch counters are shared state."""
    import numpy as np
    cases = [("aeim,ebmj->abij", dict(a=24, b=24, i=24, j=24, e=24, m=24), "f64"),
             ("aeim,ebmj->abij", dict(a=20, b=20, i=20, j=20, e=20, m=20), "f32"),
             ("adij,bcdk->abcijk", dict(a=4, b=12, c=12, i=12, j=12, k=12, d=64), "f64"),
             ("ak,kb->ab", dict(a=700, b=900, k=1024), "f64"),
             ("abc,cd->dab", dict(a=40, b=50, c=60, d=70), "f32"),
             ("dkca,bk->abcd", dict(a=32, b=8, c=32, d=32, k=8), "f32")]
    code = {"f64": 0, "f32": 1}
    tdt = {"f64": torch.float64, "f32": torch.float32}
    tol = {"f64": 1e-12, "f32": 1e-5}
    results, errors = {}, []
    gate = threading.Barrier(len(cases))
    host = {}
    for i, (spec, ext, dt) in enumerate(cases):
        ins, outm = spec.split("->")
        la, lb = ins.split(",")
        num = lambda t: int(np.prod([ext[c] for c in t]))  # noqa: E731
        host[i] = (oracle.fill(num(la), 100 + i, code[dt]), oracle.fill(num(lb), 200 + i, code[dt]), num(outm))

    def worker(i):
        try:
            spec, ext, dt = cases[i]
            torch.cuda.set_device(cuda)
            s = torch.cuda.Stream(device=cuda)
            hA, hB, nc = host[i]
            with torch.cuda.stream(s):
                dA, dB = torch.from_numpy(hA).to(cuda), torch.from_numpy(hB).to(cuda)
                dC = torch.full((nc,), float("nan"), dtype=tdt[dt], device=cuda)
            s.synchronize()
            gate.wait(timeout=60)
            for _ in range(4):
                plan = ctc_amd.Plan(spec, ext, dt)
                plan.execute(dA, dB, dC, stream=s)
                s.synchronize()
            results[i] = dC.cpu().numpy()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(len(cases))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=180)
    assert not any(t.is_alive() for t in ts), "a worker did not finish"
    assert not errors, errors
    for i, (spec, ext, dt) in enumerate(cases):
        hA, hB, _ = host[i]
        want = oracle.contract(spec, ext, hA, hB, dtype=code[dt]).astype(np.float64)
        got = results[i].astype(np.float64)
        assert np.abs(got - want).max() <= tol[dt] * np.abs(want).max(), (spec, dt)


def test_scoring_from_several_threads(cuda):
    """ctc_fastr_best_offsets stages its work table through a ring of four (pinned, device) buffer pairs per GPU:
    six threads on six streams, different batches, more calls in flight than slots.  Every call's result must
    equal the oracle's."""
    import numpy as np
    from oracle import fastr as ofastr
    from ctc_amd import fastr
    threads, reps = 6, 12
    batches = [ofastr.make_batch(seed=500 + i, n_regions=3 + i, max_consensus=4, reads=(4, 14), read_len=48, ref_len=(100, 400))
               for i in range(threads)]
    wants = [ofastr.best_offsets(b) for b in batches]
    errors, bad = [], []
    gate = threading.Barrier(threads)

    def worker(i):
        try:
            torch.cuda.set_device(cuda)
            s = torch.cuda.Stream(device=cuda)
            with torch.cuda.stream(s):
                d = fastr.DeviceBatch(batches[i], cuda)
            s.synchronize()
            gate.wait(timeout=60)
            for r in range(reps):
                d.best_score.fill_(-7)
                d.best_index.fill_(-7)
                torch.cuda.current_stream().synchronize()
                d.run(stream=s)
                s.synchronize()
                score, index = d.results()
                if not (np.array_equal(score, wants[i][0]) and np.array_equal(index, wants[i][1])):
                    bad.append((i, r))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=180)
    assert not any(t.is_alive() for t in ts), "a worker did not finish"
    assert not errors, errors
    assert not bad, bad
