"""Headline benchmark: fp64 CCSD-style contraction C[a,b,i,j] = sum_em A[a,e,i,m] * B[e,b,m,j]
(BASELINE.json configs[1]) on MI355X through the C ABI of libctc_tc.so.

  python bench.py --gpus N --steps K --warmup W [--extent 96] [--shard free|contracted]

A step is one whole contraction with operands already resident in HBM.  With
N > 1 (one process per GPU under torch.distributed.run) the SAME contraction is
split along the free index `a` (extent / N per rank): every rank owns one slab
of A and of C, B is replicated, no collective touches the data path, the step
time is the slowest rank's ("strong" scaling: value(N) / value(1) is the
speed-up).  The contraction grown with N instead (a = N * extent, "weak") is
reported under config.variants.  Rank 0 prints one JSON line.

--shard contracted (BASELINE configs[3], second variant) keeps the contraction at all extents
`extent` and splits the contracted index e over the ranks instead ("strong" scaling): every
rank holds 1/N of A and of B, computes a full-size partial C and the partials are summed by
one RCCL reduce-scatter, after which every rank owns |C|/N.  The step time includes the
collective.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import ctc_amd  # noqa: E402

SPEC = "aeim,ebmj->abij"
SEED_A, SEED_B = 0xC7C0A, 0xC7C0B
# fp64 matrix peak of MI355X, dense, public datasheet value (256 CUs x 4 SIMDs x
# 32 flop/clk x 2.4 GHz); the measured microarchitecture guide lists no fp64 row.
PEAK_FP64_MFMA_TFLOPS = 78.6
TRAFFIC_FILE = "r03_bench_traffic.json"  # counter passes of the kernel the bench currently runs


def cpu_baseline(extent, threads):
    """The CPU oracle on the same contraction at a bounded extent (a few seconds of work)."""
    from oracle import ref
    import numpy as np
    n = extent
    ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
    A = ref.fill(n ** 4, SEED_A)
    B = ref.fill(n ** 4, SEED_B)
    ref.contract(SPEC, dict(a=4, b=4, i=4, j=4, e=4, m=4), A, B, threads=threads)  # spin up the thread pool
    t0 = time.perf_counter()
    C = ref.contract(SPEC, ext, A, B, threads=threads)
    dt = time.perf_counter() - t0
    return 2.0 * n ** 6 / dt * 1e-12, dt, float(np.abs(C).max())


def cpu_sample_extent(threads):
    """The CPU sample is bounded: the largest extent whose 2 n^6 flops take at most about 30 s at 5 GFLOP/s
    per thread (the oracle measures 7-12 per thread).  A launcher that sets OMP_NUM_THREADS=1 for its
    workers must not turn the sample into minutes that the other ranks spend at a barrier."""
    for n in (96, 80, 64):
        if 2.0 * n ** 6 <= 30.0 * 5e9 * threads:
            return n
    return 48


def cpu_threads():
    """Host threads the CPU baseline runs on: the process's CPU share (its affinity mask, capped by
    OMP_NUM_THREADS when the environment sets one: on a shared GPU host os.cpu_count() counts the whole
    machine, not this job's share)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS", "")
    return max(1, min(avail, int(cap))) if cap.isdigit() and int(cap) > 0 else avail


def verify_slabs(dC, n, first_a, a_idx, b_list):
    """C[a, b, :, :] of the timed contraction recomputed by the CPU oracle from regenerated operands,
    for one a (index within this rank's slab; first_a = the slab's first global a) and a few b.
    Returns the largest |difference| / max |reference| (the parity bar is 1e-12 for fp64)."""
    from oracle import ref
    import numpy as np
    # A[a_idx] is n^3 consecutive elements of the synthetic stream; B is needed whole
    A1 = ref.fill(n ** 3, SEED_A + (first_a + a_idx) * n ** 3)
    B = ref.fill(n ** 4, SEED_B).reshape(n, n, n, n)
    got_all = dC.reshape(-1, n, n, n)[a_idx].cpu().numpy()
    worst = 0.0
    for b in b_list:
        want = ref.contract(SPEC, dict(a=1, b=1, i=n, j=n, e=n, m=n), A1,
                            np.ascontiguousarray(B[:, b:b + 1]).reshape(-1))
        err = float(np.abs(got_all[b].reshape(-1) - want).max() / np.abs(want).max())
        worst = max(worst, err)
    return worst


def measured_traffic(extent, kernel_summary):
    """HBM-side bytes per launch from the committed counter passes (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE, one pass each, tools/pmc_traffic.sh); None when they do not cover this run."""
    path = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if extent != 96 or not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    # the passes count for one kernel instantiation: the whole name must agree, template arguments included
    if kernel_summary["kernel_name"].replace(" ", "") not in t.get("kernel", "").replace(" ", ""):
        return None
    return t["per_launch_bytes"]["total_corrected"]


def _barrier(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def _max_over_ranks(values, world, dev, backend):
    if world == 1:
        return [float(v) for v in values]
    t = torch.tensor(list(values), dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def run_contracted(n, world, rank, dev, backend, steps, warmup, nchunks=4):
    """BASELINE configs[3], second variant: abij=aeim,ebmj at all extents n with the CONTRACTED index e
    split over the ranks.  Every rank holds 1/N of A and of B, the partial results are summed by RCCL
    reduce-scatters issued from C (ctc_tc_sharded_execute) chunk by chunk behind the GEMMs; afterwards
    every rank owns |C|/N.  Strong scaling; the step time includes the collectives."""
    from ctc_amd.sharded import Communicator, ShardedContraction
    ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
    sc = ShardedContraction(SPEC, ext, "f64", "e", rank, world, reduce="reduce_scatter", nchunks=nchunks)
    A = torch.empty(sc.local_numel("a"), dtype=torch.float64, device=dev)
    B = torch.empty(sc.local_numel("b"), dtype=torch.float64, device=dev)
    ctc_amd.fill(A, SEED_A + rank * 0x100000000)
    ctc_amd.fill(B, SEED_B + rank * 0x100000000)
    partial = torch.empty(sc.workspace_numel(), dtype=torch.float64, device=dev)
    C = torch.empty(sc.out_numel(), dtype=torch.float64, device=dev)
    comm = Communicator(rank, world)  # a 1-rank communicator still runs the real RCCL call
    sc.prepare()
    for _ in range(warmup):
        sc.execute(A, B, C, comm=comm, scratch=partial)
    _barrier(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        sc.execute(A, B, C, comm=comm, scratch=partial)
    _barrier(world)
    elapsed = time.perf_counter() - t0
    # the same GEMMs without the collectives
    t1 = time.perf_counter()
    for _ in range(steps):
        for c in range(sc.nchunks):
            plan, oa, ob, op, _, elems = sc.chunk(c)
            plan.execute(A[oa:], B[ob:], partial[op:op + elems])
    torch.cuda.synchronize()
    gemm_only = time.perf_counter() - t1
    elapsed, gemm_only = _max_over_ranks([elapsed, gemm_only], world, dev, backend)
    flops = 2 * n ** 6
    out = {
        "workload": "BASELINE configs[3] variant: abij=aeim,ebmj fp64, all extents %d, contracted index e split over "
                    "%d GPU(s), %d pipelined RCCL reduce-scatter(s) of C (%d MB in all) per step"
                    % (n, world, sc.nchunks, sc.full_c_numel() * 8 // 1000000),
        "scaling": "strong", "n_gpus": world, "steps": steps,
        "ms_per_step": round(elapsed / steps * 1e3, 4), "tflops": round(flops / (elapsed / steps) * 1e-12, 3),
        "ms_per_step_gemms_only": round(gemm_only / steps * 1e3, 4),
        "collective": "ncclReduceScatter x %d on a second stream, issued by libctc_tc.so" % sc.nchunks,
    }
    del comm
    return out


def run_config4(world, rank, dev, backend, steps, warmup):
    """BASELINE configs[3], first variant: abcijk=adij,bcdk fp64, free extents 40, d = 64 (|C| = 32.8 GB),
    free index a split over the ranks (40 / N each): no collective, strong scaling."""
    n, d = 40, 64
    if n % world:
        return {"skipped": "40 is not a multiple of %d" % world}
    spec = "adij,bcdk->abcijk"
    full = ctc_amd.Plan(spec, dict(a=n, b=n, c=n, i=n, j=n, k=n, d=d), "f64")
    plan, off_a, _, _ = full.shard("a", rank, world)
    a_loc = n // world
    A = torch.empty(a_loc * d * n * n, dtype=torch.float64, device=dev)
    B = torch.empty(n * n * d * n, dtype=torch.float64, device=dev)
    C = torch.empty(a_loc * n ** 5, dtype=torch.float64, device=dev)
    ctc_amd.fill(A, SEED_A, first=off_a)
    ctc_amd.fill(B, SEED_B)
    plan.prepare()
    for _ in range(warmup):
        plan.execute(A, B, C)
    _barrier(world)
    t0 = time.perf_counter()
    kernel_ms = plan.time(A, B, C, warmup=0, iters=steps)
    _barrier(world)
    elapsed = time.perf_counter() - t0
    elapsed, kernel_ms = _max_over_ranks([elapsed, kernel_ms], world, dev, backend)
    flops = 2 * n ** 6 * d
    tf_rank = plan.flops / (kernel_ms * 1e-3) * 1e-12
    return {
        "workload": "BASELINE configs[3]: abcijk=adij,bcdk fp64, free extents 40, d=64, free index a split over %d GPU(s) "
                    "(%d each), |C| %.1f GB per GPU, no collective" % (world, a_loc, a_loc * n ** 5 * 8e-9),
        "scaling": "strong", "n_gpus": world, "steps": steps,
        "ms_per_step": round(elapsed / steps * 1e3, 4), "tflops": round(flops / (elapsed / steps) * 1e-12, 3),
        "per_gpu_kernel_tflops": round(tf_rank, 3), "per_gpu_frac_of_fp64_mfma_peak": round(tf_rank / PEAK_FP64_MFMA_TFLOPS, 4),
        "per_gpu_c_write_TBps": round(a_loc * n ** 5 * 8 / (kernel_ms * 1e-3) * 1e-12, 3),
    }


def bench_contracted(args, world, rank, dev, backend):
    """--shard contracted: the contracted-index variant as the headline line."""
    v = run_contracted(args.extent, world, rank, dev, backend, args.steps, args.warmup)
    if rank == 0:
        n = args.extent
        emit(({
            "metric": "achieved TFLOP/s per contraction (fp64, C[a,b,i,j]=A[a,e,i,m]*B[e,b,m,j])",
            "value": v["tflops"], "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": v["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (splitmix64 uniform[-1,1), generated in HBM)",
            "config": {"workload": v["workload"], "collective": v["collective"], "flops_per_step": 2 * n ** 6,
                       "ms_per_step_gemms_only": v["ms_per_step_gemms_only"],
                       "pct_of_fp64_mfma_peak": round(100.0 * v["tflops"] / (PEAK_FP64_MFMA_TFLOPS * world), 2)},
        }))
    if world > 1:
        dist.destroy_process_group()


def _claim_stdout():
    """Rank 0 prints ONE JSON line on stdout.  Libraries underneath (RCCL prints a version banner when a
    communicator is created) write to file descriptor 1 as well, so the descriptor is pointed at stderr
    for the whole run and the line goes to a private copy of the original."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(keep, "w")


RESULT = None  # the original stdout, claimed in main()

# A child process that owns the original stdout and prints the LAST line it was offered, once, when its
# pipe closes, i.e. when this process is done or dies.  The headline line is offered as soon as it is
# measured, the same line with config.variants filled in after the secondary measurements: whatever a
# secondary measurement does at a world size that was never rehearsed (a collective that hangs, an abort
# inside a library), the run still prints exactly one line with the headline numbers.
KEEPER_SOURCE = (
    "import signal, sys\n"
    "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
    "signal.signal(signal.SIGINT, signal.SIG_IGN)\n"
    "last = None\n"
    "for line in sys.stdin:\n"
    "    if line.strip():\n"
    "        last = line\n"
    "if last:\n"
    "    sys.stdout.write(last)\n"
    "    sys.stdout.flush()\n"
)
KEEPER = None
KEEPER_LOCK = threading.Lock()  # the watchdog thread and the main thread both offer lines


def start_keeper():
    """Rank 0, before anything touches the GPU."""
    global KEEPER
    r, w = os.pipe()
    proc = subprocess.Popen([sys.executable, "-c", KEEPER_SOURCE], stdin=r, stdout=RESULT.fileno(), close_fds=True)
    os.close(r)
    KEEPER = (proc, os.fdopen(w, "w"))


def offer(obj):
    """The line to print unless a later one replaces it."""
    with KEEPER_LOCK:
        if KEEPER is not None:
            KEEPER[1].write(json.dumps(obj) + "\n")
            KEEPER[1].flush()


def finish():
    """Closes the keeper's pipe (it prints the last line offered) and waits for it."""
    global KEEPER
    with KEEPER_LOCK:
        keeper, KEEPER = KEEPER, None
    if keeper is not None:
        proc, w = keeper
        w.close()
        proc.wait()


def emit(obj):
    offer(obj)
    finish()


EXIT_VARIANTS_TIMED_OUT = 3


def watchdog(seconds, rank, line, state):
    """Ends this rank, with a non-zero exit status, if the secondary measurements do not return.  Rank 0
    first prints the headline line with the time-out recorded in config.variants (what had finished stays,
    the measurement that hung is named), so a collective that hangs is visible both in the record and in
    the exit status.  A thread, not a signal handler: the main thread may be stuck inside a library call."""
    def fire():
        sys.stderr.write("bench.py: secondary measurements still running after %d s on rank %d (in %r): giving up\n"
                         % (seconds, rank, state.get("running")))
        sys.stderr.flush()
        if line is not None:
            variants = dict(state.get("done", {}))
            variants["timeout"] = {"after_s": seconds, "running": state.get("running"),
                                   "exit_status": EXIT_VARIANTS_TIMED_OUT}
            line["config"]["variants"] = variants
            offer(line)
        finish()
        os._exit(EXIT_VARIANTS_TIMED_OUT)
    # rank 0 goes first: a launcher that sees another rank fail ends rank 0 before it has recorded the time-out
    # (rank 0 starts its variants later than the others by the time its CPU baseline takes, some 10 s)
    t = threading.Timer(seconds if rank == 0 else seconds + 30.0, fire)
    t.daemon = True
    t.start()
    return t


def run_weak(n, world, rank, dev, backend, steps, warmup):
    """The contraction grown with the job: a = N * n, one slab of n per rank (what every rank computes is
    the N = 1 headline contraction), no collective: weak scaling."""
    full = ctc_amd.Plan(SPEC, dict(a=n * world, b=n, i=n, j=n, e=n, m=n), "f64")
    plan, off_a, _, _ = full.shard("a", rank, world)
    A = torch.empty(n ** 4, dtype=torch.float64, device=dev)
    B = torch.empty(n ** 4, dtype=torch.float64, device=dev)
    C = torch.empty(n ** 4, dtype=torch.float64, device=dev)
    ctc_amd.fill(A, SEED_A, first=off_a)
    ctc_amd.fill(B, SEED_B)
    plan.prepare()
    for _ in range(warmup):
        plan.execute(A, B, C)
    _barrier(world)
    t0 = time.perf_counter()
    kernel_ms = plan.time(A, B, C, warmup=0, iters=steps)
    _barrier(world)
    elapsed = time.perf_counter() - t0
    elapsed, kernel_ms = _max_over_ranks([elapsed, kernel_ms], world, dev, backend)
    flops = 2 * n ** 6 * world
    return {"workload": "abij=aeim,ebmj fp64, a = %d (= %d x %d), other extents %d, free index a split over %d GPU(s), no collective"
                        % (n * world, world, n, n, world),
            "scaling": "weak", "n_gpus": world, "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 4),
            "tflops": round(flops / (elapsed / steps) * 1e-12, 3), "per_gpu_kernel_ms": round(kernel_ms, 4)}


def main():
    global RESULT
    RESULT = _claim_stdout()
    if int(os.environ.get("RANK", "0")) == 0:
        start_keeper()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--extent", type=int, default=96, help="extent of every mode (64..96 in BASELINE)")
    ap.add_argument("--cpu-extent", type=int, default=0,
                    help="extent of the CPU-oracle sample (0: the largest of 96 / 80 / 64 / 48 that the host threads finish in about 30 s)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary measurements (config.variants)")
    ap.add_argument("--no-verify", action="store_true", help="skip the check of the result against the CPU oracle")
    ap.add_argument("--shard", choices=("free", "contracted"), default="free")
    ap.add_argument("--prewarm", type=float, default=0.1, help="seconds of untimed load before the warm-up steps")
    ap.add_argument("--variant-timeout", type=float, default=240.0,
                    help="seconds after which the secondary measurements are abandoned: the headline line is printed with "
                         "the time-out recorded in config.variants and the process exits with status %d" % EXIT_VARIANTS_TIMED_OUT)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # CTC_BENCH_BACKEND=gloo rehearses the multi-process path on a box with fewer GPUs than ranks
    backend = os.environ.get("CTC_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run" % (args.gpus, world))

    n = args.extent
    if args.shard == "contracted":
        return bench_contracted(args, world, rank, dev, backend)
    if n % world:
        raise SystemExit("--extent %d is not a multiple of --gpus %d: the free index a is split evenly" % (n, world))
    # ONE contraction whatever N is: all extents n, the free index a split over the ranks (n / N each)
    a_loc = n // world
    full = ctc_amd.Plan(SPEC, dict(a=n, b=n, i=n, j=n, e=n, m=n), "f64")
    plan, off_a, off_b, off_c = full.shard("a", rank, world)
    # each rank holds only its slab of A and C; B is replicated
    slab = a_loc * n ** 3
    A = torch.empty(slab, dtype=torch.float64, device=dev)
    B = torch.empty(n ** 4, dtype=torch.float64, device=dev)
    C = torch.empty(slab, dtype=torch.float64, device=dev)
    ctc_amd.fill(A, SEED_A, first=off_a)
    ctc_amd.fill(B, SEED_B)
    plan.prepare()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The chip needs some tens of milliseconds of load before it holds its clocks (the first
    # 30 ms of launches run about 7 % slower): load it for a fixed time, untimed, before the
    # W warm-up steps, so that short configurations are measured in the same state as long ones.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm:
        plan.execute(A, B, C)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        plan.execute(A, B, C)
    barrier()
    counters = ("launches_dma_gemm", "launches_classic", "launches_stream", "launches_stream_f64_dma")
    launches_before = [ctc_amd.get_option(c) for c in counters]
    t0 = time.perf_counter()
    # K launches on torch's current stream, a HIP event on that same stream between every two
    ms_each = plan.time_each(A, B, C, iters=args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(ms_each) / len(ms_each)
    kernel_ms_median = sorted(ms_each)[len(ms_each) // 2] if len(ms_each) % 2 else \
        0.5 * (sorted(ms_each)[len(ms_each) // 2 - 1] + sorted(ms_each)[len(ms_each) // 2])
    elapsed, kernel_ms, kernel_ms_median = _max_over_ranks([elapsed, kernel_ms, kernel_ms_median], world, dev, backend)

    flops_rank = plan.flops
    flops_job = 2 * n ** 6
    assert flops_rank * world == flops_job
    ms_per_step = elapsed / args.steps * 1e3
    value = flops_job / (elapsed / args.steps) * 1e-12

    # Outside the timed region: the result the timed launches left in C, checked against the CPU oracle
    # on C[a, b, :, :] for one a of this rank's slab and three b (fp64 bar: 1e-12 of max |C|).
    rel_err = None
    if not args.no_verify:
        rel_err = verify_slabs(C, n, rank * a_loc, a_loc // 3, (0, n // 2, n - 1))
        rel_err = _max_over_ranks([rel_err], world, dev, backend)[0]

    out = None
    if rank == 0:
        # the plan whose GEMM the launches ran (the plan itself unless it goes through copies of its operands: odd
        # extents) and which tiled-GEMM kernel they took (the library counts them and reports the form)
        gemm = plan.gemm_plan()
        s = gemm.summary()
        s["transposes"] = plan.summary()["transposes"]
        took = [c for c, b in zip(counters, launches_before) if ctc_amd.get_option(c) > b]
        name, tile = gemm.gemm_kernel_name()
        if took == ["launches_dma_gemm"] or took == ["launches_classic"]:
            s["kernel_name"], s["kernel_tile"] = name, tile
        else:
            s["kernel_name"], s["kernel_tile"] = "+".join(t[9:] for t in took), list(s["tile"])
        achieved = flops_rank / (kernel_ms * 1e-3) * 1e-12
        out = {
            "metric": "achieved TFLOP/s per contraction (fp64, C[a,b,i,j]=A[a,e,i,m]*B[e,b,m,j])",
            "value": round(value, 3),
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (splitmix64 uniform[-1,1), generated in HBM)",
            "config": {
                "workload": "BASELINE configs[1]: abij=aeim,ebmj fp64, all extents %d%s" % (
                    n, "" if world == 1 else ", the same contraction split along the free index a over %d GPUs "
                                             "(%d each), no collective" % (world, a_loc)),
                "gemm_view_per_gpu": {"M": s["M"], "N": s["N"], "K": s["K"]},
                "tile": s["kernel_tile"], "layout_a": s["layout_a"], "layout_b": s["layout_b"], "operand_copies": s["transposes"],
                "flops_per_step": flops_job,
                "pct_of_fp64_mfma_peak": round(100.0 * value / (PEAK_FP64_MFMA_TFLOPS * world), 2),
                "variants": {},
            },
            "roofline": {
                "bound": "mfma",
                "achieved": round(achieved, 3),
                "peak": PEAK_FP64_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP64_MFMA_TFLOPS, 4),
                "traffic": measured_traffic(n, s) if world == 1 else None,
                "traffic_unit": "bytes per launch at the L2's memory side (separate counter passes, profiles/%s)" % TRAFFIC_FILE,
                "algorithmic_bytes_per_launch": 8 * (2 * a_loc * n ** 3 + n ** 4),
                "kernel": s["kernel_name"],
                "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_median": round(kernel_ms_median, 4),
                "frac_from_median": round(flops_rank / (kernel_ms_median * 1e-3) * 1e-12 / PEAK_FP64_MFMA_TFLOPS, 4),
                "per": "GPU (the slowest rank's launches; every rank runs the same kernel on its slab)",
                "algorithmic_flops_per_launch": flops_rank,
            },
        }
        if rel_err is not None:
            out["verified"] = bool(rel_err <= 1e-12)
            out["verify"] = {"max_rel_err": rel_err, "tolerance": 1e-12,
                             "what": "C[a,b,:,:] for one a of every rank's slab (its index %d), b in (0,%d,%d), vs the CPU "
                                     "oracle, after the timed launches" % (a_loc // 3, n // 2, n - 1)}
        offer(out)  # printed as it stands if anything below takes the process down
        if not args.no_cpu:
            # the CPU column, at every N: rank 0's host cores, while the other ranks wait at the next barrier
            threads = cpu_threads()
            cpu_n = args.cpu_extent or cpu_sample_extent(threads)
            tf, dt, _ = cpu_baseline(cpu_n, threads)
            out["cpu_baseline"] = {
                "value": round(tf, 5), "unit": "TFLOP/s", "cores": threads, "threads": threads,
                "host_logical_cpus": os.cpu_count(), "kind": "port",
                "sample": "same contraction at all extents %d (2*%d^6 flops, %.1f s) on the build's own CPU "
                          "oracle (OpenMP over the outermost free index, every CPU of this job's share); the "
                          "reference has no CPU contraction path" % (cpu_n, cpu_n, dt),
            }
            offer(out)

    # Secondary measurements, after and outside the headline's timed region (every rank takes part; a
    # failure in one of them is reported in its place and does not cost the headline line; one that does not
    # return ends the run with the time-out on record and a non-zero exit status).
    if not args.no_variants:
        state = {"running": None, "done": {}}
        guard = watchdog(args.variant_timeout, rank, out, state)
        del C
        torch.cuda.empty_cache()
        todo = [("config4_free_index_strong", lambda: run_config4(world, rank, dev, backend, 4, 2)),
                ("contracted_index_reduce_scatter", lambda: run_contracted(n, world, rank, dev, backend, 6, 2))]
        if world > 1:
            todo.insert(0, ("free_index_weak", lambda: run_weak(n, world, rank, dev, backend, args.steps, args.warmup)))
        for name, fn in todo:
            state["running"] = name
            try:
                state["done"][name] = fn()
            except Exception as e:  # noqa: BLE001
                state["done"][name] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
        state["running"] = None
        guard.cancel()
        if rank == 0:
            out["config"]["variants"] = state["done"]
            offer(out)
    if rank == 0:
        finish()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
