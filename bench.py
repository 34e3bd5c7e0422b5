#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

    python bench.py                                    # 1 GPU, 200 timed steps
    python bench.py --gpus 8 --steps 20 --warmup 5     # starts its own 8 ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W        # the same ranks, started by a launcher
    python bench.py --gpus 8 --mode single             # ONE process driving 8 GPUs (smhip_set_devices: what sm::Sharded uses)

metric   "Gelem/s" of the contiguous float32 add c = a + b (BASELINE.json `metric`, configs[1]): one step = one pass
         of smhip_contiguous over N = 2^28 elements per GPU, inputs already resident in HBM (generated on device by the
         counter-based hash), output preallocated.
N GPUs   each GPU owns one outer-dimension shard of 2^28 elements of the N * 2^28 array (config 5's partitioning).
         Elementwise work has no data-path collective, so scaling is "weak" and value = all GPUs' elements / time.
         --mode ranks (default): one process per GPU.  Plain `--gpus N` starts the N ranks itself through
         torch.distributed.run BEFORE this process has touched HIP; under a launcher (RANK in the environment) this
         process is one of the ranks.  torch.distributed (backend nccl = RCCL) provides the barrier and the
         max-over-ranks of the timings only.
         --mode single: one process, one host thread, N devices through libsmhip's device group.
         --mode single also runs config 3 over the group ("c5"."c3_sharded": 4096 rows per GPU, the row replicated device to device
         by smhip_copy_peer, smhip_sharded_elementwise -- the replicated-operand path and a real peer copy).
         Config 5's exchange step (fused add + sum per shard, then ONE ncclAllReduce of an fp64 scalar over xGMI, issued
         by libsmhip itself: smhip_allreduce_sum_async / smhip_sharded_contiguous_sum) is timed after the headline
         region and reported under "c5" (operands: config 5's seeds 6/7 in [0,1)).
roofline dominant kernel = contiguous_vec_kernel<float, AddOp<float>, 1024, false>; algorithmic bytes 12 B/elem (2 reads +
         1 write).  A step is ONE smhip_contiguous call over 2^28 elements, which the library issues as TWO launches of that
         kernel over 2^27 elements each (operands above 512 MiB go out in pieces: 1 % faster): 12 B * 2^27 =
         1 610 612 736 B per launch, `launches_per_step` 2; duration = HIP events (smhip_event_*, recorded on the stream the
         kernel runs on) over the timed region / (steps * launches_per_step); peak = 8000 GB/s (MI355X HBM3E spec).  `traffic` is NOT
         measured by this run: it is the HBM byte count of the last committed rocprofv3 --pmc passes
         (tools/pmc_traffic.sh -> profiles/traffic_latest.json) and `traffic_source` says so; null if that file is absent.
configs  the N = 1 line also carries BASELINE configs 3, 4 and 5's per-GPU step under "configs": {"c3", "c4", "c5_shard"}, each
         timed in TWO settings and priced against the same 8 TB/s:
           replay  every launch re-reads the same operands (the headline's setting).  For configs 3 and 4 the reads fit the
                   256 MiB Infinity Cache, so `bound` reads "hbm+infinity_cache": the rate is cache-fed.
           cold    launches rotate through K disjoint operand sets, K x footprint >= 2.5 GiB: nothing a launch reads or writes
                   was touched recently -- an operator's realistic first call.  `bound` "hbm".
         "chain" is the harness's operator chain (A * row + B) * 0.5f on 4096 x 4096 as ONE smhip_chain call -- what
         SMArray's operators queue for that expression (12 B/elem; the reference's three operator calls move 28).
         plus that config's cpu_baseline, and -- c5_shard -- the one-rank run of config 5's exchange step through libsmhip's
         own device group (smhip_set_devices(1) + smhip_sharded_contiguous_sum: the ncclAllReduce is issued for real, with
         one rank; no torch).  "rccl" says what RCCL itself reports: its version and the ranks the communicator counts
         (ncclCommCount) -- at N > 1 that is how the line shows the collective saw N ranks -- and the N > 1 line adds
         "per_gpu": [{rank, device, kernel_ms, frac}].  `--configs none` leaves the legs out; `--workload W --setting cold`
         runs one workload's whole timed region in the cold setting (what the rocprofv3 summaries under profiles/ time).
cpu_baseline  rank 0, N = 1 only, on a bounded sample: `value` = the reference's own operator path as shipped (oracle/_ref,
         kind "reference") if the prebuilt .so travelled, else the oracle's restatement (kind "port"); `best_effort` =
         the restatement on every core this process may use (cgroup quota / affinity / SMT accounted for), output
         preallocated -- the number that does not flatter the GPU.  Per workload, as BASELINE.md section 3 lists them.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BASELINE_METRIC = "Gelem/s + achieved HBM GB/s (% of peak), float32 add N=2^28, 1/2/4/8 GPU"  # BASELINE.json
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (bytes per element, description)
    "add": (12, "1D float32 add, N=2^28, contiguous (BASELINE config 2)"),
    "bcast_mul": (None, "2D float32 (4096x4096) * (1x4096) broadcast multiply (BASELINE config 3)"),
    "pow": (8, "1D float32 pow(a, 2.5), N=2^26 (BASELINE config 4)"),
    "add_sum": (12, "1D float32 fused add + sum, 2^28 per GPU (BASELINE config 5 shard)"),
    "transpose_add": (12, "2D float32 (8192x8192).T + (8192x8192): a transpose() view operand (SURVEY 8f rank 1)"),
    "chain": (12, "2D float32 (A * row + B) * 0.5 on 4096x4096, ONE smhip_chain call: the harness's chain_check (SURVEY 8f rank 4)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="add", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="ranks", choices=["ranks", "single"],
                    help="ranks = one process per GPU (started here if no launcher did); single = one process, N devices")
    ap.add_argument("--log2n", type=int, default=None, help="elements per GPU = 2^log2n (default: the config's size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier and the max over ranks: nccl = RCCL; gloo = CPU rehearsal")
    ap.add_argument("--cpu-log2n", type=int, default=26)
    ap.add_argument("--prewarm", type=float, default=0.2, help="seconds of untimed steps before the W warm-up steps (clock ramp)")
    ap.add_argument("--setting", default="replay", choices=["replay", "cold"],
                    help="replay = every step re-reads the same operands; cold = steps rotate through >= 2.5 GiB of disjoint operand sets")
    ap.add_argument("--configs", default="all", choices=["all", "none"],
                    help="all = the N=1 `add` line also times BASELINE configs 3, 4 and 5's shard (replay and cold) and the one-rank RCCL leg")
    ap.add_argument("--rank-timeout", type=float, default=900.0, help="seconds after which self-started ranks are killed")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------- CPU baseline

_USABLE = None


def usable_cores():
    """Physical cores this process may really use: affinity mask, SMT siblings and the cgroup CPU quota accounted for.
    Counted ONCE per process: after libgomp has loaded with a binding policy (the first cpu_baseline leg) the calling thread
    is pinned to one core and its affinity mask no longer says what the process may use."""
    global _USABLE
    env = os.environ.get("SMHIP_BENCH_CPU_THREADS")
    if env:
        return max(int(env), 1), dict((_USABLE or (0, {}))[1], override="SMHIP_BENCH_CPU_THREADS")
    if _USABLE is None:
        _USABLE = _count_usable_cores()
    return _USABLE[0], dict(_USABLE[1])


def _count_usable_cores():
    info = {}
    try:
        logical = len(os.sched_getaffinity(0))
    except AttributeError:
        logical = os.cpu_count() or 1
    smt = 1
    try:
        with open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list") as f:
            txt = f.read().strip()
        smt = 0
        for part in txt.split(","):
            lo, _, hi = part.partition("-")
            smt += (int(hi) - int(lo) + 1) if hi else 1
        smt = max(smt, 1)
    except (OSError, ValueError):
        pass
    cores = max(logical // smt, 1)
    info.update(logical_cpus=logical, smt=smt, physical_cores=cores)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2
            q, p = f.read().split()
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:  # cgroup v1
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        info["cgroup_cpu_quota"] = quota
        cores = max(min(cores, int(quota)), 1)
    # How many NUMA nodes the allowed CPUs span: the all-core ("best effort") figure moves with it -- the same 16-thread
    # streaming loop measured 32 Gelem/s on one GPU box and 46-60 on others: threads spread over the
    # CPUs of several sockets / NPS domains stream from several memory controllers, threads confined to one do not.
    try:
        allowed = os.sched_getaffinity(0)
        spanned = 0
        nodes = [d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()]
        for d in nodes:
            with open(f"/sys/devices/system/node/{d}/cpulist") as f:
                cpus = set()
                for part in f.read().strip().split(","):
                    if part:
                        lo, _, hi = part.partition("-")
                        cpus.update(range(int(lo), int(hi or lo) + 1))
            spanned += bool(cpus & allowed)
        info["numa_nodes"] = len(nodes)
        info["numa_nodes_spanned_by_allowed_cpus"] = spanned
    except (OSError, ValueError, AttributeError):
        pass
    return cores, info


def cpu_baseline(wl, log2n):
    """Reference / oracle timed on this box's host cores.  Checker code: used here only as the thing measured
    BESIDE the GPU, never on the product path."""
    cores, core_info = usable_cores()                  # before libgomp loads: with a binding policy it pins this thread to one core
    os.environ.setdefault("OMP_PLACES", "cores")       # read by libgomp when it is first loaded
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    import numpy as np
    from oracle import oracle as orc

    o = orc.Oracle()
    ref = orc.Reference() if orc.Reference.available() else None
    o.set_threads(cores)  # (libgomp's own default is the machine's logical CPUs, whatever the cgroup grants: not what any leg below runs with)
    res = {"unit": "Gelem/s", "threads_visible": o.num_threads(), "usable_cores": cores, "core_accounting": core_info}

    def med(fn, reps):
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return sorted(ts)[len(ts) // 2]

    def both(fn_shipped, threads_shipped, fn_best, units, reps=5):
        """(as shipped, best effort) Gelem/s; each leg warmed once."""
        out = {}
        o.set_threads(threads_shipped)
        fn_shipped()
        out["shipped"] = units / med(fn_shipped, reps) / 1e9
        o.set_threads(cores)
        fn_best()
        out["best"] = units / med(fn_best, reps) / 1e9
        return out

    if wl in ("add", "add_sum"):
        n = 1 << log2n
        o.set_threads(cores)  # parallel first touch by the threads that will stream the pages
        seeds, lo = ((1, 2), -1.0) if wl == "add" else ((6, 7), 0.0)
        a, b = o.uniform_f32(n, seeds[0], lo, 1.0), o.uniform_f32(n, seeds[1], lo, 1.0)
        out = np.empty_like(a)
        o.contiguous(orc.ADD, a, b, out=out, mt=True)
        if wl == "add":
            res["sample"] = f"1D float32 add, N=2^{log2n} of the 2^28 workload, uniform[-1,1) seeds 1/2"
            if ref is not None:
                r = both(lambda: ref.bench_add_f32(a, b), 1, lambda: o.contiguous(orc.ADD, a, b, out=out, mt=True), n)
                res.update(kind="reference", value=r["shipped"], cores=1,
                           note="SMArray<float>::operator+ (SMArray.h:217-225 -> calculate.h:101-134): single thread (the contiguous "
                                "path has no OpenMP), result new[]-allocated and first-touched inside the timed call, as benchmark/add.cpp times it")
            else:
                r = both(lambda: o.contiguous(orc.ADD, a, b, out=out), 1, lambda: o.contiguous(orc.ADD, a, b, out=out, mt=True), n)
                res.update(kind="port", value=r["shipped"], cores=1,
                           note="oracle restatement of handle_contiguous_arrays (calculate.h:101-134), single thread as the reference runs it, output preallocated")
            o.set_threads(1)
            res["port_1core_prealloc"] = n / med(lambda: o.contiguous(orc.ADD, a, b, out=out), 5) / 1e9
        else:
            res["sample"] = f"1D float32 add + fp64 sum, N=2^{log2n} of the 2^28-per-GPU workload, uniform[0,1) seeds 6/7"
            r = both(lambda: o.contiguous_sum(orc.ADD, a, b), 1, lambda: o.contiguous_sum_mt(orc.ADD, a, b, out=out), n)
            res.update(kind="port", value=r["shipped"], cores=1,
                       note="the reference has no sum(): its contiguous add (one thread, calculate.h:101-134) followed by an fp64 "
                            "accumulation pass over the result, restated in oracle/sm_oracle.c")
        res["best_effort"] = {"value": r["best"], "cores": cores,
                              "what": "oracle restatement, contiguous blocks over all usable cores (proc_bind spread), output preallocated and first-touched in parallel"}
    elif wl == "bcast_mul":
        rows = cols = 4096
        o.set_threads(cores)
        A, row = o.uniform_f32(rows * cols, 3, -1.0, 1.0), o.uniform_f32(cols, 4, -1.0, 1.0)
        res["sample"] = "the full config: (4096x4096) * (1x4096) float32, seeds 3/4"
        eng = ref if ref is not None else o
        fn = lambda: eng.elementwise(orc.MUL, A, [cols, 1], row, [0, 1], [rows, cols])
        # element_wise_op's general loop IS the all-thread form (OpenMP static over 1024-element chunks, calculate.h:47-49)
        r = both(fn, cores, fn, rows * cols, reps=3)
        res.update(kind="reference" if ref is not None else "port", value=r["best"], cores=cores,
                   note="element_wise_op<float, MultiplyOp> (calculate.h:5-99): scalar unravel with ndim div+mod per element, OpenMP over "
                        "1024-element chunks on all usable cores, result allocated per call as SMArray::operator* does")
    elif wl == "pow":
        n = 1 << min(log2n, 24)
        o.set_threads(cores)
        a = o.uniform_f32(n, 5, 0.01, 100.0)
        out = np.empty_like(a)
        res["sample"] = f"pow(a, 2.5f) on the first 2^{min(log2n, 24)} of the 2^26 elements, a in (0.01,100) seed 5"
        r = both(lambda: o.array_scalar(orc.POW, a, np.float32(2.5), out=out), 1,
                 lambda: o.array_scalar_mt(orc.POW, a, np.float32(2.5), out=out), n, reps=3)
        res.update(kind="port", value=r["best"], cores=cores, one_core=r["shipped"],
                   note="the reference's float pow has no array body that links (pow.h:12-13); its arithmetic is PowOp<float>::apply = "
                        "std::pow per element (pow.h:8-10), run here as array_scalar_op's OpenMP loop would (calculate.h:152) on all usable cores")
    elif wl == "chain":
        rows, cols = 1024, 4096  # a quarter of the 4096 x 4096 workload: the reference's general loop takes ~0.25 s per call at full size
        o.set_threads(cores)
        A, B = o.uniform_f32(rows * cols, 3, -1.0, 1.0), o.uniform_f32(rows * cols, 9, -1.0, 1.0)
        row = o.uniform_f32(cols, 4, -1.0, 1.0)
        res["sample"] = "(A * row + B) * 0.5f on 1024 x 4096 float32 (1/4 of the 4096 x 4096 workload), seeds 3/9/4"
        eng = ref if ref is not None else o
        def fn():
            t1 = eng.elementwise(orc.MUL, A, [cols, 1], row, [0, 1], [rows, cols])
            t2 = eng.elementwise(orc.ADD, t1, [cols, 1], B, [cols, 1], [rows, cols])
            return eng.array_scalar(orc.MUL, t2, np.float32(0.5))
        r = both(fn, cores, fn, rows * cols, reps=3)
        res.update(kind="reference" if ref is not None else "port", value=r["best"], cores=cores,
                   note="the three operator calls the reference makes for this expression (SMArray.h:217-305): element_wise_op's general loop "
                        "for A * row (OpenMP), its contiguous fast path for + B (one thread, calculate.h:101-134), array_scalar_op for * 0.5f "
                        "(OpenMP), a fresh result per operator")
    elif wl == "transpose_add":
        rows = cols = 2048  # a 2048 x 2048 sample of the 8192 x 8192 workload: the reference's loop takes ~1 s per call at full size
        o.set_threads(cores)
        A, B = o.uniform_f32(rows * cols, 1, -1.0, 1.0), o.uniform_f32(rows * cols, 2, -1.0, 1.0)
        res["sample"] = "A.T + B on 2048 x 2048 float32 (1/16 of the 8192 x 8192 workload), seeds 1/2"
        eng = ref if ref is not None else o
        fn = lambda: eng.elementwise(orc.ADD, A, [1, cols], B, [cols, 1], [rows, cols])
        r = both(fn, cores, fn, rows * cols, reps=3)
        res.update(kind="reference" if ref is not None else "port", value=r["best"], cores=cores,
                   note="element_wise_op<float, AddOp> (calculate.h:5-99) with the transposed view's strides {1, cols}: the general "
                        "scalar loop, OpenMP over 1024-element chunks on all usable cores, result allocated per call")
    else:
        return None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    res["cpu"] = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    res["flags"] = "reference: g++ -std=c++20 -O3 -DNDEBUG -fopenmp -mavx2 -mfma; port: gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp"
    if wl == "add":
        # BASELINE config 1: the reference's own CPU-runnable case (benchmark/add.cpp million_check, N = 1e6,
        # published 666 833 ns on a Ryzen 5 3600): reference / port on this host, same operator path
        o.set_threads(1)
        m = 1_000_000
        am, bm = o.uniform_f32(m, 1, -1.0, 1.0), o.uniform_f32(m, 2, -1.0, 1.0)
        om = np.empty_like(am)
        c1 = {"n": m, "port_1core_ns": med(lambda: o.contiguous(orc.ADD, am, bm, out=om), 200) * 1e9}
        if ref is not None:
            c1["reference_ns"] = med(lambda: ref.bench_add_f32(am, bm), 200) * 1e9
        res["config1_million_check"] = c1
        if ref is not None and hasattr(ref.lib, "ref_bench_tiny"):
            # the reference's tiny benchmarks on THIS host (BASELINE.md quotes them from a Ryzen 5 3600): the bodies of simple_check
            # (benchmark/add.cpp:4-19), BM_SMArrayPow_1D / _2D (benchmark/pow.cpp:5-28); the GPU side of the same bodies is what
            # simplemath_amd/bin/benchmark_add / benchmark_pow print (profiles/r04_small_array_breakdown.txt puts them side by side)
            res["tiny_benchmarks_reference_ns"] = {"simple_check": ref.bench_tiny(0), "BM_SMArrayPow_1D": ref.bench_tiny(1),
                                                   "BM_SMArrayPow_2D": ref.bench_tiny(2), "cores": 1}
    return res


# ----------------------------------------------------------------------------------------------- launching

class _StdoutToStderr:
    """RCCL prints a version banner on stdout when a communicator is first created; the contract is ONE JSON line there.
    While a communicator is being set up, file descriptor 1 points at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def gpu_count_without_hip():
    """How many GPUs a rank of this job would see -- asked of a CHILD process (libsmhip's smhip_device_count, i.e.
    hipGetDeviceCount under this environment's *_VISIBLE_DEVICES), so that this process, which is about to start the ranks,
    never initialises HIP itself.  None if the child could not say."""
    code = ("import ctypes, sys; l = ctypes.CDLL(sys.argv[1]); n = ctypes.c_int(0); "
            "rc = l.smhip_device_count(ctypes.byref(n)); print(n.value if rc == 0 else -1)")
    lib = os.path.join(ROOT, "simplemath_amd", "lib", "libsmhip.so")
    try:
        r = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True, timeout=120)
        return int(r.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
        return None


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) as children.  This process
    has not imported torch or touched HIP, and it never becomes a rank itself.  Fails fast and loudly: fewer GPUs than ranks
    is refused before anything starts (exit 2, one line); the first rank that exits non-zero takes its siblings with it
    (each rank is a fresh child in its own process group, killed by that exact group id) and its code is returned; the
    whole job has a deadline."""
    import signal
    if args.dist_backend == "nccl":
        have = gpu_count_without_hip()
        if have is not None and have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node shows {max(have, 0)} GPU(s) to a rank; refusing to start "
                  "(one process per GPU; use --dist-backend gloo to rehearse ranks on fewer devices)", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus))
    procs = []
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True))

    def stop_all():
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, signal.SIGTERM)  # q.pid is the id of the session / group this child leads
                except OSError:
                    pass
        t_end = time.time() + 10
        for q in procs:
            try:
                q.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(q.pid, signal.SIGKILL)
                except OSError:
                    pass
                q.wait()

    deadline = time.time() + args.rank_timeout
    rc = 0
    try:
        while True:
            codes = [q.poll() for q in procs]
            bad = [(i, c) for i, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                i, c = bad[0]
                print(f"bench.py: rank {i} exited with code {c}; stopping the other ranks", file=sys.stderr)
                rc = c if c > 0 else 128 - c  # a signal's negative code, shell style
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                print(f"bench.py: ranks still running after {args.rank_timeout:.0f} s; stopping them", file=sys.stderr)
                rc = 124
                break
            time.sleep(0.05)
    finally:
        stop_all()
    return rc


def traffic_from_profiles(wl):
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(tpath) as f:
            tj = json.load(f)
        t = tj.get(wl, {}).get("hbm_bytes_per_launch")
        meta = tj.get("_meta", {})
    except (OSError, ValueError):
        return None, None
    if t is None:
        return None, None
    src = ("NOT measured in this run: profiles/traffic_latest.json, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
           f"`bench.py --workload {wl}` (tools/pmc_traffic.sh; FETCH_SIZE doubled per the guide's gfx950 correction)")
    if meta:
        src += f"; collected {meta.get('date', '?')} at commit {meta.get('commit', '?')}"
    return t, src


COLD_ROTATION_BYTES = 2560 << 20  # a cold setting's operand sets cover at least this much (10 x the Infinity Cache)


def build_workload(lib, sma, np, C, wl, args, rank, bound, setting="replay", log2n=None):
    """(steps, units, algorithmic bytes, kernel name, workload text, keep-alive objects, extras) on the current device.
    `steps` holds one bound call per operand set: one set for `replay`, K disjoint sets for `cold` (step i uses set i % K)."""
    F32 = np.float32
    i64 = lambda seq: (C.c_int64 * len(seq))(*seq)
    log2n = log2n or args.log2n

    def sets_for(footprint):
        return 1 if setting == "replay" else max(2, -(-COLD_ROTATION_BYTES // footprint))

    steps, keep = [], []
    if wl in ("add", "add_sum"):
        log2n = log2n or 28
        n = 1 << log2n
        first = rank * n  # this rank's shard of the global array
        seeds, lo = ((1, 2), -1.0) if wl == "add" else ((6, 7), 0.0)
        sum_ptr = lib.alloc(8)
        K = sets_for(12 * n)
        for k in range(K):
            # set k > 0: other streams of the same generator (seed + 100 k): same distribution, disjoint memory
            a = lib.uniform_f32(n, seeds[0] + 100 * k, lo, 1.0, first=first)
            b = lib.uniform_f32(n, seeds[1] + 100 * k, lo, 1.0, first=first)
            c = lib.empty((n,), F32)
            keep.append((a, b, c))
            if wl == "add":
                steps.append(bound(lib.c.smhip_contiguous, C.c_int(sma.OP_ADD), C.c_int(sma.F32), C.c_void_p(a.ptr), C.c_void_p(b.ptr),
                                   C.c_void_p(c.ptr), C.c_size_t(n)))
            else:
                steps.append(bound(lib.c.smhip_contiguous_sum_async, C.c_int(sma.OP_ADD), C.c_int(sma.F32), C.c_void_p(a.ptr),
                                   C.c_void_p(b.ptr), C.c_void_p(c.ptr), C.c_size_t(n), C.c_void_p(sum_ptr)))
        pieces = lib.launch_pieces(4 * n)  # operands above 512 MiB go out as several launches (very large arrays)
        kernel = ("contiguous_vec_kernel<float, AddOp<float>, 1024, false>" if wl == "add"
                  else "reduce_kernel<float, AddOp<float>, kFused> (+ finish_kernel)")
        text = f"1D float32 {'add' if wl == 'add' else 'fused add+sum'}, N=2^{log2n} per GPU, contiguous, HBM-resident"
        extras = {"log2n": log2n, "n": n, "sum_ptr": sum_ptr}
        if wl == "add":
            extras["launches_per_step"] = pieces
        else:  # priced per STEP: its launches are two different kernels
            extras["main_launches_per_step"] = pieces
            extras["step_launches"] = f"{pieces} x reduce_kernel + 1 x finish_kernel; kernel_ms is their sum per step"
        return steps, n, 12 * n, kernel, text, keep, extras
    if wl == "bcast_mul":
        rows = cols = 4096
        r = lib.uniform_f32(cols, 4, -1.0, 1.0)
        keep.append(r)
        for k in range(sets_for(8 * rows * cols)):
            A = lib.uniform_f32(rows * cols, 3 + 100 * k, -1.0, 1.0)
            out = lib.empty((rows, cols), F32)
            keep.append((A, out))
            steps.append(bound(lib.c.smhip_elementwise, C.c_int(sma.OP_MUL), C.c_int(sma.F32), C.c_void_p(A.ptr), i64([cols, 1]),
                               C.c_void_p(r.ptr), i64([0, 1]), i64([rows, cols]), C.c_int(2), C.c_void_p(out.ptr)))
        return (steps, rows * cols, 4 * (2 * rows * cols + cols), "flat_tile_kernel<float, MultiplyOp<float>, 3, 2, *>",
                "2D float32 (4096x4096) * (1x4096) broadcast multiply, HBM-resident", keep, {})
    if wl == "chain":
        # (A * row + B) * 0.5f -- what SMArray's operators queue for that expression: ONE kernel, A and B streamed, the row
        # from the caches, the two temporaries of the reference's three operator calls (SMArray.h:217-305) never written
        rows = cols = 4096
        r = lib.uniform_f32(cols, 4, -1.0, 1.0)
        half = np.zeros(4, dtype=F32)
        half[3] = 0.5
        ops, swp = (C.c_int * 3)(sma.OP_MUL, sma.OP_ADD, sma.OP_MUL), (C.c_int * 3)(0, 0, 0)
        keep += [r, half, ops, swp]
        for k in range(sets_for(12 * rows * cols)):
            A = lib.uniform_f32(rows * cols, 3 + 100 * k, -1.0, 1.0)
            B = lib.uniform_f32(rows * cols, 9 + 100 * k, -1.0, 1.0)
            out = lib.empty((rows, cols), F32)
            ptrs = (C.c_void_p * 4)(A.ptr, r.ptr, B.ptr, None)
            strides = i64([cols, 1, 0, 1, cols, 1, 0, 0])
            keep.append((A, B, out, ptrs, strides))
            steps.append(bound(lib.c.smhip_chain, C.c_int(sma.F32), C.c_int(4), ptrs, strides, half.ctypes.data_as(C.c_void_p), ops, swp,
                               i64([rows, cols]), C.c_int(2), C.c_void_p(out.ptr)))
        return (steps, rows * cols, 4 * (3 * rows * cols + cols), "chain_kernel<float, 2, 1, 0, 1>",
                "2D float32 (A * row + B) * 0.5 on 4096x4096 as one smhip_chain call (3 operators, 1 launch), HBM-resident", keep,
                {"operators": 3, "eager_bytes_per_element": 28})
    if wl == "transpose_add":
        rows = cols = 8192
        for k in range(sets_for(12 * rows * cols)):
            A = lib.uniform_f32(rows * cols, 8 + 100 * k, -1.0, 1.0)
            B = lib.uniform_f32(rows * cols, 9 + 100 * k, -1.0, 1.0)
            out = lib.empty((cols, rows), F32)
            keep.append((A, B, out))
            steps.append(bound(lib.c.smhip_elementwise, C.c_int(sma.OP_ADD), C.c_int(sma.F32), C.c_void_p(A.ptr), i64([1, cols]),
                               C.c_void_p(B.ptr), i64([rows, 1]), i64([cols, rows]), C.c_int(2), C.c_void_p(out.ptr)))
        return (steps, rows * cols, 12 * rows * cols, "tile_kernel<float, AddOp<float>, true, 1, 0>",
                "2D float32 A.T + B, 8192x8192, A read through a transposed view, HBM-resident", keep, {})
    log2n = log2n or 26
    n = 1 << log2n
    exponent = C.c_float(2.5)
    keep.append(exponent)
    for k in range(sets_for(8 * n)):
        a = lib.uniform_f32(n, 5 + 100 * k, 0.01, 100.0)
        out = lib.empty((n,), F32)
        keep.append((a, out))
        steps.append(bound(lib.c.smhip_array_scalar, C.c_int(sma.OP_POW), C.c_int(sma.F32), C.c_void_p(a.ptr), C.byref(exponent),
                           C.c_size_t(n), C.c_void_p(out.ptr)))
    return (steps, n, 8 * n, "flat_tile_kernel<float, PowOp<float>, 1, 2, *>",
            f"1D float32 pow(a, 2.5), N=2^{log2n}, a in (0.01,100), HBM-resident", keep, {"log2n": log2n})


def bound_reads(wl, setting, alg_bytes):
    """`roofline.bound`: what feeds the kernel.  A replayed workload whose reads fit the 256 MiB Infinity Cache is cache-fed."""
    read_bytes = {"add": alg_bytes * 2 // 3, "add_sum": alg_bytes * 2 // 3, "transpose_add": alg_bytes * 2 // 3, "chain": alg_bytes * 2 // 3,
                  "bcast_mul": alg_bytes // 2, "pow": alg_bytes // 2}[wl]
    return "hbm+infinity_cache" if setting == "replay" and read_bytes <= (256 << 20) else "hbm"


def time_steps(lib, steps, n_steps, warmup, barrier=None):
    """Average launch duration in ms from HIP events on the kernels' stream, wall seconds of the timed region."""
    K = len(steps)
    for i in range(max(warmup, K)):  # every set is visited before the clock starts (first touches are not timed)
        steps[i % K]()
    e0, e1 = lib.event(), lib.event()
    (barrier or lib.synchronize)()
    t0 = time.perf_counter()
    lib.record(e0)
    for i in range(n_steps):
        steps[i % K]()
    lib.record(e1)
    (barrier or lib.synchronize)()
    wall = time.perf_counter() - t0
    ms = lib.elapsed_ms(e0, e1) / n_steps
    lib.event_destroy(e0)
    lib.event_destroy(e1)
    return ms, wall


def config_legs(lib, sma, np, C, args, bound):
    """BASELINE configs 3, 4 and 5's per-GPU step, each replayed and cold, on the current device (rank 0, N = 1)."""
    legs = {}
    for key, wl, n_steps in (("c3", "bcast_mul", 200), ("c4", "pow", 100), ("c5_shard", "add_sum", 60), ("chain", "chain", 200)):
        leg = None
        for setting in ("replay", "cold"):
            steps, units, alg_bytes, kernel, workload, keep, info = build_workload(lib, sma, np, C, wl, args, 0, bound, setting, log2n=0)
            t_pre, i_pre = time.perf_counter(), 0
            while time.perf_counter() - t_pre < 0.05:  # the clocks sag while operand sets are freed and rebuilt
                for _ in range(10):
                    steps[i_pre % len(steps)]()
                    i_pre += 1
                lib.synchronize()
            ms, _ = time_steps(lib, steps, n_steps, 20)
            if leg is None:
                leg = {"workload": workload, "kernel": kernel, "algorithmic_bytes_per_launch": alg_bytes, "elements": units}
                if "step_launches" in info:
                    leg["step_launches"] = info["step_launches"]
                    leg["algorithmic_bytes_per_step"] = leg.pop("algorithmic_bytes_per_launch")
            achieved = alg_bytes / (ms * 1e-3) / 1e9
            leg[setting] = {"bound": bound_reads(wl, setting, alg_bytes), "kernel_ms": ms, "achieved": achieved, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "value": units / (ms * 1e-3) / 1e9, "value_unit": "Gelem/s",
                            "launches": n_steps, "operand_sets": len(steps)}
            if setting == "cold":
                leg[setting]["rotating_bytes"] = len(steps) * alg_bytes
            if "sum_ptr" in info:
                lib.free(info["sum_ptr"])
            del steps, keep
            lib.synchronize()
            lib.pool_trim()
        legs[key] = leg
    return legs


def one_rank_rccl_leg(lib, sma, np, C, log2n=28):
    """Config 5's exchange step with ONE rank, through the product's own device group: smhip_set_devices(1) (RCCL loaded,
    ncclCommInitAll), then per step the fused add + sum and ONE ncclAllReduce(1 x fp64) inside ncclGroupStart/End, the scalar
    read back to the host.  Returns (leg, rccl) -- or an error text in both if RCCL is not usable here."""
    n = 1 << log2n
    try:
        with _StdoutToStderr():
            lib.set_devices(1)
    except sma.SmhipError as e:
        return {"error": str(e)}, {"error": str(e)}
    try:
        nr, rk, dev = lib.group_info(0)
        rccl = {"nranks": nr, "rank": rk, "device": dev, "version": lib.rccl_version(),
                "source": "ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion through libsmhip (smhip_group_info, smhip_rccl_version)"}
        a = lib.uniform_f32(n, 6, 0.0, 1.0)
        b = lib.uniform_f32(n, 7, 0.0, 1.0)
        c = lib.empty((n,), np.float32)
        pt = lambda x: (C.c_void_p * 1)(x.ptr)
        pa, pb, pc, ns = pt(a), pt(b), pt(c), (C.c_size_t * 1)(n)
        total = C.c_double(0)

        def step():
            rc = lib.c.smhip_sharded_contiguous_sum(C.c_int(sma.OP_ADD), C.c_int(sma.F32), pa, pb, pc, ns, C.byref(total))
            if rc < 0:
                raise sma.SmhipError(rc, lib.c.smhip_last_error().decode())

        for _ in range(3):
            step()
        lib.sharded_synchronize()
        t = time.perf_counter()
        reps = 20
        for _ in range(reps):
            step()
        lib.sharded_synchronize()
        t = (time.perf_counter() - t) / reps
        leg = {"workload": f"fused add+sum over 1 x 2^{log2n} f32 (uniform[0,1), seeds 6/7) + one all-reduce of 1 x fp64, scalar read back each step",
               "path": "smhip_set_devices(1) -> smhip_sharded_contiguous_sum -> ncclGroupStart / ncclAllReduce / ncclGroupEnd (no torch)",
               "ms_per_step": t * 1e3, "value": n / t / 1e9, "unit": "Gelem/s", "global_sum": total.value, "expected_sum_approx": float(n)}
        return leg, rccl
    finally:
        lib.set_devices(0)


def emit(args, wl, world, mode, value, ms_per_step, units, alg_bytes, kern_ms, singles, kernel, workload, c5, extra, setting="replay",
         launches_per_step=1, step_launches=None, main_launches_per_step=1):
    """kern_ms: HIP-event time of the timed region / steps.  A step of the headline is ONE smhip_contiguous call that the library
    issues as `launches_per_step` launches of the same kernel (two for 1 GiB operands): the roofline is priced per launch --
    algorithmic bytes per launch / average launch duration -- which is the same ratio, and is what a rocprofv3 kernel summary
    of the same command shows (its average duration = kernel_ms, its call count = launches_per_step x steps)."""
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    step_kern_ms = kern_ms
    alg_bytes_step = alg_bytes
    if launches_per_step > 1:
        kern_ms = kern_ms / launches_per_step
        alg_bytes = alg_bytes // launches_per_step
        singles = [x / launches_per_step for x in singles]
    traffic, traffic_source = traffic_from_profiles(wl)
    if traffic is not None and step_launches:
        traffic *= main_launches_per_step  # the PMC passes count per launch of the main kernel; this workload is priced per step
    line = {
        "metric": BASELINE_METRIC if wl == "add" else f"Gelem/s, {WORKLOADS[wl][1]}",
        "value": value, "unit": "Gelem/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "elements_per_gpu": units, "sharding": f"outer-dim x{world}, no data-path collective",
                   "kernel": kernel, "processes": "one per GPU" if mode == "ranks" else "one process, one host thread, all GPUs (smhip_set_devices)",
                   "setting": setting + (": every step re-reads the same operands" if setting == "replay"
                                         else ": steps rotate through >= 2.5 GiB of disjoint operand sets")},
        "roofline": {"bound": bound_reads(wl, setting, alg_bytes), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                     "launches_per_step": launches_per_step, "algorithmic_bytes_per_step": alg_bytes_step, "step_kernel_ms": step_kern_ms,
                     "kernel_ms_median_of_20_single_launches": singles[len(singles) // 2], "kernel_ms_min": singles[0],
                     "per_gpu": "slowest GPU's average launch" if world > 1 else "the GPU's average launch",
                     "peak_source": "MI355X HBM3E 8.0 TB/s spec (MI355X_MICROARCH.md); the guide's measured float4 copy is 6.29 TB/s"},
    }
    if step_launches:
        line["roofline"]["step_launches"] = step_launches
    if c5:
        line["c5"] = c5
    line.update(extra)
    print(json.dumps(line), flush=True)


# ------------------------------------------------------------------------------------- one process per GPU

def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    die = os.environ.get("SMHIP_BENCH_TEST_DIE_RANK")  # rehearsal hook (tests): this rank dies before the rendezvous
    if die is not None and int(die) == rank:
        print(f"bench.py: rank {rank} dying on request (SMHIP_BENCH_TEST_DIE_RANK)", file=sys.stderr)
        return 7

    if os.environ.get("SMHIP_BENCH_TEST_HANG_RANK") == str(rank):  # rehearsal hook (tests): this rank never gets anywhere
        time.sleep(3600)

    # SMHIP_BENCH_FORCE_DIST=1: take the multi-rank code path (torch.distributed, libsmhip's communicator, the config-5
    # leg) even with ONE rank -- how the one-GPU test box exercises it through the real RCCL calls.
    dist_on = world > 1 or os.environ.get("SMHIP_BENCH_FORCE_DIST") == "1"
    dist = torch = None
    if dist_on:
        # torch first: libsmhip then binds to the HIP runtime and the RCCL torch already loaded (same sonames)
        import datetime
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        limit = datetime.timedelta(seconds=120)  # a rank that never arrives fails the others here, not after the default half hour
        if args.dist_backend == "nccl":
            have = torch.cuda.device_count()
            if local_rank >= have:
                print(f"bench.py: rank {rank} wants GPU {local_rank} but this process sees {have}: one process per GPU "
                      "(--dist-backend gloo rehearses ranks on fewer devices)", file=sys.stderr)
                return 2
            torch.cuda.set_device(local_rank)
            with _StdoutToStderr():
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=limit)
                dist.barrier()  # torch creates the communicator lazily: do it now, while stdout is parked
        else:
            with _StdoutToStderr():
                dist.init_process_group("gloo", timeout=limit)
            local_rank %= max(torch.cuda.device_count(), 1)  # the gloo rehearsal may share one GPU among its ranks

    import ctypes as C
    import numpy as np
    import simplemath_amd as sma

    lib = sma.load()  # raises if the HIP library is missing: no CPU fallback
    lib.set_device(local_rank)

    use_lib_comm = dist_on and args.dist_backend == "nccl"
    comm_error = None
    if use_lib_comm:
        # libsmhip's own communicator for config 5's all-reduce: the unique id travels over torch.distributed.  A failure
        # here must not cost the headline measurement: every rank learns whether ALL ranks have a communicator, and if
        # not the config-5 leg says so and exchanges its scalar through torch.distributed instead.
        with _StdoutToStderr():
            try:
                box = [lib.comm_unique_id() if rank == 0 else None]
            except sma.SmhipError as e:
                box, comm_error = [None], str(e)
            dist.broadcast_object_list(box, src=0)
            if box[0] is None:
                comm_error = comm_error or "rank 0 could not create a communicator id"
            else:
                try:
                    lib.comm_init_rank(world, rank, box[0])
                except sma.SmhipError as e:
                    comm_error = str(e)
        flag = torch.tensor([1 if comm_error else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag[0]):
            use_lib_comm = False
            comm_error = comm_error or "another rank could not create its communicator"

    def barrier():
        lib.synchronize()
        if dist is not None:
            if args.dist_backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()
            if args.dist_backend == "nccl":
                torch.cuda.synchronize()

    def bound(fn, *cargs):
        """One step = one C-ABI call with its arguments converted once, outside the timed region
        (per-step Python glue would otherwise leave the GPU idle between short launches)."""
        def call():
            rc = fn(*cargs)
            if rc < 0:
                raise sma.SmhipError(rc, lib.c.smhip_last_error().decode())
        return call

    wl = args.workload
    steps, units, alg_bytes, kernel, workload, keep, info = build_workload(lib, sma, np, C, wl, args, rank, bound, args.setting)
    K = len(steps)

    # Clock ramp: the chip needs tens of milliseconds of continuous work to leave its idle clocks (a VALU-heavy launch
    # measures 115 us cold and 94 us ramped), and W short steps may not last that long.
    t_pre = time.perf_counter()
    i_pre = 0
    while time.perf_counter() - t_pre < args.prewarm:
        for _ in range(20):
            steps[i_pre % K]()
            i_pre += 1
        lib.synchronize()
    kern_ms, wall = time_steps(lib, steps, args.steps, args.warmup, barrier)  # barrier + synchronise on both sides of exactly K steps
    my_kern_ms = kern_ms

    coll_dev = "cuda" if (dist is not None and args.dist_backend == "nccl") else "cpu"
    per_gpu = None
    if dist is not None:
        t = torch.tensor([wall, kern_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kern_ms = float(t[0]), float(t[1])
        mine = torch.zeros(world, 2, dtype=torch.float64, device=coll_dev)  # every rank's own figures, for "per_gpu"
        mine[rank, 0], mine[rank, 1] = my_kern_ms, float(local_rank)
        dist.all_reduce(mine)
        lps = info.get("launches_per_step", 1)  # kernel_ms per LAUNCH, like roofline.kernel_ms
        per_gpu = [{"rank": r, "device": int(mine[r, 1]), "kernel_ms": float(mine[r, 0]) / lps,
                    "frac": alg_bytes / (float(mine[r, 0]) * 1e-3) / 1e9 / HBM_PEAK_GBS} for r in range(world)]

    # per-launch distribution (SURVEY 8d asks for median and min): 20 launches timed one by one, after the
    # contract's timed region so the extra event records do not touch it
    singles = []
    e0, e1 = lib.event(), lib.event()
    for i in range(20):
        lib.record(e0)
        steps[i % K]()
        lib.record(e1)
        lib.event_sync(e1)
        singles.append(lib.elapsed_ms(e0, e1))
    singles.sort()

    ms_per_step = wall / args.steps * 1e3
    value = world * units / (wall / args.steps) / 1e9

    # config 5's exchange step: fused add+sum per shard, then ONE all-reduce of the fp64 scalar
    c5 = None
    rccl = None
    if wl == "add" and dist_on:
        n, log2n = info["n"], info["log2n"]
        del keep, steps  # the headline operands go back to the pool; config 5 has its own (seeds 6/7 in [0,1))
        a5 = lib.uniform_f32(n, 6, 0.0, 1.0, first=rank * n)
        b5 = lib.uniform_f32(n, 7, 0.0, 1.0, first=rank * n)
        c5out = lib.empty((n,), np.float32)
        part = info["sum_ptr"]

        def c5_step():
            lib.contiguous_sum_async(sma.OP_ADD, a5, b5, c5out, part)  # the partial sum stays in HBM
            if use_lib_comm:
                lib.allreduce_sum_async(np.float64, part, 1)  # RCCL from libsmhip, same stream: 8 bytes per rank
                return None
            # no libsmhip communicator (gloo rehearsal, or its creation failed): the scalar goes through torch.distributed
            host = torch.tensor([lib.read_f64(part)], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(host)
            return float(host[0])

        for _ in range(3):
            total = c5_step()
        barrier()
        tc = time.perf_counter()
        reps = 20
        for _ in range(reps):
            total = c5_step()
        barrier()
        tc = (time.perf_counter() - tc) / reps
        if use_lib_comm:
            total = lib.read_f64(part)
        tt = torch.tensor([tc], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if use_lib_comm:
            backend = "libsmhip -> ncclAllReduce (RCCL over xGMI), on the kernel's stream"
            nr, rk = lib.comm_info()
            rccl = {"nranks": nr, "rank": rk, "version": lib.rccl_version(),
                    "source": "ncclCommCount / ncclCommUserRank / ncclGetVersion through libsmhip (smhip_comm_info, smhip_rccl_version), rank 0's communicator"}
        elif args.dist_backend == "gloo":
            backend = "gloo on the host (rehearsal, not RCCL)"
        else:
            backend = f"torch.distributed nccl after a host read-back -- libsmhip's communicator was not available: {comm_error}"
        c5 = {"workload": f"fused add+sum over {world} x 2^{log2n} f32 (uniform[0,1), seeds 6/7) + one all-reduce of 1 x fp64",
              "allreduce": backend, "ms_per_step": float(tt[0]) * 1e3, "value": world * n / float(tt[0]) / 1e9, "unit": "Gelem/s",
              "global_sum": total, "expected_sum_approx": float(world) * n}
        del a5, b5, c5out

    if rank == 0:
        extra = {}
        if per_gpu is not None:
            extra["per_gpu"] = per_gpu
        if rccl is not None:
            extra["rccl"] = rccl
        if world == 1 and wl == "add" and args.configs == "all" and not dist_on and args.setting == "replay":
            # the other BASELINE configs, replayed and cold, and config 5's exchange step with one rank through libsmhip's own group
            steps = keep = None
            lib.pool_trim()
            legs = config_legs(lib, sma, np, C, args, bound)
            legs["c5_shard"]["one_rank_rccl"], extra["rccl"] = one_rank_rccl_leg(lib, sma, np, C)
            lib.pool_trim()
            if not args.no_cpu_baseline:
                for key, cwl in (("c3", "bcast_mul"), ("c4", "pow"), ("c5_shard", "add_sum"), ("chain", "chain")):
                    legs[key]["cpu_baseline"] = cpu_baseline(cwl, args.cpu_log2n)
            extra["configs"] = legs
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(wl, args.cpu_log2n)
            if cb is not None:
                extra["cpu_baseline"] = cb
            if wl == "add":
                # the same million_check body on the GPU (operands resident, result from the pool)
                m = 1_000_000
                am, bm = lib.uniform_f32(m, 1, -1.0, 1.0), lib.uniform_f32(m, 2, -1.0, 1.0)
                om = lib.empty((m,), np.float32)
                small = bound(lib.c.smhip_contiguous, C.c_int(sma.OP_ADD), C.c_int(sma.F32), C.c_void_p(am.ptr), C.c_void_p(bm.ptr),
                              C.c_void_p(om.ptr), C.c_size_t(m))
                for _ in range(200):
                    small()
                lib.synchronize()
                tq = time.perf_counter()
                for _ in range(2000):
                    small()
                lib.synchronize()
                cb["config1_million_check"]["gpu_ns"] = (time.perf_counter() - tq) / 2000 * 1e9
        emit(args, wl, world, "ranks", value, ms_per_step, units, alg_bytes, kern_ms, singles, kernel, workload, c5, extra, args.setting,
             info.get("launches_per_step", 1), info.get("step_launches"), info.get("main_launches_per_step", 1))

    if use_lib_comm:
        lib.synchronize()
        lib.comm_destroy()
    if dist is not None:
        dist.destroy_process_group()
    return 0


# ----------------------------------------------------------------------------- one process, N devices

def run_single(args):
    """The product's own multi-GPU form: smhip_set_devices(N), per-device pointer tables, one host thread."""
    import ctypes as C
    import numpy as np
    import simplemath_amd as sma

    if args.workload != "add":
        raise SystemExit("--mode single runs the headline `add` workload (and its config-5 leg)")
    lib = sma.load()
    G = args.gpus
    have = lib.device_count()
    if G > have:  # before anything is allocated or any communicator is built
        print(f"bench.py: --mode single --gpus {G} but this process sees {have} GPU(s); refusing", file=sys.stderr)
        return 2
    with _StdoutToStderr():
        lib.set_devices(G)
    log2n = args.log2n or 28
    n = 1 << log2n
    F32 = np.float32

    def on_each(seed_a, seed_b, lo):
        aa, bb, cc = [], [], []
        for g in range(G):
            lib.set_device(g)
            aa.append(lib.uniform_f32(n, seed_a, lo, 1.0, first=g * n))
            bb.append(lib.uniform_f32(n, seed_b, lo, 1.0, first=g * n))
            cc.append(lib.empty((n,), F32))
        lib.set_device(0)
        return aa, bb, cc

    aa, bb, cc = on_each(1, 2, -1.0)
    pt = lambda arrs: (C.c_void_p * G)(*[x.ptr for x in arrs])
    ns = (C.c_size_t * G)(*([n] * G))
    pa, pb, pc = pt(aa), pt(bb), pt(cc)

    def step():
        rc = lib.c.smhip_sharded_contiguous(C.c_int(sma.OP_ADD), C.c_int(sma.F32), pa, pb, pc, ns)
        if rc < 0:
            raise sma.SmhipError(rc, lib.c.smhip_last_error().decode())

    def events():
        ev = []
        for g in range(G):
            lib.set_device(g)
            ev.append((lib.event(), lib.event()))
        lib.set_device(0)
        return ev

    def record(ev, which):
        for g in range(G):
            lib.set_device(g)
            lib.record(ev[g][which])
        lib.set_device(0)

    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm:
        for _ in range(20):
            step()
        lib.sharded_synchronize()
    for _ in range(args.warmup):
        step()
    ev = events()
    lib.sharded_synchronize()
    t0 = time.perf_counter()
    record(ev, 0)
    for _ in range(args.steps):
        step()
    record(ev, 1)
    lib.sharded_synchronize()
    wall = time.perf_counter() - t0
    each_ms = [lib.elapsed_ms(e0, e1) / args.steps for e0, e1 in ev]
    kern_ms = max(each_ms)

    lib.set_device(0)
    singles = []
    for _ in range(20):
        lib.record(ev[0][0])
        step()
        lib.record(ev[0][1])
        lib.sharded_synchronize()
        singles.append(lib.elapsed_ms(ev[0][0], ev[0][1]))
    singles.sort()

    # config 5: fused add+sum on every device + ONE ncclAllReduce inside ncclGroupStart/End, scalar back to the host
    del aa, bb, cc
    a5, b5, c5o = on_each(6, 7, 0.0)
    pa, pb, pc = pt(a5), pt(b5), pt(c5o)
    total = C.c_double(0)

    def c5_step():
        rc = lib.c.smhip_sharded_contiguous_sum(C.c_int(sma.OP_ADD), C.c_int(sma.F32), pa, pb, pc, ns, C.byref(total))
        if rc < 0:
            raise sma.SmhipError(rc, lib.c.smhip_last_error().decode())

    for _ in range(3):
        c5_step()
    lib.sharded_synchronize()
    tc = time.perf_counter()
    reps = 20
    for _ in range(reps):
        c5_step()
    lib.sharded_synchronize()
    tc = (time.perf_counter() - tc) / reps
    c5 = {"workload": f"fused add+sum over {G} x 2^{log2n} f32 (uniform[0,1), seeds 6/7) + one all-reduce of 1 x fp64",
          "allreduce": "smhip_sharded_contiguous_sum -> ncclGroupStart / ncclAllReduce x devices / ncclGroupEnd (RCCL), scalar read back each step",
          "ms_per_step": tc * 1e3, "value": G * n / tc / 1e9, "unit": "Gelem/s", "global_sum": total.value,
          "expected_sum_approx": float(G) * n}

    # config 3 over the group: the result's outermost dimension cut into G blocks of 4096 rows (weak: 4096 x 4096 per GPU), the
    # (1 x 4096) row -- broadcast along that dimension -- REPLICATED to every GPU device to device (smhip_copy_peer), then
    # smhip_sharded_elementwise: G independent launches, no collective
    del a5, b5, c5o
    rows = cols = 4096
    lib.set_device(0)
    row0 = lib.uniform_f32(cols, 4, -1.0, 1.0)
    A3, R3, O3 = [], [], []
    for g in range(G):
        lib.set_device(g)
        A3.append(lib.uniform_f32(rows * cols, 3, -1.0, 1.0, first=g * rows * cols))
        R3.append(row0 if g == 0 else lib.empty((cols,), F32))
        O3.append(lib.empty((rows * cols,), F32))
    lib.set_device(0)
    for g in range(1, G):
        lib.copy_peer(R3[g].ptr, g, row0.ptr, 0, 4 * cols)
    lib.sharded_synchronize()
    i64 = lambda seq: (C.c_int64 * len(seq))(*seq)
    pa3, pr3, po3 = pt(A3), pt(R3), pt(O3)
    sa3, sb3, sh3 = i64([cols, 1]), i64([0, 1]), i64([rows * G, cols])

    def c3_step():
        rc = lib.c.smhip_sharded_elementwise(C.c_int(sma.OP_MUL), C.c_int(sma.F32), pa3, sa3, pr3, sb3, sh3, C.c_int(2), po3)
        if rc < 0:
            raise sma.SmhipError(rc, lib.c.smhip_last_error().decode())

    for _ in range(5):
        c3_step()
    lib.sharded_synchronize()
    t3 = time.perf_counter()
    reps3 = 100
    for _ in range(reps3):
        c3_step()
    lib.sharded_synchronize()
    t3 = (time.perf_counter() - t3) / reps3
    # every block against the same multiply done by the single-device entry point on its own GPU (the replica must hold the row)
    block_ok = []
    for g in range(G):
        lib.set_device(g)
        want = lib.empty((rows * cols,), F32)
        lib.c.smhip_elementwise(C.c_int(sma.OP_MUL), C.c_int(sma.F32), C.c_void_p(A3[g].ptr), sa3, C.c_void_p(R3[g].ptr), sb3, i64([rows, cols]), C.c_int(2),
                                C.c_void_p(want.ptr))
        d = lib.empty((rows * cols,), F32)
        lib.contiguous(sma.OP_SUB, O3[g], want, out=d)
        block_ok.append(lib.sum(d) == 0.0 and lib.dot(d, d) == 0.0)
        del want, d
    lib.set_device(0)
    c3_bytes = 4 * (2 * rows * cols + cols)
    c5["c3_sharded"] = {"workload": f"({rows * G} x {cols}) * (1 x {cols}) float32 over {G} GPU(s): blocks of {rows} rows, the row replicated by smhip_copy_peer",
                        "entry": "smhip_sharded_elementwise (no collective)", "ms_per_step": t3 * 1e3, "value": G * rows * cols / t3 / 1e9, "unit": "Gelem/s",
                        "per_gpu_frac_of_8TBps": c3_bytes / t3 / 1e9 / HBM_PEAK_GBS, "peer_copies": G - 1, "blocks_match_single_device_result": all(block_ok),
                        "note": "host clock over 100 steps of G launches each (replayed operands: cache-fed, like configs.c3.replay)"}

    info = [lib.group_info(g) for g in range(G)]
    extra = {"rccl": {"nranks": info[0][0], "version": lib.rccl_version(), "communicators": [{"rank": r, "device": d, "nranks": nr} for nr, r, d in info],
                      "source": "ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion through libsmhip (smhip_group_info, smhip_rccl_version)"},
             "per_gpu": [{"rank": g, "device": g, "kernel_ms": each_ms[g] / lib.launch_pieces(4 * n),
                          "frac": 12 * n / (each_ms[g] * 1e-3) / 1e9 / HBM_PEAK_GBS} for g in range(G)]}
    if G == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline("add", args.cpu_log2n)
        if cb is not None:
            extra["cpu_baseline"] = cb
    emit(args, "add", G, "single", G * n / (wall / args.steps) / 1e9, wall / args.steps * 1e3, n, 12 * n, kern_ms, singles,
         "contiguous_vec_kernel<float, AddOp<float>, 1024, false>",
         f"1D float32 add, N=2^{log2n} per GPU, contiguous, HBM-resident", c5, extra, "replay", lib.launch_pieces(4 * n))
    lib.set_devices(0)
    return 0


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.mode == "single":
        return run_single(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args)  # before anything has touched the GPU
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
