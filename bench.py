#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

metric   "Gelem/s" of the contiguous float32 add c = a + b (BASELINE.json `metric`,
         configs[1]): one step = one pass of smhip_contiguous over N = 2^28 elements,
         inputs already resident in HBM (generated on device by the counter-based
         hash), output preallocated.
N GPUs   one process per GPU; each rank owns one outer-dimension shard of 2^28 elements
         of the N * 2^28 array (config 5's partitioning) -- elementwise work has no
         data-path collective, so scaling is "weak" and value = all ranks' elements / time.
         The reduction path of config 5 (fused add + sum, then ONE RCCL all-reduce of an
         fp64 scalar over xGMI) is timed after the headline region and reported under "c5".
roofline dominant kernel = contiguous_vec_kernel<float, AddOp<float>, 1024>;
         algorithmic bytes 12 B/elem (2 reads + 1 write) * 2^28 = 3 221 225 472 B per launch;
         duration = HIP events (smhip_event_*, recorded on the stream the kernel runs on)
         over the timed region / launches; peak = 8000 GB/s (MI355X HBM3E spec).
cpu_baseline  rank 0, N = 1 only: the reference's own operator path (oracle/_ref, kind
         "reference": SMArray<float>::operator+ exactly as benchmark/add.cpp drives it, result
         new[]-allocated inside the call, one core) if the prebuilt .so travelled, else the
         oracle's restatement (kind "port"); on a bounded 2^26-element sample.
"""
from __future__ import annotations

import argparse
import json
import os
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
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="add", choices=sorted(WORKLOADS))
    ap.add_argument("--log2n", type=int, default=None, help="elements per GPU = 2^log2n (default: the config's size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo = rehearsal of the N>1 path, e.g. 2 ranks sharing one GPU")
    ap.add_argument("--cpu-log2n", type=int, default=26)
    ap.add_argument("--prewarm", type=float, default=0.2, help="seconds of untimed steps before the W warm-up steps (clock ramp)")
    return ap.parse_args()


def cpu_baseline(log2n):
    """Reference / oracle timed on this box's host cores.  Checker code: used here only
    as the thing measured BESIDE the GPU, never on the product path."""
    import numpy as np
    from oracle import oracle as orc

    n = 1 << log2n
    o = orc.Oracle()
    a = o.uniform_f32(n, 1, -1.0, 1.0)
    b = o.uniform_f32(n, 2, -1.0, 1.0)
    out = np.empty_like(a)

    def best(fn, reps):
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return min(ts), sorted(ts)[len(ts) // 2]

    res = {"unit": "Gelem/s", "sample": f"1D float32 add, N=2^{log2n} of the 2^28 workload, uniform[-1,1) seeds 1/2",
           "threads_available": o.num_threads()}
    o.contiguous(orc.ADD, a, b, out=out)  # warm
    tmin, tmed = best(lambda: o.contiguous(orc.ADD, a, b, out=out), 5)
    res["port_1core_prealloc"] = n / tmed / 1e9
    tmin, tmed = best(lambda: o.contiguous(orc.ADD, a, b, out=out, mt=True), 5)
    res["port_allcores_prealloc"] = n / tmed / 1e9
    if orc.Reference.available():
        r = orc.Reference()
        r.bench_add_f32(a, b)
        tmin, tmed = best(lambda: r.bench_add_f32(a, b), 5)
        res.update(kind="reference", value=n / tmed / 1e9, cores=1,
                   note="SMArray<float>::operator+ (SMArray.h:217-225 -> calculate.h:101-134): single thread, "
                        "result new[]-allocated and first-touched inside the timed call, as benchmark/add.cpp times it")
    else:
        res.update(kind="port", value=res["port_1core_prealloc"], cores=1,
                   note="oracle restatement of handle_contiguous_arrays (calculate.h:101-134), single thread as the "
                        "reference runs it, output preallocated")
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    res["cpu"] = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    res["nproc"] = os.cpu_count()
    res["flags"] = "reference: g++ -std=c++20 -O3 -DNDEBUG -fopenmp -mavx2 -mfma; port: gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp"
    # BASELINE config 1: the reference's own CPU-runnable case (benchmark/add.cpp million_check, N = 1e6,
    # published 666 833 ns on a Ryzen 5 3600): reference / port on this host, same operator path
    m = 1_000_000
    am, bm = o.uniform_f32(m, 1, -1.0, 1.0), o.uniform_f32(m, 2, -1.0, 1.0)
    om = np.empty_like(am)
    c1 = {"n": m}
    tmin, tmed = best(lambda: o.contiguous(orc.ADD, am, bm, out=om), 200)
    c1["port_1core_ns"] = tmed * 1e9
    if orc.Reference.available():
        r = orc.Reference()
        tmin, tmed = best(lambda: r.bench_add_f32(am, bm), 200)
        c1["reference_ns"] = tmed * 1e9
    res["config1_million_check"] = c1
    return res


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with python -m torch.distributed.run (one rank per GPU)")

    dist = torch = None
    if world > 1:
        # torch first: libsmhip must bind to the HIP runtime torch already loaded (one runtime per process)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank %= max(torch.cuda.device_count(), 1)  # identity on a full node; lets a rehearsal share one GPU
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import numpy as np
    import simplemath_amd as sma

    lib = sma.load()  # raises if the HIP library is missing: no CPU fallback
    lib.set_device(local_rank)
    if torch is not None:
        # One explicit side stream shared by libsmhip's kernels and torch's collectives, so the all-reduce of
        # a partial sum is ordered after the kernel that produced it.  (torch's default stream is the null
        # stream, whose handle 0 libsmhip reads as "use your own stream" -- hence a real stream object.)
        side = torch.cuda.Stream()
        torch.cuda.set_stream(side)
        lib.set_stream(side.cuda_stream)

    def barrier():
        lib.synchronize()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    import ctypes as C

    def bound(fn, *cargs):
        """One step = one C-ABI call with its arguments converted once, outside the timed region
        (per-step Python glue would otherwise leave the GPU idle between short launches)."""
        def call():
            rc = fn(*cargs)
            if rc < 0:
                raise sma.SmhipError(rc, lib.c.smhip_last_error().decode())
        return call

    wl = args.workload
    F32 = np.float32
    if wl == "add" or wl == "add_sum":
        log2n = args.log2n or 28
        n = 1 << log2n
        first = rank * n  # this rank's shard of the global array
        a = lib.uniform_f32(n, 1 if wl == "add" else 6, -1.0 if wl == "add" else 0.0, 1.0, first=first)
        b = lib.uniform_f32(n, 2 if wl == "add" else 7, -1.0 if wl == "add" else 0.0, 1.0, first=first)
        c = lib.empty((n,), F32)
        sum_ptr = lib.alloc(8)
        units, alg_bytes = n, 12 * n
        if wl == "add":
            step = bound(lib.c.smhip_contiguous, C.c_int(sma.OP_ADD), C.c_int(sma.F32), C.c_void_p(a.ptr), C.c_void_p(b.ptr),
                         C.c_void_p(c.ptr), C.c_size_t(n))
            kernel = "contiguous_vec_kernel<float, AddOp<float>, 1024>"
        else:
            step = bound(lib.c.smhip_contiguous_sum_async, C.c_int(sma.OP_ADD), C.c_int(sma.F32), C.c_void_p(a.ptr), C.c_void_p(b.ptr),
                         C.c_void_p(c.ptr), C.c_size_t(n), C.c_void_p(sum_ptr))
            kernel = "reduce_kernel<float, AddOp<float>, kFused>"
        workload = f"1D float32 {'add' if wl == 'add' else 'fused add+sum'}, N=2^{log2n} per GPU, contiguous, HBM-resident"
    elif wl == "bcast_mul":
        rows = cols = 4096
        A = lib.uniform_f32(rows * cols, 3, -1.0, 1.0)
        r = lib.uniform_f32(cols, 4, -1.0, 1.0)
        A2 = sma.DeviceArray(lib, A.base_ptr, F32, (rows, cols), (cols, 1), 0, A._owner)
        r2 = sma.DeviceArray(lib, r.base_ptr, F32, (1, cols), (cols, 1), 0, r._owner)
        out = lib.empty((rows, cols), F32)
        units, alg_bytes = rows * cols, 4 * (2 * rows * cols + cols)
        i64 = lambda seq: (C.c_int64 * len(seq))(*seq)
        step = bound(lib.c.smhip_elementwise, C.c_int(sma.OP_MUL), C.c_int(sma.F32), C.c_void_p(A2.ptr), i64([cols, 1]),
                     C.c_void_p(r2.ptr), i64([0, 1]), i64([rows, cols]), C.c_int(2), C.c_void_p(out.ptr))
        kernel = "row_kernel<float, MultiplyOp<float>, 1, 1, false, true, 256, 2>"
        workload = "2D float32 (4096x4096) * (1x4096) broadcast multiply, HBM-resident"
    elif wl == "transpose_add":
        rows = cols = 8192
        A = lib.uniform_f32(rows * cols, 8, -1.0, 1.0)
        B = lib.uniform_f32(rows * cols, 9, -1.0, 1.0)
        AT = sma.DeviceArray(lib, A.base_ptr, F32, (cols, rows), (1, cols), 0, A._owner)  # A.transpose()
        B2 = sma.DeviceArray(lib, B.base_ptr, F32, (cols, rows), (rows, 1), 0, B._owner)
        out = lib.empty((cols, rows), F32)
        units, alg_bytes = rows * cols, 12 * rows * cols
        i64 = lambda seq: (C.c_int64 * len(seq))(*seq)
        step = bound(lib.c.smhip_elementwise, C.c_int(sma.OP_ADD), C.c_int(sma.F32), C.c_void_p(AT.ptr), i64([1, cols]),
                     C.c_void_p(B2.ptr), i64([rows, 1]), i64([cols, rows]), C.c_int(2), C.c_void_p(out.ptr))
        kernel = "tile_kernel<float, AddOp<float>, true, 1, 0>"
        workload = "2D float32 A.T + B, 8192x8192, A read through a transposed view, HBM-resident"
    else:  # pow
        log2n = args.log2n or 26
        n = 1 << log2n
        a = lib.uniform_f32(n, 5, 0.01, 100.0)
        out = lib.empty((n,), F32)
        units, alg_bytes = n, 8 * n
        exponent = C.c_float(2.5)
        step = bound(lib.c.smhip_array_scalar, C.c_int(sma.OP_POW), C.c_int(sma.F32), C.c_void_p(a.ptr), C.byref(exponent),
                     C.c_size_t(n), C.c_void_p(out.ptr))
        kernel = "heavy_vec_kernel<float, PowOp<float>, 1>"
        workload = f"1D float32 pow(a, 2.5), N=2^{log2n}, a in (0.01,100), HBM-resident"

    # Clock ramp: the chip needs tens of milliseconds of continuous work to leave its idle clocks (a
    # VALU-heavy launch measures 115 us cold and 94 us ramped, tools/powexp2.py), and W short steps may
    # not last that long.  Untimed pre-warm for ~0.2 s, then the contract's W warm-up steps.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm:
        for _ in range(20):
            step()
        lib.synchronize()
    for _ in range(args.warmup):
        step()
    e0, e1 = lib.event(), lib.event()
    barrier()
    t0 = time.perf_counter()
    lib.record(e0)
    for _ in range(args.steps):
        step()
    lib.record(e1)
    barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    kern_ms = lib.elapsed_ms(e0, e1) / args.steps  # average launch duration, back-to-back on one stream

    coll_dev = "cuda" if (dist is not None and args.dist_backend == "nccl") else "cpu"
    if dist is not None:
        t = torch.tensor([wall, kern_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kern_ms = float(t[0]), float(t[1])

    # per-launch distribution (SURVEY 8d asks for median and min): 20 launches timed one by one, after the
    # contract's timed region so the extra event records do not touch it
    singles = []
    for _ in range(20):
        lib.record(e0)
        step()
        lib.record(e1)
        lib.event_sync(e1)
        singles.append(lib.elapsed_ms(e0, e1))
    singles.sort()

    ms_per_step = wall / args.steps * 1e3
    value = world * units / (wall / args.steps) / 1e9

    # config 5's exchange step: fused add+sum per shard, then ONE all-reduce of the fp64 scalar
    c5 = None
    if wl == "add" and world > 1:
        part = torch.zeros(1, dtype=torch.float64, device="cuda")

        def c5_step():
            lib.contiguous_sum_async(sma.OP_ADD, a, b, c, part.data_ptr())  # partial sum stays in HBM
            if coll_dev == "cuda":
                dist.all_reduce(part)  # RCCL, same stream: 8 bytes per rank
                return part
            host = part.cpu()  # gloo rehearsal: the collective runs on the host copy
            dist.all_reduce(host)
            return host

        for _ in range(3):
            total = c5_step()
        barrier()
        tc = time.perf_counter()
        reps = 20
        for _ in range(reps):
            total = c5_step()
        barrier()
        tc = (time.perf_counter() - tc) / reps
        part = total
        tt = torch.tensor([tc], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        c5 = {"workload": f"fused add+sum over {world} x 2^{log2n} f32 + one RCCL all-reduce (1 x fp64)",
              "ms_per_step": float(tt[0]) * 1e3, "value": world * n / float(tt[0]) / 1e9, "unit": "Gelem/s",
              "global_sum": float(part[0])}

    if rank == 0:
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                traffic = tj.get(wl, {}).get("hbm_bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        line = {
            "metric": BASELINE_METRIC if wl == "add" else f"Gelem/s, {WORKLOADS[wl][1]}",
            "value": value, "unit": "Gelem/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "elements_per_gpu": units, "sharding": f"outer-dim x{world}, no data-path collective",
                       "kernel": kernel},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                         "kernel_ms_median_of_20_single_launches": singles[len(singles) // 2], "kernel_ms_min": singles[0],
                         "peak_source": "MI355X HBM3E 8.0 TB/s spec (MI355X_MICROARCH.md); the guide's measured float4 copy is 6.29 TB/s"},
        }
        if c5:
            line["c5"] = c5
        if world == 1 and not args.no_cpu_baseline and wl == "add":
            line["cpu_baseline"] = cpu_baseline(args.cpu_log2n)
            # the same million_check body on the GPU (operands resident, result from the pool)
            m = 1_000_000
            am, bm = lib.uniform_f32(m, 1, -1.0, 1.0), lib.uniform_f32(m, 2, -1.0, 1.0)
            om = lib.empty((m,), F32)
            small = bound(lib.c.smhip_contiguous, C.c_int(sma.OP_ADD), C.c_int(sma.F32), C.c_void_p(am.ptr), C.c_void_p(bm.ptr),
                          C.c_void_p(om.ptr), C.c_size_t(m))
            for _ in range(200):
                small()
            lib.synchronize()
            tq = time.perf_counter()
            for _ in range(2000):
                small()
            lib.synchronize()
            line["cpu_baseline"]["config1_million_check"]["gpu_ns"] = (time.perf_counter() - tq) / 2000 * 1e9
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
