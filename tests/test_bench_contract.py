"""bench.py pieces that need no GPU: the cpu_baseline leg (reference / oracle timed on host cores) and the
argument contract."""
import importlib.util
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_cpu_baseline_object(oracle):
    b = _bench()
    r = b.cpu_baseline("add", 20)  # 2^20-element sample keeps this test to a second or two
    assert r["kind"] in ("reference", "port") and r["cores"] == 1 and r["unit"] == "Gelem/s"
    assert r["value"] > 0 and r["port_1core_prealloc"] > 0
    be = r["best_effort"]  # all usable cores (cgroup quota / affinity / SMT accounted for), output preallocated
    assert be["value"] > 0 and 1 <= be["cores"] == r["usable_cores"] <= (os.cpu_count() or 1)
    assert "2^20" in r["sample"]
    c1 = r["config1_million_check"]  # BASELINE config 1: the reference's own CPU-runnable case
    assert c1["n"] == 1_000_000 and c1["port_1core_ns"] > 0


def test_cpu_baseline_for_the_other_configs(oracle):
    """BASELINE.md section 3's CPU legs beside configs 3-5: all-thread element_wise_op, all-thread std::pow, add + fp64 sum."""
    b = _bench()
    r = b.cpu_baseline("pow", 16)
    assert r["value"] > 0 and r["one_core"] > 0 and r["cores"] == r["usable_cores"] and "std::pow" in r["note"]
    r = b.cpu_baseline("add_sum", 18)
    assert r["value"] > 0 and r["best_effort"]["value"] > 0 and "seeds 6/7" in r["sample"]
    r = b.cpu_baseline("transpose_add", 16)   # the reference's general loop on a transposed view (a 2048 x 2048 sample)
    assert r["value"] > 0 and r["cores"] == r["usable_cores"] and "2048 x 2048" in r["sample"] and "calculate.h:5-99" in r["note"]
    assert b.cpu_baseline("no_such_workload", 16) is None


def test_usable_cores_accounts_for_quota(monkeypatch):
    b = _bench()
    n, info = b.usable_cores()
    assert 1 <= n <= info["logical_cpus"] and info["smt"] >= 1
    monkeypatch.setenv("SMHIP_BENCH_CPU_THREADS", "3")
    assert b.usable_cores()[0] == 3


def test_metric_string_is_baselines():
    import json
    b = _bench()
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert b.BASELINE_METRIC == json.load(f)["metric"]
    assert b.HBM_PEAK_GBS == 8000.0


def _run_bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"), timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    t = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    return r, time.time() - t


def test_more_ranks_than_gpus_is_refused_before_anything_starts():
    """`python bench.py --gpus 2` with no launcher would start one process per GPU itself.  With fewer GPUs than ranks (here:
    none) it refuses up front: exit 2, one line saying how many GPUs a rank would see -- the count comes from a child process,
    so the parent still has not touched HIP."""
    r, took = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 2, (r.returncode, r.stderr[-2000:])
    lines = [ln for ln in r.stderr.splitlines() if ln.strip()]
    assert len(lines) == 1 and "--gpus 2" in lines[0] and "0 GPU(s)" in lines[0] and "refusing to start" in lines[0], r.stderr[-2000:]
    assert r.stdout.strip() == "" and took < 120


def test_a_dying_rank_takes_its_siblings_with_it():
    """Self-started ranks (gloo rehearsal: no GPU count check): rank 1 exits with code 7 before the rendezvous; rank 0 is left
    waiting there.  The parent must notice, stop rank 0 (by its own process group) and return 7 -- at once, not after
    torch's rendezvous timeout."""
    r, took = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo", "--log2n", "20"],
                         {"SMHIP_BENCH_TEST_DIE_RANK": "1"})
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert "rank 1 exited with code 7; stopping the other ranks" in r.stderr
    assert took < 100, took  # init_process_group's own limit is 120 s; the parent does not wait for it


def test_self_started_ranks_have_a_deadline():
    r, took = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo", "--log2n", "20", "--rank-timeout", "3"],
                         {"SMHIP_BENCH_TEST_HANG_RANK": "0"})
    assert r.returncode != 0 and took < 100
    assert "still running after 3 s" in r.stderr or "exited with code" in r.stderr, r.stderr[-2000:]


def test_a_rank_without_its_gpu_fails_at_once():
    """Under a launcher (RANK in the environment) with the nccl backend a rank whose LOCAL_RANK has no GPU exits 2 with one
    line: ranks are never folded onto fewer devices (only `--dist-backend gloo` rehearsals share a GPU)."""
    r, took = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"],
                         {"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29571"}, drop=())
    assert r.returncode == 2, (r.returncode, r.stderr[-2000:])
    assert "wants GPU 1 but this process sees 0" in r.stderr and took < 200


def test_single_mode_refuses_more_devices_than_present():
    r, _ = _run_bench(["--gpus", "2", "--mode", "single", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 2 and "--mode single --gpus 2" in r.stderr and "refusing" in r.stderr, r.stderr[-2000:]


import json

import pytest


@pytest.mark.gpu
def test_two_ranks_rehearsal_on_one_gpu():
    """The multi-rank path of bench.py as the driver launches it (torch.distributed.run, one process per rank), with
    both ranks sharing this box's one GPU and gloo standing in for RCCL (RCCL refuses two ranks on one device): the
    JSON contract, the whole-job aggregate, and the config-5 object with its all-reduced sum."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--log2n", "24", "--dist-backend", "gloo"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["elements_per_gpu"] == 1 << 24 and "x2" in d["config"]["sharding"]
    assert abs(d["value"] - 2 * (1 << 24) / (d["ms_per_step"] * 1e-3) * 1e-9) < 1e-6 * d["value"]   # units of ALL ranks / max time
    assert d["roofline"]["bound"] == "hbm+infinity_cache" and 0 < d["roofline"]["frac"] < 1.5  # 2 x 64 MiB of reads replayed: cache-fed
    c5 = d["c5"]
    # config 5's operands (seeds 6/7, uniform[0,1)): 2 * 2^24 terms a + b of mean 1, variance 1/6 -- a 6-sigma band
    n = 2 * (1 << 24)
    assert abs(c5["global_sum"] - n) < 6 * (n / 6.0) ** 0.5 and c5["value"] > 0
    assert "gloo" in c5["allreduce"] and "seeds 6/7" in c5["workload"]


@pytest.mark.gpu
def test_single_process_mode_through_rccl():
    """`--mode single`: one process, the device group (here one device), the sharded entry points, and config 5's
    ncclAllReduce issued by libsmhip inside ncclGroupStart/End."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--mode", "single", "--steps", "5", "--warmup", "2",
                        "--log2n", "24", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and "one process" in d["config"]["processes"] and 0 < d["roofline"]["frac"] < 1
    n = 1 << 24
    c5 = d["c5"]
    assert abs(c5["global_sum"] - n) < 6 * (n / 6.0) ** 0.5 and "ncclAllReduce" in c5["allreduce"]


@pytest.mark.gpu
def test_rank_mode_through_rccl_with_one_rank():
    """The code path the driver's N > 1 runs take -- torch.distributed (nccl) for the barrier, libsmhip's own communicator
    (ncclGetUniqueId / ncclCommInitRank) and ncclAllReduce for config 5's scalar -- with ONE rank under the launcher."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SMHIP_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                        "--log2n", "24", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]   # ONE JSON line on stdout (RCCL's banner goes to stderr)
    d = json.loads(lines[0])
    n = 1 << 24
    c5 = d["c5"]
    assert "libsmhip -> ncclAllReduce" in c5["allreduce"], c5
    assert abs(c5["global_sum"] - n) < 6 * (n / 6.0) ** 0.5 and c5["value"] > 0
