"""bench.py pieces that need no GPU: the cpu_baseline leg (reference / oracle timed on host cores) and the
argument contract."""
import importlib.util
import os
import subprocess
import sys

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
    r = b.cpu_baseline(20)  # 2^20-element sample keeps this test to a second or two
    assert r["kind"] in ("reference", "port") and r["cores"] == 1 and r["unit"] == "Gelem/s"
    assert r["value"] > 0 and r["port_1core_prealloc"] > 0 and r["port_allcores_prealloc"] > 0
    assert "2^20" in r["sample"]
    c1 = r["config1_million_check"]  # BASELINE config 1: the reference's own CPU-runnable case
    assert c1["n"] == 1_000_000 and c1["port_1core_ns"] > 0


def test_metric_string_is_baselines():
    import json
    b = _bench()
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert b.BASELINE_METRIC == json.load(f)["metric"]
    assert b.HBM_PEAK_GBS == 8000.0


def test_multi_gpu_needs_torchrun():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)


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
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    c5 = d["c5"]
    # sum over both shards of (a + b), a and b uniform[-1,1): 2 * 2^24 terms of mean 0, variance 2/3 -- a 6-sigma band
    n = 2 * (1 << 24)
    assert abs(c5["global_sum"]) < 6 * (n * 2.0 / 3.0) ** 0.5 and c5["value"] > 0
