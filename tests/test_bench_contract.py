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
