"""CPU: bench.py's host-side contract — defaults, refusal without a GPU (there is no CPU path to measure),
rank/world consistency check, PMC traffic lookup from the committed profile summary."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_defaults_and_constants(monkeypatch):
    b = _load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert a.gpus == 1 and a.envs == 65536 and a.substeps == 1 and a.gravity == "j2" and a.scenario == "bare"
    assert a.steps >= 1000 and a.warmup >= 50
    assert b.BYTES_PER_ENV_STEP == 340.0 and b.HBM_PEAK_GBS == 8000.0


def test_pmc_traffic_comes_from_committed_profile():
    b = _load_bench()
    t, src = b.pmc_traffic(65536, 1)
    assert src and src.startswith("profiles/") and os.path.exists(os.path.join(ROOT, src))
    assert 22.2e6 < t < 30e6                       # above the 22.3 MB algorithmic figure, no wasted re-reads
    big, _ = b.pmc_traffic(4194304, 1)
    assert 1.4e9 < big < 1.9e9
    assert b.pmc_traffic(65536, 1800) == (None, None)       # traffic is only profiled at K = 1
    assert b.pmc_traffic(12345, 1) == (None, None)


def test_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "no HIP device" in (res.stderr + res.stdout)
    assert not res.stdout.strip().startswith("{")            # no JSON line is fabricated


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert res.returncode != 0 and "torch.distributed.run" in (res.stderr + res.stdout)


def test_committed_bench_line_has_the_contract_keys():
    with open(os.path.join(ROOT, "profiles", "r01", "bench_final.json")) as f:
        d = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] > r["algorithmic_bytes"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert d["metric"] == json.load(f)["metric"]
