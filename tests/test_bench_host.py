"""CPU: bench.py's host-side contract — defaults, refusal without a GPU (there is no CPU path to measure),
rank/world consistency check, PMC traffic lookup from the committed profile summary."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_defaults_and_constants(monkeypatch):
    b = _load_bench()
    a = b.parse([])
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


class _FakeProp(object):
    """Records what bench.py's timing helpers ask of a propagator."""

    def __init__(self):
        self.log = []

    def step_device(self, ptr, substeps):
        self.log.append("step")

    def sync(self):
        self.log.append("sync")

    def profile_begin(self, capacity, stride=1):
        self.log.append("profile_begin(%d,%d)" % (capacity, stride))

    def profile_end(self):
        self.log.append("profile_end")
        return 0.007, self.log.count("step")


def test_timed_region_is_unstamped_and_counts_exactly_k_steps():
    """`value` must not depend on dispatch stamping: the wall-timed loop arms no events and holds exactly K
    launches between its two barrier + synchronise pairs; kernel_us comes from a separate stamped pass."""
    b = _load_bench()
    p = _FakeProp()
    marks = []
    el = b.timed_run(p, 0, 1, 20, 5, lambda: marks.append(("barrier", len(p.log))), lambda: marks.append(("devsync", len(p.log))))
    assert el >= 0 and not any(x.startswith("profile") for x in p.log)
    assert p.log.count("step") == 25
    # barrier, devsync | 20 steps | devsync, barrier  -- nothing else inside the timed window
    (b0, i0), (s0, j0), (s1, j1), (b1, i1) = marks
    assert (b0, s0, s1, b1) == ("barrier", "devsync", "devsync", "barrier") and i0 == j0 and j1 - j0 == 20 and i1 == j1
    assert p.log[j0:j1] == ["step"] * 20
    q = _FakeProp()
    ms, n, stats = b.kernel_time(q, 0, 1, 7)                 # short kernel: sampled inside a burst, stride 16
    assert q.log[:2] == ["sync", "profile_begin(9,16)"] and q.log.count("step") == 7 * 16 + 2 and q.log[-1] == "profile_end"
    q = _FakeProp()
    b.kernel_time(q, 0, 1800, 3)                             # long kernel: every launch stamped
    assert q.log[:2] == ["sync", "profile_begin(5,1)"] and q.log.count("step") == 3


def test_self_launch_composes_the_launcher_as_a_child_and_relays():
    """`python bench.py --gpus N` as typed: torch.distributed.run is started as a CHILD process (never exec),
    with the original arguments, rendezvous on 127.0.0.1, and its output / exit code are passed through."""
    import io
    b = _load_bench()
    seen = {}

    class FakeProc(object):
        stdout = io.StringIO("warning line\nEXTRA {\"extra\": 1}\n{\"metric\": \"m\"}\n")

        def wait(self):
            return 7

    def fake_popen(cmd, stdout=None, text=None, env=None):
        seen.update(cmd=cmd, env=env, stdout=stdout)
        return FakeProc()

    os.environ["RANK"] = "3"          # stale variables of an outer launcher must not leak into the child
    try:
        out, err = io.StringIO(), io.StringIO()
        rc = b.self_launch(4, ["--gpus", "4", "--steps", "20"], popen=fake_popen, out=out, err=err)
    finally:
        del os.environ["RANK"]
    # (the EXTRA line - the whole record - and, LAST, the compact line the driver reads are the child's stdout; the rest is stderr)
    assert rc == 7 and out.getvalue() == 'EXTRA {"extra": 1}\n{"metric": "m"}\n' and err.getvalue() == "warning line\n"
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "20"]
    assert "RANK" not in seen["env"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["stdout"] == subprocess.PIPE


def test_parent_of_a_self_launch_never_touches_torch_or_hip():
    """The parent must not initialise the GPU before it spawns the launcher: importing bench.py and parsing
    arguments pulls in neither torch nor the HIP library."""
    code = ("import sys, importlib.util; spec = importlib.util.spec_from_file_location('b', %r); "
            "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); m.parse(['--gpus', '8']); "
            "bad = [k for k in sys.modules if k == 'torch' or k.startswith('basilisk_env_amd')]; "
            "assert not bad, bad" % os.path.join(ROOT, "bench.py"))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr


def test_gpus_2_as_typed_reaches_the_ranks_and_relays_their_failure():
    """End to end on this GPU-less box: `python bench.py --gpus 2` starts two ranks, each refuses to run without a
    HIP device, and the parent exits non-zero without fabricating a JSON line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode != 0 and "no HIP device" in (res.stderr + res.stdout)
    assert not any(line.startswith("{") for line in res.stdout.splitlines())


def test_rooflines():
    b = _load_bench()
    info = {"name": "k", "vgprs": 200, "lds_bytes": 0, "block": 64, "grid": 1024}
    h = b.hbm_roofline(65536, 7e-6, info, 25.9e6, "profiles/x", 64)
    assert h["bound"] == "hbm" and abs(h["achieved"] - 340 * 65536 / 7e-6 / 1e9) < 1e-6
    assert abs(h["frac"] - h["achieved"] / 8000.0) < 1e-12 and abs(h["frac_of_copy_ceiling"] - h["achieved"] / 6290.0) < 1e-12
    mix, src, mix_fp = b.isa_mix("bare")
    f = b.fp64_roofline("bare", 65536.0 * 1800, 1.9e-3, info)
    assert f["bound"] == "fp64" and f["peak"] == 78.6 and f["unit"] == "TFLOP/s"
    if mix:      # executed flops from the committed counter pass: can never exceed the peak
        assert src.startswith("profiles/") and 0.0 < f["frac"] < 1.0
        assert abs(f["achieved"] - (2 * mix["fma"] + mix["mul"] + mix["add"] + mix.get("trans", 0)) * 65536 * 1800 / 1.9e-3 / 1e12) < 1e-9
        assert f["isa_mix_fresh"] == (mix_fp == b.kernel_fingerprint())     # a stale instruction mix is flagged, never silent
        assert f["isa_mix_fresh"] or "isa_mix_note" in f
    else:
        assert f["achieved"] is None
    assert b.fp64_roofline("nope", 1.0, 1.0, info)["achieved"] is None


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


def _canned_run(n_gpus=1):
    """A run dict shaped like main()'s, with every optional part present and long-winded (what pushed round 5's line over)."""
    b = _load_bench()
    info = {"name": "step_kernel<PM_J2,4,diag>", "vgprs": 244, "lds_bytes": 0, "block": 64, "grid": 1024}
    roof = b.hbm_roofline(65536, 7.02e-6, info, 25.9e6, "profiles/r06/summary_latest.json", 64)
    roof.update({"kernel_us_rule": "x" * 300, "working_set": "y" * 200, "stamping": "z" * 150})
    out = {"metric": json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"], "value": 9.1e9, "unit": "env-steps/s", "n_gpus": n_gpus,
           "steps": 20, "warmup": 5, "ms_per_step": 0.0072, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic", "config": {"workload": "w" * 400, "envs_per_gpu": 65536, "substeps": 1, "scenario": "bare",
                                           "batch_stats": "per-wave sums in the step launch; join on demand",
                                           "sharding": "env ranges, no step-path collective", "kernel_fingerprint": "0123456789abcdef"},
           "value_with_join": 6.6e9, "rk4_substeps_per_s": 9.1e9, "roofline": roof, "small_batch_crossover_n": 2,
           "cpu_baseline": {"value": 1.77e7, "unit": "env-steps/s", "cores": 16, "kind": "port", "sample": "s" * 300,
                            "single_thread": {"value": 2.74e6, "unit": "env-steps/s", "cores": 1, "sample": "t" * 100}},
           "extra": {k: {"note": "n" * 400, "roofline": dict(roof)} for k in ("k1800", "power_k1800", "full_k1800", "sh70", "large_n", "config3_per_gpu",
                                                                            "fp64_ceiling", "small_batch", "host_buffers_k1", "batch_stats_us", "rollout",
                                                                            "vecenv_episode_end_ms", "rl_loop")}}
    if n_gpus > 1:
        out["ranks"] = [{"rank": r, "hip_device": r, "name": "AMD Instinct MI355X", "pci_bus_id": "0000:%02x:00.0" % r, "uuid": "u" * 36} for r in range(n_gpus)]
        out["distinct_devices"] = n_gpus
        out["gather_ms"] = 0.21
        out["gather"] = {"shard_bytes": 2621440, "gather_to_rank0_ms": 0.3, "all_gather_ms": 0.21, "direct_d2h_per_gpu_ms": 0.1,
                         "direct_rccl_gather7_to_rank0_ms": 0.2, "direct_rccl_gather7_rank_major_ms": 0.15, "all_reduce_stats_ms": 0.03,
                         "nccl_comm_count": n_gpus, "direct_rccl_form": "f" * 300, "bytes": {"a": 1}, "messages_on_root": {"x": 14}}
        out["extra"]["strong_65536_total"] = {"envs_per_gpu": 65536 // n_gpus, "k1_env_steps_per_s": 5e9, "k1800_env_steps_per_s": 3e7,
                                              "full_k1800_env_steps_per_s": 2.6e7, "full_k1800_kernel": "k" * 60}
        out["extra"]["config3"] = {"env_steps_per_s": 1.1e11, "gather": {"all_gather_ms": 1.0}}
    return b, out


def test_headline_line_is_small():
    """the LAST stdout line is a compact JSON object (< 4 096 bytes, the driver keeps a bounded tail of stdout) with
    the contract keys, the reduced roofline and the CPU baseline; the whole record goes to an EARLIER line prefixed 'EXTRA ' and
    to bench_extra.json."""
    import io
    import tempfile
    for n_gpus in (1, 8):
        b, out = _canned_run(n_gpus)
        assert len(json.dumps(out)) > 20000                   # the record itself is as large as round 5's
        line = b.headline_line(out)
        assert len(line) < b.HEADLINE_LIMIT == 4096 and "\n" not in line
        d = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline", "value_with_join"):
            assert k in d, k
        assert d["dtype"] == "f64" and d["vs_baseline"] is None and d["scaling"] == "weak" and d["n_gpus"] == n_gpus
        assert 0 < len(d["config"]["workload"]) <= 200 and "model" not in d["config"]
        assert d["config"]["batch_stats"] == "per-wave sums in the step launch; join on demand"
        r = d["roofline"]
        assert set(r) <= set(b.ROOFLINE_KEYS) and r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
        assert abs(r["achieved"] - r["algorithmic_bytes"] / r["kernel_us"] / 1e3) < 1e-2 * r["achieved"] and r["traffic"] > r["algorithmic_bytes"]
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] == 16 and c["value"] > 0 and c["single_thread"]["value"] > 0 and len(c["sample"]) <= 160
        assert "extra" not in d or isinstance(d["extra"], str)   # the sub-reports never ride on the headline
        if n_gpus > 1:
            assert d["ranks"] == n_gpus == d["distinct_devices"] == d["gather"]["nccl_comm_count"] and d["gather_ms"] == 0.21
            assert {"torch_gather_ms", "direct_7row_ms", "direct_rank_major_ms"} <= set(d["gather"])
            assert d["strong_65536_total"]["full_k1800_env_steps_per_s"] == 2.6e7 and d["strong_65536_total"]["k1_env_steps_per_s"] == 5e9
        # emit(): EXTRA line first, compact line LAST, the record in the file
        buf, path = io.StringIO(), os.path.join(tempfile.mkdtemp(), "bench_extra.json")
        b.emit(out, stream=buf, extra_path=path)
        lines = buf.getvalue().splitlines()
        assert len(lines) == 2 and lines[0].startswith(b.EXTRA_PREFIX) and lines[1] == line
        assert json.loads(lines[0][len(b.EXTRA_PREFIX):]) == json.load(open(path)) == json.loads(json.dumps(out))
    # a record with traffic unknown keeps the key (null), as the contract says
    b, out = _canned_run(1)
    out["roofline"]["traffic"] = None
    assert json.loads(b.headline_line(out))["roofline"]["traffic"] is None


def test_committed_headline_of_this_round_is_small_and_complete():
    """The line committed under profiles/ for the driver's own command (tools/round.sh TAG lines) obeys the limit."""
    path = os.path.join(ROOT, "profiles", "r06", "bench_steps20_warmup5.json")
    if not os.path.exists(path):
        pytest.skip("no committed round-6 bench line yet")
    text = open(path).read().strip().splitlines()
    assert len(text[-1]) < 4096
    d = json.loads(text[-1])
    assert d["dtype"] == "f64" and d["roofline"]["bound"] == "hbm" and d["cpu_baseline"]["kind"] == "port" and d["value_with_join"] > 0
    assert d["value"] > d["value_with_join"] and len(d["config"]["workload"]) <= 200


def test_roofline_is_priced_on_the_more_conservative_duration(monkeypatch):
    """bench.settle_roofline: frac uses max(stamped pass, committed rocprofv3 steady-state mean of the same command),
    and the committed figure only when it was measured on this very kernel source; every estimate stays in the object."""
    b = _load_bench()
    info = {"name": "k", "vgprs": 200, "lds_bytes": 0, "block": 64, "grid": 1024}
    fp = b.kernel_fingerprint()
    assert len(fp) == 16 and fp == b.kernel_fingerprint()
    work = 340.0 * 65536

    def roof():
        return b.hbm_roofline(65536, 6.6e-6, info, None, None, 64)

    monkeypatch.setattr(b, "rocprof_kernel", lambda key: {"trimmed_mean_us": 7.5, "median_us": 7.4, "source": "profiles/rXX/kernel_trace.json",
                                                          "csv": "kt_65k", "fingerprint": fp})
    r = b.settle_roofline(roof(), "65k_k1", 6.6, 7.7, work, 8000.0, fp)
    assert r["kernel_us"] == 7.5 and r["kernel_us_stamped"] == 6.6 and r["wall_us_per_launch"] == 7.7 and r["kernel_us_rocprof_fresh"]
    assert abs(r["frac"] - work / 7.5 / 1e3 / 8000.0) < 1e-12 and abs(r["frac_stamped"] - work / 6.6 / 1e3 / 8000.0) < 1e-12
    assert r["frac"] < r["frac_stamped"]
    monkeypatch.setattr(b, "rocprof_kernel", lambda key: {"trimmed_mean_us": 9.9, "median_us": 9.9, "source": "profiles/rXX/kernel_trace.json",
                                                          "csv": "kt_65k", "fingerprint": "0" * 16})
    r = b.settle_roofline(roof(), "65k_k1", 6.6, 7.7, work, 8000.0, fp)        # profile of another kernel: shown, not used
    assert r["kernel_us"] == 6.6 and r["kernel_us_rocprof"] == 9.9 and not r["kernel_us_rocprof_fresh"] and "kernel_us_rocprof_note" in r
    # a command traced on several boxes: the median of their trimmed means is the profile figure
    monkeypatch.setattr(b, "rocprof_kernel", lambda key: {"trimmed_mean_us": 7.9, "median_us": 7.8, "source": "profiles/rXX/kernel_trace.json",
                                                          "csv": "kt_65k", "fingerprint": fp, "trimmed_mean_us_median_of_boxes": 7.3,
                                                          "boxes": [{"trimmed_mean_us": 7.9}, {"trimmed_mean_us": 7.1}, {"trimmed_mean_us": 7.3}]})
    r = b.settle_roofline(roof(), "65k_k1", 6.6, 7.7, work, 8000.0, fp)
    assert r["kernel_us"] == 7.3 and r["kernel_us_rocprof"] == 7.3 and r["kernel_us_rocprof_boxes"] == [7.9, 7.1, 7.3]
    monkeypatch.setattr(b, "rocprof_kernel", lambda key: None)
    r = b.settle_roofline(roof(), None, 6.6, 7.7, work, 8000.0, fp)
    assert r["kernel_us"] == 6.6 and r["kernel_us_rocprof"] is None
    f = {"bound": "fp64", "unit": "TFLOP/s", "peak": 78.6}
    r = b.settle_roofline(f, None, 3700.0, 3710.0, 1.5e11, 78.6, fp)
    assert abs(r["achieved"] - 1.5e11 / 3700.0 / 1e6) < 1e-9 and abs(r["frac"] - r["achieved"] / 78.6) < 1e-12


def test_committed_kernel_trace_is_consistent_with_its_dispatch_files():
    """profiles/rNN/kernel_trace.json: every figure can be recomputed from the per-dispatch CSVs committed beside it, and
    where a key was traced on several boxes the median of their trimmed means is what it offers."""
    import csv
    b = _load_bench()
    path = b._latest_profile("kernel_trace.json")
    d = json.load(open(path))
    base = os.path.dirname(path)
    for key, rec in d["runs"].items():
        for box in rec.get("boxes", [rec]):
            f = os.path.join(base, box["csv"])
            if not os.path.exists(f):                      # (since round 6 the raw per-dispatch files sit in the round's traces/ sub-directory)
                f = os.path.join(base, "traces", box["csv"])
            rows = list(csv.DictReader(open(f)))
            dur = [(float(r["start_offset_us"]), int(r["duration_ns"])) for r in rows]
            n_ramp = min(sum(1 for s, _ in dur if s < 25.0 * 1e3), len(dur) // 2)
            steady = sorted(x for _, x in dur[n_ramp:])
            cut = len(steady) // 10
            core = steady[cut:len(steady) - cut] if len(steady) >= 10 else steady
            assert abs(sum(core) / len(core) / 1e3 - box["trimmed_mean_us"]) < 1e-6 * box["trimmed_mean_us"], (key, box["csv"])
        if "boxes" in rec:
            tm = sorted(x["trimmed_mean_us"] for x in rec["boxes"])
            med = tm[len(tm) // 2] if len(tm) % 2 else 0.5 * (tm[len(tm) // 2 - 1] + tm[len(tm) // 2])
            assert rec["trimmed_mean_us_median_of_boxes"] == med and rec["boxes"][0]["trimmed_mean_us"] == rec["trimmed_mean_us"]


def test_profile_keys():
    b = _load_bench()
    assert b.profile_key(b.parse([]), False) == "65k_k1"
    assert b.profile_key(b.parse(["--scenario", "full", "--substeps", "1800"]), False) == "full_k1800"
    assert b.profile_key(b.parse(["--substeps", "1800"]), False) == "bare_k1800"
    assert b.profile_key(b.parse(["--gravity", "sh"]), True) == "sh70"
    assert b.profile_key(b.parse(["--envs", "4194304"]), False) == "4m_k1"
    assert b.profile_key(b.parse(["--envs", "1000"]), False) is None
    assert b.profile_key(b.parse(["--features", "power"]), False) is None


def test_with_deadline_returns_fn_result_and_fires_only_on_overrun():
    """The watchdog around the auxiliary RCCL leg: quiet when the leg returns in time (and it must not fire later), loud
    when it does not; an exception in the leg is passed on and disarms it too."""
    import threading
    import time
    b = _load_bench()
    fired = threading.Event()
    assert b.with_deadline(0.2, fired.set, lambda: 41 + 1) == 42
    time.sleep(0.4)
    assert not fired.is_set()
    assert b.with_deadline(0.05, fired.set, lambda: (time.sleep(0.3), "late")[1]) == "late"
    assert fired.is_set()
    fired.clear()
    try:
        b.with_deadline(0.2, fired.set, lambda: 1 // 0)
    except ZeroDivisionError:
        pass
    else:
        raise AssertionError("the leg's exception was swallowed")
    time.sleep(0.4)
    assert not fired.is_set()


def test_a_hung_collective_leg_ends_in_status_3_with_exactly_one_json_line():
    """Rehearsal of the watchdog's failure path (bench.py: guarded_leg): a deliberately hung auxiliary leg
    (BENCH_FAULT_HANG_LEG=1 swaps it for one that never returns) must neither take the bench line with it nor look like
    success - rank 0 prints the ONE JSON line, the process ends with status 3 (os._exit: no re-exec, no atexit work on a
    process that has touched the GPU), every other rank prints nothing and ends with 3 as well."""
    code = "\n".join([
        "import importlib.util, sys, os",
        "spec = importlib.util.spec_from_file_location('bench_module', %r)" % os.path.join(ROOT, "bench.py"),
        "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)",
        "out = {'metric': 'm', 'value': 1.0, 'gather': {'all_gather_ms': 0.5}}",
        "b.guarded_leg(out, int(sys.argv[1]), 0.3, lambda: {'never': 'reached'})",
        "print('not reached')"])
    import tempfile
    tmp = tempfile.mkdtemp()
    env = dict(os.environ, BENCH_FAULT_HANG_LEG="1", BENCH_EXTRA_FILE=os.path.join(tmp, "bench_extra.json"))
    for rank in (0, 1):
        res = subprocess.run([sys.executable, "-c", code, str(rank)], capture_output=True, text=True, timeout=60, env=env)
        assert res.returncode == 3, (res.returncode, res.stderr[-500:])
        lines = [l for l in res.stdout.splitlines() if l.strip()]
        if rank == 0:
            # the whole record as an EXTRA line (and in bench_extra.json), then - last - the ONE compact JSON line
            assert len(lines) == 2 and lines[0].startswith("EXTRA {") and lines[1].startswith("{"), lines
            full, d = json.loads(lines[0][len("EXTRA "):]), json.loads(lines[1])
            assert full["value"] == 1.0 and full["gather"]["all_gather_ms"] == 0.5 and "timeout after 0.3 s" in full["gather"]["direct_rccl"]
            assert d["value"] == 1.0 and d["gather"]["torch_all_gather_ms"] == 0.5 and "timeout after 0.3 s" in d["gather"]["direct_rccl"]
            assert json.load(open(env["BENCH_EXTRA_FILE"])) == full
        else:
            assert lines == []
    # ... and a leg that returns in time is merged into the line, its byte counts beside the others', no exit
    b = _load_bench()
    out = {"gather": {"bytes": {"all_gather": 10}}}
    os.environ.pop("BENCH_FAULT_HANG_LEG", None)
    b.guarded_leg(out, 0, 5.0, lambda: {"x_ms": 1.0, "bytes": {"direct": 7}}, exit_fn=lambda c: (_ for _ in ()).throw(AssertionError("fired")))
    assert out["gather"] == {"bytes": {"all_gather": 10, "direct": 7}, "x_ms": 1.0}
    out = {}
    b.guarded_leg(out, 0, 5.0, lambda: 1 // 0)
    assert "ZeroDivisionError" in out["gather"]["direct_rccl_error"]


def test_round_entry_point_parses_and_refuses_unknown_phases():
    """tools/round.sh is the single entry of a round's evidence pass: it must at least parse, know its phases, and not
    start anything for a phase it does not know (a typo must not cost a GPU call)."""
    sh = os.path.join(ROOT, "tools", "round.sh")
    assert subprocess.run(["bash", "-n", sh]).returncode == 0
    assert subprocess.run(["bash", "-n", os.path.join(ROOT, "tools", "collect_evidence.sh")]).returncode == 0
    res = subprocess.run(["bash", sh, "rXX_test", "nonsense"], capture_output=True, text=True, env=dict(os.environ, GRAFT_REPO_ROOT="/tmp/bsk_round_test"))
    assert res.returncode == 2 and "unknown phase" in res.stdout and "tests traces stats counters isa lines collect" in res.stdout
    text = open(sh).read()
    for ph in ("tests", "traces", "stats", "counters", "isa", "lines", "collect"):
        assert "phase_%s()" % ph in text
