#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched HIP propagator on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: as typed (this process starts ``python -m torch.distributed.run --nproc-per-node N
bench.py ...`` as a CHILD, before anything here has touched the GPU, relays its output and exits with its
code) or already under ``torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE in the environment).

Workload (BASELINE.json configs[2], the one the metric is quoted on): 65 536 spacecraft PER GPU,
J2 gravity + 4 reaction wheels (pyramid) + nadir-pointing reward (action 0), fp64, synthetic
random-orbit batch, one RK4 sub-step of 0.1 s per env step (the HBM framing
of the metric).  Envs are independent: ranks shard them with no collective on the step path
("scaling": "weak"); the one exchange step — delivering the observation batch — is timed after
the timed region (``gather``: RCCL gather to rank 0, RCCL all-gather, and each GPU's direct D2H).

A "step" = one pass of the hot path over the whole batch (one kernel launch per GPU): mode
switch, FSW chain when due, RK4, observation, reward, done mask, and the wave-level reductions (the done
ballot and the per-wave reward sums: bsk_set_step_stats is on).  Actions and state are resident in HBM when
the timed region starts.  ``value_with_join`` is the same loop with the two batch scalars joined on the
device after every step (bsk_get_batch_stats_device).

Output.  The LAST stdout line is the compact JSON record the driver reads (< 4 096 bytes: contract keys,
``value_with_join``, ``roofline``, ``cpu_baseline``, at N > 1 the rank / device / communicator counts and the
exchange timings).  The whole record of the run - every other measurement point - is an EARLIER line
prefixed ``EXTRA `` and ``bench_extra.json`` beside this file.  (``--full-line``: tooling gets the whole record
as the one line instead.)

Timing.  ``value`` / ``ms_per_step`` come from the wall clock around EXACTLY ``--steps`` launches that
carry no timestamps (nothing but the launches is enqueued between the two synchronisations), so they do
not depend on ``--steps``.  ``roofline.kernel_us`` comes from a separate pass afterwards: dispatch stamps
(hipExtLaunchKernelGGL start/stop events on the launch stream) on a sample of the launches of a back-to-back burst,
bounded from above by the timed loop's wall time per launch (launches of one stream cannot overlap; ``kernel_us_source``
says which of the two is reported).  ``--scenario`` / ``--features`` / ``--fsw-timing`` / ``--lds-scratch`` select
other kernels of the same path for measurement; the default line is the contract's.

Besides the contract keys the record carries ``roofline`` (dominant kernel: HBM-bound at K = 1 with
340 algorithmic bytes per env-step; fp64-issue-bound for K >> 1, the harmonics and the
scenario levels, with EXECUTED flops from the committed SQ_INSTS_VALU_*_F64 counter passes),
``cpu_baseline`` (the CPU oracle on this box's host cores, rank 0 at N = 1 only) and ``extra`` (EXTRA line only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_ENV_STEP = 340.0   # config 3/4 algorithmic bytes per env-step
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0  # MI355X_MICROARCH.md: measured copy ceiling (SURVEY.md §8(d) asks for both)
FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector peak (FMA = 2 flop), per GPU
STAMPED_LAUNCHES = 64        # launches of the separate dispatch-stamped pass


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10000)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--envs", type=int, default=65536, help="spacecraft per GPU")
    p.add_argument("--substeps", type=int, default=1, help="RK4 sub-steps per env step")
    p.add_argument("--gravity", choices=["j2", "sh"], default="j2",
                   help="j2 = BASELINE configs[2] (headline); sh = configs[4], degree-70 spherical harmonics")
    p.add_argument("--scenario", choices=["bare", "power", "full"], default="bare",
                   help="bare = BASELINE configs[2] as named (headline); power / full add the reference scenario's "
                        "power system / + Sun third body, drag and desaturation (what the drop-in env runs)")
    p.add_argument("--lds-scratch", action="store_true", help="BSK_FLAG_LDS_SCRATCH kernel variant (RK4 accumulator in LDS)")
    p.add_argument("--features", default=None,
                   help="measurement only: comma list out of power,sun,drag,desat replacing the scenario's feature flags "
                        "(any of sun/drag/desat selects the full-scenario kernel)")
    p.add_argument("--fsw-timing", choices=["reference", "same-tick"], default="reference",
                   help="reference: bsk_config.fsw_lag = nav_lag = 1 (the reference's task order and priorities); "
                        "same-tick: both 0 (measurement A/B only)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra", action="store_true")
    p.add_argument("--no-join", action="store_true", help="measurement tooling only: skip the second timed loop (value_with_join)")
    p.add_argument("--full-line", action="store_true",
                   help="measurement tooling only: print the whole record as the ONE stdout line (no EXTRA line, no compact line, no "
                        "bench_extra.json) - what tools/*.sh parse; the driver's commands never pass it")
    return p.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# The ONE line the driver reads is the LAST line of stdout and stays small (the driver keeps a bounded tail of stdout: round 5's
# 20.9 KB line did not parse).  Everything else of the run - the other measurement points, sample statistics, prose - goes to
# bench_extra.json beside this file and to an EARLIER stdout line prefixed "EXTRA ".
HEADLINE_LIMIT = 4096
EXTRA_PREFIX = "EXTRA "
EXTRA_FILE = "bench_extra.json"
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_us", "algorithmic_bytes", "traffic", "traffic_source",
                 "working_set_bytes", "frac_of_copy_ceiling", "flop_per_launch")


def _sig(x, digits=6):
    """Floats of the headline at ``digits`` significant figures (the full-precision record is the EXTRA line)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float("%.*g" % (digits, x))


def _pick(d, keys, keep_null=("traffic",)):
    return {k: _sig(d[k]) for k in keys if k in d and (d[k] is not None or k in keep_null)}


def headline(out):
    """-> dict: the contract keys + ``roofline`` reduced to what a reader recomputes from + ``cpu_baseline`` + (ranks > 1) the
    rank / device / communicator counts and the exchange timings; ``json.dumps`` of it is < HEADLINE_LIMIT bytes by
    construction (tests/test_bench_host.py::test_headline_line_is_small)."""
    cfg = out.get("config", {})
    h = {k: _sig(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data") if k in out}
    h["config"] = {"workload": str(cfg.get("workload", ""))[:200]}
    for k in ("envs_per_gpu", "substeps", "scenario", "batch_stats", "sharding", "kernel_fingerprint"):
        if cfg.get(k) is not None:
            h["config"][k] = cfg[k]
    for k in ("value_with_join", "rk4_substeps_per_s"):
        if out.get(k) is not None:
            h[k] = _sig(out[k])
    if "roofline" in out:
        h["roofline"] = _pick(out["roofline"], ROOFLINE_KEYS)
    cb = out.get("cpu_baseline")
    if cb:
        h["cpu_baseline"] = {k: _sig(cb[k]) for k in ("value", "unit", "cores", "kind") if k in cb}
        h["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
        if cb.get("single_thread"):
            h["cpu_baseline"]["single_thread"] = {"value": _sig(cb["single_thread"]["value"])}
    ex = out.get("extra", {})
    if out.get("small_batch_crossover_n") is not None:
        h["small_batch_crossover_n"] = out["small_batch_crossover_n"]
    if out.get("n_gpus", 1) > 1 or "gather" in out:
        if "ranks" in out:
            h["ranks"] = len(out["ranks"])
            h["distinct_devices"] = out.get("distinct_devices")
        g = out.get("gather", {})
        keep = {"gather_to_rank0_ms": "torch_gather_ms", "all_gather_ms": "torch_all_gather_ms", "direct_rccl_gather7_to_rank0_ms": "direct_7row_ms",
                "direct_rccl_gather7_rank_major_ms": "direct_rank_major_ms", "all_reduce_stats_ms": "all_reduce_stats_ms",
                "nccl_comm_count": "nccl_comm_count", "direct_rccl": "direct_rccl", "direct_rccl_error": "direct_rccl_error"}
        hg = {v: (_sig(g[k]) if not isinstance(g[k], str) else g[k][:120]) for k, v in keep.items() if k in g}
        if hg:
            h["gather"] = hg
        if "gather_ms" in out:
            h["gather_ms"] = _sig(out["gather_ms"])
        st = ex.get("strong_65536_total")
        if st:
            h["strong_65536_total"] = {"envs_per_gpu": st.get("envs_per_gpu"), "k1_env_steps_per_s": _sig(st.get("k1_env_steps_per_s")),
                                       "full_k1800_env_steps_per_s": _sig(st.get("full_k1800_env_steps_per_s"))}
        c3 = ex.get("config3")
        if c3:
            h["config3_env_steps_per_s"] = _sig(c3.get("env_steps_per_s"))
    h["extra"] = EXTRA_FILE + " / the 'EXTRA ' line above"
    return h


def headline_line(out):
    """The compact line as text; if a future key ever pushes it over the limit the optional parts go first, never the contract keys."""
    h = headline(out)
    line = json.dumps(h)
    for k in ("gather", "strong_65536_total", "small_batch_crossover_n", "rk4_substeps_per_s", "extra"):
        if len(line) < HEADLINE_LIMIT:
            break
        h.pop(k, None)
        line = json.dumps(h)
    return line


FULL_LINE = False       # --full-line (set by main): tooling gets the record as one line


def emit(out, stream=None, extra_path=None):
    """Rank 0's output: bench_extra.json + the EXTRA line (the whole record), then - LAST - the compact line."""
    stream = stream or sys.stdout
    full = json.dumps(out)
    if FULL_LINE and extra_path is None:
        stream.write(full + "\n")
        stream.flush()
        return
    try:
        with open(extra_path or os.environ.get("BENCH_EXTRA_FILE") or os.path.join(ROOT, EXTRA_FILE), "w") as f:
            f.write(full + "\n")
    except OSError as e:          # a read-only tree must not cost the line
        sys.stderr.write("bench: could not write %s: %r\n" % (EXTRA_FILE, e))
    stream.write(EXTRA_PREFIX + full + "\n")
    stream.write(headline_line(out) + "\n")
    stream.flush()


# ---------------------------------------------------------------------------------------------
# N > 1 as typed: spawn the launcher as a child.  Nothing above or in here imports torch or touches HIP, so the
# parent never initialises the GPU (a process that has must not exec or be replaced; it may start children).
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(gpus, argv, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(gpus, argv, popen=subprocess.Popen, out=None, err=None):
    """Run ``bench.py argv`` on ``gpus`` ranks in a child ``torch.distributed.run``; relay its stdout line by
    line (rank 0 prints the EXTRA line and, last, the one compact JSON line) and return its exit code."""
    out = out or sys.stdout
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    proc = popen(launcher_command(gpus, argv, _free_port()), stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:
        # the bench line goes to stdout; whatever else the ranks or their libraries print there (gloo / RCCL
        # banners) is passed on through stderr, so that stdout carries exactly the one JSON line
        dst = out if line.lstrip().startswith(("{", EXTRA_PREFIX)) else (err or sys.stderr)
        dst.write(line)
        dst.flush()
    return proc.wait()


# ---------------------------------------------------------------------------------------------
def timed_run(prop, d_act_ptr, substeps, steps, warmup, barrier, device_sync, join=False):
    """Wall time of exactly ``steps`` un-stamped launches between two barrier + device synchronisations.  ``join``: every step is
    followed by bsk_get_batch_stats_device (the batch's sum of rewards / number of done envs joined on the device, no
    synchronisation): ``value_with_join``."""
    for _ in range(warmup):
        prop.step_device(d_act_ptr, substeps)
        if join:
            prop.batch_stats_device()
    prop.sync()
    barrier()
    device_sync()
    t0 = time.perf_counter()
    if join:
        for _ in range(steps):
            prop.step_device(d_act_ptr, substeps)
            prop.batch_stats_device()
    else:
        for _ in range(steps):
            prop.step_device(d_act_ptr, substeps)
    device_sync()
    t1 = time.perf_counter()   # this rank's K steps are done; the MAX over ranks is taken by the caller
    barrier()
    return t1 - t0


def kernel_time(prop, d_act_ptr, substeps, launches, stride=None):
    """Duration of the step kernel from dispatch stamps, in its own pass after the wall-timed region:
    -> (mean ms, launches counted, stats dict).

    Short kernels are sampled INSIDE a back-to-back burst: a pair of launches is stamped every ``stride`` launches
    and the second of each pair counted (bsk_profile_set_stride), ``launches`` samples in all.  Stamping every
    launch of a pass that follows a synchronisation measures something else — each kernel then starts on an idle
    device after the ~5 us stamping gap, and reads up to 60 % long once the clocks have settled after a long
    burst.  Long kernels (many sub-steps per launch) are stamped one by one (stride 1)."""
    if stride is None:
        stride = 16 if substeps < 16 else 1
    prop.sync()
    prop.profile_begin(launches + 2, stride=stride)
    for _ in range(launches * stride + (2 if stride > 1 else 0)):
        prop.step_device(d_act_ptr, substeps)
    if hasattr(prop, "profile_end_samples"):
        mean_ms, samples = prop.profile_end_samples()
        srt = sorted(float(x) for x in samples)
        if not srt:
            return mean_ms, 0, {}
        # the average over the samples with the top and bottom tenth set aside: a handful of stamped launches read
        # two to four times long (the stamp's own marker traffic; the un-stamped loop's wall clock per step, an upper
        # bound of the true average, sits below the plain mean whenever one of them is in the sample)
        cut = len(srt) // 10
        core = srt[cut:len(srt) - cut] if len(srt) >= 10 else srt
        stats = {"mean_all_us": mean_ms * 1e3, "median_us": srt[len(srt) // 2] * 1e3, "min_us": srt[0] * 1e3,
                 "max_us": srt[-1] * 1e3, "stamp_stride": stride, "averaging": "mean of the middle 80 % of the samples"}
        return sum(core) / len(core), len(srt), stats
    mean_ms, n = prop.profile_end()
    return mean_ms, n, {}


def cpu_baseline(cfg, n_rw, substeps, budget_s=12.0, n=8192, sh=None, single_s=0.0):
    """The CPU oracle (plain-C restatement, oracle/bsk_oracle.c) on the host cores of this box:
    OpenMP over spacecraft, bounded to ~budget_s.  A reported baseline, not the target.  ``sh``: harmonics degree
    (synthetic Kaula field) when the configuration's gravity model is GRAV_SH."""
    import numpy as np

    from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
    from oracle import oracle

    cores = oracle.set_threads(oracle.usable_cpus())
    st = sample_ic_batch(n, n_rw, seed=0)
    steps_c, ticks_c = np.zeros(n, np.int32), np.zeros(n, np.int32)
    act = np.zeros(n, np.int32)
    kw = {}
    if sh:
        from basilisk_env_amd.simulators.dynamics.gravity_sh import synthetic_sh_coefficients
        kw["cbar"], kw["sbar"] = synthetic_sh_coefficients(sh)
    if substeps == 1 and not sh:
        oracle.step(cfg, st, steps_c, ticks_c, act, substeps, omp=True, **kw)  # warm
    t0 = time.perf_counter()
    done_steps = 0
    while True:
        oracle.step(cfg, st, steps_c, ticks_c, act, substeps, omp=True, **kw)
        done_steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or done_steps >= 100000:
            break
    out = {"value": n * done_steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "sample": "%d envs x %d env-steps of %d RK4 sub-step(s), same physics/config, OpenMP over envs, %.1f s"
                     % (n, done_steps, substeps, el)}
    if single_s > 0:
        # ... and on ONE core ("on all host cores ... and single-threaded"): the same library, one thread
        n1 = max(1, n // max(cores, 1))
        st1 = sample_ic_batch(n1, n_rw, seed=0)
        s1, t1, a1 = np.zeros(n1, np.int32), np.zeros(n1, np.int32), np.zeros(n1, np.int32)
        oracle.set_threads(1)
        t0, k = time.perf_counter(), 0
        while True:
            oracle.step(cfg, st1, s1, t1, a1, substeps, omp=True, **kw)
            k += 1
            e1 = time.perf_counter() - t0
            if e1 > single_s or k >= 100000:
                break
        oracle.set_threads(cores)
        out["single_thread"] = {"value": n1 * k / e1, "unit": "env-steps/s", "cores": 1,
                                "sample": "%d envs x %d env-steps, one thread, %.1f s" % (n1, k, e1)}
    return out


KERNEL_SOURCES = ("basilisk_env_amd/csrc/bsk_kernels.hip", "basilisk_env_amd/csrc/bsk_device.hpp",
                  "basilisk_env_amd/csrc/bsk_launch.hpp", "basilisk_env_amd/csrc/bsk_probes.hpp", "basilisk_env_amd/csrc/dpp_nops.py")


def kernel_fingerprint():
    """sha256[:16] over the step kernel's sources: committed profile summaries carry the fingerprint of the tree they
    were measured on, so a bench line never silently mixes this tree's timings with another kernel's profile."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _latest_profile(name):
    d = os.path.join(ROOT, "profiles")
    best = None
    for sub in sorted(os.listdir(d)) if os.path.isdir(d) else []:
        f = os.path.join(d, sub, name)
        if os.path.exists(f):
            best = f
    return best


def rocprof_kernel(key):
    """Steady-state duration of the step kernel as rocprofv3 --kernel-trace saw it for the bench command of ``key``
    (tools/prof_summary.py -> profiles/*/kernel_trace.json).  -> dict or None."""
    best = _latest_profile("kernel_trace.json")
    if not best:
        return None
    try:
        d = json.load(open(best))
        rec = d.get("runs", {}).get(key)
        if not rec:
            return None
        rec = dict(rec)
        rec["source"] = os.path.relpath(best, ROOT)
        rec["fingerprint"] = d.get("fingerprint")
        return rec
    except Exception:
        return None


def settle_roofline(roof, key, stamped_us, wall_us, work, peak, fp=None):
    """Put every duration estimate into the roofline object and price it on the MORE CONSERVATIVE of the two kernel
    timings: the dispatch-stamped pass of this run and the committed rocprofv3 kernel trace of the same command
    (steady state; used only when it was taken on this very kernel source).  ``work`` = algorithmic bytes or flops per
    launch; ``peak`` in the roofline's unit (GB/s or TFLOP/s).  The wall time per launch of the un-stamped loop is
    reported beside them (an upper bound of the average duration of back-to-back launches)."""
    fp = fp or kernel_fingerprint()
    roof["kernel_us_stamped"] = stamped_us
    roof["wall_us_per_launch"] = wall_us
    rp = rocprof_kernel(key)
    # launches of one stream cannot overlap, so the wall time per launch of the un-stamped loop bounds the kernel's average duration
    # from above: a stamped pass that reads longer than that (the stamps' own weight on a microsecond-scale kernel: 6.5 - 7.1 us from
    # run to run against 6.35 wall) is cut down to it before the comparison with the profiler's trace
    used = min(stamped_us, wall_us) if (wall_us and wall_us > 0) else stamped_us
    roof["kernel_us_stamped_bounded"] = used
    if rp:
        fresh = rp.get("fingerprint") == fp
        if rp.get("trimmed_mean_us_median_of_boxes"):      # the command was traced on several boxes: their median
            roof["kernel_us_rocprof_boxes"] = [b["trimmed_mean_us"] for b in rp.get("boxes", [])]
            rp["trimmed_mean_us"] = rp["trimmed_mean_us_median_of_boxes"]
        roof["kernel_us_rocprof"] = rp.get("trimmed_mean_us")
        roof["kernel_us_rocprof_median"] = rp.get("median_us")
        roof["kernel_us_rocprof_source"] = "%s [%s]" % (rp["source"], rp.get("csv", key))
        roof["kernel_us_rocprof_fresh"] = bool(fresh)
        if fresh and rp.get("trimmed_mean_us"):
            used = max(used, float(rp["trimmed_mean_us"]))
        elif not fresh:
            roof["kernel_us_rocprof_note"] = "taken on other kernel sources (fingerprint %s, this tree %s): not used for frac" % (rp.get("fingerprint"), fp)
    else:
        roof["kernel_us_rocprof"] = None
    roof["kernel_us"] = used
    roof["kernel_us_rule"] = "max(min(stamped pass of this run, wall time per launch of the un-stamped loop), committed rocprofv3 steady-state trimmed mean of the same command; median over the boxes it was traced on)"
    unit = 1e3 if roof.get("unit") == "GB/s" else 1e6        # bytes/us -> GB/s ; flop/us -> TFLOP/s
    if work is not None and used > 0:
        roof["achieved"] = work / used / unit
        roof["frac"] = roof["achieved"] / peak
        roof["frac_stamped"] = work / stamped_us / unit / peak if stamped_us > 0 else None
        if roof.get("algorithmic_bytes_moved"):
            roof["frac_on_bytes_moved"] = roof["algorithmic_bytes_moved"] / used / unit / peak
    return roof


def pmc_traffic(n_envs, substeps):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC passes
    (tools/round.sh TAG counters -> profiles/*/summary_latest.json; separate FETCH_SIZE / WRITE_SIZE runs of this
    same command, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).  None when no matching profile."""
    if substeps != 1:
        return None, None
    best = _latest_profile("summary_latest.json")
    if not best:
        return None, None
    try:
        t = json.load(open(best)).get("traffic", {}).get(str(n_envs))
        return (t["traffic_bytes"], os.path.relpath(best, ROOT)) if t else (None, None)
    except Exception:
        return None, None


def isa_mix(key):
    """fp64 instructions the step kernel EXECUTES per RK4 sub-step and wave, by class, from the committed
    SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 counter passes (tools/round.sh TAG isa -> profiles/*/isa_mix.json).
    key: 'bare' | 'power' | 'full' | 'sh'.  -> (dict, source, fingerprint of the tree it was counted on)."""
    best = _latest_profile("isa_mix.json")
    if not best:
        return None, None, None
    try:
        d = json.load(open(best))
        m = d.get(key)
        return (m, os.path.relpath(best, ROOT), d.get("_meta", {}).get("fingerprint")) if m else (None, None, None)
    except Exception:
        return None, None, None


def fp64_roofline(key, rk4_steps_per_gpu, kernel_s, info):
    """Roofline object for the fp64-issue-bound regimes: executed flop = 64 lanes x (2 FMA + MUL + ADD + TRANS)
    per wave-instruction (counter passes), against the 78.6 TFLOP/s fp64 vector peak.  ``flop_per_launch`` is what
    settle_roofline() prices."""
    mix, src, mix_fp = isa_mix(key)
    out = {"bound": "fp64", "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": None, "kernel": info["name"],
           "kernel_us": kernel_s * 1e6, "vgprs": info["vgprs"], "block": info["block"], "grid": info["grid"]}
    if not mix or kernel_s <= 0:
        out.update({"achieved": None, "frac": None, "flop_per_launch": None, "note": "no isa_mix.json entry '%s' under profiles/" % key})
        return out
    # counts are per wave-instruction and RK4 step; `lanes_per_env` > 1 where several waves carry one spacecraft
    # (the two-wave harmonics form runs the cheap RK4 part redundantly in both waves)
    flop_per_lane_step = (2.0 * mix["fma"] + mix["mul"] + mix["add"] + mix.get("trans", 0.0)) * mix.get("lanes_per_env", 1.0)
    achieved = flop_per_lane_step * rk4_steps_per_gpu / kernel_s / 1e12
    # the same instructions as issue slots: one fp64 wave-instruction occupies its SIMD for 4 cycles
    out.update({"achieved": achieved, "frac": achieved / FP64_PEAK_TFLOPS, "flop_per_rk4_step_executed": flop_per_lane_step,
                "flop_per_launch": flop_per_lane_step * rk4_steps_per_gpu,
                "fp64_instr_per_rk4_step": mix["fma"] + mix["mul"] + mix["add"] + mix.get("trans", 0.0),
                "valu_instr_per_rk4_step": mix.get("valu"), "isa_mix_source": src,
                "isa_mix_fresh": mix_fp == kernel_fingerprint()})
    if not out["isa_mix_fresh"]:
        out["isa_mix_note"] = ("instruction mix counted on other kernel sources (fingerprint %s): executed-flop figures are "
                               "indicative until tools/round.sh TAG isa has been re-run on this tree" % mix_fp)
    return out


INFINITY_CACHE_BYTES = 256 << 20   # MI355X_MICROARCH.md: 256 MiB of Infinity Cache in front of HBM


def hbm_roofline(n, kernel_s, info, traffic_bytes, traffic_src, launches, static_o3=True, n_fields=47):
    """``static_o3``: the bare levels neither load the battery charge nor store obs[3] when no battery of the batch started empty
    (StepArgs::static_charge): 8 of the 340 algorithmic bytes do not move - ``algorithmic_bytes_moved`` / ``frac_on_bytes_moved``
    say so, ``frac`` stays priced on SURVEY.md section 8(d)'s 340.  ``working_set``: a batch whose slab and outputs fit the Infinity
    Cache is served from it between launches - its fraction is cache bandwidth priced on the HBM peak, not an HBM figure."""
    achieved = BYTES_PER_ENV_STEP * n / kernel_s / 1e9 if kernel_s > 0 else 0.0
    stride = (n + 255) // 256 * 256
    footprint = (n_fields + 6 + 1) * stride * 8 + stride * 2
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "algorithmic_bytes_moved": (BYTES_PER_ENV_STEP - (8.0 if static_o3 else 0.0)) * n,
            "working_set_bytes": footprint,
            "working_set": ("cache-resident working set (%.0f MB of state + outputs against the 256 MiB Infinity Cache): the fraction is cache bandwidth priced on the HBM peak" % (footprint / 1e6)
                            if footprint < INFINITY_CACHE_BYTES else "streams from HBM (%.0f MB per launch working set)" % (footprint / 1e6)),
            "frac_of_copy_ceiling": achieved / HBM_COPY_CEILING_GBS, "copy_ceiling": HBM_COPY_CEILING_GBS,
            "traffic": traffic_bytes, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
            "algorithmic_bytes": BYTES_PER_ENV_STEP * n, "kernel": info["name"], "kernel_us": kernel_s * 1e6,
            "launches_timed": launches, "stamping": "separate pass after the timed region: pairs stamped inside a back-to-back burst, second of each pair counted",
            "bytes_per_env_step": BYTES_PER_ENV_STEP, "vgprs": info["vgprs"], "lds_bytes": info["lds_bytes"],
            "block": info["block"], "grid": info["grid"]}


def _with_degree(cfg, degree):
    cfg.sh_degree = degree
    return cfg


def gather_legs(prop, dist, torch, world, n):
    """The one exchange step of the path, through torch.distributed three ways: RCCL gather of the
    observation shards to rank 0, RCCL all-gather, and every GPU copying its own shard to pinned host memory.  The shard
    sizes are exchanged once, outside the clocked region (ObsGatherer).  Max over ranks, ms."""
    from basilisk_env_amd.parallel import ObsGatherer, local_obs_tensor

    def clock(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64)
        dist.barrier()
        cpu = dist.get_backend() == "gloo"
        t = t if cpu else t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    g = ObsGatherer(prop, dist)
    local = local_obs_tensor(prop)
    host = torch.empty(tuple(local.shape), dtype=local.dtype, pin_memory=True)
    shard = int(local.numel() * 8)
    out = {"shard_bytes": shard, "total_bytes": shard * world,
           "gather_to_rank0_ms": clock(lambda: g.gather(0)),
           "all_gather_ms": clock(g.all_gather),
           "direct_d2h_per_gpu_ms": clock(lambda: host.copy_(local_obs_tensor(prop), non_blocking=True)),
           # bytes each leg moves per call: over the fabric into rank 0 / into every rank; over PCIe per GPU
           "bytes": {"gather_to_rank0": shard * (world - 1), "all_gather": shard * (world - 1) * world, "direct_d2h_per_gpu": shard}}
    full = g.all_gather()
    assert tuple(full.shape) == (world, 5, n)
    out["_clock"] = clock
    return out


def direct_rccl_leg(prop, dist, torch, world, clock):
    """The same gather through librccl directly (basilisk_env_amd/rccl.py): ONE group of ncclSend / ncclRecv from the library's
    buffers into rank 0's on the propagator handle's own stream - no torch tensor, no staging copy.  Two forms side by side:
    the five observation rows alone (what the torch legs move), and the seven rows a trainer on rank 0 needs (+ reward, done
    reason: SURVEY.md section 8(e)); then the one all-reduce of the path, {sum of rewards, number of done envs} as two doubles
    from device-side partials.  Checked against the torch leg / the host sums on rank 0."""
    from basilisk_env_amd.parallel import DirectRcclGather, ObsGatherer, concat_shards
    out = {"messages_on_root": {}}
    for rows, layout in ((5, "columns"), (7, "columns"), (7, "rank-major")):
        d = DirectRcclGather(prop, dist, root=0, rows=rows, layout=layout)
        try:
            ms = clock(d.enqueue)
            key = "direct_rccl_gather_to_rank0" if rows == 5 else ("direct_rccl_gather7_to_rank0" if layout == "columns" else "direct_rccl_gather7_rank_major")
            out[key + "_ms"] = ms
            out.setdefault("bytes", {})[key] = d.bytes_over_fabric
            out["messages_on_root"][key] = d.messages_on_root          # receives the root posts per gather
            out["nccl_comm_count"] = d.comm_count()                    # ranks RCCL itself says the direct communicator spans
            if layout == "rank-major":
                # same data as the column form, one block per rank: checked against this rank's own buffers on the root
                prop.sync()
                if dist.get_rank() == 0:
                    rb = d.result_blocks()
                    own = rb["blocks"][0]
                    got_o = torch.as_tensor(own["obs"], device="cuda")
                    got_r = torch.as_tensor(own["reward"], device="cuda")
                    v = prop.device_views()
                    out["direct_rccl_rank_major_matches_own_shard"] = bool(torch.equal(got_o, torch.as_tensor(v["obs"], device="cuda")) and
                                                                           torch.equal(got_r, torch.as_tensor(v["reward"], device="cuda")))
                continue
            if rows == 5:
                ref = ObsGatherer(prop, dist).gather(0)
                if dist.get_rank() == 0:
                    got = torch.as_tensor(d.result_view(), device="cuda")
                    out["direct_rccl_matches_torch_gather"] = bool(torch.equal(got, concat_shards(ref)))
            else:
                rsum, ndone = prop.batch_stats()
                p = d.all_reduce_stats()
                ms2 = clock(d.all_reduce_stats)
                prop.sync()
                tot = torch.zeros(2, dtype=torch.float64, device="cuda")
                import ctypes
                from basilisk_env_amd import _hip
                _hip.check(_hip.runtime().hipMemcpyAsync(ctypes.c_void_p(tot.data_ptr()), ctypes.c_void_p(p), 16, _hip.hipMemcpyDeviceToDevice, ctypes.c_void_p(0)), "hipMemcpyAsync")
                torch.cuda.synchronize()
                want = torch.tensor([rsum, float(ndone)], dtype=torch.float64, device="cuda")
                dist.all_reduce(want)
                out["all_reduce_stats_ms"] = ms2
                out["all_reduce_stats_matches_host_sums"] = bool(abs(float(tot[0]) - float(want[0])) < 1e-9 * max(1.0, abs(float(want[0]))) and float(tot[1]) == float(want[1]))
                if dist.get_rank() == 0:
                    v = d.result_views()
                    rew = torch.as_tensor(v["reward"], device="cuda")
                    why = torch.as_tensor(v["reason"], device="cuda")
                    out["direct_rccl_gather7_shapes"] = [list(rew.shape), list(why.shape)]
        finally:
            d.close()
    out["direct_rccl_form"] = ("one group of ncclSend/ncclRecv per gather, rows straight into rank 0's buffers, handle stream; 7 = obs(5) + reward + reason; "
                               "rank_major = rank 0's buffer holds one f64[6][n_r] block per rank: 2 messages per rank instead of 7 (rccl.py)")
    return out


def with_deadline(seconds, on_timeout, fn):
    """Run fn(); if it has not returned after ``seconds`` call on_timeout() from a watchdog thread (which ends the
    process): an auxiliary collective leg must never take the whole bench line with it."""
    import threading
    done = threading.Event()

    def watch():
        if not done.wait(seconds):
            on_timeout()

    t = threading.Thread(target=watch, daemon=True)
    t.start()
    try:
        return fn()
    finally:
        done.set()


def guarded_leg(out, rank, seconds, leg, exit_fn=os._exit):
    """Run the auxiliary collective ``leg()`` (-> dict merged into ``out['gather']``) under a watchdog: if it has not returned
    after ``seconds`` every rank gives up - rank 0 still prints the ONE JSON line - and the process ends with status 3, so that
    the launcher records a failure (never a re-exec: this process has initialised the GPU).  BENCH_FAULT_HANG_LEG=1 replaces the
    leg by one that never returns (rehearsal of exactly this path; tests/test_bench_host.py)."""
    def give_up():
        out.setdefault("gather", {})["direct_rccl"] = "timeout after %g s: leg abandoned" % seconds
        if rank == 0:
            emit(out)
        exit_fn(3)

    if os.environ.get("BENCH_FAULT_HANG_LEG") == "1":
        import threading
        leg = threading.Event().wait          # blocks for ever
    try:
        res = with_deadline(seconds, give_up, leg)
        out.setdefault("gather", {}).setdefault("bytes", {}).update(res.pop("bytes", {}))
        out["gather"].update(res)
    except Exception as e:
        out.setdefault("gather", {})["direct_rccl_error"] = repr(e)


def rank_identity(torch, dist, rank, local, world):
    """Which HIP device every rank bound and how many ranks its process group spans, gathered on rank 0 (and printed per rank
    on stderr): the first real multi-GPU record must show N ranks on N distinct devices."""
    p = torch.cuda.get_device_properties(local)
    me = {"rank": rank, "local_rank": local, "hip_device": int(torch.cuda.current_device()), "name": p.name,
          "pci_bus_id": getattr(p, "pci_bus_id", None), "uuid": str(getattr(p, "uuid", "")), "pid": os.getpid(),
          "process_group_size": int(dist.get_world_size()) if dist is not None else 1,
          "backend": dist.get_backend() if dist is not None else None}
    sys.stderr.write("[bench rank %d/%d] hip device %d (%s, pci bus %s), process group of %d over %s\n"
                     % (rank, world, me["hip_device"], me["name"], me["pci_bus_id"], me["process_group_size"], me["backend"]))
    sys.stderr.flush()
    if dist is None:
        return [me]
    box = [None] * world
    dist.all_gather_object(box, me)
    return box


def profile_key(a, sh):
    """Name of this configuration's run in profiles/*/kernel_trace.json (tools/round.sh produces them)."""
    if sh:
        return "sh70" if (a.substeps == 1 and a.envs == 65536) else None
    if a.features is not None or a.lds_scratch or a.fsw_timing != "reference":
        return None
    if a.envs == 65536:
        if a.scenario == "bare" and a.substeps == 1:
            return "65k_k1"
        return "%s_k%d" % (a.scenario, a.substeps)
    if a.envs == (1 << 22) and a.scenario == "bare" and a.substeps == 1:
        return "4m_k1"
    if a.envs == 131072 and a.scenario == "bare" and a.substeps == 1:
        return "131k_k1"
    return None


SH_FLOP_PER_RK4 = 4 * 2592 * 16 + 2 * 450   # config 5, algorithmic (SURVEY.md §8(d) / DESIGN.md §4)


def scenario_flags(scenario):
    from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY
    if scenario == "bare":
        return 0
    return FLAG_POWER | ((FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT) if scenario == "full" else 0)


def fp64_point(key, mix_key, prop, d_ptr, n, substeps, steps, warmup, stamped, barrier, sync, fp):
    """One fp64-issue-bound measurement point: wall-timed un-stamped launches, then the stamped pass; roofline priced on the
    conservative duration (settle_roofline)."""
    el = timed_run(prop, d_ptr, substeps, steps, warmup, barrier, sync)
    km, _, kst = kernel_time(prop, d_ptr, substeps, stamped)
    info = prop.kernel_info()
    roof = fp64_roofline(mix_key, float(n) * substeps, km * 1e-3, info)
    roof.update(kst)
    flop = roof.get("flop_per_launch")
    if mix_key == "sh":
        roof["executed_flop_per_launch"] = flop
        flop = float(n) * SH_FLOP_PER_RK4 * substeps
        roof["algorithmic_flop_per_env_step"] = SH_FLOP_PER_RK4 * substeps
        roof["flop_per_launch"] = flop
    settle_roofline(roof, key, km * 1e3, el / steps * 1e6, flop, FP64_PEAK_TFLOPS, fp)
    if mix_key == "sh" and roof.get("executed_flop_per_launch") and roof["kernel_us"] > 0:
        roof["executed_tflops"] = roof["executed_flop_per_launch"] / roof["kernel_us"] / 1e6
        roof["executed_frac"] = roof["executed_tflops"] / FP64_PEAK_TFLOPS
    return {"env_steps_per_s": n * steps / el, "rk4_substeps_per_s": n * steps * substeps / el, "ms_per_step": el / steps * 1e3,
            "kernel_ms": roof["kernel_us"] * 1e-3, "roofline": roof}


def batch_stats_point(torch, cfg, n, n_rw, local, sample_ic_batch, BatchedPropagator, steps):
    """What asking for the batch scalars after EVERY step costs (bsk_get_batch_stats_device: stats_kernel + stats_join_kernel
    enqueued behind the step kernel, no synchronisation): wall time per step of a back-to-back loop with and without the request,
    then the same with bsk_set_step_stats (the step launch forms the per-wave sums, a request is the join kernel alone).  The
    kernels' own durations are in the committed rocprofv3 traces (profiles/r05/kt_stats_*.csv)."""
    p = BatchedPropagator(cfg, n, device=local)
    p.reset(sample_ic_batch(n, n_rw, seed=5))
    act = torch.zeros(n, dtype=torch.int32, device="cuda")
    dp = act.data_ptr()

    def loop(with_stats):
        for _ in range(max(steps // 10, 5)):
            p.step_device(dp, 1)
            if with_stats:
                p.batch_stats_device()
        p.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            p.step_device(dp, 1)
            if with_stats:
                p.batch_stats_device()
        p.sync()
        return (time.perf_counter() - t0) / steps * 1e6

    base = min(loop(False) for _ in range(3))
    both = min(loop(True) for _ in range(3))
    rsum, ndone = p.batch_stats()
    # bsk_set_step_stats: the first level inside the step launch (what a consumer that asks after EVERY step switches on)
    p.set_step_stats(True)
    fused_alone = min(loop(False) for _ in range(3))
    fused = min(loop(True) for _ in range(3))
    rew = p.get_obs()[1]
    rsum2 = p.batch_stats()[0]
    ok = abs(rsum2 - float(rew.sum())) < 1e-9 * max(1.0, abs(float(rew.sum())))
    p.close()
    return {"envs": n, "step_us": base, "step_plus_stats_us": both, "added_us_per_step": both - base, "matches_host_sum": bool(ok),
            "stats_grid": max(1, min((((n + 63) // 64) + 3) // 4, 2048)),
            "in_launch_wave_sums": {"step_us": fused_alone, "step_plus_stats_us": fused, "added_us_per_step": fused - base}}


def rollout_point(torch, cfg, n, n_rw, local, sample_ic_batch, BatchedPropagator, T=541, reps=5, warm=1):
    """Open-loop rollouts (bsk_step_n): T env steps of ONE RK4 sub-step in one launch, state in registers across them - the
    reference's own mains step whole episodes under one action (envs/leoPowerAttitudeEnvironment.py:218-231).  A point of its own,
    never `value`: per env step the launch reads 4 B (0 with a constant action) and writes 49 B (five observations, reward, done
    reason); the state slab moves once per launch.  What bounds it is fp64 issue (one wave per SIMD at 65 536 spacecraft)."""
    p = BatchedPropagator(cfg, n, device=local)
    p.reset(sample_ic_batch(n, n_rw, seed=6))
    act = torch.zeros((T, n), dtype=torch.int32, device="cuda")
    ob = torch.empty((T, 5, n), dtype=torch.float64, device="cuda")
    rw = torch.empty((T, n), dtype=torch.float64, device="cuda")
    wy = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    out = {"envs": n, "steps_per_launch": T, "substeps": 1}
    for key, a in (("constant_action", None), ("device_actions", act.data_ptr())):
        for _ in range(warm):          # (a 0.9 ms launch needs ~25 ms of them before the clocks have settled)
            p.step_n(T, 1, a, 0, ob.data_ptr(), rw.data_ptr(), wy.data_ptr())
        p.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            p.step_n(T, 1, a, 0, ob.data_ptr(), rw.data_ptr(), wy.data_ptr())
        p.sync()
        dt = (time.perf_counter() - t0) / reps
        bytes_step = 49 + (4 if a else 0)
        out[key] = {"env_steps_per_s": n * T / dt, "us_per_env_step": dt / T * 1e6, "ms_per_launch": dt * 1e3,
                    "algorithmic_bytes_per_env_step": bytes_step + 2 * 128.0 / T,
                    "history_GBps": bytes_step * n * T / dt / 1e9}
    assert bool(torch.isfinite(ob).all()) and bool(torch.isfinite(rw).all())
    out["kernel"] = p.kernel_info()
    p.close()
    return out


def small_batch_crossover(small_batch, cpu_single):
    """The reference's own operating point is ONE spacecraft (simulators/leoPowerAttitudeSimulator.py:213): below which batch size is
    one host core (the oracle, single thread) the faster engine for a 180 s env step of the full scenario?  -> {"n": smallest measured
    batch the GPU wins at, "cpu_core_ms_per_env_step": ..., "gpu_ms_per_env_step": {N: ms}} or None.  (The GPU's time is flat in N
    down here - a launch of three waves per spacecraft costs what it costs - while the core's grows with N.)"""
    if not small_batch or not cpu_single or "error" in small_batch or not cpu_single.get("value"):
        return None
    cpu_ms = 1e3 / float(cpu_single["value"])                # one core: ms per env step of one spacecraft
    gpu = {int(k): float(v["ms_per_env_step"]) for k, v in small_batch.items() if k.isdigit()}
    wins = sorted(nn for nn, ms in gpu.items() if ms < nn * cpu_ms)
    return {"n": wins[0] if wins else None, "cpu_core_ms_per_env_step": cpu_ms, "gpu_ms_per_env_step": {str(k): gpu[k] for k in sorted(gpu)},
            "rule": "smallest measured N with GPU wall time of one env step (launch + kernel + sync) < N x one core's time per env step"}


def vecenv_episode_end(n, device_pool):
    """Host time of LeoPowerAttVecEnv.step_wait at the step where EVERY episode of the batch ends (max_length = 2, so every third
    step; with the reference's max_length = 540 and a common reset() it is every 541st) beside an ordinary step: the surface
    stable-baselines drops in on.  278 ms (device pool) / 425 ms (host resets) before round 5."""
    import numpy as np
    from basilisk_env_amd.envs import LeoPowerAttVecEnv
    kw = {"n_rw": 4, "step_duration": 0.1, "seed": 0, "device_reset_pool": device_pool, "device_sampler": bool(device_pool)}
    probe = LeoPowerAttVecEnv(64, **kw)
    cfg = probe.cfg.copy()
    probe.close()
    cfg.max_length = 2
    env = LeoPowerAttVecEnv(n, cfg=cfg, **kw)
    env.reset()
    acts = np.zeros(n, np.int64)
    ordinary, all_done = [], []
    for _ in range(12):
        env.step_async(acts)
        t0 = time.perf_counter()
        _, _, done, infos = env.step_wait()
        dt = (time.perf_counter() - t0) * 1e3
        (all_done if done.all() else ordinary).append(dt)
    one = infos[0] if all_done else None
    env.close()
    srt = lambda v: sorted(v)[len(v) // 2] if v else None     # noqa: E731
    return {"envs": n, "device_reset_pool": device_pool, "ordinary_step_wait_ms": srt(ordinary), "all_done_step_wait_ms": srt(all_done),
            "all_done_steps_seen": len(all_done), "terminal_info_keys": sorted(one.keys()) if one else None}


def rl_loop(torch, n, substeps, steps, warmup=10):
    """The whole on-device RL step as a training loop sees it: an on-GPU policy (linear -> argmax) reads the (N,5,1)
    observation tensor, the env steps on device int32 actions (LeoPowerAttVecEnv.step_tensors: the full reference
    scenario, device-side auto-reset), rewards accumulate on the device; nothing crosses PCIe and the host never waits
    inside the loop.  -> end-to-end env-steps/s beside the step kernel's own rate."""
    from basilisk_env_amd.envs import LeoPowerAttVecEnv
    # policy and env share ONE non-default torch stream (the legacy default stream synchronises with every other stream
    # of the process: measured 76 us per K = 1 step there against 41 us here)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        env = LeoPowerAttVecEnv(n, n_rw=4, step_duration=0.1 * substeps, seed=0, device_reset_pool=4096, device_sampler=True,
                                stream=side.cuda_stream)
        ob = env.reset_tensors()
        g = torch.Generator(device="cuda").manual_seed(0)
        w = torch.randn(5, 3, dtype=torch.float64, device="cuda", generator=g)

        def one(ob):
            act = (ob.reshape(n, 5) @ w).argmax(dim=1)          # int64, read in place by the step kernel; (N,5) is a view
            ob2, rew, done, info = env.step_tensors(act)        # no torch kernel in here; returns accumulate in the step kernel
            return ob2, info

        info = None
        for _ in range(warmup):
            ob, info = one(ob)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ob, info = one(ob)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ret = info["episode_return"]
        # the policy's two torch kernels alone, same stream, same shapes: what the loop costs without the env in it
        for _ in range(warmup):
            (ob.reshape(n, 5) @ w).argmax(dim=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            (ob.reshape(n, 5) @ w).argmax(dim=1)
        torch.cuda.synchronize()
        el_policy = time.perf_counter() - t0
    d_act = torch.zeros(n, dtype=torch.int32, device="cuda")
    km, _, _ = kernel_time(env.propagator, d_act.data_ptr(), substeps, 16 if substeps > 1 else 64)
    assert bool(torch.isfinite(ret).all())
    env.close()
    return {"env_steps_per_s": n * steps / el, "ms_per_step": el / steps * 1e3, "kernel_ms": km,
            "kernel_env_steps_per_s": n / (km * 1e-3), "loop_over_kernel_rate": (n * steps / el) / (n / (km * 1e-3)),
            "steps": steps, "torch_kernels_per_step": 2, "policy_only_ms_per_step": el_policy / steps * 1e3,
            "env_share_over_kernel": ((el - el_policy) / steps * 1e3) / km,
            "note": "loop time minus the policy's own two kernels, over the step kernel's time: what the env adds per step beyond its "
                    "kernel (1.0 = nothing); the loop captured in a HIP graph runs at the eager loop's rate (tools/attic/exp/rl_graph.py, "
                    "tests/test_gpu_device_surface.py): it is bound by the three dependent kernels, not by launches",
            "policy": "obs(N,5) @ W(5,3) -> argmax (int64, consumed in place), torch on the env's (non-default) stream; episode returns / lengths / done byte kept by the step kernel",
            "env": "LeoPowerAttVecEnv.step_tensors, full reference scenario, J2 + 4 wheels, device IC pool 4096 (Philox), device-side auto-reset"}


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    global FULL_LINE
    FULL_LINE = bool(a.full_line)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a.gpus, argv))

    import faulthandler

    import numpy as np
    import torch

    faulthandler.enable()     # a native fault leaves a Python traceback on stderr instead of a bare signal
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d (run `python bench.py --gpus N` as typed, or "
                         "torch.distributed.run --nproc-per-node N)" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: no HIP device visible (there is no CPU path to measure)")
    # BENCH_REHEARSAL=1: exercise the N > 1 control flow on a box with a single GPU (every rank on
    # device 0, gloo instead of RCCL).  Never set by the driver; numbers from it mean nothing.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    # under a launcher (WORLD_SIZE set) the process group is initialised even for a single rank, so that the
    # collective legs below run on the real RCCL backend wherever the bench is launched the multi-GPU way
    if world > 1 or "WORLD_SIZE" in os.environ:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    from basilisk_env_amd._lib import FLAG_LDS_SCRATCH, GRAV_PM_J2, GRAV_SH
    from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
    from basilisk_env_amd.simulators.dynamics.gravity_sh import synthetic_sh_coefficients
    from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch

    n_rw = 4
    sh = a.gravity == "sh"
    fp = kernel_fingerprint()
    cfg = default_config(n_rw=n_rw, gravity_model=GRAV_SH if sh else GRAV_PM_J2)
    cfg.flags |= scenario_flags(a.scenario)
    if a.features is not None:
        from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY
        names = {"power": FLAG_POWER, "sun": FLAG_SUN_THIRD_BODY, "drag": FLAG_DRAG, "desat": FLAG_DESAT}
        cfg.flags &= ~(FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT)
        for f in filter(None, a.features.split(",")):
            cfg.flags |= names[f]
    if a.lds_scratch:
        cfg.flags |= FLAG_LDS_SCRATCH
    if a.fsw_timing == "same-tick":
        cfg.fsw_lag = cfg.nav_lag = 0
    n = a.envs
    ic = sample_ic_batch(n, n_rw, seed=rank)       # rank r owns env indices [r*n, (r+1)*n)
    prop = BatchedPropagator(cfg if not sh else _with_degree(cfg, 70), n, device=local)
    if sh:
        cfg = prop.cfg
        prop.set_gravity_sh(70, *synthetic_sh_coefficients(70))
    prop.reset(ic)
    # The wave-level reductions of the path inside the timed launch: every step launch forms the done
    # ballot AND the per-wave reward sums (bsk_set_step_stats; measured at <= 0.06 us on the K = 1 launch); the join of the wave
    # sums into the two batch scalars is a request of its own (bsk_get_batch_stats_device) - `value_with_join` times it after
    # every step.
    prop.set_step_stats(True)
    d_act = torch.zeros(n, dtype=torch.int32, device="cuda")  # action 0 = nadir pointing (reward mode)
    torch.cuda.synchronize()
    sync = torch.cuda.synchronize

    el_local = timed_run(prop, d_act.data_ptr(), a.substeps, a.steps, a.warmup, barrier, sync)
    el = max_over_ranks(el_local)
    el_join = max_over_ranks(timed_run(prop, d_act.data_ptr(), a.substeps, 1 if a.no_join else a.steps, 0 if a.no_join else a.warmup, barrier, sync, join=True))
    rsum_dev, ndone_dev = prop.batch_stats()
    _, rew_j, _, why_j = prop.get_obs()
    join_ok = bool(abs(rsum_dev - float(rew_j.sum())) <= 1e-9 * max(1.0, abs(float(rew_j.sum()))) and ndone_dev == int((why_j != 0).sum()))
    assert join_ok or os.environ.get("BENCH_ALLOW_NONFINITE") == "1", "device-side batch scalars differ from the host sums of the same step"
    kernel_ms, n_launch, kstats = kernel_time(prop, d_act.data_ptr(), a.substeps, min(STAMPED_LAUNCHES, max(a.steps, 4)))
    obs, rew, done, why = prop.get_obs()
    # (BENCH_ALLOW_NONFINITE=1: timing-only ablation builds whose results are deliberately wrong)
    assert os.environ.get("BENCH_ALLOW_NONFINITE") == "1" or (np.isfinite(obs).all() and np.isfinite(rew).all())
    info = prop.kernel_info()
    key = profile_key(a, sh)
    wall_us = el_local / a.steps * 1e6

    value = n * world * a.steps / el
    kernel_s = kernel_ms * 1e-3
    traffic_bytes, traffic_src = pmc_traffic(n, a.substeps) if (not sh and a.scenario == "bare") else (None, None)
    hbm = hbm_roofline(n, kernel_s, info, traffic_bytes, traffic_src, n_launch, static_o3=a.scenario == "bare" and not sh)
    hbm.update(kstats)
    settle_roofline(hbm, key, kernel_ms * 1e3, wall_us, BYTES_PER_ENV_STEP * n, HBM_PEAK_GBS, fp)
    hbm["frac_of_copy_ceiling"] = hbm["achieved"] / HBM_COPY_CEILING_GBS
    # which roofline bounds this configuration: K = 1 of the bare / power / full propagator streams its state once
    # per launch (HBM); many sub-steps per launch and the harmonics are fp64-issue bound
    mix_key = "sh" if sh else a.scenario
    fp64_bound = sh or a.substeps >= 10
    fp64 = fp64_roofline(mix_key, float(n) * a.substeps, kernel_s, info)
    fp64.update(kstats)
    flop = fp64.get("flop_per_launch")
    if sh:
        # config 5, algorithmic flops as SURVEY.md §8(d) / DESIGN.md §4 count them: 9 fp64 instructions (7 FMA + 2 MUL
        # = 16 flop) per (l, m) entry of the padded Pines stream (2 592 entries at degree 70), four field evaluations
        # per RK4 step, + ~450 fp64 instructions for the rest of the step.  `roofline.achieved` is this figure; the
        # counter-derived one (every executed fp64 instruction, column ends and the redundant RK4 of the second wave
        # included) is kept beside it.
        fp64["executed_flop_per_launch"] = flop
        flop = float(n) * SH_FLOP_PER_RK4 * a.substeps
        fp64["algorithmic_flop_per_env_step"] = SH_FLOP_PER_RK4 * a.substeps
    settle_roofline(fp64, key, kernel_ms * 1e3, wall_us, flop, FP64_PEAK_TFLOPS, fp)
    if sh and fp64.get("executed_flop_per_launch"):
        fp64["executed_tflops"] = fp64["executed_flop_per_launch"] / fp64["kernel_us"] / 1e6
        fp64["executed_frac"] = fp64["executed_tflops"] / FP64_PEAK_TFLOPS
    out = {
        "metric": "env steps/sec at 65k parallel spacecraft, 1/2/4/8 MI355X; HBM GB/s vs roofline",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%s]: %d envs/GPU, %s + 4 reaction wheels (pyramid) + nadir reward, f64, dt 0.1 s, "
                               "K=%d RK4 sub-step(s)/env step, FSW every 10 (fsw_lag=nav_lag=1), synthetic PCG64(rank)"
                               % ("4" if sh else "2", n, "degree-70 harmonics (synthetic Kaula)" if sh else "J2", a.substeps),
                   "envs_per_gpu": n, "substeps": a.substeps, "scenario": a.scenario, "lds_scratch": bool(a.lds_scratch), "fsw_timing": a.fsw_timing, "features": a.features,
                   "sharding": "env ranges, no step-path collective", "kernel_fingerprint": fp,
                   "batch_stats": "per-wave sums in the step launch; join on demand"},
        "value_with_join": None if a.no_join else n * world * a.steps / el_join,
        "ms_per_step_with_join": None if a.no_join else el_join / a.steps * 1e3, "join_matches_host_sum": join_ok,
        "roofline": fp64 if fp64_bound else hbm,
        "rk4_substeps_per_s": value * a.substeps,
    }
    if fp64_bound:
        out["roofline_hbm"] = hbm
    if sh:
        out["sh"] = {"degree": 70, "field_evals_per_s": n * world * a.steps * a.substeps * 4 / el}

    ranks = rank_identity(torch, dist, rank, local, world)
    if rank == 0:
        out["ranks"] = ranks
        out["distinct_devices"] = len({(r["pci_bus_id"], r["uuid"], r["hip_device"]) for r in ranks})
    clock = None
    if dist is not None:
        out["gather"] = gather_legs(prop, dist, torch, world, n)
        clock = out["gather"].pop("_clock")
        out["gather_ms"] = out["gather"]["all_gather_ms"]

    extra = {}
    default_line = not a.no_extra and not sh and a.scenario == "bare" and not a.lds_scratch and a.features is None
    if world == 1 and default_line:
        dp = d_act.data_ptr()
        # reference-faithful env step: 180 s of sim time = 1 800 RK4 sub-steps, 180 FSW updates
        # (the first few 2 ms launches after the K = 1 burst run while the clocks settle: five warm-up steps)
        extra["k1800"] = fp64_point("bare_k1800", "bare", prop, dp, n, 1800, 10, 12, 5, barrier, sync, fp)
        # the kernels behind the drop-in env (leoPowerAttEnv / LeoPowerAttVecEnv): config-3 physics + the reference
        # scenario of leoPowerAttitudeSimulator.py:213-366 (power system; + Sun third body, drag, desaturation),
        # replacing the ExecuteSimulation call at :594-595, at the reference's 1 800 sub-steps per env step
        for sc in ("power", "full"):
            c2 = cfg.copy()
            c2.flags |= scenario_flags(sc)
            p2 = BatchedPropagator(c2, n, device=local)
            p2.reset(ic)
            extra["%s_k1800" % sc] = fp64_point("%s_k1800" % sc, sc, p2, dp, n, 1800, 10, 10, 5, barrier, sync, fp)
            extra["%s_k1800" % sc]["scenario"] = sc
            p2.close()
        # BASELINE configs[4]: degree-70 harmonics, K = 1 (a 250 us kernel needs ~300 launches before the clocks settle)
        c5 = default_config(n_rw=n_rw, gravity_model=GRAV_SH)
        c5.sh_degree = 70
        p5 = BatchedPropagator(c5, n, device=local)
        p5.set_gravity_sh(70, *synthetic_sh_coefficients(70))
        p5.reset(ic)
        extra["sh70"] = fp64_point("sh70", "sh", p5, dp, n, 1, 1000, 300, 32, barrier, sync, fp)
        extra["sh70"]["workload"] = "BASELINE configs[4]: 65 536 envs, degree-70 harmonics (synthetic Kaula field), 4 wheels, K = 1"
        p5.close()
        # large-batch point where the HBM roofline is the binding limit (4 Mi envs = 1.4 GB/launch)
        nl = 1 << 22
        big = BatchedPropagator(cfg, nl, device=local)
        big.reset(sample_ic_batch(nl, n_rw, seed=1))
        d_act_big = torch.zeros(nl, dtype=torch.int32, device="cuda")
        el3 = timed_run(big, d_act_big.data_ptr(), 1, 50, 5, barrier, sync)
        km3, nl3, kst3 = kernel_time(big, d_act_big.data_ptr(), 1, 20)
        tb, ts = pmc_traffic(nl, 1)
        roof3 = hbm_roofline(nl, km3 * 1e-3, big.kernel_info(), tb, ts, nl3)
        roof3.update(kst3)
        settle_roofline(roof3, "4m_k1", km3 * 1e3, el3 / 50 * 1e6, BYTES_PER_ENV_STEP * nl, HBM_PEAK_GBS, fp)
        roof3["frac_of_copy_ceiling"] = roof3["achieved"] / HBM_COPY_CEILING_GBS
        extra["large_n"] = {"envs": nl, "env_steps_per_s": nl * 50 / el3, "roofline": roof3}
        big.close()
        del d_act_big
        # BASELINE configs[3], the per-GPU half: 131 072 envs per GPU (1 048 576 on 8), config-3 physics, K = 1 - driver-timed on
        # one card while no 8-GPU node is at hand (the exchange legs need the launcher: `gather`)
        n3 = 131072
        p3 = BatchedPropagator(cfg, n3, device=local)
        p3.reset(sample_ic_batch(n3, n_rw, seed=1000))
        d_act3 = torch.zeros(n3, dtype=torch.int32, device="cuda")
        el4 = timed_run(p3, d_act3.data_ptr(), 1, 2000, 100, barrier, sync)
        km4, nl4, kst4 = kernel_time(p3, d_act3.data_ptr(), 1, 48)
        tb4, ts4 = pmc_traffic(n3, 1)
        roof4 = hbm_roofline(n3, km4 * 1e-3, p3.kernel_info(), tb4, ts4, nl4)
        roof4.update(kst4)
        settle_roofline(roof4, "131k_k1", km4 * 1e3, el4 / 2000 * 1e6, BYTES_PER_ENV_STEP * n3, HBM_PEAK_GBS, fp)
        roof4["frac_of_copy_ceiling"] = roof4["achieved"] / HBM_COPY_CEILING_GBS
        extra["config3_per_gpu"] = {"workload": "BASELINE configs[3], one GPU's share: %d envs, J2 + 4 wheels, K = 1" % n3, "envs": n3,
                                    "env_steps_per_s": n3 * 2000 / el4, "ms_per_step": el4 / 2000 * 1e3, "roofline": roof4}
        p3.close()
        del d_act3
        # what this device sustains on a pure fp64 FMA stream today (its clocks under a dense fp64 load): the rooflines
        # above stay priced on the nominal 78.6 TFLOP/s; this is printed beside them
        try:
            from basilisk_env_amd._lib import calibrate_fp64
            tf1, ns1 = calibrate_fp64(local, 1, 5)
            tf2, ns2 = calibrate_fp64(local, 2, 5)
            extra["fp64_ceiling"] = {"nominal_tflops": FP64_PEAK_TFLOPS, "measured_tflops_1_wave_per_simd": tf1, "measured_tflops_2_waves_per_simd": tf2,
                                     "ns_per_fma_per_simd_1_wave": ns1, "ns_per_fma_per_simd_2_waves": ns2,
                                     "method": "bsk_calibrate_fp64: 16 independent v_fma_f64 chains per lane, 1024 x waves workgroups of 64, median of 5 launches of ~2-4 ms"}
            for k, ceil in (("k1800", tf1), ("power_k1800", tf1), ("full_k1800", tf1), ("sh70", tf2)):
                r = extra.get(k, {}).get("roofline")
                if r and r.get("achieved") and ceil > 0:
                    r["frac_of_measured_fma_ceiling"] = r["achieved"] / ceil
        except Exception as e:
            extra["fp64_ceiling"] = {"error": repr(e)}
        # small batches (the reference itself runs ONE environment): wall time of one 180 s env step of the full reference
        # scenario - launch + kernel + stream synchronisation - for 1 / 64 / 8 192 spacecraft (pair form of the kernel)
        try:
            sb = {}
            for ns in (1, 2, 4, 64, 8192):
                c6 = cfg.copy()
                c6.flags |= scenario_flags("full")
                p6 = BatchedPropagator(c6, ns, device=local)
                p6.reset(sample_ic_batch(ns, n_rw, seed=7))
                a6 = torch.zeros(ns, dtype=torch.int32, device="cuda")
                p6.step_device(a6.data_ptr(), 1800)
                p6.sync()
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    p6.step_device(a6.data_ptr(), 1800)
                    p6.sync()
                    ts.append(time.perf_counter() - t0)
                sb[str(ns)] = {"ms_per_env_step": min(ts) * 1e3, "kernel": p6.kernel_info()["name"]}
                p6.close()
            extra["small_batch"] = sb
        except Exception as e:
            extra["small_batch"] = {"error": repr(e)}
        # the boundary with HOST buffers on both sides (bsk_step with host actions + bsk_get_obs into page-locked host arrays,
        # one synchronisation per step): the PCIe-inclusive rate of DESIGN.md section 7 - never `value`
        try:
            acts_h = np.zeros(n, np.int32)
            for _ in range(20):
                prop.step(acts_h, 1)
                prop.get_obs(copy=False)
            t0 = time.perf_counter()
            for _ in range(300):
                prop.step(acts_h, 1)
                prop.get_obs(copy=False)
            dt = (time.perf_counter() - t0) / 300
            extra["host_buffers_k1"] = {"env_steps_per_s": n / dt, "ms_per_step": dt * 1e3,
                                        "bytes_per_step": {"h2d": 4 * n, "d2h": 49 * n},
                                        "what": "bsk_step (host int32 actions) + bsk_get_obs (obs, reward, reason -> pinned host arrays), synchronised every step"}
        except Exception as e:
            extra["host_buffers_k1"] = {"error": repr(e)}
        # the batch scalars on demand (row a7): what bsk_get_batch_stats_device after every step adds, at four batch sizes
        try:
            extra["batch_stats_us"] = {str(nn): batch_stats_point(torch, cfg, nn, n_rw, local, sample_ic_batch, BatchedPropagator, st_)
                                       for nn, st_ in ((65536, 3000), (131072, 2000), (1 << 20, 400), (1 << 22, 120))}
        except Exception as e:
            extra["batch_stats_us"] = {"error": repr(e)}
        # open-loop rollouts: a whole 541-step episode per launch (bsk_step_n), 65 536 and 4 Mi spacecraft
        try:
            extra["rollout"] = {str(nn): rollout_point(torch, cfg, nn, n_rw, local, sample_ic_batch, BatchedPropagator, T=tt, reps=rr, warm=ww)
                                for nn, tt, rr, ww in ((65536, 541, 30, 30), (1 << 22, 100, 3, 2))}
            r1 = extra["rollout"]["65536"]["constant_action"]
            r1["over_one_launch_per_step"] = r1["env_steps_per_s"] / value
        except Exception as e:
            extra["rollout"] = {"error": repr(e)}
        # the VecEnv's host path at a synchronized episode end (every env of the batch finishes on the same step)
        try:
            extra["vecenv_episode_end_ms"] = {"device_pool": vecenv_episode_end(n, 4096), "host_resets": vecenv_episode_end(n, 0)}
        except Exception as e:
            extra["vecenv_episode_end_ms"] = {"error": repr(e)}
        # the device-resident RL loop (row f4): on-GPU policy -> step_tensors, at K = 1 and at the reference's K = 1 800
        try:
            extra["rl_loop"] = {"k1": rl_loop(torch, n, 1, 200), "k1800": rl_loop(torch, n, 1800, 200 if a.steps >= 1000 else 40)}
        except Exception as e:   # never lose the line to an auxiliary leg
            extra["rl_loop"] = {"error": repr(e)}
    if dist is not None and default_line:
        # BASELINE configs[3]: 131 072 envs per GPU (1 048 576 on 8), config-3 physics, + the observation exchange
        n3 = 131072
        p3 = BatchedPropagator(cfg, n3, device=local)
        p3.reset(sample_ic_batch(n3, n_rw, seed=1000 + rank))
        d_act3 = torch.zeros(n3, dtype=torch.int32, device="cuda")
        el4 = max_over_ranks(timed_run(p3, d_act3.data_ptr(), 1, 500, 20, barrier, sync))
        km4, _, _ = kernel_time(p3, d_act3.data_ptr(), 1, 32)
        extra["config3"] = {"workload": "BASELINE configs[3]: %d envs sharded over %d GPUs (%d per GPU), K = 1" % (n3 * world, world, n3),
                            "env_steps_per_s": n3 * world * 500 / el4, "ms_per_step": el4 / 500 * 1e3,
                            "roofline": hbm_roofline(n3, km4 * 1e-3, p3.kernel_info(), None, None, 32),
                            "gather": gather_legs(p3, dist, torch, world, n3)}
        extra["config3"]["gather"].pop("_clock", None)
        p3.close()
        # strong-scaling points of the literal target (65 536 spacecraft in total), K = 1 and the reference's K = 1 800
        ns = max(1, 65536 // world)
        ps = BatchedPropagator(cfg, ns, device=local)
        ps.reset(sample_ic_batch(ns, n_rw, seed=2000 + rank))
        d_acts = torch.zeros(ns, dtype=torch.int32, device="cuda")
        el5 = max_over_ranks(timed_run(ps, d_acts.data_ptr(), 1, 500, 20, barrier, sync))
        el6 = max_over_ranks(timed_run(ps, d_acts.data_ptr(), 1800, 3, 1, barrier, sync))
        extra["strong_65536_total"] = {"envs_per_gpu": ns, "k1_env_steps_per_s": ns * world * 500 / el5,
                                       "k1800_env_steps_per_s": ns * world * 3 / el6, "k1800_ms_per_step": el6 / 3 * 1e3}
        ps.close()
        # ... and with the scenario the drop-in env runs (small shards run the three-wave form of the kernel)
        cf = cfg.copy()
        cf.flags |= scenario_flags("full")
        pf = BatchedPropagator(cf, ns, device=local)
        pf.reset(sample_ic_batch(ns, n_rw, seed=2000 + rank))
        timed_run(pf, d_acts.data_ptr(), 1800, 2, 1, barrier, sync)
        el7 = max_over_ranks(timed_run(pf, d_acts.data_ptr(), 1800, 5, 1, barrier, sync))
        extra["strong_65536_total"].update({"full_k1800_env_steps_per_s": ns * world * 5 / el7, "full_k1800_ms_per_step": el7 / 5 * 1e3,
                                            "full_k1800_kernel": pf.kernel_info()["name"]})
        pf.close()
    if extra and rank == 0:
        out["extra"] = extra
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # the CPU oracle on this box's host cores, like for like beside every GPU point of the line (bounded samples)
        if sh:
            out["cpu_baseline"] = cpu_baseline(cfg, n_rw, a.substeps, budget_s=10.0, n=256, sh=70)
        else:
            out["cpu_baseline"] = cpu_baseline(cfg, n_rw, a.substeps, budget_s=8.0 if default_line else 12.0, single_s=3.0)
        if "k1800" in extra:
            extra["k1800"]["cpu_baseline"] = cpu_baseline(cfg, n_rw, 1800, budget_s=6.0, n=512)
        for sc in ("power", "full"):
            if "%s_k1800" % sc in extra:
                c2 = cfg.copy()
                c2.flags |= scenario_flags(sc)
                extra["%s_k1800" % sc]["cpu_baseline"] = cpu_baseline(c2, n_rw, 1800, budget_s=4.0, n=256, single_s=2.0 if sc == "full" else 0.0)
        if "sh70" in extra:
            c5 = default_config(n_rw=n_rw, gravity_model=GRAV_SH)
            c5.sh_degree = 70
            extra["sh70"]["cpu_baseline"] = cpu_baseline(c5, n_rw, 1, budget_s=5.0, n=256, sh=70)
        cross = small_batch_crossover(extra.get("small_batch"), extra.get("full_k1800", {}).get("cpu_baseline", {}).get("single_thread"))
        if cross:
            extra["small_batch"]["crossover"] = cross
            out["small_batch_crossover_n"] = cross["n"]
    hang = os.environ.get("BENCH_FAULT_HANG_LEG") == "1"
    if dist is not None and (not rehearsal or hang) and os.environ.get("BSKGPU_DIRECT_RCCL", "1") != "0":
        # last, and under a watchdog: if the direct-RCCL leg (a second communicator, never run on more than one rank
        # before the driver's 8-GPU node) hangs, every rank gives up after 90 s and rank 0 still prints the line
        guarded_leg(out, rank, float(os.environ.get("BENCH_LEG_DEADLINE_S", "90")), lambda: direct_rccl_leg(prop, dist, torch, world, clock))
    prop.close()
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
