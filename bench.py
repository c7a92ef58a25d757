#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched HIP propagator on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the one the metric is quoted on): 65 536 spacecraft PER GPU,
J2 gravity + 4 reaction wheels (pyramid) + nadir-pointing reward (action 0), fp64, synthetic
random-orbit batch, one RK4 sub-step of 0.1 s per env step (the HBM framing
of the metric).  Envs are independent: ranks shard them with no collective on the step path
("scaling": "weak"); one RCCL all-gather of the observation shards is exercised and timed
after the timed region (``gather_ms``).

A "step" = one pass of the hot path over the whole batch (one kernel launch per GPU): mode
switch, FSW chain when due, RK4, observation, reward, done mask, wave reductions.  Actions and
state are resident in HBM when the timed region starts.

Besides the contract keys the JSON line carries ``roofline`` (dominant kernel, algorithmic
bytes = 340 B/env-step), ``cpu_baseline`` (the CPU oracle on this box's host
cores, rank 0 at N=1 only) and ``extra`` (the reference-faithful 1 800 sub-steps/env-step rate
and a large-batch roofline point).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_ENV_STEP = 340.0   # config 3/4 algorithmic bytes per env-step
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FLOP_PER_RK4 = 4 * 330 + 110  # SURVEY.md §8(d) algorithmic flops per RK4 sub-step, config 3


def parse():
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
    p.add_argument("--stamp-every-launch", action="store_true",
                   help="dispatch-timestamp every launch of the timed region (lower throughput, every kernel isolated; "
                        "used for the rocprofv3 kernel-trace passes so that both report the same thing)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra", action="store_true")
    return p.parse_args()


def timed_run(prop, d_act_ptr, substeps, steps, warmup, barrier, torch, stamp_all=False):
    for _ in range(warmup):
        prop.step_device(d_act_ptr, substeps)
    prop.sync()
    # dispatch timestamps on a sample of the timed launches: stamping costs ~5 us of launch throughput
    # per stamped launch, so a pair is stamped every `stride` launches and its second launch counted (every
    # launch for short runs); ~128 samples per run, never more often than every 16th launch; see
    # bsk_profile_set_stride
    stride = max(16, steps // 128) if (steps >= 64 and not stamp_all) else 1
    prop.profile_begin(steps // stride + 2, stride=stride)   # capacity = launches that will be stamped and counted
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        prop.step_device(d_act_ptr, substeps)
    torch.cuda.synchronize()
    t1 = time.perf_counter()   # this rank's K steps are done; the MAX over ranks is taken by the caller
    barrier()
    kernel_ms, n_launch = prop.profile_end()
    return t1 - t0, kernel_ms, n_launch


def cpu_baseline(cfg, n_rw, substeps):
    """The CPU oracle (plain-C restatement, oracle/bsk_oracle.c) on the host cores of this box:
    OpenMP over spacecraft, bounded to ~10-20 s.  A reported baseline, not the target."""
    import numpy as np

    from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
    from oracle import oracle

    cores = oracle.set_threads(oracle.usable_cpus())
    n = 8192
    st = sample_ic_batch(n, n_rw, seed=0)
    steps_c, ticks_c = np.zeros(n, np.int32), np.zeros(n, np.int32)
    act = np.zeros(n, np.int32)
    oracle.step(cfg, st, steps_c, ticks_c, act, substeps, omp=True)  # warm
    t0 = time.perf_counter()
    done_steps = 0
    while True:
        oracle.step(cfg, st, steps_c, ticks_c, act, substeps, omp=True)
        done_steps += 1
        el = time.perf_counter() - t0
        if el > 12.0 or done_steps >= 100000:
            break
    return {"value": n * done_steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d env-steps of %d RK4 sub-step(s), same physics/config, OpenMP over envs, %.1f s"
                      % (n, done_steps, substeps, el)}


def pmc_traffic(n_envs, substeps):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC passes
    (tools/prof.sh -> profiles/*/summary_latest.json; separate FETCH_SIZE / WRITE_SIZE runs of this
    same command, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).  None when no matching profile."""
    if substeps != 1:
        return None, None
    best = None
    for d in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        f = os.path.join(ROOT, "profiles", d, "summary_latest.json")
        if os.path.exists(f):
            best = f
    if not best:
        return None, None
    try:
        t = json.load(open(best)).get("traffic", {}).get(str(n_envs))
        return (t["traffic_bytes"], os.path.relpath(best, ROOT)) if t else (None, None)
    except Exception:
        return None, None


def _with_degree(cfg, degree):
    cfg.sh_degree = degree
    return cfg


def main():
    a = parse()
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: no HIP device visible (there is no CPU path to measure)")
    # BENCH_REHEARSAL=1: exercise the N > 1 control flow on a box with a single GPU (every rank on
    # device 0, gloo instead of RCCL).  Never set by the driver; numbers from it mean nothing.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    def barrier():
        if dist is not None:
            dist.barrier()

    from basilisk_env_amd._lib import GRAV_PM_J2, GRAV_SH
    from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
    from basilisk_env_amd.simulators.dynamics.gravity_sh import synthetic_sh_coefficients
    from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch

    n_rw = 4
    sh = a.gravity == "sh"
    cfg = default_config(n_rw=n_rw, gravity_model=GRAV_SH if sh else GRAV_PM_J2)
    if a.scenario != "bare":
        from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY
        cfg.flags |= FLAG_POWER | ((FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT) if a.scenario == "full" else 0)
    n = a.envs
    ic = sample_ic_batch(n, n_rw, seed=rank)       # rank r owns env indices [r*n, (r+1)*n)
    prop = BatchedPropagator(cfg if not sh else _with_degree(cfg, 70), n, device=local)
    if sh:
        cfg = prop.cfg
        prop.set_gravity_sh(70, *synthetic_sh_coefficients(70))
    prop.reset(ic)
    d_act = torch.zeros(n, dtype=torch.int32, device="cuda")  # action 0 = nadir pointing (reward mode)
    torch.cuda.synchronize()

    el, kernel_ms, n_launch = timed_run(prop, d_act.data_ptr(), a.substeps, a.steps, a.warmup, barrier, torch,
                                        stamp_all=a.stamp_every_launch)
    el_t = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    if dist is not None:
        dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
    el = float(el_t.item())
    obs, rew, done, why = prop.get_obs()
    assert np.isfinite(obs).all() and np.isfinite(rew).all()
    info = prop.kernel_info()

    # the one exchange step of the path: all-gather of the observation shards over RCCL/xGMI
    gather_ms = None
    if dist is not None:
        from basilisk_env_amd.parallel import gather_observations
        gather_observations(prop, dist)  # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        full = gather_observations(prop, dist)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t0) * 1e3
        assert full.shape == (world, 5, n)

    value = n * world * a.steps / el
    traffic_bytes, traffic_src = pmc_traffic(n, a.substeps) if not sh else (None, None)
    kernel_s = kernel_ms * 1e-3
    achieved = BYTES_PER_ENV_STEP * n / kernel_s / 1e9 if kernel_s > 0 else 0.0
    out = {
        "metric": "env steps/sec at 65k parallel spacecraft, 1/2/4/8 MI355X; HBM GB/s vs roofline",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%s]: %d envs/GPU, %s gravity + 4 reaction wheels (pyramid) + "
                               "nadir-pointing reward, fp64, dt 0.1 s, %d RK4 sub-step(s) per env step, fsw every 10 "
                               "sub-steps, synthetic random-orbit batch PCG64(rank)"
                               % ("4" if sh else "2", n, "degree-70 spherical-harmonic (synthetic Kaula field)" if sh else "J2",
                                  a.substeps),
                   "envs_per_gpu": n, "substeps": a.substeps, "scenario": a.scenario,
                   "sharding": "env ranges, no step-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_bytes, "traffic_unit": "bytes/launch",
                     "traffic_source": traffic_src, "algorithmic_bytes": BYTES_PER_ENV_STEP * n,
                     "kernel": info["name"], "kernel_us": kernel_ms * 1e3, "launches_timed": n_launch,
                     "stamping": "every launch" if (a.stamp_every_launch or a.steps < 64) else "pairs every %d launches, second counted" % max(16, a.steps // 128),
                     "bytes_per_env_step": BYTES_PER_ENV_STEP, "vgprs": info["vgprs"], "block": info["block"],
                     "grid": info["grid"]},
        "rk4_substeps_per_s": value * a.substeps,
    }
    if gather_ms is not None:
        out["gather_ms"] = gather_ms

    if sh:
        # config 5 is fp64-VALU bound, not HBM bound: 9 fp64 instructions (7 FMA + 2 MUL = 16 flop) per
        # (l, m) entry of the padded Pines stream (2 592 entries at degree 70), four field evaluations per RK4
        # step, + ~450 fp64 instructions for the rest of the step; peak = MI355X fp64 vector 78.6 TFLOP/s
        flop = (4 * 2592 * 16 + 2 * 450) * a.substeps
        tf = n * world * a.steps * flop / el / 1e12
        out["sh"] = {"degree": 70, "fp64_tflops_executed": tf, "fp64_peak_tflops": 78.6 * world,
                     "frac_of_fp64_peak": tf / (78.6 * world),
                     "field_evals_per_s": n * world * a.steps * a.substeps * 4 / el}
    if rank == 0 and world == 1 and not a.no_extra and not sh:
        extra = {}
        # reference-faithful env step: 180 s of sim time = 1 800 RK4 sub-steps, 180 FSW updates
        ksteps = 5
        el2, km2, _ = timed_run(prop, d_act.data_ptr(), 1800, ksteps, 1, barrier, torch)
        extra["k1800"] = {"env_steps_per_s": n * ksteps / el2, "rk4_substeps_per_s": n * ksteps * 1800 / el2,
                          "fp64_tflops_algorithmic": n * ksteps * 1800 * FLOP_PER_RK4 / el2 / 1e12,
                          "kernel_ms": km2, "ms_per_step": el2 / ksteps * 1e3}
        # large-batch point where the HBM roofline is the binding limit (4 Mi envs = 1.4 GB/launch)
        nl = 1 << 22
        big = BatchedPropagator(cfg, nl, device=local)
        big.reset(sample_ic_batch(nl, n_rw, seed=1))
        d_act_big = torch.zeros(nl, dtype=torch.int32, device="cuda")
        el3, km3, _ = timed_run(big, d_act_big.data_ptr(), 1, 50, 5, barrier, torch)
        extra["large_n"] = {"envs": nl, "env_steps_per_s": nl * 50 / el3, "kernel_us": km3 * 1e3,
                            "achieved_gbs": BYTES_PER_ENV_STEP * nl / (km3 * 1e-3) / 1e9,
                            "frac_of_8TBs": BYTES_PER_ENV_STEP * nl / (km3 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        big.close()
        out["extra"] = extra
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not sh:
        out["cpu_baseline"] = cpu_baseline(cfg, n_rw, a.substeps)
    prop.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
