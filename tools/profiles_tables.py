#!/usr/bin/env python3
"""profiles/r05/README.md written straight from the committed JSON / CSV files (no hand transcription): kernel traces over the boxes,
the default bench line and its extras, the batch-scalars kernels, rollouts, VecEnv, traffic and instruction mix, the rejected list.
usage (this container, after tools/collect_evidence.sh r05): python tools/profiles_tables.py      (rounds 3 / 4 had their own table scripts:
the history of this file)"""
import json
import os

d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05") + "/"
J = lambda n: json.load(open(d + n))      # noqa: E731
kt, im, bd, st, s = J("kernel_trace.json"), J("isa_mix.json"), J("bench_default.json"), J("kt_stats.json"), J("summary_latest.json")["traffic"]
x, r = bd["extra"], bd["roofline"]
us = lambda v: ("%.2f" % v) if v < 100 else ("%.1f" % v)      # noqa: E731


def boxes(k):
    q = kt["runs"][k]
    return " / ".join(us(b["trimmed_mean_us"]) for b in q["boxes"]), q["trimmed_mean_us_median_of_boxes"], q["dispatches"]


def plain(k):
    return " / ".join(us(J("ab_%s%s_plain.json" % (k, suf))["roofline"]["kernel_us_stamped"]) for suf in ("", "_b2", "_b3"))


rows = [("65k_k1", "65k", "`--steps 40000 --warmup 4000`: 65 536 envs, K = 1, the headline kernel", "7.13"),
        ("131k_k1", "131k", "`--envs 131072 --steps 20000 --warmup 2000`: configs[3], one GPU's share", "8.15"),
        ("4m_k1", "4m", "`--envs 4194304 --steps 20 --warmup 3`", "285.5"),
        ("bare_k1800", "k1800", "`--substeps 1800 --steps 20 --warmup 10`", "1 575.8"),
        ("power_k1800", "power_k1800", "`--scenario power ...`", "2 073.2"),
        ("full_k1800", "full_k1800", "`--scenario full ...`", "3 246.1"),
        ("sh70", "sh", "`--gravity sh --steps 1000 --warmup 300`", "253.6")]
tab = "| key | dispatches | trimmed mean, box 1 / 2 / 3 | **median (used)** | round 4 (median of five) | un-profiled stamped pass, box 1 / 2 / 3 |\n|---|---|---|---|---|---|\n"
for key, ab, desc, r4 in rows:
    b, m, n = boxes(key)
    tab += "| `%s` (%s) | %d | %s | **%s** | %s | %s |\n" % (key, desc, n, b, us(m), r4, plain(ab))
fma = lambda k: im[k]["fma"] + im[k]["mul"] + im[k]["add"] + im[k]["trans"]      # noqa: E731
K = ("k1800", "power_k1800", "full_k1800")
txt = """# Round 5 profiles (MI355X, gfx950, ROCm 7.2 rocprofv3)

(Written by `tools/profiles_tables.py` from the JSON / CSV files beside it.)

Everything here was taken on the FINAL sources of the stepping path (`bench.py: kernel_fingerprint()` = `%s`, recorded in
`kernel_trace.json` and `isa_mix.json`; `bench.py` prices its rooflines on these files only while the fingerprint of the tree it runs in
agrees).  The evidence pass of the round, on the final sources: `tools/round.sh r05` (the whole GPU suite - 458 passed, 12 skipped -; kernel traces; HBM traffic and issue
counter passes; the batch-scalars kernels at four sizes; the device-resident loop under the copy tracer; plain A/B lines; latency),
`tools/kt_boxes.sh r05 2|3` (every trace again on two more boxes), `tools/isa_mix.sh r05`, then `tools/bench_lines.sh r05` with the
summaries in place; summarised by `tools/kernel_trace_summary.py`, `tools/prof_summary.py`, `tools/isa_mix_summary.py`,
`tools/collect_evidence.sh`.  Kernel traces and every `--pmc` set are separate runs.

The step kernel's arithmetic did not change this round: the code objects are round 4's instruction streams spread over eight translation
units, plus one optional block at the end of the epilogue (the per-wave reward sums of `bsk_set_step_stats`, skipped by a scalar branch when
off: 195 296 DPP instructions and 253 `s_nop` in sum) - a same-box A/B of the K = 1 launch with and without the block reads 6.34 against
6.34 us wall per launch (`ab_step_stats_block.txt`).  What is new is measured in the sections "Batch scalars", "Rollouts", "VecEnv" and
`rejected/`.

## Kernel durations: rocprofv3 per-dispatch traces in steady state on THREE boxes, beside the same commands un-profiled

`kernel_trace.json` (key -> statistics, `boxes`, `trimmed_mean_us_median_of_boxes`), `kt_<run>[_b2|_b3]_dispatches.csv` (every dispatch of the
step kernel: index, start offset, duration - recompute anything from these), `kt_<run>..._kernel_stats.csv` (rocprofv3's own `--stats`
table of the same run), `ab_<run>[_b2|_b3]_plain.json` (the identical bench command un-profiled on the same box).  "Steady state" = the
dispatches that start >= 25 ms after the kernel's first one.  `bench.py` prices on the median over the boxes.  All us:

%s
The microsecond-scale keys move with the box AND with the profiler's own timing: `trace_modes.txt` (one box, the same library traced six times) shows
two modes for the headline kernel - ~6.0 us where the tracer leaves a gap between dispatches, ~7.2 us where a dispatch's start stamp falls on the
previous one's end and the duration then contains front-end work that overlaps the previous kernel's tail in an un-profiled stream
(`share_of_dispatches_with_zero_gap` in `kernel_trace.json`).  Un-profiled the stream runs at 6.19 - 6.23 us per launch on all three boxes, which
bounds the kernel's true average from above; `bench.py` nevertheless keeps its conservative rule - max(own stamps cut down to the wall time per
launch, median trace over the boxes) - so the headline is priced on %.2f us = %.3f of 8 TB/s where its own launches average <= %.2f us (%.3f).
The fp64-bound keys agree with round 4 within 1 %%.

**The slab's padding** (the round's one change to the K = 1 launch: 256 B more per field row; measured, the mechanism not established - the
micro-benchmark of the bare access pattern, `row_channels.txt`, is FASTEST at the power-of-two stride, so it is not "46 rows in one L2 channel"): same-box un-profiled sweep `stride_pad.txt` 6.33 - 6.38 -> 6.21 us wall per launch (6.36 -> 6.13 with every row of the handle padded; 512 B a
third of it, 4 KB nothing); the three boxes' plain lines 6.33 - 6.36 (previous pass) -> 6.19 - 6.23; nothing at 131 072, 4 Mi or K = 1 800.  The tracer
does not resolve it (`trace_modes.txt`).

## Bench lines (`bench_*.json`; the default line is what the driver runs)

`bench_default.json`: **%.4g env-steps/s**, %.2f us per step; roofline (HBM, 340 B per env-step): kernel %.2f us on the rule above ->
achieved %.0f GB/s = **%.3f** of 8 TB/s (stamped %.2f us, wall per launch %.2f; trace median %.2f); `frac_on_bytes_moved` %.3f (332 B move with the
static obs[3]); counter traffic %.2f MB per launch = %.2f x the algorithmic %.2f MB; `working_set`: "%s".
`extra` (all in the one default line): bare / power / full scenario at K = 1 800: %.3f / %.3f / %.3f ms (%.3f / %.3f / %.3f of the nominal fp64 peak on
executed flops, %.2f / %.2f / %.2f of the measured one-wave FMA ceiling %.1f TFLOP/s); config 5: %.1f us, %.3f algorithmic; 4 Mi: %.1f us = %.3f of 8 TB/s;
configs[3] per GPU (131 072): %.2f us = %.3f; `host_buffers_k1` %.0f us per step; `rl_loop` k1 / k1800: loop over kernel rate %.2f / %.3f, env share
over kernel %.2f; small batches (1 / 64 / 8 192 spacecraft, full scenario, three-wave form): %.2f / %.2f / %.2f ms per 180 s env step.
CPU oracle beside it: %.3g env-steps/s on %d cores, %.3g on one (K = 1); full scenario K = 1 800: %.3g / %.3g.
Other lines: `bench_steps20_warmup5.json` (the driver's window: 20 launches), `bench_sh.json`, `bench_4m.json`, `bench_bare_k1800.json`,
`bench_scenario_{power,full}_{k1,k1800}.json`, `bench_rehearsal2.json` (two ranks on one card over gloo: `ranks`, `distinct_devices`).

## Batch scalars on demand (`kt_stats.json`, `kt_stats_<N>.csv`; `tools/exp/stats_trace.py N`: a K = 1 step + a request per iteration, 3 000 iterations)

| spacecraft | `stats_kernel` (level 1) | `stats_join_kernel` | both | added per step in the stepping loop (`extra.batch_stats_us`) | with `bsk_set_step_stats`: join alone | added per step |
|---|---|---|---|---|---|---|
""" % ((kt["fingerprint"], tab, r["kernel_us"], r["frac"], r["wall_us_per_launch"], r["algorithmic_bytes"] / r["wall_us_per_launch"] / 8e6, bd["value"], bd["ms_per_step"] * 1e3, r["kernel_us"], r["achieved"], r["frac"], r["kernel_us_stamped"], r["wall_us_per_launch"],
        r["kernel_us_rocprof"], r["frac_on_bytes_moved"], r["traffic"] / 1e6, r["traffic"] / r["algorithmic_bytes"], r["algorithmic_bytes"] / 1e6, r["working_set"])
       + tuple(x[k]["kernel_ms"] for k in K) + tuple(x[k]["roofline"]["frac"] for k in K) + tuple(x[k]["roofline"]["frac_of_measured_fma_ceiling"] for k in K)
       + (x["fp64_ceiling"]["measured_tflops_1_wave_per_simd"], x["sh70"]["kernel_ms"] * 1e3, x["sh70"]["roofline"]["frac"], x["large_n"]["roofline"]["kernel_us"],
          x["large_n"]["roofline"]["frac"], x["config3_per_gpu"]["roofline"]["kernel_us"], x["config3_per_gpu"]["roofline"]["frac"], x["host_buffers_k1"]["ms_per_step"] * 1e3,
          x["rl_loop"]["k1"]["loop_over_kernel_rate"], x["rl_loop"]["k1800"]["loop_over_kernel_rate"], x["rl_loop"]["k1"]["env_share_over_kernel"],
          x["small_batch"]["1"]["ms_per_env_step"], x["small_batch"]["64"]["ms_per_env_step"], x["small_batch"]["8192"]["ms_per_env_step"],
          bd["cpu_baseline"]["value"], bd["cpu_baseline"]["cores"], bd["cpu_baseline"]["single_thread"]["value"], x["full_k1800"]["cpu_baseline"]["value"],
          x["full_k1800"]["cpu_baseline"]["single_thread"]["value"]))
for n in ("65536", "131072", "1048576", "4194304"):
    v, f = st[n], st["fused_" + n]
    fused = "%.2f us" % f["stats_join_kernel"]["trimmed_mean_us"] if "stats_kernel" not in f else "(two-level form kept: %.2f)" % f["both_trimmed_mean_us"]
    txt += "| %s | %.2f us | %.2f | **%.2f** | %.2f us | %s | **%.2f us** |\n" % ("{:,}".format(int(n)).replace(",", " "), v["stats_kernel"]["trimmed_mean_us"], v["stats_join_kernel"]["trimmed_mean_us"],
                                                                            v["both_trimmed_mean_us"], x["batch_stats_us"][n]["added_us_per_step"], fused, x["batch_stats_us"][n]["in_launch_wave_sums"]["added_us_per_step"])
ro, ve = x["rollout"], x["vecenv_episode_end_ms"]
txt += """
Round 4's kernel (one workgroup walking everything) was never measured.  The three single-launch forms built first and their numbers:
`rejected/stats_forms.txt`.  `bsk_set_step_stats` (for consumers that ask after every step): the step launch forms the per-wave sums in its
epilogue (+0.02 ... 0.06 us on the K = 1 launch, nothing when off) and a request is the join kernel alone, up to 2 Mi spacecraft
(`kt_stats_fused_<N>.csv`).  The order of the sum is unchanged either way (tests/test_gpu_device_surface.py: bit for bit from 1 to 4 Mi
spacecraft, in every kernel form).

## Rollouts (`rollout_time.txt`, `pmc_rollout.txt`; `tools/exp/rollout_time.py`, `extra.rollout` of the default line)

`bsk_step_n`: T env steps of one RK4 sub-step per launch against one launch per env step, wall time per env step on one box.  65 536
spacecraft: 6.39 us per step as single launches, 2.48 at T = 10, 1.67 at T = 100, **1.60 at T = 541 (an episode; 4.1e10 env-steps/s)** with
a constant action, 1.70 with per-step actions; 4 Mi spacecraft: 284 -> 88.6 us per env step (4.7e10).  In the default bench line:
%.2f us / %.3g env-steps/s (constant) and %.2f us (device actions) at 65 536, %.1f us / %.3g at 4 Mi.  Counter pass (before the action block
prefetch): 594 VALU + 49 SALU per wave and env step, VALU-active 0.62, WAIT_ANY 0.25.

## VecEnv at a synchronized episode end (`extra.vecenv_episode_end_ms`)

65 536 envs, `max_length` = 2 so that every env finishes on every third step: `step_wait` **%.2f ms** when all finish against %.2f ms for an
ordinary step with the device pool (278 ms before this round on the oracle-backed engine), %.1f ms with host resets (425 ms).

## HBM traffic and issue counters (`summary_latest.json`, `isa_mix.json`)

Traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, separate passes): %.2f MB at 65 536 (%.2f x algorithmic), %.2f MB at 131 072, %.1f MB at 4 Mi
(%.2f x).  Executed instruction mix per RK4 sub-step and wave (unchanged against round 4): bare %.0f VALU (%.0f fp64), power %.0f, full %.0f
(%.0f fp64: %.0f FMA, %.0f MUL, %.0f ADD, %.0f rcp / rsq) + %.0f scalar; VALU-active %.2f / %.2f / %.2f, WAIT_ANY %.2f / %.2f / %.2f; harmonics %.0f VALU per wave
and step, VALU-active %.2f per wave with two waves per SIMD, WAIT_ANY %.2f.

## The device-resident loop under the copy tracer (`memcopy_rl.txt`)

rocprofv3 --kernel-trace --memory-copy-trace over reset_tensors + 200 x (policy, step_tensors): no memory-copy record, three kernels per
step (two of the policy's, one step kernel), the library's own counters 0 copies / 0 synchronisations inside the loop.

## Rejected this round (`rejected/`)

| file | what | result |
|---|---|---|
| `tiled_layout.txt` (+ `.diff`) | wave-tiled state layout for the K = 1 launch, plain and pair-interleaved (`global_load_dwordx4`) | -3.0 %% at 65 536 (4 %% asked), pairs slower than plain tiles, +1 ... 12 %% / +25 %% at 4 Mi: the layout stays |
| `sh_three_waves.txt` | a third wave per SIMD for config 5 (168-VGPR build, RK4 state parked around the walks) | 3.99 ns per spacecraft against 3.86 / 3.88 with two waves: not built |
| `stats_forms.txt` | single-launch forms of the batch-scalars reduction (fences; write-through + ticket), an atomic done counter | 24 - 236 us, 7.6 - 99 us, 28 us when every env is done: two launches with per-workgroup partials shipped |
| `../stride_pad.txt` (end) | the observation / reward rows padded like the slab's | at most 1 %% on the K = 1 launch, costs the contiguity of everything a consumer sees: not adopted |
| `buffer_io.txt` | the slab through buffer instructions (32-bit scalar row offsets: 48 fewer scalar, 75 fewer instructions in the K = 1 kernel) | 6.18 - 6.21 against 6.20 - 6.21 us: the launch is not bound by its wave's instruction count |
| `rollout_fsw_lds.txt` | rollout kernel: FSW constants in LDS for the launch + an explicit wait in the restart branch (so that no load queues behind the history stores), with and without a one-wave register budget; a K = 1 body of its own | 1.95 / 1.68 us per env step against 1.58, the K = 1 body 1.81 (1.56 without history): the loop does not wait for memory or for its branches (counters in the file) |
""" % (ro["65536"]["constant_action"]["us_per_env_step"], ro["65536"]["constant_action"]["env_steps_per_s"], ro["65536"]["device_actions"]["us_per_env_step"],
       ro["4194304"]["constant_action"]["us_per_env_step"], ro["4194304"]["constant_action"]["env_steps_per_s"],
       ve["device_pool"]["all_done_step_wait_ms"], ve["device_pool"]["ordinary_step_wait_ms"], ve["host_resets"]["all_done_step_wait_ms"],
       s["65536"]["traffic_bytes"] / 1e6, s["65536"]["traffic_bytes"] / (340 * 65536), s["131072"]["traffic_bytes"] / 1e6, s["4194304"]["traffic_bytes"] / 1e6,
       s["4194304"]["traffic_bytes"] / (340 * 4194304), im["bare"]["valu"], fma("bare"), im["power"]["valu"], im["full"]["valu"], fma("full"), im["full"]["fma"],
       im["full"]["mul"], im["full"]["add"], im["full"]["trans"], im["full"]["salu"],
       im["bare"]["valu_active_over_wave_cycles"], im["power"]["valu_active_over_wave_cycles"], im["full"]["valu_active_over_wave_cycles"],
       im["bare"]["wait_any_over_wave_cycles"], im["power"]["wait_any_over_wave_cycles"], im["full"]["wait_any_over_wave_cycles"],
       im["sh"]["valu"], im["sh"]["valu_active_over_wave_cycles"], im["sh"]["wait_any_over_wave_cycles"])
open(d + "README.md", "w").write(txt)
print("wrote", d + "README.md", len(txt), "bytes")
