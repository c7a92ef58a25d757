#!/usr/bin/env python3
"""profiles/TAG/README.md written straight from the committed JSON / CSV files beside it (no hand transcription): kernel traces over the
boxes, the bench lines (the compact line the driver reads + the whole record), code-object facts of the shipped library, the batch-scalars
kernels, traffic and instruction mix, this round's records.  Sections whose files are missing are left out, not guessed.
usage (this container, via `tools/round.sh TAG collect`): python tools/profiles_tables.py TAG"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
D = os.path.join(ROOT, "profiles", TAG) + "/"


def J(name):
    try:
        with open(D + name) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def us(v):
    return "-" if v is None else (("%.2f" % v) if v < 100 else ("%.1f" % v))


out = ["# Round %s profiles (MI355X, gfx950, ROCm 7.2 rocprofv3)" % TAG[1:].lstrip("0"), "",
       "(Written by `tools/profiles_tables.py %s` from the JSON / CSV files beside it; produced by `tools/round.sh %s tests traces | counters isa | lines`," % (TAG, TAG),
       "one gpurun call per group, `BOX=2|3 ... traces` for the other boxes, `collect` in the build container after each.)", ""]
kt, im, st, summ = J("kernel_trace.json"), J("isa_mix.json"), J("kt_stats.json"), J("summary_latest.json")
head, rec = J("bench_default.json"), J("bench_default.extra.json")
h20, rec20 = J("bench_steps20_warmup5.json"), J("bench_steps20_warmup5.extra.json")
if kt:
    out += ["Kernel sources fingerprint (`bench.py: kernel_fingerprint()`): `%s` - `bench.py` prices its rooflines on the files here only while the tree it" % kt.get("fingerprint"),
            "runs in has the same one.  Kernel traces and every `--pmc` set are separate runs; the program follows `--` directly.", ""]
try:
    out += ["GPU suite on the final sources (`gputest_tail.txt`): `%s`" % open(D + "gputest_tail.txt").read().strip().split("\n")[-1], ""]
except OSError:
    pass

# ---------------------------------------------------------------- kernel durations
if kt:
    rows = [("65k_k1", "65k", "65 536 envs, K = 1: the headline kernel (`--steps 40000 --warmup 4000`)"),
            ("131k_k1", "131k", "131 072 envs, K = 1: configs[3], one GPU's share"), ("4m_k1", "4m", "4 Mi envs, K = 1: streams from HBM"),
            ("bare_k1800", "k1800", "65 536 envs, K = 1 800, bare"), ("power_k1800", "power_k1800", "... power level"),
            ("full_k1800", "full_k1800", "... full scenario (the drop-in env's kernel)"), ("sh70", "sh", "65 536 envs, degree-70 harmonics, K = 1")]
    out += ["## Kernel durations: rocprofv3 per-dispatch traces in steady state, beside the same commands un-profiled", "",
            "`kernel_trace.json` (key -> statistics; `boxes`, `trimmed_mean_us_median_of_boxes` where a key was traced on several boxes),",
            "`traces/kt_<run>[_b2|_b3]_dispatches.csv` (every dispatch of the step kernel: index, start offset, duration - recompute anything from these),",
            "`traces/kt_<run>..._kernel_stats.csv` (rocprofv3's own `--stats` table of the same run), `traces/ab_<run>[_b2|_b3]_plain.json` (the identical command",
            "un-profiled on the same box: its own dispatch stamps and wall time per launch).  Steady state = dispatches that start >= 25 ms after the",
            "kernel's first one; trimmed mean = mean of their middle 80 %.  All us:", "",
            "| key | dispatches | trimmed mean per box | **median over boxes (what `bench.py` uses)** | un-profiled, per box: stamped / wall per launch |", "|---|---|---|---|---|"]
    for key, ab, desc in rows:
        q = kt["runs"].get(key)
        if not q:
            continue
        boxes = q.get("boxes") or [q]
        plain = []
        for suf in ("", "_b2", "_b3"):
            p = J("traces/ab_%s%s_plain.json" % (ab, suf))
            if p:
                plain.append("%s / %s" % (us(p["roofline"].get("kernel_us_stamped")), us(p["roofline"].get("wall_us_per_launch"))))
        out.append("| `%s` (%s) | %d | %s | **%s** | %s |" % (key, desc, q["dispatches"], " / ".join(us(b["trimmed_mean_us"]) for b in boxes),
                                                          us(q.get("trimmed_mean_us_median_of_boxes", q["trimmed_mean_us"])), "; ".join(plain)))
    q = kt["runs"].get("65k_k1", {})
    out += ["", "The microsecond-scale keys move with the box and with the tracer's own timing (`profiles/r05/trace_modes.txt`: a dispatch whose start stamp falls on",
            "the previous one's end contains front-end work an un-profiled stream overlaps with the previous kernel's tail; share of such dispatches of the",
            "headline key on box 1: %s).  Un-profiled, the stream's wall time per launch bounds the kernel's average from above; `bench.py` keeps its" % ("%.2f" % q["share_of_dispatches_with_zero_gap"] if q.get("share_of_dispatches_with_zero_gap") is not None else "-"),
            "conservative rule: max(own stamps cut down to the wall time per launch, median trace over the boxes).", ""]

# ---------------------------------------------------------------- bench lines
if head and rec:
    r = rec["roofline"]
    x = rec.get("extra", {})
    out += ["## Bench lines (`bench_*.json` = the compact LAST stdout line the driver reads, `bench_*.extra.json` = the whole record of the same run)", "",
            "`bench_default.json` (%d bytes; limit 4 096): **%.4g env-steps/s**, %.2f us per step; with the batch scalars joined on the device after every step" % (len(json.dumps(head)), head["value"], head["ms_per_step"] * 1e3),
            "(`value_with_join`) %.4g.  The timed launches form the done ballot AND the per-wave reward sums (`config.batch_stats`: \"%s\")." % (head.get("value_with_join") or 0, head["config"].get("batch_stats")),
            "Roofline (HBM, 340 B per env-step): kernel %.2f us -> %.0f GB/s = **%.3f** of 8 TB/s, %.3f of the 6.29 TB/s copy ceiling (stamped %s us, wall per launch %s, trace %s);" % (
                r["kernel_us"], r["achieved"], r["frac"], r["frac_of_copy_ceiling"], us(r.get("kernel_us_stamped")), us(r.get("wall_us_per_launch")), us(r.get("kernel_us_rocprof"))),
            "counter traffic %s MB per launch = %s x the algorithmic %.2f MB (`%s`); working set %.0f MB (cache-resident: the fraction is cache bandwidth priced on the HBM peak)." % (
                us(r["traffic"] / 1e6) if r.get("traffic") else "-", ("%.2f" % (r["traffic"] / r["algorithmic_bytes"])) if r.get("traffic") else "-", r["algorithmic_bytes"] / 1e6,
                r.get("traffic_source"), r["working_set_bytes"] / 1e6)]
    if h20:
        out.append("`bench_steps20_warmup5.json` (the driver's own command, %d bytes): %.4g env-steps/s, %.2f us per step over 20 launches; `value_with_join` %.4g." % (
            len(json.dumps(h20)), h20["value"], h20["ms_per_step"] * 1e3, h20.get("value_with_join") or 0))
    K = ("k1800", "power_k1800", "full_k1800")
    if all(k in x for k in K):
        out.append("The record's other points (`bench_default.extra.json: extra`): bare / power / full scenario at K = 1 800: %s ms (%s of the nominal fp64 peak on executed flops, %s of the measured one-wave FMA ceiling %.1f TFLOP/s);" % (
            " / ".join("%.3f" % x[k]["kernel_ms"] for k in K), " / ".join("%.3f" % x[k]["roofline"]["frac"] for k in K),
            " / ".join("%.2f" % x[k]["roofline"].get("frac_of_measured_fma_ceiling", 0) for k in K), x.get("fp64_ceiling", {}).get("measured_tflops_1_wave_per_simd", 0)))
    if "sh70" in x and "large_n" in x and "config3_per_gpu" in x:
        out.append("config 5: %.1f us, %.3f algorithmic; 4 Mi: %.1f us = %.3f of 8 TB/s; configs[3] per GPU (131 072): %.2f us = %.3f;" % (
            x["sh70"]["kernel_ms"] * 1e3, x["sh70"]["roofline"]["frac"], x["large_n"]["roofline"]["kernel_us"], x["large_n"]["roofline"]["frac"],
            x["config3_per_gpu"]["roofline"]["kernel_us"], x["config3_per_gpu"]["roofline"]["frac"]))
    sb = x.get("small_batch", {})
    if "crossover" in sb:
        c = sb["crossover"]
        out.append("small batches, full scenario, one 180 s env step (launch + kernel + sync): %s ms for %s spacecraft; one host core: %.2f ms per env step -> the GPU is the faster engine from N = %s (`small_batch_crossover_n`; `rejected/tri_roles.txt`);" % (
            " / ".join("%.2f" % v for v in c["gpu_ms_per_env_step"].values()), " / ".join(c["gpu_ms_per_env_step"].keys()), c["cpu_core_ms_per_env_step"], c["n"]))
    if "rollout" in x and "65536" in x["rollout"]:
        ro = x["rollout"]
        out.append("rollouts (`bsk_step_n`, T = 541): %.2f us per env step = %.3g env-steps/s with a constant action, %.2f us with device actions at 65 536; %.1f us at 4 Mi;" % (
            ro["65536"]["constant_action"]["us_per_env_step"], ro["65536"]["constant_action"]["env_steps_per_s"], ro["65536"]["device_actions"]["us_per_env_step"],
            ro["4194304"]["constant_action"]["us_per_env_step"]))
    if "host_buffers_k1" in x and "rl_loop" in x and "k1" in x["rl_loop"]:
        out.append("host buffers on both sides at K = 1: %.0f us per step (PCIe-inclusive, never `value`); `rl_loop` k1 / k1800: loop over kernel rate %.2f / %.3f, env share over kernel %.2f." % (
            x["host_buffers_k1"]["ms_per_step"] * 1e3, x["rl_loop"]["k1"]["loop_over_kernel_rate"], x["rl_loop"]["k1800"]["loop_over_kernel_rate"], x["rl_loop"]["k1"]["env_share_over_kernel"]))
    cb = rec.get("cpu_baseline", {})
    if cb:
        out.append("CPU oracle beside it (a reported baseline, never the target): %.3g env-steps/s on %d cores, %.3g on one (K = 1); full scenario K = 1 800: %.3g / %.3g." % (
            cb["value"], cb["cores"], cb.get("single_thread", {}).get("value", 0), x.get("full_k1800", {}).get("cpu_baseline", {}).get("value", 0),
            x.get("full_k1800", {}).get("cpu_baseline", {}).get("single_thread", {}).get("value", 0)))
    r4 = J("bench_rehearsal4.json")
    if r4:
        out.append("`bench_rehearsal4.json`: `python bench.py --gpus 4` as typed, rehearsed on ONE card over gloo (numbers meaningless): ranks %s, distinct devices %s, one compact last line of %d bytes with `gather`, `strong_65536_total`, `config3_env_steps_per_s`." % (
            r4.get("ranks"), r4.get("distinct_devices"), len(json.dumps(r4))))
    out.append("Other lines: `bench_sh.json`, `bench_4m.json`, `bench_bare_k1800.json`, `bench_scenario_{power,full}_{k1,k1800}.json`.")
    out.append("")

# ---------------------------------------------------------------- code objects
try:
    co = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "code_objects.py"), "--json"], capture_output=True, text=True, check=True).stdout)
    with open(D + "code_objects.json", "w") as f:
        json.dump(co, f, indent=1)
except Exception:
    co = J("code_objects.json")
if co:
    out += ["## Code objects of the shipped `libbskgpu.so` (`code_objects.json`; `tools/code_objects.py`: the gfx950 code objects' own `.amdhsa` metadata)", "",
            "| kernel `<GRAV, NRW, DIAG, FEAT, SPLIT>` | what | VGPR | AGPR | SGPR | VGPR spills | SGPR spills | scratch B | static LDS B |", "|---|---|---|---|---|---|---|---|---|"]
    for r in co:
        out.append("| `%s` | %s | %d | %d | %d | %d | %d | %d | %d |" % (r["kernel"], r["what"], r["vgpr"], r["agpr"], r["sgpr"], r["vgpr_spill"], r["sgpr_spill"], r["scratch_bytes"], r["lds_bytes"]))
    for r in co:
        if r.get("scratch"):
            s = r["scratch"]
            out.append("")
            out.append("`%s`: %d scratch instructions among %d; %d of them inside the %d innermost loop bodies (the Pines walks' bodies are among those) - the rest park the RK4 state around the walks." % (
                r["kernel"], s["scratch_instructions"], s["instructions"], s["inside_innermost_loops"], s["innermost_loops"]))
    out += ["", "(SGPR spills live in VGPR lanes.  The rollout kernels' ~155 are the slab's row addresses and the tail arguments: written once at the kernel's start; of the 327 static `v_readlane`",
            "289 sit in the last 870 instructions of the kernel - the device-side restart block and the launch's final stores, behind the step loop - and 38 in the per-step path; the measured",
            "576 - 594 VALU instructions per wave and env step are the tick's + the epilogue's (`profiles/r05/rejected/rollout_fsw_lds.txt`).  Dynamic LDS - pair / three-wave forms, power levels - is not in the static figure.)", ""]

# ---------------------------------------------------------------- batch scalars
if st:
    bs = (rec or {}).get("extra", {}).get("batch_stats_us", {})
    out += ["## Batch scalars (`kt_stats.json`, `traces/kt_stats_<N>.csv`; `tools/exp/stats_trace.py N`: a K = 1 step + a request per iteration, 3 000 iterations)", "",
            "| spacecraft | `stats_kernel` (level 1) | join (`stats_join1_kernel`, one wave, up to 65 536; `stats_join_kernel` above) | both | added per step in the stepping loop | with `bsk_set_step_stats` (the headline's setting): join alone | added per step |", "|---|---|---|---|---|---|---|"]
    for n in ("65536", "131072", "1048576", "4194304"):
        v, f = st.get(n), st.get("fused_" + n)
        if not v or not f:
            continue
        fused = "%.2f us" % f["stats_join_kernel"]["trimmed_mean_us"] if "stats_kernel" not in f else "(two-level form kept: %.2f)" % f["both_trimmed_mean_us"]
        b = bs.get(n, {})
        out.append("| %s | %.2f us | %.2f | **%.2f** | %s us | %s | **%s us** |" % ("{:,}".format(int(n)).replace(",", " "), v["stats_kernel"]["trimmed_mean_us"], v["stats_join_kernel"]["trimmed_mean_us"],
                                                                         v["both_trimmed_mean_us"], us(b.get("added_us_per_step")), fused, us(b.get("in_launch_wave_sums", {}).get("added_us_per_step"))))
    out.append("")

# ---------------------------------------------------------------- traffic, instruction mix
if summ and summ.get("traffic"):
    t = summ["traffic"]
    out += ["## HBM traffic and issue counters (`summary_latest.json`, `isa_mix.json`)", "",
            "Traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, separate passes; the gfx950 correction of MI355X_MICROARCH.md): " +
            ", ".join("%.2f MB at %s (%.2f x algorithmic)" % (t[n]["traffic_bytes"] / 1e6, "{:,}".format(int(n)).replace(",", " "), t[n]["traffic_bytes"] / (340.0 * int(n))) for n in ("65536", "131072", "4194304") if n in t) + "."]
    if im and all(k in im for k in ("bare", "power", "full", "sh")):
        f64 = lambda k: im[k]["fma"] + im[k]["mul"] + im[k]["add"] + im[k]["trans"]      # noqa: E731
        out.append("Executed instructions per RK4 sub-step and wave: bare %.0f VALU (%.0f fp64), power %.0f, full %.0f (%.0f fp64: %.0f FMA, %.0f MUL, %.0f ADD, %.0f rcp / rsq) + %.0f scalar; VALU-active %.2f / %.2f / %.2f,"
                   " WAIT_ANY %.2f / %.2f / %.2f; harmonics %.0f VALU per wave and step, VALU-active %.2f per wave with two waves per SIMD, WAIT_ANY %.2f." % (
                       im["bare"]["valu"], f64("bare"), im["power"]["valu"], im["full"]["valu"], f64("full"), im["full"]["fma"], im["full"]["mul"], im["full"]["add"], im["full"]["trans"], im["full"]["salu"],
                       im["bare"]["valu_active_over_wave_cycles"], im["power"]["valu_active_over_wave_cycles"], im["full"]["valu_active_over_wave_cycles"],
                       im["bare"]["wait_any_over_wave_cycles"], im["power"]["wait_any_over_wave_cycles"], im["full"]["wait_any_over_wave_cycles"],
                       im["sh"]["valu"], im["sh"]["valu_active_over_wave_cycles"], im["sh"]["wait_any_over_wave_cycles"]))
    out.append("")
if os.path.exists(D + "memcopy_rl.txt"):
    out += ["## The device-resident loop under the copy tracer (`memcopy_rl.txt`)", "",
            "rocprofv3 --kernel-trace --memory-copy-trace over reset_tensors + 200 x (policy, step_tensors): " + open(D + "memcopy_rl.txt").read().split("\n")[1].strip(), ""]

# ---------------------------------------------------------------- this round's records
notes = [("stride_pad.txt", "where the state slab's 256-B row padding pays: one box, alternating, 32 768 / 65 536 / 98 304 / 131 072 / 4 Mi -> applied for 65 536 ... 98 304 spacecraft only (`bsk_capi.hip: slab_row_pad`)"),
         ("rejected/tri_roles.txt", "the ONE-spacecraft env step, role by role (probe `BSK_PROBE_TRI_ROLE`): both dynamics halves issue back to back and are level to 0.2 %; nothing to move - rejected; the crossover N is stated"),
         ("rejected/rollout_k1_overlap.txt", "rollout kernel, K = 1: the epilogue of env step s overlapped with the tick of step s + 1 (+ `.diff`): bit-identical, -1 % at 65 536 (1.55 - 1.58 against 1.58 - 1.59 us; the bar was 1.35), -4.5 % at 4 Mi - rejected"),
         ("rejected/tri_pmc_n64.txt", "SQ counters of the three forms at 64 spacecraft (instructions per tick and workgroup)"),
         ("fuzz.txt", "the three randomized GPU tests widened to 10 000 seeds each on the final sources: 40 001 passed"),
         ("forced_forms.txt", "the whole GPU suite with the pair / three-wave form forced for every launch: 442 passed each"),
         ("dpp_hazard.txt", "`tools/dpp_hazard.py` over EVERY translation unit's code object of the shipped library (until this round it read the first offload bundle only)")]
have = [(f, w) for f, w in notes if os.path.exists(D + f)]
if have:
    out += ["## This round's records", "", "| file | what |", "|---|---|"] + ["| `%s` | %s |" % fw for fw in have] + [""]
with open(D + "README.md", "w") as f:
    f.write("\n".join(out) + "\n")
print("wrote", D + "README.md", sum(len(l) + 1 for l in out), "bytes")
