#!/usr/bin/env python3
"""Markdown rows of profiles/TAG/README.md straight from the committed JSON / CSV files (no hand transcription).
usage: tools/profiles_tables.py r03"""
import json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", tag)
J = lambda n: json.load(open(os.path.join(d, n)))
kt = J("kernel_trace.json")
print("fingerprint", kt["fingerprint"])
CMD = {"65k_k1": "`--steps 40000 --warmup 4000` (65 536 envs, K = 1: the headline kernel)", "4m_k1": "`--envs 4194304 --steps 20 --warmup 3`",
       "bare_k1800": "`--substeps 1800 --steps 20 --warmup 10`", "power_k1800": "`--scenario power --substeps 1800 --steps 20 --warmup 10`",
       "full_k1800": "`--scenario full --substeps 1800 --steps 20 --warmup 10`", "sh70": "`--gravity sh --steps 1000 --warmup 300`"}
def sp(x, nd=2):     # 1 234.56 with a thin-space-free plain space as in the README
    t = ("%%.%df" % nd) % x
    i, _, f = t.partition(".")
    i = i[::-1]; i = " ".join(i[k:k + 3] for k in range(0, len(i), 3))[::-1]
    return i + ("." + f if f else "")
print("\n| key | dispatches (ramp dropped) | mean of ALL | p10 / p90 | median | trimmed mean | un-profiled stamped pass (median) |")
ab = {"65k_k1": "ab_65k_plain.json", "4m_k1": "ab_4m_plain.json", "bare_k1800": "ab_k1800_plain.json", "power_k1800": "ab_power_k1800_plain.json",
      "full_k1800": "ab_full_k1800_plain.json", "sh70": "ab_sh_plain.json"}
for k in ("65k_k1", "4m_k1", "bare_k1800", "power_k1800", "full_k1800", "sh70"):
    v = kt["runs"][k]
    try:
        r = J(ab[k])["roofline"]; plain = "%.2f (%.2f)" % (r["kernel_us_stamped"], r.get("median_us", float("nan")))
    except Exception as e:
        plain = "n/a"
    try:
        plain = "%s (%s)" % (sp(r["kernel_us_stamped"]), sp(r.get("median_us", float("nan"))))
    except Exception:
        pass
    print("| `%s` | %s | %s (%s%s) | %s | %s / %s | %s | **%s** | %s%s |" % (k, CMD[k], sp(v["dispatches"], 0), sp(v["ramp_dispatches_dropped"], 0), " ramp dropped" if k == "65k_k1" else "", sp(v["mean_all_us"]), sp(v["p10_us"]), sp(v["p90_us"]), sp(v["median_us"]), sp(v["trimmed_mean_us"]), plain.replace("(", "(median " if k == "65k_k1" else "("), ""))
b = J("bench_default.json")
r = b["roofline"]; e = b["extra"]
print("\nbench_default: value %.3e, ms_per_step %.5f, stamped %.2f us, rocprof %.2f (fresh %s), frac %.3f, frac_stamped %.3f" % (b["value"], b["ms_per_step"], r["kernel_us_stamped"], r.get("kernel_us_rocprof") or float("nan"), r.get("kernel_us_rocprof_fresh"), r["frac"], r.get("frac_stamped", float("nan"))))
for k in ("k1800", "power_k1800", "full_k1800", "sh70"):
    x = e[k]; rr = x["roofline"]
    print("  extra.%s: kernel %.3f ms (roofline kernel_us %.1f), frac %.3f, %.3e env-steps/s, cpu %.3e" % (k, x["kernel_ms"], rr["kernel_us"], rr["frac"], x["env_steps_per_s"], x.get("cpu_baseline", {}).get("value", float("nan"))))
ln = e["large_n"]; print("  extra.large_n: frac %.3f achieved %.0f GB/s kernel_us %.1f copy-ceiling frac %.3f" % (ln["roofline"]["frac"], ln["roofline"]["achieved"], ln["roofline"]["kernel_us"], ln["roofline"].get("frac_of_copy_ceiling", float("nan"))))
print("  extra.fp64_ceiling:", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in e["fp64_ceiling"].items() if k != "method"})
print("  extra.small_batch:", {k: round(v["ms_per_env_step"], 3) for k, v in e["small_batch"].items()}, e["small_batch"]["1"]["kernel"])
for k in ("k1", "k1800"):
    x = e["rl_loop"][k]; print("  extra.rl_loop.%s: %.4f ms per step, kernel %.4f ms, loop/kernel %.3f, %.3e env-steps/s" % (k, x["ms_per_step"], x["kernel_ms"], x["loop_over_kernel_rate"], x["env_steps_per_s"]))
print("  cpu_baseline: %.3e (%d cores)" % (b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"]))
for f in ("bench_steps20_warmup5.json", "bench_scenario_power_k1.json", "bench_scenario_full_k1.json", "bench_bare_k1800.json", "bench_scenario_power_k1800.json", "bench_scenario_full_k1800.json", "bench_sh.json", "bench_4m.json"):
    try:
        x = J(f); print("%s: value %.3e ms_per_step %.5f kernel_us %.2f frac %.3f" % (f, x["value"], x["ms_per_step"], x["roofline"]["kernel_us"], x["roofline"]["frac"]))
    except Exception as ex:
        print(f, "missing", ex)
im = J("isa_mix.json")
print("\n| level | FMA | MUL | ADD | rcp/rsq | all VALU | SALU | VALU active / wave cycles | WAIT_ANY |")
for k in ("bare", "power", "full", "sh"):
    v = im[k]; print("| %s | %.1f | %.1f | %.1f | %.1f | %.0f | %.1f | %.2f | %.3f |" % (k, v["fma"], v["mul"], v["add"], v["trans"], v["valu"], v["salu"], v["valu_active_over_wave_cycles"], v["wait_any_over_wave_cycles"]))
s = J("summary_latest.json")
print("\ntraffic:", json.dumps(s.get("traffic"), indent=0)[:600])
l = J("latency_box.json"); print("\nlatency:", l)
