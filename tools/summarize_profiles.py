#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/collect_profiles.sh into small text/JSON summaries (these are what
gets copied into profiles/).  usage: summarize_profiles.py OUT_DIR TAG"""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
N_SIMD = 1024            # 256 CUs x 4 SIMDs
MFMA_FLOP_PER_CYCLE = 1024  # 16-bit 32x32x16 MFMA: 32 768 flop in 32 cycles per SIMD (2.5 PF at 2.4 GHz)


def one(pattern):
    f = glob.glob(os.path.join(out, pattern))
    return f[0] if f else None


def stats(sub, title, dst):
    ks = one(sub + "/*/*kernel_stats.csv")
    if not ks:
        return
    rows = list(csv.DictReader(open(ks)))
    lines = [title, "%-100s %7s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%")]
    for r in rows[:30]:
        lines.append("%-100s %7s %12.1f %12.2f %7.2f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                         float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    open(os.path.join(out, dst), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


stats("trace", "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-strict --no-extra "
      "(default: 2 pair-group streams + the sequence pipeline's side stream; 4 pipelined steps (1 warm-up + 3), 4 one-at-a-time steps (1 + 3) and the 2 single-stream profiling steps)",
      "%s_kernel_stats_bench.txt" % tag)
stats("trace1", "ACCFLOW_STREAMS=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --no-pipeline --steps 3 --warmup 1 --no-strict "
      "--no-extra (ONE stream, one sequence at a time: every estimator launch covers all 11 pairs; 6 steps)", "%s_kernel_stats_bench_1stream.txt" % tag)


stats("trace_c5", "ACCFLOW_STREAMS=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --ofe gma --height 720 --width 1280 "
      "--steps 2 --warmup 1 --no-pipeline --no-extra (configs[4] AccFlow(GMA) 7x720x1280, ONE stream; 5 sequence evaluations)",
      "%s_kernel_stats_c5_1stream.txt" % tag)


def counters(sub):
    """-> list of dict(kernel, grid, dur_ns, {counter: value}) per dispatch"""
    f = one(sub + "/*/*counter_collection.csv")
    if not f:
        return []
    by = {}
    for r in csv.DictReader(open(f)):
        d = by.setdefault(r["Dispatch_Id"], {"kernel": r["Kernel_Name"], "grid": int(r["Grid_Size"]),
                                              "dur_ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(by.values())


LOOKUP = "corr_lookup_disp_kernel"
ALG = 2904 * 11 * 60 * 128


def mean_big(rows, name, kernel):
    rows = [(d["grid"], d["c"][name]) for d in rows if kernel in d["kernel"] and name in d["c"]]
    if not rows:
        return None
    big = max(g for g, _ in rows)
    v = [x for g, x in rows if g == big]
    return sum(v) / len(v)


fetch_kb = mean_big(counters("pmc_fetch"), "FETCH_SIZE", LOOKUP)
write_kb = mean_big(counters("pmc_write"), "WRITE_SIZE", LOOKUP)
cal_kb = mean_big(counters("pmc_cal"), "FETCH_SIZE", LOOKUP)
known_read = (1600 + 8) * 11 * 60 * 128
res = {"kernel": LOOKUP, "launch": "the B=11-pair, 60x128 estimator lookups of bench.py (C3 workload, real flows of the "
                                   "benchmarked model), ACCFLOW_STREAMS=1",
       "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
       "algorithmic_bytes_per_launch": ALG}
if cal_kb:
    factor = known_read / (cal_kb * 1024)
    res["fetch_calibration"] = {
        "launch": "tools/lookup_bench.py --layout disp --flow 0 (B=11, 60x128): every byte of the 4 x 10x10 windows is read "
                  "exactly once, %d B per launch known a priori" % known_read,
        "FETCH_SIZE_KB_per_launch": cal_kb, "known_read_bytes": known_read, "factor_known_over_counter": round(factor, 4),
        "note": "MI355X_MICROARCH.md: FETCH_SIZE = TCC_EA0_RDREQ x 64 B reports 1/2 of wide streaming reads and is "
                "uncalibrated for other widths; this kernel issues dword-per-lane buffer loads, 256 B per wave-instruction "
                "under coherent flow, so the factor is measured here instead of assumed"}
    if fetch_kb is not None and write_kb is not None:
        res["hbm_bytes_per_launch_raw"] = int((fetch_kb + write_kb) * 1024)
        res["hbm_bytes_per_launch"] = int((factor * fetch_kb + write_kb) * 1024)
elif fetch_kb is not None and write_kb is not None:
    res["hbm_bytes_per_launch_raw"] = int((fetch_kb + write_kb) * 1024)
    res["hbm_bytes_per_launch"] = int((2 * fetch_kb + write_kb) * 1024)
    res["note"] = "no calibration run: FETCH_SIZE doubled per the guide's streaming-read correction"
# the product path runs the lookup FUSED with convc1 (csrc/corr_lookup_conv.hip): the same dword-per-lane window loads (the
# calibration factor above applies to them; the weight fragments are 16-byte loads that hit L2 and do not reach the
# memory-side counters) + the pre-split output
FUSED = "corr_lookup_convc1_ws_kernel"
ffetch = mean_big(counters("pmc_fetch"), "FETCH_SIZE", FUSED)
fwrite = mean_big(counters("pmc_write"), "WRITE_SIZE", FUSED)
if ffetch is not None and fwrite is not None:
    fac = res.get("fetch_calibration", {}).get("factor_known_over_counter", 2.0)
    res["fused"] = {"kernel": FUSED, "FETCH_SIZE_KB_per_launch": ffetch, "WRITE_SIZE_KB_per_launch": fwrite,
                    "algorithmic_bytes_per_launch": (1608 + 1024) * 11 * 60 * 128,
                    "hbm_bytes_per_launch": int((fac * ffetch + fwrite) * 1024),
                    "note": "window reads scaled by the stand-alone lookup's calibration factor (same load instructions), "
                            "writes exact; algorithmic = 1 608 B read + 1 024 B written per query pixel"}
json.dump(res, open(os.path.join(out, "%s_lookup_traffic.json" % tag), "w"), indent=1)
print(json.dumps(res, indent=1))

# matrix-pipe / stall counters per kernel family (single-stream bench step)
sq = counters("pmc_sq")
fam = {}
for d in sq:
    k = d["kernel"]
    name = None
    for key in ("corr_lookup_convc1_ws_kernel", "conv2d_direct_bf16s_kernel<2, 2, true, true, false, true>", "conv2d_direct_bf16s_kernel<1, 2, true, false, false, true>",
                "corr_disp_ring_kernel", "corr_disp_gemm_kernel", "conv2d_bf16s_kernel", "conv2d_direct_bf16s_kernel"):
        if key in k:
            name = key
            break
    if name is None or "GRBM_GUI_ACTIVE" not in d["c"]:
        continue
    f = fam.setdefault(name, {"launches": 0, "dur_ns": 0, "c": {}})
    f["launches"] += 1
    f["dur_ns"] += d["dur_ns"]
    for c, v in d["c"].items():
        f["c"][c] = f["c"].get(c, 0.0) + v
busy = {"source": "ACCFLOW_STREAMS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS "
                  "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --no-pipeline --steps 1 "
                  "--warmup 0 (sums over every launch of the kernel in 3 steps: 1 timed + 2 profiling)",
        "units": "MFMA_BUSY in cycles summed over the 1024 SIMDs (32 per 32x32x16 16-bit MFMA); GRBM_GUI_ACTIVE summed over "
                 "the 8 XCDs; SQ_WAVE_CYCLES / WAIT_* / ACTIVE_* in quad-cycles summed over waves; durations are under the "
                 "profiler (counter collection serialises launches; clocks read 2-5 % low)",
        "kernels": {}}
for name, f in fam.items():
    c = f["c"]
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    e = {"launches": f["launches"], "total_ms": round(f["dur_ns"] / 1e6, 3),
         "effective_clock_GHz": round(cyc / f["dur_ns"], 3),
         "mfma_pipe_busy_frac": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * N_SIMD), 4),
         "mfma_TFLOPs_executed": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * MFMA_FLOP_PER_CYCLE / f["dur_ns"] / 1e3, 1),
         "frac_of_2.5PF": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * MFMA_FLOP_PER_CYCLE / f["dur_ns"] / 1e3 / 2500.0, 4)}
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        e["wave_cycle_split"] = {k: round(c.get(n, 0.0) / wc, 4) for k, n in (
            ("active_inst_any", "SQ_ACTIVE_INST_ANY"), ("wait_any(s_waitcnt/barrier)", "SQ_WAIT_ANY"),
            ("wait_inst_any(issue stall)", "SQ_WAIT_INST_ANY"), ("wait_inst_lds", "SQ_WAIT_INST_LDS"))}
    busy["kernels"][name] = e
json.dump(busy, open(os.path.join(out, "%s_conv_mfma_busy.json" % tag), "w"), indent=1)
print(json.dumps(busy, indent=1))
p = os.path.join(out, "lookup_bench.log")
if os.path.exists(p):
    print(open(p).read())
