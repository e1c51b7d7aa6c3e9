#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/collect_profiles.sh into small text/JSON summaries (these are what
gets copied into profiles/)."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def one(pattern):
    f = glob.glob(os.path.join(out, pattern))
    return f[0] if f else None


lines = []
ks = one("trace/*/*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    lines.append("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-strict (6 steps incl. warm-up and the 2 profiling steps)")
    lines.append("%-100s %7s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for r in rows[:28]:
        lines.append("%-100s %7s %12.1f %12.2f %7.2f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                         float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
open(os.path.join(out, "kernel_stats_summary.txt"), "w").write("\n".join(lines) + "\n")


KERNEL = "corr_lookup_disp_kernel"


def counter(pattern, name, kernel=KERNEL):
    """mean counter value over the kernel's largest-grid launches (the B = 11 estimator lookups of the bench)"""
    f = one(pattern)
    if not f:
        return None
    rows = [(int(r["Grid_Size"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == name and kernel in r["Kernel_Name"]]
    if not rows:
        return None
    big = max(g for g, _ in rows)
    vals = [v for g, v in rows if g == big]
    return sum(vals) / len(vals)


fetch_kb = counter("pmc_fetch/*/*counter_collection.csv", "FETCH_SIZE")
write_kb = counter("pmc_write/*/*counter_collection.csv", "WRITE_SIZE")
hit = counter("pmc_l2/*/*counter_collection.csv", "TCC_HIT_sum")
miss = counter("pmc_l2/*/*counter_collection.csv", "TCC_MISS_sum")
res = {"kernel": KERNEL, "launch": "the B=11-pair, 60x128 estimator lookups of bench.py (C3 workload, real flows of the "
                                   "benchmarked model), ACCFLOW_STREAMS=1",
       "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
       "TCC_HIT_sum": hit, "TCC_MISS_sum": miss,
       "note": "gfx950: FETCH_SIZE = 64 B per memory-side read request and reports 1/2 of the bytes of wide coalesced "
               "streaming reads (MI355X_MICROARCH.md, HBM/rocprofv3 section), so it is doubled; this kernel's reads are "
               "coalesced dword loads (256 B per wave-instruction under coherent flow).  WRITE_SIZE is exact for "
               "coalesced stores."}
if fetch_kb is not None and write_kb is not None:
    res["hbm_bytes_per_launch_raw"] = int((fetch_kb + write_kb) * 1024)
    res["hbm_bytes_per_launch"] = int((2 * fetch_kb + write_kb) * 1024)
    res["algorithmic_bytes_per_launch"] = 2904 * 11 * 60 * 128
json.dump(res, open(os.path.join(out, "lookup_traffic.json"), "w"), indent=1)
print(open(os.path.join(out, "kernel_stats_summary.txt")).read())
print(json.dumps(res, indent=1))
for f in ("lookup_bench.log",):
    p = os.path.join(out, f)
    if os.path.exists(p):
        print(open(p).read())
