#!/usr/bin/env python3
"""per-kernel table of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir or kernel_stats.csv> [substring]"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(p)):
    n = r["Name"].split("(")[0].replace("void rlhip::", "")
    if pat in n:
        print(f"{n[:64]:64s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs']) / 1e3:9.2f} pct={r['Percentage']}")
