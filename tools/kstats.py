#!/usr/bin/env python3
"""Print calls and average duration of the kernels whose name contains one of the given
substrings, from a rocprofv3 --kernel-trace --stats output directory.
    python tools/kstats.py <dir> k_tg k_rg ..."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in sys.argv[2:]):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.1f} us")
