#!/usr/bin/env python3
"""Groups the pass kernels of a rocprofv3 --kernel-trace CSV by pass index (launch order modulo 4) and prints the
median / mean / min duration per pass and for the histogram kernel.   pass_parity.py <kernel_trace.csv> [skip_sorts]"""
import csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
passes = [[] for _ in range(4)]
hist = []
k = 0
sorts = 0
for r in rows:
    name = r["Kernel_Name"]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "histogram_kernel" in name:
        sorts += 1
        k = 0
        if sorts > skip:
            hist.append(dur)
    elif "onesweep" in name:
        if sorts > skip:
            passes[k % 4].append(dur)
        k += 1
def line(label, v):
    if v:
        print("  %-10s n=%3d  median %8.2f  mean %8.2f  min %8.2f  max %8.2f" % (label, len(v), statistics.median(v), statistics.mean(v), min(v), max(v)))
line("histogram", hist)
for p in range(4):
    line("pass %d" % p, passes[p])
