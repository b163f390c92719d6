#!/usr/bin/env python3
"""Prints the vrdx rows of a rocprofv3 *kernel_stats.csv: calls, average / min / max duration in us."""
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Name"]
    if "vrdx" not in name:
        continue
    short = name.replace("void vrdx::", "").split("(")[0]
    print("    %-62s calls %5s avg_us %9.2f min_us %9.2f max_us %9.2f" % (
        short[:62], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
