#!/usr/bin/env python3
"""Prints the vrdx rows of a rocprofv3 *kernel_stats.csv: calls, average / min / max duration in us.
With --json OUT: also writes them, stamped with the digest of the kernel sources and the library's tile choice, as the
profiles/kernel_stats.json that bench.py quotes next to its own event-timed figures (roofline.rocprof)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Name"]
    if "vrdx" not in name:
        continue
    short = name.replace("void vrdx::", "").split("(")[0]
    print("    %-62s calls %5s avg_us %9.2f min_us %9.2f max_us %9.2f" % (
        short[:62], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
    if "lds_order_check" not in short and "spin_kernel" not in short:
        rows[short] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3,
                       "min_us": float(row["MinNs"]) / 1e3, "max_us": float(row["MaxNs"]) / 1e3}
if "--json" in sys.argv:
    sys.path.insert(0, ROOT)
    import bench
    import vulkan_radix_sort_amd as vrdx
    out = sys.argv[sys.argv.index("--json") + 1]
    with open(out, "w") as f:
        json.dump({"file": os.path.basename(sys.argv[1]), "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-sweep",
                   "measured_on": os.uname().nodename,
                   "kernel_source_sha256": bench.kernel_source_digest(), "library": vrdx.version_string(),
                   "kernels": rows}, f, indent=1)
        f.write("\n")
