#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/probes/pmc_calibrate into HBM bytes
per launch for the sort kernels, calibrated on the copy kernels of the same run.

    python tools/pmc_report.py <fetch_counter_collection.csv> <write_counter_collection.csv> N [out.json]

The output is stamped with the library it was measured on (its version string and the SHA-256 of the
kernel / host sources): bench.py quotes `roofline.traffic` from profiles/pmc_traffic.json only while
that stamp matches the library it is benchmarking, and reports null otherwise.
"""
import csv
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_digest():
    h = hashlib.sha256()
    for name in ("vrdx_kernels.hip", "vrdx_kernels.h", "vrdx_layout.h", "vrdx_api.cpp"):
        with open(os.path.join(ROOT, "vulkan_radix_sort_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def library_version():
    sys.path.insert(0, ROOT)
    try:
        import vulkan_radix_sort_amd as vrdx
        return vrdx.version_string()
    except Exception as e:  # the stamp is then unusable, and bench.py will say traffic = null
        return "unknown (%s)" % e


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def pick(d, needle):
    for k, v in d.items():
        if needle in k:
            return v
    raise KeyError(needle)


def main():
    fetch_csv, write_csv, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fetch, calls = per_kernel(fetch_csv, "FETCH_SIZE")
    write, _ = per_kernel(write_csv, "WRITE_SIZE")
    known = 4.0 * n
    f4 = known / pick(fetch, "copy_dword")      # bytes per FETCH_SIZE unit, 4 B/lane reads
    f16 = known / pick(fetch, "copy_uint4")     # 16 B/lane reads
    w4 = known / pick(write, "copy_dword")
    w16 = known / pick(write, "copy_uint4")
    out = {"n": n, "library": library_version(), "kernel_source_sha256": kernel_source_digest(),
           "measured_on": os.uname().nodename, "plan_kernels": {}, "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; counter units "
                             "calibrated on copy kernels of known size in the same run (4 B/lane and 16 B/lane)",
           "bytes_per_fetch_unit": {"dword": f4, "uint4": f16}, "bytes_per_write_unit": {"dword": w4, "uint4": w16},
           "kernels": {}}
    for name in fetch:
        if "onesweep" in name:
            rd, wr = fetch[name] * f4, write.get(name, 0.0) * w4
            # onesweep_kernel<THREADS, KPT, KV, ATOMIC_RANK, DYN>: third argument; onesweep_pair_kernel<THREADS, KPT, DYN> is keys-only
            targs = [t.strip() for t in name.split("<", 1)[1].split(">", 1)[0].split(",")]
            kind = "key_value" if "pair" not in name and len(targs) > 2 and targs[2] == "true" else "keys"
        elif "scatter_msd" in name or "bucket_sort2" in name or "msd_scatter_or_pass0" in name or "msd_buckets_or_pass1" in name:
            # the MSD plan's kernels: 4-byte loads; the scatter stores quads (16 B), the bucket kernel words
            # scatter_msd_kernel<BITS, KV> | bucket_sort2_kernel<BITS, KPT, KV> | msd_*_or_pass*_kernel<BITS, KV, DYN> (keys-only)
            targs = [t.strip() for t in name.split("<", 1)[1].split(">", 1)[0].split(",")]
            kv_arg = targs[2] if "bucket_sort2" in name else targs[1]
            mode = "key_value" if kv_arg == "true" else "keys"
            part = "scatter" if "scatter" in name else "bucket"
            rd = fetch[name] * f4
            wr = write.get(name, 0.0) * (w16 if part == "scatter" else w4)
            kind = "msd_%s_%s" % (part, mode)
            out["plan_kernels"]["%s:%s" % (mode, part)] = rd + wr
        elif "histogram" in name:
            rd, wr = fetch[name] * f16, write.get(name, 0.0) * w4
            kind = "histogram"
            if "histogram_msd" in name:
                out["plan_kernels"]["keys:histogram"] = out["plan_kernels"]["key_value:histogram"] = rd + wr
        else:
            continue
        out["kernels"][name] = {"kind": kind, "launches": calls[name], "hbm_read_bytes_per_launch": rd,
                                "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
        if kind == "keys":
            out["onesweep_keys_bytes_per_launch"] = rd + wr
            out["onesweep_keys_algorithmic_bytes_per_launch"] = 8.0 * n
        if kind == "key_value":
            out["onesweep_key_value_bytes_per_launch"] = rd + wr
            out["onesweep_key_value_algorithmic_bytes_per_launch"] = 16.0 * n
    text = json.dumps(out, indent=1)
    print(text)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(text + "\n")


if __name__ == "__main__":
    main()
