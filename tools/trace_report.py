#!/usr/bin/env python3
"""Summarises a VRDX_TRACE dump (tools/trace.sh): per pass, per phase durations in microseconds."""
import struct
import sys

import numpy as np

PHASES = ["ticket", "load+rank", "hist/scan", "regroup", "look-back", "scatter"]


def main(path):
    raw = open(path, "rb").read()
    tiles = struct.unpack_from("<I", raw, 0)[0]
    data = np.frombuffer(raw, dtype=np.uint64, offset=4).reshape(4, tiles, 8)
    for p in range(4):
        t = data[p, :, :7].astype(np.int64)
        t0 = t[:, 0].min()
        rel = (t - t0) / 100.0  # 100 MHz wall clock -> microseconds
        dur = np.diff(rel, axis=1)
        total = rel[:, 6] - rel[:, 0]
        print(f"pass {p}: tiles {tiles}  span {rel[:, 6].max():.1f} us  tile life mean {total.mean():.2f} "
              f"p50 {np.median(total):.2f} p95 {np.percentile(total, 95):.2f} us")
        for i, name in enumerate(PHASES):
            c = dur[:, i]
            print(f"    {name:10s} mean {c.mean():7.2f}  p50 {np.median(c):7.2f}  p95 {np.percentile(c, 95):7.2f}  "
                  f"max {c.max():7.2f}  sum/span {c.sum() / rel[:, 6].max():7.1f}")
        # concurrency and start profile
        starts = np.sort(rel[:, 0])
        print("    tile start times (us) at 10% steps:", " ".join(f"{starts[int(q * (tiles - 1))]:.1f}" for q in np.linspace(0, 1, 11)))
        extra = data[p, :, 7]
        trips = ((extra >> np.uint64(24 + 16)) & np.uint64(0xFFFF)).astype(np.int64)
        rows = ((extra >> np.uint64(24)) & np.uint64(0xFFFF)).astype(np.int64)
        print(f"    look-back trips mean {trips.mean():.2f} p50 {np.median(trips):.0f} p95 {np.percentile(trips, 95):.0f} max {trips.max()};"
              f" rows walked (digit 0) mean {rows.mean():.1f} p50 {np.median(rows):.0f} p95 {np.percentile(rows, 95):.0f}")
        order = np.argsort(rel[:, 0])
        first = order[: min(600, tiles)]
        lb = dur[:, 4]
        print(f"    look-back of the first {len(first)} started tiles: mean {lb[first].mean():.2f} max {lb[first].max():.2f};"
              f" of the rest: mean {np.delete(lb, first).mean() if tiles > len(first) else 0:.2f}")


def timeline(path, bin_us=2.0):
    """Lock-step evidence: for every pass, how many tiles are in which phase in each time bin.  The
    phases that move HBM bytes are load+rank (loads in flight) and scatter (stores issued, draining into
    the next bins); with one workgroup per CU at most 256 tiles exist at a time."""
    raw = np.fromfile(path, dtype=np.uint32, count=1)
    tiles = int(raw[0])
    data = np.fromfile(path, dtype=np.uint64, offset=4).reshape(4, tiles, 8)
    print("\n# tiles per phase in %.0f us bins (columns: t_us ticket load+rank hist/scan regroup look-back scatter | in flight)" % bin_us)
    for p in range(4):
        t = data[p, :, :7].astype(np.int64)
        rel = (t - t[:, 0].min()) / 100.0
        end = rel[:, 6].max()
        print(f"pass {p}:")
        idle = 0
        edges = np.arange(0.0, end + bin_us, bin_us)
        for lo in edges[:-1]:
            mid = lo + bin_us / 2
            counts = [int(((rel[:, i] <= mid) & (mid < rel[:, i + 1])).sum()) for i in range(6)]
            memory = counts[1] + counts[5]
            if memory < 64:
                idle += 1
            print("  %6.1f  %s | %d" % (lo, " ".join("%4d" % c for c in counts), sum(counts)))
        print(f"  bins with fewer than 64 of 256 CUs in a memory phase (load+rank or scatter): {idle} of {len(edges) - 1}")


if __name__ == "__main__":
    main(sys.argv[1])
    if len(sys.argv) > 2 and sys.argv[2] == "timeline":
        timeline(sys.argv[1])
