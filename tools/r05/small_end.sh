#!/bin/bash
# Where do the 32 us of a 2^18 sort go?  Kernel start/end times of `bench 18` under rocprofv3 --kernel-trace.
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/r05_small_end; mkdir -p $OUT
export TMPDIR=/tmp
for lg in 18 20; do
rm -rf /tmp/pse
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/pse -o t -- $ROOT/tests/native/vrdx_selftest bench $lg > $OUT/bench_$lg.txt 2>&1)
python3 - "$(find /tmp/pse -name '*kernel_trace.csv' | head -1)" > $OUT/timeline_$lg.txt <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last keys-only sort without timestamps: find sequences fill..pass3; print the last 3 sorts
seq = []
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void vrdx::", "")
    seq.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
# sorts start with a fill
starts = [i for i, (n, s, e) in enumerate(seq) if "fillBuffer" in n]
for si in starts[-4:]:
    t0 = seq[si][1]
    prev_end = None
    for n, s, e in seq[si:si + 8]:
        if "fillBuffer" in n and prev_end is not None: break
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print(f"{n[:60]:60s} start {(s - t0)/1e3:7.2f} dur {(e - s)/1e3:6.2f} gap_before {gap:5.2f}")
        prev_end = e
    print("  total %.2f us" % ((prev_end - t0) / 1e3))
PY
tail -32 $OUT/timeline_$lg.txt
done
