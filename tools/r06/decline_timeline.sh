#!/bin/bash
# Kernel-by-kernel timeline of the adversarial inputs at 2^25 (the LAST of the eleven sorts of every pattern x mode: no timestamps), with the
# MSD plan recorded in front of the passes and with VRDX_MSD=0: what does a plan that the device turns down cost, and where?
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/${TAG:-r06_decline}; mkdir -p $OUT
export TMPDIR=/tmp
for msd in 1 0; do
rm -rf /tmp/pdt
(cd /tmp && VRDX_MSD=$msd VRDX_SELFTEST_PATTERNS=${PATTERNS:-6} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pdt -o t -- $ROOT/tests/native/vrdx_selftest adversarial 25 > $OUT/adversarial_msd$msd.txt 2>&1)
python3 - "$(find /tmp/pdt -name '*kernel_trace.csv' | head -1)" > $OUT/timeline_msd$msd.txt <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void vrdx::", "")
    seq.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size", ""), r.get("LDS_Block_Size", "")))
# a sort starts with its histogram kernel (the fill in front of it, when there is one, is printed with it)
starts = [i for i, s in enumerate(seq) if s[0].startswith("histogram")]
sorts = []
for k, si in enumerate(starts):
    end = starts[k + 1] if k + 1 < len(starts) else len(seq)
    first = si - 1 if si > 0 and ("fill" in seq[si - 1][0].lower() or "prologue" in seq[si - 1][0]) else si
    body = [s for s in seq[first:end] if not ("fill" in s[0].lower() and s is not seq[first])]
    sorts.append(body)
# eleven sorts per (pattern, mode): six with the 15 timestamps, five without; print the last (unstamped) of each group
G = 11
for g in range(0, len(sorts), G):
    grp = sorts[g:g + G]
    if len(grp) < G: break
    body = grp[-1]
    t0 = body[0][1]; prev = None
    print(f"--- sort group {g // G} (pattern {g // (2 * G)}, {'kv' if (g // G) % 2 else 'keys'})")
    for n, s, e, grid, lds in body:
        gap = (s - prev) / 1e3 if prev else 0.0
        print(f"{n[:64]:64s} start {(s - t0)/1e3:8.2f} dur {(e - s)/1e3:7.2f} gap {gap:5.2f} grid {grid} lds {lds}")
        prev = e
    print("  total %.2f us" % ((prev - t0) / 1e3))
PY
done
cat $OUT/adversarial_msd1.txt $OUT/adversarial_msd0.txt
