#!/bin/bash
# Round 6, VERDICT item 4: the key+value fused launches (msd_scatter_or_pass0_kernel<.., true, ..>) against the same sort with
# its four returning passes (VRDX_MSD_FUSED=0, the recorder of HEAD a5914c1 for key+value) and against the tree of a5914c1
# (`git worktree add build/headtree a5914c1`, built there before the gpurun call) for the keys-only kernels, whose shape changed.
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/${TAG:-r06_kv_fused}; mkdir -p $OUT
OLD=$ROOT/build/headtree/tests/native/vrdx_selftest; NEW=$ROOT/tests/native/vrdx_selftest
export TMPDIR=/tmp
if [ -z "$SKIP_PARITY" ]; then
timeout 600 $NEW msd > $OUT/msd.txt 2>&1; echo "msd battery: exit $? $(tail -1 $OUT/msd.txt)"
timeout 600 $NEW quick > $OUT/quick.txt 2>&1; echo "quick: exit $? $(tail -1 $OUT/quick.txt)"
timeout 600 $NEW adversarial > $OUT/adversarial.txt 2>&1; echo "adversarial: exit $? $(tail -1 $OUT/adversarial.txt)"
fi
for rep in 1 2 3; do
  timeout 300 $OLD bench ${SIZES:-23 24 25 26} 2>&1 | grep -v "^vrdx-hip\|^n " | sed "s/^/head /" >> $OUT/bench.txt
  timeout 300 $NEW bench ${SIZES:-23 24 25 26} 2>&1 | grep -v "^vrdx-hip\|^n " | sed "s/^/new /" >> $OUT/bench.txt
  VRDX_MSD_FUSED=0 timeout 300 $NEW bench ${SIZES:-23 24 25 26} 2>&1 | grep -v "^vrdx-hip\|^n " | sed "s/^/unfused /" >> $OUT/bench.txt
done
sort -k2,2n -k3,3 -k1,1 $OUT/bench.txt | awk '{print $1, $2, $3, $4, $6}'
for which in ${B2B:-head new}; do
  bin=$([ $which = head ] && echo $OLD || echo $NEW)
  for mode in ${MODES:-keys kv}; do
    rm -rf /tmp/ab_prof
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -o t -- $bin backtoback ${AB_LG:-25} $mode 10) > $OUT/b2b_${which}_$mode.log 2>&1
    S=$(find /tmp/ab_prof -name '*kernel_stats.csv' | head -1); cp $S $OUT/b2b_${which}_${mode}_kernel_stats.csv
    echo "== $which $mode: $(grep 'back to back' $OUT/b2b_${which}_$mode.log)"
    python3 - "$S" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"]).replace("void vrdx::", "")
    print(f"   {name[:56]:56s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
  done
done
