#!/bin/bash
# (the round-5 tree: `git worktree add build/r05tree a7fc09d`, then `make -C vulkan_radix_sort_amd/csrc && make -C oracle &&
#  make -C tests/native vrdx_selftest` inside it, here, before the gpurun call: build/ travels with the snapshot)
# A/B on ONE box: the round-5 tree (git worktree build/r05tree, built there) against the working tree.
# usage: ab.sh [log2n ...]   -> gpurun_out/$TAG/{old,new}_*.txt
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/${TAG:-r06_ab}; mkdir -p $OUT
OLD=$ROOT/build/r05tree
export TMPDIR=/tmp
SIZES=${@:-25}
for rep in 1 2 3; do
  for which in old new; do
    bin=$([ $which = old ] && echo $OLD/tests/native/vrdx_selftest || echo $ROOT/tests/native/vrdx_selftest)
    timeout 300 $bin bench $SIZES 2>&1 | grep -v "^vrdx-hip\|^n " | sed "s/^/$which /" >> $OUT/bench.txt
  done
done
sort -k2,2n -k3,3 -k1,1 $OUT/bench.txt | awk '{print $1, $2, $3, $4, $6}' | column -t
# per-kernel averages of ten sorts back to back
for which in old new; do
  bin=$([ $which = old ] && echo $OLD/tests/native/vrdx_selftest || echo $ROOT/tests/native/vrdx_selftest)
  for mode in keys kv; do
    rm -rf /tmp/ab_prof
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -o t -- $bin backtoback ${AB_LG:-25} $mode 10) > $OUT/b2b_${which}_$mode.log 2>&1
    S=$(find /tmp/ab_prof -name '*kernel_stats.csv' | head -1)
    echo "== $which $mode: $(grep 'back to back' $OUT/b2b_${which}_$mode.log)"
    python3 - "$S" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"]).replace("void vrdx::", "")
    print(f"   {name[:56]:56s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
  done
done
