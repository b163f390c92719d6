#!/bin/bash
# (the round-5 tree: `git worktree add build/r05tree a7fc09d`, then `make -C vulkan_radix_sort_amd/csrc && make -C oracle &&
#  make -C tests/native vrdx_selftest` inside it, here, before the gpurun call: build/ travels with the snapshot)
# Per-kernel averages (rocprofv3, ten sorts back to back at 2^25) of the round-5 tree, the working tree and its compile-time
# variants under build/variants/<name>/ (tools/build_variants.sh), all on ONE box.
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/${TAG:-r06_abv}; mkdir -p $OUT
export TMPDIR=/tmp
MODES=${MODES:-keys kv}
for which in old new "$@"; do
  bin=$ROOT/tests/native/vrdx_selftest; lib=
  [ $which = old ] && bin=$ROOT/build/r05tree/tests/native/vrdx_selftest
  [ $which != old ] && [ $which != new ] && lib=$ROOT/build/variants/$which
  for mode in $MODES; do
    rm -rf /tmp/ab_prof
    (cd /tmp && LD_LIBRARY_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -o t -- $bin backtoback ${AB_LG:-25} $mode 10) > $OUT/b2b_${which}_$mode.log 2>&1
    S=$(find /tmp/ab_prof -name '*kernel_stats.csv' | head -1)
    echo "== $which $mode: $(grep 'back to back' $OUT/b2b_${which}_$mode.log)"
    python3 - "$S" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"]).replace("void vrdx::", "")
    if "order_check" in name or "copyBuffer" in name: continue
    print(f"   {name[:56]:56s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
  done
done
