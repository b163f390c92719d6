#!/bin/bash
# (the round-5 tree: see ab_bench.sh)
# A/B on ONE box over the MSD plan's size range: `vrdx_selftest sweep lo hi points [kv]` (reference protocol per size) of the
# round-5 tree and of the working tree, alternating.
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/${TAG:-r06_ab_sweep}; mkdir -p $OUT; rm -f $OUT/*.txt
OLD=$ROOT/build/r05tree
LO=${1:-22.95}; HI=${2:-25.15}; PTS=${3:-16}
for rep in 1 2; do
  for which in old new; do
    bin=$([ $which = old ] && echo $OLD/tests/native/vrdx_selftest || echo $ROOT/tests/native/vrdx_selftest)
    for mode in "" kv; do
      timeout 600 $bin sweep $LO $HI $PTS $mode 2>&1 | grep "^[0-9]" | sed "s/^/$which /" >> $OUT/sweep.txt
    done
  done
done
python3 - $OUT/sweep.txt <<'PY'
import sys, collections
best = collections.defaultdict(lambda: 1e9)
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) < 5: continue
    which, n, mode, ms = f[0], int(f[1]), f[2], float(f[3])
    best[(n, mode, which)] = min(best[(n, mode, which)], ms)
ns = sorted({k[0] for k in best})
for mode in ("keys", "kv"):
    print(f"== {mode}: n, old gpu_ms, new gpu_ms (best of two), new/old")
    for n in ns:
        o, w = best.get((n, mode, "old")), best.get((n, mode, "new"))
        if o and w: print(f"{n:10d} {o:8.4f} {w:8.4f} {w / o:6.3f}")
PY
