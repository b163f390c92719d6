#!/bin/bash
# (the round-5 tree: `git worktree add build/r05tree a7fc09d`, then `make -C vulkan_radix_sort_amd/csrc && make -C oracle &&
#  make -C tests/native vrdx_selftest` inside it, here, before the gpurun call: build/ travels with the snapshot)
# A/B on ONE box in the reference protocol (`vrdx_selftest bench`: 1 warm-up + 10 timed sorts of fresh data, median GPU time):
# the round-5 tree (git worktree build/r05tree, built there) against the working tree, alternating, three times.
# usage: ab_bench.sh [log2n ...]
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/${TAG:-r06_ab_bench}; mkdir -p $OUT; rm -f $OUT/bench.txt
OLD=$ROOT/build/r05tree
SIZES=${@:-23 24 25 26}
for rep in 1 2 3; do
  for which in old new; do
    bin=$([ $which = old ] && echo $OLD/tests/native/vrdx_selftest || echo $ROOT/tests/native/vrdx_selftest)
    timeout 600 $bin bench $SIZES 2>&1 | grep -v "^vrdx-hip\|^n " | sed "s/^/$which /" >> $OUT/bench.txt
  done
done
sort -k2,2n -k3,3 -k1,1 $OUT/bench.txt | awk '{printf "%-4s %10s %-5s gpu_ms %s GItems/s %s stamped %s hist %s | %s %s %s %s\n", $1, $2, $3, $4, $6, $11, $12, $19, $20, $21, $22}'
