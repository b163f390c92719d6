#!/bin/bash
# Histogram kernel duration against its grid size, run ON the GPU box:  gpurun -- 'bash tools/hist_grid.sh 23 "64 128 256 512"'
LOG2N=${1:-23}
GRIDS=${2:-"64 128 256 512"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for g in $GRIDS; do
  export VRDX_HIST_GRID=$g
  rm -rf /tmp/hg
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hg -o s -- $ROOT/tests/native/vrdx_selftest trace $LOG2N keys > /dev/null 2>&1
  f=$(find /tmp/hg -name "*kernel_stats.csv" | head -1)
  echo "n=2^$LOG2N grid=$g $(grep histogram $f | awk -F, '{print "hist avg ns", $(NF-4), "min", $(NF-2)}')"
done
