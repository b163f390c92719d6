#!/bin/bash
# The part of tools/profile_round.sh that depends on the final kernels: bench line, rocprofv3 stats of the same command,
# PMC traffic, native bench at the headline sizes.  Writes gpurun_out/r05f/.
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/r05f; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_stats -o stats -- python3 bench.py --no-cpu-baseline --no-sweep > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
S=$(find $OUT/kernel_stats -name '*kernel_stats.csv' | head -1)
cp "$S" $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
python3 tools/kernel_stats.py "$S" --json $OUT/kernel_stats.json > $OUT/kernel_stats.txt 2>&1
rm -rf $OUT/kernel_stats
(cd tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 pmc_calibrate.hip -I$ROOT/include -L$ROOT/vulkan_radix_sort_amd -lvrdx_hip \
    -Wl,-rpath,$ROOT/vulkan_radix_sort_amd -o /tmp/pmc_calibrate) 2> $OUT/pmc_build.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -o fetch -- /tmp/pmc_calibrate 25 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -o write -- /tmp/pmc_calibrate 25 > $OUT/pmc_write.log 2>&1
F=$(find /tmp/pmc_fetch -name '*counter_collection.csv' | head -1)
W=$(find /tmp/pmc_write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_report.py "$F" "$W" 33554432 $OUT/pmc_traffic.json > $OUT/pmc_report.log 2>&1
timeout 600 tests/native/vrdx_selftest bench 23 24 25 26 > $OUT/native_23_26.txt 2>&1
cat $OUT/kernel_stats.txt; tail -25 $OUT/pmc_report.log; cat $OUT/native_23_26.txt
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05f/bench.json"))
print("keys", d["value"], "kv", d["key_value"]["value"], d["setup"])
PY
