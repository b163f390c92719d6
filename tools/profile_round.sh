#!/bin/bash
# Round-end measurement bundle, run ON the GPU box:  gpurun --timeout 1800 -- 'bash tools/profile_round.sh r01'
# Writes everything under gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
# 1. bench line
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
# 2. kernel trace + stats of the same command (without the CPU baseline, the size curve and the adversarial inputs: every
#    launch the profiler then sees is a sort of uniform keys of the headline size, so its per-kernel averages are comparable
#    with the line's)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_stats -o stats -- python3 bench.py --no-cpu-baseline --no-sweep --no-adversarial > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
S=$(find $OUT/kernel_stats -name '*kernel_stats.csv' | head -1)
cp "$S" $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
python3 tools/kernel_stats.py "$S" --json $OUT/kernel_stats.json > $OUT/kernel_stats.txt 2>&1
# kernel durations by pass index of ten sorts back to back, keys-only and key+value apart (bench.py's run mixes them)
for mode in keys kv; do
  rm -rf /tmp/pr_b2b
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pr_b2b -o t -- $ROOT/tests/native/vrdx_selftest backtoback 25 $mode 10) \
      2>&1 | grep "back to back" >> $OUT/pass_times_back_to_back.txt
  python3 tools/pass_parity.py "$(find /tmp/pr_b2b -name '*kernel_trace.csv' | head -1)" 1 >> $OUT/pass_times_back_to_back.txt 2>&1
done
# 3. PMC passes (separate runs, counters only + kernel trace)
(cd tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 pmc_calibrate.hip -I$ROOT/include -L$ROOT/vulkan_radix_sort_amd -lvrdx_hip \
    -Wl,-rpath,$ROOT/vulkan_radix_sort_amd -o /tmp/pmc_calibrate) 2> $OUT/pmc_build.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- /tmp/pmc_calibrate 25 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- /tmp/pmc_calibrate 25 > $OUT/pmc_write.log 2>&1
F=$(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/pmc_write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_report.py "$F" "$W" 33554432 $OUT/pmc_traffic.json > $OUT/pmc_report.log 2>&1
# 4. native sweep over N (the reference's curve: bench/bench.cc:17-20)
timeout 900 tests/native/vrdx_selftest bench 15 16 17 18 19 20 21 22 23 24 25 26 27 > $OUT/native_sweep.txt 2>&1
ls -R $OUT | head -40
# 5. the rest of the round's evidence: adversarial inputs at 2^25, a soak of overlapping sorts, the bench driver's sweep
#    (hip and rocprim backends), the smoke entry and the GPU test log
timeout 900 tests/native/vrdx_selftest adversarial 25 > $OUT/adversarial.txt 2>&1
# ... and the same inputs with the four passes alone (VRDX_MSD=0), on the same box: what a plan that is turned down costs
echo "# the same with VRDX_MSD=0 (the four passes alone)" >> $OUT/adversarial.txt
VRDX_MSD=0 timeout 900 tests/native/vrdx_selftest adversarial 25 >> $OUT/adversarial.txt 2>&1
timeout 300 tests/native/vrdx_selftest soak 120 > $OUT/soak.txt 2>&1
if [ "${WITH_DRIVER:-0}" = 1 ]; then  # (ten minutes each: the reference's sweep, 64 sizes x 11 runs x fresh mt19937 data)
  timeout 900 bench/bench hip --no-verify -o $OUT/bench_driver_hip.csv > $OUT/bench_driver_hip.log 2>&1
  timeout 900 bench/bench rocprim --no-verify -o $OUT/bench_driver_rocprim.csv > $OUT/bench_driver_rocprim.log 2>&1
  # record once / submit many: every sort captured into a hipGraph once per (N, mode), the replay timed (cpu_ms column)
  timeout 900 bench/bench hip --graph --no-verify -o $OUT/bench_driver_hip_graph.csv > $OUT/bench_driver_hip_graph.log 2>&1
fi
# 6. per-tile phase timelines of the headline kernels (the -DVRDX_TRACE build of `make -C vulkan_radix_sort_amd/csrc trace`)
bash tools/trace.sh 1024x32x2 25 keys uniform > /dev/null 2>&1
bash tools/trace.sh 1024x32 25 kv uniform > /dev/null 2>&1
cp gpurun_out/trace_1024x32x2_keys_uniform.txt gpurun_out/trace_1024x32_kv_uniform.txt $OUT/ 2>/dev/null
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
if [ "${WITH_TESTS:-0}" = 1 ]; then timeout 1800 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; fi
