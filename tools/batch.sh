cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round.sh r03 > /dev/null 2>&1
OUT=gpurun_out/r03
timeout 900 tests/native/vrdx_selftest bench 26 27 > $OUT/native_sweep_large.txt 2>&1
timeout 900 tests/native/vrdx_selftest adversarial 25 > $OUT/adversarial.txt 2>&1
timeout 600 tests/native/vrdx_selftest soak 120 > $OUT/soak.txt 2>&1
timeout 900 bench/bench hip --points 16 -o $OUT/bench_driver_hip.csv > $OUT/bench_driver_hip.log 2>&1
timeout 900 bench/bench rocprim --points 16 -o $OUT/bench_driver_rocprim.csv > $OUT/bench_driver_rocprim.log 2>&1
VRDX_RANK=ballot timeout 600 tests/native/vrdx_selftest bench 20 23 25 > $OUT/ballot_bench.txt 2>&1
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log; cat $OUT/bench.json | head -c 1500; echo; tail -4 $OUT/soak.txt; tail -3 $OUT/bench_driver_hip.log
