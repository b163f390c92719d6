#!/bin/bash
# After a change to the stamped sources: GPU tests, native parity, then the stamped evidence (bench line, rocprofv3 stats, PMC
# traffic) and the bench line once more with that traffic in place (profiles/pmc_traffic.json of the box's copy).
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r05_refreeze; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 900 tests/native/vrdx_selftest parity > $OUT/native_parity.txt 2>&1; tail -1 $OUT/native_parity.txt
timeout 900 tests/native/vrdx_selftest msd > $OUT/msd_parity.txt 2>&1; tail -1 $OUT/msd_parity.txt
bash tools/r05/final_numbers.sh > $OUT/final_numbers.log 2>&1; tail -12 $OUT/final_numbers.log
cp gpurun_out/r05f/pmc_traffic.json profiles/pmc_traffic.json; cp gpurun_out/r05f/kernel_stats.json profiles/kernel_stats.json
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05_refreeze/bench.json")); r=d["roofline"]
print("keys", d["value"], "kv", d["key_value"]["value"], "dominant", r["kernel"], r["frac"], "traffic", r.get("traffic"), r.get("traffic_frac"))
PY
