#!/bin/bash
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT
OUT=gpurun_out/r05_bench; mkdir -p $OUT
S=$(date +%s); python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "default run: $(( $(date +%s) - S )) s"; tail -3 $OUT/bench.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05_bench/bench.json"))
print("keys", d["value"], "kv", d["key_value"]["value"], "ms", d["ms_per_step"], d["setup"])
for mode, ks in (("keys", d["roofline"]["kernels"]), ("kv", d["roofline"]["key_value"]["kernels"])):
    for part, e in ks.items(): print(mode, part, e["kernel"][:48], round(e["avg_launch_ms"]*1e3,1), "us", round(e.get("frac",0),3))
print("dominant", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["whole_sort"])
for p in d.get("sweep", []): print(p["log2n"], round(p["keys"]["gitems_per_s"],1), p["keys"]["plan"], round(p["keys"]["hbm_fraction"],3), round(p["key_value"]["gitems_per_s"],1), p["key_value"]["plan"], round(p["key_value"]["hbm_fraction"],3))
print(d.get("hbm_resident")); print(d.get("sweep_protocol")); print(d.get("cpu_baseline"))
PY
S=$(date +%s); python3 bench.py --steps 20 --warmup 5 --no-sweep --no-cpu-baseline > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; echo "driver-args run: $(( $(date +%s) - S )) s"
python3 -c "
import json
d=json.load(open('gpurun_out/r05_bench/bench_driver_args.json')); print(d['value'], d['key_value']['value'], d['setup'])"
