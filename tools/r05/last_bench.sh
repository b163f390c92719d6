#!/bin/bash
# The bench line once more with the final tree's PMC traffic committed (profiles/pmc_traffic.json carries the kernel-source
# digest bench.py checks), and the bench driver's graph-replay sweep on the same tree.
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT
OUT=gpurun_out/r05_last; mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05_last/bench.json"))
r=d["roofline"]
print("keys", d["value"], "kv", d["key_value"]["value"], "dominant", r["kernel"], r["frac"], "traffic", r.get("traffic"), r.get("traffic_frac"))
for mode, ks in (("keys", r["kernels"]), ("kv", r["key_value"]["kernels"])):
    for part, e in ks.items(): print(mode, part, e["kernel"][:48], round(e["avg_launch_ms"]*1e3,1), "us", round(e.get("frac",0),3), e.get("traffic"), e.get("traffic_frac"))
PY
timeout 900 bench/bench hip --graph --no-verify -o $OUT/bench_driver_hip_graph.csv > $OUT/bench_driver_hip_graph.log 2>&1; tail -2 $OUT/bench_driver_hip_graph.log
