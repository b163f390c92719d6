#!/bin/bash
# Instruction-cache and in-flight-level counters of the onesweep launches (keys-only, 2^25), run ON the GPU box:  gpurun -- "bash tools/pmc_icache.sh"
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp; out=/tmp/pmc_ic; rm -rf $out
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS" "SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL" "SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "SQ_INSTS_BRANCH SQ_INST_LEVEL_SMEM"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o p -- $ROOT/tests/native/vrdx_selftest trace 25 keys uniform > $out.log 2>&1 || { echo "set '$set' failed"; tail -2 $out.log; }
done
python3 - $out <<'PY'
import csv, sys, glob, collections
d = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "onesweep" not in name: continue
        e = d.setdefault(int(r["Dispatch_Id"]), {})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(d)[-2:]:
    print(k)
    for a, b in sorted(d[k].items()): print("    %-30s %16d" % (a, int(b)))
PY
