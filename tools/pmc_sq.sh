#!/bin/bash
# Wave-state counters (what the waves of a pass wait for) of the onesweep launches, run ON the GPU box:
#   gpurun -- 'bash tools/pmc_sq.sh [keys|kv] [log2n]'
# One rocprofv3 run per counter pair (counters only + kernel trace); prints per-launch sums of the last sort.
KV=${1:-keys}
LOG2N=${2:-25}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
out=/tmp/pmc_sq_$KV
rm -rf $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "GRBM_GUI_ACTIVE SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o p -- \
      $ROOT/tests/native/vrdx_selftest trace $LOG2N $KV uniform > $out.log 2>&1 || { echo "counter set '$set' failed:"; tail -3 $out.log; }
done
python3 - $out <<'PY'
import csv, sys, glob, collections
d = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "onesweep" not in name and "histogram" not in name:
            continue
        e = d.setdefault(int(r["Dispatch_Id"]), {"kernel": "hist" if "histogram" in name else "onesweep"})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(d)[-5:]:
    print(k, d[k]["kernel"])
    for a, b in sorted(d[k].items()):
        if a != "kernel":
            print("    %-28s %16d" % (a, int(b)))
PY
