#!/bin/bash
# LDS conflict counters of the histogram launch and every onesweep pass for one input pattern, run ON the GPU box:
#   gpurun -- 'bash tools/pmc_lds.sh ascending [keys|kv] [log2n]'
# (counters in their own rocprofv3 run, kernel trace only)
PATTERN=${1:-uniform}
KV=${2:-keys}
LOG2N=${3:-25}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
out=/tmp/pmc_lds_$PATTERN
rm -rf $out
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT" "SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o p -- \
      $ROOT/tests/native/vrdx_selftest trace $LOG2N $KV $PATTERN > $out.log 2>&1 || tail -5 $out.log
done
python3 - $out <<'EOF'
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
    print(k, d[k]["kernel"], {a: int(b) for a, b in sorted(d[k].items()) if a != "kernel"})
EOF
