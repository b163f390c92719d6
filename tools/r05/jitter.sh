#!/bin/bash
# Where do the slow medians of the 128-point sweep come from (1.57 M, 4.98 M keys-only in r05_bench_driver_hip.csv)?
# Every sample with its stage intervals, three processes per size.
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT; OUT=gpurun_out/r05_jitter; mkdir -p $OUT
for n in 1310720 1572864 1835008 4718592 4980736 5242880; do
  for proc in 1 2 3; do
    timeout 120 tests/native/vrdx_selftest jitter $n keys 2 >> $OUT/jitter_keys.txt 2>&1
  done
done
for n in 786432 1310720; do
  for proc in 1 2; do timeout 120 tests/native/vrdx_selftest jitter $n kv 2 >> $OUT/jitter_kv.txt 2>&1; done
done
python3 - <<'PY'
import re, collections, statistics
for f in ("gpurun_out/r05_jitter/jitter_keys.txt", "gpurun_out/r05_jitter/jitter_kv.txt"):
    groups=collections.OrderedDict(); proc=0
    for line in open(f):
        if line.startswith('#'): proc+=1; print(line.strip()); continue
        m=re.match(r"(\d+) (\w+) round (\d+) run +(\d+) total +([\d.]+)", line)
        if not m or int(m.group(4))==0: continue
        groups.setdefault((m.group(1),proc,m.group(3)),[]).append(float(m.group(5)))
    for k,v in groups.items(): print(k, "median %.1f min %.1f max %.1f"%(statistics.median(v),min(v),max(v)), " ".join("%.0f"%x for x in v))
PY
