#!/bin/bash
# The reference's sweep twice with the driver's growing buffers and twice with the buffers reserved once for 2^25 pairs:
# are the slow sizes of r05_bench_driver_hip.csv (1.57 M, 4.98 M ...) the sort, or the mapping a fresh hipMalloc got?
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT; OUT=gpurun_out/r05_reserve; mkdir -p $OUT
for i in 1 2; do
  timeout 600 bench/bench hip --no-verify -o $OUT/grow_$i.csv > $OUT/grow_$i.log 2>&1
  VRDX_BENCH_RESERVE=33554432 timeout 600 bench/bench hip --no-verify -o $OUT/reserved_$i.csv > $OUT/reserved_$i.log 2>&1
done
python3 - <<'PY'
def load(path):
    rows={}
    for line in open(path):
        if line.startswith('#') or line.startswith('backend'): continue
        f=line.strip().split(',')
        rows[(int(f[1]), f[2])]=float(f[3])*1e3
    return rows
d={k:load("gpurun_out/r05_reserve/%s.csv"%k) for k in ("grow_1","grow_2","reserved_1","reserved_2")}
for s in ("keys","kv"):
    print(s, "us:   n      grow_1 grow_2 | reserved_1 reserved_2")
    for k in range(1,129):
        n=k<<18
        print(f"{n/1e6:6.2f}M " + " ".join(f"{d[x][(n,s)]:7.1f}" for x in d))
    for x in d:
        pts=[(n,n/t) for (n,m),t in sorted(d[x].items()) if m==s]
        steps=sorted(((b[1]-a[1])/a[1]*100,a[0]) for a,b in zip(pts,pts[1:]))
        print(x, s, "worst steps:", [(round(p,1), f"{n/1e6:.2f}M") for p,n in steps[:5]])
PY
