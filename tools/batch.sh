cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== parity"; timeout 600 tests/native/vrdx_selftest quick 2>&1 | tail -2
for c in 1024x16 1024x32 1024x32x2; do echo "== parity $c"; VRDX_TILE_CONFIG=$c timeout 600 tests/native/vrdx_selftest quick 2>&1 | tail -1; done
for v in base cur base cur; do
PROF=1 bash tools/run_variants.sh "$v" "auto" "sweep 25 25 1" b7 > /dev/null 2>&1
PROF=1 bash tools/run_variants.sh "$v" "auto" "sweep 25 25 1 kv" b7 > /dev/null 2>&1
done
PROF=1 bash tools/run_variants.sh "base cur" "auto" "sweep 24 24 1" b7 > /dev/null 2>&1
PROF=1 bash tools/run_variants.sh "base cur" "auto" "sweep 23 23 1 kv" b7 > /dev/null 2>&1
grep -v "^n \|order_check\|^vrdx" gpurun_out/b7.log
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/b7_pytest.log 2>&1; tail -5 gpurun_out/b7_pytest.log
