cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in 1 2 3; do
echo "=== pass0 config $c keys"; VRDX_PASS0_CONFIG=$c timeout 300 tests/native/vrdx_selftest bench 25 2>&1 | grep "^3355"
done
echo "=== parity pass0=1"; VRDX_PASS0_CONFIG=1 timeout 600 tests/native/vrdx_selftest quick 2>&1 | tail -2
