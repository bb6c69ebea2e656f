set -u
mkdir -p gpurun_out/c10
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --deselect tests/test_kernels_gpu.py > gpurun_out/c10/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/c10/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|FAILED\|assert" gpurun_out/c10/pytest.log | tail -20; exit 1; }
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for net in 80sec 18pct conv-local imagenet; do timeout -k 10 200 python3 tools/net_bench.py $net 128 30 3 2>&1 | grep -v amdgpu | head -2; done
