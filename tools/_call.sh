set -u
mkdir -p gpurun_out/c11
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "tile_choice or test_gemm or weight or bias or fused or no_alloc or nin" > gpurun_out/c11/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/c11/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|FAILED\|assert" gpurun_out/c11/pytest.log | tail -20; exit 1; }
timeout -k 10 200 python3 tools/layer_times.py nin wgrad 2>&1 | grep wgrad
timeout -k 10 300 python3 tools/net_bench.py imagenet 128 30 8 2>&1 | grep -v amdgpu
timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200
