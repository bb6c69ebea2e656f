set -u
timeout -k 10 300 python3 -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "band" --timeout 120 --timeout-method thread 2>&1 | tail -15 || exit 1
MC_CONV_BAND_VERBOSE=1 timeout -k 10 200 python3 tools/net_bench.py 80sec 128 30 12 2>&1 | grep -v amdgpu.ids | head -12 || exit 1
MC_CONV_BAND=0 timeout -k 10 200 python3 tools/net_bench.py 80sec 128 30 12 2>&1 | grep -v amdgpu.ids | head -3 || exit 1
MC_CONV_BAND_VERBOSE=1 timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | cut -c1-300
