set -u
mkdir -p gpurun_out/c9
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "dropout or test_pool or nin or masks or warm_up" > gpurun_out/c9/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/c9/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|FAILED\|assert" gpurun_out/c9/pytest.log | tail -20; exit 1; }
timeout -k 10 200 python3 tools/layer_times.py nin "pool" > gpurun_out/c9/lt.txt 2>&1 || exit 1
grep -v amdgpu gpurun_out/c9/lt.txt
timeout -k 10 200 python3 tools/layer_times.py nin "dropout" 2>&1 | grep -v amdgpu
timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/c9/bench.json 2>gpurun_out/c9/bench.err || exit 1
cut -c1-200 gpurun_out/c9/bench.json
