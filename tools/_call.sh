set -u
mkdir -p gpurun_out/c8
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "no_alloc or tile_choice" > gpurun_out/c8/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/c8/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|FAILED\|assert" gpurun_out/c8/pytest.log | tail -20; exit 1; }
MC_IGEMM_AUTOTUNE=2 timeout -k 10 200 python3 tools/layer_times.py nin wgrad > gpurun_out/c8/lt_mf.txt 2>&1 || exit 1
grep wgrad gpurun_out/c8/lt_mf.txt; grep "autotune" gpurun_out/c8/lt_mf.txt | grep "kc/kc" | cut -c1-420
timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/c8/bench.json 2>gpurun_out/c8/bench.err || exit 1
cut -c1-200 gpurun_out/c8/bench.json
