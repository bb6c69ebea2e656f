set -u
timeout -k 10 1000 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_full.log 2>&1; rc=$?
tail -4 gpurun_out/pytest_full.log
[ $rc -eq 0 ] || { grep -n "Error\|FAILED\|assert" gpurun_out/pytest_full.log | tail -20; exit 1; }
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | cut -c1-200
