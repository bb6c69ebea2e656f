set -u
timeout -k 10 300 python3 -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "pool or undo or dropout" --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
for e in "" MC_POOL_NO_XCD_ORDER=1; do
  echo "== $e"
  env $e timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --kernel-table gpurun_out/kt.json > gpurun_out/b.json 2>/dev/null || exit 1
  python3 - <<'PY'
import json
b=json.load(open('gpurun_out/b.json')); print(round(b['value']), b['ms_per_step'])
for r in json.load(open('gpurun_out/kt.json')):
    if r['kernel'].startswith('pool'): print("  %-50s %6.1f us  %.3f"%(r['kernel'], r['avg_ms']*1e3, r.get('frac_of_peak',0)))
PY
done
