set -u
timeout -k 10 600 python3 -u -m pytest tests/test_kernels_gpu.py tests/test_runtime_gpu.py -m gpu -x -q -k "pool or relu_fused or whole_step or one_step or captured or forked" --timeout 300 --timeout-method thread 2>&1 | tail -8 || exit 1
for e in "" MC_NO_FUSE_POOL_RELU=1; do
  for net in 80sec 18pct conv-local; do env $e timeout -k 10 200 python3 tools/net_bench.py $net 128 30 12 2>&1 | grep "images/s"; done
done
