set -u
MC_KERNELS_LIB=tools/variants/libmck_tr2.so timeout -k 10 300 python3 -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv_triplet or gemm or tile_choice" --timeout 200 --timeout-method thread 2>&1 | tail -4
for rep in 1 2; do
for lib in "" tools/variants/libmck_tr2.so; do
  echo "== ${lib:-default} rep $rep"
  for o in f b; do
    MC_KERNELS_LIB=$lib MC_CONV_BAND=0 timeout -k 10 200 python3 tools/tile_sweep.py $o 2>&1 | grep -v "amdgpu.ids\|cccp5\|cccp6" || exit 1
  done
done
done
