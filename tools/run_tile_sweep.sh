#!/bin/bash
for t in 1,1 2,1 3,1 5,1 2,3 3,2 3,3; do
  MC_IGEMM_TILE=$t timeout -k 10 200 python3 tools/tile_sweep.py 2>&1 | grep -v amdgpu.ids || exit 1
done
for c in 32 64 128 256; do for t in 2,3 3,3 3,1 5,1; do
  MC_GEMM_CHUNKS=$c MC_IGEMM_TILE=$t timeout -k 10 200 python3 tools/tile_sweep.py w 2>&1 | grep cccp || exit 1
done; done
