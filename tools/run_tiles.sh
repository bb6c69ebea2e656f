#!/bin/bash
# sweep: tools/run_tiles.sh <what> <tiles...> ; for each lib in LIBS (default: product lib)
what=$1; shift
for lib in ${LIBS:-default}; do
  for tile in "$@"; do
    if [ "$lib" = default ]; then unset MC_KERNELS_LIB; else export MC_KERNELS_LIB=build/variants/libmck_$lib.so; fi
    if [ "$tile" = auto ]; then unset MC_IGEMM_TILE; else export MC_IGEMM_TILE=$tile; fi
    timeout -k 10 120 python3 tools/igemm_sweep.py $what 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
