#!/bin/bash
# runs tools/igemm_sweep.py <what> for the default library and each build/variants/libmck_*.so given
what=$1; shift
timeout -k 10 120 python3 tools/igemm_sweep.py $what || exit 1
for v in "$@"; do
  MC_KERNELS_LIB=build/variants/libmck_$v.so timeout -k 10 120 python3 tools/igemm_sweep.py $what || exit 1
done
