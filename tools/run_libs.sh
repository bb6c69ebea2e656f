#!/bin/bash
# tools/run_libs.sh <script> <libs...>: runs a sweep script for the product library ("default") and variants
script=$1; shift
for lib in "$@"; do
  if [ "$lib" = default ]; then unset MC_KERNELS_LIB; else export MC_KERNELS_LIB=build/variants/libmck_$lib.so; fi
  echo "== $lib"
  timeout -k 10 200 python3 $script 2>&1 | grep -v amdgpu.ids || exit 1
done
