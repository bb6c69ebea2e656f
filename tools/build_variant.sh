#!/bin/bash
# builds a tuning variant of libmckernels.so: tools/build_variant.sh name -DMC_BK=32 ...
set -e
name=$1; shift
cd "$(dirname "$0")/../cuda-convnet_amd"
mkdir -p ../build/variants/$name
for f in igemm_fwd igemm_bwd igemm_wgrad igemm_gemm conv_band elementwise pool_norm image_ops api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../include -Icsrc/kernels -Wno-unused-function "$@" -c csrc/kernels/$f.hip -o ../build/variants/$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../build/variants/libmck_$name.so ../build/variants/$name/*.o
echo built build/variants/libmck_$name.so
