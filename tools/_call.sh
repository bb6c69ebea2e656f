set -u
mkdir -p gpurun_out/c6
export MC_IGEMM_AUTOTUNE=0
for k in 65536 131072 262144; do
  MC_IGEMM_TILE=5,6 MC_GEMM_CHUNKS=256 timeout -k 10 100 python3 tools/mf_probe.py 160 192 $k || exit 1
  MC_IGEMM_TILE=5,6 MC_GEMM_CHUNKS=256 MC_IGEMM_KINTERLEAVE=1 timeout -k 10 100 python3 tools/mf_probe.py 160 192 $k || exit 1
done
MC_IGEMM_TILE=5,1 MC_GEMM_CHUNKS=128 timeout -k 10 100 python3 tools/mf_probe.py 160 192 131072 || exit 1
MC_IGEMM_TILE=5,1 MC_GEMM_CHUNKS=128 MC_IGEMM_KINTERLEAVE=1 timeout -k 10 100 python3 tools/mf_probe.py 160 192 131072 || exit 1
MC_IGEMM_TILE=3,6 MC_GEMM_CHUNKS=128 timeout -k 10 100 python3 tools/mf_probe.py 192 192 32768 || exit 1
MC_IGEMM_TILE=3,6 MC_GEMM_CHUNKS=128 MC_IGEMM_KINTERLEAVE=1 timeout -k 10 100 python3 tools/mf_probe.py 192 192 32768 || exit 1
MC_IGEMM_TILE=3,6 MC_GEMM_CHUNKS=128 timeout -k 10 100 python3 tools/mf_probe.py 192 192 65536 || exit 1
MC_IGEMM_TILE=6,6 MC_GEMM_CHUNKS=256 timeout -k 10 100 python3 tools/mf_probe.py 192 192 131072 || exit 1
MC_IGEMM_TILE=6,6 MC_GEMM_CHUNKS=256 timeout -k 10 100 python3 tools/mf_probe.py 192 192 262144 || exit 1
