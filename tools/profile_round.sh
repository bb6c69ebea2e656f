#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box: kernel-trace stats of bench.py, then PMC
# passes (each in a run of its own, as the MI355X guide prescribes).  usage: tools/profile_round.sh r01b
set -u
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o nin -- $B > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
S="python3 bench.py --steps 4 --warmup 1 --resident-minibatches 2 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o nin -- $S > $out/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $out/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o nin -- $S > $out/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $out/pmc_write.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $out/pmc_sq -o nin -- $S > $out/pmc_sq.log 2>&1 || { echo "pmc sq failed"; tail -5 $out/pmc_sq.log; exit 1; }
find $out -name '*.db' -o -name '*.csv' | head -20
