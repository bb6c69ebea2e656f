#!/bin/bash
# Round-3 evidence, part 1 (run on the GPU box): the GPU test suite, then UNCONTENDED per-kernel evidence -- kernel trace and
# counter passes of the NIN step with the weight-gradient branch off (MC_NO_STEP_FORK=1: one stream, so a kernel's duration is its
# own), and the event tables (every launch timed) of all four BASELINE nets.  usage: tools/profile_r03_base.sh <tag>
set -u
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
bad() { case $1 in 124|137|134|139) echo "GPU step died ($1): stopping"; exit 1;; esac; }
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1
  rc=$?; tail -5 $out/pytest.log; [ $rc -eq 0 ] || { echo "pytest failed rc=$rc"; grep -n "Error\|FAILED\|assert" $out/pytest.log | tail -30; exit 1; }
fi
export MC_NO_STEP_FORK=1
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_nofork -o nin -- $B --kernel-table $out/${tag}_nin_bs128_nofork_event_table.json > $out/trace_nofork.log 2>&1; rc=$?; bad $rc
[ $rc -eq 0 ] && python3 tools/rocprof_summary.py $(find $out/trace_nofork -name 'nin_results.db' | head -1) $out/${tag}_nin_bs128_nofork_kernel_stats.txt "MC_NO_STEP_FORK=1 rocprofv3 --kernel-trace --stats -- $B" > /dev/null
S="python3 bench.py --steps 4 --warmup 1 --resident-minibatches 2 --no-cpu-baseline"
pass() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" -d $out/pmc_$name -o nin -- $S > $out/pmc_$name.log 2>&1; local rc=$?; bad $rc
  if [ $rc -eq 0 ]; then python3 tools/pmc_summary.py $(find $out/pmc_$name -name 'nin_results.db' | head -1) mc --out $out/${tag}_nin_bs128_nofork_pmc_$name.txt; else echo "pmc pass $name failed rc=$rc"; tail -3 $out/pmc_$name.log; fi
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT
pass lds SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD
unset MC_NO_STEP_FORK
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --kernel-table $out/${tag}_nin_bs128_event_table.json > $out/${tag}_bench_n1.json 2> $out/bench.err; rc=$?; bad $rc
for net in 80sec 18pct conv-local imagenet; do
  timeout -k 10 300 python3 tools/net_bench.py $net 128 30 12 $out/${tag}_$(echo $net | tr -d '-')_bs128_event_table.json > $out/${tag}_$(echo $net | tr -d '-')_bs128_net_bench.txt 2>&1; rc=$?; bad $rc
  head -2 $out/${tag}_$(echo $net | tr -d '-')_bs128_net_bench.txt
done
cat $out/${tag}_bench_n1.json | cut -c1-600
ls $out/*.txt $out/*.json
