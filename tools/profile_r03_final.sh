#!/bin/bash
# Round-3 evidence (run on the GPU box after tools/profile_r03_base.sh <tag>, same tag): the conv2-forward counter passes with each
# of the two tiles the autotuner may pick (tools/profile_pmc.sh; tools/make_pmc_kernels.py then adds the "kernels" section from
# part 1's single-stream passes, off the box, where both parts' files are), the kernel trace of the forked NIN step and of the 18pct step (its rnorm kernels), and bench.py read back with
# the new pmc_traffic.json in place.  usage: tools/profile_r03_final.sh <tag>
set -u
tag=${1:-r03}
out=gpurun_out/prof_$tag
export TMPDIR=/tmp
bad() { case $1 in 124|137|134|139) echo "GPU step died ($1): stopping"; exit 1;; esac; }
mkdir -p $out
if [ "${SKIP_PMC:-0}" != 1 ]; then
  bash tools/profile_pmc.sh $tag || exit 1
  cp $out/pmc_traffic.json $out/pmc_traffic_launches.json
fi
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_fork -o nin -- $B > $out/trace_fork.log 2>&1; rc=$?; bad $rc
[ $rc -eq 0 ] && python3 tools/rocprof_summary.py $(find $out/trace_fork -name 'nin_results.db' | head -1) $out/${tag}_nin_bs128_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- $B" > /dev/null
N="python3 tools/net_bench.py 18pct 128 30 12"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_18pct -o net -- $N > $out/trace_18pct.log 2>&1; rc=$?; bad $rc
[ $rc -eq 0 ] && python3 tools/rocprof_summary.py $(find $out/trace_18pct -name 'net_results.db' | head -1) $out/${tag}_18pct_bs128_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- $N" > /dev/null
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --kernel-table $out/${tag}_nin_bs128_event_table.json > $out/${tag}_bench_n1.json 2> $out/bench.err; rc=$?; bad $rc
cut -c1-2500 $out/${tag}_bench_n1.json
ls $out/*.txt $out/*.json
