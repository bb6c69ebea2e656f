#!/bin/bash
# The counter part of tools/profile_round.sh for BOTH tiles the autotuner may pick for NIN's conv2 forward (96x128 and 192x128
# time within a few per cent of each other): for each, three PMC passes of bench.py with the tile forced, a kernel trace, a bench
# line, and the entry merged into pmc_traffic.json.  usage: tools/profile_pmc.sh r02   (results under gpurun_out/prof_<tag>/)
set -u
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rm -f $out/pmc_traffic.json
for tile in 3,1 3,2; do
  t=$(echo $tile | tr -d ,)
  export MC_IGEMM_TILE=$tile
  S="python3 bench.py --steps 4 --warmup 1 --resident-minibatches 2 --no-cpu-baseline"
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch_$t -o nin -- $S > $out/pmc_fetch_$t.log 2>&1 || { echo "pmc fetch failed"; tail -5 $out/pmc_fetch_$t.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write_$t -o nin -- $S > $out/pmc_write_$t.log 2>&1 || { echo "pmc write failed"; tail -5 $out/pmc_write_$t.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $out/pmc_sq_$t -o nin -- $S > $out/pmc_sq_$t.log 2>&1 || { echo "pmc sq failed"; tail -5 $out/pmc_sq_$t.log; exit 1; }
  for p in fetch write sq; do
    python3 tools/pmc_summary.py $(find $out/pmc_${p}_$t -name 'nin_results.db' | head -1) mc_ --out $out/${tag}_nin_bs128_tile${t}_pmc_$p.txt || exit 1
  done
  B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_$t -o nin -- $B > $out/trace_$t.log 2>&1 || { echo "trace failed"; tail -5 $out/trace_$t.log; exit 1; }
  python3 tools/rocprof_summary.py $(find $out/trace_$t -name 'nin_results.db' | head -1) $out/${tag}_nin_bs128_tile${t}_kernel_stats.txt "MC_IGEMM_TILE=$tile rocprofv3 --kernel-trace --stats -- $B" > /dev/null || exit 1
  timeout -k 10 300 $B > $out/bench_tile$t.json 2> $out/bench_tile$t.err || { echo "bench failed"; tail -5 $out/bench_tile$t.err; exit 1; }
  python3 tools/make_pmc_traffic.py $out/${tag}_nin_bs128_tile${t}_pmc_fetch.txt $out/${tag}_nin_bs128_tile${t}_pmc_write.txt $out/${tag}_nin_bs128_tile${t}_pmc_sq.txt $out/bench_tile$t.json $out/${tag}_nin_bs128_tile${t}_kernel_stats.txt $out/pmc_traffic.json > $out/pmc_traffic.new || exit 1
  mv $out/pmc_traffic.new $out/pmc_traffic.json
done
unset MC_IGEMM_TILE
grep -o '"launch": "[^"]*"' $out/pmc_traffic.json; grep -o '"hbm_bytes_per_launch": [0-9]*' $out/pmc_traffic.json
