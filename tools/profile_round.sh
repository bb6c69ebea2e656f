#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box and writes the text summaries that get committed under profiles/:
# kernel-trace stats of bench.py (NIN) and of the other BASELINE nets, then the PMC passes of bench.py (each in a run of its
# own, as the MI355X guide prescribes), pmc_traffic.json for bench.py's roofline.traffic, and the in-run event table.
# usage: tools/profile_round.sh r02      (results under gpurun_out/prof_r02/)
set -u
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o nin -- $B > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
python3 tools/rocprof_summary.py $(find $out/trace -name 'nin_results.db' | head -1) $out/${tag}_nin_bs128_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- $B" || exit 1
for net in imagenet 80sec conv-local; do
  C="python3 tools/net_bench.py $net 128 20"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_$net -o $net -- $C > $out/trace_$net.log 2>&1 || { echo "trace $net failed"; tail -5 $out/trace_$net.log; exit 1; }
  python3 tools/rocprof_summary.py $(find $out/trace_$net -name "${net}_results.db" | head -1) $out/${tag}_$(echo $net | tr -d '-')_bs128_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- $C" || exit 1
  grep -A10 "images/s" $out/trace_$net.log > $out/${tag}_$(echo $net | tr -d '-')_bs128_net_bench.txt
done
S="python3 bench.py --steps 4 --warmup 1 --resident-minibatches 2 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o nin -- $S > $out/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $out/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o nin -- $S > $out/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $out/pmc_write.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $out/pmc_sq -o nin -- $S > $out/pmc_sq.log 2>&1 || { echo "pmc sq failed"; tail -5 $out/pmc_sq.log; exit 1; }
for p in fetch write sq; do   # (tools/profile_pmc.sh collects the same per candidate tile and writes the pmc_traffic.json that is committed)
  python3 tools/pmc_summary.py $(find $out/pmc_$p -name 'nin_results.db' | head -1) mc_ --out $out/${tag}_nin_bs128_pmc_$p.txt || exit 1
done
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --kernel-table $out/${tag}_nin_bs128_event_table.json > $out/${tag}_bench_n1.json 2> $out/bench.err || { echo "bench failed"; tail -5 $out/bench.err; exit 1; }
python3 tools/make_pmc_traffic.py $out/${tag}_nin_bs128_pmc_fetch.txt $out/${tag}_nin_bs128_pmc_write.txt $out/${tag}_nin_bs128_pmc_sq.txt $out/${tag}_bench_n1.json $out/${tag}_nin_bs128_kernel_stats.txt > $out/pmc_traffic.json || exit 1
ls $out/*.txt $out/*.json
