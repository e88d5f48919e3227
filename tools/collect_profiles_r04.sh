#!/bin/bash
# Round-end measurement run (one gpurun call, ~10 min of box time): everything profiles/r04_* is derived from, written under
# gpurun_out/r04/.  tools/update_profiles_r04.py turns it into the committed summaries.  PMC passes run `bench.py --no-graph` under
# `timeout` (counter collection on hipGraph replays hung in earlier rounds) and never together with a trace domain other than kernel-trace.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { echo "== $*" >> $O/log.txt; "$@" >> $O/log.txt 2>&1 < /dev/null; }
python3 $R/bench.py > $O/bench_cfg2.out 2>> $O/log.txt < /dev/null; tail -1 $O/bench_cfg2.out > $O/bench_cfg2_result.json
for c in 3 4 5; do timeout 600 python3 $R/bench.py --config $c > $O/bench_cfg$c.out 2>> $O/log.txt < /dev/null; tail -1 $O/bench_cfg$c.out > $O/bench_cfg${c}_result.json; done
python3 $R/bench.py --no-cpu-baseline --ids zipf 2>> $O/log.txt < /dev/null | tail -1 > $O/bench_cfg2_zipf_result.json
python3 $R/tools/step_table.py > $O/step_table_cfg2.txt 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg2 -- python3 $R/bench.py --no-cpu-baseline > $O/bench_cfg2_under_profiler.out 2>&1 < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg3 -- python3 $R/bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3_under_profiler.out 2>&1 < /dev/null
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_cfg2_$ctr -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-graph > /dev/null 2>&1 < /dev/null
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_cfg3_$ctr -- python3 $R/bench.py --config 3 --steps 12 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_cfg2_mfma -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-graph > /dev/null 2>&1 < /dev/null
for c in 3 4 5; do timeout 300 python3 $R/tools/host_time_supernet.py $c >> $O/host_time_supernet.txt 2>> $O/log.txt < /dev/null; done
timeout 300 python3 $R/tools/bench_supernet.py --strategy full-path --steps 10 --warmup 3 > $O/supernet_fullpath_step.txt 2>> $O/log.txt < /dev/null
timeout 300 python3 $R/tools/parser_bench.py > $O/parser_bench.txt 2>> $O/log.txt < /dev/null
timeout 900 python3 $R/tools/e2e_tsv_run.py --rows 3000000 > $O/e2e_tsv_run.txt 2>> $O/log.txt < /dev/null
timeout 300 python3 $R/tools/dedup_cost.py > $O/dedup_cost.txt 2>> $O/log.txt < /dev/null
# round 4: the data-parallel exchange step on one rank (whole step captured as one graph), what its collectives cost, row-sharded tables
python3 $R/bench.py --force-dp-path --no-cpu-baseline 2>> $O/log.txt < /dev/null | tail -1 > $O/bench_cfg2_dp_path_result.json
python3 $R/bench.py --force-dp-path --no-cpu-baseline --no-graph 2>> $O/log.txt < /dev/null | tail -1 > $O/bench_cfg2_dp_path_eager_result.json
(cd $R && bash tools/dp_overhead.sh > /dev/null 2>&1; cp gpurun_out/dp_overhead/result.txt $O/dp_overhead.txt)
python3 $R/bench.py --table-sharding row --no-cpu-baseline 2>> $O/log.txt < /dev/null | tail -1 > $O/bench_cfg2_row_sharded_result.json
(cd $R && bash tools/r04_ab.sh balance_final NASREC_WL_BALANCE=0 > /dev/null 2>&1; cp gpurun_out/ab/balance_final.txt $O/ab_level_balance.txt)
CONFIG=3 TOP=60 timeout 300 python3 $R/tools/supernet_step_table.py > $O/supernet_step_table_cfg3.txt 2>> $O/log.txt < /dev/null
# the throughput GEMM: per-k-tile cycle budget (needs `tools/build_variant.sh ftstamps -DFT_STAMPS` of the same sources) and the vendor library on five products
[ -f $R/nasrec_amd/lib/variants/ftstamps.so ] && NASREC_HIP_LIB=$R/nasrec_amd/lib/variants/ftstamps.so timeout 300 python3 $R/tools/gemm_fast_stamps.py > $O/gemm_fast_stamps.txt 2>> $O/log.txt < /dev/null
timeout 300 python3 $R/tools/gemm_vs_vendor.py > $O/gemm_vs_vendor.txt 2>> $O/log.txt < /dev/null
bash $R/tools/mha_pmc.sh $O/mha_pmc > /dev/null 2>&1
# keep what travels back small: per-dispatch traces are summarised on the box
python3 $R/tools/update_profiles_r04.py --summarise $O >> $O/log.txt 2>&1 < /dev/null
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
du -sh $O | tail -1
