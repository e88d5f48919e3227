#!/bin/bash
# Round-end measurement run (one gpurun call, ~15 min of box time): everything profiles/<tag>_* is derived from, written under
# gpurun_out/<tag>/ (tag = $1 or NASREC_PROFILE_TAG, default r06).  tools/update_profiles.py turns it into the committed summaries.  PMC passes run `bench.py --no-graph` (cfg 2; the
# supernet configs never capture) with `--steps-only` (the counter window holds real steps only: no repeated timing launches of the live
# roofline measurement) under `timeout`, and never together with a trace domain other than kernel-trace.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-${NASREC_PROFILE_TAG:-r06}}
export NASREC_PROFILE_TAG=$TAG
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_cfg2.out 2>> $O/log.txt < /dev/null; tail -1 $O/bench_cfg2.out > $O/bench_cfg2_result.json
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $O/log.txt < /dev/null | tail -1 > $O/bench_cfg2_driver_flags_result.json
for c in 3 4 5; do timeout 600 python3 $R/bench.py --config $c > $O/bench_cfg$c.out 2>> $O/log.txt < /dev/null; tail -1 $O/bench_cfg$c.out > $O/bench_cfg${c}_result.json; done
python3 $R/bench.py --no-cpu-baseline --ids zipf 2>> $O/log.txt < /dev/null | tail -1 > $O/bench_cfg2_zipf_result.json
python3 $R/tools/step_table.py > $O/step_table_cfg2.txt 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg2 -- python3 $R/bench.py --no-cpu-baseline > $O/bench_cfg2_under_profiler.out 2>&1 < /dev/null
for c in 3 4 5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg$c -- python3 $R/bench.py --config $c --no-cpu-baseline > $O/bench_cfg${c}_under_profiler.out 2>&1 < /dev/null
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_cfg2_$ctr -- python3 $R/bench.py --steps 12 --warmup 3 --steps-only --no-graph > /dev/null 2>&1 < /dev/null
  for c in 3 4 5; do
    timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_cfg${c}_$ctr -- python3 $R/bench.py --config $c --steps 12 --warmup 3 --steps-only > /dev/null 2>&1 < /dev/null
  done
done
# the largest GEMM launch's own rows: windows that also hold the live roofline timing (cfg 2 and 3)
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmcfull_cfg2_$ctr -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-graph > /dev/null 2>&1 < /dev/null
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmcfull_cfg3_$ctr -- python3 $R/bench.py --config 3 --steps 12 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_cfg2_mfma -- python3 $R/bench.py --steps 12 --warmup 3 --steps-only --no-graph > /dev/null 2>&1 < /dev/null
# keep what travels back small: per-dispatch traces are summarised on the box, right away (the rest of the run may be cut short)
python3 $R/tools/update_profiles.py --summarise $O >> $O/log.txt 2>&1 < /dev/null
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete  # (gpurun copies back at most 64 MiB; the summaries hold what is kept)
for c in 3 4 5; do timeout 300 python3 $R/tools/host_time_supernet.py $c >> $O/host_time_supernet.txt 2>> $O/log.txt < /dev/null; done
timeout 300 python3 $R/tools/bench_supernet.py --strategy full-path --steps 10 --warmup 3 > $O/supernet_fullpath_step.txt 2>> $O/log.txt < /dev/null
timeout 300 python3 $R/tools/dedup_cost.py > $O/dedup_cost.txt 2>> $O/log.txt < /dev/null
# the data-parallel exchange step on one rank (whole step captured as one graph) with elided and with real collectives, piece by piece; row-sharded tables
python3 $R/bench.py --force-dp-path --no-cpu-baseline 2>> $O/log.txt < /dev/null | tail -1 > $O/bench_cfg2_dp_path_result.json
python3 $R/bench.py --force-dp-path --real-collectives --no-cpu-baseline 2>> $O/log.txt < /dev/null | tail -1 > $O/bench_cfg2_dp_path_real_result.json
python3 $R/bench.py --force-dp-path --no-cpu-baseline --no-graph 2>> $O/log.txt < /dev/null | tail -1 > $O/bench_cfg2_dp_path_eager_result.json
(cd $R && bash tools/dp_overhead.sh > /dev/null 2>&1; cp gpurun_out/dp_overhead/result.txt $O/dp_overhead.txt)
python3 $R/bench.py --table-sharding row --no-cpu-baseline 2>> $O/log.txt < /dev/null | tail -1 > $O/bench_cfg2_row_sharded_result.json
(cd $R && rm -f gpurun_out/ab/fuse_final.txt && bash tools/r04_ab.sh fuse_final NASREC_FUSE_FINAL=0 > /dev/null 2>&1; cp gpurun_out/ab/fuse_final.txt $O/ab_fuse_final.txt)
# program launch against graph replay of the cfg-2 step (bench.py decides by engine.prefers_graph; --graph forces the replay), and the gaps between kernels in both
for i in 1 2 3; do for a in "" "--graph"; do
  python3 $R/bench.py --no-cpu-baseline --steps 1000 --warmup 100 --steps-only $a 2>> $O/log.txt < /dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('bench.py --steps 1000 --warmup 100 $a:', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms mean', round(r['median_ms_per_step'],4), 'median;', r['config']['step_submission'])" >> $O/ab_graph_vs_launch.txt
done; done
for a in "" "--graph"; do
  rm -rf /tmp/gaps; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -- python3 $R/bench.py --steps 200 --warmup 20 --steps-only $a > /dev/null 2>&1 < /dev/null
  echo "== rocprofv3 --kernel-trace -- python3 bench.py --steps 200 --warmup 20 --steps-only $a" >> $O/launch_gaps.txt; python3 $R/tools/launch_gaps.py /tmp/gaps >> $O/launch_gaps.txt 2>&1
done
CONFIG=3 TOP=60 timeout 300 python3 $R/tools/supernet_step_table.py > $O/supernet_step_table_cfg3.txt 2>> $O/log.txt < /dev/null
timeout 300 python3 $R/tools/gemm_vs_vendor.py > $O/gemm_vs_vendor.txt 2>> $O/log.txt < /dev/null
timeout 300 python3 $R/tools/parser_bench.py > $O/parser_bench.txt 2>> $O/log.txt < /dev/null
timeout 900 python3 $R/tools/e2e_tsv_run.py --rows 3000000 > $O/e2e_tsv_run.txt 2>> $O/log.txt < /dev/null
# round 6: the search loop at its operating point, the persistent step against the level launches (+ its timeline), the uncached-arena A/B,
# the seam and any-order probes (tools/micro/*.hip are built by the caller: hipcc -O3 --offload-arch=gfx950 -o X X.hip)
timeout 600 python3 $R/tools/search_operating_point.py --candidates 3 2>> $O/log.txt < /dev/null | grep -E "candidate|checkpoint|steady|resident" > $O/search_operating_point.txt
ab() { label=$1; shift; env "$@" python3 $R/bench.py --steps-only --no-cpu-baseline 2>> $O/log.txt < /dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$label:', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms mean', round(r['median_ms_per_step'],4), 'median;', r['config']['plan_buffers'], '/', r['config']['joint_program'])"; }
for i in 1 2 3; do
  ab "level launches, uncached plan buffers + gradient arena (default)" NASREC_PERSIST_DEFAULT=0 >> $O/ab_persist_uc.txt
  ab "level launches, uncached plan buffers, cached gradient arena" NASREC_UC_FLAT= >> $O/ab_persist_uc.txt
  ab "level launches, torch allocator" NASREC_UC_ARENA=0 >> $O/ab_persist_uc.txt
  ab "persistent step (NASREC_PERSIST_DEFAULT=1)" NASREC_PERSIST_DEFAULT=1 >> $O/ab_persist_uc.txt
done
NASREC_PERSIST_THROTTLE=0 timeout 300 python3 $R/tools/persist_timeline.py > $O/persist_timeline.txt 2>> $O/log.txt < /dev/null
[ -x $R/tools/micro/seam_probe ] && for sk in 1 3; do timeout 200 $R/tools/micro/seam_probe --skew $sk --groups 512 --reps 100 >> $O/seam_probe.txt 2>&1; done
[ -x $R/tools/micro/anyorder_probe ] && timeout 60 $R/tools/micro/anyorder_probe >> $O/anyorder_probe.txt 2>&1
rm -rf /tmp/nasrec_tsv_* /tmp/nasrec_search_point /tmp/nasrec_search_ckpt* 2>/dev/null  # (the suite writes ~25 GB of checkpoints itself: one box ran out of disk behind the runs above)
(cd $R && timeout 2400 python3 -m pytest tests -q -m gpu 2>&1 | grep -v "^\[Gloo\]" | tail -6 > $O/gpu_tests.txt)
du -sh $O | tail -1
