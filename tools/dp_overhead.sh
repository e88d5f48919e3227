#!/bin/bash
# Where the single-rank cost of the data-parallel exchange step goes (bench.py --force-dp-path, cfg 2): the captured step with all
# collectives, without the all-gathers, without the all-reduces, without any, with 1 / 2 / 8 pieces, next to the plain step.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/dp_overhead
mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 200 python bench.py --force-dp-path --no-cpu-baseline --steps 600 --warmup 50 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$name', round(r['ms_per_step'],4), round(r['median_ms_per_step'],4), r['config'].get('dp_exchange',{}).get('launches_up_to_cut'))" >> $O/result.txt; }
rm -f $O/result.txt
timeout 200 python bench.py --no-cpu-baseline --steps 600 --warmup 50 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('plain', round(r['ms_per_step'],4), round(r['median_ms_per_step'],4))" >> $O/result.txt
run all A=1
run no_gather NASREC_BENCH_DP_SKIP=gather
run no_reduce NASREC_BENCH_DP_SKIP=reduce
run none NASREC_BENCH_DP_SKIP=gather,reduce
run pieces1 NASREC_DP_SEGMENTS=1
run pieces2 NASREC_DP_SEGMENTS=2
run pieces8 NASREC_DP_SEGMENTS=8
run no_pack NASREC_DP_PACK_TAIL=0
run linear NASREC_DP_SEGMENTS=1 NASREC_DP_LATE_IDS=1
run ids_on_side_stream NASREC_DP_IDS_MODE=side
# ... and with RCCL's own kernels inside the captured step (a one-rank gather is a copy kernel otherwise)
runreal() { name=$1; shift; env "$@" timeout 200 python bench.py --force-dp-path --real-collectives --no-cpu-baseline --steps 600 --warmup 50 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$name', round(r['ms_per_step'],4), round(r['median_ms_per_step'],4), r['config'].get('dp_exchange',{}).get('launches_up_to_cut'))" >> $O/result.txt; }
runreal real_all A=1
runreal real_no_gather NASREC_BENCH_DP_SKIP=gather
runreal real_no_reduce NASREC_BENCH_DP_SKIP=reduce
runreal real_no_pack NASREC_DP_PACK_TAIL=0
runreal real_linear NASREC_DP_SEGMENTS=1 NASREC_DP_LATE_IDS=1
cat $O/result.txt
