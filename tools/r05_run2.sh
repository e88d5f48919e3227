#!/bin/bash
# round 5, second GPU call: mask-based id half on the staging launch, LDS-resident OPT_REDUCE2 (quads, 1024 threads above 256 samples),
# where the id half of a data-parallel step should run (NASREC_DP_IDS_MODE)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dedup_split_gpu.py -x -q > $O/t_dedup.txt 2>&1; echo "dedup tests rc $?" >> $O/summary.txt
tail -3 $O/t_dedup.txt >> $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc $?" >> $O/summary.txt
tail -15 $O/gpu_tests.txt | grep -v "^$" >> $O/summary.txt
b() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/bench_$name.json 2>> $O/log.txt; }
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>> $O/log.txt
b default A=1
b default2 A=1
b one_launch_dedup NASREC_DEDUP_SPLIT_MAX_B=0
EXTRA="--force-dp-path" b dp_side NASREC_DP_IDS_MODE=side
EXTRA="--force-dp-path" b dp_chain NASREC_DP_IDS_MODE=chain
EXTRA="--force-dp-path" b dp_main NASREC_DP_IDS_MODE=main
EXTRA="--force-dp-path" b dp_old_dedup NASREC_DEDUP_SPLIT_MAX_B=0
EXTRA="--force-dp-path" b dp_old_dedup_nopack NASREC_DEDUP_SPLIT_MAX_B=0 NASREC_DP_PACK_TAIL=0
EXTRA="--force-dp-path --real-collectives" b dpreal_side NASREC_DP_IDS_MODE=side
EXTRA="--force-dp-path --real-collectives" b dpreal_chain NASREC_DP_IDS_MODE=chain
EXTRA="--force-dp-path --real-collectives" b dpreal_main NASREC_DP_IDS_MODE=main
EXTRA="--force-dp-path --real-collectives" b dpreal_main_nopack NASREC_DP_IDS_MODE=main NASREC_DP_PACK_TAIL=0
EXTRA="--force-dp-path --real-collectives" b dpreal_old_dedup_nopack NASREC_DEDUP_SPLIT_MAX_B=0 NASREC_DP_PACK_TAIL=0
timeout 300 python tools/dedup_cost.py > $O/dedup_cost.txt 2>> $O/log.txt
timeout 300 python tools/step_table.py > $O/step_table.txt 2>> $O/log.txt
for f in $O/bench_*.json; do python - "$f" <<'P' >> $O/summary.txt
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms', 'median', round(r['median_ms_per_step'],4), r.get('setup_steps'), r['config'].get('dp_exchange',{}).get('captured_in_one_graph'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
P
done
cat $O/summary.txt
