#!/bin/bash
# round 5, third GPU call: the id half as a worklist item of the joint program; the data-parallel step without side branches; row-sharded
# tables over fixed-capacity slots with the whole step captured
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc $?" >> $O/summary.txt
tail -12 $O/gpu_tests.txt | grep -v "^$" >> $O/summary.txt
b() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/bench_$name.json 2>> $O/log.txt; }
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>> $O/log.txt
b default A=1
b default2 A=1
b ids_on_stage NASREC_IDS_AS_ITEM=0
b one_launch_dedup NASREC_DEDUP_SPLIT_MAX_B=0
b one_launch_dedup2 NASREC_DEDUP_SPLIT_MAX_B=0
EXTRA="--force-dp-path" b dp A=1
EXTRA="--force-dp-path" b dp_nopack NASREC_DP_PACK_TAIL=0
EXTRA="--force-dp-path" b dp_old_dedup NASREC_DEDUP_SPLIT_MAX_B=0
EXTRA="--force-dp-path" b dp_old_dedup_nopack NASREC_DEDUP_SPLIT_MAX_B=0 NASREC_DP_PACK_TAIL=0
EXTRA="--force-dp-path --real-collectives" b dpreal A=1
EXTRA="--table-sharding row" b sharded A=1
EXTRA="--table-sharding row --no-graph" b sharded_nograph A=1
timeout 300 python tools/step_table.py > $O/step_table.txt 2>> $O/log.txt
FB=1 timeout 300 python tools/step_table.py > $O/step_table_fb.txt 2>> $O/log.txt
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
