#!/bin/bash
# round 5, first GPU call: the two-halves row dedup + the packed data-parallel tail, against tests, then the numbers they move
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dedup_split_gpu.py -x -q > $O/t_dedup.txt 2>&1; echo "dedup tests rc $?" >> $O/summary.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc $?" >> $O/summary.txt
tail -3 $O/gpu_tests.txt >> $O/summary.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>> $O/log.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_flags2.json 2>> $O/log.txt
python bench.py --no-cpu-baseline > $O/bench_default.json 2>> $O/log.txt
NASREC_DEDUP_SPLIT_MAX_B=0 python bench.py --no-cpu-baseline > $O/bench_default_one_launch_dedup.json 2>> $O/log.txt
python bench.py --force-dp-path --no-cpu-baseline > $O/bench_dp.json 2>> $O/log.txt
python bench.py --force-dp-path --real-collectives --no-cpu-baseline > $O/bench_dp_real.json 2>> $O/log.txt
NASREC_DP_PACK_TAIL=0 python bench.py --force-dp-path --real-collectives --no-cpu-baseline > $O/bench_dp_real_nopack.json 2>> $O/log.txt
timeout 300 python tools/dedup_cost.py > $O/dedup_cost.txt 2>> $O/log.txt
timeout 300 python tools/step_table.py > $O/step_table.txt 2>> $O/log.txt
for c in 3 5; do timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_cfg$c.json 2>> $O/log.txt; NASREC_DEDUP_SPLIT_MAX_B=0 timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_cfg${c}_one_launch_dedup.json 2>> $O/log.txt; done
for f in $O/bench_*.json; do python - "$f" <<'P' >> $O/summary.txt
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms', 'median', round(r['median_ms_per_step'],4), r.get('setup_steps'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
P
done
cat $O/summary.txt
