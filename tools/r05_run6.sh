#!/bin/bash
# round 5, sixth GPU call: the state after the memset fix — whole GPU suite, cfg 2 at the driver's flags and at the defaults, the
# data-parallel step on one rank piece by piece, dedup cost at global batch sizes, row-sharded tables
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc $?" >> $O/summary.txt
tail -4 $O/gpu_tests.txt | grep -v "^$" >> $O/summary.txt
b() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/bench_$name.json 2>> $O/log.txt; }
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>> $O/log.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags2.json 2>> $O/log.txt
b default A=1
b default2 A=1
EXTRA="--table-sharding row" b sharded A=1
EXTRA="--table-sharding row" b sharded2 A=1
EXTRA="--ids zipf" b zipf A=1
NASREC_DEDUP_SPLIT_MAX_B=2048 timeout 300 python tools/dedup_cost.py > $O/dedup_cost.txt 2>> $O/log.txt
bash tools/dp_overhead.sh > /dev/null 2>&1; cp gpurun_out/dp_overhead/result.txt $O/dp_overhead.txt
FB=1 timeout 300 python tools/step_table.py > $O/step_table_fb.txt 2>> $O/log.txt
for f in $O/bench_*.json; do python - "$f" <<'P' >> $O/summary.txt
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms', 'median', round(r['median_ms_per_step'],4), r.get('setup_steps'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
P
done
cat $O/summary.txt; cat $O/dp_overhead.txt
