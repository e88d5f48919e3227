#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06f
mkdir -p $O
cd $R
timeout 900 python3 -m pytest -q -x tests/test_persist_gpu.py -m gpu 2>&1 | tail -30 > $O/persist_tests.txt
one() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python3 bench.py --steps-only --no-cpu-baseline 2>> $O/err.txt | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$label', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms mean', round(r['median_ms_per_step'],4), 'median')" >> $O/ab_persist.txt
}
one "level-launches" NASREC_PERSIST_DEFAULT=0
one "persist start a=0.5 wg512" NASREC_PERSIST_DEFAULT=1
one "persist level wg512" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_ORDER=level
one "persist start a=0 wg512" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_ALPHA=0
one "persist start a=1 wg512" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_ALPHA=1
one "persist start a=0.5 wg256" NASREC_PERSIST_DEFAULT=1 NASREC_PS_MAX_WG=256
one "persist start a=0.5 wg128" NASREC_PERSIST_DEFAULT=1 NASREC_PS_MAX_WG=128
one "persist start a=0.5 wg100000" NASREC_PERSIST_DEFAULT=1 NASREC_PS_MAX_WG=100000
one "level-launches" NASREC_PERSIST_DEFAULT=0
NASREC_PERSIST_DEFAULT=1 timeout 300 python3 bench.py --no-cpu-baseline 2>> $O/err.txt | tail -1 > $O/bench_persist_result.json
python3 -c "
import json; r=json.load(open('$O/bench_persist_result.json'))
for row in r['roofline_levels']: print(row['kernel'], round(row['us'],2), round(row['us_isolated'],2), len(row['items']), row['items'][:6])
print(r['roofline_in_step_timing'])" > $O/persist_levels.txt
cat $O/persist_tests.txt | tail -5; cat $O/ab_persist.txt; cat $O/persist_levels.txt; tail -5 $O/err.txt
