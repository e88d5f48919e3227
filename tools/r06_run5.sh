#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06e
mkdir -p $O
cd $R
timeout 900 python3 -m pytest -q -x tests/test_persist_gpu.py -m gpu 2>&1 | tail -30 > $O/persist_tests.txt
for i in 1 2; do for ps in 0 1; do
  NASREC_PERSIST_DEFAULT=$ps timeout 300 python3 bench.py --steps-only --no-cpu-baseline 2>> $O/err.txt | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('NASREC_PERSIST_DEFAULT=$ps', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms mean', round(r['median_ms_per_step'],4), 'median')" >> $O/ab_persist.txt
done; done
cat $O/persist_tests.txt; cat $O/ab_persist.txt; tail -5 $O/err.txt
