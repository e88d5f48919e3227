#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06k
mkdir -p $O
cd $R
timeout 900 python3 -m pytest -q -x tests/test_persist_gpu.py -m gpu 2>&1 | tail -30 > $O/persist_tests.txt
one() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python3 bench.py --steps-only --no-cpu-baseline 2>> $O/err.txt | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$label', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms mean', round(r['median_ms_per_step'],4), 'median')" >> $O/ab_persist.txt
}
one "level-launches" NASREC_PERSIST_DEFAULT=0
for th in 0 0.25 0.5 1.0; do for a in 0.5 1.0; do
  one "persist throttle=$th alpha=$a" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_THROTTLE=$th NASREC_PERSIST_ALPHA=$a
done; done
one "persist throttle=0.5 alpha=0.5 resplit=slack" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_RESPLIT=slack
one "persist throttle=0.5 alpha=0.5 resplit=all" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_RESPLIT=all
one "level-launches" NASREC_PERSIST_DEFAULT=0
tail -4 $O/persist_tests.txt; cat $O/ab_persist.txt; grep -i "refused" $O/err.txt | head -3
