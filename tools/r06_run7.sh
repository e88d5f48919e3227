#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06g
mkdir -p $O
cd $R
timeout 200 tools/micro/seam_probe --skew 1 --groups 512 --reps 100 --modes 0,2,6,9 > $O/seam_probe.txt 2>&1
timeout 200 tools/micro/seam_probe --skew 3 --groups 512 --reps 100 --modes 0,2,6,9 >> $O/seam_probe.txt 2>&1
timeout 900 python3 -m pytest -q -x tests/test_persist_gpu.py -m gpu 2>&1 | tail -30 > $O/persist_tests.txt
NASREC_PERSIST_RESPLIT=all timeout 900 python3 -m pytest -q -x tests/test_persist_gpu.py -m gpu 2>&1 | tail -30 >> $O/persist_tests.txt
one() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python3 bench.py --steps-only --no-cpu-baseline 2>> $O/err.txt | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$label', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms mean', round(r['median_ms_per_step'],4), 'median')" >> $O/ab_persist.txt
}
one "level-launches" NASREC_PERSIST_DEFAULT=0
one "persist resplit=slack" NASREC_PERSIST_DEFAULT=1
one "persist resplit=off" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_RESPLIT=off
one "persist resplit=all" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_RESPLIT=all
one "persist resplit=all shard_above=128" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_RESPLIT=all NASREC_PS_SHARD_ABOVE=128
one "persist resplit=all shard_above=256" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_RESPLIT=all NASREC_PS_SHARD_ABOVE=256
one "persist resplit=all shard_above=100000" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_RESPLIT=all NASREC_PS_SHARD_ABOVE=100000
one "persist resplit=all alpha=0.3" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_RESPLIT=all NASREC_PERSIST_ALPHA=0.3
one "persist resplit=all alpha=0.7" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_RESPLIT=all NASREC_PERSIST_ALPHA=0.7
one "level-launches" NASREC_PERSIST_DEFAULT=0
cat $O/seam_probe.txt; cat $O/persist_tests.txt | tail -8; cat $O/ab_persist.txt; tail -3 $O/err.txt
