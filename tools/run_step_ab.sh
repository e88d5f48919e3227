#!/bin/bash
# cfg 2-5 bench lines (steps only) on the default build and on a variant (NASREC_HIP_LIB), interleaved: bash tools/run_step_ab.sh VARIANT [cfgs]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$1; shift
O=$R/gpurun_out/step_ab_$V
mkdir -p $O
cd $R
one() { label=$1; shift; env "$@" python3 bench.py --steps-only --no-cpu-baseline $EXTRA 2>> $O/log.txt < /dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$label:', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms mean', round(r['median_ms_per_step'],4), 'median')"; }
for c in ${@:-2 3 5}; do
  EXTRA="--config $c"
  for i in 1 2 3; do
    one "cfg $c default build" X=1 >> $O/ab.txt
    one "cfg $c variant $V" NASREC_HIP_LIB=$R/nasrec_amd/lib/variants/$V.so >> $O/ab.txt
  done
done
cat $O/ab.txt
