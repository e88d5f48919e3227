#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06m
mkdir -p $O; rm -f $O/ab2.txt
cd $R
one() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python3 bench.py --steps-only --no-cpu-baseline 2>> $O/err.txt | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$label', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms mean', round(r['median_ms_per_step'],4), 'median')" >> $O/ab2.txt
}
for i in 1 2; do
one "NASREC_UC_FLAT=g" NASREC_UC_FLAT=g
one "NASREC_UC_FLAT=gs" NASREC_UC_FLAT=gs
one "NASREC_UC_FLAT=gsp" NASREC_UC_FLAT=gsp
one "NASREC_UC_FLAT=(none)" NASREC_UC_FLAT=
done
cat $O/ab2.txt; tail -2 $O/err.txt
