#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06i
mkdir -p $O
cd $R
one() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python3 bench.py --steps-only --no-cpu-baseline 2>> $O/err.txt | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$label', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms mean', round(r['median_ms_per_step'],4), 'median')" >> $O/ab_persist.txt
}
one "level-launches" NASREC_PERSIST_DEFAULT=0
for rs in slack all; do for wg in 128 512 4096; do
  one "persist resplit=$rs max_wg=$wg a=0.5" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_RESPLIT=$rs NASREC_PS_MAX_WG=$wg
done; done
for a in 0.8 1.0; do for rs in slack all; do
  one "persist resplit=$rs max_wg=512 a=$a" NASREC_PERSIST_DEFAULT=1 NASREC_PERSIST_RESPLIT=$rs NASREC_PS_MAX_WG=512 NASREC_PERSIST_ALPHA=$a
done; done
one "level-launches" NASREC_PERSIST_DEFAULT=0
cat $O/ab_persist.txt; grep -i "refused" $O/err.txt | head -3
