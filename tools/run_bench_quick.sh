#!/bin/bash
# quick bench lines (steps only, no CPU baseline), three runs per configuration: tools/run_bench_quick.sh [cfgs]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/quick
mkdir -p $O
cd $R
rm -f $O/bench.txt
for c in ${@:-2 3}; do for i in 1 2 3; do
  python3 bench.py --steps-only --no-cpu-baseline --config $c 2>> $O/log.txt < /dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg $c:', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms mean', round(r['median_ms_per_step'],4), 'median')" >> $O/bench.txt
done; done
cat $O/bench.txt
