#!/bin/bash
# several settings of one environment knob on the cfg-2 bench, interleaved, three rounds: tools/run_env_ab.sh NAME KNOB VAL1 VAL2 ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
name=$1; knob=$2; shift 2
one() { env "$@" timeout 200 python3 bench.py --no-cpu-baseline --steps-only --steps 1000 --warmup 100 2>/dev/null | grep '^{' | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],4), round(r['median_ms_per_step'],4))"; }
rm -f $O/$name.txt
for i in 1 2 3; do
  for v in "$@"; do
    echo "$knob=$v: $(one "$knob=$v")" >> $O/$name.txt
  done
done
sort $O/$name.txt
