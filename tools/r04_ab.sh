#!/bin/bash
# A/B of one environment knob on the cfg-2 bench: tools/r04_ab.sh NAME KNOB=VAL ... (three runs each of off / on, interleaved)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
name=$1; shift
one() { env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 1000 --warmup 100 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],4), round(r['median_ms_per_step'],4), len(r.get('roofline_levels',[])))"; }
for i in 1 2 3; do
  echo "$name base: $(one A=1)" >> $O/$name.txt
  echo "$name with $*: $(one "$@")" >> $O/$name.txt
done
cat $O/$name.txt
