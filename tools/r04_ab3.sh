#!/bin/bash
# A/B of one environment knob on a supernet bench config: tools/r04_ab3.sh NAME CONFIG KNOB=VAL ... (two runs each, interleaved)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
name=$1; cfg=$2; shift; shift
one() { env "$@" timeout 400 python bench.py --config $cfg --no-cpu-baseline --steps 120 --warmup 10 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],4), round(r['median_ms_per_step'],4), r['roofline_step']['launches_per_step'], round(r['value']))"; }
for i in 1 2; do
  echo "$name cfg$cfg base: $(one A=1)" >> $O/$name.txt
  echo "$name cfg$cfg with $*: $(one "$@")" >> $O/$name.txt
done
cat $O/$name.txt
