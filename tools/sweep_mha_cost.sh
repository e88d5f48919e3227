#!/bin/bash
# the balancing pass's Transformer-item durations as tuning knobs: one cfg-2 bench run per setting (tools/run_env_ab.sh confirms the winners)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
one() { env "$@" timeout 200 python3 bench.py --no-cpu-baseline --steps-only --steps 1000 --warmup 100 2>/dev/null | grep '^{' | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],4), round(r['median_ms_per_step'],4))"; }
rm -f $O/mha_cost_sweep.txt
for f in ${FWD:-6000 8000 10500 13000}; do for b in ${BWD:-12000 14000 16000 18500 21000 24000}; do for dv in ${DIV:-4}; do
  echo "fwd $f bwd $b div $dv: $(one NASREC_WL_MHA_NS=$f,0,$b,0,$dv,$dv)" >> $O/mha_cost_sweep.txt
done; done; done
sort -k7 -n $O/mha_cost_sweep.txt | head -40
