#!/bin/bash
# the balancing model's level latency and product speed as knobs: one cfg-2 bench run per setting
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
one() { env "$@" timeout 200 python3 bench.py --no-cpu-baseline --steps-only --steps 1000 --warmup 100 2>/dev/null | grep '^{' | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],4), round(r['median_ms_per_step'],4))"; }
rm -f $O/balance_model_sweep.txt
for l in ${LAT:-3000 4000 5000 6000 7000}; do for g in ${DIV:-12000 16000 20000 26000 34000}; do
  echo "lat $l div $g: $(one NASREC_WL_LAT_NS=$l NASREC_WL_GEMM_DIV=$g)" >> $O/balance_model_sweep.txt
done; done
sort -k5 -n $O/balance_model_sweep.txt | head -30
