#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06d
mkdir -p $O
cd $R
for i in 1 2; do for uc in 0 1; do
  NASREC_UC_ARENA=$uc timeout 300 python3 bench.py --steps-only --no-cpu-baseline 2>> $O/err.txt | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('NASREC_UC_ARENA=$uc', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms mean', round(r['median_ms_per_step'],4), 'median')" >> $O/ab_uc_arena.txt
done; done
NASREC_UC_ARENA=1 timeout 600 python3 -m pytest -q -x tests/test_parity_gpu.py -m gpu -k "level_scheduled or trajectory or logits_match" 2>&1 | tail -4 >> $O/ab_uc_arena.txt
timeout 600 python3 tools/search_operating_point.py --candidates 3 > $O/search_operating_point.txt 2> $O/search.err
cat $O/ab_uc_arena.txt; grep "candidate\|steady\|resident" $O/search_operating_point.txt; tail -3 $O/err.txt
