#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06l
mkdir -p $O; rm -f $O/ab.txt
cd $R
timeout 900 python3 -m pytest -q -x tests/test_gemm_skinny_gpu.py -m gpu 2>&1 | tail -30 > $O/tests.txt
CONFIG=3 TOP=70 timeout 300 python3 tools/supernet_step_table.py > $O/supernet_step_table_cfg3.txt 2>> $O/err.txt
for i in 1 2 3; do
NASREC_TINYK=0 timeout 600 python3 bench.py --config 3 --steps-only --no-cpu-baseline 2>> $O/err.txt | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg3 NASREC_TINYK=0', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms')" >> $O/ab.txt
timeout 600 python3 bench.py --config 3 --steps-only --no-cpu-baseline 2>> $O/err.txt | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg3 tiny-K kernel', round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms')" >> $O/ab.txt
done
tail -4 $O/tests.txt; cat $O/ab.txt; grep -n "gemm_tinyk\|sum of isolated" $O/supernet_step_table_cfg3.txt | head -14; tail -3 $O/err.txt
