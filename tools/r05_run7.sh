#!/bin/bash
# round 5, seventh GPU call: world size 2 on one GPU (gloo), the data-parallel step piece by piece once more (optimizer over the trained ranges, linear form)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05g
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_data_parallel_2rank_gpu.py -q -x > $O/t_2rank.txt 2>&1; echo "2-rank tests rc $?" >> $O/summary.txt
tail -15 $O/t_2rank.txt | grep -v "^$" >> $O/summary.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -q -k "data_parallel" > $O/t_dp.txt 2>&1; echo "dp single-rank test rc $?" >> $O/summary.txt
tail -3 $O/t_dp.txt | grep -v "^$" >> $O/summary.txt
bash tools/dp_overhead.sh > /dev/null 2>&1; cp gpurun_out/dp_overhead/result.txt $O/dp_overhead.txt
cat $O/summary.txt; cat $O/dp_overhead.txt
