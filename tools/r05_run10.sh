#!/bin/bash
# round 5, tenth GPU call: the id half above 256 samples through an LDS hash table, compacted entries in the sums
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dedup_split_gpu.py -q -x > $O/t_dedup.txt 2>&1; echo "dedup tests rc $?" >> $O/summary.txt
tail -5 $O/t_dedup.txt | grep -v "^$" >> $O/summary.txt
NASREC_DEDUP_SPLIT_MAX_B=2048 timeout 600 python tools/dedup_cost.py > $O/dedup_cost.txt 2>&1
NASREC_DEDUP_SPLIT_MAX_B=2048 timeout 900 python -m pytest tests/test_data_parallel_2rank_gpu.py -q -x > $O/t_2rank.txt 2>&1; echo "2-rank tests rc $?" >> $O/summary.txt
cat $O/summary.txt; head -12 $O/dedup_cost.txt
