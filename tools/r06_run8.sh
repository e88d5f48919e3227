#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06h
mkdir -p $O
cd $R
NASREC_PERSIST_RESPLIT=all timeout 300 python3 tools/persist_timeline.py > $O/timeline.txt 2> $O/err.txt
cat $O/timeline.txt; tail -5 $O/err.txt
