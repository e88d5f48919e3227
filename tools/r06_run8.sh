#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06h
mkdir -p $O
cd $R
timeout 300 python3 tools/persist_timeline.py > $O/timeline_default.txt 2> $O/err.txt
cat $O/timeline_default.txt; tail -3 $O/err.txt
