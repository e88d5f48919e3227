#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06h
mkdir -p $O
cd $R
NASREC_PERSIST_THROTTLE=0 NASREC_TIMELINE_JSON=$O/tl_nothrottle.json timeout 300 python3 tools/persist_timeline.py > $O/timeline_nothrottle.txt 2> $O/err.txt
tail -3 $O/err.txt; head -3 $O/timeline_nothrottle.txt
