#!/bin/bash
# the whole GPU suite, twice, full log kept on failure (a flaky hand-off or a worker that dies at exit shows up in repetition)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06n
mkdir -p $O
cd $R
df -h / | tail -1
for i in 1 2; do
  timeout 1500 python3 -m pytest tests -q -m gpu 2>&1 | grep -v "^\[Gloo\]" > $O/suite_$i.txt
  tail -3 $O/suite_$i.txt
  grep -n "FAILED\|ERROR" $O/suite_$i.txt | head -10
  df -h / | tail -1; du -sh /tmp 2>/dev/null
done
