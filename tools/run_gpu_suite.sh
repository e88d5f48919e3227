#!/bin/bash
# the whole GPU suite with its full log kept (gpurun_out/suite/): `gpurun --timeout 2400 -- bash tools/run_gpu_suite.sh [repeats]`
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/suite
mkdir -p $O
cd $R
df -h / /tmp | tail -2 > $O/df.txt
for i in $(seq 1 ${1:-1}); do
  timeout 2000 python3 -m pytest tests -q -m gpu -x 2>&1 | grep -v "^\[Gloo\]" > $O/suite_$i.txt
  echo "exit ${PIPESTATUS[0]}" >> $O/suite_$i.txt
  tail -4 $O/suite_$i.txt
  df -h / /tmp | tail -2 >> $O/df.txt; du -sh /tmp 2>/dev/null | tail -1 >> $O/df.txt
done
