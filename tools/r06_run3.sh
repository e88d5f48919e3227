#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06c
mkdir -p $O
cd $R
for sk in 1 3; do
  timeout 200 tools/micro/seam_probe --skew $sk --groups 512 --reps 100 --modes 0,2,4,5,6,7,8 >> $O/seam_probe.txt 2>&1
done
timeout 900 python3 tools/search_operating_point.py --candidates 2 --profile > $O/search_profile.txt 2> $O/search_profile.err
cat $O/seam_probe.txt; grep -v "^Evaluating\|^Test\|seconds elasped\|^Finetune\|^\[\|^done\|^Done\|^Epoch\|^Learning\|^Data\|^Train" $O/search_profile.txt | tail -75
