#!/bin/bash
# round 6, first GPU call: seam probe (go / no-go for the persistent step), the GPU suite on the fixed tree, a default bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06a
mkdir -p $O
cd $R
for sk in 1 3; do for g in 512 1024; do
  timeout 120 tools/micro/seam_probe --skew $sk --groups $g --reps 100 >> $O/seam_probe.txt 2>&1
done; done
timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.txt
timeout 600 python3 bench.py > $O/bench_cfg2.out 2> $O/bench_cfg2.err; tail -1 $O/bench_cfg2.out > $O/bench_cfg2_result.json
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $O/bench_cfg2.err | tail -1 > $O/bench_cfg2_driver_flags_result.json
tail -5 $O/seam_probe.txt; cat $O/gpu_tests.txt | tail -5; python3 -c "
import json; r=json.load(open('$O/bench_cfg2_result.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_us'], r['roofline']['avg_launch_us_isolated'])"
