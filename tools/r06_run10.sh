#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06j
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.txt
timeout 600 python3 bench.py > $O/bench_cfg2.out 2> $O/bench_cfg2.err; tail -1 $O/bench_cfg2.out > $O/bench_cfg2_result.json
tail -6 $O/gpu_tests.txt; python3 -c "
import json; r=json.load(open('$O/bench_cfg2_result.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_us'], r['roofline']['avg_launch_us_isolated'], r.get('roofline_in_step_timing'))"
