#!/bin/bash
# round 6, second GPU call: seam probe with the item form, the new tests, the search loop's operating point, bench with the harness-route leg
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06b
mkdir -p $O
cd $R
for sk in 1 3; do for g in 512 1024; do
  timeout 120 tools/micro/seam_probe --skew $sk --groups $g --reps 100 >> $O/seam_probe.txt 2>&1
done; done
timeout 1200 python3 -m pytest -q -x tests/test_device_guard.py tests/test_data_parallel_2rank_gpu.py tests/test_parity_gpu.py::test_exchange_step_falls_back_to_eager_when_the_capture_is_refused tests/test_parity_gpu.py::test_data_parallel_code_path_single_rank tests/test_operating_point_parity_gpu.py::test_search_point_batch_512_last_layer_only_against_the_oracle -m gpu 2>&1 | tail -25 > $O/new_tests.txt
timeout 1200 python3 tools/search_operating_point.py > $O/search_operating_point.txt 2> $O/search_operating_point.err
timeout 600 python3 bench.py > $O/bench_cfg2.out 2> $O/bench_cfg2.err; tail -1 $O/bench_cfg2.out > $O/bench_cfg2_result.json
tail -6 $O/seam_probe.txt; tail -8 $O/new_tests.txt; grep -v "^Evaluating\|^Test\|seconds elasped\|^Finetune\|^\[" $O/search_operating_point.txt | tail -12; tail -3 $O/search_operating_point.err
python3 -c "
import json; r=json.load(open('$O/bench_cfg2_result.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_us'], r['roofline']['avg_launch_us_isolated'], r.get('roofline_in_step_timing')); print(r.get('unchanged_harness_route'))"
