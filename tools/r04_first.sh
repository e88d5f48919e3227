#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 300 python bench.py > $O/bench_cfg2.out 2> $O/bench_cfg2.err; tail -1 $O/bench_cfg2.out > $O/bench_cfg2.json
timeout 300 python bench.py --force-dp-path --no-cpu-baseline > $O/bench_dp.out 2> $O/bench_dp.err; tail -1 $O/bench_dp.out > $O/bench_dp.json
timeout 300 python bench.py --force-dp-path --no-cpu-baseline --no-graph > $O/bench_dp_nograph.out 2> $O/bench_dp_nograph.err
tail -3 $O/pytest.txt
