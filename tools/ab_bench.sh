#!/bin/bash
# ab_bench.sh: runs bench.py (no CPU baseline) over every ab/*.so twice, interleaved, printing ms/step per variant
for rep in 1 2; do
for f in ab/*.so; do
  r=$(NASREC_HIP_LIB=$PWD/$f timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4f ms  fwd %.0f/s  gemm %.2f us" % (d["ms_per_step"], d["forward_only_samples_per_s"], d["roofline"]["avg_launch_us"]))')
  echo "$f rep$rep: $r"
done; done
