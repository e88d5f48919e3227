#!/bin/bash
# round 5, ninth GPU call: final logit forward + per-sample backward as one operator; the all-pairs id half with batched loads
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05i
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dedup_split_gpu.py -q -x > $O/t_dedup.txt 2>&1; echo "dedup tests rc $?" >> $O/summary.txt
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_operating_point_parity_gpu.py tests/test_fullsize_gpu.py -q -x > $O/t_parity.txt 2>&1; echo "parity tests rc $?" >> $O/summary.txt
tail -6 $O/t_parity.txt | grep -v "^$" >> $O/summary.txt
NASREC_DEDUP_SPLIT_MAX_B=2048 timeout 600 python tools/dedup_cost.py > $O/dedup_cost.txt 2>&1
for i in 1 2; do
python bench.py --no-cpu-baseline > $O/bench_default$i.json 2>/dev/null
NASREC_FUSE_FINAL=0 python bench.py --no-cpu-baseline > $O/bench_nofuse$i.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_driver_flags$i.json 2>/dev/null
done
python tools/step_table.py > $O/step_table.txt 2>&1
python - <<'P' >> $O/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.environ.get("O","gpurun_out/r05i")+"/bench_*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), round(d["value"]), d["ms_per_step"], d.get("median_ms_per_step"), d.get("kernel_launches_per_step"))
    except Exception as e: print(f, "bad", e)
P
cat $O/summary.txt; head -8 $O/dedup_cost.txt
