#!/bin/bash
# round 5, eleventh GPU call: Transformer backward with its weight-gradient chains in two stages; full GPU suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05k
mkdir -p $O
cd $R
export TMPDIR=/tmp
for i in 1 2; do
python bench.py --no-cpu-baseline > $O/bench_default$i.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_driver_flags$i.json 2>/dev/null
done
python tools/step_table.py > $O/step_table.txt 2>&1
timeout 600 python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3.json 2>/dev/null
timeout 2400 python -m pytest tests -q -x -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc $?" >> $O/summary.txt
tail -6 $O/gpu_tests.txt | grep -v "^$" >> $O/summary.txt
python - <<'P' >> $O/summary.txt
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r05k/bench_*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), round(d["value"]), d["ms_per_step"], d.get("median_ms_per_step"))
    except Exception as e: print(f, "bad", e)
P
cat $O/summary.txt
