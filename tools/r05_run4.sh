#!/bin/bash
# round 5, fourth GPU call: split-K products with few output tiles as ONE worklist item (main + second pass fused, bit-identical)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_worklist_items_gpu.py tests/test_operating_point_parity_gpu.py -q > $O/gpu_tests_parity.txt 2>&1; echo "parity tests rc $?" >> $O/summary.txt
tail -4 $O/gpu_tests_parity.txt | grep -v "^$" >> $O/summary.txt
b() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/bench_$name.json 2>> $O/log.txt; }
b fused A=1
b unfused NASREC_WL_FUSE_SPLITK=0
b fused2 A=1
b unfused2 NASREC_WL_FUSE_SPLITK=0
b fused_tiles128 NASREC_WL_FUSE_TILES=128
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_flags.json 2>> $O/log.txt
FB=1 ITEMS=5 timeout 300 python tools/step_table.py > $O/step_table_fb.txt 2>> $O/log.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc $?" >> $O/summary.txt
tail -4 $O/gpu_tests.txt | grep -v "^$" >> $O/summary.txt
for f in $O/bench_*.json; do python - "$f" <<'P' >> $O/summary.txt
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(r['value']), 'samples/s', round(r['ms_per_step'],4), 'ms', 'median', round(r['median_ms_per_step'],4), r.get('setup_steps'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
P
done
cat $O/summary.txt
