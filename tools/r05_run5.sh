#!/bin/bash
# round 5, fifth GPU call: fused split-K restricted to >= 64 tiles; optimizer of a fixed sub-network over its trained ranges; the inf of the
# sharded trajectory test
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python -m pytest tests/test_sharded_tables_gpu.py -q -x > $O/t_sharded_fused_$i.txt 2>&1; echo "sharded (fused) run $i rc $?" >> $O/summary.txt; done
for i in 1 2; do NASREC_WL_FUSE_SPLITK=0 timeout 300 python -m pytest tests/test_sharded_tables_gpu.py -q -x > $O/t_sharded_unfused_$i.txt 2>&1; echo "sharded (unfused) run $i rc $?" >> $O/summary.txt; done
b() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/bench_$name.json 2>> $O/log.txt; }
b fused_min64 A=1
b unfused NASREC_WL_FUSE_SPLITK=0
b fused_min64_2 A=1
b unfused2 NASREC_WL_FUSE_SPLITK=0
b unfused_whole_arena_opt NASREC_WL_FUSE_SPLITK=0 NASREC_FIXED_OPT_TABLE=0
b unfused_whole_arena_opt2 NASREC_WL_FUSE_SPLITK=0 NASREC_FIXED_OPT_TABLE=0
NASREC_WL_FUSE_SPLITK=0 FB=1 timeout 300 python tools/step_table.py > $O/step_table_fb.txt 2>> $O/log.txt
NASREC_WL_FUSE_SPLITK=0 timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests (unfused) rc $?" >> $O/summary.txt
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
