#!/bin/bash
# round 5, twelfth GPU call: half-width tiles of the throughput GEMM for launches of at most 256 tiles
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05l
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_fast_gpu.py tests/test_gemm_skinny_gpu.py -q -x > $O/t_gemm.txt 2>&1; echo "gemm tests rc $?" >> $O/summary.txt
tail -5 $O/t_gemm.txt | grep -v "^$" >> $O/summary.txt
python tools/gemm_vs_vendor.py > $O/gemm_vs_vendor.txt 2>&1
NASREC_FAST_HALF_TILES=0 python tools/gemm_vs_vendor.py > $O/gemm_vs_vendor_full_tiles.txt 2>&1
for i in 1 2; do
timeout 600 python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3_$i.json 2>/dev/null
NASREC_FAST_HALF_TILES=0 timeout 600 python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3_full_tiles_$i.json 2>/dev/null
done
timeout 600 python bench.py --config 5 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null
NASREC_FAST_HALF_TILES=0 timeout 600 python bench.py --config 5 --no-cpu-baseline > $O/bench_cfg5_full_tiles.json 2>/dev/null
timeout 900 python -m pytest tests/test_supernet_fullsize_gpu.py tests/test_operating_point_parity_gpu.py -q -x > $O/t_full.txt 2>&1; echo "fullsize tests rc $?" >> $O/summary.txt
tail -4 $O/t_full.txt | grep -v "^$" >> $O/summary.txt
python - <<'P' >> $O/summary.txt
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r05l/bench_*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), round(d["value"]), d["ms_per_step"], d.get("median_ms_per_step"))
    except Exception as e: print(f, "bad", e)
P
cat $O/summary.txt; cat $O/gemm_vs_vendor.txt; echo; cat $O/gemm_vs_vendor_full_tiles.txt
