#!/bin/bash
# Transformer kernels: parity tests, the fp64 comparison of every intermediate, stand-alone launch times (optionally against a variant build:
# tools/run_mha_ab.sh [VARIANT], nasrec_amd/lib/variants/VARIANT.so from tools/build_variant.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/mha_ab
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_ops_gpu.py -q -x -k "mha" 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
(python3 tools/mha_debug.py; N=7 B=2 python3 tools/mha_debug.py; N=48 DIMS=32 B=5 python3 tools/mha_debug.py) 2>&1 | grep -v "e-0[5-9]\|amdgpu.ids" > $O/debug.txt; cat $O/debug.txt
echo "this build:" > $O/mha_bench.txt; timeout 300 python3 tools/mha_bench.py 2>/dev/null >> $O/mha_bench.txt
if [ -n "$1" ]; then echo "variant $1:" >> $O/mha_bench.txt; NASREC_HIP_LIB=$R/nasrec_amd/lib/variants/$1.so timeout 300 python3 tools/mha_bench.py 2>/dev/null >> $O/mha_bench.txt; fi
cat $O/mha_bench.txt
