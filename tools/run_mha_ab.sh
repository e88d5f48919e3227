#!/bin/bash
# Transformer bodies A/B (token-major against column-slice, the latter as nasrec_amd/lib/variants/mhaold.so = tools/build_variant.sh mhaold -DMHA_TOK=0)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/mha_ab
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_ops_gpu.py -q -x -k "mha" 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
echo "token-major:" > $O/mha_bench.txt; timeout 300 python3 tools/mha_bench.py >> $O/mha_bench.txt 2>&1
echo "column-slice (MHA_TOK=0):" >> $O/mha_bench.txt; NASREC_HIP_LIB=$R/nasrec_amd/lib/variants/mhaold.so timeout 300 python3 tools/mha_bench.py >> $O/mha_bench.txt 2>&1
cat $O/mha_bench.txt
