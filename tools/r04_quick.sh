#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_sharded_tables_gpu.py tests/test_supernet_fullsize_gpu.py tests/test_worklist_items_gpu.py tests/test_harness_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
