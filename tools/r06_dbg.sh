#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06o
mkdir -p $O
cd $R
df -h / /tmp /dev/shm $R 2>&1 | tee $O/df_before.txt
ulimit -c 0
cat /proc/sys/kernel/core_pattern
timeout 900 python3 -m pytest tests/test_bench_spawn_cpu.py tests/test_data_parallel_2rank_gpu.py tests/test_dedup_split_gpu.py tests/test_device_guard.py -q -m gpu -x 2>&1 | grep -v "^\[Gloo\]" | tail -15
df -h / /tmp /dev/shm $R 2>&1 | tee $O/df_after.txt
find / -xdev -name "core*" -size +10M 2>/dev/null | head; du -sh /tmp 2>/dev/null; ls -la /tmp | head -20
