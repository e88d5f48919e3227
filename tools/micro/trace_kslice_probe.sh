cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3p
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3p/trace -- python3 $R/tools/kslice_probe.py > $R/gpurun_out/r3p/trace.log 2>&1
grep "^K=" $R/gpurun_out/r3p/trace.log
for f in $R/gpurun_out/r3p/trace/*/*kernel_stats.csv; do head -6 "$f" | cut -c1-220; done
