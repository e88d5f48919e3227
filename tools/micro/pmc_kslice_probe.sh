cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r3p/direct -- python3 $R/tools/kslice_probe.py > $R/gpurun_out/r3p/direct.log 2>&1
