# SQ instruction / wait counters of the LDS-DMA probe kernel (tools/micro/kdma_probe.hip) and of the product kernel under tools/kslice_probe.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3p
hipcc -w -O3 --offload-arch=gfx950 -o /tmp/kd $R/tools/micro/kdma_probe.hip
for grp in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INSTS_MFMA"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/r3p/pmc_probe_$tag -- /tmp/kd 0 > /dev/null 2>&1 < /dev/null
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/r3p/pmc_prod_$tag -- python3 $R/tools/kslice_probe.py > /dev/null 2>&1 < /dev/null
done
ls $R/gpurun_out/r3p | head -30
