#!/bin/bash
# Counter-based bound of the Transformer kernels at large batch (mha_fwd_kernel<4>, mha_bwd_kernel<4>, B = 4096, full-path supernet step):
# two rocprofv3 --pmc passes (SQ block: 8 slots each; --kernel-trace only), summarised per kernel by tools/summarize_pmc.py.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=${1:-$R/gpurun_out/mha_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
timeout 400 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/passA -- python3 $R/tools/bench_supernet.py --strategy full-path --steps 3 --warmup 1 > $O/passA.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc $B --kernel-trace --output-format csv -d $O/passB -- python3 $R/tools/bench_supernet.py --strategy full-path --steps 3 --warmup 1 > $O/passB.log 2>&1 < /dev/null
for p in A B; do
  f=$(find $O/pass$p -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/summarize_pmc.py $f | grep -E "^kernel|mha_" > $O/mha_pmc_pass$p.csv
done
find $O -name "*.csv" -size +4M -delete
cat $O/mha_pmc_passA.csv $O/mha_pmc_passB.csv 2>/dev/null | head -60; tail -3 $O/passB.log
