#!/bin/bash
# diagnostic build of the library with in-kernel cycle stamps in gemm_fast_kernel (never shipped: writes to nasrec_amd/lib/libnasrec_hip_stamps${TAG}.so)
set -e
cd "$(dirname "$0")/.."
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -Inasrec_amd/csrc -Wno-unused-result -ffp-contract=on"
/opt/rocm/bin/hipcc -c nasrec_amd/csrc/gemm_fast.hip -o /tmp/gemm_fast_stamps.o $F -DFT_STAMPS $EXTRA
OBJS=$(ls nasrec_amd/lib/*.o | grep -v gemm_fast.o)
/opt/rocm/bin/hipcc -shared -o nasrec_amd/lib/libnasrec_hip_stamps${TAG}.so $OBJS /tmp/gemm_fast_stamps.o --offload-arch=gfx950
