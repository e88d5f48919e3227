#!/bin/bash
# build_variant.sh NAME [extra hipcc flags]: builds nasrec_amd/lib/variants/NAME.so (ships to the GPU box; select with NASREC_HIP_LIB) from the current csrc (for kernel A/B runs on the GPU box)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p ab/$name nasrec_amd/lib/variants
for f in nasrec_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -c $f -o ab/$name/$(basename $f .hip).o -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -Inasrec_amd/csrc -Wno-unused-result -ffp-contract=on -mllvm -amdgpu-kernarg-preload-count=12 "$@" &
done
wait
/opt/rocm/bin/hipcc -shared -o nasrec_amd/lib/variants/$name.so ab/$name/*.o --offload-arch=gfx950
rm -rf ab/$name
