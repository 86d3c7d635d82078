#!/bin/bash
# A/B builds of the library: tools/build_variant.sh NAME "-DHV_DEPTH=4 ..."  ->  nellie_amd/variants/libnellie_hip_NAME.so
# (select with NELLIE_HIP_LIB=nellie_amd/variants/libnellie_hip_NAME.so; *.so is git-ignored but travels with gpurun)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/nellie_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wno-unused-value $2 \
  -o $R/nellie_amd/variants/libnellie_hip_$1.so $R/nellie_amd/csrc/nellie_hip.hip -ldl 2>&1 | grep -E "error|spill" ; echo "built $1"
