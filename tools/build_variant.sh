#!/bin/bash
# A/B builds of the library: tools/build_variant.sh NAME "-DHV_DEPTH=4 ..."  ->  nellie_amd/variants/libnellie_hip_NAME.so
# (select with NELLIE_HIP_LIB=nellie_amd/variants/libnellie_hip_NAME.so; *.so is git-ignored but travels with gpurun)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R && python -m nellie_amd.build --variant $1 $2 2>&1 | grep -E "error|spill" ; ls $R/nellie_amd/variants/libnellie_hip_$1.so > /dev/null && echo "built $1"
