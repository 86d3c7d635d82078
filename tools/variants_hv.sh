#!/bin/bash
# Walk-only A/B of library variants on the GPU box: tools/variants_hv.sh OUTNAME Z Y X REPS variant...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/ab/$1.txt; shift; mkdir -p gpurun_out/ab
Z=$1; Y=$2; X=$3; REPS=$4; shift 4
CFG=()
for v in "$@"; do CFG+=("NELLIE_HIP_LIB=$R/nellie_amd/variants/libnellie_hip_$v.so"); done
python tools/hv_time.py $Z $Y $X $REPS -- "${CFG[@]}" 2>&1 | sed -e "s#NELLIE_HIP_LIB=$R/nellie_amd/variants/libnellie_hip_##" | tee $OUT
