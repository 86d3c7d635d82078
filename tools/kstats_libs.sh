#!/bin/bash
# the walk's and the resolve kernel's average launch for several builds of the library: tools/kstats_libs.sh Z Y X name ...   (- = the shipping library)
R=${GRAFT_REPO_ROOT:-/root/repo}
Z=$1; Y=$2; X=$3; shift 3
for v in "$@"; do
  if [ "$v" = "-" ]; then unset NELLIE_HIP_LIB; else export NELLIE_HIP_LIB=$R/nellie_amd/variants/libnellie_hip_$v.so; fi
  echo "== $v"; $R/tools/kstats.sh $Z $Y $X 2 2>&1 | grep -E "hessian_|vesselness_queue|hd_count|fillBuffer" | cut -c1-150
done
