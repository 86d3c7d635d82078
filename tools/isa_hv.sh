#!/bin/bash
# ISA summary of hessian_v_kernel<2,8,2> (the default variant: two-instruction division) compiled alone (tools/ubench/hv_only.hip): tools/isa_hv.sh [TAG] [-D flags]
TAG=${1:-base}; shift
OUT=/tmp/isa; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -S --cuda-device-only "$@" \
  -o $OUT/hv_$TAG.s $(dirname $0)/ubench/hv_only.hip 2>&1 | grep -E "error|warning: .*spill" | head
grep -E "; (NumVgprs|ScratchSize|Occupancy|LDSByteSize):" $OUT/hv_$TAG.s | head -4
python3 - $OUT/hv_$TAG.s <<'PY'
import sys, collections
lines = [l.split()[0] for l in open(sys.argv[1]) if l.startswith("\t") and not l.strip().startswith((";", "."))]
idx = [i for i, l in enumerate(lines) if l == "s_barrier"]
gaps = [b - a for a, b in zip(idx, idx[1:])]
print("instructions between consecutive barriers:", gaps[:26])
PY
