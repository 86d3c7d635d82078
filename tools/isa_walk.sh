#!/bin/bash
# ISA summary of the one-pass walk hessian_v_kernel<2,8,true>: registers, scratch, instructions per interior plane step.
# tools/isa_walk.sh [extra -D flags]
OUT=/tmp/isa; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only "$@" \
  -o $OUT/nellie.s $(dirname $0)/../nellie_amd/csrc/nellie_hip.hip 2>/dev/null
awk '/^_Z16hessian_v_kernelILi2ELi8ELb1E/{f=1} f{print} f&&/^\.Lfunc_end/{exit}' $OUT/nellie.s > $OUT/hv.s
awk '/^_Z16hessian_v_kernelILi2ELi8ELb1E/{f=1} f&&/; (NumVgprs|NumSgprs|ScratchSize|Occupancy):/{print} f&&/; Occupancy/{exit}' $OUT/nellie.s
python3 - $OUT/hv.s <<'PY'
import sys, collections
lines = [l.split()[0] for l in open(sys.argv[1]) if l.startswith("\t") and not l.strip().startswith((";", "."))]
idx = [i for i, l in enumerate(lines) if l == "s_barrier"]
gaps = [b - a for a, b in zip(idx, idx[1:])]
print("instructions between consecutive barriers:", gaps[:40])
# the interior unrolled steps are the first run of 6+ similar gaps
for k in range(len(gaps) - 5):
    if max(gaps[k:k + 6]) - min(gaps[k:k + 6]) < 25 and min(gaps[k:k + 6]) > 120:
        seg = lines[idx[k + 1]:idx[k + 2]]
        c = collections.Counter(seg)
        print("interior step:", len(seg), "instructions;", ", ".join(f"{n} {m}" for m, n in c.most_common(14)))
        print("spill traffic in it: writelane", c["v_writelane_b32"], "readlane", c["v_readlane_b32"], "s_nop", c["s_nop"])
        break
PY
