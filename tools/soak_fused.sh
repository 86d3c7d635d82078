#!/bin/bash
# Soak of the fused cascade kernel: tools/soak_crc.py prints the CRCs of the Frangi frame and the labels of N different frames (tubes, scaled
# intensities, pure noise, sinusoidal textures; iso / anisotropic) -- once with the Z march + Y+X pass (NELLIE_GAUSS_FUSED=0), once with the
# fused kernel forced (=1), on several shapes; the outputs must be identical.     tools/soak_fused.sh > gpurun_out/r5/soak_fused.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
bad=0
for cfg in "64 160 200 150" "40 97 263 80" "150 300 330 24" "300 520 640 6"; do
  set -- $cfg
  NELLIE_GAUSS_FUSED=0 python $R/tools/soak_crc.py $1 $2 $3 $4 > /tmp/soak_two.txt 2>/tmp/soak_two.err
  NELLIE_GAUSS_FUSED=1 python $R/tools/soak_crc.py $1 $2 $3 $4 > /tmp/soak_fused.txt 2>/tmp/soak_fused.err
  if cmp -s /tmp/soak_two.txt /tmp/soak_fused.txt && [ -s /tmp/soak_two.txt ]; then
    echo "$1 x $2 x $3: $4 frames, $(awk '{s+=$5} END {print s}' /tmp/soak_two.txt) voxels > 0, two kernels == fused kernel (Frangi CRC, label CRC, label count per frame)"
  else
    echo "$1 x $2 x $3: DIFFERENT"; diff /tmp/soak_two.txt /tmp/soak_fused.txt | head -5; tail -2 /tmp/soak_fused.err; bad=1
  fi
done
exit $bad
