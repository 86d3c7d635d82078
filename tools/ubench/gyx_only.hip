// Compile-only wrapper: the fused Y+X Gaussian pass alone, for ISA inspection.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only -o /tmp/isa/gyx_only.s tools/ubench/gyx_only.hip
#include "../../nellie_amd/csrc/nl_common.h"
#include <type_traits>
#include "../../nellie_amd/csrc/device_math.inc"
#include "../../nellie_amd/csrc/gauss.inc"
template __global__ void gauss_yx_tile_kernel<4, false>(const float *, float *, VolGeom, i64, i64, GaussWS, GaussWS, int, int, int);
