// Compile-only wrapper: the Hessian walk alone (seconds instead of minutes), for ISA inspection of variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only [-D...] -o /tmp/isa/hv_only.s tools/ubench/hv_only.hip
#include "../../nellie_amd/csrc/nl_common.h"
#include <type_traits>
#include "../../nellie_amd/csrc/device_math.inc"
#include "../../nellie_amd/csrc/hessian.inc"
#include "../../nellie_amd/csrc/hessian_pair.inc"
template __global__ void hessian_v_kernel<2, 8, 2>(const float *, unsigned long long *, const unsigned long long *, int, VolGeom, HessDv<2>, VessP,
                                                      VQueue, int, int, int, int, unsigned int *, unsigned long long *, const float *);
#ifdef HV_ONLY_NP2
template __global__ void hessian_v_kernel<2, 8, 2, 2>(const float *, unsigned long long *, const unsigned long long *, int, VolGeom, HessDv<2>, VessP,
                                                         VQueue, int, int, int, int, unsigned int *, unsigned long long *, const float *);
#endif
