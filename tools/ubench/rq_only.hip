// Compile-only wrapper: the resolve kernel alone (seconds instead of minutes), for ISA inspection.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only -o /tmp/isa/rq_only.s tools/ubench/rq_only.hip
#include "../../nellie_amd/csrc/nl_common.h"
#include <type_traits>
#include "../../nellie_amd/csrc/device_math.inc"
#include "../../nellie_amd/csrc/hessian.inc"
template __global__ void vesselness_queue_kernel<true>(const float4 *, const unsigned int *, unsigned int, float *, i64, VessP, unsigned long long *,
                                                        const unsigned long long *, int, int, int, i64, unsigned long long *, const float *);
