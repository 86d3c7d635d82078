// Second translation unit of libnellie_hip.so (gfx950): the pair walk, compiled with the ILP-first scheduler (see hv_launch.h and
// nellie_amd/build.py).  Only the kernel and its launch live here.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <type_traits>
#define NL_HV_UNIT 1
#include "nl_common.h"
#include "device_math.inc"
#include "hessian.inc"
#include "hessian_pair.inc"
#include "hessian_dpp.inc"
#include "hv_launch.h"

// dynamic LDS beyond 64 KiB has to be allowed per kernel (the RS = 16 tile of the pair kernel takes 121 KiB)
template <typename K> static void allow_lds(K kernel, int bytes) {
    if (bytes > (64 << 10)) (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int MODE, int RS, int FAST, int NP>
static void launch_np(const HvLaunch &a, const HessDv<FAST> &hr) {
    allow_lds(hessian_v_kernel<MODE, RS, FAST, NP>, HVCfg<RS, NP>::lds_bytes());
    hessian_v_kernel<MODE, RS, FAST, NP><<<a.nblocks, HVCfg<RS, NP>::NT, HVCfg<RS, NP>::lds_bytes(), a.stream>>>(
        a.g, a.cmask, a.pmask, a.wpr, a.geom, hr, a.vp, a.vq, a.z0, a.z1, a.ntx, a.nty, a.res, a.d_cnt, a.dev_lohi);
}
// The shipping library holds SIX instantiations: modes 0-2 x {the two-instruction division, the float64 division}, 16-row tiles, two voxels
// per lane.  What measurement rejected is built on request only (tools/build_variant.sh hv_all "-DNL_HV_VARIANTS=1"): 32-row tiles
// (NELLIE_HV_RS=16), four voxels per lane (NELLIE_HV_NP=2: 26 % slower, docs/HISTORY.md round 5) and the three-instruction division
// (NELLIE_EXACT_DIV=3; a divisor the two-instruction form is not proven for takes the float64 form here).
template <int MODE, int RS, int FAST>
static void launch_one(const HvLaunch &a, const HessDv<FAST> &hr) {
#if NL_HV_VARIANTS
    if constexpr (RS == 8 && FAST == 2) { if (a.np == 2) { launch_np<MODE, RS, FAST, 2>(a, hr); return; } }
#endif
    launch_np<MODE, RS, FAST, 1>(a, hr);
}
template <int MODE, int RS>
static void launch_div(const HvLaunch &a) {
    if (a.fastv == 2) launch_one<MODE, RS, 2>(a, hessdv_two(a.hp));
#if NL_HV_VARIANTS
    else if (a.fastv == 1) launch_one<MODE, RS, 1>(a, hessdv_fast(a.hp));
#endif
    else launch_one<MODE, RS, 0>(a, hessdv_exact(a.hp));
}
template <int MODE>
static void launch_rs(const HvLaunch &a) {
#if NL_HV_VARIANTS
    if (a.rs == 16) { launch_div<MODE, 16>(a); return; }
#endif
    launch_div<MODE, 8>(a);
}

// the wave-autonomous walk (hessian_dpp.inc): a.nblocks counts waves, four to a workgroup
template <int MODE, int FAST>
static void launch_dpp(const HvLaunch &a, const HessDv<FAST> &hr) {
    hessian_d_kernel<MODE, 4, FAST><<<(a.nblocks + 3u) / 4u, HD_NT, 0, a.stream>>>(
        a.g, a.cmask, a.pmask, a.wpr, a.geom, hr, a.vp, a.vq, a.z0, a.z1, a.ntx, a.nty, a.res, a.d_cnt, a.dev_lohi);
    if (MODE != 0 && a.d_cnt) hd_count_kernel<<<1, 1024, 0, a.stream>>>(a.vq.count + a.nblocks, a.nblocks, a.d_cnt);
}
template <int MODE>
static void launch_dpp_div(const HvLaunch &a) {
    if (a.fastv == 2) launch_dpp<MODE, 2>(a, hessdv_two(a.hp)); else launch_dpp<MODE, 0>(a, hessdv_exact(a.hp));
}

hipError_t nl_hv_launch(const HvLaunch &a) {
    if (a.np == 0) {          // (the one-pass mode only: statistics and known-threshold walks are the rare two-pass fallback and stay with the pair kernel)
        if (a.rs != 4 || a.mode != 2) return hipErrorInvalidValue;
        launch_dpp_div<2>(a);
        return hipGetLastError();
    }
    if (a.rs != 8 && !(NL_HV_VARIANTS && a.rs == 16)) return hipErrorInvalidValue;
    if (a.np != 1 && !(NL_HV_VARIANTS && a.np == 2 && a.rs == 8 && a.fastv == 2)) return hipErrorInvalidValue;
    switch (a.mode) {
        case 0: launch_rs<0>(a); break;
        case 1: launch_rs<1>(a); break;
        case 2: launch_rs<2>(a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
