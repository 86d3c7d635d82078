// Part of libnellie_hip.so (gfx950).  The pair walk (hessian_pair.inc: hessian_v_kernel) lives in its OWN translation unit,
// nellie_hv.hip, compiled with `-mllvm -amdgpu-sched-strategy=max-ilp`: the walk is bound by instruction issue and LDS latency at four
// waves per SIMD, and the ILP-first scheduler orders its 280-instruction steps 5 % better (14.5 -> 13.7 ms per 1024^3 frame) -- while the
// same switch costs the fused Gaussian pass 15 % (9.6 -> 11.0 ms), so it cannot be a flag of the whole library.  Scheduling moves no
// rounding point: the kernels' results are bit for bit the same either way (the whole GPU suite runs on this build).
// nellie_hip.hip fills an HvLaunch and calls nl_hv_launch; everything else about the walk (queue, records, streams) stays there.
#pragma once
#ifndef NL_HV_VARIANTS
#define NL_HV_VARIANTS 0       // 1: the rejected forms of the walk are built too (nellie_hv.hip)
#endif

struct HvLaunch {
    int mode;                 // 0 statistics, 1 known threshold, 2 one pass with a bracket (hessian.inc)
    int rs;                   // rows per wave pair: 8 or 16 (HVCfg<RS>)
    int np;                   // pair-rows per lane: 1, or 2 with rs = 8 (four voxels per lane)
                              // np = 0: the wave-autonomous walk (hessian_dpp.inc), rs = rows of a wave's strip (4), nblocks = WAVES (tiles)
    int fastv;                // division: 0 float64, 1 three instructions, 2 two instructions (proven exact per divisor first)
    unsigned int nblocks;
    hipStream_t stream;
    const float *g;
    unsigned long long *cmask;
    const unsigned long long *pmask;
    int wpr;
    VolGeom geom;
    HessP hp;
    VessP vp;
    VQueue vq;
    int z0, z1, ntx, nty;
    unsigned int *res;
    unsigned long long *d_cnt;
    const float *dev_lohi;
};

hipError_t nl_hv_launch(const HvLaunch &a);

// divisors of np.gradient as the kernels take them (FAST: see hessian.inc)
static inline Dv<1> dv_fast(float d) { return Dv<1>{d, (float)(1.0 / (double)d)}; }
static inline Dv<2> dv_two(float d) {       // yh = RN32(1/d), yl = RN32(1/d - yh), both from the float64 quotient
    const double inv = 1.0 / (double)d;
    const float yh = (float)inv;
    return Dv<2>{(float)(inv - (double)yh), yh};
}
static inline Dv<0> dv_exact(float d) { return Dv<0>{1.0 / (double)d}; }
static inline HessDv<1> hessdv_fast(const HessP &h) { return HessDv<1>{dv_fast(h.hz), dv_fast(h.hy), dv_fast(h.hx), dv_fast(h.hz2), dv_fast(h.hy2), dv_fast(h.hx2)}; }
static inline HessDv<2> hessdv_two(const HessP &h) { return HessDv<2>{dv_two(h.hz), dv_two(h.hy), dv_two(h.hx), dv_two(h.hz2), dv_two(h.hy2), dv_two(h.hx2)}; }
static inline HessDv<0> hessdv_exact(const HessP &h) { return HessDv<0>{dv_exact(h.hz), dv_exact(h.hy), dv_exact(h.hx), dv_exact(h.hz2), dv_exact(h.hy2), dv_exact(h.hx2)}; }
