// libnellie_hip.so -- hand-written HIP for gfx950 (MI355X): Nellie's Filter -> Label hot path.
// C-ABI in include/nellie_amd.h.  Compile with -ffp-contract=off: every float operation
// below is meant to round exactly where numpy/scipy round.
#include <stdarg.h>
#include <stdlib.h>
#include <type_traits>
#include <rccl/rccl.h>
#include "nl_common.h"

#define NL_MASK_SLOTS 8      // per-scale mask bit planes kept before falling back to read-modify-write
#define NL_VERSION "nellie_amd-hip 0.1.0 (gfx950)"

// =================================================================================================
// device helpers
// =================================================================================================
struct VolGeom {
    i64 nzl, ny, nx;   // local shape
    i64 gz0, gnz;      // global placement of local plane 0, global plane count
};

struct HessP {
    float hz, hy, hx;      // float32(h)      : one-sided divisor at a face
    float hz2, hy2, hx2;   // float32(2.0*h)  : central divisor
};

// scipy NI_EXTEND_REFLECT (d c b a | a b c d | d c b a)
__device__ __forceinline__ i64 reflect_idx(i64 i, i64 n) {
    if (i >= 0 && i < n) return i;
    i64 p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - 1 - i : i;
}

// Exactly rounded float32 division by a constant d, given rcp = RN64(1/d):
//   a/d = (float)((double)a * rcp).
// Why it is exact: a/d can never sit closer than ~2^-48 (relative) to a float32 rounding boundary
// (a - m*d is a non-zero integer multiple of 2^-47 for 24-bit a, d and a 25-bit midpoint m), while the
// float64 product carries a relative error <= 2^-52, so the final rounding lands on the IEEE quotient.
// (Sub-normal quotients are the one exception; differences of image intensities never get there.)
// 3 instructions instead of the ~12 of an IEEE division; checked bit-for-bit on 1.2e9 operands.
__device__ __forceinline__ float div_c(float a, double rcp) { return (float)((double)a * rcp); }

// wave64 reductions (CDNA wavefront = 64 lanes)
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_or_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}

// -------------------------------------------------------------------------------------------------
// np.gradient applied twice (filtering.py:518-536), fused: every first derivative is rounded to
// float32 (true IEEE division by float32(2h) / float32(h)) before it is differenced again.
// Faces use clamped indices + the one-sided divisor, independently at both stages.
// Component order = the reference's (hxx,hxy,hxz,hyy,hyz,hzz) with "x" = axis 0 (Z).
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void hessian_at(const float *__restrict__ g, const VolGeom &v, const HessP &hp,
                                           i64 z, i64 y, i64 x, float h[6]) {
    const i64 sy = v.nx, sz = v.ny * v.nx;
    const i64 gz = v.gz0 + z;
    // outer-difference sites and divisors
    const bool z_lo = (gz == 0), z_hi = (gz == v.gnz - 1);
    const bool y_lo = (y == 0), y_hi = (y == v.ny - 1);
    const bool x_lo = (x == 0), x_hi = (x == v.nx - 1);
    const i64 zl = z_lo ? z : z - 1, zh = z_hi ? z : z + 1;
    const i64 yl = y_lo ? y : y - 1, yh = y_hi ? y : y + 1;
    const i64 xl = x_lo ? x : x - 1, xh = x_hi ? x : x + 1;
    const float dz = (z_lo || z_hi) ? hp.hz : hp.hz2;
    const float dy = (y_lo || y_hi) ? hp.hy : hp.hy2;
    const float dx = (x_lo || x_hi) ? hp.hx : hp.hx2;

    auto F = [&](i64 zz, i64 yy, i64 xx) -> float { return g[zz * sz + yy * sy + xx]; };
    auto GZ = [&](i64 zz, i64 yy, i64 xx) -> float {
        const i64 gg = v.gz0 + zz;
        const bool lo = (gg == 0), hi = (gg == v.gnz - 1);
        const float a = F(lo ? zz : zz - 1, yy, xx), b = F(hi ? zz : zz + 1, yy, xx);
        return (b - a) / ((lo || hi) ? hp.hz : hp.hz2);
    };
    auto GY = [&](i64 zz, i64 yy, i64 xx) -> float {
        const bool lo = (yy == 0), hi = (yy == v.ny - 1);
        const float a = F(zz, lo ? yy : yy - 1, xx), b = F(zz, hi ? yy : yy + 1, xx);
        return (b - a) / ((lo || hi) ? hp.hy : hp.hy2);
    };
    auto GX = [&](i64 zz, i64 yy, i64 xx) -> float {
        const bool lo = (xx == 0), hi = (xx == v.nx - 1);
        const float a = F(zz, yy, lo ? xx : xx - 1), b = F(zz, yy, hi ? xx : xx + 1);
        return (b - a) / ((lo || hi) ? hp.hx : hp.hx2);
    };
    h[0] = (GZ(zh, y, x) - GZ(zl, y, x)) / dz;   // hxx = d0 g0
    h[1] = (GZ(z, yh, x) - GZ(z, yl, x)) / dy;   // hxy = d1 g0
    h[2] = (GZ(z, y, xh) - GZ(z, y, xl)) / dx;   // hxz = d2 g0
    h[3] = (GY(z, yh, x) - GY(z, yl, x)) / dy;   // hyy = d1 g1
    h[4] = (GY(z, y, xh) - GY(z, y, xl)) / dx;   // hyz = d2 g1
    h[5] = (GX(z, y, xh) - GX(z, y, xl)) / dx;   // hzz = d2 g2
}

// filtering.py:538-543, float32, numpy's left-to-right association
__device__ __forceinline__ float frob_sq_of(const float h[6]) {
    const float a = h[0] * h[0] + h[3] * h[3] + h[5] * h[5];
    const float b = 2.0f * (h[1] * h[1] + h[2] * h[2] + h[4] * h[4]);
    return a + b;
}

// frob = sqrt(frob_sq)/max_abs with +inf -> max finite (filtering.py:421-426, 562)
__device__ __forceinline__ float frob_norm(float fsq, float max_abs, float max_finite) {
    float fr = sqrtf(fsq) / max_abs;
    if (isinf(fr)) fr = max_finite;
    return fr;
}

// -------------------------------------------------------------------------------------------------
// numpy.linalg.eigvalsh on a float32 3x3 symmetric matrix = float64 LAPACK result cast to
// float32 (numpy/linalg/_linalg.py computes in double).  Closed form in float64 (Smith 1961),
// then the |lambda| ordering of filtering.py:583-584 and the Frangi response of
// filtering.py:744-766 in float32.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void eig3_sorted_abs_libm(const float h[6], float &l1, float &l2, float &l3) {
    const double a00 = h[0], a01 = h[1], a02 = h[2], a11 = h[3], a12 = h[4], a22 = h[5];
    const double q = (a00 + a11 + a22) / 3.0;
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double p2 = (b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * (a01 * a01 + a02 * a02 + a12 * a12)) / 6.0;
    const double p = sqrt(p2);
    const double det = b00 * (b11 * b22 - a12 * a12) - a01 * (a01 * b22 - a12 * a02) + a02 * (a01 * a12 - b11 * a02);
    double r = (p2 > 0.0) ? det / (2.0 * p2 * p) : 0.0;
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    const double phi = acos(r) / 3.0;
    const double e_max = q + 2.0 * p * cos(phi);
    const double e_min = q + 2.0 * p * cos(phi + 2.0943951023931953 /* 2*pi/3 */);
    const double e_mid = 3.0 * q - e_max - e_min;
    float a = (float)e_min, b = (float)e_mid, c = (float)e_max;   // ascending, as eigvalsh returns
    // stable insertion sort by |.| (numpy argsort of 3 elements)
    float ka = fabsf(a), kb = fabsf(b), kc = fabsf(c);
    if (kb < ka) { float t = a; a = b; b = t; t = ka; ka = kb; kb = t; }
    if (kc < kb) {
        float t = b; b = c; c = t; t = kb; kb = kc; kc = t;
        if (kb < ka) { t = a; a = b; b = t; t = ka; ka = kb; kb = t; }
    }
    l1 = a; l2 = b; l3 = c;
}

// 1/sqrt(x) in float64 from the hardware estimate + two Newton steps (x > 0, normal range)
__device__ __forceinline__ double rsqrt_f64(double x) {
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double t = x * y;
        const double e = fma(-t, y, 1.0);
        y = fma(0.5 * y, e, y);
    }
    return y;
}

// Stable |lambda| ordering of three ascending values (numpy argsort on 3 elements = insertion sort)
__device__ __forceinline__ void sort3_abs(float a, float b, float c, float &l1, float &l2, float &l3) {
    float ka = fabsf(a), kb = fabsf(b), kc = fabsf(c);
    if (kb < ka) { float t = a; a = b; b = t; t = ka; ka = kb; kb = t; }
    if (kc < kb) {
        float t = b; b = c; c = t; t = kb; kb = kc; kc = t;
        if (kb < ka) { t = a; a = b; b = t; t = ka; ka = kb; kb = t; }
    }
    l1 = a; l2 = b; l3 = c;
}

// Production eigen-solve: the same closed form, no libm.  With r = det(B)/(2 p^3) in [-1,1] and
// phi = acos(|r|)/3 in [0, pi/6], c = cos(phi) is the largest root of 4c^3 - 3c = |r| (c in [0.866,1],
// derivative 12c^2-3 in [6,9]: well conditioned): degree-4 starting polynomial (4.5e-6) + two Newton
// steps with a float32 reciprocal of the derivative -> 1e-16.  s = sin(phi) = sqrt(1-c^2).
//   r >= 0: e_max = q + 2pc, e_min = q - p(c + sqrt3 s);   r < 0: e_min = q - 2pc, e_max = q + p(c + sqrt3 s)
// Accuracy ~1e-16 * ||A|| (near-degenerate pairs ~1e-8 * ||A||, exactly like the acos form), so the
// float32-rounded eigenvalues equal LAPACK's except with probability ~1e-9 per value.
__device__ __forceinline__ void eig3_sorted_abs(const float h[6], float &l1, float &l2, float &l3) {
    const double a00 = h[0], a01 = h[1], a02 = h[2], a11 = h[3], a12 = h[4], a22 = h[5];
    const double q = (a00 + a11 + a22) * (1.0 / 3.0);
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double off = fma(a01, a01, fma(a02, a02, a12 * a12));
    const double p2 = (fma(b00, b00, fma(b11, b11, b22 * b22)) + 2.0 * off) * (1.0 / 6.0);
    const double m0 = fma(b11, b22, -(a12 * a12));
    const double m1 = fma(a01, b22, -(a12 * a02));
    const double m2 = fma(a01, a12, -(b11 * a02));
    const double det = fma(b00, m0, fma(-a01, m1, a02 * m2));
    float ea, eb, ec;
    if (p2 > 0.0) {
        const double y = rsqrt_f64(p2);
        const double p = p2 * y;
        double r = 0.5 * det * (y * y * y);
        const bool neg = r < 0.0;
        double ra = fabs(r);
        ra = ra > 1.0 ? 1.0 : ra;
        double c = fma(fma(fma(fma(-0.004099405456413993, ra, 0.017642364887819385), ra, -0.046005261357895906), ra,
                           0.16642901838062307), ra, 0.866029853359238);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double t = c * c;
            const double f = fma(c, fma(4.0, t, -3.0), -ra);
            const double fp = fma(12.0, t, -3.0);
            c = fma(-f, (double)__builtin_amdgcn_rcpf((float)fp), c);
        }
        c = c > 1.0 ? 1.0 : c;
        const double s2 = fma(-c, c, 1.0);
        const double s = s2 > 0.0 ? s2 * rsqrt_f64(s2) : 0.0;
        const double t1 = 2.0 * p * c;
        const double t2 = p * fma(1.7320508075688772, s, c);
        const double e_max = neg ? q + t2 : q + t1;
        const double e_min = neg ? q - t1 : q - t2;
        const double e_mid = 3.0 * q - e_max - e_min;
        ea = (float)e_min; eb = (float)e_mid; ec = (float)e_max;
    } else {
        ea = eb = ec = (float)q;          // p2 == 0 (multiple of the identity) or NaN
    }
    sort3_abs(ea, eb, ec, l1, l2, l3);
}

__device__ __forceinline__ float frangi3(float l1, float l2, float l3, float alpha_sq, float beta_sq, float gamma_sq) {
    const float al2 = fabsf(l2), al3 = fabsf(l3);
    const float ra = al2 / (al3 + 1e-12f);
    const float ra_sq = ra * ra;
    const float rb = al2 / (sqrtf(fabsf(l2 * l3)) + 1e-12f);
    const float rb_sq = rb * rb;
    const float s_sq = l1 * l1 + l2 * l2 + l3 * l3;
    const float A = 1.0f - expf(-(ra_sq / alpha_sq));
    const float B = expf(-(rb_sq / beta_sq));
    const float C = 1.0f - expf(-(s_sq / gamma_sq));
    float v = A * B * C;
    if (l3 > 0.0f) v = 0.0f;
    if (l2 > 0.0f) v = 0.0f;
    if (!(fabsf(v) <= 3.402823466e38f)) v = 0.0f;   // nan_to_num(nan=0, posinf=0, neginf=0)
    return v;
}

// debug / known-answer kernel: eigenvalues (sorted by |.|) and Frangi response of explicit Hessians
__global__ void __launch_bounds__(256)
debug_eig_kernel(const float *__restrict__ h6, i64 n, int impl, float alpha_sq, float beta_sq, float gamma_sq,
                 float *__restrict__ out4) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float h[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) h[k] = h6[i * 6 + k];
    float l1, l2, l3;
    if (impl == 0) eig3_sorted_abs(h, l1, l2, l3); else eig3_sorted_abs_libm(h, l1, l2, l3);
    out4[i * 4 + 0] = l1; out4[i * 4 + 1] = l2; out4[i * 4 + 2] = l3;
    out4[i * 4 + 3] = frangi3(l1, l2, l3, alpha_sq, beta_sq, gamma_sq);
}

// =================================================================================================
// kernels: dtype conversion
// =================================================================================================
template <typename T>
__global__ void convert_kernel(const T *__restrict__ src, float *__restrict__ dst, i64 n) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (float)src[i];
}

template <typename T>
__global__ void intensity_mask_kernel(const T *__restrict__ orig, float *__restrict__ frangi, double thresh, i64 n) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        // frangi * mask with mask bool: False -> +0.0 * x (keeps numpy's signed zero / nan semantics simple: x finite >= 0)
        if (!((double)orig[i] > thresh)) frangi[i] = frangi[i] * 0.0f;
    }
}

__global__ void fill_f32_kernel(float *p, float v, i64 n) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// =================================================================================================
// kernels: separable Gaussian (scipy correlate1d, symmetric branch, float64 accumulate)
//   tmp = in[0]*w[r];  for j = r..1: tmp += (in[-j] + in[+j]) * w[r-j];  out = (float)tmp
// One thread per output voxel; lanes run along X so every tap is a coalesced row read.
// =================================================================================================
#define NL_MAX_RADIUS 63
struct GaussW { double w[NL_MAX_RADIUS + 1]; int r; };   // w[k] = weight at distance k from the centre

template <int AXIS>
__global__ void __launch_bounds__(256)
gauss_axis_kernel(const float *__restrict__ in, float *__restrict__ out, VolGeom v, i64 z0, i64 z1, GaussW gw) {
    const i64 x = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 y = blockIdx.y;
    const i64 z = z0 + blockIdx.z;
    if (x >= v.nx) return;
    const i64 sy = v.nx, sz = v.ny * v.nx;
    const i64 c = z * sz + y * sy + x;
    double tmp = (double)in[c] * gw.w[0];
    for (int j = gw.r; j >= 1; --j) {
        i64 il, ih;
        if (AXIS == 0) {
            il = (reflect_idx(v.gz0 + z - j, v.gnz) - v.gz0) * sz + y * sy + x;
            ih = (reflect_idx(v.gz0 + z + j, v.gnz) - v.gz0) * sz + y * sy + x;
        } else if (AXIS == 1) {
            il = z * sz + reflect_idx(y - j, v.ny) * sy + x;
            ih = z * sz + reflect_idx(y + j, v.ny) * sy + x;
        } else {
            il = z * sz + y * sy + reflect_idx(x - j, v.nx);
            ih = z * sz + y * sy + reflect_idx(x + j, v.nx);
        }
        const double s = (double)in[il] + (double)in[ih];
        tmp = tmp + s * gw.w[j];
    }
    out[c] = (float)tmp;
}

// -------------------------------------------------------------------------------------------------
// v2 Gaussian passes.  Same arithmetic (scipy's order, float64), every input read from HBM once:
//  * Z and Y passes: a thread owns one (.., x) column position and MARCHES along the filtered axis
//    with a sliding window of 2R+1 float64 values in registers (static indexing through a fully
//    unrolled phase loop); lanes run along X so every load/store is a coalesced 256-B row piece.
//  * X pass: a workgroup stages a row piece (+-R halo, reflected) in LDS, each thread then computes
//    4 consecutive outputs from 4+2R staged values and stores them as one float4.
// Radii above GM_MAX_R fall back to the one-thread-per-voxel kernel above.
// -------------------------------------------------------------------------------------------------
#define GM_MAX_R 8
#define GM_CHUNK 128
struct GaussWS { double w[GM_MAX_R + 1]; };

// AXIS 0: line = Z (grid over y-groups, x-tiles, z-chunks); AXIS 1: line = Y (grid over z-groups, x-tiles, y-chunks)
template <int AXIS, int R>
__global__ void __launch_bounds__(256)
gauss_march_kernel(const float *__restrict__ in, float *__restrict__ out, VolGeom v, i64 z0, i64 z1, GaussWS gw) {
    constexpr int W = 2 * R + 1;
    const int lx = threadIdx.x & 63, lo = threadIdx.x >> 6;      // 64 x-positions x 4 "other" positions
    const i64 x = (i64)blockIdx.x * 64 + lx;
    const i64 sy = v.nx, sz = v.ny * v.nx;
    i64 other, c0, c1, n_line, line_stride, base;
    if (AXIS == 0) {
        other = (i64)blockIdx.y * 4 + lo;                         // y
        c0 = z0 + (i64)blockIdx.z * GM_CHUNK;
        c1 = c0 + GM_CHUNK < z1 ? c0 + GM_CHUNK : z1;
        if (other >= v.ny || x >= v.nx) return;
        line_stride = sz; base = other * sy + x;
    } else {
        other = z0 + (i64)blockIdx.y * 4 + lo;                    // z
        c0 = (i64)blockIdx.z * GM_CHUNK;
        c1 = c0 + GM_CHUNK < v.ny ? c0 + GM_CHUNK : v.ny;
        if (other >= z1 || x >= v.nx) return;
        line_stride = sy; base = other * sz + x;
    }
    n_line = (AXIS == 0) ? v.gnz : v.ny;
    const i64 goff = (AXIS == 0) ? v.gz0 : 0;                     // line coordinate of local index 0
    auto ld = [&](i64 p) -> double {                               // p = local line index, may be outside
        const i64 q = reflect_idx(goff + p, n_line) - goff;
        return (double)in[base + q * line_stride];
    };
    double win[W];
#pragma unroll
    for (int k = 0; k < W - 1; ++k) win[k + 1] = ld(c0 - R + k);   // slots 1..W-1 hold c0-R .. c0+R-1
    for (i64 p0 = c0; p0 < c1; p0 += W) {
#pragma unroll
        for (int ph = 0; ph < W; ++ph) {
            const i64 p = p0 + ph;
            if (p < c1) {
                // the newest value (p+R) replaces the oldest slot, which is slot `ph`
                win[ph] = ld(p + R);
                // window position t (0..2R) = line index p-R+t lives in slot (ph + 1 + t) % W
                double tmp = win[(ph + 1 + R) % W] * gw.w[0];
#pragma unroll
                for (int j = R; j >= 1; --j) {
                    const double s = win[(ph + 1 + R - j) % W] + win[(ph + 1 + R + j) % W];
                    tmp = tmp + s * gw.w[j];
                }
                out[base + p * line_stride] = (float)tmp;
            }
        }
    }
}

// Fused Y + X pass (the Y and X radii of a cascade step are always equal: sigma_vec = (s/z_ratio, s, s)).
// A 320-thread workgroup owns 256 output columns of one Z plane and walks a chunk of rows: every thread marches
// down ITS column (256 outputs + R reflected halo columns on each side) with the float64 sliding window of the Y
// pass, rounds to float32 exactly like the stand-alone pass, and drops the value into a double-buffered LDS row;
// after one barrier the 256 output threads take the X pass from that row.  The intermediate volume between the Y
// and the X pass never exists in HBM: 8 B/voxel instead of 16.
#define GYX_COLS 256
#define GYX_THREADS 320
template <int R>
__global__ void __launch_bounds__(GYX_THREADS)
gauss_yx_kernel(const float *__restrict__ in, float *__restrict__ out, VolGeom v, i64 z0, i64 z1, GaussWS gwy, GaussWS gwx) {
    constexpr int W = 2 * R + 1;
    __shared__ float rowbuf[2][GYX_COLS + 2 * GM_MAX_R];
    const int pos = threadIdx.x;                                  // position in the staged row
    const i64 x0 = (i64)blockIdx.x * GYX_COLS;
    const i64 z = z0 + blockIdx.z;
    const i64 c0 = (i64)blockIdx.y * GM_CHUNK;
    const i64 c1 = c0 + GM_CHUNK < v.ny ? c0 + GM_CHUNK : v.ny;
    const bool active = pos < GYX_COLS + 2 * R;
    const i64 xcol = reflect_idx(x0 - R + pos, v.nx);             // the column this thread filters along Y
    const i64 xout = x0 + pos - R;                                // the output column of an output thread
    const bool writer = pos >= R && pos < R + GYX_COLS && xout < v.nx;
    const i64 sy = v.nx;
    const i64 base = z * v.ny * v.nx + xcol;
    auto ld = [&](i64 p) -> double { return (double)in[base + reflect_idx(p, v.ny) * sy]; };
    double win[W];
    if (active) {
#pragma unroll
        for (int k = 0; k < W - 1; ++k) win[k + 1] = ld(c0 - R + k);
    }
    for (i64 p0 = c0; p0 < c1; p0 += W) {
#pragma unroll
        for (int ph = 0; ph < W; ++ph) {
            const i64 p = p0 + ph;
            if (p < c1) {                                          // uniform
                const int buf = (int)(p & 1);
                if (active) {
                    win[ph] = ld(p + R);
                    double tmp = win[(ph + 1 + R) % W] * gwy.w[0];
#pragma unroll
                    for (int j = R; j >= 1; --j) {
                        const double s = win[(ph + 1 + R - j) % W] + win[(ph + 1 + R + j) % W];
                        tmp = tmp + s * gwy.w[j];
                    }
                    rowbuf[buf][pos] = (float)tmp;                 // the float32 store between the two passes
                }
                __syncthreads();
                if (writer) {
                    double tmp = (double)rowbuf[buf][pos] * gwx.w[0];
#pragma unroll
                    for (int j = R; j >= 1; --j) {
                        const double s = (double)rowbuf[buf][pos - j] + (double)rowbuf[buf][pos + j];
                        tmp = tmp + s * gwx.w[j];
                    }
                    out[(z * v.ny + p) * v.nx + xout] = (float)tmp;
                }
            }
        }
    }
}

#define GX_SEG 1024
template <int R>
__global__ void __launch_bounds__(256)
gauss_x_kernel(const float *__restrict__ in, float *__restrict__ out, VolGeom v, i64 z0, i64 z1, GaussWS gw, int vec4) {
    __shared__ float row[GX_SEG + 2 * GM_MAX_R];
    const i64 y = blockIdx.y, z = z0 + blockIdx.z;
    const i64 xs = (i64)blockIdx.x * GX_SEG;                       // first output of this segment
    const i64 rb = (z * v.ny + y) * v.nx;
    const int tid = threadIdx.x;
    const i64 seg = (v.nx - xs < GX_SEG) ? v.nx - xs : GX_SEG;
    for (int e = tid; e < seg + 2 * R; e += 256) row[e] = in[rb + reflect_idx(xs - R + e, v.nx)];
    __syncthreads();
    const int o = tid * 4;
    if (o >= seg) return;
    double d[4 + 2 * R];
#pragma unroll
    for (int k = 0; k < 4 + 2 * R; ++k) d[k] = (o + k < seg + 2 * R) ? (double)row[o + k] : 0.0;
    float res[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double tmp = d[q + R] * gw.w[0];
#pragma unroll
        for (int j = R; j >= 1; --j) tmp = tmp + (d[q + R - j] + d[q + R + j]) * gw.w[j];
        res[q] = (float)tmp;
    }
    float *dst = out + rb + xs + o;
    if (vec4 && o + 3 < seg) {
        *reinterpret_cast<float4 *>(dst) = make_float4(res[0], res[1], res[2], res[3]);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (o + q < seg) dst[q] = res[q];
    }
}

// =================================================================================================
// kernels: lattice sampling (strided subsample for the thresholds)
// =================================================================================================
struct Lattice {
    i64 sz, sy, sx;      // strides (global lattice)
    i64 cz, cy, cx;      // lattice extent on the owned planes
    i64 zfirst;          // local z of the first owned lattice plane
};

struct FieldSrc {
    const float *p;      // gauss (GAUSS/FROB) or frangi
    int field;
    HessP hp;
    float max_abs, max_finite;
};

__device__ __forceinline__ float field_value(const FieldSrc &fs, const VolGeom &v, i64 z, i64 y, i64 x) {
    if (fs.field == NL_FIELD_FROB) {
        float h[6];
        hessian_at(fs.p, v, fs.hp, z, y, x, h);
        return frob_norm(frob_sq_of(h), fs.max_abs, fs.max_finite);
    }
    return fs.p[(z * v.ny + y) * v.nx + x];
}

__global__ void __launch_bounds__(256)
sample_gather_kernel(FieldSrc fs, VolGeom v, Lattice L, float *__restrict__ out) {
    const i64 total = L.cz * L.cy * L.cx;
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const i64 ix = i % L.cx, iy = (i / L.cx) % L.cy, iz = i / (L.cx * L.cy);
    out[i] = field_value(fs, v, L.zfirst + iz * L.sz, iy * L.sy, ix * L.sx);
}

// results: [0] = min bits (uint), [1] = max bits (uint), [2..3] = count (u64).  Grid-stride, one set of
// atomics per workgroup (per-wave atomics on three shared words cost ~0.5 ms at 1e6 samples).
__global__ void __launch_bounds__(256)
sample_minmax_kernel(FieldSrc fs, VolGeom v, Lattice L, unsigned int *__restrict__ res) {
    __shared__ float s_mn[4], s_mx[4];
    __shared__ unsigned long long s_c[4];
    const i64 total = L.cz * L.cy * L.cx;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    float mn = __int_as_float(0x7f800000), mx = 0.0f;
    unsigned long long cnt = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const i64 ix = i % L.cx, iy = (i / L.cx) % L.cy, iz = i / (L.cx * L.cy);
        const float val = field_value(fs, v, L.zfirst + iz * L.sz, iy * L.sy, ix * L.sx);
        if (val > 0.0f) { mn = fminf(mn, val); mx = fmaxf(mx, val); cnt++; }
    }
    mn = wave_min_f(mn); mx = wave_max_f(mx); cnt = wave_sum_u64(cnt);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_mn[w] = mn; s_mx[w] = mx; s_c[w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { mn = fminf(mn, s_mn[k]); mx = fmaxf(mx, s_mx[k]); cnt += s_c[k]; }
        if (cnt) {
            // positive floats order like their bit patterns
            atomicMin(&res[0], __float_as_uint(mn));
            atomicMax(&res[1], __float_as_uint(mx));
            atomicAdd((unsigned long long *)(res + 2), cnt);
        }
    }
}

// numpy histogram fast path, float32 (numpy/lib/_histograms_impl.py): see include/nellie_amd.h
__global__ void __launch_bounds__(256)
sample_hist_kernel(FieldSrc fs, VolGeom v, Lattice L, const float *__restrict__ edges, int nbins,
                   unsigned long long *__restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_edges = (float *)smem;                          // nbins+1
    unsigned int *s_cnt = (unsigned int *)(s_edges + nbins + 1 + ((nbins + 1) & 1));   // nbins
    for (int k = threadIdx.x; k <= nbins; k += blockDim.x) s_edges[k] = edges[k];
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) s_cnt[k] = 0;
    __syncthreads();
    const float first = s_edges[0], last = s_edges[nbins];
    const float denom = last - first;
    const i64 total = L.cz * L.cy * L.cx;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const i64 ix = i % L.cx, iy = (i / L.cx) % L.cy, iz = i / (L.cx * L.cy);
        const float a = field_value(fs, v, L.zfirst + iz * L.sz, iy * L.sy, ix * L.sx);
        if (a > 0.0f && a >= first && a <= last) {
            const float fi = ((a - first) / denom) * (float)nbins;
            int idx = (int)fi;
            if (idx == nbins) idx -= 1;
            if (a < s_edges[idx]) idx -= 1;
            if (a >= s_edges[idx + 1] && idx != nbins - 1) idx += 1;
            atomicAdd(&s_cnt[idx], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nbins; k += blockDim.x)
        if (s_cnt[k]) atomicAdd(&counts[k], (unsigned long long)s_cnt[k]);
}

__global__ void __launch_bounds__(256)
flat_gather_kernel(const float *__restrict__ p, i64 base, i64 offset, i64 step, i64 count, float *__restrict__ out) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = p[base + offset + i * step];
}

// =================================================================================================
// kernels: Hessian statistics and the per-scale vesselness update
// =================================================================================================
// res: [0] max|h| bits, [1] max finite frob_sq bits, [2] any_inf
__global__ void __launch_bounds__(256)
hessian_stats_kernel(const float *__restrict__ g, VolGeom v, HessP hp, i64 z0, i64 z1, unsigned int *__restrict__ res) {
    const i64 x = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 y = blockIdx.y;
    const i64 z = z0 + blockIdx.z;
    float mabs = 0.0f, mfrob = 0.0f;
    int inf = 0;
    if (x < v.nx) {
        float h[6];
        hessian_at(g, v, hp, z, y, x, h);
#pragma unroll
        for (int k = 0; k < 6; ++k) mabs = fmaxf(mabs, fabsf(h[k]));
        const float fsq = frob_sq_of(h);
        if (isinf(fsq)) inf = 1; else mfrob = fmaxf(mfrob, fsq);   // fmaxf drops NaN
    }
    mabs = wave_max_f(mabs); mfrob = wave_max_f(mfrob); inf = wave_or_i(inf);
    if ((threadIdx.x & 63) == 0) {
        if (mabs > 0.0f) atomicMax(&res[0], __float_as_uint(mabs));
        if (mfrob > 0.0f) atomicMax(&res[1], __float_as_uint(mfrob));
        if (inf) atomicOr(&res[2], 1u);
    }
}

struct VessP {
    float gamma_sq, alpha_sq, beta_sq;
    int use_thr;
    float thr;
    float max_abs, max_finite;
    int mask_rmw;            // 1: AND into the slot (more scales than mask slots)
    int cnt_lo, cnt_hi;      // planes whose masked voxels are counted (the owned ones)
    int first;               // 1: first evaluated scale of the frame -> vesselness is written, not max-ed (no memset needed)
};

__global__ void __launch_bounds__(256)
vesselness_kernel(const float *__restrict__ g, float *__restrict__ vmax, uint8_t *__restrict__ cmask,
                  VolGeom v, HessP hp, VessP vp, i64 z0, i64 z1, unsigned long long *__restrict__ mask_count) {
    const i64 x = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 y = blockIdx.y;
    const i64 z = z0 + blockIdx.z;
    unsigned long long cnt = 0;
    if (x < v.nx) {
        const i64 c = (z * v.ny + y) * v.nx + x;
        float h[6];
        hessian_at(g, v, hp, z, y, x, h);
        const float fr = frob_norm(frob_sq_of(h), vp.max_abs, vp.max_finite);
        const bool m = vp.use_thr ? (fr > vp.thr) : (fr > 0.0f);
        if (m) {
            float l1, l2, l3;
            eig3_sorted_abs(h, l1, l2, l3);
            const float val = frangi3(l1, l2, l3, vp.alpha_sq, vp.beta_sq, vp.gamma_sq);
            const float old = vmax[c];
            if (val > old) vmax[c] = val;      // np.maximum(vesselness, vessel_scale); both finite, >= 0
            cnt = 1;
        } else {
            cmask[c] = 0;                      // masks &= h_mask
        }
    }
    cnt = wave_sum_u64(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(mask_count, cnt);
}

// -------------------------------------------------------------------------------------------------
// v3: Z-marching, LDS-tiled Hessian kernel (MODE 0 = statistics, MODE 1 = vesselness update).
// A 1024-thread workgroup owns a 16(y) x 64(x) column tile and walks a chunk of Z planes.  Each
// plane tile (+-2 halo, indices clamped at the volume faces so the padding replicates the edge
// voxels) is read from HBM once into an 8-slot LDS ring; the next plane is fetched into registers
// while the current one is computed (one barrier per plane).  All 24 stencil taps come from LDS.
// Edge rules: with replicated padding f(clamp(q-1)) is simply tile[q-1]; only the divisor
// (float32(h) at a face, float32(2h) inside) and the outer difference sites need selects.
// Divisions by the six constant spacings go through div_c (exactly rounded, see above).
// MODE 1 only evaluates eigenvalues where the Frobenius mask holds (15-25 % of the voxels, scattered):
// masked voxels are COMPACTED into an LDS queue (6 Hessian components + voxel index) and the
// float64 eigen-solve + Frangi response runs on dense batches of 1024 queue entries.
// -------------------------------------------------------------------------------------------------
#define HM_TX 64
#define HM_PW (HM_TX + 4)
#define HM_SLOTS 8
#define HM_RSLOT(zz) ((int)(((unsigned)((zz) - zc0 + HM_SLOTS)) & (HM_SLOTS - 1)))   // ring slot relative to the chunk start
#define HM_ZCHUNK 64
#define HM_DEPTH 3
template <int TY> struct HMCfg {
    static constexpr int NT = HM_TX * TY;               // threads per workgroup
    static constexpr int PH = TY + 4;
    static constexpr int PLANE = HM_PW * PH;
    static constexpr int QCAP = 2 * NT;                 // queue ring: < NT waiting + <= NT appended per plane
    static constexpr int lds_floats(int mode) { return HM_SLOTS * PLANE + 64 + (mode == 1 ? 7 * QCAP : 0); }
};

// A constant float32 divisor, two exact implementations of a/d:
//  FAST : q = a*y; r = fma(-q,d,a); q' = fma(r,y,q) with y = RN(1/d) -- three float32 instructions.  This
//         sequence is correctly rounded for most but not all d, so nl_hessian_stats PROVES it for the
//         six divisors in use by exhaustion (divcheck_kernel: all 2^23 significands; scaling a by 2^k
//         scales q, r, q' exactly, so one binade covers every normal a) before selecting this path.
//  !FAST: (float)((double)a * RN64(1/d)), exact for every d (see div_c).
template <bool FAST> struct Dv;
template <> struct Dv<true> {
    float d, y;
    __device__ __forceinline__ float div(float a) const { const float q = a * y; const float r = fmaf(-q, d, a); return fmaf(r, y, q); }
};
template <> struct Dv<false> {
    double r;
    __device__ __forceinline__ float div(float a) const { return (float)((double)a * r); }
};
template <bool FAST> struct HessDv { Dv<FAST> z, y, x, z2, y2, x2; };   // float32(h) and float32(2h) per axis

__global__ void __launch_bounds__(256)
divcheck_kernel(float d, float y, unsigned int *__restrict__ bad) {
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;               // 2^23 significands of [1,2)
    const float a = __uint_as_float(0x3f800000u | i);
    Dv<true> dv{d, y};
    if (dv.div(a) != a / d) atomicOr(bad, 1u);
}

template <int MODE, int TY, bool FAST>
__global__ void __launch_bounds__(HM_TX * TY)
hessian_march_kernel(const float *__restrict__ g, float *__restrict__ vmax, unsigned long long *__restrict__ cmask64, int wpr,
                     VolGeom v, HessDv<FAST> hr, VessP vp, int z0, int z1, int ntx, int nty,
                     unsigned int *__restrict__ res, unsigned long long *__restrict__ mask_count) {
    constexpr int NT = HMCfg<TY>::NT, HM_PLANE = HMCfg<TY>::PLANE, HM_QCAP = HMCfg<TY>::QCAP, HM_TY = TY;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *sp = (float *)smem;                               // [HM_SLOTS][HM_PLANE]
    float *s_red = sp + HM_SLOTS * HM_PLANE;               // 64 floats of reduction scratch
    float *q_h = s_red + 64;                                 // [6][HM_QCAP]   (MODE 1)
    int *q_i = (int *)(q_h + 6 * HM_QCAP);                   // [HM_QCAP]      (MODE 1)
    int *s_tail = (int *)s_red + 48;                         // queue tail     (MODE 1)
    const int tid = threadIdx.x;
    const int lx = tid & 63, ly = tid >> 6;
    // XCD-aware remap: workgroup b runs on XCD b % 8; give each XCD a contiguous run of tiles so that
    // neighbouring tiles (shared halos) meet in the same L2.  Speed only, never correctness.
    const unsigned nblk = gridDim.x;
    unsigned bid = blockIdx.x;
    if ((nblk & 7u) == 0u) bid = (bid & 7u) * (nblk >> 3) + (bid >> 3);
    const int tx = bid % ntx;
    const int ty = (bid / ntx) % nty;
    const int zc = bid / (ntx * nty);
    const int zc0 = z0 + zc * HM_ZCHUNK;
    const int zc1 = (zc0 + HM_ZCHUNK < z1) ? zc0 + HM_ZCHUNK : z1;
    const int nx = (int)v.nx, ny = (int)v.ny;
    const i64 sz = v.ny * v.nx;
    const int gz0 = (int)v.gz0, gnz = (int)v.gnz;
    const int xbase = tx * HM_TX - 2, ybase = ty * HM_TY - 2;
    const int x = xbase + 2 + lx, y = ybase + 2 + ly;
    const bool valid = (x < nx) && (y < ny);

    // the (up to) two tile elements this thread stages per plane (in-plane offsets fit 32 bits)
    int off0, off1 = -1;
    {
        int yy = ybase + tid / HM_PW, xx = xbase + tid % HM_PW;
        yy = yy < 0 ? 0 : (yy > ny - 1 ? ny - 1 : yy);
        xx = xx < 0 ? 0 : (xx > nx - 1 ? nx - 1 : xx);
        off0 = yy * nx + xx;
        const int e1 = tid + NT;
        if (e1 < HM_PLANE) {
            int y1 = ybase + e1 / HM_PW, x1 = xbase + e1 % HM_PW;
            y1 = y1 < 0 ? 0 : (y1 > ny - 1 ? ny - 1 : y1);
            x1 = x1 < 0 ? 0 : (x1 > nx - 1 ? nx - 1 : x1);
            off1 = y1 * nx + x1;
        }
    }
    // clamp a local plane index to the GLOBAL volume
    auto zclamp = [&](int zz) -> int {
        const int gg = gz0 + zz;
        return gg < 0 ? -gz0 : (gg > gnz - 1 ? gnz - 1 - gz0 : zz);
    };

    if (MODE == 1 && tid == 0) *s_tail = 0;
    // Planes this chunk ever touches form the contiguous range [pmin, pmax].  The first five go straight
    // to LDS; the following HM_DEPTH planes are put in flight into registers (memory-level parallelism:
    // a 3.2 KB plane per workgroup is far too little to cover HBM latency on its own).
    const int pmin = zclamp(zc0 - 2), pmax = zclamp(zc1 + 1);
    for (int pz = pmin; pz <= pmax && pz <= zc0 + 2; ++pz) {
        float *dst = sp + HM_RSLOT(pz) * HM_PLANE;
        const float *src = g + (i64)pz * sz;
        dst[tid] = src[off0];
        if (off1 >= 0) dst[tid + NT] = src[off1];
    }
    float ra[HM_DEPTH], rb[HM_DEPTH];
#pragma unroll
    for (int d = 0; d < HM_DEPTH; ++d) {
        ra[d] = 0.0f; rb[d] = 0.0f;
        const int pz = zc0 + 3 + d;
        if (pz <= pmax) {
            const float *src = g + (i64)pz * sz;
            ra[d] = src[off0];
            if (off1 >= 0) rb[d] = src[off1];
        }
    }
    __syncthreads();

    // per-lane in-plane geometry (constant along Z)
    const bool y_lo = (y == 0), y_hi = (y == ny - 1), x_lo = (x == 0), x_hi = (x == nx - 1);
    const int jc = ly + 2, ic = lx + 2;
    const int jl = jc - (y_lo ? 0 : 1), jh = jc + (y_hi ? 0 : 1);
    const int il = ic - (x_lo ? 0 : 1), ih = ic + (x_hi ? 0 : 1);
    const Dv<FAST> rdy = (y_lo || y_hi) ? hr.y : hr.y2;
    const Dv<FAST> rdx = (x_lo || x_hi) ? hr.x : hr.x2;
    // reciprocal divisor of the first derivative taken AT row j / column i of the tile
    auto rdy_at = [&](int j) -> Dv<FAST> { const int yy = ybase + j; return (yy == 0 || yy == ny - 1) ? hr.y : hr.y2; };
    auto rdx_at = [&](int i) -> Dv<FAST> { const int xx = xbase + i; return (xx == 0 || xx == nx - 1) ? hr.x : hr.x2; };
    const Dv<FAST> rdy_jl = rdy_at(jl), rdy_jh = rdy_at(jh), rdy_jc = rdy_at(jc);
    const Dv<FAST> rdx_il = rdx_at(il), rdx_ih = rdx_at(ih);
    const int o_cc = jc * HM_PW + ic;
    const int o_hc = jh * HM_PW + ic, o_lc = jl * HM_PW + ic;     // rows jh / jl, column ic
    const int o_ch = jc * HM_PW + ih, o_cl = jc * HM_PW + il;     // row jc, columns ih / il

    float mabs = 0.0f, mfrob = 0.0f;
    int anyinf = 0;
    unsigned long long cnt = 0;
    int head = 0;

    auto process_entry = [&](int e) {
        const int slot = e & (HM_QCAP - 1);
        float h[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) h[k] = q_h[k * HM_QCAP + slot];
        const int c = q_i[slot];
        float l1, l2, l3;
        eig3_sorted_abs(h, l1, l2, l3);
        const float val = frangi3(l1, l2, l3, vp.alpha_sq, vp.beta_sq, vp.gamma_sq);
        if (vp.first || val > vmax[c]) vmax[c] = val;
    };

    // Hessian of the voxel from five plane tiles (float32, numpy's rounding points)
    auto compute_h = [&](const float *Pm2, const float *Pm1, const float *P0, const float *Pp1, const float *Pp2,
                         const bool z_lo, const bool z_hi, const Dv<FAST> rdz, const Dv<FAST> rdz_m1, const Dv<FAST> rdz_p1,
                         float h[6]) {
        const Dv<FAST> rdz_0 = rdz;
        // h_zz: outer sites are z+1 (or z at the top face) and z-1 (or z at the bottom face); the first derivatives
        // along Z at planes z-1, z, z+1 use planes (z-2,z), (z-1,z+1), (z,z+2)
        const float gz_hi = z_hi ? rdz_0.div(P0[o_cc] - Pm1[o_cc]) : rdz_p1.div(Pp2[o_cc] - P0[o_cc]);
        const float gz_lo = z_lo ? rdz_0.div(Pp1[o_cc] - P0[o_cc]) : rdz_m1.div(P0[o_cc] - Pm2[o_cc]);
        h[0] = rdz.div(gz_hi - gz_lo);
        h[1] = rdy.div(rdz_0.div(Pp1[o_hc] - Pm1[o_hc]) - rdz_0.div(Pp1[o_lc] - Pm1[o_lc]));
        h[2] = rdx.div(rdz_0.div(Pp1[o_ch] - Pm1[o_ch]) - rdz_0.div(Pp1[o_cl] - Pm1[o_cl]));
        h[3] = rdy.div(rdy_jh.div(P0[o_hc + HM_PW] - P0[o_hc - HM_PW]) - rdy_jl.div(P0[o_lc + HM_PW] - P0[o_lc - HM_PW]));
        h[4] = rdx.div(rdy_jc.div(P0[o_ch + HM_PW] - P0[o_ch - HM_PW]) - rdy_jc.div(P0[o_cl + HM_PW] - P0[o_cl - HM_PW]));
        h[5] = rdx.div(rdx_ih.div(P0[o_ch + 1] - P0[o_ch - 1]) - rdx_il.div(P0[o_cl + 1] - P0[o_cl - 1]));
    };

    // One plane step.  U >= 0: the ring slot of plane z is the compile-time constant U (the Z loop is unrolled by
    // HM_SLOTS and slots are taken relative to the chunk start), the planes z-2..z+2 all exist (interior), so every
    // LDS address is a per-lane base + immediate and no select is needed.  U < 0: generic step (faces, loop tail).
    auto step = [&](const int z, auto uc) {
        constexpr int U = decltype(uc)::value;
        if (MODE == 1) {
            // dense eigen batch once the queue holds a full workgroup's worth
            const int tail = *(volatile int *)s_tail;
            if (tail - head >= NT) {              // uniform
                process_entry(head + tid);
                head += NT;
                __syncthreads();                  // entries read before their ring slots can be re-used
            }
        }
        bool m = false;
        float h[6];
        if (valid) {
            if (U >= 0) {
                compute_h(sp + ((U + 6) & 7) * HM_PLANE, sp + ((U + 7) & 7) * HM_PLANE, sp + (U & 7) * HM_PLANE,
                          sp + ((U + 1) & 7) * HM_PLANE, sp + ((U + 2) & 7) * HM_PLANE, false, false, hr.z2, hr.z2, hr.z2, h);
            } else {
                const int gz = gz0 + z;
                const bool z_lo = (gz == 0), z_hi = (gz == gnz - 1);
                compute_h(sp + HM_RSLOT(zclamp(z - 2)) * HM_PLANE, sp + HM_RSLOT(zclamp(z - 1)) * HM_PLANE,
                          sp + HM_RSLOT(z) * HM_PLANE, sp + HM_RSLOT(zclamp(z + 1)) * HM_PLANE,
                          sp + HM_RSLOT(zclamp(z + 2)) * HM_PLANE, z_lo, z_hi, (z_lo || z_hi) ? hr.z : hr.z2,
                          (gz - 1 == 0) ? hr.z : hr.z2, (gz + 1 == gnz - 1) ? hr.z : hr.z2, h);
            }
            const float fsq = frob_sq_of(h);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) mabs = fmaxf(mabs, fabsf(h[k]));
                if (isinf(fsq)) anyinf = 1; else mfrob = fmaxf(mfrob, fsq);
            } else {
                const float fr = frob_norm(fsq, vp.max_abs, vp.max_finite);
                m = vp.use_thr ? (fr > vp.thr) : (fr > 0.0f);
            }
        }
        if (MODE == 1) {
            // append the masked voxels of this wave to the queue (one LDS atomic per wave)
            const unsigned long long bal = __ballot(m);
            const int lane = tid & 63;
            // h_mask of this scale as a bit mask: this wave owns exactly one 64-voxel word of the row.  Every
            // scale writes its own slot (no read-modify-write on the critical path); nl_filter_finish ANDs them.
            if (lane == 0 && y < ny) {
                unsigned long long *wp = cmask64 + ((i64)z * ny + y) * wpr + tx;
                *wp = vp.mask_rmw ? (*wp & bal) : bal;
            }
            if (bal) {
                const int leader = __builtin_ctzll(bal);
                int base = 0;
                if (lane == leader) base = atomicAdd(s_tail, (int)__builtin_popcountll(bal));
                base = __shfl(base, leader, 64);
                if (m) {
                    const int rank = (int)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                    const int slot = (base + rank) & (HM_QCAP - 1);
#pragma unroll
                    for (int k = 0; k < 6; ++k) q_h[k * HM_QCAP + slot] = h[k];
                    q_i[slot] = (int)((i64)z * sz + (i64)y * nx + x);
                    if (z >= vp.cnt_lo && z < vp.cnt_hi) cnt++;
                }
            }
        }
        // plane z+3 (in flight since HM_DEPTH steps) lands in the ring; plane z+3+HM_DEPTH takes its place in flight
        if (z + 3 <= pmax) {
            float *dst = sp + ((U >= 0) ? ((U + 3) & 7) : HM_RSLOT(z + 3)) * HM_PLANE;
            dst[tid] = ra[0];
            if (off1 >= 0) dst[tid + NT] = rb[0];
        }
#pragma unroll
        for (int d = 0; d + 1 < HM_DEPTH; ++d) { ra[d] = ra[d + 1]; rb[d] = rb[d + 1]; }
        {
            const int pz = z + 3 + HM_DEPTH;
            if (pz <= pmax) {
                const float *src = g + (i64)pz * sz;
                ra[HM_DEPTH - 1] = src[off0];
                if (off1 >= 0) rb[HM_DEPTH - 1] = src[off1];
            }
        }
        __syncthreads();
    };

    {
        int z = zc0;
        for (; z + HM_SLOTS <= zc1; z += HM_SLOTS) {
            // all eight planes interior (z-2 >= global 0, z+7+2 <= global last)?  (z - zc0) % 8 == 0 here.
            // (only the statistics kernel is unrolled: in the vesselness kernel the eight inlined copies of the
            //  eigen batch push the register count from 84 to 135 and cost more than the static slots gain)
            if (MODE == 0 && gz0 + z >= 2 && gz0 + z + HM_SLOTS - 1 <= gnz - 3) {
                step(z + 0, std::integral_constant<int, 0>{}); step(z + 1, std::integral_constant<int, 1>{});
                step(z + 2, std::integral_constant<int, 2>{}); step(z + 3, std::integral_constant<int, 3>{});
                step(z + 4, std::integral_constant<int, 4>{}); step(z + 5, std::integral_constant<int, 5>{});
                step(z + 6, std::integral_constant<int, 6>{}); step(z + 7, std::integral_constant<int, 7>{});
            } else {
                for (int u = 0; u < HM_SLOTS; ++u) step(z + u, std::integral_constant<int, -1>{});
            }
        }
        for (; z < zc1; ++z) step(z, std::integral_constant<int, -1>{});
    }
    if (MODE == 1) {
        // drain the queue
        const int tail = *(volatile int *)s_tail;
        while (tail - head > 0) {
            if (head + tid < tail) process_entry(head + tid);
            head += NT;
        }
    }

    // one set of atomics per workgroup
    const int w = tid >> 6;
    __syncthreads();
    if (MODE == 0) {
        mabs = wave_max_f(mabs); mfrob = wave_max_f(mfrob); anyinf = wave_or_i(anyinf);
        if ((tid & 63) == 0) { s_red[w] = mabs; s_red[16 + w] = mfrob; s_red[32 + w] = anyinf ? 1.0f : 0.0f; }
        __syncthreads();
        if (tid == 0) {
            float a = 0.0f, b = 0.0f, c = 0.0f;
            for (int k = 0; k < NT / 64; ++k) { a = fmaxf(a, s_red[k]); b = fmaxf(b, s_red[16 + k]); c = fmaxf(c, s_red[32 + k]); }
            if (a > 0.0f && __float_as_uint(a) > __hip_atomic_load(&res[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&res[0], __float_as_uint(a));
            if (b > 0.0f && __float_as_uint(b) > __hip_atomic_load(&res[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&res[1], __float_as_uint(b));
            if (c > 0.0f) atomicOr(&res[2], 1u);
        }
    } else {
        cnt = wave_sum_u64(cnt);
        unsigned long long *s_cnt = (unsigned long long *)s_red;
        if ((tid & 63) == 0) s_cnt[w] = cnt;
        __syncthreads();
        if (tid == 0) {
            unsigned long long t = 0;
            for (int k = 0; k < NT / 64; ++k) t += s_cnt[k];
            if (t) atomicAdd(mask_count, t);
        }
    }
}

// vesselness * masks (filtering.py:926); counts voxels > 0 on the owned planes.
// The cumulative mask is a bit mask: one 64-bit word per 64 consecutive x of a row (row pitch `wpr` words).
__global__ void __launch_bounds__(256)
finish_kernel(float *__restrict__ vmax, const unsigned long long *__restrict__ cmask64, int nslots, i64 slot_words, int wpr,
              VolGeom v, i64 z0, i64 z1, i64 cnt_lo, i64 cnt_hi, unsigned long long *__restrict__ npos) {
    // one thread = 4 consecutive x of one row (they share a mask word); rows are padded to a multiple of 4 here
    const i64 qpr = (v.nx + 3) / 4;                                    // quads per row
    const i64 total = (z1 - z0) * v.ny * qpr;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const bool vec = (v.nx & 3) == 0;
    unsigned long long cnt = 0;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const i64 row = t / qpr;
        const i64 x0 = (t % qpr) * 4;
        const i64 zloc = z0 + row / v.ny;
        const bool counted = zloc >= cnt_lo && zloc < cnt_hi;
        unsigned long long bits = ~0ull;
        for (int k = 0; k < nslots; ++k) bits &= cmask64[k * slot_words + (z0 * v.ny + row) * wpr + (x0 >> 6)];
        const unsigned int b4 = (unsigned int)(bits >> (x0 & 63)) & 0xFu;
        float *p = vmax + (z0 * v.ny + row) * v.nx + x0;
        if (vec) {
            float4 val = *reinterpret_cast<float4 *>(p);
            if (b4 != 0xFu) {
                if (!(b4 & 1u)) val.x = 0.0f;
                if (!(b4 & 2u)) val.y = 0.0f;
                if (!(b4 & 4u)) val.z = 0.0f;
                if (!(b4 & 8u)) val.w = 0.0f;
                *reinterpret_cast<float4 *>(p) = val;
            }
            if (counted) cnt += (val.x > 0.0f) + (val.y > 0.0f) + (val.z > 0.0f) + (val.w > 0.0f);
        } else {
            for (int q = 0; q < 4 && x0 + q < v.nx; ++q) {
                float val = p[q];
                if (!((b4 >> q) & 1u)) { val = 0.0f; p[q] = 0.0f; }
                if (counted && val > 0.0f) cnt++;
            }
        }
    }
    cnt = wave_sum_u64(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(npos, cnt);
}

// filtering.py:964-966: mask = f > thr; binary_opening (6-conn cross, one iteration, border_value 0); f * mask.
// Done on BIT masks: pack (rl_threshold_pack_kernel), erode, dilate (word-wide logic on 1 bit/voxel), apply.
// Planes [z0, z1) are produced from planes [z0-1, z1+1) of the input bits; neighbours outside the GLOBAL volume
// count as 0 (border_value), ghost planes of a slab are real neighbours.
template <int DILATE>
__global__ void __launch_bounds__(256)
bits_morph6_kernel(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out, VolGeom v, int wpr, i64 z0, i64 z1) {
    const i64 nw = (z1 - z0) * v.ny * wpr;
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nw) return;
    const int w = (int)(t % wpr);
    const i64 row_in_range = t / wpr;
    const i64 y = row_in_range % v.ny, z = z0 + row_in_range / v.ny;
    const i64 row = z * v.ny + y;
    const unsigned long long cur = in[row * wpr + w];
    const unsigned long long lft = (cur << 1) | ((w > 0) ? (in[row * wpr + w - 1] >> 63) : 0ull);           // x-1
    const unsigned long long rgt = (cur >> 1) | ((w + 1 < wpr) ? (in[row * wpr + w + 1] << 63) : 0ull);    // x+1
    const i64 gz = v.gz0 + z;
    const unsigned long long up = (gz > 0) ? in[(row - v.ny) * wpr + w] : 0ull;
    const unsigned long long dn = (gz < v.gnz - 1) ? in[(row + v.ny) * wpr + w] : 0ull;
    const unsigned long long no = (y > 0) ? in[(row - 1) * wpr + w] : 0ull;
    const unsigned long long so = (y < v.ny - 1) ? in[(row + 1) * wpr + w] : 0ull;
    unsigned long long r = DILATE ? (cur | lft | rgt | up | dn | no | so) : (cur & lft & rgt & up & dn & no & so);
    const int rem = (int)v.nx - w * 64;
    if (rem < 64) r &= (1ull << rem) - 1ull;          // keep the tail bits of a row at 0
    out[row * wpr + w] = r;
}

__global__ void __launch_bounds__(256)
apply_bits_kernel(const float *__restrict__ f, const unsigned long long *__restrict__ bits, float *__restrict__ out, VolGeom v,
                  int wpr, i64 z0, i64 z1) {
    const i64 qpr = (v.nx + 3) / 4;
    const i64 total = (z1 - z0) * v.ny * qpr;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const bool vec = (v.nx & 3) == 0;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const i64 row = z0 * v.ny + t / qpr;
        const i64 x0 = (t % qpr) * 4;
        const unsigned int b4 = (unsigned int)(bits[row * wpr + (x0 >> 6)] >> (x0 & 63)) & 0xFu;
        const float *p = f + row * v.nx + x0;
        float *o = out + row * v.nx + x0;
        if (vec) {
            float4 a = *reinterpret_cast<const float4 *>(p);
            // frame * mask: a False mask gives +-0 with the sign of the value, exactly numpy's float * bool
            if (!(b4 & 1u)) a.x *= 0.0f;
            if (!(b4 & 2u)) a.y *= 0.0f;
            if (!(b4 & 4u)) a.z *= 0.0f;
            if (!(b4 & 8u)) a.w *= 0.0f;
            *reinterpret_cast<float4 *>(o) = a;
        } else {
            for (int k = 0; k < 4 && x0 + k < v.nx; ++k) o[k] = ((b4 >> k) & 1u) ? p[k] : p[k] * 0.0f;
        }
    }
}

// =================================================================================================
// kernels: Label (labelling.py:467-509)
// =================================================================================================
__global__ void __launch_bounds__(256)
threshold_kernel(const float *__restrict__ f, uint8_t *__restrict__ m, int has_thr, float thr, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        m[i] = (has_thr && f[i] > thr) ? 1 : 0;
}

// ---- lock-free union-find on int32 parent array; root = minimum raster index of the set ----------
__device__ __forceinline__ int uf_load(const int *L, int i) {
    return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_find(const int *L, int i) {
    int p = uf_load(L, i);
    while (p != i) { i = p; p = uf_load(L, i); }
    return i;
}
__device__ __forceinline__ void uf_union(int *L, int a, int b) {
    while (true) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }          // a > b : hang a under b
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;                                          // someone re-rooted a meanwhile: retry from there
    }
}

// FG = 1: components of m != 0; FG = 0: components of m == 0 (background, for fill-holes)
template <int FG>
__device__ __forceinline__ bool is_set(const uint8_t *m, i64 i) { return FG ? (m[i] != 0) : (m[i] == 0); }

// init: every voxel of the set points at the start of its X-run within its 64-lane segment.
template <int FG>
__global__ void __launch_bounds__(256)
ccl_init_kernel(const uint8_t *__restrict__ m, int *__restrict__ L, i64 nx, i64 nrows) {
    // one wave per 64-voxel row segment
    const i64 segs_per_row = (nx + 63) / 64;
    const i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= nrows * segs_per_row) return;
    const i64 row = wave / segs_per_row, seg = wave % segs_per_row;
    const i64 x = seg * 64 + lane;
    const bool inb = x < nx;
    const i64 i = row * nx + x;
    const bool s = inb && is_set<FG>(m, i);
    const unsigned long long bal = __ballot(s);
    if (!inb) return;
    if (!s) { L[i] = -1; return; }
    // highest zero bit below `lane` -> run start
    const unsigned long long below = (~bal) & ((lane == 0) ? 0ull : ((~0ull) >> (64 - lane)));
    const int start = below ? (64 - __builtin_clzll(below)) : 0;
    L[i] = (int)(row * nx + seg * 64 + start);
}

// merge: unions with the raster-preceding rows, skipping connections the X-neighbour already made.
//   CONN = 26: rows (dz,dy) in {(-1,-1),(-1,0),(-1,+1),(0,-1)}, columns x-1..x+1
//   CONN = 6 : rows (-1,0) and (0,-1), column x
template <int FG, int CONN>
__global__ void __launch_bounds__(256)
ccl_merge_kernel(const uint8_t *__restrict__ m, int *__restrict__ L, i64 nz, i64 ny, i64 nx) {
    const i64 x = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 y = blockIdx.y, z = blockIdx.z;
    if (x >= nx) return;
    const i64 i = (z * ny + y) * nx + x;
    if (!is_set<FG>(m, i)) return;
    const bool left = (x > 0) && is_set<FG>(m, i - 1);
    if (left && (x & 63) == 0) uf_union(L, (int)i, (int)(i - 1));   // stitch 64-lane segments of a run
    auto row = [&](i64 zz, i64 yy) {
        if (zz < 0 || yy < 0 || yy >= ny) return;
        const i64 b = (zz * ny + yy) * nx;
        const bool c0 = is_set<FG>(m, b + x);
        if (CONN == 6) {
            if (!c0) return;
            if (left && is_set<FG>(m, b + x - 1)) return;          // same two runs already joined at x-1
            uf_union(L, (int)i, (int)(b + x));
        } else {
            const bool cm = (x > 0) && is_set<FG>(m, b + x - 1);
            const bool cp = (x + 1 < nx) && is_set<FG>(m, b + x + 1);
            if (left) {
                // x-1 (same run as us) already reached every set voxel in columns <= x of that row,
                // and column x+1 hangs off column x when that one is set
                if (cp && !c0) uf_union(L, (int)i, (int)(b + x + 1));
            } else {
                if (c0) uf_union(L, (int)i, (int)(b + x));
                else {
                    if (cm) uf_union(L, (int)i, (int)(b + x - 1));
                    if (cp) uf_union(L, (int)i, (int)(b + x + 1));
                }
            }
        }
    };
    if (CONN == 6) {
        row(z - 1, y);
        row(z, y - 1);
    } else {
        row(z - 1, y - 1);
        row(z - 1, y);
        row(z - 1, y + 1);
        row(z, y - 1);
    }
}

__global__ void __launch_bounds__(256)
ccl_flatten_kernel(int *__restrict__ L, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int p = L[i];
        if (p >= 0 && p != (int)i) {
            const int r = uf_find(L, p);
            if (r != p) L[i] = r;
        }
    }
}

// fill holes: background components touching the GLOBAL volume border stay background
__global__ void __launch_bounds__(256)
border_mark_kernel(const int *__restrict__ L, uint8_t *__restrict__ flag, VolGeom v) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 n = v.nzl * v.ny * v.nx;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int r = L[i];
        if (r < 0) continue;
        const i64 x = i % v.nx, y = (i / v.nx) % v.ny, z = i / (v.nx * v.ny);
        const i64 gz = v.gz0 + z;
        if (gz == 0 || gz == v.gnz - 1 || y == 0 || y == v.ny - 1 || x == 0 || x == v.nx - 1) flag[r] = 1;
    }
}
__global__ void __launch_bounds__(256)
clear_root_flags_kernel(const int *__restrict__ L, uint8_t *__restrict__ flag, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (L[i] == (int)i) flag[i] = 0;
}
__global__ void __launch_bounds__(256)
fill_holes_apply_kernel(const int *__restrict__ L, const uint8_t *__restrict__ flag, uint8_t *__restrict__ m, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int r = L[i];
        if (r >= 0 && !flag[r]) m[i] = 1;       // enclosed background -> foreground
    }
}

// areas: zero at roots, then one atomicAdd per X-run segment (bincount, labelling.py:495)
__global__ void __launch_bounds__(256)
zero_at_roots_kernel(const int *__restrict__ L, int *__restrict__ area, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (L[i] == (int)i) area[i] = 0;
}
__global__ void __launch_bounds__(256)
area_count_kernel(const int *__restrict__ L, int *__restrict__ area, i64 nx, i64 nrows) {
    const i64 segs_per_row = (nx + 63) / 64;
    const i64 nwaves = nrows * segs_per_row;
    const i64 wstride = ((i64)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wave < nwaves; wave += wstride) {
        const i64 row = wave / segs_per_row, seg = wave % segs_per_row;
        const i64 x = seg * 64 + lane;
        const int r = (x < nx) ? L[row * nx + x] : -1;
        const unsigned long long bal = __ballot(r >= 0);
        if (!bal) continue;                                            // wave-uniform
        // run leader = set lane whose left neighbour is not set; it carries the run length
        const bool leader = (r >= 0) && ((lane == 0) || !((bal >> (lane - 1)) & 1ull));
        int len = 0;
        if (leader) {
            const unsigned long long rest = ~(bal >> lane);            // first zero above lane ends the run
            len = rest ? __builtin_ctzll(rest) : (64 - lane);
        }
        // one atomic per DISTINCT root in the segment (runs of one object are usually neighbours)
        unsigned long long todo = __ballot(leader);
        while (todo) {
            const int first = __builtin_ctzll(todo);
            const int r0 = __shfl(r, first, 64);
            const bool mine = leader && (r == r0);
            int sum = mine ? len : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            if (lane == first) atomicAdd(&area[r0], sum);
            todo &= ~__ballot(mine);
        }
    }
}
__global__ void __launch_bounds__(256)
keep_large_kernel(const int *__restrict__ L, const int *__restrict__ area, uint8_t *__restrict__ m, int min_area, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int r = L[i];
        m[i] = (r >= 0 && area[r] >= min_area) ? 1 : 0;
    }
}

// uniform_filter(float32 mask, size=3, mode='reflect') > 0.5  ==  >= 14 of the 27 clamped neighbours
__global__ void __launch_bounds__(256)
majority_kernel(const uint8_t *__restrict__ m, uint8_t *__restrict__ out, VolGeom v) {
    const i64 x = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 y = blockIdx.y, z = blockIdx.z;
    if (x >= v.nx) return;
    int cnt = 0;
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz) {
        i64 zz = z + dz;
        const i64 gg = v.gz0 + zz;
        if (gg < 0) zz = z; else if (gg >= v.gnz) zz = z;          // reflect == clamp for a 3-window
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            i64 yy = y + dy;
            yy = yy < 0 ? 0 : (yy >= v.ny ? v.ny - 1 : yy);
            const i64 b = (zz * v.ny + yy) * v.nx;
            const i64 xm = x > 0 ? x - 1 : 0, xp = x + 1 < v.nx ? x + 1 : v.nx - 1;
            cnt += (int)m[b + xm] + (int)m[b + x] + (int)m[b + xp];
        }
    }
    out[(z * v.ny + y) * v.nx + x] = cnt >= 14 ? 1 : 0;
}

// raster renumbering of roots: ids 1..K in order of the root (= first voxel) index
#define SCAN_CHUNK 4096
__global__ void __launch_bounds__(256)
root_count_kernel(const int *__restrict__ L, i64 n, unsigned int *__restrict__ blk) {
    const i64 base = (i64)blockIdx.x * SCAN_CHUNK;
    unsigned long long cnt = 0;
    for (int k = threadIdx.x; k < SCAN_CHUNK; k += 256) {
        const i64 i = base + k;
        if (i < n && L[i] == (int)i) cnt++;
    }
    __shared__ unsigned int s[4];
    cnt = wave_sum_u64(cnt);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = (unsigned int)cnt;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
// single block exclusive scan of the per-chunk counts (nblk <= a few 1e5)
__global__ void __launch_bounds__(1024)
blk_scan_kernel(unsigned int *__restrict__ blk, i64 nblk, unsigned long long *__restrict__ total) {
    __shared__ unsigned int s_w[16];
    __shared__ unsigned int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (i64 base = 0; base < nblk; base += 1024) {
        const i64 i = base + threadIdx.x;
        const unsigned int val = (i < nblk) ? blk[i] : 0u;
        unsigned int inc = val;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t = __shfl_up(inc, o, 64);
            if ((threadIdx.x & 63) >= o) inc += t;
        }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
        __syncthreads();
        unsigned int woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += s_w[w];
        const unsigned int carry = s_carry;
        if (i < nblk) blk[i] = carry + woff + inc - val;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}
// newid[root] = rank + 1 (ordered within a chunk by a serial-in-wave ballot scan)
__global__ void __launch_bounds__(256)
root_assign_kernel(const int *__restrict__ L, i64 n, const unsigned int *__restrict__ blk, int *__restrict__ newid) {
    const i64 base = (i64)blockIdx.x * SCAN_CHUNK;
    __shared__ unsigned int s_w[4];
    __shared__ unsigned int s_run;
    if (threadIdx.x == 0) s_run = blk[blockIdx.x];
    __syncthreads();
    for (int k0 = 0; k0 < SCAN_CHUNK; k0 += 256) {
        const i64 i = base + k0 + threadIdx.x;
        const bool isroot = (i < n) && (L[i] == (int)i);
        const unsigned long long bal = __ballot(isroot);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const unsigned int before = (unsigned int)__builtin_popcountll(bal & ((lane == 0) ? 0ull : ((~0ull) >> (64 - lane))));
        if (lane == 0) s_w[w] = (unsigned int)__builtin_popcountll(bal);
        __syncthreads();
        unsigned int woff = 0;
        for (int q = 0; q < w; ++q) woff += s_w[q];
        const unsigned int run = s_run;
        if (isroot) newid[i] = (int)(run + woff + before + 1);
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256)
relabel_kernel(const int *__restrict__ L, const int *__restrict__ newid, int *__restrict__ out, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int r = L[i];
        out[i] = r >= 0 ? newid[r] : 0;
    }
}

#include "label_runs.inc"

// =================================================================================================
// host side: context, launch helpers, C-ABI
// =================================================================================================
static inline unsigned int grid1d(i64 n, int block = 256, i64 cap = 256 * 32) {
    i64 g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned int)g;
}

static VolGeom geom(const nl_ctx *c) { return VolGeom{c->nzl, c->ny, c->nx, c->gz0, c->gnz}; }
// tile height of the Hessian kernels (experiment knob; 8 -> 512-thread workgroups, 16 -> 1024)
static int hm_ty() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("NELLIE_HM_TY"); v = (e && atoi(e) == 16) ? 16 : 8; }
    return v;
}
static Dv<true> dv_fast(float d) { return Dv<true>{d, (float)(1.0 / (double)d)}; }
static Dv<false> dv_exact(float d) { return Dv<false>{1.0 / (double)d}; }
static HessDv<true> hessdv_fast(const nl_ctx *c) {
    return HessDv<true>{dv_fast(c->hz), dv_fast(c->hy), dv_fast(c->hx), dv_fast(c->hz2), dv_fast(c->hy2), dv_fast(c->hx2)};
}
static HessDv<false> hessdv_exact(const nl_ctx *c) {
    return HessDv<false>{dv_exact(c->hz), dv_exact(c->hy), dv_exact(c->hx), dv_exact(c->hz2), dv_exact(c->hy2), dv_exact(c->hx2)};
}
// Exhaustive proof that the 3-instruction division is exact for the six divisors in use.
static int check_fast_div(nl_ctx *c, char *err, size_t errlen) {
    const float ds[6] = {c->hz, c->hy, c->hx, c->hz2, c->hy2, c->hx2};
    unsigned int *bad = (unsigned int *)c->d_small + 64;
    NL_HIP(hipMemsetAsync(bad, 0, 6 * 4, c->stream));
    for (int k = 0; k < 6; ++k) {
        const Dv<true> dv = dv_fast(ds[k]);
        divcheck_kernel<<<(1u << 23) / 256, 256, 0, c->stream>>>(dv.d, dv.y, bad + k);
    }
    NL_CHECK_LAUNCH();
    unsigned int *h = (unsigned int *)c->h_small + 64;
    NL_HIP(hipMemcpyAsync(h, bad, 6 * 4, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    c->fast_div = 1;
    for (int k = 0; k < 6; ++k) {
        const bool normal = ds[k] > 1e-30f && ds[k] < 1e30f;
        if (h[k] || !normal) c->fast_div = 0;
    }
    const char *e = getenv("NELLIE_EXACT_DIV");
    if (e && atoi(e)) c->fast_div = 0;
    return NL_OK;
}
static HessP hessp(const nl_ctx *c) { return HessP{c->hz, c->hy, c->hx, c->hz2, c->hy2, c->hx2}; }

static size_t dtype_size(int dt) {
    switch (dt) {
        case NL_U8: case NL_I8: return 1;
        case NL_U16: case NL_I16: return 2;
        case NL_U32: case NL_I32: case NL_F32: return 4;
        case NL_F64: case NL_U64: case NL_I64: return 8;
    }
    return 0;
}

extern "C" const char *nl_version(void) { return NL_VERSION; }

extern "C" int nl_device_count(int *count, char *err, size_t errlen) {
    if (!count) return nl_fail(err, errlen, NL_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return nl_fail(err, errlen, NL_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return NL_OK;
}

extern "C" int nl_device_mem_info(int device, int64_t *free_bytes, int64_t *total_bytes, char *err, size_t errlen) {
    NL_HIP(hipSetDevice(device));
    size_t f = 0, t = 0;
    NL_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return NL_OK;
}

extern "C" int nl_device_name(int device, char *name, size_t namelen, char *err, size_t errlen) {
    hipDeviceProp_t p;
    NL_HIP(hipGetDeviceProperties(&p, device));
    if (name && namelen) snprintf(name, namelen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return NL_OK;
}

extern "C" int64_t nl_ctx_bytes(int64_t nz_local, int64_t ny, int64_t nx) {
    const int64_t n = nz_local * ny * nx;
    return n * (4 * 4 + 3) + (1 << 16) + ((n + SCAN_CHUNK - 1) / SCAN_CHUNK + 1) * 4;
}

extern "C" int nl_ctx_destroy(nl_ctx *c) {
    if (!c) return NL_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto &kv : c->prof) for (auto &r : kv.second) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (int k = 0; k < 4; ++k) if (c->f[k]) hipFree(c->f[k]);
    for (int k = 0; k < 3; ++k) if (c->m[k]) hipFree(c->m[k]);
    if (c->d_small) hipFree(c->d_small);
    if (c->d_input) hipFree(c->d_input);
    if (c->d_blk) hipFree(c->d_blk);
    if (c->d_rows) hipFree(c->d_rows);
    if (c->gbits[0]) hipFree(c->gbits[0]);
    if (c->gbits[1]) hipFree(c->gbits[1]);
    if (c->grows) hipFree(c->grows);
    if (c->h_small) hipHostFree(c->h_small);
    if (c->t0) hipEventDestroy(c->t0);
    if (c->t1) hipEventDestroy(c->t1);
    if (c->comm) ncclCommDestroy((ncclComm_t)c->comm);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return NL_OK;
}

extern "C" int nl_ctx_create(nl_ctx **out, int device, int64_t nzl, int64_t ny, int64_t nx,
                             int64_t gz0, int64_t gnz, int64_t own_lo, int64_t own_hi, char *err, size_t errlen) {
    if (!out) return nl_fail(err, errlen, NL_EINVAL, "out is NULL");
    *out = nullptr;
    if (nzl < 1 || ny < 1 || nx < 1) return nl_fail(err, errlen, NL_EINVAL, "empty volume (%lld,%lld,%lld)", (i64)nzl, (i64)ny, (i64)nx);
    if (gz0 < 0 || gz0 + nzl > gnz) return nl_fail(err, errlen, NL_EINVAL, "slab [%lld,%lld) outside the global volume of %lld planes", (i64)gz0, (i64)(gz0 + nzl), (i64)gnz);
    if (own_lo < 0 || own_hi > nzl || own_lo >= own_hi) return nl_fail(err, errlen, NL_EINVAL, "bad owned range [%lld,%lld)", (i64)own_lo, (i64)own_hi);
    const i64 n = (i64)nzl * ny * nx;
    if (n >= ((i64)1 << 31)) return nl_fail(err, errlen, NL_EINVAL, "local slab of %lld voxels exceeds the int32 label index range; shard over Z", n);
    if (ny > 65535 || nzl > 65535) return nl_fail(err, errlen, NL_EINVAL, "Y and local Z extents must be <= 65535");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return nl_fail(err, errlen, NL_ENODEV, "GPU backend requested but no HIP device is available (%s)", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return nl_fail(err, errlen, NL_ENODEV, "GPU backend requested but device %d does not exist (%d devices)", device, ndev);
    NL_HIP(hipSetDevice(device));
    nl_ctx *c = new nl_ctx();
    c->device = device; c->nzl = nzl; c->ny = ny; c->nx = nx; c->gz0 = gz0; c->gnz = gnz;
    c->own_lo = own_lo; c->own_hi = own_hi; c->n = n;
    int rc = NL_OK;
    auto alloc = [&](void **p, size_t bytes) -> bool {
        hipError_t ee = hipMalloc(p, bytes);
        if (ee != hipSuccess) {
            rc = nl_fail(err, errlen, ee == hipErrorOutOfMemory ? NL_ENOMEM : NL_EHIP,
                         "hipMalloc(%zu bytes): %s%s", bytes, hipGetErrorString(ee), ee == hipErrorOutOfMemory ? " [out of memory]" : "");
            return false;
        }
        return true;
    };
    bool ok = true;
    for (int k = 0; k < 4 && ok; ++k) ok = alloc((void **)&c->f[k], (size_t)n * 4);
    // m[0] doubles as the cumulative bit mask of Filter: one 64-bit word per 64 x-voxels of a row
    const size_t mask_words = (size_t)NL_MASK_SLOTS * nzl * ny * ((nx + 63) / 64);
    const size_t plane_words_bytes = (size_t)nzl * ny * ((nx + 63) / 64) * 8;     // one bit plane
    for (int k = 0; k < 3 && ok; ++k) {
        size_t bytes = (size_t)n;
        if (k == 0 && mask_words * 8 > bytes) bytes = mask_words * 8;
        if (plane_words_bytes > bytes) bytes = plane_words_bytes;
        ok = alloc((void **)&c->m[k], bytes);
    }
    if (ok) ok = alloc((void **)&c->d_rows, ((size_t)nzl * ny + 2) * 2 * 4);
    if (ok) ok = alloc(&c->d_small, 1 << 16);
    c->blk_cap = (n + SCAN_CHUNK - 1) / SCAN_CHUNK + 1;
    if (ok) ok = alloc(&c->d_blk, (size_t)c->blk_cap * 4);
    if (ok && hipHostMalloc(&c->h_small, 1 << 16, hipHostMallocDefault) != hipSuccess) {
        rc = nl_fail(err, errlen, NL_ENOMEM, "hipHostMalloc failed [out of memory]"); ok = false;
    }
    if (ok && (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
               hipEventCreate(&c->t0) != hipSuccess || hipEventCreate(&c->t1) != hipSuccess)) {
        rc = nl_fail(err, errlen, NL_EHIP, "stream/event creation failed"); ok = false;
    }
    if (!ok) { nl_ctx_destroy(c); return rc; }
    *out = c;
    return NL_OK;
}

extern "C" int nl_sync(nl_ctx *c, char *err, size_t errlen) {
    if (!c) return nl_fail(err, errlen, NL_EINVAL, "ctx is NULL");
    NL_HIP(hipSetDevice(c->device));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

#define NL_ENTER(c)                                                    \
    if (!(c)) return nl_fail(err, errlen, NL_EINVAL, "ctx is NULL");   \
    NL_HIP(hipSetDevice((c)->device));

static int upload_convert(nl_ctx *c, const void *host, int dtype, float *dst, i64 count, char *err, size_t errlen) {
    const size_t es = dtype_size(dtype);
    if (!es) return nl_fail(err, errlen, NL_EINVAL, "unsupported dtype code %d", dtype);
    if (dtype == NL_F32) {
        NL_HIP(hipMemcpyAsync(dst, host, (size_t)count * 4, hipMemcpyHostToDevice, c->stream));
        NL_HIP(hipStreamSynchronize(c->stream));
        return NL_OK;
    }
    // stage raw bytes in free float volumes (2 consecutive volumes cover 8-byte types)
    void *raw = nullptr;
    bool own = false;
    // find a free f[] buffer that is not dst's buffer
    for (int k = 0; k < 4 && !raw; ++k) {
        const bool contains = (dst >= c->f[k] && dst < c->f[k] + c->n);
        if (!contains && k != c->i_vmax && es <= 4) raw = c->f[k];
    }
    if (!raw) { NL_HIP(hipMalloc(&raw, (size_t)count * es)); own = true; }
    NL_HIP(hipMemcpyAsync(raw, host, (size_t)count * es, hipMemcpyHostToDevice, c->stream));
    const unsigned int g = grid1d(count);
    switch (dtype) {
        case NL_U8: convert_kernel<uint8_t><<<g, 256, 0, c->stream>>>((const uint8_t *)raw, dst, count); break;
        case NL_I8: convert_kernel<int8_t><<<g, 256, 0, c->stream>>>((const int8_t *)raw, dst, count); break;
        case NL_U16: convert_kernel<uint16_t><<<g, 256, 0, c->stream>>>((const uint16_t *)raw, dst, count); break;
        case NL_I16: convert_kernel<int16_t><<<g, 256, 0, c->stream>>>((const int16_t *)raw, dst, count); break;
        case NL_U32: convert_kernel<uint32_t><<<g, 256, 0, c->stream>>>((const uint32_t *)raw, dst, count); break;
        case NL_I32: convert_kernel<int32_t><<<g, 256, 0, c->stream>>>((const int32_t *)raw, dst, count); break;
        case NL_F64: convert_kernel<double><<<g, 256, 0, c->stream>>>((const double *)raw, dst, count); break;
        case NL_U64: convert_kernel<uint64_t><<<g, 256, 0, c->stream>>>((const uint64_t *)raw, dst, count); break;
        case NL_I64: convert_kernel<int64_t><<<g, 256, 0, c->stream>>>((const int64_t *)raw, dst, count); break;
    }
    NL_CHECK_LAUNCH();
    NL_HIP(hipStreamSynchronize(c->stream));
    if (own) hipFree(raw);
    return NL_OK;
}

extern "C" int nl_filter_load(nl_ctx *c, const void *host, int dtype, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!host || z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    c->i_gauss = 0; c->i_vmax = 3; c->i_labels = -1; c->frangi_ready = 0;
    const i64 plane = c->ny * c->nx;
    int rc = upload_convert(c, host, dtype, c->f[0] + z0 * plane, (z1 - z0) * plane, err, errlen);
    if (rc) return rc;
    // vesselness = zeros, masks = ones (filtering.py:807-808): implicit -- the first evaluated scale writes
    // vesselness instead of max-ing it and nl_filter_finish zeroes whatever the masks reject
    c->mask_slots_used = 0;
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

// Keep the raw frame resident in HBM (any dtype) so that a timed region can start from device memory.
extern "C" int nl_input_load(nl_ctx *c, const void *host, int dtype, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    const size_t es = dtype_size(dtype);
    if (!es) return nl_fail(err, errlen, NL_EINVAL, "unsupported dtype code %d", dtype);
    if (!host || z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    if (c->d_input && c->input_dtype != dtype) { hipFree(c->d_input); c->d_input = nullptr; }
    if (!c->d_input) NL_HIP(hipMalloc(&c->d_input, (size_t)c->n * es));
    c->input_dtype = dtype;
    const i64 plane = c->ny * c->nx;
    NL_HIP(hipMemcpyAsync((char *)c->d_input + (size_t)z0 * plane * es, host, (size_t)(z1 - z0) * plane * es, hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

// frame = xp.asarray(resident input, dtype=float32); vesselness = 0; masks = 1.  Asynchronous.
extern "C" int nl_filter_begin(nl_ctx *c, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->d_input) return nl_fail(err, errlen, NL_ESTATE, "nl_filter_begin before nl_input_load");
    c->i_gauss = 0; c->i_vmax = 3; c->i_labels = -1; c->frangi_ready = 0;
    ProfScope ps(c, "load");
    const unsigned int g = grid1d(c->n);
    switch (c->input_dtype) {
        case NL_U8: convert_kernel<uint8_t><<<g, 256, 0, c->stream>>>((const uint8_t *)c->d_input, c->f[0], c->n); break;
        case NL_I8: convert_kernel<int8_t><<<g, 256, 0, c->stream>>>((const int8_t *)c->d_input, c->f[0], c->n); break;
        case NL_U16: convert_kernel<uint16_t><<<g, 256, 0, c->stream>>>((const uint16_t *)c->d_input, c->f[0], c->n); break;
        case NL_I16: convert_kernel<int16_t><<<g, 256, 0, c->stream>>>((const int16_t *)c->d_input, c->f[0], c->n); break;
        case NL_U32: convert_kernel<uint32_t><<<g, 256, 0, c->stream>>>((const uint32_t *)c->d_input, c->f[0], c->n); break;
        case NL_I32: convert_kernel<int32_t><<<g, 256, 0, c->stream>>>((const int32_t *)c->d_input, c->f[0], c->n); break;
        case NL_F32: convert_kernel<float><<<g, 256, 0, c->stream>>>((const float *)c->d_input, c->f[0], c->n); break;
        case NL_F64: convert_kernel<double><<<g, 256, 0, c->stream>>>((const double *)c->d_input, c->f[0], c->n); break;
        case NL_U64: convert_kernel<uint64_t><<<g, 256, 0, c->stream>>>((const uint64_t *)c->d_input, c->f[0], c->n); break;
        case NL_I64: convert_kernel<int64_t><<<g, 256, 0, c->stream>>>((const int64_t *)c->d_input, c->f[0], c->n); break;
    }
    NL_CHECK_LAUNCH();
    c->mask_slots_used = 0;
    return NL_OK;
}

static int fill_gw(GaussW &gw, const double *w, int r, char *err, size_t errlen) {
    if (r < 0 || r > NL_MAX_RADIUS) return nl_fail(err, errlen, NL_EINVAL, "Gaussian radius %d outside [0,%d]", r, NL_MAX_RADIUS);
    gw.r = r;
    for (int k = 0; k <= r; ++k) gw.w[k] = w[r + k];   // w[] has 2r+1 entries centred at r (symmetric)
    return NL_OK;
}

template <int AXIS, int R>
static void launch_gauss_march(nl_ctx *c, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussW &gw) {
    GaussWS ws;
    for (int k = 0; k <= GM_MAX_R; ++k) ws.w[k] = k <= R ? gw.w[k] : 0.0;
    dim3 grid;
    if (AXIS == 0) grid = dim3((unsigned)((c->nx + 63) / 64), (unsigned)((c->ny + 3) / 4), (unsigned)((z1 - z0 + GM_CHUNK - 1) / GM_CHUNK));
    else grid = dim3((unsigned)((c->nx + 63) / 64), (unsigned)((z1 - z0 + 3) / 4), (unsigned)((c->ny + GM_CHUNK - 1) / GM_CHUNK));
    gauss_march_kernel<AXIS, R><<<grid, 256, 0, c->stream>>>(src, dst, v, z0, z1, ws);
}
template <int R>
static void launch_gauss_x(nl_ctx *c, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussW &gw) {
    GaussWS ws;
    for (int k = 0; k <= GM_MAX_R; ++k) ws.w[k] = k <= R ? gw.w[k] : 0.0;
    const dim3 grid((unsigned)((c->nx + GX_SEG - 1) / GX_SEG), (unsigned)c->ny, (unsigned)(z1 - z0));
    gauss_x_kernel<R><<<grid, 256, 0, c->stream>>>(src, dst, v, z0, z1, ws, (c->nx % 4 == 0) ? 1 : 0);
}
// returns false when the radius has no specialised kernel
template <int AXIS>
static bool launch_gauss_fast(nl_ctx *c, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussW &gw) {
#define NL_GCASE(RR)                                                                         \
    case RR:                                                                                 \
        if (AXIS == 2) launch_gauss_x<RR>(c, src, dst, v, z0, z1, gw);                       \
        else launch_gauss_march<(AXIS == 2 ? 0 : AXIS), RR>(c, src, dst, v, z0, z1, gw);     \
        return true;
    switch (gw.r) {
        NL_GCASE(1) NL_GCASE(2) NL_GCASE(3) NL_GCASE(4) NL_GCASE(5) NL_GCASE(6) NL_GCASE(7) NL_GCASE(8)
        default: return false;
    }
#undef NL_GCASE
}

extern "C" int nl_gauss_step(nl_ctx *c, const double *wz, int rz, const double *wy, int ry, const double *wx, int rx,
                             int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    if (z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    const VolGeom v = geom(c);
    // three ping-pong volumes f[0..2]: the source of a pass is dead once the pass has run,
    // so "the next one" is always a legal destination
    const dim3 blk(256, 1, 1);
    const dim3 grid((unsigned)((c->nx + 255) / 256), (unsigned)c->ny, (unsigned)(z1 - z0));
    ProfScope ps(c, "gauss");
    int src = c->i_gauss;
    GaussW gw;
    int rc;
    if (wz) {
        if ((rc = fill_gw(gw, wz, rz, err, errlen))) return rc;
        // every tap must land inside the local slab unless it reflects at a true face
        if ((z0 - rz < 0 && c->gz0 > 0) || (z1 - 1 + rz >= c->nzl && c->gz0 + c->nzl < c->gnz))
            return nl_fail(err, errlen, NL_EINVAL, "Z pass of radius %d on planes [%lld,%lld) reaches outside the local slab", rz, (i64)z0, (i64)z1);
        const int dst = (src + 1) % 3;
        if (!launch_gauss_fast<0>(c, c->f[src], c->f[dst], v, z0, z1, gw))
            gauss_axis_kernel<0><<<grid, blk, 0, c->stream>>>(c->f[src], c->f[dst], v, z0, z1, gw);
        NL_CHECK_LAUNCH();
        src = dst;
    }
    bool fused_yx = false;
    if (wy && wx && ry == rx && ry >= 1 && ry <= GM_MAX_R && !getenv("NELLIE_NO_FUSED_YX")) {
        GaussW gy, gx;
        if ((rc = fill_gw(gy, wy, ry, err, errlen))) return rc;
        if ((rc = fill_gw(gx, wx, rx, err, errlen))) return rc;
        GaussWS wsy, wsx;
        for (int k = 0; k <= GM_MAX_R; ++k) { wsy.w[k] = k <= ry ? gy.w[k] : 0.0; wsx.w[k] = k <= rx ? gx.w[k] : 0.0; }
        const int dst = (src + 1) % 3;
        const dim3 g2((unsigned)((c->nx + GYX_COLS - 1) / GYX_COLS), (unsigned)((c->ny + GM_CHUNK - 1) / GM_CHUNK), (unsigned)(z1 - z0));
        switch (ry) {
#define NL_YX(RR) case RR: gauss_yx_kernel<RR><<<g2, GYX_THREADS, 0, c->stream>>>(c->f[src], c->f[dst], v, z0, z1, wsy, wsx); break;
            NL_YX(1) NL_YX(2) NL_YX(3) NL_YX(4) NL_YX(5) NL_YX(6) NL_YX(7) NL_YX(8)
#undef NL_YX
        }
        NL_CHECK_LAUNCH();
        src = dst;
        fused_yx = true;
    }
    if (wy && !fused_yx) {
        if ((rc = fill_gw(gw, wy, ry, err, errlen))) return rc;
        const int dst = (src + 1) % 3;
        if (!launch_gauss_fast<1>(c, c->f[src], c->f[dst], v, z0, z1, gw))
            gauss_axis_kernel<1><<<grid, blk, 0, c->stream>>>(c->f[src], c->f[dst], v, z0, z1, gw);
        NL_CHECK_LAUNCH();
        src = dst;
    }
    if (wx && !fused_yx) {
        if ((rc = fill_gw(gw, wx, rx, err, errlen))) return rc;
        const int dst = (src + 1) % 3;
        if (!launch_gauss_fast<2>(c, c->f[src], c->f[dst], v, z0, z1, gw))
            gauss_axis_kernel<2><<<grid, blk, 0, c->stream>>>(c->f[src], c->f[dst], v, z0, z1, gw);
        NL_CHECK_LAUNCH();
        src = dst;
    }
    c->i_gauss = src;
    return NL_OK;
}

static int make_lattice(const nl_ctx *c, i64 sz, i64 sy, i64 sx, Lattice &L, char *err, size_t errlen) {
    if (sz < 1 || sy < 1 || sx < 1) return nl_fail(err, errlen, NL_EINVAL, "strides must be >= 1");
    L.sz = sz; L.sy = sy; L.sx = sx;
    // owned global planes [g_lo, g_hi): lattice planes are global z = k*sz
    const i64 g_lo = c->gz0 + c->own_lo, g_hi = c->gz0 + c->own_hi;
    const i64 k_lo = (g_lo + sz - 1) / sz, k_hi = (g_hi + sz - 1) / sz;   // k in [k_lo, k_hi)
    L.cz = k_hi > k_lo ? k_hi - k_lo : 0;
    L.zfirst = k_lo * sz - c->gz0;
    L.cy = (c->ny + sy - 1) / sy;
    L.cx = (c->nx + sx - 1) / sx;
    return NL_OK;
}

static int make_field(nl_ctx *c, int field, FieldSrc &fs, char *err, size_t errlen) {
    fs.field = field; fs.hp = hessp(c); fs.max_abs = c->frob_max_abs; fs.max_finite = c->frob_max_finite;
    if (field == NL_FIELD_GAUSS) fs.p = c->f[c->i_gauss];
    else if (field == NL_FIELD_FROB) {
        if (!c->have_spacing) return nl_fail(err, errlen, NL_ESTATE, "NL_FIELD_FROB before nl_hessian_stats");
        fs.p = c->f[c->i_gauss];
    } else if (field == NL_FIELD_FRANGI) fs.p = c->f[c->i_vmax];
    else return nl_fail(err, errlen, NL_EINVAL, "unknown field %d", field);
    return NL_OK;
}

extern "C" int nl_sample_gather(nl_ctx *c, int field, int64_t sz, int64_t sy, int64_t sx, float *out, int64_t cap,
                                int64_t *n, char *err, size_t errlen) {
    NL_ENTER(c);
    Lattice L; FieldSrc fs; int rc;
    if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
    if ((rc = make_field(c, field, fs, err, errlen))) return rc;
    const i64 total = L.cz * L.cy * L.cx;
    if (n) *n = total;
    if (total == 0 || (!out && cap == 0)) return NL_OK;   // size query
    if (!out || cap < total) return nl_fail(err, errlen, NL_EINVAL, "output capacity %lld < %lld samples", (i64)cap, total);
    // a free float volume as staging: whichever of f[0..2] is not the current gauss
    float *stage = c->f[(c->i_gauss + 1) % 3];
    if (total > c->n) return nl_fail(err, errlen, NL_EINVAL, "lattice larger than the volume");
    {
        ProfScope ps(c, "sample");
        sample_gather_kernel<<<(unsigned)((total + 255) / 256), 256, 0, c->stream>>>(fs, geom(c), L, stage);
        NL_CHECK_LAUNCH();
    }
    NL_HIP(hipMemcpyAsync(out, stage, (size_t)total * 4, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

extern "C" int nl_sample_minmax(nl_ctx *c, int field, int64_t sz, int64_t sy, int64_t sx, float *mn, float *mx,
                                int64_t *npos, char *err, size_t errlen) {
    NL_ENTER(c);
    Lattice L; FieldSrc fs; int rc;
    if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
    if ((rc = make_field(c, field, fs, err, errlen))) return rc;
    const i64 total = L.cz * L.cy * L.cx;
    unsigned int *res = (unsigned int *)c->d_small;
    unsigned int *h = (unsigned int *)c->h_small;
    h[0] = 0xffffffffu; h[1] = 0; h[2] = 0; h[3] = 0;
    NL_HIP(hipMemcpyAsync(res, h, 16, hipMemcpyHostToDevice, c->stream));
    if (total > 0) {
        ProfScope ps(c, "sample");
        sample_minmax_kernel<<<grid1d(total, 256, 1024), 256, 0, c->stream>>>(fs, geom(c), L, res);
        NL_CHECK_LAUNCH();
    }
    NL_HIP(hipMemcpyAsync(h, res, 16, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    const unsigned long long cnt = *(unsigned long long *)(h + 2);
    if (npos) *npos = (int64_t)cnt;
    if (cnt) {
        if (mn) memcpy(mn, &h[0], 4);
        if (mx) memcpy(mx, &h[1], 4);
    }
    return NL_OK;
}

extern "C" int nl_sample_hist(nl_ctx *c, int field, int64_t sz, int64_t sy, int64_t sx, const float *edges, int nbins,
                              int64_t *counts, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!edges || !counts || nbins < 1 || nbins > 4096) return nl_fail(err, errlen, NL_EINVAL, "bad histogram arguments (nbins=%d)", nbins);
    Lattice L; FieldSrc fs; int rc;
    if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
    if ((rc = make_field(c, field, fs, err, errlen))) return rc;
    const i64 total = L.cz * L.cy * L.cx;
    // d_small layout: [0, 32K) counts (u64 x nbins), [32K, 64K) edges (f32 x nbins+1)
    unsigned long long *d_counts = (unsigned long long *)c->d_small;
    float *d_edges = (float *)((char *)c->d_small + (1 << 15));
    NL_HIP(hipMemsetAsync(d_counts, 0, (size_t)nbins * 8, c->stream));
    memcpy((char *)c->h_small + (1 << 15), edges, (size_t)(nbins + 1) * 4);
    NL_HIP(hipMemcpyAsync(d_edges, (char *)c->h_small + (1 << 15), (size_t)(nbins + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (total > 0) {
        ProfScope ps(c, "sample");
        const size_t sh = (size_t)(nbins + 2) * 4 + (size_t)nbins * 4;
        sample_hist_kernel<<<grid1d(total, 256, 1024), 256, sh, c->stream>>>(fs, geom(c), L, d_edges, nbins, d_counts);
        NL_CHECK_LAUNCH();
    }
    NL_HIP(hipMemcpyAsync(c->h_small, d_counts, (size_t)nbins * 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    memcpy(counts, c->h_small, (size_t)nbins * 8);
    return NL_OK;
}

extern "C" int nl_hessian_stats(nl_ctx *c, const double spacing[3], float *max_abs, float *max_frob_sq, int *any_inf,
                                char *err, size_t errlen) {
    NL_ENTER(c);
    if (!spacing) return nl_fail(err, errlen, NL_EINVAL, "spacing is NULL");
    if (c->gnz < 2 || c->ny < 2 || c->nx < 2)
        return nl_fail(err, errlen, NL_EINVAL, "Shape of array too small to calculate a numerical gradient, at least (edge_order + 1) elements are required.");
    c->hz = (float)spacing[0]; c->hy = (float)spacing[1]; c->hx = (float)spacing[2];
    c->hz2 = (float)(2.0 * spacing[0]); c->hy2 = (float)(2.0 * spacing[1]); c->hx2 = (float)(2.0 * spacing[2]);
    if (!c->have_spacing || c->chk_spacing[0] != spacing[0] || c->chk_spacing[1] != spacing[1] || c->chk_spacing[2] != spacing[2]) {
        int rcx = check_fast_div(c, err, errlen);
        if (rcx) return rcx;
        c->chk_spacing[0] = spacing[0]; c->chk_spacing[1] = spacing[1]; c->chk_spacing[2] = spacing[2];
    }
    c->have_spacing = 1;
    unsigned int *res = (unsigned int *)c->d_small;
    NL_HIP(hipMemsetAsync(res, 0, 16, c->stream));
    {
        ProfScope ps(c, "hessian_stats");
        const int ntx = (int)((c->nx + HM_TX - 1) / HM_TX), nty = (int)((c->ny + 15) / 16);
        const int nzc = (int)((c->own_hi - c->own_lo + HM_ZCHUNK - 1) / HM_ZCHUNK);
        VessP vp{};
#define NL_LAUNCH_STATS(TYV, FASTV, HR)                                                                                   \
        hessian_march_kernel<0, TYV, FASTV><<<(unsigned)(ntx * (int)((c->ny + TYV - 1) / TYV) * nzc), HMCfg<TYV>::NT,     \
                                              HMCfg<TYV>::lds_floats(0) * 4, c->stream>>>(                                \
            c->f[c->i_gauss], nullptr, nullptr, 0, geom(c), HR, vp, (int)c->own_lo, (int)c->own_hi, ntx,                  \
            (int)((c->ny + TYV - 1) / TYV), res, nullptr)
        if (hm_ty() == 8) { if (c->fast_div) NL_LAUNCH_STATS(8, true, hessdv_fast(c)); else NL_LAUNCH_STATS(8, false, hessdv_exact(c)); }
        else { if (c->fast_div) NL_LAUNCH_STATS(16, true, hessdv_fast(c)); else NL_LAUNCH_STATS(16, false, hessdv_exact(c)); }
#undef NL_LAUNCH_STATS
        NL_CHECK_LAUNCH();
    }
    unsigned int *h = (unsigned int *)c->h_small;
    NL_HIP(hipMemcpyAsync(h, res, 16, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    if (max_abs) memcpy(max_abs, &h[0], 4);
    if (max_frob_sq) memcpy(max_frob_sq, &h[1], 4);
    if (any_inf) *any_inf = (int)h[2];
    return NL_OK;
}

extern "C" int nl_set_frob_norm(nl_ctx *c, float max_abs, float max_finite, char *err, size_t errlen) {
    if (!c) return nl_fail(err, errlen, NL_EINVAL, "ctx is NULL");
    c->frob_max_abs = max_abs; c->frob_max_finite = max_finite;
    return NL_OK;
}

extern "C" int nl_vesselness_step(nl_ctx *c, float gamma_sq, float alpha_sq, float beta_sq, int use_thr, float thr,
                                  int64_t z0, int64_t z1, int64_t *mask_count, char *err, size_t errlen) {
    NL_ENTER(c);
    if (z0 < 0 && z1 < 0) { z0 = c->own_lo; z1 = c->own_hi; }
    if (z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    if (!c->have_spacing) return nl_fail(err, errlen, NL_ESTATE, "nl_vesselness_step before nl_hessian_stats");
    unsigned long long *d_cnt = (unsigned long long *)c->d_small;
    NL_HIP(hipMemsetAsync(d_cnt, 0, 8, c->stream));
    VessP vp{gamma_sq, alpha_sq, beta_sq, use_thr, thr, c->frob_max_abs, c->frob_max_finite, 0, (int)c->own_lo, (int)c->own_hi,
             c->mask_slots_used == 0 ? 1 : 0};
    {
        ProfScope ps(c, "vesselness");
        const int ntx = (int)((c->nx + HM_TX - 1) / HM_TX), nty = (int)((c->ny + 15) / 16);
        const int nzc = (int)((z1 - z0 + HM_ZCHUNK - 1) / HM_ZCHUNK);
        static bool attr_set = false;
        if (!attr_set) {
            NL_HIP(hipFuncSetAttribute((const void *)hessian_march_kernel<1, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, HMCfg<16>::lds_floats(1) * 4));
            NL_HIP(hipFuncSetAttribute((const void *)hessian_march_kernel<1, 16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, HMCfg<16>::lds_floats(1) * 4));
            NL_HIP(hipFuncSetAttribute((const void *)hessian_march_kernel<1, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, HMCfg<8>::lds_floats(1) * 4));
            NL_HIP(hipFuncSetAttribute((const void *)hessian_march_kernel<1, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, HMCfg<8>::lds_floats(1) * 4));
            attr_set = true;
        }
        const int wpr = (int)((c->nx + 63) / 64);
        const i64 slot_words = c->nzl * c->ny * wpr;
        int slot = c->mask_slots_used;
        if (slot >= NL_MASK_SLOTS) { slot = NL_MASK_SLOTS - 1; vp.mask_rmw = 1; } else c->mask_slots_used++;
        unsigned long long *cm = (unsigned long long *)c->m[0] + (i64)slot * slot_words;
#define NL_LAUNCH_VESS(TYV, FASTV, HR)                                                                                    \
        hessian_march_kernel<1, TYV, FASTV><<<(unsigned)(ntx * (int)((c->ny + TYV - 1) / TYV) * nzc), HMCfg<TYV>::NT,     \
                                              HMCfg<TYV>::lds_floats(1) * 4, c->stream>>>(                                \
            c->f[c->i_gauss], c->f[c->i_vmax], cm, wpr, geom(c), HR, vp, (int)z0, (int)z1, ntx,             \
            (int)((c->ny + TYV - 1) / TYV), nullptr, d_cnt)
        if (hm_ty() == 8) { if (c->fast_div) NL_LAUNCH_VESS(8, true, hessdv_fast(c)); else NL_LAUNCH_VESS(8, false, hessdv_exact(c)); }
        else { if (c->fast_div) NL_LAUNCH_VESS(16, true, hessdv_fast(c)); else NL_LAUNCH_VESS(16, false, hessdv_exact(c)); }
#undef NL_LAUNCH_VESS
        NL_CHECK_LAUNCH();
    }
    if (mask_count) {
        NL_HIP(hipMemcpyAsync(c->h_small, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
        NL_HIP(hipStreamSynchronize(c->stream));
        *mask_count = (int64_t)(*(unsigned long long *)c->h_small);
    }
    return NL_OK;
}

extern "C" int nl_filter_finish(nl_ctx *c, int64_t z0, int64_t z1, int64_t *n_positive, char *err, size_t errlen) {
    NL_ENTER(c);
    if (z0 < 0 && z1 < 0) { z0 = c->own_lo; z1 = c->own_hi; }
    if (z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    unsigned long long *d_cnt = (unsigned long long *)c->d_small;
    NL_HIP(hipMemsetAsync(d_cnt, 0, 8, c->stream));
    const i64 plane = c->ny * c->nx;
    if (c->mask_slots_used == 0)       // every scale was skipped: vesselness was never written
        NL_HIP(hipMemsetAsync(c->f[c->i_vmax] + z0 * plane, 0, (size_t)(z1 - z0) * plane * 4, c->stream));
    {
        ProfScope ps(c, "finish");
        const int wpr = (int)((c->nx + 63) / 64);
        const i64 quads = (z1 - z0) * c->ny * ((c->nx + 3) / 4);
        finish_kernel<<<grid1d(quads, 256, 256 * 16), 256, 0, c->stream>>>(c->f[c->i_vmax], (const unsigned long long *)c->m[0], c->mask_slots_used,
                                                               c->nzl * c->ny * wpr, wpr, geom(c), z0, z1, c->own_lo, c->own_hi, d_cnt);
        NL_CHECK_LAUNCH();
    }
    NL_HIP(hipMemcpyAsync(c->h_small, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    if (n_positive) *n_positive = (int64_t)(*(unsigned long long *)c->h_small);
    c->frangi_ready = 1;
    return NL_OK;
}

extern "C" int nl_mask_volume(nl_ctx *c, float thr, char *err, size_t errlen) {
    NL_ENTER(c);
    // result goes to a free gauss volume, which then becomes the Frangi volume
    int dst = -1;
    for (int k = 0; k < 3; ++k) if (k != c->i_gauss) { dst = k; break; }
    {
        ProfScope ps(c, "mask_volume");
        const int wpr = (int)((c->nx + 63) / 64);
        const VolGeom v = geom(c);
        // planes whose bits exist: own +-2 clipped to the slab (ghost planes beyond a true face do not exist)
        const i64 m0 = c->own_lo - 2 > 0 ? c->own_lo - 2 : 0, m1 = c->own_hi + 2 < c->nzl ? c->own_hi + 2 : c->nzl;
        const i64 e0 = c->own_lo - 1 > 0 ? c->own_lo - 1 : 0, e1 = c->own_hi + 1 < c->nzl ? c->own_hi + 1 : c->nzl;
        unsigned long long *bM = (unsigned long long *)c->m[1], *bE = (unsigned long long *)c->m[2], *bD = (unsigned long long *)c->m[0];
        rl_threshold_pack_kernel<<<grid1d((m1 - m0) * c->ny * wpr * 64, 256, 256 * 32), 256, 0, c->stream>>>(
            c->f[c->i_vmax] + m0 * c->ny * c->nx, bM + m0 * c->ny * wpr, 1, thr, (int)c->nx, (m1 - m0) * c->ny, wpr);
        NL_CHECK_LAUNCH();
        bits_morph6_kernel<0><<<(unsigned)(((e1 - e0) * c->ny * wpr + 255) / 256), 256, 0, c->stream>>>(bM, bE, v, wpr, e0, e1);
        NL_CHECK_LAUNCH();
        bits_morph6_kernel<1><<<(unsigned)(((c->own_hi - c->own_lo) * c->ny * wpr + 255) / 256), 256, 0, c->stream>>>(bE, bD, v, wpr, c->own_lo, c->own_hi);
        NL_CHECK_LAUNCH();
        apply_bits_kernel<<<grid1d((c->own_hi - c->own_lo) * c->ny * ((c->nx + 3) / 4), 256, 256 * 32), 256, 0, c->stream>>>(
            c->f[c->i_vmax], bD, c->f[dst], v, wpr, c->own_lo, c->own_hi);
        NL_CHECK_LAUNCH();
    }
    // swap roles: old vmax volume joins the gauss ping-pong set
    float *tmp = c->f[c->i_vmax];
    c->f[c->i_vmax] = c->f[dst];
    c->f[dst] = tmp;
    return NL_OK;
}

static int store_planes(nl_ctx *c, const void *dev_base, void *host, size_t elem, int64_t z0, int64_t z1, char *err, size_t errlen) {
    if (!host || z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    const i64 plane = c->ny * c->nx;
    NL_HIP(hipMemcpyAsync(host, (const char *)dev_base + (size_t)z0 * plane * elem, (size_t)(z1 - z0) * plane * elem,
                          hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

extern "C" int nl_filter_store(nl_ctx *c, float *host, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    return store_planes(c, c->f[c->i_vmax], host, 4, z0, z1, err, errlen);
}
extern "C" int nl_gauss_store(nl_ctx *c, float *host, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    return store_planes(c, c->f[c->i_gauss], host, 4, z0, z1, err, errlen);
}

// ------------------------------------------------------------------------------ slab helpers ----
static float *field_ptr(nl_ctx *c, int field) {
    if (field == NL_FIELD_GAUSS) return c->f[c->i_gauss];
    if (field == NL_FIELD_FRANGI) return c->f[c->i_vmax];
    return nullptr;
}

extern "C" int nl_planes_get(nl_ctx *c, int field, int64_t z0, int64_t z1, float *host, char *err, size_t errlen) {
    NL_ENTER(c);
    float *p = field_ptr(c, field);
    if (!p) return nl_fail(err, errlen, NL_EINVAL, "nl_planes_get: field %d has no volume", field);
    return store_planes(c, p, host, 4, z0, z1, err, errlen);
}

extern "C" int nl_planes_put(nl_ctx *c, int field, int64_t z0, int64_t z1, const float *host, char *err, size_t errlen) {
    NL_ENTER(c);
    float *p = field_ptr(c, field);
    if (!p || !host || z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "nl_planes_put: bad field or plane range");
    const i64 plane = c->ny * c->nx;
    NL_HIP(hipMemcpyAsync(p + z0 * plane, host, (size_t)(z1 - z0) * plane * 4, hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

#define NL_NCCL(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) return nl_fail(err, errlen, NL_ECOMM, "%s: %s", #expr, ncclGetErrorString(r_)); \
    } while (0)

extern "C" int nl_comm_unique_id(char *id128, char *err, size_t errlen) {
    if (!id128) return nl_fail(err, errlen, NL_EINVAL, "id buffer is NULL");
    ncclUniqueId id;
    NL_NCCL(ncclGetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return NL_OK;
}

extern "C" int nl_comm_init(nl_ctx *c, int world, int rank, const char *id128, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!id128 || world < 1 || rank < 0 || rank >= world) return nl_fail(err, errlen, NL_EINVAL, "bad communicator arguments");
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm;
    NL_NCCL(ncclCommInitRank(&comm, world, id, rank));
    c->comm = comm; c->world = world; c->rank = rank;
    return NL_OK;
}

// Ghost-plane exchange with the Z neighbours over RCCL (xGMI): this rank's first / last `depth` owned planes
// go to the neighbour's ghost planes, and the neighbours' go into ours.  Asynchronous on the context stream.
extern "C" int nl_halo_exchange(nl_ctx *c, int field, int64_t depth, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->comm) return nl_fail(err, errlen, NL_ESTATE, "nl_halo_exchange before nl_comm_init");
    float *p = field_ptr(c, field);
    if (!p) return nl_fail(err, errlen, NL_EINVAL, "nl_halo_exchange: field %d has no volume", field);
    const i64 plane = c->ny * c->nx;
    const bool has_lo = c->rank > 0, has_hi = c->rank + 1 < c->world;
    if (depth < 1 || depth > c->own_hi - c->own_lo || (has_lo && depth > c->own_lo) || (has_hi && depth > c->nzl - c->own_hi))
        return nl_fail(err, errlen, NL_EINVAL, "halo depth %lld does not fit the slab (own %lld, ghosts %lld/%lld)", (i64)depth,
                       (i64)(c->own_hi - c->own_lo), (i64)c->own_lo, (i64)(c->nzl - c->own_hi));
    ncclComm_t comm = (ncclComm_t)c->comm;
    ProfScope ps(c, "halo");
    NL_NCCL(ncclGroupStart());
    if (has_lo) {
        NL_NCCL(ncclSend(p + c->own_lo * plane, (size_t)(depth * plane), ncclFloat, c->rank - 1, comm, c->stream));
        NL_NCCL(ncclRecv(p + (c->own_lo - depth) * plane, (size_t)(depth * plane), ncclFloat, c->rank - 1, comm, c->stream));
    }
    if (has_hi) {
        NL_NCCL(ncclSend(p + (c->own_hi - depth) * plane, (size_t)(depth * plane), ncclFloat, c->rank + 1, comm, c->stream));
        NL_NCCL(ncclRecv(p + c->own_hi * plane, (size_t)(depth * plane), ncclFloat, c->rank + 1, comm, c->stream));
    }
    NL_NCCL(ncclGroupEnd());
    return NL_OK;
}

// Small all-reduce of host values through RCCL: dtype 0 = int64, 1 = float32; op 0 = sum, 1 = min, 2 = max.
extern "C" int nl_allreduce(nl_ctx *c, void *host_inout, int64_t count, int dtype, int op, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->comm) return nl_fail(err, errlen, NL_ESTATE, "nl_allreduce before nl_comm_init");
    const size_t es = dtype == 0 ? 8 : 4;
    if (!host_inout || count < 1 || (size_t)count * es > (1 << 15) || dtype < 0 || dtype > 1 || op < 0 || op > 2)
        return nl_fail(err, errlen, NL_EINVAL, "bad all-reduce arguments");
    memcpy(c->h_small, host_inout, (size_t)count * es);
    NL_HIP(hipMemcpyAsync(c->d_small, c->h_small, (size_t)count * es, hipMemcpyHostToDevice, c->stream));
    const ncclRedOp_t ops[3] = {ncclSum, ncclMin, ncclMax};
    NL_NCCL(ncclAllReduce(c->d_small, c->d_small, (size_t)count, dtype == 0 ? ncclInt64 : ncclFloat, ops[op], (ncclComm_t)c->comm, c->stream));
    NL_HIP(hipMemcpyAsync(c->h_small, c->d_small, (size_t)count * es, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    memcpy(host_inout, c->h_small, (size_t)count * es);
    return NL_OK;
}

// ---------------------------------------------------------------------------------- Label -------
extern "C" int nl_label_load_frangi(nl_ctx *c, const float *host, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!host || z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    const i64 plane = c->ny * c->nx;
    c->i_vmax = 3; c->i_gauss = 0; c->i_labels = -1;
    NL_HIP(hipMemcpyAsync(c->f[c->i_vmax] + z0 * plane, host, (size_t)(z1 - z0) * plane * 4, hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    c->frangi_ready = 1;
    return NL_OK;
}

extern "C" int nl_label_intensity_mask(nl_ctx *c, const void *host_original, int dtype, double thresh, char *err, size_t errlen) {
    NL_ENTER(c);
    const size_t es = dtype_size(dtype);
    if (!es || !host_original) return nl_fail(err, errlen, NL_EINVAL, "bad original image (dtype code %d)", dtype);
    void *raw = nullptr;
    NL_HIP(hipMalloc(&raw, (size_t)c->n * es));
    hipError_t e = hipMemcpyAsync(raw, host_original, (size_t)c->n * es, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { hipFree(raw); return nl_fail(err, errlen, NL_EHIP, "upload of the original image failed: %s", hipGetErrorString(e)); }
    float *fr = c->f[c->i_vmax];
    const unsigned int g = grid1d(c->n);
    switch (dtype) {
        case NL_U8: intensity_mask_kernel<uint8_t><<<g, 256, 0, c->stream>>>((const uint8_t *)raw, fr, thresh, c->n); break;
        case NL_I8: intensity_mask_kernel<int8_t><<<g, 256, 0, c->stream>>>((const int8_t *)raw, fr, thresh, c->n); break;
        case NL_U16: intensity_mask_kernel<uint16_t><<<g, 256, 0, c->stream>>>((const uint16_t *)raw, fr, thresh, c->n); break;
        case NL_I16: intensity_mask_kernel<int16_t><<<g, 256, 0, c->stream>>>((const int16_t *)raw, fr, thresh, c->n); break;
        case NL_U32: intensity_mask_kernel<uint32_t><<<g, 256, 0, c->stream>>>((const uint32_t *)raw, fr, thresh, c->n); break;
        case NL_I32: intensity_mask_kernel<int32_t><<<g, 256, 0, c->stream>>>((const int32_t *)raw, fr, thresh, c->n); break;
        case NL_F32: intensity_mask_kernel<float><<<g, 256, 0, c->stream>>>((const float *)raw, fr, thresh, c->n); break;
        case NL_F64: intensity_mask_kernel<double><<<g, 256, 0, c->stream>>>((const double *)raw, fr, thresh, c->n); break;
        case NL_U64: intensity_mask_kernel<uint64_t><<<g, 256, 0, c->stream>>>((const uint64_t *)raw, fr, thresh, c->n); break;
        case NL_I64: intensity_mask_kernel<int64_t><<<g, 256, 0, c->stream>>>((const int64_t *)raw, fr, thresh, c->n); break;
    }
    e = hipGetLastError();
    hipStreamSynchronize(c->stream);
    hipFree(raw);
    if (e != hipSuccess) return nl_fail(err, errlen, NL_EHIP, "intensity mask kernel: %s", hipGetErrorString(e));
    return NL_OK;
}

extern "C" int nl_flat_sample_gather(nl_ctx *c, int field, int64_t offset, int64_t step, float *out, int64_t cap, int64_t *n,
                                     char *err, size_t errlen) {
    NL_ENTER(c);
    if (step < 1 || offset < 0) return nl_fail(err, errlen, NL_EINVAL, "bad offset/step");
    if (field != NL_FIELD_FRANGI && field != NL_FIELD_GAUSS) return nl_fail(err, errlen, NL_EINVAL, "flat sampling supports GAUSS/FRANGI");
    // flat index runs over the GLOBAL volume; this rank contributes indices inside its owned planes
    const i64 plane = c->ny * c->nx;
    const i64 g_begin = (c->gz0 + c->own_lo) * plane, g_end = (c->gz0 + c->own_hi) * plane;
    i64 k0 = 0;
    if (g_begin > offset) k0 = (g_begin - offset + step - 1) / step;
    i64 k1 = (g_end > offset) ? (g_end - offset + step - 1) / step : 0;    // k in [k0,k1)
    const i64 count = k1 > k0 ? k1 - k0 : 0;
    if (n) *n = count;
    if (count == 0 || (!out && cap == 0)) return NL_OK;   // size query
    if (!out || cap < count) return nl_fail(err, errlen, NL_EINVAL, "output capacity %lld < %lld samples", (i64)cap, count);
    const float *src = (field == NL_FIELD_FRANGI) ? c->f[c->i_vmax] : c->f[c->i_gauss];
    float *stage = nullptr;
    for (int k = 0; k < 3; ++k) if (k != c->i_gauss && c->f[k] != src) { stage = c->f[k]; break; }
    {
        ProfScope ps(c, "sample");
        // local flat index = global - gz0*plane
        flat_gather_kernel<<<(unsigned)((count + 255) / 256), 256, 0, c->stream>>>(src, -c->gz0 * plane, offset + k0 * step, step, count, stage);
        NL_CHECK_LAUNCH();
    }
    NL_HIP(hipMemcpyAsync(out, stage, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

template <int FG, int CONN>
static int run_ccl(nl_ctx *c, const uint8_t *mask, int *L, char *err, size_t errlen) {
    const i64 nrows = c->nzl * c->ny;
    const i64 waves = nrows * ((c->nx + 63) / 64);
    ccl_init_kernel<FG><<<(unsigned)((waves * 64 + 255) / 256), 256, 0, c->stream>>>(mask, L, c->nx, nrows);
    NL_CHECK_LAUNCH();
    const dim3 grid((unsigned)((c->nx + 255) / 256), (unsigned)c->ny, (unsigned)c->nzl);
    ccl_merge_kernel<FG, CONN><<<grid, 256, 0, c->stream>>>(mask, L, c->nzl, c->ny, c->nx);
    NL_CHECK_LAUNCH();
    ccl_flatten_kernel<<<grid1d(c->n), 256, 0, c->stream>>>(L, c->n);
    NL_CHECK_LAUNCH();
    return NL_OK;
}

// Voxel-level variant (first implementation): kept as the fallback for rows longer than 65535 voxels or
// pathological masks with more than N/2 runs, and as an A/B reference (NELLIE_LABEL_VOXEL=1).
static int label_run_voxels(nl_ctx *c, int has_thr, float thr, int64_t min_area, int fill_holes, int64_t *n_labels,
                            char *err, size_t errlen) {
    // buffers: frangi = f[i_vmax]; the other three float volumes serve as int32 scratch
    int free_idx[3], nf = 0;
    for (int k = 0; k < 4; ++k) if (k != c->i_vmax) free_idx[nf++] = k;
    int *L = (int *)c->f[free_idx[0]];
    int *aux = (int *)c->f[free_idx[1]];
    int *out = (int *)c->f[free_idx[2]];
    uint8_t *mA = c->m[1], *mB = c->m[2], *flag = c->m[0];
    const i64 n = c->n;
    const i64 nrows = c->nzl * c->ny;
    const i64 waves = nrows * ((c->nx + 63) / 64);
    int rc;
    ProfScope ps(c, "label");
    threshold_kernel<<<grid1d(n), 256, 0, c->stream>>>(c->f[c->i_vmax], mA, has_thr, thr, n);
    NL_CHECK_LAUNCH();
    if (fill_holes) {
        if ((rc = run_ccl<0, 6>(c, mA, L, err, errlen))) return rc;
        clear_root_flags_kernel<<<grid1d(n), 256, 0, c->stream>>>(L, flag, n);
        NL_CHECK_LAUNCH();
        border_mark_kernel<<<grid1d(n), 256, 0, c->stream>>>(L, flag, geom(c));
        NL_CHECK_LAUNCH();
        fill_holes_apply_kernel<<<grid1d(n), 256, 0, c->stream>>>(L, flag, mA, n);
        NL_CHECK_LAUNCH();
    }
    if ((rc = run_ccl<1, 26>(c, mA, L, err, errlen))) return rc;
    zero_at_roots_kernel<<<grid1d(n), 256, 0, c->stream>>>(L, aux, n);
    NL_CHECK_LAUNCH();
    area_count_kernel<<<grid1d(waves * 64, 256, 256 * 16), 256, 0, c->stream>>>(L, aux, c->nx, nrows);
    NL_CHECK_LAUNCH();
    const int ma = (int)(min_area > 0x7fffffff ? 0x7fffffff : min_area);
    keep_large_kernel<<<grid1d(n), 256, 0, c->stream>>>(L, aux, mB, ma, n);
    NL_CHECK_LAUNCH();
    {
        const dim3 grid((unsigned)((c->nx + 255) / 256), (unsigned)c->ny, (unsigned)c->nzl);
        majority_kernel<<<grid, 256, 0, c->stream>>>(mB, mA, geom(c));
        NL_CHECK_LAUNCH();
    }
    if ((rc = run_ccl<1, 26>(c, mA, L, err, errlen))) return rc;
    const i64 nblk = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    unsigned int *blk = (unsigned int *)c->d_blk;
    unsigned long long *d_total = (unsigned long long *)c->d_small;
    root_count_kernel<<<(unsigned)nblk, 256, 0, c->stream>>>(L, n, blk);
    NL_CHECK_LAUNCH();
    blk_scan_kernel<<<1, 1024, 0, c->stream>>>(blk, nblk, d_total);
    NL_CHECK_LAUNCH();
    root_assign_kernel<<<(unsigned)nblk, 256, 0, c->stream>>>(L, n, blk, aux);
    NL_CHECK_LAUNCH();
    relabel_kernel<<<grid1d(n), 256, 0, c->stream>>>(L, aux, out, n);
    NL_CHECK_LAUNCH();
    NL_HIP(hipMemcpyAsync(c->h_small, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    if (n_labels) *n_labels = (int64_t)(*(unsigned long long *)c->h_small);
    c->i_labels = free_idx[2];
    return NL_OK;
}


// exclusive scan of n u32 values (in -> out); returns nothing, total = out[n-1] + in[n-1]
static int scan_excl_u32(nl_ctx *c, const unsigned int *in, unsigned int *out, i64 n, char *err, size_t errlen) {
    const i64 nblk = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    unsigned int *blk = (unsigned int *)c->d_blk;
    unsigned long long *d_total = (unsigned long long *)c->d_small + 16;
    chunk_sum_kernel<<<(unsigned)nblk, 256, 0, c->stream>>>(in, n, blk);
    NL_CHECK_LAUNCH();
    blk_scan_kernel<<<1, 1024, 0, c->stream>>>(blk, nblk, d_total);
    NL_CHECK_LAUNCH();
    chunk_scan_kernel<<<(unsigned)nblk, 256, 0, c->stream>>>(in, out, n, blk);
    NL_CHECK_LAUNCH();
    return NL_OK;
}

struct RunSet { RunRec *runs; int *parent; unsigned int *row_off; i64 nruns; };

// Geometry the run-level Label works on: the whole (global) volume as rows of bit-packed words.
struct LabelGeo {
    i64 nz, ny, nx;            // volume the masks describe (the global one for a Z-slab run)
    i64 nrows; int wpr; i64 nwords;
    unsigned int *rows;        // 2 x (nrows + 2) u32: run counts, run offsets
    unsigned long long *bitsA, *bitsB;
    i64 paint_row0, paint_row1;   // rows this context paints ...
    int *paint_out;               // ... into this int32 buffer (row paint_row0 first)
};

// runs of `bits` (or of its complement) + union-find over them, flattened
template <int CONN>
static int build_components(nl_ctx *c, const LabelGeo &g, const unsigned long long *bits, int invert, RunSet &rs, i64 cap,
                            bool *overflow, char *err, size_t errlen) {
    unsigned int *counts = g.rows, *row_off = g.rows + (g.nrows + 2);
    NL_HIP(hipMemsetAsync(counts + g.nrows, 0, 4, c->stream));
    rl_count_kernel<<<(unsigned)((g.nrows + 255) / 256), 256, 0, c->stream>>>(bits, invert, counts, g.nrows, g.wpr, (int)g.nx);
    NL_CHECK_LAUNCH();
    int rc = scan_excl_u32(c, counts, row_off, g.nrows + 1, err, errlen);
    if (rc) return rc;
    NL_HIP(hipMemcpyAsync(c->h_small, row_off + g.nrows, 4, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    rs.nruns = (i64)(*(unsigned int *)c->h_small);
    rs.row_off = row_off;
    *overflow = rs.nruns > cap;
    if (*overflow || rs.nruns == 0) return NL_OK;
    rl_emit_kernel<<<(unsigned)((g.nrows + 255) / 256), 256, 0, c->stream>>>(bits, invert, row_off, rs.runs, rs.parent, g.nrows, g.wpr, (int)g.nx);
    NL_CHECK_LAUNCH();
    const unsigned gr = (unsigned)((rs.nruns + 255) / 256);
    rl_union_kernel<CONN><<<gr, 256, 0, c->stream>>>(rs.runs, row_off, rs.parent, rs.nruns, (int)g.ny);
    NL_CHECK_LAUNCH();
    ccl_flatten_kernel<<<grid1d(rs.nruns), 256, 0, c->stream>>>(rs.parent, rs.nruns);
    NL_CHECK_LAUNCH();
    return NL_OK;
}

// labelling.py:484-509 on a bit-packed mask (bitsA holds `frame > thr` on entry).  *overflow: more runs than scratch.
static int label_core(nl_ctx *c, const LabelGeo &g, int64_t min_area, int fill_holes, int64_t *n_labels, bool *overflow,
                      char *err, size_t errlen) {
    int free_idx[3], nf = 0;
    for (int k = 0; k < 4; ++k) if (k != c->i_vmax) free_idx[nf++] = k;
    const i64 cap = c->n / 2;                                 // runs that fit the scratch volumes
    RunSet rs;
    rs.runs = (RunRec *)c->f[free_idx[0]];                    // 8 B x cap  = 4N bytes
    rs.parent = (int *)c->f[free_idx[1]];                     // 4 B x cap  = 2N bytes
    int *aux = rs.parent + cap;                               // 4 B x cap  = 2N bytes (areas, then new ids)
    uint8_t *flag = c->m[0];
    const VolGeom vg{g.nz, g.ny, g.nx, 0, g.nz};             // boundary rules of the whole volume
    int rc;
    *overflow = false;
    if (fill_holes) {
        // binary_fill_holes: 6-connected background components that reach no face become foreground
        if ((rc = build_components<6>(c, g, g.bitsA, 1, rs, cap, overflow, err, errlen))) return rc;
        if (*overflow) return NL_OK;
        if (rs.nruns) {
            const unsigned gr = (unsigned)((rs.nruns + 255) / 256);
            NL_HIP(hipMemsetAsync(flag, 0, (size_t)rs.nruns, c->stream));
            rl_border_kernel<<<gr, 256, 0, c->stream>>>(rs.runs, rs.parent, flag, rs.nruns, vg);
            NL_CHECK_LAUNCH();
            rl_fill_kernel<<<gr, 256, 0, c->stream>>>(rs.runs, rs.parent, flag, g.bitsA, rs.nruns, g.wpr);
            NL_CHECK_LAUNCH();
        }
    }
    // first labelling + small-object removal
    if ((rc = build_components<26>(c, g, g.bitsA, 0, rs, cap, overflow, err, errlen))) return rc;
    if (*overflow) return NL_OK;
    NL_HIP(hipMemsetAsync(g.bitsB, 0, (size_t)g.nwords * 8, c->stream));
    if (rs.nruns) {
        const unsigned gr = (unsigned)((rs.nruns + 255) / 256);
        NL_HIP(hipMemsetAsync(aux, 0, (size_t)rs.nruns * 4, c->stream));
        rl_area_kernel<<<(unsigned)((rs.nruns + RL_CHUNK - 1) / RL_CHUNK), 256, 0, c->stream>>>(rs.runs, rs.parent, aux, rs.nruns);
        NL_CHECK_LAUNCH();
        const int ma = (int)(min_area > 0x7fffffff ? 0x7fffffff : min_area);
        rl_keep_kernel<<<gr, 256, 0, c->stream>>>(rs.runs, rs.parent, aux, ma, g.bitsB, rs.nruns, g.wpr);
        NL_CHECK_LAUNCH();
    }
    // majority smoothing, second labelling
    majority_bits_kernel<<<(unsigned)((g.nwords + 255) / 256), 256, 0, c->stream>>>(g.bitsB, g.bitsA, vg, g.wpr);
    NL_CHECK_LAUNCH();
    if ((rc = build_components<26>(c, g, g.bitsA, 0, rs, cap, overflow, err, errlen))) return rc;
    if (*overflow) return NL_OK;
    unsigned long long total = 0;
    if (rs.nruns) {
        const i64 nblk = (rs.nruns + SCAN_CHUNK - 1) / SCAN_CHUNK;
        unsigned int *blk = (unsigned int *)c->d_blk;
        unsigned long long *d_total = (unsigned long long *)c->d_small;
        root_count_kernel<<<(unsigned)nblk, 256, 0, c->stream>>>(rs.parent, rs.nruns, blk);
        NL_CHECK_LAUNCH();
        blk_scan_kernel<<<1, 1024, 0, c->stream>>>(blk, nblk, d_total);
        NL_CHECK_LAUNCH();
        root_assign_kernel<<<(unsigned)nblk, 256, 0, c->stream>>>(rs.parent, rs.nruns, blk, aux);
        NL_CHECK_LAUNCH();
        NL_HIP(hipMemcpyAsync(c->h_small, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    }
    rl_paint_kernel<<<grid1d((g.paint_row1 - g.paint_row0) * 64, 256, 256 * 32), 256, 0, c->stream>>>(
        g.bitsA, rs.row_off, rs.parent, aux, g.paint_out, g.paint_row0, g.paint_row1, g.wpr, (int)g.nx);
    NL_CHECK_LAUNCH();
    NL_HIP(hipStreamSynchronize(c->stream));
    if (rs.nruns) total = *(unsigned long long *)c->h_small;
    if (n_labels) *n_labels = (int64_t)total;
    c->i_labels = free_idx[2];
    return NL_OK;
}

static int label_out_index(const nl_ctx *c) {
    int last = -1;
    for (int k = 0; k < 4; ++k) if (k != c->i_vmax) last = k;
    return last;
}

extern "C" int nl_label_run(nl_ctx *c, int has_thr, float thr, int64_t min_area, int fill_holes, int64_t *n_labels,
                            char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->frangi_ready) return nl_fail(err, errlen, NL_ESTATE, "nl_label_run before a Frangi volume exists");
    if (c->nzl != c->gnz) return nl_fail(err, errlen, NL_EINVAL, "nl_label_run works on a whole volume (Z-slabs: nl_label_pack / nl_label_run_global)");
    static int force_voxel = -1;
    if (force_voxel < 0) { const char *e = getenv("NELLIE_LABEL_VOXEL"); force_voxel = (e && atoi(e)) ? 1 : 0; }
    if (force_voxel || c->nx > 65535) return label_run_voxels(c, has_thr, thr, min_area, fill_holes, n_labels, err, errlen);
    LabelGeo g;
    g.nz = c->nzl; g.ny = c->ny; g.nx = c->nx;
    g.nrows = c->nzl * c->ny; g.wpr = (int)((c->nx + 63) / 64); g.nwords = g.nrows * g.wpr;
    g.rows = c->d_rows;
    g.bitsA = (unsigned long long *)c->m[1]; g.bitsB = (unsigned long long *)c->m[2];
    g.paint_row0 = 0; g.paint_row1 = g.nrows; g.paint_out = (int *)c->f[label_out_index(c)];
    ProfScope ps(c, "label");
    rl_threshold_pack_kernel<<<grid1d(g.nwords * 64, 256, 256 * 32), 256, 0, c->stream>>>(c->f[c->i_vmax], g.bitsA, has_thr, thr,
                                                                                          (int)c->nx, g.nrows, g.wpr);
    NL_CHECK_LAUNCH();
    bool overflow = false;
    int rc = label_core(c, g, min_area, fill_holes, n_labels, &overflow, err, errlen);
    if (rc) return rc;
    if (overflow) return label_run_voxels(c, has_thr, thr, min_area, fill_holes, n_labels, err, errlen);
    return NL_OK;
}

// ---- Z-slab Label: every rank packs the mask bits of its own planes into a GLOBAL bit mask (1 bit/voxel,
// gnz*ny*nx/8 bytes), the bit planes are all-gathered, and the run-level labelling (cheap: it scales with the
// number of runs, not voxels) runs redundantly on the global mask on every rank, which then paints only its own
// planes.  Exact by construction: it IS the single-volume algorithm.
static int ensure_global_label_buffers(nl_ctx *c, char *err, size_t errlen) {
    const i64 grows = c->gnz * c->ny;
    const int wpr = (int)((c->nx + 63) / 64);
    if (!c->gbits[0]) {
        NL_HIP(hipMalloc((void **)&c->gbits[0], (size_t)grows * wpr * 8));
        NL_HIP(hipMalloc((void **)&c->gbits[1], (size_t)grows * wpr * 8));
        NL_HIP(hipMalloc((void **)&c->grows, ((size_t)grows + 2) * 2 * 4));
        if ((grows + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK + 1 > c->blk_cap) {
            hipFree(c->d_blk);
            c->blk_cap = (grows + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK + 1;
            NL_HIP(hipMalloc(&c->d_blk, (size_t)c->blk_cap * 4));
        }
    }
    return NL_OK;
}

extern "C" int nl_label_pack(nl_ctx *c, int has_thr, float thr, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->frangi_ready) return nl_fail(err, errlen, NL_ESTATE, "nl_label_pack before a Frangi volume exists");
    if (c->nx > 65535) return nl_fail(err, errlen, NL_EINVAL, "rows longer than 65535 voxels are not supported on Z-slabs");
    int rc = ensure_global_label_buffers(c, err, errlen);
    if (rc) return rc;
    const int wpr = (int)((c->nx + 63) / 64);
    const i64 own_rows = (c->own_hi - c->own_lo) * c->ny;
    const i64 row0 = (c->gz0 + c->own_lo) * c->ny;
    ProfScope ps(c, "label");
    rl_threshold_pack_kernel<<<grid1d(own_rows * wpr * 64, 256, 256 * 32), 256, 0, c->stream>>>(
        c->f[c->i_vmax] + c->own_lo * c->ny * c->nx, c->gbits[0] + row0 * wpr, has_thr, thr, (int)c->nx, own_rows, wpr);
    NL_CHECK_LAUNCH();
    return NL_OK;
}

// host access to bit-mask rows [row0, row0+nrows) of the global mask (tests / communicators without RCCL)
extern "C" int nl_label_bits_get(nl_ctx *c, int64_t row0, int64_t nrows, uint64_t *host, char *err, size_t errlen) {
    NL_ENTER(c);
    const int wpr = (int)((c->nx + 63) / 64);
    if (!c->gbits[0] || !host || row0 < 0 || nrows < 1 || row0 + nrows > c->gnz * c->ny) return nl_fail(err, errlen, NL_EINVAL, "bad bit-mask row range");
    NL_HIP(hipMemcpyAsync(host, c->gbits[0] + row0 * wpr, (size_t)nrows * wpr * 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}
extern "C" int nl_label_bits_put(nl_ctx *c, int64_t row0, int64_t nrows, const uint64_t *host, char *err, size_t errlen) {
    NL_ENTER(c);
    const int wpr = (int)((c->nx + 63) / 64);
    if (!c->gbits[0] || !host || row0 < 0 || nrows < 1 || row0 + nrows > c->gnz * c->ny) return nl_fail(err, errlen, NL_EINVAL, "bad bit-mask row range");
    NL_HIP(hipMemcpyAsync(c->gbits[0] + row0 * wpr, host, (size_t)nrows * wpr * 8, hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

// all-gather of the mask bit planes over RCCL: rank r broadcasts the rows of its own planes (slab_plane0[r] ..)
extern "C" int nl_label_bits_allgather(nl_ctx *c, const int64_t *slab_plane0, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->comm) return nl_fail(err, errlen, NL_ESTATE, "nl_label_bits_allgather before nl_comm_init");
    if (!c->gbits[0] || !slab_plane0) return nl_fail(err, errlen, NL_ESTATE, "nl_label_bits_allgather before nl_label_pack");
    const int wpr = (int)((c->nx + 63) / 64);
    ProfScope ps(c, "halo");
    NL_NCCL(ncclGroupStart());
    for (int r = 0; r < c->world; ++r) {
        const i64 p0 = slab_plane0[r], p1 = slab_plane0[r + 1];        // world + 1 entries, last = gnz
        unsigned long long *ptr = c->gbits[0] + p0 * c->ny * wpr;
        NL_NCCL(ncclBroadcast(ptr, ptr, (size_t)((p1 - p0) * c->ny * wpr), ncclUint64, r, (ncclComm_t)c->comm, c->stream));
    }
    NL_NCCL(ncclGroupEnd());
    return NL_OK;
}

extern "C" int nl_label_run_global(nl_ctx *c, int64_t min_area, int fill_holes, int64_t *n_labels, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->gbits[0]) return nl_fail(err, errlen, NL_ESTATE, "nl_label_run_global before nl_label_pack");
    LabelGeo g;
    g.nz = c->gnz; g.ny = c->ny; g.nx = c->nx;
    g.nrows = c->gnz * c->ny; g.wpr = (int)((c->nx + 63) / 64); g.nwords = g.nrows * g.wpr;
    g.rows = c->grows;
    g.bitsA = c->gbits[0]; g.bitsB = c->gbits[1];
    g.paint_row0 = (c->gz0 + c->own_lo) * c->ny; g.paint_row1 = (c->gz0 + c->own_hi) * c->ny;
    g.paint_out = (int *)c->f[label_out_index(c)] + c->own_lo * c->ny * c->nx;
    ProfScope ps(c, "label");
    bool overflow = false;
    int rc = label_core(c, g, min_area, fill_holes, n_labels, &overflow, err, errlen);
    if (rc) return rc;
    if (overflow) return nl_fail(err, errlen, NL_ENOMEM, "the global mask has more runs than this slab's scratch volumes hold [out of memory]");
    return NL_OK;
}

extern "C" int nl_label_store(nl_ctx *c, int32_t *host, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->i_labels < 0) return nl_fail(err, errlen, NL_ESTATE, "nl_label_store before nl_label_run");
    return store_planes(c, c->f[c->i_labels], host, 4, z0, z1, err, errlen);
}

// ---------------------------------------------------------------------------------- debug -------
extern "C" int nl_ctx_info(nl_ctx *c, const char *key, double *value) {
    if (!c || !key || !value) return NL_EINVAL;
    if (!strcmp(key, "fast_div")) *value = c->fast_div;
    else if (!strcmp(key, "hessian_tile_rows")) *value = hm_ty();
    else if (!strcmp(key, "device_bytes")) *value = (double)nl_ctx_bytes(c->nzl, c->ny, c->nx);
    else return NL_EINVAL;
    return NL_OK;
}

extern "C" int nl_debug_eig_frangi(nl_ctx *c, const float *h6, int64_t n, int impl, float alpha_sq, float beta_sq,
                                   float gamma_sq, float *out4, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!h6 || !out4 || n < 1 || n * 6 > c->n) return nl_fail(err, errlen, NL_EINVAL, "bad debug batch (n=%lld)", (i64)n);
    float *d_in = c->f[(c->i_gauss + 1) % 3], *d_out = c->f[(c->i_gauss + 2) % 3];
    NL_HIP(hipMemcpyAsync(d_in, h6, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
    debug_eig_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(d_in, n, impl, alpha_sq, beta_sq, gamma_sq, d_out);
    NL_CHECK_LAUNCH();
    NL_HIP(hipMemcpyAsync(out4, d_out, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

// --------------------------------------------------------------------------------- timing -------
extern "C" int nl_timer_begin(nl_ctx *c, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_HIP(hipEventRecord(c->t0, c->stream));
    return NL_OK;
}
extern "C" int nl_timer_end_ms(nl_ctx *c, float *ms, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_HIP(hipEventRecord(c->t1, c->stream));
    NL_HIP(hipEventSynchronize(c->t1));
    float t = 0;
    NL_HIP(hipEventElapsedTime(&t, c->t0, c->t1));
    if (ms) *ms = t;
    return NL_OK;
}
extern "C" int nl_prof_enable(nl_ctx *c, int on) { if (c) c->prof_on = on; return NL_OK; }
extern "C" int nl_prof_reset(nl_ctx *c) {
    if (!c) return NL_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (auto &kv : c->prof) for (auto &r : kv.second) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    c->prof.clear();
    return NL_OK;
}
extern "C" int nl_prof_get(nl_ctx *c, const char *name, double *ms, int64_t *launches) {
    if (!c || !name) return NL_EINVAL;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    double tot = 0; int64_t k = 0;
    auto it = c->prof.find(name);
    if (it != c->prof.end())
        for (auto &r : it->second) { float t = 0; if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { tot += t; ++k; } }
    if (ms) *ms = tot;
    if (launches) *launches = k;
    return NL_OK;
}
