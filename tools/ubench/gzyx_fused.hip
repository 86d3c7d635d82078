// Micro-benchmark (round 5): ONE kernel for a whole cascade step -- Z, Y and X Gaussian passes fused -- against
// the two kernels the library runs today (gauss_march_z2_kernel + gauss_yx_tile_kernel), same scipy arithmetic, bit-compared.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -o tools/ubench/gzyx_fused tools/ubench/gzyx_fused.hip
//   tools/ubench/gzyx_fused [nz ny nx [R]]        (default 1024 1024 1024, R = 4)
//
// Design of the fused kernel.  A workgroup owns TY x TX output columns and marches along Z over a chunk of planes.  Per plane:
//   (1) every thread advances the Z window of ITS columns of the halo'd footprint (TY + 2R) x (TX + 2R): the 2R+1 window lives in
//       registers (float64, or float32 converted at use: WIN64), the next plane's value is already in flight; the Z-filtered value is
//       rounded to float32 exactly like the stand-alone pass and dropped into an LDS tile;
//   (2) barrier; the Y pass runs out of that tile, register-blocked GY outputs per thread, result (float32) into a second LDS tile;
//   (3) barrier; the X pass runs out of the second tile, four outputs per thread, one float4 store.
// One read and one write of the volume per cascade step (plus the halo columns, which neighbouring workgroups read again: L2 /
// Infinity Cache hits mostly) instead of two of each; the price is the redundant Z and Y work on the halo and one more
// float32 -> float64 conversion per tap where the window is kept in float32.
#include "../../nellie_amd/csrc/nl_common.h"
#include <type_traits>
#include <vector>
#include <stdlib.h>
#include "../../nellie_amd/csrc/device_math.inc"
#include "../../nellie_amd/csrc/gauss.inc"
#include "../../nellie_amd/csrc/gauss_zyx.inc"

// float32 window: the conversion is repeated at every use -- unless the compiler sees that phase ph+1 converts the value phase ph
// converted and keeps the float64 copy alive as well (then the window costs 3 registers per value instead of 1): the empty asm makes
// each conversion's source opaque.
__device__ __forceinline__ double cvt_opaque(float a) { asm volatile("" : "+v"(a)); return (double)a; }
__device__ __forceinline__ double cvt_opaque(double a) { return a; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// SKIP (ablation, results then wrong): 1 no global loads in the march, 2 no Z arithmetic, 4 no Y stage, 8 no X arithmetic, 16 no stores
template <int RZ, int R, int TY, int TX, int NT, bool WIN64, int GY, int SKIP = 0>
__global__ void __launch_bounds__(NT)
gauss_zyx_v1_kernel(const float *__restrict__ in, float *__restrict__ out, VolGeom v, int z0, int z1, int zchunk, GaussWS gz, GaussWS gyx,
                 int ntx, int nty) {
    constexpr int WZ = 2 * RZ + 1, HY = TY + 2 * R, HX = TX + 2 * R;
    constexpr int NCOL = HY * HX, CPT = (NCOL + NT - 1) / NT;
    constexpr int YP = HX + 8;                                      // pitch of the Y-filtered tile (bank spread of the float4 reads)
    constexpr int NYT = HX * (TY / GY);                             // Y-pass tasks: (column, group of GY rows)
    constexpr int XG = TX / 4, NXT = TY * XG;                       // X-pass tasks: (row, group of 4 outputs)
    constexpr int NV = (4 + 2 * R + 3) / 4;
    static_assert(TY % GY == 0 && TX % 4 == 0, "tile shape");
    typedef typename std::conditional<WIN64, double, float>::type WT;
    __shared__ float zt[NCOL];
    __shared__ __attribute__((aligned(16))) float yt[TY * YP + 16];
    const int t = threadIdx.x;
    const unsigned nblk = gridDim.x;
    unsigned bid = blockIdx.x;
    if ((nblk & 7u) == 0u) bid = (bid & 7u) * (nblk >> 3) + (bid >> 3);       // contiguous runs of tiles per XCD
    const int bx = (int)(bid % (unsigned)ntx), by = (int)((bid / (unsigned)ntx) % (unsigned)nty), bz = (int)(bid / ((unsigned)ntx * (unsigned)nty));
    const int x0 = bx * TX, y0 = by * TY;
    const int c0 = z0 + bz * zchunk;
    const int c1 = c0 + zchunk < z1 ? c0 + zchunk : z1;
    const int nx = (int)v.nx, ny = (int)v.ny;
    const i64 sz = (i64)ny * nx;
    const int n_line = (int)v.gnz, goff = (int)v.gz0;
    int off[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        int q = t + k * NT;
        if (q >= NCOL) q = NCOL - 1;                                // duplicate column: cheaper than a branch around the loads
        const int row = q / HX, col = q - row * HX;
        off[k] = reflect_once(y0 - R + row, ny) * nx + reflect_once(x0 - R + col, nx);
    }
    auto ld = [&](int k, int p) -> float { return in[(i64)(reflect_once(goff + p, n_line) - goff) * sz + off[k]]; };
    WT w[CPT][WZ];
    float nxt[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
#pragma unroll
        for (int s = 0; s < WZ - 1; ++s) w[k][s + 1] = (WT)ld(k, c0 - RZ + s);   // slots 1..WZ-1 hold c0-RZ .. c0+RZ-1
        nxt[k] = ld(k, c0 + RZ);
    }
    const int last = c1 - 1 + RZ;
#define ZYX_STEP(ph, p)                                                                                       \
    {                                                                                                         \
        _Pragma("unroll") for (int k = 0; k < CPT; ++k) {                                                     \
            w[k][ph] = (WT)nxt[k];                                                                            \
            if (!(SKIP & 1)) nxt[k] = ld(k, (p) + RZ + 1 < last ? (p) + RZ + 1 : last);                        \
            double tmp = cvt_opaque(w[k][((ph) + 1 + RZ) % WZ]) * gz.w[0];                                    \
            if (!(SKIP & 2)) { _Pragma("unroll") for (int j = RZ; j >= 1; --j) {                              \
                const double s_ = cvt_opaque(w[k][((ph) + 1 + RZ - j) % WZ]) + cvt_opaque(w[k][((ph) + 1 + RZ + j) % WZ]); \
                tmp = tmp + s_ * gz.w[j];                                                                     \
            } }                                                                                               \
            if (k < CPT - 1 || t + k * NT < NCOL) zt[t + k * NT] = (float)tmp;                                \
        }                                                                                                     \
        __syncthreads();                                                                                      \
        if (!(SKIP & 4)) for (int task = t; task < NYT; task += NT) {                                         \
            const int g = task / HX, col = task - g * HX;                                                     \
            double d[GY + 2 * R];                                                                             \
            _Pragma("unroll") for (int i = 0; i < GY + 2 * R; ++i) d[i] = (double)zt[(g * GY + i) * HX + col]; \
            _Pragma("unroll") for (int o = 0; o < GY; ++o) {                                                  \
                double t2 = d[o + R] * gyx.w[0];                                                              \
                _Pragma("unroll") for (int j = R; j >= 1; --j) t2 = t2 + (d[o + R - j] + d[o + R + j]) * gyx.w[j]; \
                yt[(g * GY + o) * YP + col] = (float)t2;                                                      \
            }                                                                                                 \
        }                                                                                                     \
        __syncthreads();                                                                                      \
        for (int task = t; task < NXT; task += NT) {                                                          \
            const int row = task / XG, g = task - row * XG;                                                   \
            const int y = y0 + row, xo = x0 + 4 * g;                                                          \
            double d[4 * NV];                                                                                 \
            _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                                  \
                const float4 q4 = *reinterpret_cast<const float4 *>(&yt[row * YP + 4 * g + 4 * i]);           \
                d[4 * i] = (double)q4.x; d[4 * i + 1] = (double)q4.y; d[4 * i + 2] = (double)q4.z; d[4 * i + 3] = (double)q4.w; \
            }                                                                                                 \
            float res[4];                                                                                     \
            _Pragma("unroll") for (int o = 0; o < 4; ++o) {                                                   \
                double t3 = d[o + R] * gyx.w[0];                                                              \
                if (!(SKIP & 8)) { _Pragma("unroll") for (int j = R; j >= 1; --j) t3 = t3 + (d[o + R - j] + d[o + R + j]) * gyx.w[j]; } \
                res[o] = (float)t3;                                                                           \
            }                                                                                                 \
            if (y < ny && xo < nx && (!(SKIP & 16) || res[0] == 123.456f)) {                                  \
                float *op = out + (i64)(p) * sz + (i64)y * nx + xo;                                           \
                if (xo + 3 < nx) *reinterpret_cast<float4 *>(op) = make_float4(res[0], res[1], res[2], res[3]); \
                else { _Pragma("unroll") for (int o = 0; o < 4; ++o) if (xo + o < nx) op[o] = res[o]; }       \
            }                                                                                                 \
        }                                                                                                     \
    }
    int p0 = c0;
    for (; p0 + WZ <= c1; p0 += WZ) {
#pragma unroll
        for (int ph = 0; ph < WZ; ++ph) ZYX_STEP(ph, p0 + ph)
    }
#pragma unroll
    for (int ph = 0; ph < WZ; ++ph) {
        if (p0 + ph < c1) ZYX_STEP(ph, p0 + ph)                     // uniform
    }
#undef ZYX_STEP
}


// Version 2: the three stages of a plane software-pipelined across planes -- in one barrier interval the workgroup runs the Z stage of
// plane i, the Y stage of plane i-1 and the X stage of plane i-2 on double-buffered LDS tiles: ONE barrier per plane instead of two,
// and three independent instruction streams for the scheduler to interleave (LDS latency of one under the arithmetic of another).
template <int RZ, int R, int TY, int TX, int NT, bool WIN64, int GY>
__global__ void __launch_bounds__(NT)
gauss_zyx2_kernel(const float *__restrict__ in, float *__restrict__ out, VolGeom v, int z0, int z1, int zchunk, GaussWS gz, GaussWS gyx,
                  int ntx, int nty) {
    constexpr int WZ = 2 * RZ + 1, HY = TY + 2 * R, HX = TX + 2 * R;
    constexpr int NCOL = HY * HX, CPT = (NCOL + NT - 1) / NT;
    constexpr int YP = HX + 8;
    constexpr int NYT = HX * (TY / GY);
    constexpr int XG = TX / 4, NXT = TY * XG;
    constexpr int NV = (4 + 2 * R + 3) / 4;
    constexpr int YTS = TY * YP + 16;
    static_assert(TY % GY == 0 && TX % 4 == 0, "tile shape");
    typedef typename std::conditional<WIN64, double, float>::type WT;
    __shared__ float zt[2 * NCOL];
    __shared__ __attribute__((aligned(16))) float yt[2 * YTS];
    const int t = threadIdx.x;
    const unsigned nblk = gridDim.x;
    unsigned bid = blockIdx.x;
    if ((nblk & 7u) == 0u) bid = (bid & 7u) * (nblk >> 3) + (bid >> 3);
    const int bx = (int)(bid % (unsigned)ntx), by = (int)((bid / (unsigned)ntx) % (unsigned)nty), bz = (int)(bid / ((unsigned)ntx * (unsigned)nty));
    const int x0 = bx * TX, y0 = by * TY;
    const int c0 = z0 + bz * zchunk;
    const int c1 = c0 + zchunk < z1 ? c0 + zchunk : z1;
    const int nx = (int)v.nx, ny = (int)v.ny;
    const i64 sz = (i64)ny * nx;
    const int n_line = (int)v.gnz, goff = (int)v.gz0;
    int off[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        int q = t + k * NT;
        if (q >= NCOL) q = NCOL - 1;
        const int row = q / HX, col = q - row * HX;
        off[k] = reflect_once(y0 - R + row, ny) * nx + reflect_once(x0 - R + col, nx);
    }
    auto ld = [&](int k, int p) -> float { return in[(i64)(reflect_once(goff + p, n_line) - goff) * sz + off[k]]; };
    WT w[CPT][WZ];
    float nxt[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
#pragma unroll
        for (int s = 0; s < WZ - 1; ++s) w[k][s + 1] = (WT)ld(k, c0 - RZ + s);
        nxt[k] = ld(k, c0 + RZ);
    }
    const int last = c1 - 1 + RZ;
    // the Y task and the X task of this thread are the same every plane
    constexpr int XT = (NXT + NT - 1) / NT;                        // X tasks per thread
    const bool ytask = t < NYT;
    static_assert(NYT <= NT, "one Y task per thread at most");
    const int yg = t / HX, ycol = t - yg * HX;
    int xlds[XT], xn[XT]; bool xst[XT]; i64 xoff[XT];
    const bool vec4 = (nx & 3) == 0;
#pragma unroll
    for (int u = 0; u < XT; ++u) {
        int task = t + u * NT;
        const bool valid = task < NXT;
        if (!valid) task = NXT - 1;
        const int xrow = task / XG, xg = task - xrow * XG;
        const int xy = y0 + xrow, xo = x0 + 4 * xg;
        xlds[u] = xrow * YP + 4 * xg;
        xst[u] = valid && xy < ny && xo < nx;
        xn[u] = nx - xo;
        xoff[u] = (i64)xy * nx + xo;
    }
#define ZYX2_STEP(ph, i)                                                                                      \
    {                                                                                                         \
        const int p = (i);                                                                                    \
        if (p < c1) {                                                                                         \
            float *ztw = zt + (((p) - c0) & 1) * NCOL;                                                        \
            _Pragma("unroll") for (int k = 0; k < CPT; ++k) {                                                 \
                w[k][ph] = (WT)nxt[k];                                                                        \
                nxt[k] = ld(k, p + RZ + 1 < last ? p + RZ + 1 : last);                                        \
                double tmp = cvt_opaque(w[k][((ph) + 1 + RZ) % WZ]) * gz.w[0];                                \
                _Pragma("unroll") for (int j = RZ; j >= 1; --j) {                                             \
                    const double s_ = cvt_opaque(w[k][((ph) + 1 + RZ - j) % WZ]) + cvt_opaque(w[k][((ph) + 1 + RZ + j) % WZ]); \
                    tmp = tmp + s_ * gz.w[j];                                                                 \
                }                                                                                             \
                if (k < CPT - 1 || t + k * NT < NCOL) ztw[t + k * NT] = (float)tmp;                           \
            }                                                                                                 \
        }                                                                                                     \
        if (p - 1 >= c0 && p - 1 < c1 && ytask) {                                                             \
            const float *ztr = zt + (((p) - 1 - c0) & 1) * NCOL;                                              \
            float *ytw = yt + (((p) - 1 - c0) & 1) * YTS;                                                     \
            double d[GY + 2 * R];                                                                             \
            _Pragma("unroll") for (int q = 0; q < GY + 2 * R; ++q) d[q] = (double)ztr[(yg * GY + q) * HX + ycol]; \
            _Pragma("unroll") for (int o = 0; o < GY; ++o) {                                                  \
                double t2 = d[o + R] * gyx.w[0];                                                              \
                _Pragma("unroll") for (int j = R; j >= 1; --j) t2 = t2 + (d[o + R - j] + d[o + R + j]) * gyx.w[j]; \
                ytw[(yg * GY + o) * YP + ycol] = (float)t2;                                                   \
            }                                                                                                 \
        }                                                                                                     \
        if (p - 2 >= c0 && p - 2 < c1) {                                                                      \
            const float *ytr = yt + (((p) - 2 - c0) & 1) * YTS;                                               \
            _Pragma("unroll") for (int u = 0; u < XT; ++u) {                                                  \
                if (XT * NT == NXT || u < XT - 1 || t + u * NT < NXT) {                                       \
                    double d[4 * NV];                                                                         \
                    _Pragma("unroll") for (int q = 0; q < NV; ++q) {                                          \
                        const float4 q4 = *reinterpret_cast<const float4 *>(&ytr[xlds[u] + 4 * q]);           \
                        d[4 * q] = (double)q4.x; d[4 * q + 1] = (double)q4.y; d[4 * q + 2] = (double)q4.z; d[4 * q + 3] = (double)q4.w; \
                    }                                                                                         \
                    float res[4];                                                                             \
                    _Pragma("unroll") for (int o = 0; o < 4; ++o) {                                           \
                        double t3 = d[o + R] * gyx.w[0];                                                      \
                        _Pragma("unroll") for (int j = R; j >= 1; --j) t3 = t3 + (d[o + R - j] + d[o + R + j]) * gyx.w[j]; \
                        res[o] = (float)t3;                                                                   \
                    }                                                                                         \
                    if (xst[u]) {                                                                             \
                        float *op = out + (i64)(p - 2) * sz + xoff[u];                                        \
                        if (vec4 && xn[u] >= 4) __builtin_nontemporal_store(gm_v4f{res[0], res[1], res[2], res[3]}, reinterpret_cast<gm_v4f *>(op)); \
                        else { _Pragma("unroll") for (int o = 0; o < 4; ++o) if (o < xn[u]) op[o] = res[o]; } \
                    }                                                                                         \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
        __syncthreads();                                                                                      \
    }
    int p0 = c0;
    for (; p0 + WZ <= c1 + 2; p0 += WZ) {
#pragma unroll
        for (int ph = 0; ph < WZ; ++ph) ZYX2_STEP(ph, p0 + ph)
    }
#pragma unroll
    for (int ph = 0; ph < WZ; ++ph) {
        if (p0 + ph < c1 + 2) ZYX2_STEP(ph, p0 + ph)                // uniform
    }
#undef ZYX2_STEP
}

__global__ void diff_kernel(const unsigned int *a, const unsigned int *b, i64 n, unsigned long long *cnt) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (i64)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(cnt, c);
}

static void weights(double sigma, int r, GaussWS &ws) {           // scipy _gaussian_kernel1d, order 0
    std::vector<double> w(2 * r + 1);
    double sum = 0;
    for (int x = -r; x <= r; ++x) { w[x + r] = exp(-0.5 / (sigma * sigma) * x * x); sum += w[x + r]; }
    for (int k = 0; k <= GM_MAX_R; ++k) ws.w[k] = k <= r ? w[r + k] / sum : 0.0;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <typename F> float run(F f, int reps) {
        f(); CK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0;
        for (int i = 0; i < reps; ++i) {
            CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); sum += ms; if (ms < best) best = ms;
        }
        last_avg = sum / reps;
        return best;
    }
    float last_avg = 0;
};

template <int VER, int R, int TY, int TX, int NT, bool WIN64, int GY, int SKIP = 0>
static void bench_fused(const char *name, const float *in, float *out, const float *ref, const VolGeom &v, int zchunk, const GaussWS &gz,
                        const GaussWS &gyx, unsigned long long *d_cnt, Timer &tm) {
    const int ntx = (int)((v.nx + TX - 1) / TX), nty = (int)((v.ny + TY - 1) / TY), nzc = (int)((v.nzl + zchunk - 1) / zchunk);
    const unsigned nb = (unsigned)ntx * nty * nzc;
    CK(hipMemset(out, 0xff, (size_t)v.nzl * v.ny * v.nx * 4));
    auto f = [&]() {
        if (VER == 2) gauss_zyx2_kernel<R, R, TY, TX, NT, WIN64, GY><<<nb, NT>>>(in, out, v, 0, (int)v.nzl, zchunk, gz, gyx, ntx, nty);
        else gauss_zyx_v1_kernel<R, R, TY, TX, NT, WIN64, GY, SKIP><<<nb, NT>>>(in, out, v, 0, (int)v.nzl, zchunk, gz, gyx, ntx, nty);
    };
    const float best = tm.run(f, 5);
    CK(hipGetLastError());
    CK(hipMemset(d_cnt, 0, 8));
    const i64 n = v.nzl * v.ny * v.nx;
    diff_kernel<<<4096, 256>>>((const unsigned int *)out, (const unsigned int *)ref, n, d_cnt);
    unsigned long long bad = 0;
    CK(hipMemcpy(&bad, d_cnt, 8, hipMemcpyDeviceToHost));
    hipFuncAttributes fa;
    if (VER == 2) CK(hipFuncGetAttributes(&fa, (const void *)gauss_zyx2_kernel<R, R, TY, TX, NT, WIN64, GY>));
    else CK(hipFuncGetAttributes(&fa, (const void *)gauss_zyx_v1_kernel<R, R, TY, TX, NT, WIN64, GY, SKIP>));
    printf("%-44s zchunk %4d  %7.3f ms best %7.3f avg  %5.2f TB/s(8 B/voxel)  vgpr %3d lds %6zu  %s (%llu words differ)\n", name, zchunk, best, tm.last_avg,
           8.0 * n / best / 1e9, fa.numRegs, (size_t)fa.sharedSizeBytes, bad ? "MISMATCH" : "bit-exact", bad);
    fflush(stdout);
}

template <int R>
static void bench_lib(const float *in, float *out, const float *ref, const VolGeom &v, int zchunk, const GaussWS &gz, const GaussWS &gyx,
                      unsigned long long *d_cnt, Timer &tm) {
    constexpr int TY = GzyxCfg<R>::TY;
    const int ntx = (int)((v.nx + 63) / 64), nty = (int)((v.ny + TY - 1) / TY), nzc = (int)((v.nzl + zchunk - 1) / zchunk);
    const unsigned nb = (unsigned)ntx * nty * nzc;
    CK(hipMemset(out, 0xff, (size_t)v.nzl * v.ny * v.nx * 4));
    auto f = [&]() { gauss_zyx_kernel<R, R><<<nb, GZ_NT>>>(in, out, v, 0, (int)v.nzl, zchunk, gz, gyx, ntx, nty, nullptr); };
    const float best = tm.run(f, 5);
    CK(hipGetLastError());
    CK(hipMemset(d_cnt, 0, 8));
    const i64 n = v.nzl * v.ny * v.nx;
    diff_kernel<<<4096, 256>>>((const unsigned int *)out, (const unsigned int *)ref, n, d_cnt);
    unsigned long long bad = 0;
    CK(hipMemcpy(&bad, d_cnt, 8, hipMemcpyDeviceToHost));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void *)gauss_zyx_kernel<R, R>));
    char name[96];
    snprintf(name, sizeof name, "LIBRARY gauss_zyx_kernel %dx64 tile 512 thr", TY);
    printf("%-44s zchunk %4d  %7.3f ms best %7.3f avg  %5.2f TB/s(8 B/voxel)  vgpr %3d lds %6zu  %s (%llu words differ)\n", name, zchunk, best, tm.last_avg,
           8.0 * n / best / 1e9, fa.numRegs, (size_t)fa.sharedSizeBytes, bad ? "MISMATCH" : "bit-exact", bad);
    fflush(stdout);
}

template <int R>
static void run_all(int nz, int ny, int nx) {
    const i64 n = (i64)nz * ny * nx;
    float *in, *mid, *ref, *out;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&mid, n * 4)); CK(hipMalloc(&ref, n * 4)); CK(hipMalloc(&out, n * 4));
    {
        std::vector<float> h((size_t)ny * nx);
        unsigned s = 12345u;
        for (int z = 0; z < nz; ++z) {                                 // noise + a gradient: every plane differs
            for (auto &x : h) { s = s * 1664525u + 1013904223u; x = 100.0f + (float)(s >> 8) * (1.0f / 16777216.0f) * 40.0f + 0.01f * z; }
            CK(hipMemcpy(in + (i64)z * ny * nx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        }
    }
    VolGeom v{nz, ny, nx, 0, nz};
    v.chunk = n >= ((i64)1 << 29) ? 256 : 128;
    GaussWS gz, gyx;
    const double sigma = R / 3.0 - 0.05;
    weights(sigma, R, gz); weights(sigma, R, gyx);
    unsigned long long *d_cnt; CK(hipMalloc(&d_cnt, 8));
    Timer tm;
    printf("# volume %d x %d x %d float32, cascade radius %d on all three axes (MI355X, tools/ubench/gzyx_fused.hip)\n", nz, ny, nx, R);
    // today's two kernels
    {
        dim3 gzg((unsigned)((nx / 2 + 63) / 64), (unsigned)((ny + 3) / 4), (unsigned)((nz + v.chunk - 1) / v.chunk));
        auto fz = [&]() { gauss_march_z2_kernel<(R <= 6 ? R : 1)><<<gzg, 256>>>(in, mid, v, 0, nz, gz); };
        const unsigned ntx = (unsigned)((nx + GYX_COLS - 1) / GYX_COLS), nty = (unsigned)((ny + v.chunk - 1) / v.chunk);
        auto fyx = [&]() { gauss_yx_tile_kernel<R, false><<<ntx * nty * nz, GYX_THREADS>>>(mid, ref, v, 0, nz, gyx, gyx, (nx % 4 == 0) ? 1 : 0, (int)ntx, (int)nty); };
        const float tz = tm.run(fz, 5), tza = tm.last_avg;
        const float tyx = tm.run(fyx, 5), tyxa = tm.last_avg;
        auto both = [&]() { fz(); fyx(); };
        const float tb = tm.run(both, 5);
        printf("%-44s              %7.3f ms best %7.3f avg\n", "gauss_march_z2_kernel (Z)", tz, tza);
        printf("%-44s              %7.3f ms best %7.3f avg\n", "gauss_yx_tile_kernel (Y+X)", tyx, tyxa);
        printf("%-44s              %7.3f ms best %7.3f avg  <- the cascade step today\n", "Z then Y+X, back to back", tb, tm.last_avg);
        CK(hipGetLastError());
    }
    for (int zc : {64, 128, 256}) bench_lib<R>(in, out, ref, v, zc, gz, gyx, d_cnt, tm);
    const bool ablate = getenv("GZYX_ABLATE") != nullptr;
    if (getenv("GZYX_LIB_ONLY")) return;
    for (int zc : {128}) {
        if (ablate) {
#define AB(SK, txt) bench_fused<1, R, 64, ((64 - 2 * R) & ~3), 512, true, 8, SK>("v1 f64 64x56 512thr  " txt, in, out, ref, v, zc, gz, gyx, d_cnt, tm); \
                    bench_fused<1, R, 32, ((64 - 2 * R) & ~3), 512, false, 4, SK>("v1 f32 32x56 512thr  " txt, in, out, ref, v, zc, gz, gyx, d_cnt, tm);
            AB(0, "full") AB(1, "no global loads") AB(2, "no Z arithmetic") AB(4, "no Y stage") AB(8, "no X arithmetic") AB(16, "no stores")
            AB(2 + 4 + 8, "loads + LDS + stores only") AB(1 + 16, "no loads, no stores") AB(1 + 2 + 4 + 8 + 16, "skeleton: barriers, LDS")
#undef AB
            continue;
        }
        bench_fused<1, R, 32, ((64 - 2 * R) & ~3), 512, false, 4>("v1 f32 window 32x(64-2R) 512 thr GY4", in, out, ref, v, zc, gz, gyx, d_cnt, tm);
        bench_fused<1, R, 64, ((64 - 2 * R) & ~3), 512, true, 8>("v1 f64 window 64x(64-2R) 512 thr GY8", in, out, ref, v, zc, gz, gyx, d_cnt, tm);
        bench_fused<2, R, 64, ((64 - 2 * R) & ~3), 512, true, 8>("v2 f64 window 64x(64-2R) 512 thr GY8", in, out, ref, v, zc, gz, gyx, d_cnt, tm);
    }
    hipFree(in); hipFree(mid); hipFree(ref); hipFree(out); hipFree(d_cnt);
}

int main(int argc, char **argv) {
    const int nz = argc > 3 ? atoi(argv[1]) : 1024, ny = argc > 3 ? atoi(argv[2]) : 1024, nx = argc > 3 ? atoi(argv[3]) : 1024;
    const int r = argc > 4 ? atoi(argv[4]) : 4;
    if (r == 3) run_all<3>(nz, ny, nx);
    else if (r == 5) run_all<5>(nz, ny, nx);
    else run_all<4>(nz, ny, nx);
    return 0;
}
