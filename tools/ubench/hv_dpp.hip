// Micro-benchmark (round 6, VERDICT r05 "next 1"): the Hessian walk re-decomposed WITHOUT LDS and WITHOUT the per-plane barrier, against the
// production kernel (hessian_pair.inc: raw planes and first derivatives through an LDS ring, one s_barrier per plane) doing the same work.
//
// The skeleton: a WAVE owns a strip of SR rows x 60 valid columns (lanes 2 .. 61 of the 64 it loads: the X neighbours of the first
// derivatives come from lanes +-1 through DPP wave shifts, so two lanes per side only carry halo) and marches along Z by itself:
//   * raw planes z .. z+3 of the strip (rows -2 .. SR+1) live in registers (a four-deep ring, the plane entering next in flight);
//   * first derivatives of plane z+1: d/dz and d/dy on rows -1 .. SR (Y neighbours of the second stage: REGISTERS of the same lane, paid
//     for with the two halo rows' divisions), d/dx on rows 0 .. SR-1 from the raw X neighbours (DPP);
//   * Hessian of plane z: Z from a register ring of d/dz, Y from the neighbouring rows' registers, X through DPP; frob_sq; the statistics
//     of the production kernel's MODE 0 (max |H|, max frob_sq) per lane, reduced at the end.
// Same float32 operations in the same order as the production kernel on interior voxels (checked here: frob_sq of a small volume bit for
// bit against a plain one-thread-per-voxel kernel).  It is a SKELETON: faces of the volume, masks, the eigen queue are not in it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize [-mllvm -amdgpu-sched-strategy=max-ilp] -o tools/ubench/hv_dpp tools/ubench/hv_dpp.hip
//   tools/ubench/hv_dpp [nz ny nx] [reps]          (default 512 1024 1024, 20)
#include "../../nellie_amd/csrc/nellie_hv.hip"       // the production walk and its launcher, as the library builds them
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// value of lane - 1 / lane + 1 (wave shifts, gfx9 DPP; the lane without a source gets 0: it is a halo lane, its results are dropped)
__device__ __forceinline__ float from_left(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true)); }    // wave_shr:1
__device__ __forceinline__ float from_right(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true)); }   // wave_shl:1
__device__ __forceinline__ v2f dx_of(v2f a) { return v2f{from_right(a.x) - from_left(a.x), from_right(a.y) - from_left(a.y)}; }

#define DPP_VALID 60
template <int SR, bool DUMP>
__global__ void __launch_bounds__(256)
hv_dpp_kernel(const float *__restrict__ g, int nz, int ny, int nx, Dv1<2> dz, Dv1<2> dy, Dv1<2> dx, int zchunk, int ntx, int nty,
              unsigned int *__restrict__ res, float *__restrict__ dump) {
    static_assert(SR % 2 == 0, "rows in pairs");
    constexpr int P = SR / 2;                    // aligned row pairs (0,1), (2,3), ...
    constexpr int NR = SR + 4;                   // raw rows -2 .. SR+1
    const int lane = threadIdx.x & 63;
    const unsigned nblk = gridDim.x;
    unsigned bid = blockIdx.x;
    if ((nblk & 7u) == 0u) bid = (bid & 7u) * (nblk >> 3) + (bid >> 3);
    const unsigned tile = bid * 4u + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tx = (int)(tile % (unsigned)ntx), ty = (int)((tile / (unsigned)ntx) % (unsigned)nty), zc = (int)(tile / ((unsigned)ntx * (unsigned)nty));
    const int zc0 = zc * zchunk, zc1 = (zc0 + zchunk < nz) ? zc0 + zchunk : nz;
    if (zc0 >= nz) return;
    const int x = tx * DPP_VALID - 2 + lane, y0 = ty * SR;
    auto clampi = [](int q, int n) -> int { return q < 0 ? 0 : (q > n - 1 ? n - 1 : q); };
    const int xc = clampi(x, nx);
    const i64 sz = (i64)ny * nx;
    const int pmin = clampi(zc0 - 2, nz), pmax = clampi(zc1 + 1, nz);
    const unsigned int plane_bytes = (unsigned int)sz * 4u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(g + (i64)pmin * sz), 0, (int)((unsigned int)(pmax - pmin + 1) * plane_bytes), 0x00020000);
    unsigned int rowoff[NR];                     // wave-uniform: row offsets are scalars
#pragma unroll
    for (int r = 0; r < NR; ++r) rowoff[r] = (unsigned int)__builtin_amdgcn_readfirstlane(clampi(y0 - 2 + r, ny) * nx) * 4u;
    const int xoff = xc * 4;
    auto ld = [&](int pz, int r) -> float {
        const int pc = clampi(pz, nz);
        const unsigned int so = (unsigned int)((pc < pmax ? pc : pmax) - pmin) * plane_bytes + rowoff[r];
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, xoff, (int)so, 0));
    };
    float raw[4][NR];                            // ring: plane z + k in raw[(u + k) & 3] at unrolled step u
    // first derivatives, two planes (b = plane parity of the unrolled step): d/dz, d/dy on row pairs (-1,0), (1,2), ... (SR-1,SR);
    // row-ALIGNED copies of d/dz (for the Z ring and the X differences) and d/dy (X differences); d/dx on the aligned pairs
    v2f GZ[2][P + 1], GY[2][P + 1], GZA[2][P], GYA[2][P], GX[2][P];
    v2f gz_m1[P];
    auto first = [&](const float (&Pm)[NR], const float (&P0)[NR], const float (&Pp)[NR], int b) {
#pragma unroll
        for (int k = 0; k <= P; ++k) {           // rows 2k-1, 2k  = raw index 2k+1, 2k+2
            v2f a = v2f{Pp[2 * k + 1], Pp[2 * k + 2]} - v2f{Pm[2 * k + 1], Pm[2 * k + 2]};
            v2f c = v2f{P0[2 * k + 2], P0[2 * k + 3]} - v2f{P0[2 * k], P0[2 * k + 1]};
            GZ[b][k] = dz.div(a); GY[b][k] = dy.div(c);
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {            // rows 2k, 2k+1 = raw index 2k+2, 2k+3
            GZA[b][k] = v2f{GZ[b][k].y, GZ[b][k + 1].x};
            GYA[b][k] = v2f{GY[b][k].y, GY[b][k + 1].x};
            GX[b][k] = dx.div(dx_of(v2f{P0[2 * k + 2], P0[2 * k + 3]}));
        }
    };
    // ---- prologue: planes zc0-1 .. zc0+2 in the ring; d/dz of zc0-1 (own rows), first derivatives of zc0 -> buffer 0
#pragma unroll
    for (int r = 0; r < NR; ++r) { raw[3][r] = ld(zc0 - 2, r); raw[0][r] = ld(zc0 - 1, r); raw[1][r] = ld(zc0, r); raw[2][r] = ld(zc0 + 1, r); }
#pragma unroll
    for (int k = 0; k < P; ++k) gz_m1[k] = dz.div(v2f{raw[1][2 * k + 2], raw[1][2 * k + 3]} - v2f{raw[3][2 * k + 2], raw[3][2 * k + 3]});
    first(raw[0], raw[1], raw[2], 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) raw[3][r] = ld(zc0 + 2, r);
    float mabs = 0.0f, mabs2 = 0.0f, mfrob = 0.0f;
    // step u (compile time): plane z = zc0 + ..., ring slots (u+1)&3 = plane z, (u+2)&3 = z+1, (u+3)&3 = z+2, u&3 = the plane entering (z+3)
    auto step = [&](const int z, auto uc) {
        constexpr int U = decltype(uc)::value;
        constexpr int b0 = U & 1, b1 = (U + 1) & 1;
        // plane z+3 goes in flight into the slot plane z-1 leaves (its last use was the previous step's d/dz)
#pragma unroll
        for (int r = 0; r < NR; ++r) raw[U & 3][r] = ld(z + 3, r);
        first(raw[(U + 1) & 3], raw[(U + 2) & 3], raw[(U + 3) & 3], b1);          // plane z+1 from planes z, z+1, z+2
#pragma unroll
        for (int k = 0; k < P; ++k) {
            v2f h0 = GZA[b1][k] - gz_m1[k];
            v2f h1 = GZ[b0][k + 1] - GZ[b0][k];
            v2f h2 = dx_of(GZA[b0][k]);
            v2f h3 = GY[b0][k + 1] - GY[b0][k];
            v2f h4 = dx_of(GYA[b0][k]);
            v2f h5 = dx_of(GX[b0][k]);
            div3(dz, h0, dy, h1, dx, h2);
            div3(dy, h3, dx, h4, dx, h5);
            const v2f fa = h0 * h0 + h3 * h3 + h5 * h5;
            const v2f fb = 2.0f * (h1 * h1 + h2 * h2 + h4 * h4);
            const v2f fsq = fa + fb;
            max3_abs(mabs, h0.x, h0.y); max3_abs(mabs2, h1.x, h1.y); max3_abs(mabs, h2.x, h2.y);
            max3_abs(mabs2, h3.x, h3.y); max3_abs(mabs, h4.x, h4.y); max3_abs(mabs2, h5.x, h5.y);
            max3_f(mfrob, fsq.x, fsq.y);
            gz_m1[k] = GZA[b0][k];
            if (DUMP) {
                if (lane >= 2 && lane < 2 + DPP_VALID && x < nx) {
                    if (y0 + 2 * k < ny) dump[(i64)z * sz + (i64)(y0 + 2 * k) * nx + x] = fsq.x;
                    if (y0 + 2 * k + 1 < ny) dump[(i64)z * sz + (i64)(y0 + 2 * k + 1) * nx + x] = fsq.y;
                }
            }
        }
    };
    int z = zc0;
    for (; z + 4 <= zc1; z += 4) {
        step(z, std::integral_constant<int, 0>{}); step(z + 1, std::integral_constant<int, 1>{});
        step(z + 2, std::integral_constant<int, 2>{}); step(z + 3, std::integral_constant<int, 3>{});
    }
    // (chunks are multiples of four planes in this benchmark)
    const bool valid = lane >= 2 && lane < 2 + DPP_VALID && x < nx;
    float a = valid ? fmaxf(mabs, mabs2) : 0.0f, b = valid ? mfrob : 0.0f;
    a = wave_max_f(a); b = wave_max_f(b);
    if (lane == 0) {
        if (a > 0.0f && __float_as_uint(a) > __hip_atomic_load(&res[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&res[0], __float_as_uint(a));
        if (b > 0.0f && __float_as_uint(b) > __hip_atomic_load(&res[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&res[1], __float_as_uint(b));
    }
}

// plain reference: frob_sq of interior voxels, one thread per voxel, the production kernel's operations (np.gradient twice, central differences)
__global__ void naive_fsq_kernel(const float *__restrict__ g, int nz, int ny, int nx, Dv1<2> dz, Dv1<2> dy, Dv1<2> dx, float *__restrict__ out) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 sz = (i64)ny * nx;
    if (i >= (i64)nz * sz) return;
    const int xx = (int)(i % nx), yy = (int)((i / nx) % ny), zz = (int)(i / sz);
    if (xx < 2 || xx > nx - 3 || yy < 2 || yy > ny - 3 || zz < 2 || zz > nz - 3) { out[i] = -1.0f; return; }
    auto F = [&](int dzz, int dyy, int dxx) -> float { return g[i + (i64)dzz * sz + (i64)dyy * nx + dxx]; };
    auto d1 = [&](const Dv1<2> &d, float p, float m) -> float { v2f q = d.div(v2f{p - m, 0.0f}); return q.x; };
    auto gz = [&](int a, int b, int c) { return d1(dz, F(a + 1, b, c), F(a - 1, b, c)); };
    auto gy = [&](int a, int b, int c) { return d1(dy, F(a, b + 1, c), F(a, b - 1, c)); };
    auto gx = [&](int a, int b, int c) { return d1(dx, F(a, b, c + 1), F(a, b, c - 1)); };
    const float h0 = d1(dz, gz(1, 0, 0), gz(-1, 0, 0)), h1 = d1(dy, gz(0, 1, 0), gz(0, -1, 0)), h2 = d1(dx, gz(0, 0, 1), gz(0, 0, -1));
    const float h3 = d1(dy, gy(0, 1, 0), gy(0, -1, 0)), h4 = d1(dx, gy(0, 0, 1), gy(0, 0, -1)), h5 = d1(dx, gx(0, 0, 1), gx(0, 0, -1));
    const float fa = h0 * h0 + h3 * h3 + h5 * h5, fb = 2.0f * (h1 * h1 + h2 * h2 + h4 * h4);
    out[i] = fa + fb;
}

template <int SR>
static float run_dpp(const float *d_g, int nz, int ny, int nx, const HessDv<2> &hr, int zchunk, unsigned int *d_res, int reps, float *d_dump, unsigned int *h_res) {
    const int ntx = (nx + DPP_VALID - 1) / DPP_VALID, nty = (ny + SR - 1) / SR, nzc = (nz + zchunk - 1) / zchunk;
    const unsigned waves = (unsigned)ntx * nty * nzc, blocks = (waves + 3) / 4;
    const Dv1<2> dz{hr.z2.yl, hr.z2.y}, dy{hr.y2.yl, hr.y2.y}, dx{hr.x2.yl, hr.x2.y};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(d_res, 0, 16));
    if (d_dump) hv_dpp_kernel<SR, true><<<blocks, 256>>>(d_g, nz, ny, nx, dz, dy, dx, zchunk, ntx, nty, d_res, d_dump);
    else hv_dpp_kernel<SR, false><<<blocks, 256>>>(d_g, nz, ny, nx, dz, dy, dx, zchunk, ntx, nty, d_res, nullptr);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h_res, d_res, 16, hipMemcpyDeviceToHost));
    if (d_dump) return 0.0f;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hv_dpp_kernel<SR, false><<<blocks, 256>>>(d_g, nz, ny, nx, dz, dy, dx, zchunk, ntx, nty, d_res, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv) {
    int nz = argc > 3 ? atoi(argv[1]) : 512, ny = argc > 3 ? atoi(argv[2]) : 1024, nx = argc > 3 ? atoi(argv[3]) : 1024;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const HessP hp{0.1f, 0.1f, 0.1f, (float)(2.0 * 0.1), (float)(2.0 * 0.1), (float)(2.0 * 0.1)};
    const HessDv<2> hr = hessdv_two(hp);
    // ---- correctness on a small volume: frob_sq of the skeleton against the plain kernel, interior voxels, bit for bit
    {
        const int sz_ = 24, sy = 50, sx = 150;
        const size_t n = (size_t)sz_ * sy * sx;
        std::vector<float> h(n);
        unsigned s = 12345u;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = 100.0f + (float)(s >> 8) * (1.0f / 16777216.0f) * 37.0f; }
        float *d_g, *d_a, *d_b; unsigned int *d_res; unsigned int h_res[4];
        CK(hipMalloc(&d_g, n * 4)); CK(hipMalloc(&d_a, n * 4)); CK(hipMalloc(&d_b, n * 4)); CK(hipMalloc(&d_res, 16));
        CK(hipMemcpy(d_g, h.data(), n * 4, hipMemcpyHostToDevice));
        const Dv1<2> dz{hr.z2.yl, hr.z2.y}, dy{hr.y2.yl, hr.y2.y}, dx{hr.x2.yl, hr.x2.y};
        naive_fsq_kernel<<<(unsigned)((n + 255) / 256), 256>>>(d_g, sz_, sy, sx, dz, dy, dx, d_a);
        for (int sr : {4, 6}) {
            CK(hipMemset(d_b, 0xff, n * 4));
            if (sr == 4) run_dpp<4>(d_g, sz_, sy, sx, hr, 8, d_res, 1, d_b, h_res); else run_dpp<6>(d_g, sz_, sy, sx, hr, 8, d_res, 1, d_b, h_res);
            std::vector<float> a(n), b(n);
            CK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
            size_t cmp = 0, bad = 0;
            for (size_t i = 0; i < n; ++i) if (a[i] >= 0.0f) { ++cmp; if (memcmp(&a[i], &b[i], 4)) ++bad; }
            printf("check SR=%d: %zu interior voxels compared, %zu differ\n", sr, cmp, bad);
        }
        hipFree(d_g); hipFree(d_a); hipFree(d_b); hipFree(d_res);
    }
    // ---- timing
    const size_t n = (size_t)nz * ny * nx;
    float *d_g; unsigned int *d_res; unsigned int h_res[4];
    CK(hipMalloc(&d_g, n * 4)); CK(hipMalloc(&d_res, 16));
    {
        std::vector<float> h(n);
        unsigned s = 777u;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = 100.0f + (float)(s >> 8) * (1.0f / 16777216.0f) * 20.0f; }
        CK(hipMemcpy(d_g, h.data(), n * 4, hipMemcpyHostToDevice));
    }
    // production kernel, MODE 0 (statistics only): as nl_hessian_stats launches it
    {
        VolGeom v{}; v.nzl = nz; v.ny = ny; v.nx = nx; v.gz0 = 0; v.gnz = nz;
        const int ntx = (nx + HM_TX - 1) / HM_TX, nty = (ny + 15) / 16, nzc = (nz + HM_ZCHUNK - 1) / HM_ZCHUNK;
        VessP vp{};
        HvLaunch L{0, 8, 1, 2, (unsigned)(ntx * nty * nzc), 0, d_g, nullptr, nullptr, 0, v, hp, vp, VQueue{}, 0, nz, ntx, nty, d_res, nullptr, nullptr};
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipMemset(d_res, 0, 16));
        CK(nl_hv_launch(L)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h_res, d_res, 16, hipMemcpyDeviceToHost));
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) CK(nl_hv_launch(L));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%d x %d x %d  production walk, MODE 0 (LDS ring, barrier per plane; faces handled): %.4f ms per launch   max|H| %08x max frob_sq %08x\n", nz, ny, nx, ms / reps, h_res[0], h_res[1]);
    }
    for (int zchunk : {64, 128, 256}) {
        unsigned int r4[4], r6[4];
        const float t4 = run_dpp<4>(d_g, nz, ny, nx, hr, zchunk, d_res, reps, nullptr, r4);
        const float t6 = run_dpp<6>(d_g, nz, ny, nx, hr, zchunk, d_res, reps, nullptr, r6);
        printf("%d x %d x %d  DPP skeleton (no LDS, no barrier; interior arithmetic only), Z chunk %3d:  SR=4 %.4f ms   SR=6 %.4f ms   (max|H| %08x / %08x, max frob_sq %08x / %08x: faces differ by design)\n",
               nz, ny, nx, zchunk, t4, t6, r4[0], r6[0], r4[1], r6[1]);
    }
    return 0;
}
