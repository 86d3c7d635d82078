// Micro-benchmark (gfx950): what an 8 B/voxel read+write stream reaches on this part as a function of its ACCESS SHAPE --
// the ceiling of the Gaussian passes.  hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_patterns tools/ubench/stream_patterns.hip
//   copy4        grid-stride float4 copy (the guide's 6.29 TB/s reference)
//   march<V,YB>  the Z pass's shape: a thread walks Z for V adjacent x (float / float2 / float4 loads), 64 lanes along X,
//                YB rows per workgroup, W planes in flight, chunk planes per workgroup; out[z] = in[z] (no arithmetic)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) copy4(const float4 *in, float4 *out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 1; };
template <> struct VecOf<float2> { static constexpr int N = 2; };
template <> struct VecOf<float4> { static constexpr int N = 4; };

template <typename T, int YB, int W>
__global__ void __launch_bounds__(64 * YB) march(const T *in, T *out, int nz, int ny, int nxv, int chunk) {
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lx, y = blockIdx.y * YB + ly;
    if (x >= nxv || y >= ny) return;
    const size_t plane = (size_t)ny * nxv, base = (size_t)y * nxv + x;
    const int c0 = blockIdx.z * chunk, c1 = c0 + chunk < nz ? c0 + chunk : nz;
    T nxt[W];
#pragma unroll
    for (int k = 0; k < W; ++k) nxt[k] = in[base + (size_t)(c0 + k < nz ? c0 + k : nz - 1) * plane];
    int p = c0;
    for (; p + W <= c1; p += W) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const T v = nxt[k];
            const int q = p + k + W;
            nxt[k] = in[base + (size_t)(q < nz ? q : nz - 1) * plane];
            out[base + (size_t)(p + k) * plane] = v;
        }
    }
    for (int k = 0; p + k < c1; ++k) out[base + (size_t)(p + k) * plane] = in[base + (size_t)(p + k) * plane];
}

// the fused Y+X pass's shape: a workgroup of T threads owns T*V columns of ONE plane and walks a chunk of rows (W rows in
// flight per thread), writing the rows it read: lanes along X, consecutive rows 4 KiB apart
template <typename T, int NT, int W>
__global__ void __launch_bounds__(NT) ymarch(const T *in, T *out, int nz, int ny, int nxv, int chunk) {
    const int x = blockIdx.x * NT + threadIdx.x;
    if (x >= nxv) return;
    const int z = blockIdx.z;
    const int c0 = blockIdx.y * chunk, c1 = c0 + chunk < ny ? c0 + chunk : ny;
    const size_t base = (size_t)z * ny * nxv + x;
    T nxt[W];
#pragma unroll
    for (int k = 0; k < W; ++k) nxt[k] = in[base + (size_t)(c0 + k < ny ? c0 + k : ny - 1) * nxv];
    int p = c0;
    for (; p + W <= c1; p += W) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const T v = nxt[k];
            const int q = p + k + W;
            nxt[k] = in[base + (size_t)(q < ny ? q : ny - 1) * nxv];
            out[base + (size_t)(p + k) * nxv] = v;
        }
    }
    for (int k = 0; p + k < c1; ++k) out[base + (size_t)(p + k) * nxv] = in[base + (size_t)(p + k) * nxv];
}

template <typename F> static float time_ms(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const int nz = 1024, ny = 1024, nx = 1024;
    const size_t n = (size_t)nz * ny * nx;
    float *in, *out;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
    CK(hipMemset(in, 1, n * 4));
    const double gb = 8.0 * n / 1e9;
    auto report = [&](const char *name, float ms) { printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, gb / ms); };
    report("hipMemcpyAsync D2D", time_ms([&] { CK(hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, 0)); }, 5));
    for (int g : {2048, 8192, 32768, 131072}) {
        char nm[64]; snprintf(nm, 64, "copy4 grid %d", g);
        report(nm, time_ms([&] { copy4<<<g, 256>>>((const float4 *)in, (float4 *)out, n / 4); }, 5));
    }
#define MARCH(T, YB, W, CH)                                                                                   \
    {                                                                                                         \
        const int nxv = nx / VecOf<T>::N;                                                                     \
        dim3 grid((nxv + 63) / 64, (ny + YB - 1) / YB, (nz + CH - 1) / CH);                                   \
        char nm[64]; snprintf(nm, 64, "march %-6s rows/wg %d  in flight %2d  chunk %4d", #T, YB, W, CH);      \
        report(nm, time_ms([&] { march<T, YB, W><<<grid, 64 * YB>>>((const T *)in, (T *)out, nz, ny, nxv, CH); }, 5)); \
    }
    MARCH(float, 4, 9, 128) MARCH(float, 4, 9, 256) MARCH(float, 4, 9, 1024) MARCH(float, 4, 4, 256) MARCH(float, 4, 16, 256)
    MARCH(float, 1, 9, 256) MARCH(float, 2, 9, 256) MARCH(float, 8, 9, 256) MARCH(float, 16, 9, 256)
    MARCH(float2, 4, 9, 256) MARCH(float2, 2, 9, 256) MARCH(float2, 8, 9, 256) MARCH(float2, 4, 4, 256)
    MARCH(float4, 4, 9, 256) MARCH(float4, 2, 9, 256) MARCH(float4, 4, 4, 256) MARCH(float4, 1, 9, 256)
#define YMARCH(T, NT, W, CH)                                                                                  \
    {                                                                                                         \
        const int nxv = nx / VecOf<T>::N;                                                                     \
        dim3 grid((nxv + NT - 1) / NT, (ny + CH - 1) / CH, nz);                                               \
        char nm[64]; snprintf(nm, 64, "ymarch %-6s threads %3d  in flight %2d  chunk %4d", #T, NT, W, CH);    \
        report(nm, time_ms([&] { ymarch<T, NT, W><<<grid, NT>>>((const T *)in, (T *)out, nz, ny, nxv, CH); }, 5)); \
    }
    YMARCH(float, 256, 9, 256) YMARCH(float, 256, 9, 128) YMARCH(float, 320, 9, 256) YMARCH(float2, 128, 9, 256) YMARCH(float2, 256, 9, 256)
    YMARCH(float2, 160, 9, 256) YMARCH(float4, 64, 9, 256) YMARCH(float4, 128, 9, 256) YMARCH(float2, 256, 4, 256)
    return 0;
}
