// Micro-benchmark (gfx950): issue cost per SIMD of the float64 instructions the Gaussian passes are made of (scipy's float64
// accumulation): v_add_f64, v_mul_f64, v_fma_f64, v_cvt_f64_f32, v_cvt_f32_f64, at 1/2/4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_issue tools/ubench/f64_issue.hip && /tmp/f64_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ void __launch_bounds__(1024) k(double *out, int iters) {
    double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const double b = 1.0000001, c = 0.5;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            REP16(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                               "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b));)
        } else if (KIND == 1) {
            REP16(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                               "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b));)
        } else if (KIND == 2) {
            REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                               "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b), "v"(c));)
        } else if (KIND == 3) {
            REP16(asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n"
                               "v_cvt_f64_f32 %4, %12\n v_cvt_f64_f32 %5, %13\n v_cvt_f64_f32 %6, %14\n v_cvt_f64_f32 %7, %15\n"
                               : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7)
                               : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));)
        } else if (KIND == 4) {
            REP16(asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n"
                               "v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15\n"
                               : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3), "=v"(f4), "=v"(f5), "=v"(f6), "=v"(f7)
                               : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));)
        } else if (KIND == 5) {   // the Gaussian's inner pattern: add of two neighbours, mul by a weight, add to the sum (dependent on the sum only)
            REP16(asm volatile("v_add_f64 %4, %0, %1\n v_mul_f64 %4, %4, %8\n v_add_f64 %6, %6, %4\n v_add_f64 %5, %2, %3\n v_mul_f64 %5, %5, %8\n v_add_f64 %7, %7, %5\n"
                               "v_add_f64 %4, %1, %2\n v_mul_f64 %4, %4, %8\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
template <int KIND> void run(const char *name) {
    double *out;
    hipMalloc(&out, 1024 * 1024 * 8);
    const int iters = 200;
    printf("%-22s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int grid = 256;
        k<KIND><<<grid, 256 * wps, 0, 0>>>(out, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<KIND><<<grid, 256 * wps, 0, 0>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n_inst = (double)iters * 16 * 8;
        printf(" | wps %d: %.3f ms => SIMD cycles per instruction %.2f", wps, ms, (ms * 1e-3 * 2.4e9) / (n_inst * wps));
    }
    printf("\n");
    hipFree(out);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s CUs %d clock %d kHz (cycles assume 2.4 GHz)\n", p.name, p.multiProcessorCount, p.clockRate);
    run<0>("v_add_f64"); run<1>("v_mul_f64"); run<2>("v_fma_f64"); run<3>("v_cvt_f64_f32"); run<4>("v_cvt_f32_f64"); run<5>("add,mul,add pattern");
    return 0;
}
