// Micro-benchmark (gfx950): issue cost of the instruction kinds the Hessian walk is made of, per SIMD, at 1/2/4 waves
// per SIMD.  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue tools/ubench/valu_issue.hip && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *cyc, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float b = 1.0001f, c = 0.5f;
    const f2 pb = {b, b}, pc = {c, c};
    __shared__ float lds[4096];
    lds[threadIdx.x] = a0; lds[threadIdx.x + 1024] = a1; lds[threadIdx.x + 2048] = a2; lds[threadIdx.x + 3072] = a3;
    __syncthreads();
    int s0 = iters, s1 = 3;
    const float *lp = lds + (threadIdx.x & 1023);
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {        // 16 independent v_fma_f32
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 1) { // v_pk_fma_f32
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
        } else if (KIND == 2) { // v_add_f32
            REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 3) { // v_pk_add_f32
            REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));)
        } else if (KIND == 4) { // v_pk_mul_f32
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));)
        } else if (KIND == 5) { // 8 VALU + 8 SALU interleaved
            REP16(asm volatile("v_fma_f32 %0, %0, %10, %11\n s_add_u32 %8, %8, %9\n v_fma_f32 %1, %1, %10, %11\n s_add_u32 %8, %8, %9\n v_fma_f32 %2, %2, %10, %11\n s_add_u32 %8, %8, %9\n v_fma_f32 %3, %3, %10, %11\n s_add_u32 %8, %8, %9\n"
                         "v_fma_f32 %4, %4, %10, %11\n s_add_u32 %8, %8, %9\n v_fma_f32 %5, %5, %10, %11\n s_add_u32 %8, %8, %9\n v_fma_f32 %6, %6, %10, %11\n s_add_u32 %8, %8, %9\n v_fma_f32 %7, %7, %10, %11\n s_add_u32 %8, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0) : "s"(s1), "v"(b), "v"(c) : "scc");)
        } else if (KIND == 6) { // v_max3_f32
            REP16(asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                         "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 7) { // v_cndmask / v_cmp pairs
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
        } else if (KIND == 8) { // ds_read_b32 x8 (no conflicts)
            REP16(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4096\n ds_read_b32 %2, %8 offset:8192\n ds_read_b32 %3, %8 offset:12288\n"
                         "ds_read_b32 %4, %8 offset:256\n ds_read_b32 %5, %8 offset:4352\n ds_read_b32 %6, %8 offset:8448\n ds_read_b32 %7, %8 offset:12544\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"((unsigned)((threadIdx.x & 63) * 4)) : "memory");)
        } else if (KIND == 9) { // ds_read_b128 x4
            f2 q0, q1; 
            REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4 offset:8192\n ds_read_b128 %3, %4 offset:1024\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(*(float4*)&p0), "=v"(*(float4*)&p2), "=v"(*(float4*)&p4), "=v"(*(float4*)&p6) : "v"((unsigned)((threadIdx.x & 63) * 16)) : "memory");)
        } else if (KIND == 10) { // ds_read2_b32 x8
            REP16(asm volatile("ds_read2_b32 %0, %8 offset0:0 offset1:64\n ds_read2_b32 %1, %8 offset0:128 offset1:192\n ds_read2_b32 %2, %8 offset0:1 offset1:65\n ds_read2_b32 %3, %8 offset0:129 offset1:193\n"
                         "ds_read2_b32 %4, %8 offset0:2 offset1:66\n ds_read2_b32 %5, %8 offset0:130 offset1:194\n ds_read2_b32 %6, %8 offset0:3 offset1:67\n ds_read2_b32 %7, %8 offset0:131 offset1:195\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3), "=v"(p4), "=v"(p5), "=v"(p6), "=v"(p7) : "v"((unsigned)((threadIdx.x & 63) * 4)) : "memory");)
        } else if (KIND == 11) { // v_mul_f32 + v_fma + v_fma dependent chain of 3 (the exact division), 8 chains... as 3 groups
            REP16(asm volatile("v_mul_f32 %0, %4, %8\n v_mul_f32 %1, %5, %8\n v_mul_f32 %2, %6, %8\n v_mul_f32 %3, %7, %8\n"
                         "v_fma_f32 %4, -%0, %9, %4\n v_fma_f32 %5, -%1, %9, %5\n v_fma_f32 %6, -%2, %9, %6\n v_fma_f32 %7, -%3, %9, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        }
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    const unsigned long long t1 = __builtin_readcyclecounter();
    a0 += a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y + s0 + lp[0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
    if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = m1 - m0; cyc[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = t1 - t0; }
}
template <int KIND> void run(const char *name, int per_iter) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 1024 * 16 * 16);
    const int iters = 200;
    printf("%-28s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD: block = 256*wps threads, one block per CU
        const int grid = 256;
        k<KIND><<<grid, 256 * wps, 0, 0>>>(out, cyc, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<KIND><<<grid, 256 * wps, 0, 0>>>(out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(grid * 4 * wps * 2);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0, avg2 = 0; for (size_t i = 0; i < h.size() / 2; ++i) { avg += h[2 * i]; avg2 += h[2 * i + 1]; }
        avg /= h.size() / 2; avg2 /= h.size() / 2;
        const double n_inst = (double)iters * 16 * per_iter;     // per wave
        // per SIMD: wps waves each issue n_inst instructions in `avg` memtime ticks (100 MHz?) / cycle counter ticks
        printf(" | wps %d: %.3f ms  memtime/inst %.3f  cyc/inst/wave %.2f  => SIMD cyc per inst %.2f", wps, ms, avg / n_inst, avg2 / n_inst, (ms * 1e-3 * 2.4e9) / (n_inst * wps));
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    run<0>("v_fma_f32", 8); run<1>("v_pk_fma_f32", 8); run<2>("v_add_f32", 8); run<3>("v_pk_add_f32", 8); run<4>("v_pk_mul_f32", 8);
    run<5>("v_fma + s_add interleaved", 16); run<6>("v_max3_f32", 8); run<7>("v_cmp+v_cndmask", 8); run<8>("ds_read_b32 (8/wait)", 8);
    run<9>("ds_read_b128 (4/wait)", 4); run<10>("ds_read2_b32 (8/wait)", 8); run<11>("mul+fma dep pairs", 8);
    return 0;
}
