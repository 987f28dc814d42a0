// VALU issue rate on gfx950: wave64 v_fma_f32 / v_pk_fma_f32 / v_exp_f32 / v_mfma_f32_16x16x4_f32 instructions per CU-cycle,
// for 1, 2, 5 and 8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, int iters, unsigned long long* cyc)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    f4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
    const float b = 1.0000001f, c = 1e-9f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
                a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(f2{b, b}), "v"(f2{c, c}));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
                a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, a3, d1, 0, 0, 0);
            }
        } else {      // 8 FMAs + 1 MFMA interleaved: do they overlap?
#pragma unroll
            for (int u = 0; u < 4; u++) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, d0, 0, 0, 0);
                a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c); a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c);
                a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
            }
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.y + p3.y + d0.x + d1.y;
}
template <int MODE>
void run(const char* name, int per_iter)
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 32 * 64 * 4); hipMalloc(&cyc, 256 * 32 * 8);
    const int iters = 2000;
    for (int wps : {1, 2, 5, 8}) {
        const int blocks = 256 * 4 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 10, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
        const double insts_per_wave = (double)iters * per_iter;
        printf("%-28s waves/SIMD %d: %.1f clock64 ticks per instruction per wave, %.2f ticks per instruction per SIMD; wall %.3f ms\n", name, wps,
               mean / insts_per_wave, mean / insts_per_wave / wps, ms);
    }
}
int main()
{
    run<0>("v_fma_f32", 32);
    run<1>("v_pk_fma_f32", 32);
    run<2>("v_exp_f32", 32);
    run<3>("v_mfma_f32_16x16x4_f32", 32);
    run<4>("1 mfma + 8 fma (per group)", 4);
    return 0;
}
