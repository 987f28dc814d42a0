// What does DPP row_newbcast:n do on gfx950 for 32-bit VOP2?  (hipcc --offload-arch=gfx950 -O2 -o /tmp/dppt tools/micro/dpp_newbcast.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out)
{
    const int lane = threadIdx.x;
    float v = (float)lane, one = 1.0f, r_mov, r_sub, r_mul, acc = 1000.f;
    asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(r_mov) : "v"(v));
    asm volatile("v_sub_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(r_sub) : "v"(v), "v"(one));
    asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "=v"(r_mul) : "v"(v), "v"(v));
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(one));
    out[lane] = r_mov; out[64 + lane] = r_sub; out[128 + lane] = r_mul; out[192 + lane] = acc;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"mov newbcast:3 (expect 16*row+3)", "sub newbcast:5 (expect 16*row+5-1)", "mul newbcast:7 (expect (16*row+7)*lane)", "fmac newbcast:2 (expect 1000+16*row+2)"};
    for (int q = 0; q < 4; q++) { printf("%s\n", names[q]); for (int i = 0; i < 64; i++) printf("%g%c", h[64 * q + i], (i & 15) == 15 ? '\n' : ' '); }
    return 0;
}
