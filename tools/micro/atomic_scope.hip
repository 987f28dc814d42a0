// Micro-benchmark (round 6): what does a returning atomicAdd on a hot tile cursor cost, by scope and by sharing across XCDs?
// 4096 waves x 64 lanes, `per` atomics per lane onto `naddr` cursors 64 B apart (the binning appends of k_preprocess_lean):
//   mode 0: agent scope, one cursor array shared by all XCDs (what the kernels do)
//   mode 1: workgroup scope (an XCD-local L2 atomic), one cursor array PER XCD (XCC_ID), so no line is shared between XCDs
//   mode 2: agent scope, one cursor array per XCD
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/atomic_scope tools/micro/atomic_scope.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(256) k(uint32_t* cur, uint32_t* out, int naddr, int per, int mode, int hot)
{
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    uint32_t xcc = 0;
    if (mode != 0) xcc = __builtin_amdgcn_s_getreg(0x14 | (0 << 6) | (3 << 11)) & 7u;      // HW_REG_XCC_ID (id 20), bits [3:0]
    uint32_t acc = 0, h = g * 2654435761u;
    for (int i = 0; i < per; i++) {
        h = h * 1664525u + 1013904223u;
        // `hot` per cent of the appends go to the first 32 cursors (a dense object), the rest anywhere
        const uint32_t a = ((h >> 8) % 100u < (uint32_t)hot) ? ((h >> 16) % 32u) : ((h >> 16) % (uint32_t)naddr);
        uint32_t* p = cur + ((size_t)xcc * naddr + a) * 16;
        if (mode == 1) acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 0xFFFFFFFFu) out[g] = acc;
}
int main(int argc, char** argv)
{
    const int naddr = argc > 1 ? atoi(argv[1]) : 1200, per = argc > 2 ? atoi(argv[2]) : 4, hot = argc > 3 ? atoi(argv[3]) : 0;
    uint32_t *cur, *out;
    hipMalloc(&cur, (size_t)8 * naddr * 64); hipMalloc(&out, 4096 * 64 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f;
        uint32_t total = 0;
        for (int rep = 0; rep < 5; rep++) {
            hipMemset(cur, 0, (size_t)8 * naddr * 64);
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, cur, out, naddr, per, mode, hot);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        uint32_t* hostc = (uint32_t*)malloc((size_t)8 * naddr * 64);
        hipMemcpy(hostc, cur, (size_t)8 * naddr * 64, hipMemcpyDeviceToHost);
        for (int i = 0; i < 8 * naddr; i++) total += hostc[i * 16];
        free(hostc);
        printf("cursors %d, %d atomics per lane (%.2f M), hot %d%%, mode %d: %.1f us; sum of cursors %u (expected %u)\n", naddr, per, 1024.0 * 256 * per / 1e6, hot, mode, best * 1e3f, total,
               1024u * 256u * (uint32_t)per);
    }
    return 0;
}
