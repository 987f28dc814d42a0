// Micro-benchmark: returning global atomicAdd under the access patterns of the bin-by-tile append.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/atomic_bench tools/micro/atomic_bench.hip && tools/micro/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_atomics(uint32_t* cur, int M, int stride, int per_lane, int active_lanes, unsigned long long* out, uint32_t seed)
{
    const int lane = threadIdx.x & 63;
    if (lane >= active_lanes) return;
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
    unsigned long long acc = 0;
    for (int k = 0; k < per_lane; k++) {
        x = x * 1664525u + 1013904223u;
        const int t = (int)((x >> 8) % (uint32_t)M);
        const uint32_t pos = atomicAdd(&cur[(size_t)t * stride], 1u);
        acc += pos;                                   // dependent use, like the bin store
    }
    if (acc == 0xFFFFFFFFFFFFull) out[0] = acc;
}
__global__ void k_atomics_noret(uint32_t* cur, int M, int stride, int per_lane, int active_lanes, uint32_t seed)
{
    const int lane = threadIdx.x & 63;
    if (lane >= active_lanes) return;
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
    for (int k = 0; k < per_lane; k++) {
        x = x * 1664525u + 1013904223u;
        atomicAdd(&cur[(size_t)((x >> 8) % (uint32_t)M) * stride], 1u);
    }
}
int main()
{
    uint32_t* cur; unsigned long long* out;
    hipMalloc(&cur, (size_t)4096 * 1024 * 4); hipMalloc(&out, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct Cfg { int M, stride, waves, per_lane, lanes; };
    const Cfg cfgs[] = {
        {1200, 16, 6144, 1, 64}, {1200, 16, 1536, 4, 64}, {1200, 16, 6144 * 4, 1, 16}, {1200, 256, 6144, 1, 64},
        {600, 16, 1024, 1, 64}, {75, 16, 1024, 1, 64}, {1200, 16, 24576, 1, 16}, {100000, 16, 6144, 1, 64},
    };
    for (const Cfg& c : cfgs) {
        for (int ret = 1; ret >= 0; ret--) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                hipMemset(cur, 0, (size_t)4096 * 1024 * 4);
                hipEventRecord(a);
                if (ret) hipLaunchKernelGGL(k_atomics, dim3(c.waves / 4), dim3(256), 0, 0, cur, c.M, c.stride, c.per_lane, c.lanes, out, 17u * rep);
                else hipLaunchKernelGGL(k_atomics_noret, dim3(c.waves / 4), dim3(256), 0, 0, cur, c.M, c.stride, c.per_lane, c.lanes, 17u * rep);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
            }
            const double n = (double)c.waves * c.lanes * c.per_lane;
            printf("%s M=%6d stride=%4d words  waves=%5d x %2d lanes x %d per lane = %7.0f atomics : %7.1f us  (%.2f atomics/ns)\n",
                   ret ? "returning" : "no-return", c.M, c.stride, c.waves, c.lanes, c.per_lane, n, best * 1e3, n / (best * 1e6));
        }
    }
    return 0;
}
