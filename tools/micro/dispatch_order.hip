// Which CU does workgroup b of a 1200-workgroup launch (256 threads, ~31 KB of LDS: five per CU, all resident at once) land on?
// Decides how a work-sorted tile order has to be laid out so that every CU gets a mix of heavy and light tiles.
// build: hipcc --offload-arch=gfx950 -O2 -o dispatch_order dispatch_order.hip ; prints per-CU lists of block ids.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
__global__ void __launch_bounds__(256) k(uint32_t* out, int spin)
{
    __shared__ float pad[31 * 256];
    pad[threadIdx.x] = (float)spin;
    __syncthreads();
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // HW_REG_XCC_ID, bits 3:0
    const unsigned long long t0 = __builtin_readcyclecounter();
    float acc = pad[threadIdx.x];
    for (int i = 0; i < spin; i++) acc = acc * 1.0001f + 0.5f;
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = hw; out[blockIdx.x * 4 + 1] = xcc; out[blockIdx.x * 4 + 2] = (uint32_t)t0; out[blockIdx.x * 4 + 3] = (uint32_t)acc; }
}
int main()
{
    const int n = 1200;
    uint32_t* d; hipMalloc(&d, n * 16);
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d, 20000); hipDeviceSynchronize(); }
    std::vector<uint32_t> h(n * 4); hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    std::map<uint32_t, std::vector<int>> cu;
    for (int b = 0; b < n; b++) {
        const uint32_t hw = h[b * 4], xcc = h[b * 4 + 1] & 15;
        const uint32_t cu_id = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(b);
    }
    printf("%zu distinct CUs\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) {
        if (shown++ < 12 || shown > (int)cu.size() - 3) {
            printf("xcc %u se %u sh %u cu %2u :", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
            for (int b : kv.second) printf(" %4d", b);
            printf("\n");
        }
    }
    // how far apart (in block id) are the workgroups sharing a CU?
    std::map<int, int> hist;
    for (auto& kv : cu) hist[(int)kv.second.size()]++;
    for (auto& kv : hist) printf("%d CUs hold %d workgroups\n", kv.second, kv.first);
    return 0;
}
