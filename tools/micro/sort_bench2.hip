#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <class Config, class K, class V>
float run(const char* name, const K* kin, K* kout, const V* vin, V* vout, unsigned n, int b0, int b1, int reps = 20)
{
    size_t tmp_bytes = 0;
    CK((rocprim::radix_sort_pairs<Config>(nullptr, tmp_bytes, kin, kout, vin, vout, n, b0, b1)));
    void* tmp; CK(hipMalloc(&tmp, tmp_bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) CK((rocprim::radix_sort_pairs<Config>(tmp, tmp_bytes, kin, kout, vin, vout, n, b0, b1)));
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) CK((rocprim::radix_sort_pairs<Config>(tmp, tmp_bytes, kin, kout, vin, vout, n, b0, b1)));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-40s n=%8u bits[%d,%d) : %8.1f us\n", name, n, b0, b1, 1000.f * ms / reps);
    CK(hipFree(tmp));
    return ms / reps;
}
int main()
{
    using namespace rocprim;
    const unsigned P = 1000000, R = 3370000;
    std::mt19937 rng(1);
    std::vector<unsigned> hk(P), hv(P);
    for (unsigned i = 0; i < P; i++) { float z = 0.5f + 5.5f * (rng() / 4294967296.f); memcpy(&hk[i], &z, 4); hv[i] = i; }
    unsigned *k0, *k1, *v0, *v1;
    CK(hipMalloc(&k0, P * 4)); CK(hipMalloc(&k1, P * 4)); CK(hipMalloc(&v0, P * 4)); CK(hipMalloc(&v1, P * 4));
    CK(hipMemcpy(k0, hk.data(), P * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v0, hv.data(), P * 4, hipMemcpyHostToDevice));
    std::vector<unsigned short> tk(R); std::vector<unsigned> tv(R);
    for (unsigned i = 0; i < R; i++) { tk[i] = rng() % 1200; tv[i] = rng() % P; }
    unsigned short *t0, *t1; unsigned *w0, *w1;
    CK(hipMalloc(&t0, R * 2)); CK(hipMalloc(&t1, R * 2)); CK(hipMalloc(&w0, R * 4)); CK(hipMalloc(&w1, R * 4));
    CK(hipMemcpy(t0, tk.data(), R * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w0, tv.data(), R * 4, hipMemcpyHostToDevice));
    constexpr auto M = block_radix_rank_algorithm::match;
#define CFG(BS, IPT, BITS, ALG) radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<BS, IPT>, kernel_config<BS, IPT>, BITS, ALG>, 1024>
#define BOTH(BS, IPT, BITS, ALG, nm) { using C = CFG(BS, IPT, BITS, ALG); run<C>("depth " nm, k0, k1, v0, v1, P, 0, 32); run<C>("tile  " nm, t0, t1, w0, w1, R, 0, 11); }
    BOTH(VBS, VIPT, VBITS, VALG, VNAME)
    return 0;
}
