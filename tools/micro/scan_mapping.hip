// Micro-benchmark: the two ways of laying a tile's compositing onto a wave of 64 lanes (forward pass only, synthetic tiles).
//   A  "lane = pixel"      what k_render_fwd does and what forward.cu:261-379 does on 32-wide warps: a wave owns 8 x 8 pixels, walks
//                          the tile's depth-ordered list entry by entry (record at a wave-uniform address: scalar loads), every lane
//                          carries its pixel's transmittance and colour in registers;
//   B  "lane = Gaussian"   the alternative named in the survey's north star: a wave takes 64 list entries at a time, one per lane, and
//                          visits its 64 pixels one after the other: alpha of all 64 entries for that pixel, the transmittance as a
//                          PREFIX PRODUCT over the lanes (6 DPP steps), the pixel's colour as four wave sums of alpha T c.  No lane
//                          idles on an entry whose footprint misses its pixel's neighbourhood -- every lane always has an entry --
//                          but the scan and the sums are paid per (pixel, 64 entries).
// Both produce the same image (checked).  Per (pixel, entry) pair: A spends 22 vector instructions per 64 pairs, B 50-60.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/scan_mapping tools/micro/scan_mapping.hip && tools/micro/scan_mapping
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REC 12      // x y B2 C2 A2 opacity depth - | r g b -      (the product's record layout, conic pre-scaled by log2 e)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define DPP(old, src, ctrl, rm) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(src)), ctrl, rm, 0xf, false))
#define RDL(v, l) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (float)(v)), (l)))

__device__ __forceinline__ float sum_to_lane63(float v)
{
    v += DPP(0.f, v, 0x111, 0xf); v += DPP(0.f, v, 0x112, 0xf); v += DPP(0.f, v, 0x114, 0xf); v += DPP(0.f, v, 0x118, 0xf);
    v += DPP(0.f, v, 0x142, 0xa); v += DPP(0.f, v, 0x143, 0xc);
    return v;
}
__device__ __forceinline__ float prefix_product(float v)          // inclusive, over the 64 lanes
{
    v *= DPP(1.f, v, 0x111, 0xf); v *= DPP(1.f, v, 0x112, 0xf); v *= DPP(1.f, v, 0x114, 0xf); v *= DPP(1.f, v, 0x118, 0xf);
    v *= DPP(1.f, v, 0x142, 0xa); v *= DPP(1.f, v, 0x143, 0xc);
    return v;
}

// ---- A: lane = pixel ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_lane_is_pixel(const float* __restrict__ rec, int L, int tiles_x, float* __restrict__ out)
{
    const int tile = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int px = (tile % tiles_x) * 16 + (wv & 1) * 8 + (lane & 7), py = (tile / tiles_x) * 16 + (wv >> 1) * 8 + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    float T = 1.f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Cd = 0.f;
    bool done = false;
    const float* r0 = rec + (size_t)tile * L * REC;
    for (int e = 0; e < L; e++) {
        if ((e & 7) == 0 && __ballot(!done) == 0ull) break;
        const float* r = r0 + (size_t)e * REC;          // wave-uniform: scalar loads
        const float dx = r[0] - pxf, dy = r[1] - pyf;
        const float p2 = fmaf(dx, fmaf(r[4], dx, r[2] * dy), r[3] * dy * dy);
        const float G = __builtin_amdgcn_exp2f(p2);
        const float alpha = fminf(0.99f, r[5] * G);
        const bool use = !done && !(alpha < 1.0f / 255.0f) && !(p2 > 0.f);
        const float test_T = T * (1.f - alpha);
        if (use && test_T < 0.0001f) done = true;
        if (use && !done) {
            const float w = alpha * T;
            Cr = fmaf(r[8], w, Cr); Cg = fmaf(r[9], w, Cg); Cb = fmaf(r[10], w, Cb); Cd = fmaf(r[6], w, Cd);
            T = test_T;
        }
    }
    float* o = out + ((size_t)py * tiles_x * 16 + px) * 5;
    o[0] = Cr; o[1] = Cg; o[2] = Cb; o[3] = Cd; o[4] = T;
}

// ---- B: lane = Gaussian, transmittance by prefix product ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_lane_is_gaussian(const float* __restrict__ rec, int L, int tiles_x, float* __restrict__ out)
{
    const int tile = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int qx = (tile % tiles_x) * 16 + (wv & 1) * 8, qy = (tile / tiles_x) * 16 + (wv >> 1) * 8;
    // lane p of these registers belongs to pixel p of the wave's 8 x 8 block
    float Tv = 1.f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Cd = 0.f;      // Tv <= 0: finished, final transmittance -Tv
    const float* r0 = rec + (size_t)tile * L * REC;
    for (int c0 = 0; c0 < L; c0 += 64) {
        if (__ballot(Tv > 0.f) == 0ull) break;
        const bool have = c0 + lane < L;
        const float4* r = reinterpret_cast<const float4*>(r0 + (size_t)(c0 + (have ? lane : 0)) * REC);
        const float4 a = r[0], b = r[1], c = r[2];      // x y B2 C2 | A2 opacity depth - | r g b -
        const float opac = have ? b.y : 0.f;
        for (int p = 0; p < 64; p++) {
            const float Tin = RDL(Tv, p);
            if (!(Tin > 0.f)) continue;                 // (wave-uniform)
            const float pxf = (float)(qx + (p & 7)), pyf = (float)(qy + (p >> 3));
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float p2 = fmaf(dx, fmaf(b.x, dx, a.z * dy), a.w * dy * dy);
            const float G = __builtin_amdgcn_exp2f(p2);
            const float al = fminf(0.99f, opac * G);
            const float alpha = (!(al < 1.0f / 255.0f) && !(p2 > 0.f)) ? al : 0.f;
            const float incl = Tin * prefix_product(1.f - alpha);      // transmittance BEHIND each lane's entry
            const float excl = DPP(Tin, incl, 0x138, 0xf);             // wave_shr:1 -> in front of it (lane 0: Tin)
            // the reference stops in front of the first entry that would take T below 1e-4 (forward.cu:336-341): incl is
            // non-increasing over the lanes, so the entries that count form a prefix
            const bool live = !(incl < 0.0001f);                       // (a skipped entry has its predecessor's incl)
            const unsigned long long dead = __ballot(!live);
            const int nlive = dead ? (int)__builtin_ctzll(dead) : 64;
            const float w = (lane < nlive) ? alpha * excl : 0.f;
            const float sr = RDL(sum_to_lane63(w * c.x), 63), sg = RDL(sum_to_lane63(w * c.y), 63), sb = RDL(sum_to_lane63(w * c.z), 63),
                        sd = RDL(sum_to_lane63(w * b.z), 63);
            const float Tout = dead ? -RDL(excl, nlive) : RDL(incl, 63);
            if (lane == p) { Cr += sr; Cg += sg; Cb += sb; Cd += sd; Tv = Tout; }
        }
    }
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    float* o = out + ((size_t)py * tiles_x * 16 + px) * 5;
    o[0] = Cr; o[1] = Cg; o[2] = Cb; o[3] = Cd; o[4] = fabsf(Tv);
}

int main(int argc, char** argv)
{
    const int tiles_x = 40, tiles_y = 30, NT = tiles_x * tiles_y;
    const int Ls[] = {128, 384, 1024};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; pass++)
    for (int L : Ls) {
        // pass 0: splats a few pixels wide scattered over the tile and its surroundings (a lane = pixel wave finds a third of its
        // lanes outside an entry's footprint); pass 1: splats covering the whole tile (every lane busy in both mappings)
        std::vector<float> h((size_t)NT * L * REC);
        srand(7);
        auto U = [](float lo, float hi) { return lo + (hi - lo) * (float)rand() / (float)RAND_MAX; };
        const float LOG2E = 1.4426950408889634f;
        for (int t = 0; t < NT; t++)
            for (int e = 0; e < L; e++) {
                float* r = &h[((size_t)t * L + e) * REC];
                const float cx = (t % tiles_x) * 16 + 8.f, cy = (t / tiles_x) * 16 + 8.f;
                const float sx = pass ? U(10.f, 30.f) : U(1.5f, 8.f), sy = pass ? U(10.f, 30.f) : U(1.5f, 8.f);
                const float ca = 1.f / (sx * sx), cc = 1.f / (sy * sy), cb = 0.3f * sqrtf(ca * cc) * U(-1.f, 1.f);
                r[0] = cx + U(-16.f, 16.f); r[1] = cy + U(-16.f, 16.f);
                r[2] = -cb * LOG2E; r[3] = -0.5f * cc * LOG2E; r[4] = -0.5f * ca * LOG2E;
                r[5] = pass ? U(0.01f, 0.08f) : U(0.05f, 0.5f); r[6] = 0.5f + 5.5f * (float)e / L; r[7] = 0.f;
                r[8] = U(0.f, 1.f); r[9] = U(0.f, 1.f); r[10] = U(0.f, 1.f); r[11] = 0.f;
            }
        float *d_rec, *d_a, *d_b;
        const size_t img = (size_t)tiles_x * 16 * tiles_y * 16 * 5;
        CK(hipMalloc(&d_rec, h.size() * 4)); CK(hipMalloc(&d_a, img * 4)); CK(hipMalloc(&d_b, img * 4));
        CK(hipMemcpy(d_rec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        float ms[2] = {1e9f, 1e9f};
        for (int rep = 0; rep < 6; rep++) {
            float t;
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lane_is_pixel, dim3(NT), dim3(256), 0, 0, d_rec, L, tiles_x, d_a); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); ms[0] = fminf(ms[0], t);
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lane_is_gaussian, dim3(NT), dim3(256), 0, 0, d_rec, L, tiles_x, d_b); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); ms[1] = fminf(ms[1], t);
        }
        std::vector<float> ha(img), hb(img);
        CK(hipMemcpy(ha.data(), d_a, img * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), d_b, img * 4, hipMemcpyDeviceToHost));
        double worst = 0.0, meanT = 0.0;
        for (size_t i = 0; i < img; i++) { worst = fmax(worst, fabs((double)ha[i] - hb[i])); if (i % 5 == 4) meanT += ha[i]; }
        printf("%s  L = %4d entries per tile, %d tiles: lane = pixel %7.1f us   lane = Gaussian %7.1f us   (x %.2f)   max |A - B| = %.2e   mean final T %.3f\n",
               pass ? "wide splats " : "small splats", L, NT, ms[0] * 1e3f, ms[1] * 1e3f, ms[1] / ms[0], worst, meanT / (img / 5));
        CK(hipFree(d_rec)); CK(hipFree(d_a)); CK(hipFree(d_b));
    }
    return 0;
}
