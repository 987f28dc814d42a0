// gsr_gradmask.h -- the per-frame gradient mask of the localisation scripts, on the device.
// Replaces Camera.compute_grad_mask (gs_localization/pipelines/tools/camera_utils.py:164-193) with image_gradient /
// image_gradient_mask (tools/descent_utils.py:33-67) behind it -- a mean over channels, two reflect-padded 3x3 convolutions, a
// 3x3 validity test, a global torch.median (a full sort in torch) and a comparison: ~25 torch launches and a sort of H*W floats
// per query frame -- and the `grad_mask | create_mask(keypoints)` step of the scripts (7scenes_localize_full_dslam.py:126-149,
// 355-360: a Python loop over the keypoints on the host and an upload).  SURVEY.md section 8(f)-1.
//
// Arithmetic (what tests/golden/grad_mask_vectors.npz, produced by the imported reference on torch's CPU backend, pins bit for bit
// on the mask): gray = ((r + g) + b) / 3; each gradient a chain of fused multiply-adds over the filter taps in row-major order,
// then x 1/32; intensity = sqrt(gv * gv + gh * gh), every step rounded to float32 (IEEE square root: what a CUDA run of the
// reference computes; torch's CPU sqrt is 1 ulp low on 0.6 % of inputs); the LOWER median by an exact three-level radix select on
// the bit patterns (non-negative floats order like their bits); threshold = fl32(median * fl32(edge_threshold)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsr {

#define GSR_GM_TW 64            // pixels per tile row: one wave
#define GSR_GM_TH 16            // tile rows: four per wave of the workgroup (a 64 x 4 tile re-read its halo rows 1.5 x: 5.3 MB fetched for a
                                // 3.7 MB picture, profiles/r06_traffic.json; 64 x 16: 1.16 x, and a quarter of the histogram flushes)
#define GSR_GM_BINS 2048        // levels 1 and 2: 11 bits each; level 3: the remaining 9 bits (512 bins)

struct GradMaskArgs {
    int W, H;
    const float* image;         // [3, H, W]
    float* intensity;           // [H, W]
    uint32_t* hist;             // [3][GSR_GM_BINS]
    uint32_t rank;              // (H W - 1) / 2: torch.median's element
    float edge_threshold;
    uint8_t* mask;              // [H, W]
    float* median_out;          // nullable: [0] median, [1] threshold
};

__device__ __forceinline__ uint32_t gm_bits(float q) { return __float_as_uint(q) & 0x7fffffffu; }      // (a NaN may carry a sign bit)

__device__ __forceinline__ int gm_reflect(int i, int n)          // torch's "reflect" padding by one; beyond that: clamped (unused lanes)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// Which bin of hist[0 .. nb) holds the element of rank k (0-based), and its rank inside that bin.  256 threads, nb = 256 * per.
__device__ __forceinline__ void gm_pick(const uint32_t* __restrict__ hist, int per, uint32_t k, uint32_t* s_tmp /*[8]*/, uint32_t& bin, uint32_t& k_rem)
{
    const int t = threadIdx.x;
    uint32_t c[8];
    uint32_t sum = 0;
    for (int j = 0; j < per; j++) { c[j] = hist[t * per + j]; sum += c[j]; }
    uint32_t incl = sum;          // inclusive scan over the 256 threads: inside the wave by shuffles, across the four waves through LDS
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((t & 63) >= off) incl += o;
    }
    __syncthreads();          // (s_tmp may still be read from a previous pick)
    if ((t & 63) == 63) s_tmp[t >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (t >> 6); w++) base += s_tmp[w];
    incl += base;
    const uint32_t excl = incl - sum;
    if (k >= excl && k < incl) {
        uint32_t before = excl;
        int j = 0;
        while (j < per - 1 && k >= before + c[j]) { before += c[j]; j++; }
        s_tmp[4] = (uint32_t)(t * per + j);
        s_tmp[5] = k - before;
    }
    __syncthreads();
    bin = s_tmp[4];
    k_rem = s_tmp[5];
}

__device__ __forceinline__ void gm_flush(uint32_t* s_hist, uint32_t* __restrict__ ghist, int nb)
{
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += 256) {
        const uint32_t c = s_hist[b];
        if (c) atomicAdd(&ghist[b], c);
    }
}

// Launch 1: intensity of every pixel + the histogram of its top 11 value bits.
__global__ void __launch_bounds__(256) k_gradmask_intensity(GradMaskArgs a)
{
    __shared__ float s_gray[GSR_GM_TH + 2][GSR_GM_TW + 2];
    __shared__ uint32_t s_hist[GSR_GM_BINS];
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * GSR_GM_TW, y0 = blockIdx.y * GSR_GM_TH;
    const size_t n = (size_t)a.W * a.H;
    for (int b = t; b < GSR_GM_BINS; b += 256) s_hist[b] = 0u;
    for (int e = t; e < (GSR_GM_TH + 2) * (GSR_GM_TW + 2); e += 256) {
        const int i = e / (GSR_GM_TW + 2), j = e - i * (GSR_GM_TW + 2);
        const size_t o = (size_t)gm_reflect(y0 - 1 + i, a.H) * a.W + gm_reflect(x0 - 1 + j, a.W);
        s_gray[i][j] = ((a.image[o] + a.image[n + o]) + a.image[2 * n + o]) / 3.0f;
    }
    __syncthreads();
    const int lx = t & 63;
    const int x = x0 + lx;
#pragma unroll
    for (int r = 0; r < GSR_GM_TH / 4; r++) {
        const int ly = (t >> 6) + 4 * r, y = y0 + ly;
        if (!(x < a.W && y < a.H)) continue;
        const float p00 = s_gray[ly][lx], p01 = s_gray[ly][lx + 1], p02 = s_gray[ly][lx + 2];
        const float p10 = s_gray[ly + 1][lx], p11 = s_gray[ly + 1][lx + 1], p12 = s_gray[ly + 1][lx + 2];
        const float p20 = s_gray[ly + 2][lx], p21 = s_gray[ly + 2][lx + 1], p22 = s_gray[ly + 2][lx + 2];
        // image_gradient (descent_utils.py:33-50): taps in row-major order, fused multiply-adds from 0, then the normaliser 1/32
        float gv = __builtin_fmaf(p00, 3.f, 0.f);
        gv = __builtin_fmaf(p01, 10.f, gv); gv = __builtin_fmaf(p02, 3.f, gv);
        gv = __builtin_fmaf(p20, -3.f, gv); gv = __builtin_fmaf(p21, -10.f, gv); gv = __builtin_fmaf(p22, -3.f, gv);
        float gh = __builtin_fmaf(p00, 3.f, 0.f);
        gh = __builtin_fmaf(p02, -3.f, gh); gh = __builtin_fmaf(p10, 10.f, gh);
        gh = __builtin_fmaf(p12, -10.f, gh); gh = __builtin_fmaf(p20, 3.f, gh); gh = __builtin_fmaf(p22, -3.f, gh);
        gv *= 0.03125f; gh *= 0.03125f;
        // image_gradient_mask (descent_utils.py:53-67): all nine |gray| > 0.01
        const float eps = 0.01f;
        const bool ok = fabsf(p00) > eps && fabsf(p01) > eps && fabsf(p02) > eps && fabsf(p10) > eps && fabsf(p11) > eps &&
                        fabsf(p12) > eps && fabsf(p20) > eps && fabsf(p21) > eps && fabsf(p22) > eps;
        const float m = ok ? 1.f : 0.f;
        gv *= m; gh *= m;
        const float q = sqrtf(gv * gv + gh * gh);          // camera_utils.py:172
        a.intensity[(size_t)y * a.W + x] = q;
        atomicAdd(&s_hist[gm_bits(q) >> 20], 1u);
    }
    gm_flush(s_hist, a.hist, GSR_GM_BINS);
}

// Launches 2 and 3: the histogram of the next bits among the elements that share the median's leading bits.
template <int LEVEL>
__global__ void __launch_bounds__(256) k_gradmask_hist(GradMaskArgs a)
{
    __shared__ uint32_t s_hist[GSR_GM_BINS];
    __shared__ uint32_t s_tmp[8];
    const int t = threadIdx.x;
    constexpr int NB = LEVEL == 2 ? GSR_GM_BINS : 512;
    for (int b = t; b < NB; b += 256) s_hist[b] = 0u;
    uint32_t b1, k1, prefix, k = a.rank;
    gm_pick(a.hist, GSR_GM_BINS / 256, k, s_tmp, b1, k1);
    prefix = b1;
    if (LEVEL == 3) {
        uint32_t b2, k2;
        gm_pick(a.hist + GSR_GM_BINS, GSR_GM_BINS / 256, k1, s_tmp, b2, k2);
        prefix = (b1 << 11) | b2;
    }
    __syncthreads();
    const size_t n = (size_t)a.W * a.H;
    for (size_t i = (size_t)blockIdx.x * 256 + t; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t v = gm_bits(a.intensity[i]);
        if (LEVEL == 2) { if ((v >> 20) == prefix) atomicAdd(&s_hist[(v >> 9) & 2047u], 1u); }
        else            { if ((v >> 9) == prefix) atomicAdd(&s_hist[v & 511u], 1u); }
    }
    gm_flush(s_hist, a.hist + (LEVEL - 1) * GSR_GM_BINS, NB);
}

// Launch 4: the median from the three histograms, the threshold, the mask (camera_utils.py:189-193).
__global__ void __launch_bounds__(256) k_gradmask_threshold(GradMaskArgs a)
{
    __shared__ uint32_t s_tmp[8];
    uint32_t b1, k1, b2, k2, b3, k3;
    gm_pick(a.hist, GSR_GM_BINS / 256, a.rank, s_tmp, b1, k1);
    gm_pick(a.hist + GSR_GM_BINS, GSR_GM_BINS / 256, k1, s_tmp, b2, k2);
    gm_pick(a.hist + 2 * GSR_GM_BINS, 512 / 256, k2, s_tmp, b3, k3);
    const float median = __uint_as_float((b1 << 20) | (b2 << 9) | b3);
    const float thr = median * a.edge_threshold;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.median_out != nullptr) { a.median_out[0] = median; a.median_out[1] = thr; }
    const size_t n = (size_t)a.W * a.H;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        a.mask[i] = a.intensity[i] > thr ? 1 : 0;
}

// Launch 5 (only with keypoints): create_mask (7scenes_localize_full_dslam.py:126-149) OR-ed in -- a box of 2 (k / 2) + 1 pixels
// around (int(x), int(y)) of every keypoint, clipped to the image.  One wave per keypoint.
__global__ void __launch_bounds__(256) k_gradmask_boxes(int W, int H, const float* __restrict__ kp, int nk, int half, uint8_t* __restrict__ mask)
{
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= nk) return;
    const int cx = (int)kp[2 * k], cy = (int)kp[2 * k + 1];          // Python's int(): truncation towards zero
    const int side = 2 * half + 1;
    for (int e = threadIdx.x & 63; e < side * side; e += 64) {
        const int y = cy - half + e / side, x = cx - half + e % side;
        if (x >= 0 && x < W && y >= 0 && y < H) mask[(size_t)y * W + x] = 1;
    }
}

// The "replica" branch (camera_utils.py:174-188): a rows x cols grid of int(H / rows) x int(W / cols) blocks, each thresholded at
// its own lower median x multiplier; the result is a FLOAT image: 1 / 0 inside the grid -- the two in-place writes of the reference
// run one after the other, so the ones written first are cleared again whenever 1 <= threshold -- and the raw intensity outside it
// (the caller copies the intensity image into `out` first).  One workgroup per block; the same three-level radix select in LDS.
__global__ void __launch_bounds__(256) k_gradmask_replica(int W, int H, int bw, int bh, int cols, const float* __restrict__ intensity,
                                                          float multiplier, float* __restrict__ out)
{
    __shared__ uint32_t s_hist[GSR_GM_BINS];
    __shared__ uint32_t s_tmp[8];
    const int t = threadIdx.x;
    const int r = blockIdx.x / cols, c = blockIdx.x - r * cols;
    const int n = bw * bh;
    const float* src = intensity + (size_t)r * bh * W + (size_t)c * bw;
    uint32_t prefix = 0, k = (uint32_t)(n - 1) / 2u, bin;
#pragma unroll 1
    for (int level = 1; level <= 3; level++) {
        const int nb = level == 3 ? 512 : GSR_GM_BINS;
        __syncthreads();
        for (int b = t; b < nb; b += 256) s_hist[b] = 0u;
        __syncthreads();
        for (int e = t; e < n; e += 256) {
            const uint32_t v = gm_bits(src[(size_t)(e / bw) * W + (e % bw)]);
            if (level == 1) atomicAdd(&s_hist[v >> 20], 1u);
            else if (level == 2) { if ((v >> 20) == prefix) atomicAdd(&s_hist[(v >> 9) & 2047u], 1u); }
            else if ((v >> 9) == prefix) atomicAdd(&s_hist[v & 511u], 1u);
        }
        __syncthreads();
        gm_pick(s_hist, nb / 256, k, s_tmp, bin, k);
        prefix = level == 1 ? bin : (level == 2 ? ((prefix << 11) | bin) : ((prefix << 9) | bin));
    }
    const float thr = __uint_as_float(prefix) * multiplier;
    const bool ones_survive = !(1.0f <= thr);
    for (int e = t; e < n; e += 256) {
        const size_t o = (size_t)(e / bw) * W + (e % bw);
        out[(size_t)r * bh * W + (size_t)c * bw + o] = (src[o] > thr && ones_survive) ? 1.f : 0.f;
    }
}

}  // namespace gsr
