// gsr_device.h -- device-side helpers shared by the gfx950 kernels.
// wave = 64 lanes, 16x16 pixel tile = 4 waves.  fp32 throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GSR_TILE 16
#define GSR_BLOCK 256
#define GSR_WAVE 64

namespace gsr {

__device__ __constant__ const float kSH_C0 = 0.28209479177387814f;
__device__ __constant__ const float kSH_C1 = 0.4886025119029199f;
__device__ __constant__ const float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                                 -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ const float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                                 0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                                 -0.5900435899266435f};

// 3x3 matrix stored column-major: m[c][r]  (same convention as the reference's maths library)
struct M3 {
    float m[3][3];
};

__device__ __forceinline__ M3 m3_mul(const M3& a, const M3& b)
{
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            r.m[i][j] = a.m[0][j] * b.m[i][0] + a.m[1][j] * b.m[i][1] + a.m[2][j] * b.m[i][2];
    return r;
}
__device__ __forceinline__ M3 m3_T(const M3& a)
{
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            r.m[i][j] = a.m[j][i];
    return r;
}
__device__ __forceinline__ M3 m3_cols(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1,
                                      float c2)
{
    M3 r;
    r.m[0][0] = a0; r.m[0][1] = a1; r.m[0][2] = a2;
    r.m[1][0] = b0; r.m[1][1] = b1; r.m[1][2] = b2;
    r.m[2][0] = c0; r.m[2][1] = c1; r.m[2][2] = c2;
    return r;
}

struct Cam {               // camera constants, loaded once per thread from uniform addresses (SGPRs)
    float v[16];           // viewmatrix  = (W2C)^T row-major
    float p[16];           // projmatrix  = (P*W2C)^T row-major
    float c[3];            // camera centre
};
__device__ __forceinline__ void load_cam(Cam& cam, const float* view, const float* proj, const float* campos)
{
#pragma unroll
    for (int i = 0; i < 16; i++) { cam.v[i] = view[i]; cam.p[i] = proj[i]; }
    cam.c[0] = campos[0]; cam.c[1] = campos[1]; cam.c[2] = campos[2];
}

__device__ __forceinline__ float3 xform4x3(float3 p, const float* m)
{
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(float3 p, const float* m)
{
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

// pixel centre from NDC; evaluated in fp64 like the reference (auxiliary.h:41-44)
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// tile rectangle of the reference's bounding rule (auxiliary.h:46-56); C truncation toward zero
__device__ __forceinline__ void get_rect(float px, float py, int rad, int gx, int gy, int& x0, int& y0, int& x1, int& y1)
{
    x0 = min(gx, max(0, (int)((px - rad) / GSR_TILE)));
    y0 = min(gy, max(0, (int)((py - rad) / GSR_TILE)));
    x1 = min(gx, max(0, (int)((px + rad + GSR_TILE - 1) / GSR_TILE)));
    y1 = min(gy, max(0, (int)((py + rad + GSR_TILE - 1) / GSR_TILE)));
}

// world covariance (6 unique entries) from scale + quaternion used AS GIVEN (forward.cu:118-152)
__device__ __forceinline__ void cov3d_from_scale_rot(const float* s3, float mod, const float* q4, float* cov6)
{
    M3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * s3[0];
    S.m[1][1] = mod * s3[1];
    S.m[2][2] = mod * s3[2];
    float r = q4[0], x = q4[1], y = q4[2], z = q4[3];
    M3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    M3 Mm = m3_mul(S, R);
    M3 Sig = m3_mul(m3_T(Mm), Mm);
    cov6[0] = Sig.m[0][0]; cov6[1] = Sig.m[0][1]; cov6[2] = Sig.m[0][2];
    cov6[3] = Sig.m[1][1]; cov6[4] = Sig.m[1][2]; cov6[5] = Sig.m[2][2];
}

struct Cov2DTerms {
    float3 t;        // camera-space mean with the 1.3*tanfov clamp applied to x,y
    float txtz, tytz;
    M3 T, Vrk, W, cov;
};
// EWA projection (forward.cu:74-113); also the recompute step of backward.cu:166-199
__device__ __forceinline__ void cov2d_terms(float3 mean, float fx, float fy, float tanx, float tany, const float* cov6,
                                            const float* view, Cov2DTerms& o)
{
    float3 t = xform4x3(mean, view);
    const float limx = 1.3f * tanx;
    const float limy = 1.3f * tany;
    o.txtz = t.x / t.z;
    o.tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, o.txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, o.tytz)) * t.z;
    M3 J = m3_cols(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z), 0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z), 0, 0, 0);
    o.W = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    o.T = m3_mul(o.W, J);
    o.Vrk = m3_cols(cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]);
    o.cov = m3_mul(m3_mul(m3_T(o.T), m3_T(o.Vrk)), o.T);
    o.t = t;
}

// ---- wave64 DPP reductions (gfx9 DPP: row_shr within 16-lane rows, then row_bcast15/31) ----
#define GSR_DPP_ADD(v, ctrl, rm) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rm, 0xf, true))
// total of all 64 lanes ends up in lane 63 (other lanes hold partial sums)
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    GSR_DPP_ADD(v, 0x111, 0xf);   // row_shr:1
    GSR_DPP_ADD(v, 0x112, 0xf);   // row_shr:2
    GSR_DPP_ADD(v, 0x114, 0xf);   // row_shr:4
    GSR_DPP_ADD(v, 0x118, 0xf);   // row_shr:8   -> lane 15 of each row = row total
    GSR_DPP_ADD(v, 0x142, 0xa);   // row_bcast:15 into rows 1,3
    GSR_DPP_ADD(v, 0x143, 0xc);   // row_bcast:31 into rows 2,3
    return v;
}
// Ten 16-lane row sums at once; the total of row r ends up in lane 16 r + 15.  Written as ONE asm block so
// that the four DPP steps of the ten independent values interleave: a DPP source written by the previous
// VALU instruction needs two wait states, which the nine other values provide for free (hipcc serialises
// each value's chain with s_nop pairs).  40 VALU instructions; the four row totals are then combined by
// the LDS float atomics that merge the four waves anyway.
#define GSR_R10_STEP(ctrl) \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\tv_add_f32_dpp %2, %2, %2 " ctrl "\n\t" \
    "v_add_f32_dpp %3, %3, %3 " ctrl "\n\tv_add_f32_dpp %4, %4, %4 " ctrl "\n\tv_add_f32_dpp %5, %5, %5 " ctrl "\n\t" \
    "v_add_f32_dpp %6, %6, %6 " ctrl "\n\tv_add_f32_dpp %7, %7, %7 " ctrl "\n\tv_add_f32_dpp %8, %8, %8 " ctrl "\n\t" \
    "v_add_f32_dpp %9, %9, %9 " ctrl "\n\t"
__device__ __forceinline__ void row_sum10_to_lane15(float (&v)[10])
{
    asm volatile("s_nop 1\n\t"
                 GSR_R10_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 GSR_R10_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 GSR_R10_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 GSR_R10_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                   "+v"(v[8]), "+v"(v[9]));
}

__device__ __forceinline__ float wave_sum(float v)
{
    v = wave_sum_to_lane63(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// fp64 total of all 64 lanes in lane 63 (other lanes hold partial sums), on the DPP path like wave_sum_to_lane63: two 32-bit DPP
// moves + one v_add_f64 per step instead of two ds_bpermute round trips through the LDS pipe (a dozen of these sums close the
// chain-rule kernel's critical path: 6.5 k cycles per wave with the shuffles)
__device__ __forceinline__ double wave_sum_d_to_lane63(double v)
{
#define GSR_DPP_ADD_D(ctrl, rm)                                                                                          \
    {                                                                                                                    \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rm, 0xf, true);                          \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rm, 0xf, true);                          \
        v += __hiloint2double(hi_, lo_);                                                                                 \
    }
    GSR_DPP_ADD_D(0x111, 0xf)   // row_shr:1
    GSR_DPP_ADD_D(0x112, 0xf)   // row_shr:2
    GSR_DPP_ADD_D(0x114, 0xf)   // row_shr:4
    GSR_DPP_ADD_D(0x118, 0xf)   // row_shr:8   -> lane 15 of each row = row total
    GSR_DPP_ADD_D(0x142, 0xa)   // row_bcast:15 into rows 1, 3
    GSR_DPP_ADD_D(0x143, 0xc)   // row_bcast:31 into rows 2, 3
#undef GSR_DPP_ADD_D
    return v;
}

// ---- deterministic accumulation (the `deterministic` option: gsr.h GSR_REFINE_DETERMINISTIC, debug bit 2 of the backward) ----
// Floating-point atomics add in whatever order the workgroups arrive, and fp32 / fp64 addition is not associative: two runs of the same
// backward differ in the last bits of every gradient that more than one tile contributes to.  In the deterministic mode everything
// that is summed ACROSS workgroups -- the per-(tile, splat) gradient sums, the per-Gaussian terms of dL/dtau, the fused loss's partial
// sums -- is converted to 64-bit fixed point first and added as integers: associative, so the result no longer depends on the order
// (sums WITHIN a workgroup already run in a fixed order).  The per-(tile, Gaussian) gradient sums span many orders of magnitude -- the
// conic's moments of a splat hundreds of pixels wide under O(1) pixel gradients reach 1e10 per tile and cancel to 1e3 over the tiles,
// under a mean loss they are 1e-6 -- so each takes TWO words: a coarse one (multiples of 2^-8, range +-2^55) and the remainder (2^-56);
// neither sum can overflow, no carry passes between them, and their total is exact to 2^-57 per addend (fixed_split / fixed_join).
// Pose terms: 2^-32 / +-2^31 per Gaussian; loss sums: 2^-30 / +-2^33 per tile (an addend outside the range wraps: INTEGRATION.md).
#define GSR_FIX_TAU 4294967296.0          // 2^32
#define GSR_FIX_LOSS 1073741824.0         // 2^30
__device__ __forceinline__ long long to_fixed(float v, double scale) { return __double2ll_rn((double)v * scale); }
__device__ __forceinline__ double from_fixed(long long v, double scale) { return (double)v * (1.0 / scale); }
__device__ __forceinline__ void fixed_split(float v, long long& hi, long long& lo)
{
    const double d = (double)v;
    hi = __double2ll_rn(d * 256.0);
    lo = __double2ll_rn((d - (double)hi * (1.0 / 256.0)) * 72057594037927936.0);      // (the remainder, at most 2^-9 in magnitude, in units of 2^-56)
}
__device__ __forceinline__ double fixed_join(long long hi, long long lo) { return (double)hi * (1.0 / 256.0) + (double)lo * (1.0 / 72057594037927936.0); }
// 64-bit integer total of all 64 lanes in lane 63 (other lanes: partial sums), DPP like wave_sum_d_to_lane63
__device__ __forceinline__ long long wave_sum_ll_to_lane63(long long v)
{
#define GSR_DPP_ADD_LL(ctrl, rm)                                                                                         \
    {                                                                                                                    \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(unsigned long long)v, ctrl, rm, 0xf, true); \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((unsigned long long)v >> 32), ctrl, rm, 0xf, true); \
        v += (long long)(((unsigned long long)hi_ << 32) | lo_);                                                         \
    }
    GSR_DPP_ADD_LL(0x111, 0xf)
    GSR_DPP_ADD_LL(0x112, 0xf)
    GSR_DPP_ADD_LL(0x114, 0xf)
    GSR_DPP_ADD_LL(0x118, 0xf)
    GSR_DPP_ADD_LL(0x142, 0xa)
    GSR_DPP_ADD_LL(0x143, 0xc)
#undef GSR_DPP_ADD_LL
    return v;
}

// XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give each XCD a contiguous
// run of tiles; bijective for any n (cdna_hip_programming.md section 5, "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int b, int n)
{
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

}  // namespace gsr
