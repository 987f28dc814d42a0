/* Plain-C consumer of include/gsr.h: what a maintainer's binding sees without torch or Python.
 * Renders 3 splats on a 48x32 image through gsr_forward, differentiates through gsr_backward (pose package), checks the
 * status codes, that the centre pixel is covered, that the image is finite and that dL/dtau is non-zero; then the per-frame
 * gradient mask of the localisers (gsr_grad_mask) on that image.
 * Build (tests/test_c_abi.py does it): gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude smoke.c
 *                                      -L<repo>/gs_localization_amd -lgsr_hip -L/opt/rocm/lib -lamdhip64 -lm */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gsr.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef struct { void* p; size_t cap; } buf_t;
static void* resize_cb(void* ctx, size_t bytes)          /* the std::function<char*(size_t)> of rasterize_points.cu:27-33 */
{
    buf_t* b = (buf_t*)ctx;
    if (bytes > b->cap) {
        if (b->p) (void)hipFree(b->p);
        if (hipMalloc(&b->p, bytes) != hipSuccess) return NULL;
        b->cap = bytes;
    }
    return b->p;
}
static float* upload(const float* h, size_t n)
{
    float* d = NULL;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
    if (hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}
static float* device_floats(size_t n)
{
    float* d = NULL;
    return hipMalloc((void**)&d, n * sizeof(float)) == hipSuccess ? d : NULL;
}

int main(void)
{
    if (gsr_abi_version() != GSR_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    if (!gsr_device_ok()) { fprintf(stderr, "no gfx950 device\n"); return 77; }
    enum { P = 3, W = 48, H = 32, N = W * H };
    const float fx = 40.f, fy = 40.f, zn = 0.01f, zf = 100.f;
    const float tanx = W / (2.f * fx), tany = H / (2.f * fy);
    /* identity camera: viewmatrix = (W2C)^T = I; projmatrix = viewmatrix * P^T (row-major), P of graphics_utils.py:77-98 */
    float view[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, proj[16];
    memset(proj, 0, sizeof(proj));
    proj[0] = 1.f / tanx; proj[5] = 1.f / tany; proj[10] = zf / (zf - zn); proj[14] = -(zf * zn) / (zf - zn); proj[11] = 1.f;
    const float campos[3] = {0, 0, 0}, bg[3] = {0, 0, 0};
    const float means[P * 3] = {0.f, 0.f, 2.f, 0.3f, -0.1f, 3.f, -0.4f, 0.2f, 2.5f};
    const float scales[P * 3] = {0.2f, 0.1f, 0.1f, 0.15f, 0.15f, 0.1f, 0.1f, 0.2f, 0.1f};
    const float rots[P * 4] = {1, 0, 0, 0, 0.9238795f, 0, 0.3826834f, 0, 0.7071068f, 0.7071068f, 0, 0};
    const float opac[P] = {0.9f, 0.8f, 0.7f};
    const float colors[P * 3] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float *d_means = upload(means, P * 3), *d_scales = upload(scales, P * 3), *d_rots = upload(rots, P * 4), *d_opac = upload(opac, P),
          *d_col = upload(colors, P * 3), *d_view = upload(view, 16), *d_proj = upload(proj, 16), *d_campos = upload(campos, 3),
          *d_bg = upload(bg, 3);
    float *out_color = device_floats(3 * N), *out_depth = device_floats(N), *out_alpha = device_floats(N);
    int *radii = NULL, *touched = NULL;
    CK(hipMalloc((void**)&radii, P * sizeof(int)));
    CK(hipMalloc((void**)&touched, P * sizeof(int)));
    if (!d_means || !d_scales || !d_rots || !d_opac || !d_col || !d_view || !d_proj || !d_campos || !d_bg || !out_color || !out_depth || !out_alpha)
        return 2;
    buf_t geom = {0, 0}, binning = {0, 0}, img = {0, 0};
    hipStream_t st;
    CK(hipStreamCreate(&st));

    const int R = gsr_forward(resize_cb, &geom, resize_cb, &binning, resize_cb, &img, P, 0, 0, d_bg, W, H, d_means, NULL, d_col, d_opac, d_scales,
                              1.0f, d_rots, NULL, d_view, d_proj, d_campos, tanx, tany, 0, out_color, out_depth, out_alpha, radii,
                              /* debug bit 2: this forward's buffers will also feed a DETERMINISTIC backward below (64-bit accumulator records) */ 4, touched, st);
    if (R < 0) { fprintf(stderr, "gsr_forward: %s\n", gsr_last_error()); return 1; }
    CK(hipStreamSynchronize(st));
    static float h_color[3 * N], h_alpha[N];
    int h_radii[P];
    CK(hipMemcpy(h_color, out_color, sizeof(h_color), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_alpha, out_alpha, sizeof(h_alpha), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_radii, radii, sizeof(h_radii), hipMemcpyDeviceToHost));
    for (int i = 0; i < 3 * N; i++)
        if (!isfinite(h_color[i])) { fprintf(stderr, "non-finite colour\n"); return 1; }
    const int centre = (H / 2) * W + W / 2;
    if (!(R > 0 && h_radii[0] > 0 && h_alpha[centre] > 0.5f && h_color[centre] > 0.4f)) {
        fprintf(stderr, "unexpected render: R=%d radius0=%d alpha=%g red=%g\n", R, h_radii[0], h_alpha[centre], h_color[centre]);
        return 1;
    }

    /* backward, pose package semantics: dL/dcolour = 1 everywhere */
    static float ones[3 * N];
    for (int i = 0; i < 3 * N; i++) ones[i] = 1.f;
    float *g_pix = upload(ones, 3 * N), *g_depth = device_floats(N), *g_alpha = device_floats(N);
    CK(hipMemset(g_depth, 0, N * sizeof(float)));
    CK(hipMemset(g_alpha, 0, N * sizeof(float)));
    float *g_m2d = device_floats(P * 3), *g_conic = device_floats(P * 4), *g_opac = device_floats(P), *g_col = device_floats(P * 3),
          *g_m3d = device_floats(P * 3), *g_cov = device_floats(P * 6), *g_scale = device_floats(P * 3), *g_rot = device_floats(P * 4),
          *g_tau = device_floats(6);
    int rc = gsr_backward(P, 0, 0, R, d_bg, W, H, d_means, NULL, d_col, out_alpha, d_scales, 1.0f, d_rots, NULL, d_view, d_proj, d_campos, tanx, tany,
                          radii, (char*)geom.p, (char*)binning.p, (char*)img.p, g_pix, g_depth, g_alpha, g_m2d, g_conic, g_opac, g_col, g_m3d, g_cov,
                          NULL, g_scale, g_rot, 0, 1, g_tau, st);
    if (rc < 0) { fprintf(stderr, "gsr_backward: %s\n", gsr_last_error()); return 1; }
    CK(hipStreamSynchronize(st));
    float h_tau[6], h_gcol[P * 3], tau_norm = 0.f;
    CK(hipMemcpy(h_tau, g_tau, sizeof(h_tau), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_gcol, g_col, sizeof(h_gcol), hipMemcpyDeviceToHost));
    for (int i = 0; i < 6; i++) tau_norm += h_tau[i] * h_tau[i];
    if (!(tau_norm > 0.f && isfinite(tau_norm) && h_gcol[0] > 1.f)) { fprintf(stderr, "unexpected gradients: |tau|^2=%g dL/dcol0=%g\n", tau_norm, h_gcol[0]); return 1; }

    /* the deterministic option (debug bit 2): twice the same bits, and the default mode's numbers up to rounding */
    float h_det[2][6], h_detcol[2][P * 3];
    for (int rep = 0; rep < 2; rep++) {
        rc = gsr_backward(P, 0, 0, R, d_bg, W, H, d_means, NULL, d_col, out_alpha, d_scales, 1.0f, d_rots, NULL, d_view, d_proj, d_campos, tanx, tany,
                          radii, (char*)geom.p, (char*)binning.p, (char*)img.p, g_pix, g_depth, g_alpha, g_m2d, g_conic, g_opac, g_col, g_m3d, g_cov,
                          NULL, g_scale, g_rot, 4, 1, g_tau, st);
        if (rc < 0) { fprintf(stderr, "gsr_backward (deterministic): %s\n", gsr_last_error()); return 1; }
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h_det[rep], g_tau, sizeof(h_tau), hipMemcpyDeviceToHost));
        CK(hipMemcpy(h_detcol[rep], g_col, sizeof(h_gcol), hipMemcpyDeviceToHost));
    }
    if (memcmp(h_det[0], h_det[1], sizeof(h_tau)) != 0 || memcmp(h_detcol[0], h_detcol[1], sizeof(h_gcol)) != 0) { fprintf(stderr, "deterministic backward: two runs differ\n"); return 1; }
    for (int i = 0; i < 6; i++)
        if (fabsf(h_det[0][i] - h_tau[i]) > 1e-4f * sqrtf(tau_norm)) { fprintf(stderr, "deterministic dL/dtau[%d] = %g against %g\n", i, h_det[0][i], h_tau[i]); return 1; }

    /* error path: exactly one of shs / colors_precomp */
    rc = gsr_forward(resize_cb, &geom, resize_cb, &binning, resize_cb, &img, P, 0, 0, d_bg, W, H, d_means, NULL, NULL, d_opac, d_scales, 1.0f, d_rots,
                     NULL, d_view, d_proj, d_campos, tanx, tany, 0, out_color, out_depth, out_alpha, radii, 0, NULL, st);
    if (rc != GSR_E_INVALID || strlen(gsr_last_error()) == 0) { fprintf(stderr, "missing colours must be GSR_E_INVALID with a message\n"); return 1; }
    /* the per-frame gradient mask of the localisers (gsr_grad_mask, ABI 5) on the rendered image + one keypoint box: a picture of three
     * splats on black has a median gradient of 0, so the mask is "any gradient at all" -- it must cover part of the frame, not all of
     * it, and the box around (2, 2) -- an empty corner -- must be set */
    {
        buf_t scratch = {NULL, 0};
        unsigned char *d_mask = NULL, h_mask[N];
        const float kp[2] = {2.2f, 2.9f};
        float* d_kp = upload(kp, 2);
        CK(hipMalloc((void**)&d_mask, N));
        rc = gsr_grad_mask(W, H, out_color, 1.1f, d_kp, 1, 10, d_mask, NULL, NULL, resize_cb, &scratch, NULL);
        if (rc < 0) { fprintf(stderr, "gsr_grad_mask: %s\n", gsr_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h_mask, d_mask, N, hipMemcpyDeviceToHost));
        int set = 0;
        for (int i = 0; i < N; i++) { if (h_mask[i] > 1) { fprintf(stderr, "mask byte %d = %d\n", i, h_mask[i]); return 1; } set += h_mask[i]; }
        if (!(set > 49 && set < N) || !h_mask[2 * W + 2] || !h_mask[0] || !h_mask[7 * W + 7] || h_mask[(H - 1) * W + W - 1]) {
            fprintf(stderr, "unexpected mask: %d of %d pixels set\n", set, N); return 1; }
        if (gsr_grad_mask(1, H, out_color, 1.1f, NULL, 0, 10, d_mask, NULL, NULL, resize_cb, &scratch, NULL) != GSR_E_INVALID) {
            fprintf(stderr, "a 1-pixel-wide image must be GSR_E_INVALID (reflect padding)\n"); return 1; }
        printf("grad mask: %d of %d pixels\n", set, N);
    }
    printf("c abi smoke ok: R=%d alpha(centre)=%.3f |dL/dtau|=%.4g\n", R, h_alpha[centre], sqrt(tau_norm));
    return 0;
}
