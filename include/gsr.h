/*
 * gsr.h -- C ABI of the MI355X-native differentiable Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary for the ONE hot path of RPL-CS-UCL/gs_localization:
 * everything below `CudaRasterizer::Rasterizer` in the reference
 *   (gaussian_splatting/submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:20-90)
 * i.e. what rasterize_points.cu:35-227 (the pybind11/torch glue) calls.  Plain pointers and
 * sizes only -- no torch types.  All pointers are DEVICE pointers (HBM) unless stated otherwise;
 * all arrays are dense row-major fp32 unless stated otherwise; "nullable" means the reference
 * passes an empty tensor there (rasterize_points.cu:96-115 hands the dangling data_ptr to
 * kernels that test it against nullptr, forward.cu:205,241).
 *
 * Thread-safety: entry points are re-entrant; they select the device that owns `means3D`
 * themselves and enqueue on the caller's `stream` (a hipStream_t; NULL = legacy default
 * stream, which is what the reference's `<<<grid,block>>>` launches use, forward.cu:396,437).
 * Backward may be called from a different host thread than forward (PyTorch's autograd worker).
 *
 * Errors: every entry point returns >= 0 on success and a negative GSR_E_* code on failure;
 * gsr_last_error() returns the thread-local message (the reference throws std::runtime_error,
 * rasterizer_impl.cu:243-246, auxiliary.h:166-173).
 */
#ifndef GSR_H_INCLUDED
#define GSR_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: the pose state is 112 floats (words 96..109: pose and exposure of the last executed forward / backward); GSR_REFINE_NO_SPLIT;
 *    gsr_debug_seg_stats; gsr_backward refuses debug bit 2 on a geometry workspace whose forward ran without it.
 * 3: gsr_forward_packed / gsr_backward_packed (one struct pointer instead of 34 / 40 arguments); gsr_refine_args gained
 *    colors_precomp / cov3D_precomp at its end; gsr_geometry_bytes_det, gsr_spec_state_bounds_bytes.
 * 2: gsr_refine_args gained `carry_state` (round 2) and the fields behind `stream` (flags, lean_min_P); stats_out is int[4];
 *    pose-state words 41 (ticket) and 84..87 (Adam beta products); gsr_debug_lean_check.  A caller must compare
 *    gsr_abi_version() with the GSR_ABI_VERSION it was compiled against before it passes any struct. */
#define GSR_ABI_VERSION 5

enum {
    GSR_OK = 0,
    GSR_E_INVALID = -1,   /* bad argument (message says which) */
    GSR_E_HIP = -2,       /* a HIP runtime call or kernel failed */
    GSR_E_ALLOC = -3,     /* a resize callback returned NULL */
    GSR_E_NODEVICE = -4   /* no gfx950 device / pointer is not device memory */
};

/* Workspace request callback.  Replaces the three `std::function<char*(size_t N)>` buffer
 * functors of rasterizer.h:31-34 / rasterize_points.cu:27-33: must return a device buffer of
 * at least `bytes` bytes, 256-byte aligned, that stays valid until the matching backward. */
typedef void* (*gsr_resize_fn)(void* ctx, size_t bytes);

/* Replaces CudaRasterizer::Rasterizer::forward (rasterizer.h:31-60, rasterizer_impl.cu:197-339).
 * Same argument order; additions are at the end.
 *   D = active SH degree, M = SH coefficients per channel (0 if colors_precomp is given).
 *   shs [P,M,3] nullable | colors_precomp [P,3] nullable   (exactly one)
 *   scales [P,3] + rotations [P,4] (w,x,y,z; used as given, forward.cu:127) | cov3D_precomp [P,6]
 *   viewmatrix, projmatrix: 16 floats each, (W2C)^T and (P*W2C)^T row-major; cam_pos: 3 floats
 *   out_color [3,H,W], out_depth [1,H,W], out_alpha [1,H,W]  (fully written; need not be zeroed)
 *   radii [P] int32 (fully written)
 *   n_touched [P] int32 nullable -- `diff_gaussian_rasterization_pose` 5th output
 *       (gs_localization/pipelines/tools/__init__.py:130); fully written when given.
 *   debug: bit 0 = the reference's debug flag (synchronise and check after every kernel, auxiliary.h:166-173);
 *       bit 1 (value 2) = diagnostics: SH colours of every visible Gaussian up front (k_sh_color) instead of lazily in the
 *       compositing kernel -- same results (tests/test_gpu_parity.py); the library reads no environment variable here.
 *       bit 2 (value 4) = the deterministic option, see GSR_REFINE_DETERMINISTIC: gsr_backward sums in 64-bit fixed point; the FORWARD
 *       whose buffers it will use must be given the bit too (it asks its geometry callback for the larger accumulator records).
 * Returns num_rendered (the `int rendered` of rasterize_points.cu:82) or a negative error.
 * Performs one blocking device->host read of num_rendered, like rasterizer_impl.cu:282. */
int gsr_forward(gsr_resize_fn geometry_buffer, void* geometry_ctx,
                gsr_resize_fn binning_buffer, void* binning_ctx,
                gsr_resize_fn image_buffer, void* image_ctx,
                int P, int D, int M,
                const float* background,
                int width, int height,
                const float* means3D,
                const float* shs,
                const float* colors_precomp,
                const float* opacities,
                const float* scales,
                float scale_modifier,
                const float* rotations,
                const float* cov3D_precomp,
                const float* viewmatrix,
                const float* projmatrix,
                const float* cam_pos,
                float tan_fovx, float tan_fovy,
                int prefiltered,
                float* out_color,
                float* out_depth,
                float* out_alpha,
                int* radii,
                int debug,
                int* n_touched,
                void* stream);

/* gsr_forward for callers that render a sequence of nearby views through the stateless boundary (the reference's
 * own loops: 7scenes_localize_full_dslam.py:66-91 calls render() fifty times per frame, a few millimetres apart).
 * `state` carries, from one call to the next, how deep every 16x16 tile had to look before all its pixels had
 * terminated.  The next call bins only what lies in front of those depths (x1.05 + 5 cm) -- per tile, without the
 * two global sorts -- and the compositing kernel VERIFIES the guess: a tile that reaches the end of a shortened
 * list with an unsaturated pixel raises a flag, and the forward is redone with complete lists before this function
 * returns.  Outputs are therefore always those of gsr_forward (same lists up to each pixel's termination, same
 * summation order); a wrong guess (a new frame, another map) only costs time, and after repeated misses the guess
 * is tried less often (2, 4, ... 64 calls apart).  One blocking device->host read per forward, as gsr_forward.
 * The saved buffers feed gsr_backward unchanged; the return value is what gsr_backward expects as R (0 when the
 * lists were binned by tile: the backward then finds them through the image buffer).
 *   state->device_buffer: gsr_spec_state_bytes(width, height) bytes on the GPU, owned by the caller, contents opaque,
 *   need not be initialised; all other fields zero before the first call.  One state per (thread, stream, image size).
 *   state == NULL or device_buffer == NULL: plain gsr_forward. */
typedef struct gsr_spec_state {
    void* device_buffer;
    int width, height;          /* set by the first call; later calls must match */
    int valid;                  /* bounds of a previous call are recorded (write 0 to drop them) */
    int parity;                 /* which of the two bound buffers is current */
    int fail_streak, skip;      /* back-off after misses */
    int last_speculative;       /* out: 1 if the last forward was composed from speculative bins */
    int n_speculative, n_failed;/* counters: verified / missed guesses */
} gsr_spec_state;
size_t gsr_spec_state_bytes(int width, int height);
/* The leading bytes of a state's device buffer that hold the recorded depth bounds (the rest is scratch): a caller that keeps one
 * state per camera seeds a NEW camera's state by copying that much from the state of the view it rendered last (and the host
 * fields valid / parity) -- a guess like any other, verified like any other. */
size_t gsr_spec_state_bounds_bytes(int width, int height);
int gsr_forward_speculative(gsr_spec_state* state,
                            gsr_resize_fn geometry_buffer, void* geometry_ctx,
                            gsr_resize_fn binning_buffer, void* binning_ctx,
                            gsr_resize_fn image_buffer, void* image_ctx,
                            int P, int D, int M, const float* background, int width, int height,
                            const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                            const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                            const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                            float tan_fovx, float tan_fovy, int prefiltered,
                            float* out_color, float* out_depth, float* out_alpha, int* radii, int debug,
                            int* n_touched, void* stream);

/* Replaces CudaRasterizer::Rasterizer::backward (rasterizer.h:62-89, rasterizer_impl.cu:343-444).
 *   R = value returned by gsr_forward; geom/binning/img buffers = the ones it filled.
 *   dL_dpix [3,H,W], dL_ddepths [1,H,W], dL_dalphas [1,H,W].
 *   Outputs (all fully written by this call -- no pre-zeroing needed, unlike
 *   rasterize_points.cu:158-166):
 *     dL_dmean2D [P,3] (z = 0), dL_dconic [P,4] (x,y,_,w used), dL_dopacity [P], dL_dcolor [P,3],
 *     dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3] (nullable iff M == 0),
 *     dL_dscale [P,3], dL_drot [P,4].
 *   pose_mode = 0: package (A) semantics, faithful to the vendored backward.cu (incl. its quirks).
 *   pose_mode = 1: package (B): the per-Gaussian depth also back-propagates into dL_dmean3D and
 *     dL_dtau [6] = [dL/drho(3), dL/dtheta(3)] is written: gradient w.r.t. the left SE(3)
 *     perturbation T_w2c <- SE3_exp([rho,theta]) * T_w2c at 0
 *     (gs_localization/pipelines/tools/pose_utils.py:105-122). */
int gsr_backward(int P, int D, int M, int R,
                 const float* background,
                 int width, int height,
                 const float* means3D,
                 const float* shs,
                 const float* colors_precomp,
                 const float* alphas,
                 const float* scales,
                 float scale_modifier,
                 const float* rotations,
                 const float* cov3D_precomp,
                 const float* viewmatrix,
                 const float* projmatrix,
                 const float* campos,
                 float tan_fovx, float tan_fovy,
                 const int* radii,
                 char* geom_buffer,
                 char* binning_buffer,
                 char* img_buffer,
                 const float* dL_dpix,
                 const float* dL_ddepths,
                 const float* dL_dalphas,
                 float* dL_dmean2D,
                 float* dL_dconic,
                 float* dL_dopacity,
                 float* dL_dcolor,
                 float* dL_dmean3D,
                 float* dL_dcov3D,
                 float* dL_dsh,
                 float* dL_dscale,
                 float* dL_drot,
                 int debug,
                 int pose_mode,
                 float* dL_dtau,
                 void* stream);

/* The same two entry points for callers whose foreign-function layer is slow per ARGUMENT (Python's ctypes converts and checks
 * every one of the 34 / 40 scalars and pointers: ~100 us per call, more than the launches): all arguments in one struct the caller
 * keeps and updates in place, ONE pointer across the boundary.  Field for field the parameters of gsr_forward_speculative /
 * gsr_backward above, same meaning, same order (rasterizer.h:31-89); nothing is retained after the call returns.
 * (GSR_ABI_VERSION 3.) */
typedef struct gsr_forward_args {
    gsr_spec_state* state;             /* NULL: plain gsr_forward */
    gsr_resize_fn geometry_buffer; void* geometry_ctx;
    gsr_resize_fn binning_buffer; void* binning_ctx;
    gsr_resize_fn image_buffer; void* image_ctx;
    int P, D, M;
    const float* background;
    int width, height;
    const float* means3D; const float* shs; const float* colors_precomp; const float* opacities;
    const float* scales; float scale_modifier; const float* rotations; const float* cov3D_precomp;
    const float* viewmatrix; const float* projmatrix; const float* cam_pos;
    float tan_fovx, tan_fovy;
    int prefiltered;
    float* out_color; float* out_depth; float* out_alpha;
    int* radii;
    int debug;
    int* n_touched;
    void* stream;
} gsr_forward_args;
int gsr_forward_packed(const gsr_forward_args* args);

typedef struct gsr_backward_args {
    int P, D, M, R;
    const float* background;
    int width, height;
    const float* means3D; const float* shs; const float* colors_precomp; const float* alphas;
    const float* scales; float scale_modifier; const float* rotations; const float* cov3D_precomp;
    const float* viewmatrix; const float* projmatrix; const float* campos;
    float tan_fovx, tan_fovy;
    const int* radii;
    char* geom_buffer; char* binning_buffer; char* img_buffer;
    const float* dL_dpix; const float* dL_ddepths; const float* dL_dalphas;
    float* dL_dmean2D; float* dL_dconic; float* dL_dopacity; float* dL_dcolor;
    float* dL_dmean3D; float* dL_dcov3D; float* dL_dsh; float* dL_dscale; float* dL_drot;
    int debug;
    int pose_mode;
    float* dL_dtau;
    void* stream;
} gsr_backward_args;
int gsr_backward_packed(const gsr_backward_args* args);

/* Replaces CudaRasterizer::Rasterizer::markVisible (rasterizer.h:24-29, rasterizer_impl.cu:141-153).
 * present [P] uint8 (0/1). */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* Fixed-size parts of the workspace, for callers that pre-allocate instead of resizing
 * (replaces `required<GeometryState>(P)` etc., rasterizer_impl.h:66-72). */
size_t gsr_geometry_bytes(int P);
size_t gsr_geometry_bytes_det(int P);      /* with the deterministic option (debug bit 2 of BOTH passes / GSR_REFINE_DETERMINISTIC): 64-bit accumulator records */
size_t gsr_image_bytes(int width, int height);
size_t gsr_binning_bytes(int num_rendered);

/* For callers that allocate the three workspaces of a forward up front instead of growing them on demand (a Python caller pays
 * three callbacks into the interpreter per forward otherwise).  gsr_fixed_buffer_resize is a gsr_resize_fn whose context is a
 * gsr_fixed_buffer: it returns `ptr` when the request fits `capacity` and NULL otherwise -- the forward then fails with
 * GSR_E_ALLOC and the caller repeats it with growing buffers; `requested` receives the size asked for either way.
 * gsr_binning_bytes_bins: an upper bound of what gsr_forward / gsr_forward_speculative ask their binning callback for while the
 * lists are binned into per-tile bins; 0 for image sizes whose complete lists go through count -> scan -> emit, whose size
 * depends on the scene (rasterizer_impl.cu:282: the reference sizes that buffer from a device read-back too). */
typedef struct gsr_fixed_buffer { void* ptr; size_t capacity; size_t requested; } gsr_fixed_buffer;
void* gsr_fixed_buffer_resize(void* ctx, size_t bytes);
size_t gsr_binning_bytes_bins(int P, int width, int height);

/* Statistics of the last forward on this buffer set (host ints, filled by a blocking copy):
 * stats[0] = visible Gaussians V (radii > 0), stats[1] = R (reference rule),
 * stats[2] = instances actually emitted after exact tile culling, stats[3] = R_eff
 * (sum over tiles of the max per-pixel n_contrib). Used by bench.py only. */
int gsr_forward_stats(int P, int width, int height, const int* radii, const char* geom_buffer,
                      const char* img_buffer, long long stats[4], void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pose-refinement epilogue, SURVEY.md section 8(f)-1 (the caller-side row next to the rasterizer): fused,
 * device-resident replacements for what the reference's Python loop does with ~130 tiny torch launches
 * per iteration (gs_localization/pipelines/7scenes_localize_full_dslam.py:66-91).
 * ------------------------------------------------------------------------------------------------ */

/* Tracking loss + gradient.  Replaces get_loss_tracking / _rgb / _rgbd
 * (gs_localization/pipelines/tools/descent_utils.py:85-123) and their autograd backward.
 *   image [3,H,W], depth [1,H,W], opacity [1,H,W]: rasterizer outputs; gt_image [3,H,W];
 *   gt_depth [H,W] (ignored when monocular != 0); grad_mask [H,W] uint8 0/1;
 *   exposure: 2 device floats (exposure_a, exposure_b).  depth_weight = 1 - config["Training"]["alpha"].
 *   Writes dL_dimage [3,H,W], dL_ddepth [1,H,W], dL_dalpha [1,H,W] (zeros: the opacity only feeds a mask)
 *   and out[0..2] = (loss, dL/dexposure_a, dL/dexposure_b)  (device, 4 floats). */
int gsr_tracking_loss(int width, int height, const float* image, const float* depth, const float* opacity,
                      const float* gt_image, const float* gt_depth, const uint8_t* grad_mask,
                      const float* exposure, float opacity_threshold, float depth_weight, int monocular,
                      float* dL_dimage, float* dL_ddepth, float* dL_dalpha, float* out, void* stream);

/* The per-frame gradient mask every localisation script refines under (GSR_ABI_VERSION 5).  Replaces
 * Camera.compute_grad_mask (gs_localization/pipelines/tools/camera_utils.py:164-193; image_gradient / image_gradient_mask,
 * tools/descent_utils.py:33-67, behind it) and the `viewpoint.grad_mask | create_mask(keypoints, k = 10)` step of the scripts
 * (7scenes_localize_full_dslam.py:126-149,355-360):
 *   gray = mean over the channels; Scharr gradients of the reflect-padded gray image, zeroed where a pixel of the 3 x 3
 *   neighbourhood has |gray| <= 0.01; intensity = sqrt(gv^2 + gh^2);
 *   grad_mask = intensity > median(intensity) * edge_threshold        (torch.median: the LOWER median, found exactly)
 *   grad_mask |= a box of 2 (box_k / 2) + 1 pixels around (int(x), int(y)) of every keypoint.
 *   image [3,H,W] float (device); keypoints: nullable device float [num_keypoints, 2] (x, y), every x, y > -1 (further outside
 *   the image the reference's numpy slice bounds turn negative and wrap around; a box that leaves the image is clipped); grad_mask [H,W] uint8 0/1 -- what
 *   gsr_tracking_loss / gsr_refine take; intensity_out: nullable device float [H,W] (the intensity image); median_out: nullable
 *   device float[2] = {median, threshold}.  workspace: resize callback for gsr_grad_mask_bytes(width, height) bytes.
 * Four launches (five with keypoints), no host synchronisation.  The mask is bit-identical to the reference's own Python run on
 * torch's CPU backend for the committed fixtures (tests/golden/grad_mask_vectors.npz; arithmetic: csrc/gsr_gradmask.h). */
size_t gsr_grad_mask_bytes(int width, int height);
int gsr_grad_mask(int width, int height, const float* image, float edge_threshold, const float* keypoints, int num_keypoints, int box_k,
                  uint8_t* grad_mask, float* intensity_out, float* median_out, gsr_resize_fn workspace, void* workspace_ctx, void* stream);
/* The config["Dataset"]["type"] == "replica" branch of the same function (camera_utils.py:174-188): a rows x cols (32 x 32 there)
 * grid of int(H / rows) x int(W / cols) blocks, each thresholded at its own lower median x edge_threshold.  The result is a FLOAT
 * image as in the reference, quirks included: 1 / 0 inside the grid (the reference's two in-place writes run in sequence, so the
 * ones are cleared again whenever 1 <= threshold), the raw intensity in the pixels right of / below the grid.  (No localisation
 * script can take this branch -- their next line ORs the mask with a boolean one, which torch refuses for a float tensor.) */
int gsr_grad_mask_replica(int width, int height, const float* image, float edge_threshold, int rows, int cols, float* grad_mask,
                          gsr_resize_fn workspace, void* workspace_ctx, void* stream);

/* Device-resident pose state: GSR_POSE_STATE_FLOATS floats, layout
 *   [0..8] R (row-major W2C rotation) [9..11] T [12..14] cam_rot_delta [15..17] cam_trans_delta
 *   [18] exposure_a [19] exposure_b [20..27] Adam exp_avg [28..35] Adam exp_avg_sq [36] Adam step
 *   [37] converged [38] last loss [39] |tau| [40] poison word of gsr_refine (uint32) [41] workgroup ticket of gsr_refine (uint32)
 *   [48..63] viewmatrix [64..79] projmatrix [80..82] campos [84..87] beta1^step, beta2^step as two doubles (Adam's bias corrections)
 *   [96..104] R, [105..107] T, [108] exposure_a, [109] exposure_b BEFORE the most recent pose step, i.e. the pose the last executed
 *   forward / backward ran with (GSR_ABI_VERSION 4; zeros until a step has run)
 * All zeros + R, T (+ exposure) is a valid initial state.
 * (viewmatrix/projmatrix/campos are what gsr_forward / gsr_backward take).
 * gsr_pose_init fills [48..82] from R, T and projmatrix_raw (16 floats, P^T row-major), replacing
 * Camera.world_view_transform / full_proj_transform / camera_center (tools/camera_utils.py:144-158). */
#define GSR_POSE_STATE_FLOATS 112
int gsr_pose_init(float* pose_state, const float* projmatrix_raw, void* stream);

/* One optimiser step + update_pose on the device.  Replaces torch.optim.Adam.step() over the four
 * parameter groups (7scenes_localize_full_dslam.py:33-64, torch defaults, single lr) and update_pose
 * (tools/pose_utils.py:105-122).  dL_dtau = gsr_backward's output; loss_out = gsr_tracking_loss's out. */
int gsr_pose_step(float* pose_state, const float* dL_dtau, const float* loss_out, const float* projmatrix_raw,
                  float lr, float converged_threshold, void* stream);

/* The whole refinement loop of gradient_decent() (7scenes_localize_full_dslam.py:29-93) in one call:
 * up to max_iters x { gsr_forward (pose package) -> gsr_tracking_loss -> gsr_backward -> gsr_pose_step },
 * stopping like the reference when update_pose reports convergence.  Every pointer is a device pointer
 * owned by the caller; workspaces are requested through the resize callbacks only when they must grow.
 * loss_out is scratch here (the last loss is pose_state[38]).  In the steady state no call inside blocks on the
 * device: every kernel of the loop checks the converged / poison words itself and the host reads each
 * iteration's 16-byte status one iteration late. */
typedef struct gsr_refine_args {
    int P, D, M;
    const float* means3D; const float* shs; const float* opacities; const float* scales; const float* rotations;
    float scale_modifier;
    int width, height; float tan_fovx, tan_fovy;
    const float* background; const float* projmatrix_raw;
    const float* gt_image; const float* gt_depth; const uint8_t* grad_mask;
    float opacity_threshold, depth_weight; int monocular;
    float* pose_state;
    float* out_color; float* out_depth; float* out_alpha; int* radii; int* n_touched;
    float* dL_dimage; float* dL_ddepth; float* dL_dalpha;
    float* dL_dmean2D; float* dL_dconic; float* dL_dopacity; float* dL_dcolor;
    float* dL_dmean3D; float* dL_dcov3D; float* dL_dsh; float* dL_dscale; float* dL_drot;   /* nullable: pose-only */
    /* On return the nine gradient tensors above hold the gradients of the LAST iteration whose pose step ran, alone (with the early
     * exit: the iteration whose update reported convergence, not the render at the final pose that follows it).  (Round 6, stated
     * exactly: the reference's loop only calls pose_optimizer.zero_grad(), 7scenes_localize_full_dslam.py:78, so the .grad of its map
     * tensors holds the SUM over all iterations; nobody reads that either.  What this call returns is what ONE loss.backward() of that
     * last iteration adds.)  Nobody can read them before the call returns, so the iterations in between only compute dL/dtau and
     * the rows are written once, from that iteration's records, when the loop ends. */
    float* dL_dtau; float* loss_out;
    gsr_resize_fn geometry_buffer; void* geometry_ctx;
    gsr_resize_fn binning_buffer; void* binning_ctx;
    gsr_resize_fn image_buffer; void* image_ctx;
    float lr, converged_threshold; int max_iters;
    int stop_on_converged;      /* 1 = reference behaviour; 0 = always run max_iters (benchmarks) */
    /* Speculative binning (exact: verified on the device, a failed speculation is redone with full lists):
     * from the 2nd iteration on, tile instances deeper than bound_margin_mul * z + bound_margin_add, z = the
     * depth the tile had to look at in the previous iteration, are not binned.  0 disables. */
    int speculative; float bound_margin_mul, bound_margin_add;     /* mul <= 0: adaptive, (1+m) z + m with m in [0.01, 0.05] */
    int* stats_out;             /* nullable host int[4]: [0] number of forwards that failed their verification, [1] last num_rendered (set [1] = -1 before the
                                 * call to skip that count when the last forward binned by tile: it costs a device->host copy),
                                 * [2] forwards that ran k_preprocess_lean (the conservative-bound preprocess), [3] how many of the
                                 * [0] failed forwards the HOST redid with complete lists (the others were retried on the device) */
    /* Nullable HOST int, in/out: warm start of the speculation for frame sequences.  0 on input = the image workspace
     * holds no depth bounds (the first iteration bins with the global sorts).  Pass the value the previous call on the
     * SAME image workspace (same size) left here to start speculating from that frame's bounds at once -- consecutive
     * frames of a sequence see almost the same depths.  As always the speculation is verified and redone if it
     * fails, so a stale or unrelated set of bounds costs time, never exactness.  (Opaque: the low byte names the bounds buffer,
     * the bits above it carry the adaptive margin the call ended with, so that the next frame does not start widening again.) */
    int* warm_state;
    /* Nullable HOST int, in/out: what the previous call left behind that this one may rely on.  0 on input = nothing.
     * Pass back the value the previous call wrote here ONLY if nothing below was touched in between -- the same geometry
     * workspace, the same gradient tensors (all dL_d* of this struct), the same Gaussians (P, scales, rotations,
     * scale_modifier):
     *   bit 0  the gradient tensors are exactly what that call left (zero except the rows its last backward wrote, which
     *          the workspace's dirty bits list): this call does not zero-fill them again (300 MB at 1 M Gaussians);
     *   bit 1  the workspace holds every Gaussian's 3D covariance: the first forward reads them instead of rebuilding them.
     * Set to 0 when the call fails. */
    int* carry_state;
    void* stream;
    /* ---- appended with GSR_ABI_VERSION 2 (new fields only ever go here, behind `stream`) ---- */
    unsigned flags;             /* GSR_REFINE_* bits below; 0 = product behaviour */
    int lean_min_P;             /* k_preprocess_lean is used from this many Gaussians on; 0 = default (200 000) */
    /* Start pose given as the caller's own device tensors (nullable, all four or none): the call begins by building pose_state
     * from them on the device -- zeros, R [9] row-major, T [3], exposure_a [1], exposure_b [1], then what gsr_pose_init does --
     * in ONE launch, instead of the caller assembling the state with five tiny copies (viewpoint.update_RT + Camera properties,
     * tools/camera_utils.py:124-158).  NULL: pose_state is taken as it is (gsr_pose_init already applied). */
    const float* init_R; const float* init_T; const float* init_exposure_a; const float* init_exposure_b;
    /* Nullable HOST float[GSR_POSE_STATE_FLOATS]: receives the final pose state (R, T, exposure, last loss ...) before the call
     * returns -- the call ends with a stream synchronisation anyway, so this costs one small copy and saves the caller a second
     * blocking read of the pose (update_RT of 7scenes_localize_full_dslam.py:84). */
    float* pose_state_host;
    /* ---- appended with GSR_ABI_VERSION 3 ---- */
    /* The precomputed-input modes of render() (B) (gs_localization/pipelines/tools/__init__.py:85-112: pipe.convert_SHs_python /
     * pipe.compute_cov3D_python), nullable.  colors_precomp [P,3]: used instead of `shs` (pass shs = NULL, M = 0, dL_dsh = NULL;
     * dL_dcolor is then the gradient of these colours).  cov3D_precomp [P,6]: used instead of `scales` / `rotations` (pass both NULL,
     * dL_dscale = dL_drot = NULL; dL_dcov3D is the gradient of these covariances, and dL/dtau's covariance term uses them). */
    const float* colors_precomp; const float* cov3D_precomp;
} gsr_refine_args;
/* diagnostic switches of gsr_refine (tests / tools; the results must not depend on any of them) */
#define GSR_REFINE_NO_LEAN     1u   /* k_preprocess + k_sh_color in every iteration instead of k_preprocess_lean */
#define GSR_REFINE_SH_SEPARATE 2u   /* k_preprocess_lean without the fused SH colour (k_sh_color behind it) */
#define GSR_REFINE_NO_BALANCE  4u   /* compositing kernels in XCD order instead of the work-balanced tile order */
#define GSR_REFINE_LOG_REDO    8u   /* one stderr line per redone (failed-speculation) forward */
/* The deterministic option (the determinism the survey's section 5 asks for "for tests"; the reference's backward adds with float
 * atomics, backward.cu:560-577, and is not reproducible either).  Every sum that crosses workgroups -- per-Gaussian gradient sums,
 * the terms of dL/dtau, the fused loss -- is accumulated in 64-bit fixed point: two runs on the same inputs give the same bits, and
 * so do the speculative, the plain and the GSR_REFINE_NO_LEAN loop among each other (tests/test_gpu_deterministic.py).  Results differ
 * from the default mode's by rounding only.  Per-(tile, Gaussian) gradient sums are kept as a coarse word (2^-8, +-2^55) plus a
 * remainder word (2^-56): no practical range limit; a per-Gaussian pose term must stay below 2^31 in magnitude, a per-tile loss sum
 * below 2^33 (beyond that the integer wraps).  Costs 2-4 % of a speculative iteration, 15 % with complete lists; the geometry buffer
 * then holds 192 B of accumulator records per Gaussian instead of 48 (gsr_geometry_bytes_det). */
#define GSR_REFINE_DETERMINISTIC 16u
/* Diagnostics: never split a heavy tile's list across workgroups.  (By default the speculative iterations of a loop cut the list of a
 * tile whose compositing took more than twice the mean into depth ranges that as many workgroups walk in parallel -- same results up
 * to the rounding of a transmittance carried as a product of per-range products; the deterministic option implies this flag.) */
#define GSR_REFINE_NO_SPLIT 32u
/* Diagnostics: do not widen the speculative depth bounds at depth discontinuities (a tile next to one that had to look much deeper takes
 * its neighbour's bound: fewer failed verifications along silhouettes).  Never changes a result either way. */
#define GSR_REFINE_NO_DILATE 64u
/* Diagnostics: write the gradient tensors of the Gaussians' own parameters in EVERY iteration (as a sequence of gsr_backward calls would)
 * instead of once, when the loop ends, from the last stepped iteration's records.  Nobody can read them in between; what the call
 * returns is the same either way -- bit for bit under the deterministic option (tests/test_gpu_deterministic.py) -- and the default is
 * 2 us (uniform cloud) to 26 us (structured scenes) per iteration cheaper. */
#define GSR_REFINE_GRADS_EVERY_ITERATION 128u
int gsr_refine(const gsr_refine_args* args, int* iters_done, int* converged);

/* Differential check of k_preprocess_lean's conservative test (tests only; replaces nothing in the reference -- it guards the
 * kernel that stands in for forward.cu:155-256 + rasterizer_impl.cu:70-111 in the loop's steady state).  Call right after a
 * gsr_refine that returned *warm_state != 0, with the same args (same workspaces): at the pose the state now holds, with the
 * depth bounds that call left behind (what its NEXT iteration would have binned with), every Gaussian goes through
 * (1) the conservative radius-bound test of k_preprocess_lean and (2) the exact geometry + exact footprint walk of k_preprocess
 * against the same per-tile bounds.  out[0] = Gaussians the conservative test settles ("nothing to do"), out[1] = candidates it
 * leaves, out[2] = Gaussians the exact walk bins into at least one tile, out[3] = VIOLATIONS: settled although the exact walk
 * bins them (must be 0), out[4] = index of the first violation or -1.  Blocking. */
int gsr_debug_lean_check(const gsr_refine_args* args, long long out[5]);
/* Tests only: what the LAST speculative iteration of the gsr_refine call that just returned on these workspaces did about heavy tiles
 * (see GSR_REFINE_NO_SPLIT).  out[0] = blocks of its compositing launches that had work, out[1] = tiles cut into more than one depth
 * range, out[2] = the largest number of ranges of a tile, out[3] = blocks the launches had room for (0: this image size is never
 * split).  Blocking. */
int gsr_debug_seg_stats(const gsr_refine_args* args, long long out[4]);
/* Byte offset of the per-Gaussian (mean, extent bound) quads inside a geometry workspace of gsr_geometry_bytes(P) bytes (tests only:
 * tests/test_gpu_lean.py overwrites the bounds to see the check above fail). */
size_t gsr_debug_lam_offset(int P);

/* Map on-disk rows -> device layout.  Replaces the per-property column gathering of
 * load_ply (gs_localization/pipelines/tools/gaussian_model.py:377-467; gaussian_splatting/scene/
 * gaussian_model.py:215-256) and, with activate = 1, the activation getters the reference re-evaluates in every
 * render() of a map that never changes during localisation (tools/gaussian_model.py:77-96).
 *   rows        device, P * row_floats f32: the vertex rows exactly as stored in point_cloud.ply
 *   cols        HOST array of 14 + n_rest ints: float index inside a row of
 *               x, y, z, f_dc_0..2, f_rest_0..n_rest-1, opacity, scale_0..2, rot_0..3   (in this order)
 *   n_rest      3 * ((sh_degree + 1)^2 - 1), at most 45
 *   outputs     means3D [P,3], shs [P,M,3] with M = 1 + n_rest / 3 (coefficient-major, RGB innermost: what the
 *               rasterizer takes), opacities [P], scales [P,3], rotations [P,4]
 *   activate    1: sigmoid(opacity), exp(scale), normalize(rot) applied once here; 0: raw parameters */
int gsr_map_from_ply_rows(int P, const float* rows, int row_floats, const int* cols, int n_rest, int activate,
                          float* means3D, float* shs, float* opacities, float* scales, float* rotations, void* stream);

/* Training-step loss epilogue: gaussian_splatting/train.py:92-108 with
 * utils/loss_utils.py:17-63 (l1_loss, ssim with the 11x11 Gaussian window) and their autograd backward:
 *   loss = (1 - lambda_dssim) * L1(image, gt) + lambda_dssim * (1 - SSIM(image, gt))
 *          + depth_weight * min(1 - pearson(-pseudo, depth), 1 - pearson(1 / (pseudo + 200), depth))
 * image, gt_image: [3, H, W]; depth, pseudo_depth: [H, W] or both NULL (no depth term; dL_ddepth may then be NULL).
 * Writes dL_dimage [3, H, W], dL_ddepth [H, W] and out[4] = {loss, Ll1, ssim, pseudo-depth loss} (device).
 * workspace: resize callback for 36 * W * H + 256 bytes of scratch. */
size_t gsr_training_loss_bytes(int width, int height);
int gsr_training_loss(int width, int height, const float* image, const float* gt_image, float lambda_dssim,
                      const float* depth, const float* pseudo_depth, float depth_weight, float* dL_dimage,
                      float* dL_ddepth, float* out, gsr_resize_fn workspace, void* workspace_ctx, void* stream);
/* Densification statistics of train.py:142-145 + GaussianModel.add_densification_stats (gaussian_model.py:405-407)
 * for the Gaussians with radii > 0: max_radii2D = max(max_radii2D, radii), xyz_gradient_accum += |dL_dmean2D.xy|,
 * denom += 1.  dL_dmean2D is the [P, 3] screen-space gradient the backward returns. */
int gsr_densification_stats(int P, const int* radii, const float* dL_dmean2D, float* max_radii2D,
                            float* xyz_gradient_accum, float* denom, void* stream);

/* distCUDA2 (gaussian_splatting/submodules/simple-knn/simple_knn.cu:183-220 behind
 * spatial.cu:15-26): mean_dist2[i] = mean squared distance from point i to its three nearest OTHER points.
 * points [P,3], mean_dist2 [P] on the device; workspace: resize callback for gsr_knn_bytes(P) bytes. */
size_t gsr_knn_bytes(int P);
int gsr_dist2_knn3(int P, const float* points, float* mean_dist2, gsr_resize_fn workspace, void* workspace_ctx, void* stream);

/* Optional per-kernel timing with HIP events recorded on the caller's stream around each kernel
 * (bench.py's roofline leg).  mask bit i enables kernel id i; 0 disables (the default, zero cost).
 * gsr_profile_collect waits for the recorded events, ADDS elapsed milliseconds / launch counts per
 * kernel id into ms[] / launches[] (arrays of gsr_profile_kernel_count() entries) and forgets them. */
int gsr_profile_enable(unsigned mask);
/* Bracket only one launch in `every` of each enabled kernel (default 1 = all): an event pair around a kernel keeps it
 * from overlapping its neighbours, which costs ~5 % with several frames in flight. */
int gsr_profile_sampling(unsigned every);
/* Diagnostic builds only (compiled with -DGSR_TIMING=1): copies out and clears 64 shader-clock phase totals
 * (slots 0-15 compositing forward, 16-31 compositing backward, 32-47 preprocess, 48-63 chain rule).  Returns -1 in product builds. */
int gsr_debug_timing(unsigned long long* out64);
/* Tests only: the launch order the stateless backward derives from the forward's per-tile work (heaviest tile first; image
 * workspace, tiles 1 ... 16 384): order[0 .. ntiles) must come back a permutation of the tile numbers whatever work[] holds.
 * Device pointers, uint32 each. */
int gsr_debug_tile_order(const unsigned* work, unsigned* order, int ntiles, void* stream);
int gsr_profile_collect(double* ms, long long* launches);
int gsr_profile_kernel_count(void);
const char* gsr_profile_kernel_name(int id);

const char* gsr_last_error(void);
int gsr_abi_version(void);
/* 1 if a gfx950 device is visible, 0 otherwise (never initialises more than device enumeration) */
int gsr_device_ok(void);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H_INCLUDED */
