// gsr_api.hip -- host orchestration + the C ABI declared in include/gsr.h.
// Replaces CudaRasterizer::Rasterizer::{forward,backward,markVisible}
// (reference: cuda_rasterizer/rasterizer_impl.cu:141-153,197-339,343-444).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>          // distCUDA2 only (Morton sort); the rasterizer itself uses no library kernels
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>
#include <atomic>
#include <thread>
#include <chrono>
#include <algorithm>

#include "gsr.h"
#include "gsr_kernels.h"
#include "gsr_gradmask.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, const char* a = "", const char* b = "")
{
    char buf[512];
    snprintf(buf, sizeof(buf), fmt, a, b);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(GSR_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define LAUNCHCHK(name)                                                                     \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) return fail(GSR_E_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
        if (debug) {                                                                        \
            e_ = hipStreamSynchronize(st);                                                  \
            if (e_ != hipSuccess) return fail(GSR_E_HIP, "kernel %s faulted: %s", name, hipGetErrorString(e_)); \
        }                                                                                   \
    } while (0)

// ---- optional per-kernel HIP-event timing (gsr_profile_*) ----
enum KernelId { K_PREPROCESS = 0, K_SH_COLOR, K_TILE_COUNT, K_TILE_SCAN, K_TILE_EMIT, K_RENDER_FWD, K_BWD_ZERO, K_RENDER_BWD, K_PREPROCESS_BWD,
                K_POSE_STEP, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"preprocess_fwd", "sh_color", "tile_count", "tile_scan", "tile_emit", "render_fwd",
                                           "bwd_zero", "render_bwd", "preprocess_bwd", "pose_step"};
struct Profiler {
    std::mutex mu;
    unsigned mask = 0;
    unsigned every = 1;                         // bracket one launch in `every` (gsr_profile_sampling)
    std::atomic<unsigned> seen[K_COUNT];
    std::vector<hipEvent_t> pool;
    struct Pending { int id; hipEvent_t a, b; };
    std::vector<Pending> pending;
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
Profiler g_prof;
struct ProfScope {
    int id; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool on;
    ProfScope(int id_, hipStream_t st_) : id(id_), st(st_), on((g_prof.mask >> id_) & 1u)
    {
        if (on && g_prof.every > 1) on = (g_prof.seen[id_].fetch_add(1u, std::memory_order_relaxed) % g_prof.every) == 0u;
        if (!on) return;
        std::lock_guard<std::mutex> l(g_prof.mu);
        a = g_prof.get(); b = g_prof.get();
        (void)hipEventRecord(a, st);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> l(g_prof.mu);
        g_prof.pending.push_back({id, a, b});
    }
};

// 256-byte aligned carving of an opaque workspace; with base == nullptr it only measures.
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(char* b) : base(b) {}
    template <class T>
    T* take(size_t n)
    {
        off = (off + 255) & ~(size_t)255;
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return r;
    }
    size_t size() const { return (off + 255) & ~(size_t)255; }
};

struct Geom {   // per-Gaussian state carried from forward to backward
    float* cov3D; uint8_t* clamped;
    uint32_t* tiles_touched; ushort4* rects; float* acc; double* tau_acc; uint8_t* dirty;
    gsr::SurvLists surv;      // work lists of the forward's survivors (k_preprocess -> k_sh_color, k_preprocess_bwd)
    float* rec;               // packed splat records (GSR_REC_*), P + 1
    float* lam;               // per-Gaussian float4 (mean, bound on sqrt(lambda_max(Sigma))) (PreArgs::lam), valid whenever cov3D holds every covariance
    uint8_t* aflag;           // 2 n bytes: "the compositing backward added to this Gaussian's record / to its colour sums" (K7 sets, K8 clears)
    // Native loop (carve_geom(..., loop = true)): what one iteration hands from its forward to its chain-rule kernel exists TWICE and the
    // kernel groups alternate (gsr::PreBwdArgs::role): set [0] is the arrays above, set [1] sits behind them.  select_set() points the
    // plain members at one set and o_* at the other.
    gsr::SurvLists surv2[2]; uint8_t* aflag2[2]; float* acc2[2]; float* rec2[2]; uint8_t* clamped2[2];
    gsr::SurvLists o_surv; uint8_t* o_aflag; float* o_acc; float* o_rec; uint8_t* o_clamped;
    float* prev_cam;          // 35 floats: the camera in front of the most recent pose step (gsr::PoseStepArgs::prev_cam)
    uint32_t* final_done;     // one word (gsr::PreBwdArgs::final_done)
    void select_set(int k)
    {
        surv = surv2[k]; aflag = aflag2[k]; acc = acc2[k]; rec = rec2[k]; clamped = clamped2[k];
        o_surv = surv2[k ^ 1]; o_aflag = aflag2[k ^ 1]; o_acc = acc2[k ^ 1]; o_rec = rec2[k ^ 1]; o_clamped = clamped2[k ^ 1];
    }
};
// `det`: the deterministic option's accumulator records are twelve PAIRS of 64-bit words (192 B) instead of twelve floats (48 B).
// The records sit at the END of the workspace, so that every other array has the same place in either mode and only the size asked
// for depends on it (every caller used to pay 192 B per Gaussian for an option only tests use).
size_t carve_geom(char* base, int P, Geom& g, bool det = false, bool loop = false)
{
    Carver c(base);
    const size_t n = P > 0 ? (size_t)P : 1;
    g.cov3D = c.take<float>(6 * n);
    g.clamped = c.take<uint8_t>(n);
    g.tiles_touched = c.take<uint32_t>(n);
    g.rects = c.take<ushort4>(n);
    g.dirty = c.take<uint8_t>(n);
    g.tau_acc = c.take<double>(16 * GSR_TAU_SLOTS);      // (second half: the deterministic option's world-frame sums 6 ... 11)
    g.rec = c.take<float>((n + 1) * GSR_REC_STRIDE);
    g.surv.cap = gsr::surv_cap(P);
    g.surv.n = c.take<uint32_t>((size_t)GSR_SURV_LISTS * GSR_SURV_CSTRIDE);
    g.surv.ids = c.take<uint32_t>((size_t)GSR_SURV_LISTS * g.surv.cap);
    g.lam = c.take<float>(4 * n);
    g.aflag = c.take<uint8_t>(2 * n);
    g.surv2[0] = g.surv; g.aflag2[0] = g.aflag; g.rec2[0] = g.rec; g.clamped2[0] = g.clamped;
    g.surv2[1] = g.surv; g.aflag2[1] = g.aflag; g.rec2[1] = g.rec; g.clamped2[1] = g.clamped;
    g.prev_cam = nullptr; g.final_done = nullptr;
    if (loop) {      // (in front of the records, whose size depends on `det`: a loop passes the same `det` to every carving of its workspace)
        g.surv2[1].n = c.take<uint32_t>((size_t)GSR_SURV_LISTS * GSR_SURV_CSTRIDE);
        g.surv2[1].ids = c.take<uint32_t>((size_t)GSR_SURV_LISTS * g.surv.cap);
        g.aflag2[1] = c.take<uint8_t>(2 * n);
        g.rec2[1] = c.take<float>((n + 1) * GSR_REC_STRIDE);
        g.clamped2[1] = c.take<uint8_t>(n);
        g.prev_cam = c.take<float>(64);
        g.final_done = reinterpret_cast<uint32_t*>(g.prev_cam ? g.prev_cam + 48 : nullptr);
    }
    g.acc = c.take<float>((det ? 4 : 1) * GSR_ACC_STRIDE * n);
    g.acc2[0] = g.acc; g.acc2[1] = g.acc;
    if (loop) g.acc2[1] = c.take<float>((det ? 4 : 1) * GSR_ACC_STRIDE * n);
    g.select_set(0);
    return c.size();
}

// The aggregated binning kernels keep one (count) or two (emit) LDS words per tile; images with more tiles than this fall
// back to per-instance atomics in HBM.  At most kTileBinMaxGroups workgroups share the Gaussians (tile_bin_gpb).
constexpr int kTileBinLdsTiles = 16 * 1024;
constexpr int kTileBinMaxGroups = 2048;
// (re-measured in round 4 with 4 608: the training step's forward 190 us slower -- bins of 4 096 overflow there --, S-3M-cam at 1024x576 1 380 against 1 674 it/s)
constexpr int kFullBinMaxTiles = 2048;      // k_preprocess_bin (complete lists in one kernel) up to this many tiles

struct Img {
    uint32_t* n_contrib; uint2* ranges;
    // exact bins: instances per tile, their exclusive prefix sum (ntiles + 1), how much of each segment has been handed out,
    // and the total (one word)
    uint32_t* tile_count; uint32_t* tile_start; uint32_t* tile_offset; uint32_t* tile_fill; uint32_t* total; uint16_t* block_counts; int copies;
    float* zb[2]; uint32_t* fail;                       // speculative depth bounds of the native loop, verification flag
    float* zbc[2]; int sbx, sby, nsb;                    // bounds per 4x4-tile superblock
    uint32_t* tile_cursor; size_t clear_words;           // bin-by-tile path: per-tile append cursors (GSR_CURSOR_STRIDE apart)
    float* loss_shards;                                   // native loop: GSR_LOSS_SHARDS x 16 floats (fused tracking loss)
    uint32_t* tile_work[2]; uint32_t* tile_order[2];      // native loop: per-tile work of the last forward [0] / backward [1] compositing -> their launch orders
    uint32_t* tile_hold;                                  // native loop: forwards a tile still goes without a depth bound after a failed verification
    // native loop, heavy tiles split across workgroups (gsr::SegCtl): launch list, per-block words and records, per-tile tickets
    uint32_t* seg_list[2]; uint32_t* seg_cnt; uint32_t* seg_pub; uint32_t* seg_ticket; uint32_t* seg_nosplit; uint32_t* seg_len; float* seg_rec; int seg_budget;
    float* zb_own[2]; uint32_t* nodilate;                  // (same block) every tile's own depth bound before dilate_bounds widened it; forwards a tile still goes without widening
};
// Blocks the compositing kernels are launched with when tiles may be split: every tile once + room for the heavy ones' extra segments
constexpr int kSegMaxTiles = 4096;
#ifndef GSR_SEG_BUDGET_MUL
#define GSR_SEG_BUDGET_MUL 3
#endif
int seg_budget_of(int ntiles) { return ntiles <= kSegMaxTiles ? GSR_SEG_BUDGET_MUL * ntiles : 0; }      // (every block without work costs the launch a little: S-1M-640, no tile split, 3 x: +1.3 us per kernel)
// (seg: with the split-tile arrays -- at the END, so that everything else has the same place either way; only gsr_refine asks for them)
size_t carve_img(char* base, int W, int H, Img& im, bool seg = false)
{
    Carver c(base);
    const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
    im.n_contrib = c.take<uint32_t>((size_t)W * H);
    im.ranges = c.take<uint2>((size_t)gx * gy);
    // (counter copies and the per-workgroup count rows only exist on the LDS-aggregated path: up to kTileBinLdsTiles tiles)
    const size_t nt = (size_t)gx * gy;
    im.copies = (nt <= (size_t)kTileBinLdsTiles) ? GSR_TBIN_COPIES : 1;
    im.tile_count = c.take<uint32_t>(nt * im.copies);
    im.tile_start = c.take<uint32_t>(nt * im.copies);
    im.tile_offset = c.take<uint32_t>(nt + 1);
    im.tile_fill = c.take<uint32_t>(nt * im.copies);
    im.total = c.take<uint32_t>(1);
    im.block_counts = c.take<uint16_t>(nt <= (size_t)kTileBinLdsTiles ? nt * kTileBinMaxGroups : 1);
    im.zb[0] = c.take<float>((size_t)gx * gy);
    im.zb[1] = c.take<float>((size_t)gx * gy);
    im.sbx = (gx + 3) / 4;
    im.sby = (gy + 3) / 4;
    im.nsb = im.sbx * im.sby;
    im.zbc[0] = c.take<float>((size_t)im.nsb);
    im.zbc[1] = c.take<float>((size_t)im.nsb);
    // fail | tile_cursor are contiguous: one memset clears them
    im.clear_words = 16 + (size_t)gx * gy * GSR_CURSOR_STRIDE;
    im.fail = c.take<uint32_t>(im.clear_words);
    im.tile_cursor = base ? im.fail + 16 : nullptr;
    im.loss_shards = c.take<float>(GSR_LOSS_SHARDS * 16);
    im.tile_work[0] = c.take<uint32_t>(nt); im.tile_work[1] = c.take<uint32_t>(nt);      // (contiguous: one memset)
    im.tile_order[0] = c.take<uint32_t>(nt); im.tile_order[1] = c.take<uint32_t>(nt);
    im.tile_hold = c.take<uint32_t>(nt);
    im.seg_budget = seg ? seg_budget_of((int)nt) : 0;
    im.seg_list[0] = im.seg_list[1] = im.seg_cnt = im.seg_pub = im.seg_ticket = im.seg_nosplit = im.seg_len = im.nodilate = nullptr; im.seg_rec = nullptr; im.zb_own[0] = im.zb_own[1] = nullptr;
    if (im.seg_budget > 0) {
        im.seg_list[0] = c.take<uint32_t>((size_t)im.seg_budget);      // (a launch walks one list while its extra workgroup builds the other)
        im.seg_list[1] = c.take<uint32_t>((size_t)im.seg_budget);
        im.seg_cnt = c.take<uint32_t>((size_t)im.seg_budget);          // cnt | pub | ticket | nosplit are contiguous: one clear
        im.seg_pub = c.take<uint32_t>((size_t)im.seg_budget);
        im.seg_ticket = c.take<uint32_t>(2 * nt);
        im.seg_nosplit = c.take<uint32_t>(nt);
        im.seg_len = c.take<uint32_t>(nt);
        im.nodilate = c.take<uint32_t>(nt);
        im.zb_own[0] = c.take<float>(nt);
        im.zb_own[1] = c.take<float>(nt);
        im.seg_rec = c.take<float>((size_t)im.seg_budget * GSR_SEG_REC_Q * GSR_BLOCK);
    }
    return c.size();
}

struct Bin {      // exact bins, per tile instance: the ordered index lists (what the backward reads) + the unordered keys
    uint32_t* vals; unsigned long long* keys;
};
size_t carve_bin(char* base, int R, Bin& b)
{
    Carver c(base);
    const size_t n = R > 0 ? (size_t)R : 1;
    b.vals = c.take<uint32_t>(n);
    b.keys = c.take<unsigned long long>(n);
    return c.size();
}

// bin-by-tile path: sorted index lists (same place as Bin::vals, which is all the backward reads) + the bins
struct BinLocal { uint32_t* vals; unsigned long long* bins; };
// Entries per bin.  What a saturating tile needs behind its depth bound is a few hundred; a tile WITHOUT a bound (one that does not
// saturate) gets its complete list, ~R / tiles: room for four times the in-LDS sort's limit while that is affordable
// (96 KB per tile: 118 MB at 640x480).
int bin_capacity(int ntiles) { return ntiles <= 4096 ? 4 * GSR_LSORT_CAP : GSR_LSORT_CAP; }
// Entries per bin for COMPLETE lists (k_preprocess_bin): eight times the mean number of Gaussians per tile, a power of two,
// at least 8 192.  (S-1M-640: 8 192 for a mean list of 2 800 after exact culling; S-3M-cam: 16 384 for 7 200.)  A tile that
// needs more makes the forward fall back to the exact count -> scan -> emit path; 12 bytes per entry.
int full_bin_capacity(int P, int ntiles)
{
    long long want = 8ll * P / (ntiles > 0 ? ntiles : 1);
    int cap = 4 * GSR_LSORT_CAP;
    while (cap < want && cap < (1 << 20)) cap <<= 1;
    return cap;
}
size_t carve_bin_local(char* base, int ntiles, int cap, BinLocal& b)
{
    Carver c(base);
    b.vals = c.take<uint32_t>((size_t)ntiles * (cap + GSR_BIN_PAD));
    b.bins = c.take<unsigned long long>((size_t)ntiles * (cap + GSR_BIN_PAD));
    return c.size();
}

// What the callers inside this library add to a forward / backward pass.  The public gsr_forward / gsr_backward run with a
// default-constructed context; gsr_forward_speculative fills `spec`; gsr_refine fills everything, per iteration.  (Passed
// explicitly: the backward may run on another host thread than the forward -- PyTorch's autograd worker -- and a re-entrant
// caller must never inherit a previous call's settings.)
// Speculative per-tile depth bounds: mode 0 = off, 1 = bin with the bounds the previous forward recorded and record new
// ones, 2 = bin everything but record bounds.  parity picks the buffer that is written.
struct SpecCtx { int mode = 0; int parity = 0; float mul = 1.05f, add = 0.05f; char* state = nullptr; };
struct PassCtx {
    bool split_ok_now = true;      // gsr_refine: may the launch list this group builds split heavy tiles?  (not with many calls in flight, see enqueue)
    bool native_loop = false;      // gsr_refine: gradient tensors and accumulators are maintained by the kernels, not re-zeroed here
    SpecCtx spec;
    gsr::LoopGuard guard = {nullptr, nullptr, 0u};      // device-side poison / converged words and this group's tag (see LoopGuard)
    gsr::FusedLoss floss = {};     // tracking loss evaluated in the compositing kernel's epilogue (out == nullptr: not fused)
    int cov_cache = 0;             // 1 = this forward stores every Gaussian's 3D covariance in the geometry buffer, 2 = reads them back
    bool lean = false;             // this forward's radii are not an output (see k_preprocess)
    uint32_t* ticket = nullptr;    // backward: the chain-rule kernel's last workgroup runs the pose step `fold` (see PreBwdArgs)
    gsr::PoseStepArgs fold = {};
    gsr::GradRows rows = {};       // native loop: the gradient tensors the kernels keep consistent through the dirty bits
    bool balance = false;          // native loop: compositing kernels launched in work-balanced tile order (tile_order_from_work)
    unsigned flags = 0;            // GSR_REFINE_* diagnostics of the caller
    int lean_min_P = 200000;       // k_preprocess_lean from this many Gaussians on
    int* n_lean = nullptr;         // counts the forwards that ran k_preprocess_lean (gsr_refine_args.stats_out[2])
    bool sh_eager = false;         // diagnostics (debug bit 1 of gsr_forward): k_sh_color for every visible Gaussian instead of lazy colours
    bool exact_bins = false;       // complete lists through count -> scan -> emit (after a bin of k_preprocess_bin overflowed; diagnostics)
    bool* used_full_bins = nullptr;   // out: this forward binned its complete lists into fixed-capacity bins (k_preprocess_bin)
    bool det = false;              // deterministic option (GSR_REFINE_DETERMINISTIC / debug bit 2 of the backward): integer sums across workgroups
    gsr::PreBwdArgs* pb_out = nullptr;   // native loop: the chain-rule kernel's arguments of this group, kept for the final pass (gsr_refine)
    int set = 0;                   // native loop: which of the two sets of work lists / flags / records / splat records this group uses (Geom::select_set)
    bool seg = false;              // native loop: the image workspace has the split-tile arrays (gsr::SegCtl) and speculative forwards may use them
    bool* seg_used = nullptr;      // out (forward) / in (backward): this group's compositing kernels run the split-tile launch list
    bool seg_ready = false;        // the previous group's backward built a launch list for this one (list[spec.parity ^ 1])
    uint32_t hold_after = 2u;      // native loop: failed verifications of a tile before it is left with its complete list for a while (k_render_fwd, tile_hold)
    // Blocks of a launch list.  A list is built one group ahead (by the previous group's backward) for a launch whose size the host
    // fixed when it enqueued THAT group: seg_grid = blocks this group's compositing kernels are launched with (what its list was built
    // for), seg_grid_next = what the list this group builds may use.  Both 0: the workspace's full budget.
    int seg_grid = 0, seg_grid_next = 0;
    uint32_t* seg_host_total = nullptr;      // pinned: blocks the most recently built list holds (the host's hint for later launches)
};
// gsr_forward_speculative: bounds, flags, cursors and the unsorted bins live in the caller's persistent state buffer
// instead of the per-call image / binning buffers (which then only hold what the backward reads)
size_t carve_spec(char* base, int W, int H, Img& im, unsigned long long** bins)
{
    Carver c(base);
    const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
    im.zb[0] = c.take<float>((size_t)gx * gy);
    im.zb[1] = c.take<float>((size_t)gx * gy);
    im.sbx = (gx + 3) / 4;
    im.sby = (gy + 3) / 4;
    im.nsb = im.sbx * im.sby;
    im.zbc[0] = c.take<float>((size_t)im.nsb);
    im.zbc[1] = c.take<float>((size_t)im.nsb);
    im.clear_words = 16 + (size_t)gx * gy * GSR_CURSOR_STRIDE;
    im.fail = c.take<uint32_t>(im.clear_words);
    im.tile_cursor = base ? im.fail + 16 : nullptr;
    unsigned long long* b = c.take<unsigned long long>((size_t)gx * gy * (GSR_LSORT_CAP + GSR_BIN_PAD));
    if (bins) *bins = b;
    return c.size();
}

// One side stream + a fork/join event pair per device: independent work (SH colours, zero fills) runs next to
// the latency-bound sort chain / the VALU-bound backward compositing instead of in front of them.
// Re-recording an event after a wait on it has been enqueued is safe (the wait binds to the record that
// preceded it), so one pair per (host thread, device) is enough.
struct Side {
    hipStream_t st = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
// Side streams are pooled per device (host threads come and go -- one per frame in flight -- and creating a
// stream costs milliseconds).  A Side is held for the duration of one API call; the next holder's work simply
// queues behind whatever the previous one left on the stream.
struct SidePool { std::mutex mu; std::vector<Side*> free_list[64]; };
SidePool g_sides;
Side* side_acquire(int dev)
{
    static const bool disabled = getenv("GSR_NO_SIDE_STREAM") != nullptr;     // diagnostics
    if (disabled || dev < 0 || dev >= 64) return nullptr;
    {
        std::lock_guard<std::mutex> l(g_sides.mu);
        auto& fl = g_sides.free_list[dev];
        if (!fl.empty()) { Side* sd = fl.back(); fl.pop_back(); return sd; }
    }
    Side* sd = new Side();
    if (hipStreamCreateWithFlags(&sd->st, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&sd->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sd->join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        delete sd;           // (whatever was created is leaked: this only happens when the runtime is out of resources)
        return nullptr;
    }
    return sd;
}
struct SideLease {          // RAII: returns the Side to its device's pool
    Side* sd; int dev;
    SideLease(bool want, int dev_) : sd(want ? side_acquire(dev_) : nullptr), dev(dev_) {}
    ~SideLease()
    {
        if (!sd) return;
        std::lock_guard<std::mutex> l(g_sides.mu);
        g_sides.free_list[dev].push_back(sd);
    }
    SideLease(const SideLease&) = delete;
    SideLease& operator=(const SideLease&) = delete;
};

// Pinned status slots + events of one gsr_refine call, pooled for the same reason.
struct LoopCtx { float* h_status = nullptr; };      // pinned, device-visible host memory: 2 slots x 8 floats + a copy of the pose state
struct LoopCtxPool { std::mutex mu; std::vector<LoopCtx*> free_list; };
LoopCtxPool g_loop_ctx;
LoopCtx* loop_ctx_acquire()
{
    {
        std::lock_guard<std::mutex> l(g_loop_ctx.mu);
        if (!g_loop_ctx.free_list.empty()) { LoopCtx* c = g_loop_ctx.free_list.back(); g_loop_ctx.free_list.pop_back(); return c; }
    }
    LoopCtx* c = new LoopCtx();
    if (hipHostMalloc((void**)&c->h_status, (2 * 8 + GSR_POSE_STATE_FLOATS) * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        delete c;
        return nullptr;
    }
    return c;
}
void loop_ctx_release(LoopCtx* c)
{
    std::lock_guard<std::mutex> l(g_loop_ctx.mu);
    g_loop_ctx.free_list.push_back(c);
}

int select_device_of(const void* p, int* dev_out = nullptr)
{
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(GSR_E_NODEVICE, "means3D is not a device pointer (%s)", hipGetErrorString(e)); }
    if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged)
        return fail(GSR_E_NODEVICE, "means3D must live in device memory%s", "");
    e = hipSetDevice(attr.device);
    if (e != hipSuccess) return fail(GSR_E_HIP, "hipSetDevice failed: %s", hipGetErrorString(e));
    if (dev_out) *dev_out = attr.device;
    return GSR_OK;
}

// Zero fills of a stateless forward / backward: the small ones (counters, flags, cursors -- a few KB to a few hundred KB each) leave as
// ONE kernel launch instead of one hipMemsetAsync each (every enqueue costs the host 5-8 us; the reference-style Python loop is
// host-bound); large ranges keep the runtime's fill kernel.
struct ZeroList {
    gsr::ClearRanges cr = {};
    int k = 0;
    size_t small_words = 0;
    int add(void* ptr, size_t bytes, hipStream_t st)
    {
        if (!ptr || bytes == 0) return GSR_OK;
        if (bytes > (256u << 10) || (bytes & 3u) != 0 || k >= 10) { HIPCHK(hipMemsetAsync(ptr, 0, bytes, st)); return GSR_OK; }
        cr.p[k] = static_cast<uint32_t*>(ptr); cr.n[k] = (uint32_t)(bytes / 4); k++;
        small_words += bytes / 4;
        return GSR_OK;
    }
    int flush(hipStream_t st)
    {
        if (k == 0) return GSR_OK;
        const int blocks = (int)std::min<size_t>(256, std::max<size_t>(1, small_words / 2048));
        hipLaunchKernelGGL(gsr::k_refine_init, dim3(blocks), dim3(GSR_BLOCK), 0, st, cr, gsr::PoseLoadArgs{});
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) return fail(GSR_E_HIP, "launch of %s failed: %s", "k_refine_init", hipGetErrorString(e_));
        return GSR_OK;
    }
};

}  // namespace

extern "C" {

const char* gsr_last_error(void) { return g_err.c_str(); }

// diagnostic builds only (GSR_TIMING): copies the 32 phase counters out and clears them; -1 in product builds
int gsr_debug_timing(unsigned long long* out48)          // (64 entries since the chain-rule kernel has its slots: 48-63)
{
#if GSR_TIMING
    static std::vector<unsigned long long> h((size_t)4 * GSR_TIM_WAVES * 12);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(gsr::g_tim), h.size() * 8) != hipSuccess) return -2;
    for (int k = 0; k < 4; k++)
        for (int q = 0; q < 12; q++) {
            unsigned long long sum = 0;
            for (size_t w = 0; w < GSR_TIM_WAVES; w++) sum += h[((size_t)k * GSR_TIM_WAVES + w) * 12 + q];
            out48[k * 16 + q] = sum;
        }
    // slots 12..15 of each kernel: the largest accumulated wave lifetime (slot 9) of any row, the number of rows used, and the
    // 50th / 99th percentile of the rows' lifetimes -- is the kernel's duration its mean wave or its slowest one?
    for (int k = 0; k < 4; k++) {
        std::vector<unsigned long long> life;
        for (size_t w = 0; w < GSR_TIM_WAVES; w++) {
            const unsigned long long v = h[((size_t)k * GSR_TIM_WAVES + w) * 12 + 9];
            if (v) life.push_back(v);
        }
        std::sort(life.begin(), life.end());
        out48[k * 16 + 12] = life.empty() ? 0 : life.back();
        out48[k * 16 + 13] = life.size();
        out48[k * 16 + 14] = life.empty() ? 0 : life[life.size() / 2];
        out48[k * 16 + 15] = life.empty() ? 0 : life[life.size() * 99 / 100];
    }
    if (const char* dump = getenv("GSR_TIM_DUMP")) {          // raw per-wave rows: "<kernel> <row> <12 slots> <12 slots of the last launch> <first, last instant of the row's last launch, 100 MHz>" (diagnostics)
        static std::vector<unsigned long long> sp((size_t)4 * GSR_TIM_WAVES * 2);
        if (hipMemcpyFromSymbol(sp.data(), HIP_SYMBOL(gsr::g_tim_span), sp.size() * 8) != hipSuccess) return -2;
        static std::vector<unsigned long long> la((size_t)4 * GSR_TIM_WAVES * 12);
        if (hipMemcpyFromSymbol(la.data(), HIP_SYMBOL(gsr::g_tim_last), la.size() * 8) != hipSuccess) return -2;
        if (FILE* f = fopen(dump, "w")) {
            for (int k = 0; k < 4; k++)
                for (size_t w = 0; w < GSR_TIM_WAVES; w++) {
                    const unsigned long long* r = &h[((size_t)k * GSR_TIM_WAVES + w) * 12];
                    if (r[9] == 0ull) continue;
                    fprintf(f, "%d %zu", k, w);
                    for (int q = 0; q < 12; q++) fprintf(f, " %llu", r[q]);
                    for (int q = 0; q < 12; q++) fprintf(f, " %llu", la[((size_t)k * GSR_TIM_WAVES + w) * 12 + q]);
                    fprintf(f, " %llu %llu\n", sp[((size_t)k * GSR_TIM_WAVES + w) * 2], sp[((size_t)k * GSR_TIM_WAVES + w) * 2 + 1]);
                }
            fclose(f);
        }
    }
    std::fill(h.begin(), h.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(gsr::g_tim), h.data(), h.size() * 8) != hipSuccess) return -2;
    return 0;
#else
    (void)out48;
    return -1;
#endif
}

int gsr_profile_enable(unsigned mask)
{
    std::lock_guard<std::mutex> l(g_prof.mu);
    g_prof.mask = mask;
    return 0;
}
int gsr_profile_sampling(unsigned every)
{
    std::lock_guard<std::mutex> l(g_prof.mu);
    g_prof.every = every ? every : 1u;
    return 0;
}
int gsr_profile_kernel_count(void) { return K_COUNT; }
const char* gsr_profile_kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : ""; }
int gsr_profile_collect(double* ms, long long* launches)
{
    std::vector<Profiler::Pending> todo;
    {
        std::lock_guard<std::mutex> l(g_prof.mu);
        todo.swap(g_prof.pending);
    }
    for (auto& p : todo) {
        HIPCHK(hipEventSynchronize(p.b));
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, p.a, p.b));
        if (ms) ms[p.id] += t;
        if (launches) launches[p.id] += 1;
    }
    std::lock_guard<std::mutex> l(g_prof.mu);
    for (auto& p : todo) { g_prof.pool.push_back(p.a); g_prof.pool.push_back(p.b); }
    return 0;
}
int gsr_abi_version(void) { return GSR_ABI_VERSION; }

int gsr_device_ok(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

size_t gsr_geometry_bytes(int P) { Geom g; return carve_geom(nullptr, P, g, false); }
size_t gsr_geometry_bytes_det(int P) { Geom g; return carve_geom(nullptr, P, g, true); }
size_t gsr_image_bytes(int width, int height) { Img im; return carve_img(nullptr, width, height, im); }
size_t gsr_binning_bytes(int num_rendered) { Bin b; return carve_bin(nullptr, num_rendered, b); }
size_t gsr_binning_bytes_bins(int P, int width, int height)
{
    if (P <= 0 || width <= 0 || height <= 0) return 0;
    const int ntiles = ((width + GSR_TILE - 1) / GSR_TILE) * ((height + GSR_TILE - 1) / GSR_TILE);
    if (ntiles > kFullBinMaxTiles) return 0;
    BinLocal bl;
    return carve_bin_local(nullptr, ntiles, std::max(full_bin_capacity(P, ntiles), (int)GSR_LSORT_CAP), bl);
}
void* gsr_fixed_buffer_resize(void* ctx, size_t bytes)
{
    gsr_fixed_buffer* b = static_cast<gsr_fixed_buffer*>(ctx);
    if (!b) return nullptr;
    b->requested = bytes;
    return (bytes <= b->capacity) ? b->ptr : nullptr;
}

namespace {

// Gaussians per workgroup of k_tile_count / k_tile_emit: 2 per lane (2048) aggregate enough in LDS (one add to a tile's counter
// in HBM per workgroup instead of one per instance) and leave every CU a workgroup or two; more per lane only when the
// workgroup count would exceed kTileBinMaxGroups (the rows of per-workgroup counts are sized for that many).
int tile_bin_gpb(int P)
{
    int k = (P + kTileBinMaxGroups * GSR_TBIN_THREADS - 1) / (kTileBinMaxGroups * GSR_TBIN_THREADS);
    if (k < 2) k = 2;
    return k * GSR_TBIN_THREADS;
}
// 16 K tiles = 128 KB of the CU's 160 KB in k_tile_emit: beyond HIP's default 64 KB of dynamic LDS, so the limit is raised
// once per device.
int allow_large_lds(int dev)
{
    static std::mutex mu;
    static bool done[64] = {};
    if (dev < 0 || dev >= 64) return GSR_OK;
    std::lock_guard<std::mutex> l(mu);
    if (done[dev]) return GSR_OK;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(gsr::k_tile_count<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(gsr::k_tile_count<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(gsr::k_tile_emit<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(gsr::k_tile_emit<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(gsr::k_preprocess_bin), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    done[dev] = true;
    return GSR_OK;
}

// The deterministic option widens K7's accumulator records from 48 to 192 bytes per Gaussian, and it is the FORWARD that sizes the
// geometry workspace.  A backward that asks for the option on a workspace whose forward ran without it would write 144 P bytes past
// the end.  The stateless forwards therefore note (workspace address -> sized for the option?) in a small process-wide
// table and the backward refuses the mismatch; a workspace the table no longer knows (256 forwards ago) is taken at the caller's word.
struct DetNote { const void* geom; bool det; };
std::mutex g_det_mu;
DetNote g_det_notes[256];
unsigned g_det_next = 0;
void note_geometry(const void* geom, bool det)
{
    if (!geom) return;
    std::lock_guard<std::mutex> lk(g_det_mu);
    for (DetNote& n : g_det_notes)
        if (n.geom == geom) { n.det = det; return; }
    g_det_notes[g_det_next++ & 255u] = DetNote{geom, det};
}
int geometry_sized_for_det(const void* geom)          // 1 / 0, -1 = unknown
{
    std::lock_guard<std::mutex> lk(g_det_mu);
    for (const DetNote& n : g_det_notes)
        if (n.geom == geom) return n.det ? 1 : 0;
    return -1;
}

#define GSR_FWD_PARAMS gsr_resize_fn geometry_buffer, void* geometry_ctx, gsr_resize_fn binning_buffer, void* binning_ctx,                 \
                       gsr_resize_fn image_buffer, void* image_ctx, int P, int D, int M, const float* background, int width, int height,    \
                       const float* means3D, const float* shs, const float* colors_precomp, const float* opacities, const float* scales,    \
                       float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,                   \
                       const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,    \
                       float* out_depth, float* out_alpha, int* radii, int debug, int* n_touched, void* stream
#define GSR_FWD_PASS geometry_buffer, geometry_ctx, binning_buffer, binning_ctx, image_buffer, image_ctx, P, D, M, background, width, \
                     height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, \
                     projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth, out_alpha, radii, debug, n_touched, stream

// Work-balanced launch order of the compositing kernels outside the native loop (tile_order_block in k_backward_prologue): pays when there are more tiles
// than resident workgroups (256 CUs x 5), i.e. when the order in which tiles START decides how long the last one runs.
#ifndef GSR_STATELESS_BALANCE_MIN_TILES
#define GSR_STATELESS_BALANCE_MIN_TILES 1536      // (0: never)
#endif
static inline bool stateless_balanced(const PassCtx& cx, int ntiles)
{
    return GSR_STATELESS_BALANCE_MIN_TILES > 0 && !cx.native_loop && ntiles >= GSR_STATELESS_BALANCE_MIN_TILES && ntiles <= GSR_STATELESS_BALANCE_MAX_TILES;
}

int forward_impl(const PassCtx& cx, GSR_FWD_PARAMS)
{
    (void)prefiltered;   // the reference only uses it to trap on a culled point (auxiliary.h:152-156)
    using namespace gsr;
    hipStream_t st = (hipStream_t)stream;
    if (P < 0 || width <= 0 || height <= 0) return fail(GSR_E_INVALID, "P >= 0 and positive image size required%s", "");
    if (!geometry_buffer || !binning_buffer || !image_buffer) return fail(GSR_E_INVALID, "resize callbacks must not be NULL%s", "");
    if (!out_color || !out_depth || !out_alpha) return fail(GSR_E_INVALID, "output images must not be NULL%s", "");
    const size_t N = (size_t)width * height;
    if (P == 0) {
        // rasterize_points.cu:81 skips the rasterizer entirely: outputs stay at their zero fill
        int rc = select_device_of(out_color);
        if (rc != GSR_OK) return rc;
        HIPCHK(hipMemsetAsync(out_color, 0, 3 * N * sizeof(float), st));
        HIPCHK(hipMemsetAsync(out_depth, 0, N * sizeof(float), st));
        HIPCHK(hipMemsetAsync(out_alpha, 0, N * sizeof(float), st));
        return 0;
    }
    if (!means3D || !opacities || !background || !viewmatrix || !projmatrix || !cam_pos || !radii)
        return fail(GSR_E_INVALID, "means3D/opacities/background/viewmatrix/projmatrix/cam_pos/radii must not be NULL%s", "");
    if ((shs == nullptr) == (colors_precomp == nullptr))
        return fail(GSR_E_INVALID, "provide exactly one of shs / colors_precomp%s", "");
    if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr))
        return fail(GSR_E_INVALID, "provide exactly one of (scales, rotations) / cov3D_precomp%s", "");
    if (shs && (M <= 0 || (D + 1) * (D + 1) > M || D < 0 || D > 3))
        return fail(GSR_E_INVALID, "SH degree / coefficient count mismatch%s", "");
    int dev = 0;
    int rc = select_device_of(means3D, &dev);
    if (rc != GSR_OK) return rc;

    const int gx = (width + GSR_TILE - 1) / GSR_TILE, gy = (height + GSR_TILE - 1) / GSR_TILE;
    if (gx > 65535 || gy > 65535) return fail(GSR_E_INVALID, "image too large%s", "");
    const int ntiles = gx * gy;
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    const SpecCtx& sp = cx.spec;

    Geom g;
    const size_t gbytes = carve_geom(nullptr, P, g, cx.det, cx.native_loop);
    char* gptr = (char*)geometry_buffer(geometry_ctx, gbytes);
    if (!gptr) return fail(GSR_E_ALLOC, "geometry buffer callback returned NULL%s", "");
    carve_geom(gptr, P, g, cx.det, cx.native_loop);
    if (cx.native_loop) g.select_set(cx.set);
    if (!cx.native_loop) note_geometry(gptr, cx.det);      // (gsr_backward checks it: see geometry_sized_for_det)
    Img im;
    const size_t ibytes = carve_img(nullptr, width, height, im, cx.seg);
    char* iptr = (char*)image_buffer(image_ctx, ibytes);
    if (!iptr) return fail(GSR_E_ALLOC, "image buffer callback returned NULL%s", "");
    carve_img(iptr, width, height, im, cx.seg);
    unsigned long long* state_bins = nullptr;
    if (sp.state) carve_spec(sp.state, width, height, im, &state_bins);

    const int pblocks = (P + GSR_BLOCK - 1) / GSR_BLOCK;
    PreArgs pa;
    pa.P = P; pa.D = D; pa.M = M; pa.W = width; pa.H = height; pa.gx = gx; pa.gy = gy;
    pa.means = means3D; pa.scales = scales; pa.mod = scale_modifier; pa.rots = rotations; pa.opac = opacities;
    pa.shs = shs; pa.cov3D_pre = cov3D_precomp; pa.colors_pre = colors_precomp;
    pa.cov_all = (cx.cov_cache == 1 && cov3D_precomp == nullptr) ? 1 : 0;
    if (cx.cov_cache == 2 && cov3D_precomp == nullptr) pa.cov3D_pre = g.cov3D;
    pa.view = viewmatrix; pa.proj = projmatrix; pa.campos = cam_pos;
    pa.tanx = tan_fovx; pa.tany = tan_fovy; pa.fx = focal_x; pa.fy = focal_y;
    pa.radii = radii; pa.cov3D = g.cov3D; pa.lam = g.lam;
    pa.clamped = g.clamped; pa.tiles_touched = g.tiles_touched; pa.rects = g.rects;
    pa.guard = cx.guard;
    pa.n_touched = n_touched;
    pa.surv = g.surv;
    pa.rec = g.rec;
    const bool balanced = cx.balance && cx.native_loop && ntiles <= GSR_ORDER_MAX_TILES && !sp.state;
    for (int k = 0; k < 2; k++) { pa.tile_work[k] = balanced ? im.tile_work[k] : nullptr; pa.tile_order[k] = balanced ? im.tile_order[k] : nullptr; }
    // the stateless entry points: no previous iteration to learn the tiles' weights from -- see tile_order_block
    const bool stateless_balance = stateless_balanced(cx, ntiles);
    pa.order_tiles = ntiles;
    pa.dirty = cx.native_loop ? g.dirty : nullptr;
    pa.rows = cx.rows;
    // (inside gsr_refine the list counters are cleared by the chain-rule kernel's last workgroup, once every consumer is done)
    ZeroList zl;
    if (!cx.native_loop) { rc = zl.add(g.surv.n, (size_t)GSR_SURV_LISTS * GSR_SURV_CSTRIDE * sizeof(uint32_t), st); if (rc != GSR_OK) return rc; }
    // Two ways to a tile's list.  With depth bounds from a previous forward (speculation) the few surviving instances are
    // appended to fixed-capacity per-tile bins by the preprocess itself; without them every instance is binned exactly
    // (count -> scan -> emit) and the compositing kernel orders each tile's segment lazily.  (More than 65 536 tiles: the
    // fixed-capacity bins would not fit; such a forward bins exactly and only records bounds.)
    const bool by_tile = sp.mode == 1 && ntiles <= 65536;
    // (complete lists: the compositing kernel evaluates the colours of the splats it stages, see LazySH)
    // (also for short SH rows: with degree-1 maps -- train.py's first iterations -- evaluating the colours up front in k_sh_color, overlapped
    // with the binning on the side stream, takes 32 us off this kernel and puts 27 us onto k_tile_count: measured, a wash)
    pa.lazy_sh = (!by_tile && colors_precomp == nullptr && M <= 16 && !debug && !cx.sh_eager) ? 1 : 0;
    const float* zb_prev = by_tile ? im.zb[sp.parity ^ 1] : nullptr;
    float* zb_next = (sp.mode != 0) ? im.zb[sp.parity] : nullptr;
    pa.zb = zb_prev;
    pa.zb_mul = sp.mul; pa.zb_add = sp.add;
    pa.zbc = zb_prev ? im.zbc[sp.parity ^ 1] : nullptr; pa.sbx = im.sbx; pa.sby = im.sby;
    float* zbc_next = (sp.mode != 0) ? im.zbc[sp.parity] : nullptr;
    BinLocal bl{nullptr, nullptr};
    // Complete lists (no depth bounds to speculate with): binned by the preprocess kernel itself into fixed-capacity bins
    // (k_preprocess_bin) while their counters fit the LDS; otherwise, and after a bin overflowed, count -> scan -> emit.
    // (measured: 85 us against 30 + 20 + 8 + 43 us and a blocking read at 1 200 tiles / 1 M Gaussians, but 300 against 253 us at the
    // training configuration's 4 293 tiles / 1.5 M: large images keep the three-kernel path)
    const bool full_bins = !by_tile && !cx.exact_bins && ntiles <= kFullBinMaxTiles;
    if (cx.used_full_bins) *cx.used_full_bins = full_bins;
    const int bin_cap = full_bins ? full_bin_capacity(P, ntiles)
                                  : ((sp.state != nullptr) ? GSR_LSORT_CAP : bin_capacity(ntiles));      // (a caller's state buffer holds 2048-entry bins)
    pa.bin_cap = bin_cap;
    if (full_bins) {
        char* lptr = (char*)binning_buffer(binning_ctx, carve_bin_local(nullptr, ntiles, bin_cap, bl));
        if (!lptr) return fail(GSR_E_ALLOC, "binning buffer callback returned NULL%s", "");
        carve_bin_local(lptr, ntiles, bin_cap, bl);
    }
    if (by_tile) {
        // (with a state buffer the unsorted bins live there and the per-call buffer only holds the sorted lists)
        const size_t lbytes = state_bins ? (size_t)ntiles * (GSR_LSORT_CAP + GSR_BIN_PAD) * sizeof(uint32_t) : carve_bin_local(nullptr, ntiles, bin_cap, bl);
        char* lptr = (char*)binning_buffer(binning_ctx, lbytes);
        if (!lptr) return fail(GSR_E_ALLOC, "binning buffer callback returned NULL%s", "");
        if (state_bins) { bl.vals = reinterpret_cast<uint32_t*>(lptr); bl.bins = state_bins; }
        else carve_bin_local(lptr, ntiles, bin_cap, bl);
    }
    // Heavy tiles split across workgroups (gsr::SegCtl): the speculative iterations of the native loop, on the launch list the previous
    // group's backward built from the work its forward measured.
    const bool use_seg = cx.seg && cx.seg_ready && by_tile && balanced && im.seg_budget > 0 && !sp.state && cx.guard.poison != nullptr;
    if (cx.seg_used) *cx.seg_used = use_seg;
    gsr::SegCtl sg = {};
    if (use_seg) sg = gsr::SegCtl{im.seg_list[sp.parity ^ 1], im.seg_cnt, im.seg_pub, im.seg_rec, im.seg_ticket, im.seg_nosplit, cx.guard.tag, im.seg_len};
    else if (cx.seg && balanced && im.seg_budget > 0) sg.len = im.seg_len;      // (every forward of the loop reports how much of each tile's list it ordered)
    sg.hold_after = cx.hold_after;
    // (bounds widened by the previous group's backward -- dilate_bounds -- come with the tiles' own bounds next to them)
    if (cx.seg && cx.seg_ready && by_tile && im.seg_budget > 0 && !sp.state && !(cx.flags & GSR_REFINE_NO_DILATE)) { sg.zb_own_used = im.zb_own[sp.parity ^ 1]; sg.nodilate = im.nodilate; }
    pa.tile_cursor = (by_tile || full_bins) ? im.tile_cursor : nullptr;
    pa.bins = bl.bins;
    pa.tile_count = (by_tile || full_bins) ? nullptr : im.tile_count;
    pa.ntiles = ntiles;
    // (on the by-tile path inside gsr_refine these words are cleared by the kernels that consume them: the tile cursors
    // by the compositing kernel, the superblock bounds by the pose step)
    if (sp.mode != 0 && !(by_tile && cx.native_loop)) {
        rc = zl.add(im.fail, im.clear_words * sizeof(uint32_t), st); if (rc != GSR_OK) return rc;
        rc = zl.add(zbc_next, (size_t)im.nsb * sizeof(float), st); if (rc != GSR_OK) return rc;
    } else if (full_bins && !cx.native_loop) {     // (the stateless entry points get a fresh image buffer per call: flag + cursors)
        rc = zl.add(im.fail, im.clear_words * sizeof(uint32_t), st); if (rc != GSR_OK) return rc;
    }
    rc = zl.flush(st); if (rc != GSR_OK) return rc;
    {
        ProfScope ps(K_PREPROCESS, st);
        // (the bounds the conservative tests look at: per tile while they fit the LDS comfortably -- 4 096 tiles: 16 KB + the coarser levels --,
        // per 4 x 4-tile superblock above that)
// (measured in round 5 with 4 096: nothing gained -- S-1M-640-object's preprocess kernel 72.7 us either way, the uniform cloud's +1 us for
// the larger table -- so it is off; -DGSR_PYR_TILES_MAX=4096 builds it)
#ifndef GSR_PYR_TILES_MAX
#define GSR_PYR_TILES_MAX 0
#endif
        pa.pyr_tiles = (pa.zbc != nullptr && pa.zb != nullptr && ntiles <= GSR_PYR_TILES_MAX) ? 1 : 0;
        pa.zbc_lds = pa.pyr_tiles ? ntiles : ((pa.zbc != nullptr && im.nsb <= 4096) ? im.nsb : 0);
        const size_t pyr_bytes = (pa.zbc_lds > 0 ? (pa.pyr_tiles ? bound_pyramid_floats(gx, gy) : bound_pyramid_floats(im.sbx, im.sby)) : 0) * sizeof(float);
        // (k_preprocess_lean trades parallelism for instruction count -- a wave per 256 Gaussians: it pays from a few hundred thousand
        // Gaussians on; a 50 k map keeps the one-lane-per-Gaussian kernel: 8 270 against 7 015 it/s)
        // (cov_cache == 2: the workspace holds every Gaussian's covariance AND the extent bound the conservative test reads)
        pa.lean = (cx.lean && cx.native_loop && by_tile && cx.cov_cache == 2 && pa.cov_all == 0 && pa.zbc_lds > 0 && scales != nullptr &&
                   cov3D_precomp == nullptr && P >= cx.lean_min_P && !(cx.flags & GSR_REFINE_NO_LEAN)) ? 1 : 0;
        pa.sh_here = (pa.lean && colors_precomp == nullptr && M <= 16 && !(cx.flags & GSR_REFINE_SH_SEPARATE)) ? 1 : 0;
        if (pa.lean && cx.n_lean) ++*cx.n_lean;
        // (the exact-bin path has the preprocess zero the per-tile counters, all copies: at least that many threads)
        pa.ntiles = ntiles * im.copies;
        // (the first two workgroups also compute the launch orders of the compositing kernels: there must be two)
        const int blocks = std::max(by_tile ? pblocks : std::max(pblocks, (pa.ntiles + GSR_BLOCK - 1) / GSR_BLOCK), balanced ? 2 : 1);
        if (full_bins) {
            rc = allow_large_lds(dev);
            if (rc != GSR_OK) return rc;
            pa.lean = 0; pa.sh_here = 0;
            pa.ntiles = ntiles;
            // bands of tile rows the emit walks one after the other (see k_tile_emit): every band is another pass over the wave's row
            // lists (~5 us on S-1M-640), so as few as keep a band's scattered 8-byte stores inside the L2s until their lines are
            // complete -- measured on the MI355X: 1 band 4 040 it/s, 2 bands 4 180, 3 bands 4 110, 5 bands 3 960 (S-1M-640, plain loop)
            int bands = (int)(((size_t)P * 20 + (12u << 20) - 1) / (12u << 20));
#ifndef GSR_BIN_BANDS_MAX
#define GSR_BIN_BANDS_MAX 4
#endif
            bands = std::max(1, std::min(bands, std::min(gy, GSR_BIN_BANDS_MAX)));
            const int gpb = GSR_PBIN_KPT * GSR_PBIN_THREADS;
            hipLaunchKernelGGL(k_preprocess_bin, dim3(std::max((P + gpb - 1) / gpb, balanced ? 2 : 1)), dim3(GSR_PBIN_THREADS),
                               (size_t)2 * ntiles * sizeof(uint32_t), st, pa, bands);
        } else if (pa.lean) {
            // radii are not an output of this forward: conservative test for all Gaussians, exact geometry for the few it leaves
            // (whole windows of GSR_LEAN_WINDOW waves: the kernel deals the Gaussians of a window out segment by segment)
            const int wblocks = GSR_LEAN_WINDOW / 4;
            const int lblocks = std::max(((P + GSR_LEAN_PER_LANE * GSR_BLOCK - 1) / (GSR_LEAN_PER_LANE * GSR_BLOCK) + wblocks - 1) / wblocks * wblocks, balanced ? 2 : 1);
            // (LDS: the superblock bounds and the coarser levels the kernel builds behind them)
            hipLaunchKernelGGL(k_preprocess_lean, dim3(lblocks), dim3(GSR_BLOCK), pyr_bytes, st, pa);
        } else
            hipLaunchKernelGGL(k_preprocess, dim3(blocks), dim3(GSR_BLOCK), pyr_bytes, st, pa);
    }
    LAUNCHCHK("k_preprocess");
    // SH colours only feed the compositing kernel: fork them onto the side stream, join before K6.  (Not on the
    // by-tile path: without the binning chain there is nothing latency-bound to hide them under, and the
    // fork/join costs more than it gains -- measured 0.466 vs 0.434 ms per iteration.)
    SideLease side_lease(colors_precomp == nullptr && !debug && !by_tile && !pa.lazy_sh, dev);
    Side* side = side_lease.sd;
    if (colors_precomp == nullptr && !pa.sh_here && !pa.lazy_sh) {
        if (side) {
            HIPCHK(hipEventRecord(side->fork, st));
            HIPCHK(hipStreamWaitEvent(side->st, side->fork, 0));
            hipLaunchKernelGGL(k_sh_color, dim3(surv_grid(P, GSR_SHC_RESIDENT)), dim3(64), 0, side->st, pa);
            HIPCHK(hipEventRecord(side->join, side->st));
        } else {
            ProfScope ps(K_SH_COLOR, st);
            hipLaunchKernelGGL(k_sh_color, dim3(surv_grid(P, GSR_SHC_RESIDENT)), dim3(64), 0, st, pa);
        }
        LAUNCHCHK("k_sh_color");
    }
    int R = 0;
    Bin b{nullptr, nullptr};
    if (!by_tile && !full_bins) {
        TileBinArgs ta;
        ta.P = P; ta.gx = gx; ta.gy = gy; ta.ntiles = ntiles; ta.gpb = tile_bin_gpb(P);
        ta.tiles_touched = g.tiles_touched; ta.rects = g.rects; ta.rec = g.rec;
        ta.tile_count = im.tile_count; ta.tile_start = im.tile_start; ta.tile_offset = im.tile_offset; ta.tile_fill = im.tile_fill; ta.keys = nullptr;
        ta.block_counts = im.block_counts; ta.copies = im.copies;
        const int tblocks = (P + ta.gpb - 1) / ta.gpb;
        // (LDS aggregation: counters for every tile fit, and a lane keeps its 2 or 4 Gaussians in registers)
        const int kpt = ta.gpb / GSR_TBIN_THREADS;
        const bool agg = ntiles <= kTileBinLdsTiles && kpt <= 4;
        if (agg && ntiles > 8 * 1024) { rc = allow_large_lds(dev); if (rc != GSR_OK) return rc; }
        {
            ProfScope ps(K_TILE_COUNT, st);
            const size_t lds = (size_t)ntiles * sizeof(uint32_t);
            if (agg && kpt <= 2) hipLaunchKernelGGL((k_tile_count<true, 2>), dim3(tblocks), dim3(GSR_TBIN_THREADS), lds, st, ta);
            else if (agg) hipLaunchKernelGGL((k_tile_count<true, 4>), dim3(tblocks), dim3(GSR_TBIN_THREADS), lds, st, ta);
            else hipLaunchKernelGGL((k_tile_count<false, 2>), dim3(tblocks), dim3(GSR_TBIN_THREADS), 0, st, ta);
        }
        LAUNCHCHK("k_tile_count");
        {
            ProfScope ps(K_TILE_SCAN, st);
            hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, ntiles, agg ? im.copies : 1, (const uint32_t*)im.tile_count, im.tile_start,
                               im.tile_offset, im.tile_fill, im.total, (volatile uint32_t*)nullptr, 0u);
        }
        LAUNCHCHK("k_tile_scan");
        // one blocking 4-byte read, as rasterizer_impl.cu:282: the instance arrays are sized by it.  (Having the scan kernel write
        // the total into pinned host memory that the host polls instead was measured: no difference, 0.336 vs 0.337 ms / iteration.)
        uint32_t num_rendered_u = 0;
        HIPCHK(hipMemcpyAsync(&num_rendered_u, im.total, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (num_rendered_u > 0x7fffffffu) return fail(GSR_E_INVALID, "more than 2^31 tile instances%s", "");
        R = (int)num_rendered_u;
        const size_t bbytes = carve_bin(nullptr, R, b);
        char* bptr = (char*)binning_buffer(binning_ctx, bbytes);
        if (!bptr) return fail(GSR_E_ALLOC, "binning buffer callback returned NULL%s", "");
        carve_bin(bptr, R, b);
        if (R > 0) {
            ta.keys = b.keys;
            // bands: each band's share of the key array should fit the L2s (8 x 4 MB) with room to spare
            int bands = (int)(((size_t)R * 8 + (12u << 20) - 1) / (12u << 20));
#ifndef GSR_EMIT_BANDS_MAX
#define GSR_EMIT_BANDS_MAX 16
#endif
            bands = std::max(1, std::min(bands, std::min(gy, GSR_EMIT_BANDS_MAX)));
            ProfScope ps(K_TILE_EMIT, st);
            const size_t lds = (size_t)2 * ntiles * sizeof(uint32_t);
            if (agg && kpt <= 2) hipLaunchKernelGGL((k_tile_emit<true, 2>), dim3(tblocks), dim3(GSR_TBIN_THREADS), lds, st, ta, bands);
            else if (agg) hipLaunchKernelGGL((k_tile_emit<true, 4>), dim3(tblocks), dim3(GSR_TBIN_THREADS), lds, st, ta, bands);
            else hipLaunchKernelGGL((k_tile_emit<false, 2>), dim3(tblocks), dim3(GSR_TBIN_THREADS), 0, st, ta, 1);
        }
        LAUNCHCHK("k_tile_emit");
    }
    if (side) HIPCHK(hipStreamWaitEvent(st, side->join, 0));
    {
        ProfScope psr(K_RENDER_FWD, st);
        const bool local = by_tile || full_bins;
        // (stateless: the tiles' work is recorded for the backward's launch order, tile_order_block.  The forward itself keeps the natural
        // order: list LENGTHS, known after the scan, are a poor predictor of a tile's work -- measured at the train step's 4 293 tiles /
        // 1.5 M Gaussians: 214 against 216 us, where the order by the TRUE work would give 150 against 180 on a second run of the
        // same frame -- and nothing better is known before the kernel has run)
#define GSR_FWD_ARGS im.ranges, local ? bl.vals : b.vals, local ? (const unsigned long long*)bl.bins : (const unsigned long long*)b.keys, \
                     local ? im.tile_cursor : im.tile_offset, width, height, gx, ntiles, (const float*)g.rec, background, out_color, out_depth, out_alpha, im.n_contrib, n_touched, \
                     zb_next, zb_prev, cx.guard.poison ? const_cast<uint32_t*>(cx.guard.poison) : im.fail, \
                     sp.mul, sp.add, zbc_next, im.sbx, cx.floss, (const uint32_t*)pa.tile_order[0], (balanced || stateless_balance) ? im.tile_work[0] : (uint32_t*)nullptr, bin_cap, \
                     LazySH{pa.lazy_sh ? shs : nullptr, means3D, cam_pos, D, M, g.clamped, g.rec}, (cx.guard.poison ? cx.guard.tag << 2 : 0u), \
                     full_bins ? im.tile_count : (uint32_t*)nullptr, (P < (1 << 28)) ? 1 : 0, \
                     (cx.native_loop && !sp.state && sp.mode != 0) ? im.tile_hold : (uint32_t*)nullptr, sg
        if (by_tile) {
            const int fgrid = use_seg ? ((cx.seg_grid > 0 && cx.seg_grid <= im.seg_budget) ? cx.seg_grid : im.seg_budget) : ntiles;
            if (n_touched) hipLaunchKernelGGL((k_render_fwd<true, GSR_LIST_BINS>), dim3(fgrid), dim3(GSR_BLOCK), 0, st, GSR_FWD_ARGS);
            else hipLaunchKernelGGL((k_render_fwd<false, GSR_LIST_BINS>), dim3(fgrid), dim3(GSR_BLOCK), 0, st, GSR_FWD_ARGS);
        } else if (full_bins) {
            if (n_touched) hipLaunchKernelGGL((k_render_fwd<true, GSR_LIST_BINS_FULL>), dim3(ntiles), dim3(GSR_BLOCK), 0, st, GSR_FWD_ARGS);
            else hipLaunchKernelGGL((k_render_fwd<false, GSR_LIST_BINS_FULL>), dim3(ntiles), dim3(GSR_BLOCK), 0, st, GSR_FWD_ARGS);
        } else {
            if (n_touched) hipLaunchKernelGGL((k_render_fwd<true, GSR_LIST_EXACT>), dim3(ntiles), dim3(GSR_BLOCK), 0, st, GSR_FWD_ARGS);
            else hipLaunchKernelGGL((k_render_fwd<false, GSR_LIST_EXACT>), dim3(ntiles), dim3(GSR_BLOCK), 0, st, GSR_FWD_ARGS);
        }
#undef GSR_FWD_ARGS
    }
    LAUNCHCHK("k_render_fwd");
    if (full_bins && !cx.native_loop) {
        // The stateless entry points: one blocking 4-byte read per forward, as rasterizer_impl.cu:282 -- there the instance count,
        // here the flag word: did a tile's complete list overflow its bin?  Then the forward is redone on the exact path.
        // (gsr_refine learns the same from the group's status word.)
        uint32_t flag = 0;
        std::vector<uint32_t> per_tile((size_t)ntiles);
        HIPCHK(hipMemcpyAsync(&flag, im.fail, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(per_tile.data(), im.tile_count, per_tile.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        unsigned long long total = 0;
        for (uint32_t c : per_tile) total += c;
        R = (int)std::min<unsigned long long>(total, 0x7fffffffull);          // num_rendered (after exact tile culling), as the reference returns it
        if (flag & GSR_FAIL_OVERFLOW) {
            PassCtx cx2 = cx;
            cx2.exact_bins = true;
            return forward_impl(cx2, GSR_FWD_PASS);
        }
    }
    return R;
}

#define GSR_BWD_PARAMS int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,                    \
                       const float* shs, const float* colors_precomp, const float* alphas, const float* scales, float scale_modifier,       \
                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,                \
                       const float* campos, float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,      \
                       char* img_buffer, const float* dL_dpix, const float* dL_ddepths, const float* dL_dalphas, float* dL_dmean2D,         \
                       float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,          \
                       float* dL_dscale, float* dL_drot, int debug, int pose_mode, float* dL_dtau, void* stream
#define GSR_BWD_PASS P, D, M, R, background, width, height, means3D, shs, colors_precomp, alphas, scales, scale_modifier, rotations,        \
                     cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, img_buffer,     \
                     dL_dpix, dL_ddepths, dL_dalphas, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,         \
                     dL_dscale, dL_drot, debug, pose_mode, dL_dtau, stream

int backward_impl(const PassCtx& cx, GSR_BWD_PARAMS)
{
    // dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot may be NULL: that gradient is then not written
    using namespace gsr;
    hipStream_t st = (hipStream_t)stream;
    if (P < 0 || R < 0 || width <= 0 || height <= 0) return fail(GSR_E_INVALID, "bad sizes%s", "");
    if (P == 0) {
        if (pose_mode && dL_dtau) {
            int rc0 = select_device_of(dL_dtau);
            if (rc0 != GSR_OK) return rc0;
            HIPCHK(hipMemsetAsync(dL_dtau, 0, 6 * sizeof(float), st));
        }
        return 0;
    }
    if (!means3D || !radii || !geom_buffer || !binning_buffer || !img_buffer || !alphas || !dL_dpix || !dL_ddepths ||
        !dL_dalphas || !dL_dmean2D || !dL_dconic || !dL_dopacity || !dL_dcolor)
        return fail(GSR_E_INVALID, "a required backward pointer is NULL%s", "");
    if (pose_mode && !dL_dtau) return fail(GSR_E_INVALID, "pose_mode needs dL_dtau%s", "");
    int dev = 0;
    int rc = select_device_of(means3D, &dev);
    if (rc != GSR_OK) return rc;

    const int gx = (width + GSR_TILE - 1) / GSR_TILE, gy = (height + GSR_TILE - 1) / GSR_TILE;
    const int ntiles = gx * gy;
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    Geom g; carve_geom(geom_buffer, P, g, cx.det, cx.native_loop);
    if (cx.native_loop) g.select_set(cx.set);
    // (the ordered index lists sit at the start of the binning buffer on both binning paths; ranges[] says where)
    const uint32_t* point_list = reinterpret_cast<const uint32_t*>(binning_buffer);
    Img im; carve_img(img_buffer, width, height, im, cx.seg);

    // Gradient tensors are zero-filled on the side stream while K7 runs; K8/K9 then only writes non-zero rows.
    // (the native loop zero-fills once per frame and keeps the tensors consistent through the dirty bits)
    // (a side stream only pays when there is something sizeable to overlap: below ~100 k Gaussians the fork / join events cost the
    // host more than the fill takes)
    SideLease side_lease(!(debug || cx.native_loop) && P >= 100000, dev);
    Side* side = side_lease.sd;
    hipStream_t zs = side ? side->st : st;
    if (side) {
        HIPCHK(hipEventRecord(side->fork, st));
        HIPCHK(hipStreamWaitEvent(side->st, side->fork, 0));
    }
    if (!cx.native_loop) {
        // (a caller that carves its gradient tensors out of one allocation -- this repo's Python layer does -- gets ONE memset:
        // exactly adjacent ranges are merged)
        const size_t Pn = (size_t)P;
        struct Range { char* p; size_t n; };
        Range rg[9];
        int nr = 0;
        auto add = [&](float* ptr, size_t bytes) { if (ptr && bytes) rg[nr++] = Range{reinterpret_cast<char*>(ptr), bytes}; };
        add(dL_dmean2D, Pn * 3 * sizeof(float)); add(dL_dconic, Pn * 4 * sizeof(float)); add(dL_dopacity, Pn * sizeof(float));
        add(dL_dcolor, Pn * 3 * sizeof(float)); add(dL_dmean3D, Pn * 3 * sizeof(float)); add(dL_dcov3D, Pn * 6 * sizeof(float));
        if (M > 0) add(dL_dsh, Pn * M * 3 * sizeof(float));
        add(dL_dscale, Pn * 3 * sizeof(float)); add(dL_drot, Pn * 4 * sizeof(float));
        std::sort(rg, rg + nr, [](const Range& x, const Range& y) { return x.p < y.p; });
        for (int i = 0; i < nr;) {
            char* p0 = rg[i].p;
            size_t n = rg[i].n;
            int j = i + 1;
            while (j < nr && rg[j].p == p0 + n) { n += rg[j].n; j++; }
            HIPCHK(hipMemsetAsync(p0, 0, n, zs));
            i = j;
        }
    }
    if (side) HIPCHK(hipEventRecord(side->join, side->st));
    if (!cx.native_loop) {      // accumulators of K7 (atomically summed), its flags
        ProfScope psz(K_BWD_ZERO, st);
        ZeroList zl;
        // (a large map: only the survivors' records, through the work lists -- k_backward_prologue)
#ifndef GSR_ACC_CLEAR_MIN_P
#define GSR_ACC_CLEAR_MIN_P 100000
#endif
        const bool by_list = P >= GSR_ACC_CLEAR_MIN_P;
        if (!by_list) {
            rc = zl.add(g.acc, (size_t)P * GSR_ACC_STRIDE * (cx.det ? 2 * sizeof(long long) : sizeof(float)), st); if (rc != GSR_OK) return rc;
            rc = zl.add(g.aflag, (2 * (size_t)P + 3) & ~(size_t)3, st); if (rc != GSR_OK) return rc;
        }
        if (pose_mode) { rc = zl.add(g.tau_acc, (cx.det ? 16 : 8) * GSR_TAU_SLOTS * sizeof(double), st); if (rc != GSR_OK) return rc; }
        rc = zl.flush(st); if (rc != GSR_OK) return rc;
        // one launch: the survivors' records cleared through the work lists and, in its first workgroup, K7's launch order (k_backward_prologue)
        const bool order_wanted = stateless_balanced(cx, ntiles);
        if (by_list || order_wanted) {
            const int clear_blocks = by_list ? surv_grid(P, 512) : 0;          // (1 024-entry chunks)
            if (cx.det) hipLaunchKernelGGL(k_backward_prologue<true>, dim3(clear_blocks + (order_wanted ? 1 : 0)), dim3(GSR_TILE_ORDER_THREADS), 0, st, g.surv, g.acc, g.aflag, P,
                                           clear_blocks, (const uint32_t*)im.tile_work[0], order_wanted ? im.tile_order[1] : (uint32_t*)nullptr, ntiles);
            else hipLaunchKernelGGL(k_backward_prologue<false>, dim3(clear_blocks + (order_wanted ? 1 : 0)), dim3(GSR_TILE_ORDER_THREADS), 0, st, g.surv, g.acc, g.aflag, P,
                                    clear_blocks, (const uint32_t*)im.tile_work[0], order_wanted ? im.tile_order[1] : (uint32_t*)nullptr, ntiles);
            LAUNCHCHK("k_backward_prologue");
        }
    }
    {
        ProfScope psb(K_RENDER_BWD, st);
#define GSR_BWD_ARGS (const uint2*)im.ranges, point_list, width, height, gx, ntiles, background, alphas, (const uint32_t*)im.n_contrib, dL_dpix, \
                     dL_ddepths, dL_dalphas, g.acc
        const bool balanced = cx.balance && cx.native_loop && ntiles <= GSR_ORDER_MAX_TILES;
        const uint32_t* order = balanced ? im.tile_order[1] : nullptr;
        uint32_t* work = balanced ? im.tile_work[1] : nullptr;
        if (stateless_balanced(cx, ntiles)) order = im.tile_order[1];      // heaviest tiles first, by what this call's forward measured (k_backward_prologue)
        // (the forward of this group split its heavy tiles: the same launch list, the same blocks -- gsr::SegCtl)
        const bool use_seg = cx.seg && cx.seg_used && *cx.seg_used && im.seg_budget > 0;
        const gsr::SegCtl sg = use_seg ? gsr::SegCtl{im.seg_list[cx.spec.parity ^ 1], im.seg_cnt, im.seg_pub, im.seg_rec, im.seg_ticket, im.seg_nosplit, cx.guard.tag, im.seg_len} : gsr::SegCtl{};
        const int kgrid = use_seg ? ((cx.seg_grid > 0 && cx.seg_grid <= im.seg_budget) ? cx.seg_grid : im.seg_budget) : ntiles;
        // (work measured by a forward with COMPLETE lists says little about the speculative forward that follows -- its ordering overhead is
        // that of lists several times as long: the list built from it splits nobody.  Round 5: S-1M-640-walls, whose speculative iterations
        // never split a tile, had 460 of its 1 200 tiles split in the iteration after each complete-list forward.)
        const int next_budget = (cx.seg_grid_next > 0 && cx.seg_grid_next <= im.seg_budget) ? cx.seg_grid_next : im.seg_budget;
        // (... and one more workgroup builds the next group's list from the work this group's forward measured; the forward's order array is
        // its scratch: the next forward either runs the list or has the preprocess kernel compute its order afresh)
        const bool build = cx.seg && im.seg_budget > 0 && cx.native_loop && balanced;
        const gsr::SegBuild sb = build ? gsr::SegBuild{im.tile_work[0], im.tile_order[0], im.seg_list[cx.spec.parity], im.seg_nosplit, ntiles, next_budget, kgrid, im.seg_len,
                                                        (cx.spec.mode != 0 && !cx.spec.state && !(cx.flags & GSR_REFINE_NO_DILATE)) ? im.zb[cx.spec.parity] : (float*)nullptr, im.zbc[cx.spec.parity], gx, gy, im.sbx,
                                                        im.zb_own[cx.spec.parity], im.nodilate, (cx.spec.mode == 2 || !cx.split_ok_now) ? 0 : 1, cx.seg_host_total} : gsr::SegBuild{};
        if (pose_mode) hipLaunchKernelGGL(k_render_bwd_mfma<true>, dim3(kgrid + (build ? 1 : 0)), dim3(GSR_BLOCK), 0, st, GSR_BWD_ARGS, cx.guard, order, work, (const float*)g.rec, P, (P < (1 << 28)) ? 1 : 0, cx.det ? 1 : 0, g.aflag, sg, sb);
        else hipLaunchKernelGGL(k_render_bwd_mfma<false>, dim3(kgrid + (build ? 1 : 0)), dim3(GSR_BLOCK), 0, st, GSR_BWD_ARGS, cx.guard, order, work, (const float*)g.rec, P, (P < (1 << 28)) ? 1 : 0, cx.det ? 1 : 0, g.aflag, sg, sb);
#undef GSR_BWD_ARGS
    }
    LAUNCHCHK("k_render_bwd");

    if (side) HIPCHK(hipStreamWaitEvent(st, side->join, 0));
    PreBwdArgs pb;
    pb.P = P; pb.D = D; pb.M = M;
    pb.means = means3D; pb.radii = radii; pb.shs = shs; pb.clamped = g.clamped;
    pb.scales = scales; pb.rots = rotations; pb.mod = scale_modifier;
    pb.cov3D = cov3D_precomp ? cov3D_precomp : g.cov3D;
    pb.rec = g.rec;
    pb.view = viewmatrix; pb.proj = projmatrix; pb.campos = campos;
    pb.fx = focal_x; pb.fy = focal_y; pb.tanx = tan_fovx; pb.tany = tan_fovy;
    pb.acc = g.acc;
    pb.dL_dmean2D = dL_dmean2D; pb.dL_dconic = dL_dconic; pb.dL_dopacity = dL_dopacity; pb.dL_dcolor = dL_dcolor;
    pb.dL_dmean3D = dL_dmean3D; pb.dL_dcov3D = dL_dcov3D; pb.dL_dsh = dL_dsh; pb.dL_dscale = dL_dscale; pb.dL_drot = dL_drot;
    pb.pose = pose_mode ? 1 : 0; pb.tau_acc = g.tau_acc;
    pb.dirty = cx.native_loop ? g.dirty : nullptr;
    pb.aflag = g.aflag;
    pb.role = cx.native_loop ? 1 : 0;
    pb.rows_every = (cx.native_loop && (cx.flags & GSR_REFINE_GRADS_EVERY_ITERATION)) ? 1 : 0;
    pb.o_surv = g.o_surv; pb.o_aflag = g.o_aflag; pb.o_acc = g.o_acc; pb.o_rec = g.o_rec; pb.o_clamped = g.o_clamped;
    pb.o_cam = g.prev_cam; pb.final_done = g.final_done;
    pb.guard = cx.guard;
    pb.ticket = cx.ticket; pb.fold = cx.fold;
    if (pb.ticket) { pb.fold.tau_acc = g.tau_acc; pb.fold.prev_cam = g.prev_cam; }
    pb.fold.det = cx.det ? 1 : 0;
    {
        ProfScope ps(K_PREPROCESS_BWD, st);
        pb.surv = g.surv;
        if (cx.pb_out) *cx.pb_out = pb;
        // (launching only as many waves as the survivors' lists have chunks -- 256 instead of 2 048 in a speculative iteration --
        // was measured in round 4: 24.0 against 24.3 us; the waves that find nothing cost nothing)
        const int k8_grid = surv_grid(P, GSR_K8_RESIDENT);
        if (cx.det) hipLaunchKernelGGL(k_preprocess_bwd<true>, dim3(k8_grid), dim3(64), 0, st, pb);
        else hipLaunchKernelGGL(k_preprocess_bwd<false>, dim3(k8_grid), dim3(64), 0, st, pb);
    }
    LAUNCHCHK("k_preprocess_bwd");
    if (pose_mode && !cx.native_loop) {
        if (cx.det) hipLaunchKernelGGL(k_tau_finish_det, dim3(1), dim3(64), 0, st, (const long long*)g.tau_acc, viewmatrix, dL_dtau);
        else hipLaunchKernelGGL(k_tau_finish, dim3(1), dim3(64), 0, st, (const double*)g.tau_acc, dL_dtau);
        LAUNCHCHK("k_tau_finish");
    }
    return 0;
}

}  // namespace

// (debug bit 1, value 2: diagnostics -- SH colours of every visible Gaussian up front instead of lazily in the compositing kernel)
static PassCtx dropin_ctx(int& debug)
{
    PassCtx cx;
    cx.sh_eager = (debug & 2) != 0;
    cx.det = (debug & 4) != 0;
    debug &= 1;
    return cx;
}
int gsr_forward(GSR_FWD_PARAMS) { const PassCtx cx = dropin_ctx(debug); return forward_impl(cx, GSR_FWD_PASS); }

size_t gsr_spec_state_bytes(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    Img im; return carve_spec(nullptr, width, height, im, nullptr);
}
size_t gsr_spec_state_bounds_bytes(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    Img im;
    char* base = reinterpret_cast<char*>((uintptr_t)4096);      // (never dereferenced: carve_spec only does pointer arithmetic)
    carve_spec(base, width, height, im, nullptr);
    return (size_t)(reinterpret_cast<char*>(im.fail) - base);
}

int gsr_forward_speculative(gsr_spec_state* s, GSR_FWD_PARAMS)
{
    using namespace gsr;
    PassCtx cx = dropin_ctx(debug);
    if (!s || !s->device_buffer || P <= 0 || width <= 0 || height <= 0) {
        if (s) { s->valid = 0; s->last_speculative = 0; }
        return forward_impl(cx, GSR_FWD_PASS);
    }
    if (s->width == 0 && s->height == 0) { s->width = width; s->height = height; }
    if (s->width != width || s->height != height)
        return fail(GSR_E_INVALID, "gsr_forward_speculative: the state was sized for another image size%s", "");
    hipStream_t st = (hipStream_t)stream;
    cx.spec.state = static_cast<char*>(s->device_buffer);
    s->last_speculative = 0;
    const int next = (s->parity ^ 1) & 1;          // the bound buffer this forward writes
    const int ntiles = ((width + GSR_TILE - 1) / GSR_TILE) * ((height + GSR_TILE - 1) / GSR_TILE);
    if (s->valid && s->skip == 0 && ntiles <= 65536) {
        cx.spec.mode = 1; cx.spec.parity = next;
        const int R = forward_impl(cx, GSR_FWD_PASS);
        if (R < 0) { s->valid = 0; return R; }
        Img im; carve_spec(cx.spec.state, width, height, im, nullptr);
        uint32_t failed = 0;      // the one blocking read of this forward
        HIPCHK(hipMemcpyAsync(&failed, im.fail, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (failed == 0u) {
            s->parity = next; s->fail_streak = 0; s->n_speculative++; s->last_speculative = 1;
            return R;
        }
        s->n_failed++; s->fail_streak++;
        if (s->fail_streak >= 2) s->skip = 1 << (s->fail_streak < 7 ? s->fail_streak - 1 : 6);      // 2, 4, ... 64
    } else if (s->skip > 0) s->skip--;
    cx.spec.mode = 2; cx.spec.parity = next;         // complete lists (exact bins); records the bounds
    const int R = forward_impl(cx, GSR_FWD_PASS);
    if (R < 0) { s->valid = 0; return R; }
    s->parity = next; s->valid = 1;
    return R;
}

int gsr_backward(GSR_BWD_PARAMS)
{
    const PassCtx cx = dropin_ctx(debug);
    if (cx.det && geometry_sized_for_det(geom_buffer) == 0)
        return fail(GSR_E_INVALID, "gsr_backward: debug bit 2 (deterministic sums) needs a geometry workspace whose FORWARD ran with the same bit "
                                   "(it sizes the 64-bit accumulator records: gsr_geometry_bytes_det)%s", "");
    return backward_impl(cx, GSR_BWD_PASS);
}

// one struct pointer across the foreign-function boundary instead of 34 / 40 arguments (include/gsr.h)
int gsr_forward_packed(const gsr_forward_args* a)
{
    if (!a) return fail(GSR_E_INVALID, "gsr_forward_packed: NULL argument%s", "");
    return gsr_forward_speculative(a->state, a->geometry_buffer, a->geometry_ctx, a->binning_buffer, a->binning_ctx, a->image_buffer, a->image_ctx,
                                   a->P, a->D, a->M, a->background, a->width, a->height, a->means3D, a->shs, a->colors_precomp, a->opacities,
                                   a->scales, a->scale_modifier, a->rotations, a->cov3D_precomp, a->viewmatrix, a->projmatrix, a->cam_pos,
                                   a->tan_fovx, a->tan_fovy, a->prefiltered, a->out_color, a->out_depth, a->out_alpha, a->radii, a->debug,
                                   a->n_touched, a->stream);
}
int gsr_backward_packed(const gsr_backward_args* a)
{
    if (!a) return fail(GSR_E_INVALID, "gsr_backward_packed: NULL argument%s", "");
    return gsr_backward(a->P, a->D, a->M, a->R, a->background, a->width, a->height, a->means3D, a->shs, a->colors_precomp, a->alphas, a->scales,
                        a->scale_modifier, a->rotations, a->cov3D_precomp, a->viewmatrix, a->projmatrix, a->campos, a->tan_fovx, a->tan_fovy,
                        a->radii, a->geom_buffer, a->binning_buffer, a->img_buffer, a->dL_dpix, a->dL_ddepths, a->dL_dalphas, a->dL_dmean2D,
                        a->dL_dconic, a->dL_dopacity, a->dL_dcolor, a->dL_dmean3D, a->dL_dcov3D, a->dL_dsh, a->dL_dscale, a->dL_drot, a->debug,
                        a->pose_mode, a->dL_dtau, a->stream);
}

int gsr_tracking_loss(int width, int height, const float* image, const float* depth, const float* opacity,
                      const float* gt_image, const float* gt_depth, const uint8_t* grad_mask, const float* exposure,
                      float opacity_threshold, float depth_weight, int monocular, float* dL_dimage, float* dL_ddepth,
                      float* dL_dalpha, float* out, void* stream)
{
    using namespace gsr;
    const int debug = 0;
    hipStream_t st = (hipStream_t)stream;
    if (width <= 0 || height <= 0) return fail(GSR_E_INVALID, "positive image size required%s", "");
    if (!image || !depth || !opacity || !gt_image || !grad_mask || !exposure || !dL_dimage || !dL_ddepth || !dL_dalpha || !out ||
        (!monocular && !gt_depth))
        return fail(GSR_E_INVALID, "gsr_tracking_loss: NULL pointer%s", "");
    int rc = select_device_of(image);
    if (rc != GSR_OK) return rc;
    HIPCHK(hipMemsetAsync(out, 0, 4 * sizeof(float), st));
    LossArgs la;
    la.guard = LoopGuard{nullptr, nullptr, 0u};
    la.clear_a = nullptr; la.clear_b = nullptr; la.clear_n = 0;
    la.W = width; la.H = height; la.image = image; la.depth = depth; la.opacity = opacity; la.gt_image = gt_image;
    la.gt_depth = gt_depth; la.grad_mask = grad_mask; la.exposure = exposure; la.opacity_thr = opacity_threshold;
    la.depth_w = depth_weight; la.monocular = monocular; la.dL_dimage = dL_dimage; la.dL_ddepth = dL_ddepth;
    la.dL_dalpha = dL_dalpha; la.out = out;
    const int n = width * height;
    const int lblocks = (n + GSR_BLOCK - 1) / GSR_BLOCK;
    hipLaunchKernelGGL(k_tracking_loss, dim3(lblocks < 256 ? lblocks : 256), dim3(GSR_BLOCK), 0, st, la);
    LAUNCHCHK("k_tracking_loss");
    return 0;
}

int gsr_pose_init(float* pose_state, const float* projmatrix_raw, void* stream)
{
    const int debug = 0;
    hipStream_t st = (hipStream_t)stream;
    if (!pose_state || !projmatrix_raw) return fail(GSR_E_INVALID, "gsr_pose_init: NULL pointer%s", "");
    int rc = select_device_of(pose_state);
    if (rc != GSR_OK) return rc;
    hipLaunchKernelGGL(gsr::k_pose_init, dim3(1), dim3(64), 0, st, pose_state, projmatrix_raw);
    LAUNCHCHK("k_pose_init");
    return 0;
}

int gsr_pose_step(float* pose_state, const float* dL_dtau, const float* loss_out, const float* projmatrix_raw, float lr,
                  float converged_threshold, void* stream)
{
    const int debug = 0;
    hipStream_t st = (hipStream_t)stream;
    if (!pose_state || !dL_dtau || !loss_out || !projmatrix_raw) return fail(GSR_E_INVALID, "gsr_pose_step: NULL pointer%s", "");
    int rc = select_device_of(pose_state);
    if (rc != GSR_OK) return rc;
    gsr::PoseStepArgs q = {};
    q.st = pose_state; q.dL_dtau = dL_dtau; q.loss_out = loss_out; q.proj_raw = projmatrix_raw; q.lr = lr; q.conv_thr = converged_threshold;
    hipLaunchKernelGGL(gsr::k_pose_step, dim3(1), dim3(64), 0, st, q, gsr::LoopGuard{nullptr, nullptr, 0u});
    LAUNCHCHK("k_pose_step");
    return 0;
}

namespace {
// resize callback wrapper that only calls the user's callback when the workspace must grow
struct CachedBuf { gsr_resize_fn fn; void* ctx; void* ptr; size_t cap; };
void* cached_resize(void* c, size_t bytes)
{
    CachedBuf* b = static_cast<CachedBuf*>(c);
    if (bytes > b->cap || !b->ptr) {
        const size_t want = bytes + bytes / 4;       // head-room: R changes a little from iteration to iteration
        b->ptr = b->fn(b->ctx, want);
        b->cap = b->ptr ? want : 0;
    }
    return b->ptr;
}
}  // namespace

static std::atomic<int> g_refine_calls{0};      // gsr_refine calls in flight in this process (wait_status naps only when there are several)
int gsr_refine(const gsr_refine_args* a, int* iters_done, int* converged)
{
    using namespace gsr;
    struct InFlight { InFlight() { g_refine_calls.fetch_add(1, std::memory_order_relaxed); } ~InFlight() { g_refine_calls.fetch_sub(1, std::memory_order_relaxed); } } in_flight;
    if (!a || !iters_done || !converged) return fail(GSR_E_INVALID, "gsr_refine: NULL argument%s", "");
    if (!a->pose_state || !a->projmatrix_raw || !a->gt_image || !a->grad_mask || !a->dL_dimage || !a->dL_ddepth ||
        !a->dL_dalpha || !a->dL_dtau || !a->loss_out || !a->n_touched)
        return fail(GSR_E_INVALID, "gsr_refine: a required pointer is NULL%s", "");
    if (!a->dL_dmean2D || !a->dL_dconic || !a->dL_dopacity || !a->dL_dcolor)
        return fail(GSR_E_INVALID, "gsr_refine: dL_dmean2D/dL_dconic/dL_dopacity/dL_dcolor are required%s", "");
    hipStream_t st = (hipStream_t)a->stream;
    int rc = select_device_of(a->pose_state);
    if (rc != GSR_OK) return rc;
    // Pinned status slots, one per group parity: ONE word each, (sequence number << 4) | overflow << 2 | bound failure << 1 |
    // converged, written by the pose step that closes each kernel group (pose_step_wave).  On an error return kernels that
    // write these slots may still be in flight: the stream is drained before the slots go back to the pool, so that the next
    // holder never sees a stale word.
    struct CtxLease {
        LoopCtx* c; hipStream_t st; bool clean = false;
        explicit CtxLease(hipStream_t s_) : c(loop_ctx_acquire()), st(s_) {}
        ~CtxLease()
        {
            if (!c) return;
            if (!clean) { (void)hipStreamSynchronize(st); (void)hipGetLastError(); }
            loop_ctx_release(c);
        }
    } ctx_lease(st);
    if (!ctx_lease.c) return fail(GSR_E_HIP, "gsr_refine: could not create the pinned status slots%s", "");
    float* h_status = ctx_lease.c->h_status;
    auto slot_of = [&](int g) { return reinterpret_cast<volatile uint32_t*>(h_status + 8 * (g & 1)); };
    *slot_of(0) = 0u; *slot_of(1) = 0u;
    *reinterpret_cast<volatile uint32_t*>(h_status + 6) = 0u;          // (blocks the most recent split-tile launch list holds: see seg_grid)
    // Waits until the pose step of group `g` has published its status word (sequence bits == g + 1) and returns it in `word`.
    // The kernel writes the slot itself, so there is no copy and no event.  A few thousand polls cover the common case (the
    // status is at most one group away); after that the thread yields between polls -- with several frames in flight per GPU
    // and eight GPUs per node, dozens of these loops share the host's cores -- and the stream is queried now and then so that
    // a failed launch cannot turn this into an endless wait.
    // How long the last group took from status to status.  One frame alone: ~0.15 ms, and the host must have the group after next
    // enqueued within that time -- it spins.  Many frames sharing the GPU (a thread per frame): milliseconds per group, and sixteen
    // spinning threads are sixteen busy cores for nothing (a container with a CPU quota then throttles the whole process, which showed
    // as frames falling behind their neighbours by a third): the thread sleeps between polls, an eighth of a group at a time.
    double group_us = 0.0;
    auto t_last = std::chrono::steady_clock::now();
    auto wait_status = [&](int g, uint32_t& word) -> int {
        volatile uint32_t* w = slot_of(g);
#ifndef GSR_POLL_NAP
#define GSR_POLL_NAP 1
#endif
        // (napping is for MANY frames sharing the GPU: only with other gsr_refine calls in flight in this process -- a single frame whose
        // groups take 0.6 ms (S-3M-cam at 1024x576 on complete lists) would otherwise flip between spinning and napping)
        const long nap_us = (GSR_POLL_NAP && group_us > 600.0 && g_refine_calls.load(std::memory_order_relaxed) > 1) ? (long)std::min(250.0, group_us / 8.0) : 0;
        for (unsigned spins = 0;; spins++) {
            const uint32_t v = *w;
            if ((v >> 4) == (uint32_t)(g + 1)) {
                word = v;
                const auto now = std::chrono::steady_clock::now();
                const double dt = std::chrono::duration<double, std::micro>(now - t_last).count();
                t_last = now;
                group_us = (group_us == 0.0) ? dt : 0.75 * group_us + 0.25 * dt;
                return 0;
            }
            if (nap_us > 0 && spins >= 64u) std::this_thread::sleep_for(std::chrono::microseconds(nap_us));
            else if (spins > 4096u) std::this_thread::yield();
            if ((nap_us > 0 && (spins & 0x3Fu) == 0x3Fu) || (spins & 0x3FFFu) == 0x3FFFu) {
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) {          // everything enqueued has run: one last look, then give up
                    const uint32_t v2 = *w;
                    if ((v2 >> 4) == (uint32_t)(g + 1)) { word = v2; return 0; }
                    return fail(GSR_E_HIP, "gsr_refine: the iteration finished without publishing its status%s", "");
                }
                if (q != hipErrorNotReady) return fail(GSR_E_HIP, "gsr_refine: %s", hipGetErrorString(q));
            }
        }
    };
    CachedBuf gb{a->geometry_buffer, a->geometry_ctx, nullptr, 0}, bb{a->binning_buffer, a->binning_ctx, nullptr, 0},
        ib{a->image_buffer, a->image_ctx, nullptr, 0};
    float* ps = a->pose_state;
    const bool host_mirror = a->pose_state_host != nullptr && a->init_R != nullptr;
    *iters_done = 0;
    *converged = 0;
    uint32_t* poison = reinterpret_cast<uint32_t*>(ps + GSR_PS_POISON);
    PassCtx cx;                    // what every forward / backward of this call runs with; spec / cov_cache / lean change per iteration
    cx.native_loop = true;
    cx.guard.poison = poison;
    cx.guard.conv = a->stop_on_converged ? ps + GSR_PS_CONV : nullptr;
    cx.flags = a->flags;
    cx.det = (a->flags & GSR_REFINE_DETERMINISTIC) != 0;
    if (a->lean_min_P > 0) cx.lean_min_P = a->lean_min_P;
    // heavy tiles split across workgroups (gsr::SegCtl): speculative loops, not under the deterministic option (a split tile's sums round
    // differently from the unsplit walk's, and that option promises the same bits whatever the lists looked like)
    cx.seg = a->speculative && !cx.det && !(a->flags & GSR_REFINE_NO_SPLIT) &&
             seg_budget_of(((a->width + GSR_TILE - 1) / GSR_TILE) * ((a->height + GSR_TILE - 1) / GSR_TILE)) > 0;
    bool seg_used = false, seg_built = false;
    int seg_next_grid = 0;
    cx.seg_used = &seg_used;
    cx.seg_host_total = reinterpret_cast<uint32_t*>(h_status + 6);
    int n_lean = 0;
    cx.n_lean = &n_lean;
    // warm start: the previous call on this workspace left its last bounds in buffer warm_buf (0 / 1); iteration 0 must
    // READ that buffer, i.e. write the other one
    // (bits 8+ of the word: the adaptive margin the previous call ended with, in 1e-4 -- a 20-iteration call would otherwise spend all
    // its iterations tightening it again: 6 800 it/s at m = 0.02 against 7 030 at 0.01 on S-1M-640)
    const int warm_word = (a->speculative && a->warm_state) ? *a->warm_state : 0;
    const int warm_buf = ((warm_word & 0xFF) == 1 || (warm_word & 0xFF) == 2) ? (warm_word & 0xFF) - 1 : -1;
#ifndef GSR_CARRY_MARGIN
#define GSR_CARRY_MARGIN 1
#endif
    const float warm_margin = GSR_CARRY_MARGIN ? (float)((warm_word >> 8) & 0xFFFF) * 1e-4f : 0.f;
    const int poff = (warm_buf >= 0) ? (warm_buf ^ 1) : 0;
    int n_fallbacks = 0, last_R = 0, fail_streak = 0, spec_resume = 0;
    bool last_local = false;
    // What the previous call on these buffers left behind (see gsr_refine_args.carry_state); dropped until this call succeeds.
    const int carried = a->carry_state ? *a->carry_state : 0;
    if (a->carry_state) *a->carry_state = 0;
    {   // gradient tensors: zero-filled once, then maintained row by row (PreBwdArgs::dirty) -- across calls too when the caller
        // vouches for them (carried bit 0)
        const size_t Pn = (size_t)a->P;
        Geom gg;
        char* gptr = (char*)cached_resize(&gb, carve_geom(nullptr, a->P, gg, cx.det, true));
        if (!gptr) return fail(GSR_E_ALLOC, "geometry buffer callback returned NULL%s", "");
        carve_geom(gptr, a->P, gg, cx.det, true);
        if (!(carried & 1)) {
            HIPCHK(hipMemsetAsync(a->dL_dmean2D, 0, Pn * 3 * sizeof(float), st));
            HIPCHK(hipMemsetAsync(a->dL_dconic, 0, Pn * 4 * sizeof(float), st));
            HIPCHK(hipMemsetAsync(a->dL_dopacity, 0, Pn * sizeof(float), st));
            HIPCHK(hipMemsetAsync(a->dL_dcolor, 0, Pn * 3 * sizeof(float), st));
            if (a->dL_dmean3D) HIPCHK(hipMemsetAsync(a->dL_dmean3D, 0, Pn * 3 * sizeof(float), st));
            if (a->dL_dcov3D) HIPCHK(hipMemsetAsync(a->dL_dcov3D, 0, Pn * 6 * sizeof(float), st));
            if (a->dL_dsh && a->M > 0) HIPCHK(hipMemsetAsync(a->dL_dsh, 0, Pn * a->M * 3 * sizeof(float), st));
            if (a->dL_dscale) HIPCHK(hipMemsetAsync(a->dL_dscale, 0, Pn * 3 * sizeof(float), st));
            if (a->dL_drot) HIPCHK(hipMemsetAsync(a->dL_drot, 0, Pn * 4 * sizeof(float), st));
            // K7's accumulator records: cleared once here, afterwards K8 clears every record it consumes
            for (int k = 0; k < 2; k++) {
                HIPCHK(hipMemsetAsync(gg.acc2[k], 0, Pn * GSR_ACC_STRIDE * (cx.det ? 2 * sizeof(long long) : sizeof(float)), st));
                HIPCHK(hipMemsetAsync(gg.aflag2[k], 0, 2 * Pn, st));
            }
            HIPCHK(hipMemsetAsync(gg.dirty, 0, Pn, st));
        }
        PoseLoadArgs pl{};
        if (a->init_R) {
            if (!a->init_T || !a->init_exposure_a || !a->init_exposure_b) return fail(GSR_E_INVALID, "gsr_refine: init_R, init_T, init_exposure_a/b go together%s", "");
            pl = PoseLoadArgs{ps, a->init_R, a->init_T, a->init_exposure_a, a->init_exposure_b, a->projmatrix_raw,
                              a->pose_state_host ? h_status + 16 : (float*)nullptr};      // (loaded by k_refine_init's last workgroup below)
        }
        cx.rows = GradRows{a->dL_dmean2D, a->dL_dconic, a->dL_dopacity, a->dL_dcolor, a->dL_dmean3D, a->dL_dcov3D, a->dL_dsh, a->dL_dscale, a->dL_drot, a->M};
        // image workspace: flags, cursors and both bound buffers start from zero; afterwards the kernels keep them so
        Img im0;
        char* iptr = (char*)cached_resize(&ib, carve_img(nullptr, a->width, a->height, im0, cx.seg));
        if (!iptr) return fail(GSR_E_ALLOC, "image buffer callback returned NULL%s", "");
        carve_img(iptr, a->width, a->height, im0, cx.seg);
        if (!(carried & 1)) HIPCHK(hipMemsetAsync(a->dL_dalpha, 0, (size_t)a->width * a->height * sizeof(float), st));      // no gradient flows into opacity; nothing writes it afterwards
        {   // the small arrays, one launch (k_refine_init)
            ClearRanges cr = {};
            int k = 0;
            auto add = [&](void* ptr, size_t words) { cr.p[k] = static_cast<uint32_t*>(ptr); cr.n[k] = (uint32_t)words; k++; };
            add(ps + GSR_PS_CONV, 5);                                                   // converged, loss, |tau|, poison, ticket
            add(gg.tau_acc, 2 * 16 * GSR_TAU_SLOTS);                                    // (doubles) then kept clean by the pose step
            add(gg.surv2[0].n, (size_t)GSR_SURV_LISTS * GSR_SURV_CSTRIDE);              // ... and the work-list counters by the chain-rule kernel
            add(gg.surv2[1].n, (size_t)GSR_SURV_LISTS * GSR_SURV_CSTRIDE);
            add(gg.final_done, 1);
            add(im0.fail, im0.clear_words);
            add(im0.loss_shards, GSR_LOSS_SHARDS * 16);
            add(im0.tile_work[0], (size_t)(im0.tile_work[1] - im0.tile_work[0]) * 2);
            // (a warm start keeps the bounds the previous call recorded in buffer `warm_buf`; zbc[0], zbc[1] are carved 256 B apart at least)
            if (warm_buf != 0) add(im0.zbc[0], (size_t)im0.nsb);
            if (warm_buf != 1) add(im0.zbc[1], (size_t)im0.nsb);
            add(a->loss_out, 4);
            // (holds are per call.  Keeping them across warm-started calls and holding a tile after its FIRST failure were both tried in
            // round 5 for S-room-640's silhouette tiles -- 30 -> 21 failed forwards per 50-iteration call -- and cost S-1M-640-object dearly:
            // a held tile is binned completely, and in front of a dense object that overflows its bin and sends the rest of the call
            // through count -> scan -> emit; what fixed the room is the widening of bounds at depth discontinuities, dilate_bounds)
            add(im0.tile_hold, (size_t)(im0.tile_work[1] - im0.tile_work[0]));
            // (split tiles: the per-block count / publication words, the per-tile tickets and hold counters -- carved back to back)
            if (im0.seg_budget > 0) add(im0.seg_cnt, (size_t)(reinterpret_cast<uint32_t*>(im0.seg_rec) - im0.seg_cnt));
            static_assert(sizeof(cr.p) / sizeof(cr.p[0]) >= 13, "ClearRanges too small");
            hipLaunchKernelGGL(k_refine_init, dim3(32 + (pl.st != nullptr ? 1 : 0)), dim3(GSR_BLOCK), 0, st, cr, pl);
            { const int debug = 0; LAUNCHCHK("k_refine_init"); }
        }
        cx.balance = !(a->flags & GSR_REFINE_NO_BALANCE);
        cx.floss.gt_image = a->gt_image; cx.floss.gt_depth = a->gt_depth; cx.floss.grad_mask = a->grad_mask;
        cx.floss.exposure = ps + GSR_PS_PARAM + 6; cx.floss.opacity_thr = a->opacity_threshold; cx.floss.depth_w = a->depth_weight;
        cx.floss.monocular = a->monocular; cx.floss.dL_dimage = a->dL_dimage; cx.floss.dL_ddepth = a->dL_ddepth;
        cx.floss.out = im0.loss_shards; cx.floss.conv = a->stop_on_converged ? ps + GSR_PS_CONV : nullptr;
        cx.floss.det = cx.det ? 1 : 0;
    }
    bool cov_cached = (carried & 2) != 0;      // the first forward stores every Gaussian's 3D covariance, the others (and, vouched for, later calls) reuse it
    const int debug = 0;
    auto par = [&](int it) { return (it + poff) & 1; };      // which of the two bound buffers iteration `it` WRITES
    // Margin of the speculative bounds.  Given by the caller: fixed.  Otherwise adaptive: bound = (1 + m) z + m metres
    // with m between 0.01 and 0.05 -- tightened by a fifth after eight verified iterations in a row, doubled when a
    // speculation fails (tight bounds mean shorter lists; a failure costs one forward with complete lists).
    const bool adaptive_margin = !(a->bound_margin_mul > 0.f);
    float margin_m = (warm_buf >= 0 && warm_margin >= 0.01f && warm_margin <= 0.05f) ? warm_margin : 0.02f;
    int margin_streak = 0;
    if (!adaptive_margin) { cx.spec.mul = a->bound_margin_mul; cx.spec.add = a->bound_margin_add; }

    // One kernel GROUP = forward (with the tracking loss in its compositing epilogue), backward, chain rule with Adam + update_pose
    // in its last workgroup -- all enqueued without waiting for the device (a forward with complete lists still reads its
    // instance count back, as the reference does).  The pose step that closes a group publishes the group's status word.
    // Groups are numbered g = 0, 1, ...; group g writes bounds buffer par(g), reads par(g) ^ 1, carries tag g + 1.
    // A group whose speculative forward fails its verification skips its own loss / backward / pose step ON THE DEVICE and the
    // next group -- enqueued one ahead, as always -- runs as the retry of the same iteration with the bounds the failed forward
    // recorded (LoopGuard).  The host only keeps count: a failed group does not advance the iteration number.  It steps in
    // (drain, complete lists, back-off) when a bin overflowed or the retry failed as well.
    Img imv_loop{};                // the image workspace's carving (for the pose step launch)
    int last_enq = -1;            // last group whose forward was enqueued: its bounds are the newest
    bool last_full = false;       // ... binned complete lists into fixed-capacity bins (k_preprocess_bin)
    bool last_counted = false;    // ... and it already counted n_touched
    gsr::PreBwdArgs pb_hist[4] = {};      // the chain-rule launches of the last groups (the final pass below reuses the last stepped one's)
    auto enqueue = [&](int g, int logical, int mode) -> int {
        last_enq = g;
        if (adaptive_margin) {
#ifndef GSR_WARM_MARGIN
#define GSR_WARM_MARGIN 0.05f      // (0.03: 6 of 16 warm-started frames fail their first verification, 0.02: 9 of 16 -- bench.py, K = 20: value 8 830 / 8 670 against 9 170)
#endif
            const float m = (g == 0 && warm_buf >= 0) ? GSR_WARM_MARGIN : margin_m;      // (bounds recorded for another frame: be generous)
            cx.spec.mul = 1.f + m; cx.spec.add = m;
        }
        // Round 6: a split tile's segments WAIT for the lower-numbered blocks of their launch (the products of the ranges in front).  Workgroups
        // of one launch start in order -- per XCD: block b goes to XCD b mod 8, each XCD takes its share when IT has room.  A launch alone on
        // the GPU loads the eight alike; with eight or more refinement calls in flight (sixteen is bench.py's default) they drift apart, a
        // range becomes resident while its predecessor still queues behind other calls' blocks on another XCD, spinning ranges fill the slots
        // their predecessors need, and the bounded waits (seconds) were what ended it: S-room-640 2 811 it/s with four frames in flight, 130
        // with eight, 81 with sixteen; a trained 800 k map 2 359 / 243 / 128 (tools/dbg/inflight_probe.py, trained_probe.py -- the
        // scene_variants leg runs ONE frame and `value` runs a scene that splits nothing, so no bench line saw it).  With more than
        // GSR_SPLIT_MAX_CALLS calls in flight the lists are built without splits (the other frames fill the GPU while a heavy tile's wave
        // walks alone -- what splitting was for); everything else the builder block does (launch order, widened bounds) stays.
#ifndef GSR_SPLIT_MAX_CALLS
#define GSR_SPLIT_MAX_CALLS 4
#endif
        cx.split_ok_now = g_refine_calls.load(std::memory_order_relaxed) <= GSR_SPLIT_MAX_CALLS;
        *slot_of(g) = 0u;      // (nothing in flight writes this slot any more: group g - 2 has been settled)
        cx.spec.mode = mode;
        cx.spec.parity = par(g);
        cx.guard.tag = (uint32_t)(g + 1);
        cx.cov_cache = cov_cached ? 2 : 1;
        cov_cached = true;
        cx.set = g & 1;                  // (work lists, flags, records, splat records: two sets, see Geom)
        cx.pb_out = &pb_hist[g & 3];
        cx.seg_ready = seg_built;      // (the group enqueued before this one left a launch list for it)
        // launch sizes: this group runs the list its predecessor built for `seg_next_grid` blocks; the list it builds itself may use what
        // the most recent finished builder needed (+ an eighth + 64), never less than a block per tile, never more than the workspace holds
        cx.seg_grid = seg_next_grid;
        {
            const int nt_ = ((a->width + GSR_TILE - 1) / GSR_TILE) * ((a->height + GSR_TILE - 1) / GSR_TILE);
            const int budget_ = seg_budget_of(nt_);
            const uint32_t hint = *reinterpret_cast<volatile uint32_t*>(h_status + 6);
            seg_next_grid = (hint == 0u) ? budget_ : std::min(budget_, std::max(nt_, (int)(hint + hint / 8u + 64u)));
            cx.seg_grid_next = seg_next_grid;
        }
#ifndef GSR_HOLD_AFTER
#define GSR_HOLD_AFTER 2u
#endif
        cx.hold_after = (g == 0 && warm_buf >= 0) ? 2u : GSR_HOLD_AFTER;
        seg_built = cx.seg && cx.balance;
        const bool maybe_last = (logical == a->max_iters - 1);
        cx.lean = (mode == 1) && !maybe_last;
        Img imv; carve_img((char*)ib.ptr, a->width, a->height, imv, cx.seg);
        imv_loop = imv;
        // Adam + update_pose run in the chain-rule kernel's last workgroup (no launch of their own); they also finish the fp64
        // dL/dtau reduction, clear the superblock bounds buffer the next group accumulates into and publish the status
        cx.ticket = reinterpret_cast<uint32_t*>(ps + GSR_PS_TICKET);
        cx.fold = PoseStepArgs{};
        cx.fold.st = ps; cx.fold.dL_dtau = a->dL_dtau; cx.fold.dL_dtau_out = a->dL_dtau; cx.fold.loss_out = a->loss_out;
        cx.fold.proj_raw = a->projmatrix_raw; cx.fold.lr = a->lr; cx.fold.conv_thr = a->converged_threshold; cx.fold.loss_zero = a->loss_out;
        cx.fold.host_status = const_cast<uint32_t*>(slot_of(g)); cx.fold.seq = g + 1; cx.fold.loss_shards = imv.loss_shards;
        cx.fold.host_state = host_mirror ? h_status + 16 : nullptr;
        cx.fold.clear_b = (mode != 0) ? imv.zbc[par(g) ^ 1] : nullptr; cx.fold.clear_n = imv.nsb;
        // n_touched is wanted for the LAST forward only (see the end): a group that may be the last counts it itself
        const bool count_touched = maybe_last && a->n_touched != nullptr;
        last_counted = count_touched;
        cx.used_full_bins = &last_full;
        int R = forward_impl(cx, cached_resize, &gb, cached_resize, &bb, cached_resize, &ib, a->P, a->D, a->M, a->background,
                             a->width, a->height, a->means3D, a->shs, a->colors_precomp, a->opacities, a->scales, a->scale_modifier,
                             a->rotations, a->cov3D_precomp, ps + GSR_PS_VIEW, ps + GSR_PS_PROJ, ps + GSR_PS_CAMPOS, a->tan_fovx,
                             a->tan_fovy, 0, a->out_color, a->out_depth, a->out_alpha, a->radii, 0, count_touched ? a->n_touched : nullptr, a->stream);
        if (R < 0) return R;
        last_R = R;
        last_local = (mode == 1) || last_full;      // a forward that bins into per-tile bins does not bring its instance count to the host
        // (the tracking loss was evaluated in the compositing kernel's epilogue: cx.floss)
        int rc2 = backward_impl(cx, a->P, a->D, a->M, R, a->background, a->width, a->height, a->means3D, a->shs, a->colors_precomp, a->out_alpha,
                                a->scales, a->scale_modifier, a->rotations, a->cov3D_precomp, ps + GSR_PS_VIEW, ps + GSR_PS_PROJ,
                                ps + GSR_PS_CAMPOS, a->tan_fovx, a->tan_fovy, a->radii, (char*)gb.ptr, (char*)bb.ptr, (char*)ib.ptr,
                                a->dL_dimage, a->dL_ddepth, a->dL_dalpha, a->dL_dmean2D, a->dL_dconic, a->dL_dopacity, a->dL_dcolor,
                                a->dL_dmean3D, a->dL_dcov3D, a->dL_dsh, a->dL_dscale, a->dL_drot, 0, 1, a->dL_dtau, a->stream);
        if (rc2 < 0) return rc2;
        return 0;
    };
    // Speculative binning: from the second iteration on, instances lying behind what their tile needed in the previous
    // iteration (x margin) are not binned; the compositing kernel verifies the speculation, so results never depend on it.
    auto mode_of = [&](int logical) { return a->speculative ? (((logical == 0 && warm_buf < 0) || logical < spec_resume) ? 2 : 1) : 0; };
    std::vector<int> group_mode;   // mode each group was enqueued with
    int enq = 0, settled_n = 0;    // groups enqueued / whose status the host has seen (in order)
    int succ = 0;                  // iterations completed (groups whose pose step ran)
    int streak = 0;                // failed groups in a row
    int n_host_redo = 0;           // forwards the HOST had to redo with complete lists (everything else was retried on the device)
    bool conv_seen = false;        // (stop_on_converged) an update reported convergence: what follows is the frozen render at the final pose
    int last_step_g = -1;
    bool final_rendered = false;   // ... and that render has been enqueued / verified
    auto enqueue_next = [&](int logical, int mode) -> int {
        const int rc2 = enqueue(enq, logical, mode);
        if (rc2 < 0) return rc2;
        group_mode.push_back(mode);
        enq++;
        return 0;
    };
    auto after_success = [&](int g, uint32_t w) {      // bookkeeping for a group that passed its verification
        streak = 0;
        if (!conv_seen) last_step_g = g;                // (the last group whose pose step ran: its records become the gradient rows)
        if (group_mode[g] == 1) {
            fail_streak = 0;
            if (adaptive_margin && ++margin_streak >= 8) { margin_m = fmaxf(0.01f, margin_m * 0.8f); margin_streak = 0; }
        }
        if (conv_seen) return;                          // (a frozen group: the render at the final pose)
        succ++;
        if (a->stop_on_converged && (w & 1u)) { conv_seen = true; *converged = 1; }
    };
    while (true) {
        const int inflight = enq - settled_n;
        // keep two groups in flight -- the one waited for and one behind it -- while iterations remain
        if (!conv_seen && inflight < 2 && succ + inflight < a->max_iters) {
            rc = enqueue_next(succ + inflight, mode_of(succ + inflight));
            if (rc < 0) return rc;
            continue;
        }
        if (inflight == 0) {
            // reference: `if converged: break` -- the caller gets the render at the final pose: one more (frozen) forward, unless
            // the iterations are used up (then, like the reference, the last loop body's render)
            if (conv_seen && succ < a->max_iters && !final_rendered) {
                rc = enqueue_next(succ, mode_of(succ));
                if (rc < 0) return rc;
                final_rendered = true;
                continue;
            }
            break;
        }
        const int g = settled_n;
        uint32_t w = 0;
        { const int wrc = wait_status(g, w); if (wrc < 0) return wrc; }
        settled_n++;
        if (conv_seen) final_rendered = true;           // (the group behind a converged one is that frozen forward)
        if ((w & 6u) == 0u) { after_success(g, w); continue; }
        // ---- group g failed its verification: nothing of it counts; the device is already retrying with the next group
        n_fallbacks++;
        streak++;
        const bool overflow = (w & 4u) != 0u;
        if (a->flags & GSR_REFINE_LOG_REDO) {
            uint32_t who[2] = {0u, 0u}, holdw = 0u;
            Img imd; carve_img((char*)ib.ptr, a->width, a->height, imd, cx.seg);
            float zbd[2] = {0.f, 0.f};
            if (overflow) {          // (diagnostics only: which tile, how many entries -- a blocking read)
                (void)hipStreamSynchronize(st);
                (void)hipMemcpy(who, ps + GSR_PS_POISON + 2, sizeof(who), hipMemcpyDeviceToHost);
                const int ntd = ((a->width + GSR_TILE - 1) / GSR_TILE) * ((a->height + GSR_TILE - 1) / GSR_TILE);
                if ((int)who[0] < ntd) { (void)hipMemcpy(&zbd[0], imd.zb[0] + who[0], 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&zbd[1], imd.zb[1] + who[0], 4, hipMemcpyDeviceToHost);
                                         (void)hipMemcpy(&holdw, imd.tile_hold + who[0], 4, hipMemcpyDeviceToHost); }
            }
            fprintf(stderr, "[gsr] group %d (mode %d, margin %.3f): speculation failed (%s)%s", g, group_mode[g], margin_m, overflow ? "bin overflow" : "unsaturated tile behind a finite bound",
                    (streak < 2 && !overflow && !conv_seen) ? ", retried on the device" : ", host steps in");
            if (overflow) fprintf(stderr, " [tile %u: %u entries; its bounds in the two buffers: %g / %g; failures so far %u, held for %u more forwards]", who[0], who[1], zbd[0], zbd[1], holdw >> 8, holdw & 0xFFu);
            fprintf(stderr, "\n");
        }
        if (adaptive_margin) { margin_m = fminf(0.05f, margin_m * 2.f); margin_streak = 0; }
        if (overflow) cx.exact_bins = true;             // (complete lists go through count -> scan -> emit for the rest of this call)
        if (!conv_seen && !overflow && streak < 2) continue;
        // ---- the host steps in: drain what is in flight (a retry that may well have succeeded), then complete lists if need be
        HIPCHK(hipStreamSynchronize(st));
        bool resolved = false;
        while (settled_n < enq) {
            const int g2 = settled_n;
            uint32_t w2 = 0;
            { const int wrc = wait_status(g2, w2); if (wrc < 0) return wrc; }
            settled_n++;
            if (w2 & 6u) { n_fallbacks++; resolved = false; if (w2 & 4u) cx.exact_bins = true; }
            else { after_success(g2, w2); resolved = true; }
        }
        if (!resolved) {
            n_host_redo++;
            // Complete lists.  Up to kFullBinMaxTiles tiles they go into fixed-capacity bins first (k_preprocess_bin), and a tile
            // whose complete list is longer than its bin reports an overflow: only then -- once -- count -> scan -> emit, which
            // cannot fail.  (A warm-started call reaches this point without ever having run a complete-list forward.)
            uint32_t w3 = 0;
            int g3 = 0;
            for (int attempt = 0;; attempt++) {
                g3 = enq;
                rc = enqueue_next(succ, 2);
                if (rc < 0) return rc;
                { const int wrc = wait_status(g3, w3); if (wrc < 0) return wrc; }
                settled_n++;
                if ((w3 & 6u) == 0u) break;
                n_fallbacks++;
                if ((w3 & 4u) && !cx.exact_bins && attempt == 0) { cx.exact_bins = true; continue; }
                return fail(GSR_E_HIP, "gsr_refine: a forward with complete lists failed its verification%s", "");
            }
            after_success(g3, w3);
            // back off: a scene whose lists stay long after culling would otherwise pay for failed forwards again and again
            fail_streak++;
            if (fail_streak >= 2) spec_resume = succ + (1 << (fail_streak < 6 ? fail_streak : 6));
        }
    }
    *iters_done = succ;
    if (last_step_g >= 0 && !(a->flags & GSR_REFINE_GRADS_EVERY_ITERATION)) {
        // The gradients of the Gaussians' own parameters: rows written once, from the records of the last iteration whose pose step
        // ran (PreBwdArgs::role).  When the loop converged with a frozen forward behind that iteration, that group's launch has done it
        // already (and says so in *final_done).
        gsr::PreBwdArgs f = pb_hist[last_step_g & 3];
        f.role = 2; f.ticket = nullptr; f.guard = gsr::LoopGuard{nullptr, nullptr, 0u};
        f.o_surv = f.surv; f.o_aflag = f.aflag; f.o_acc = f.acc; f.o_rec = f.rec; f.o_clamped = f.clamped;
        ProfScope pfs(K_PREPROCESS_BWD, st);
        const int k8_grid = surv_grid(a->P, GSR_K8_RESIDENT);
        if (cx.det) hipLaunchKernelGGL(k_preprocess_bwd<true>, dim3(k8_grid), dim3(64), 0, st, f);
        else hipLaunchKernelGGL(k_preprocess_bwd<false>, dim3(k8_grid), dim3(64), 0, st, f);
        LAUNCHCHK("k_preprocess_bwd (final pass)");
    }
    if (last_enq >= 0 && !last_counted && a->n_touched && gb.ptr && bb.ptr && ib.ptr) {
        // n_touched (fifth output of the pose package's forward) is only wanted for the LAST forward, and counting it
        // costs every iteration's compositing kernel an eighth of its instructions: the loop runs the variant
        // without it and the lists of the last forward are walked once more here, with the counters: pixel p over positions
        // 1 .. n_contrib[p] of its tile's list, the same blend tests, nothing written but the counters (RECOUNT in k_render_fwd --
        // round 6: the last forward may have split tiles, whose images this pass must not rewrite with another rounding and whose
        // lists are only valid up to the tile's deepest contributor).
        const int gx = (a->width + GSR_TILE - 1) / GSR_TILE, gy = (a->height + GSR_TILE - 1) / GSR_TILE;
        Geom g; carve_geom((char*)gb.ptr, a->P, g, cx.det, true);
        g.select_set(last_enq & 1);
        Img im; carve_img((char*)ib.ptr, a->width, a->height, im);
        HIPCHK(hipMemsetAsync(a->n_touched, 0, (size_t)a->P * sizeof(int), st));
        hipLaunchKernelGGL((k_render_fwd<true, GSR_LIST_SORTED>), dim3(gx * gy), dim3(GSR_BLOCK), 0, st, im.ranges,
                           reinterpret_cast<uint32_t*>(bb.ptr), (const unsigned long long*)nullptr, (uint32_t*)nullptr, a->width,
                           a->height, gx, gx * gy, (const float*)g.rec, a->background, a->out_color, a->out_depth, a->out_alpha, im.n_contrib,
                           a->n_touched, (float*)nullptr, (const float*)nullptr, im.fail, 1.f, 0.f, (float*)nullptr,
                           im.sbx, FusedLoss{}, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0, LazySH{}, 0u, (uint32_t*)nullptr, (a->P < (1 << 28)) ? 1 : 0, (uint32_t*)nullptr, SegCtl{});
        LAUNCHCHK("k_render_fwd (n_touched)");
    }
    // (with init_* the kernels have kept a mirror of the state in pinned memory -- the pose load of k_refine_init and every pose step that ran: no copy)
    if (a->pose_state_host && !host_mirror) HIPCHK(hipMemcpyAsync(h_status + 16, ps, GSR_POSE_STATE_FLOATS * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (a->pose_state_host) memcpy(a->pose_state_host, h_status + 16, GSR_POSE_STATE_FLOATS * sizeof(float));
    ctx_lease.clean = true;
    if (a->carry_state) *a->carry_state = (cov_cached ? 2 : 0) | 1;
    if (a->warm_state)
        *a->warm_state = (a->speculative && last_enq >= 0) ? ((par(last_enq) + 1) | (adaptive_margin ? ((int)lrintf(margin_m * 1e4f) << 8) : 0)) : 0;
    if (a->stats_out) {
        const bool want_count = a->stats_out[1] != -1;
        if (!want_count) last_R = -1;
        if (want_count && last_local && ib.ptr) {      // instances binned by the last forward = sum of the tile list lengths
            Img imv; carve_img((char*)ib.ptr, a->width, a->height, imv);
            const int nt = ((a->width + GSR_TILE - 1) / GSR_TILE) * ((a->height + GSR_TILE - 1) / GSR_TILE);
            long long sum = 0;
            if (last_full) {          // (complete lists are ordered lazily: the ranges only cover what was ordered; the full counts sit in tile_count)
                std::vector<uint32_t> tc((size_t)nt);
                HIPCHK(hipMemcpyAsync(tc.data(), imv.tile_count, tc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                for (int i = 0; i < nt; i++) sum += (long long)tc[i];
            } else {
                std::vector<uint2> rg((size_t)nt);
                HIPCHK(hipMemcpyAsync(rg.data(), imv.ranges, rg.size() * sizeof(uint2), hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                for (int i = 0; i < nt; i++) sum += (long long)(rg[i].y - rg[i].x);
            }
            last_R = (int)sum;
        }
        a->stats_out[0] = n_fallbacks; a->stats_out[1] = last_R; a->stats_out[2] = n_lean; a->stats_out[3] = n_host_redo;
    }
    return 0;
}

size_t gsr_debug_lam_offset(int P)
{
    Geom g;
    char* base = reinterpret_cast<char*>((uintptr_t)4096);      // (never dereferenced: carve_geom only does pointer arithmetic)
    carve_geom(base, P, g);
    return (size_t)(reinterpret_cast<char*>(g.lam) - base);
}

int gsr_debug_tile_order(const unsigned* work, unsigned* order, int ntiles, void* stream)
{
    using namespace gsr;
    const int debug = 0;
    if (!work || !order || ntiles < 1 || ntiles > GSR_STATELESS_BALANCE_MAX_TILES) return fail(GSR_E_INVALID, "gsr_debug_tile_order: bad arguments%s", "");
    int rc = select_device_of(work);
    if (rc != GSR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_backward_prologue<false>, dim3(1), dim3(GSR_TILE_ORDER_THREADS), 0, st, SurvLists{nullptr, nullptr, 0u}, (float*)nullptr, (uint8_t*)nullptr, 0,
                       0, (const uint32_t*)work, (uint32_t*)order, ntiles);
    LAUNCHCHK("k_backward_prologue (order only)");
    return 0;
}

int gsr_debug_seg_stats(const gsr_refine_args* a, long long out[4])
{
    using namespace gsr;
    if (!a || !out) return fail(GSR_E_INVALID, "gsr_debug_seg_stats: NULL argument%s", "");
    if (!a->image_buffer || !a->pose_state || a->width <= 0 || a->height <= 0) return fail(GSR_E_INVALID, "gsr_debug_seg_stats: a required pointer is NULL%s", "");
    hipStream_t st = (hipStream_t)a->stream;
    int rc = select_device_of(a->pose_state);
    if (rc != GSR_OK) return rc;
    Img im;
    out[0] = out[1] = out[2] = out[3] = 0;
    // (only a call that RAN with the split arrays has them in its image workspace -- the same condition gsr_refine uses;
    // asking the caller's callback for the larger carving after a deterministic / GSR_REFINE_NO_SPLIT / non-speculative call would grow
    // or move a live workspace and read a launch list nobody wrote)
    const bool det = (a->flags & GSR_REFINE_DETERMINISTIC) != 0;
    if (!(a->speculative && !det && !(a->flags & GSR_REFINE_NO_SPLIT)) ||
        seg_budget_of(((a->width + GSR_TILE - 1) / GSR_TILE) * ((a->height + GSR_TILE - 1) / GSR_TILE)) <= 0) return 0;
    char* iptr = (char*)a->image_buffer(a->image_ctx, carve_img(nullptr, a->width, a->height, im, true));
    if (!iptr) return fail(GSR_E_ALLOC, "image buffer callback returned NULL%s", "");
    carve_img(iptr, a->width, a->height, im, true);
    out[3] = im.seg_budget;
    if (im.seg_budget <= 0) return 0;
    // (the list the call's last speculative group walked: the one its predecessor built -- the parity the warm-state word remembers)
    const int last_par = a->warm_state ? ((*a->warm_state & 0xFF) - 1) : -1;
    if (last_par < 0 || last_par > 1) return fail(GSR_E_INVALID, "gsr_debug_seg_stats: needs the warm_state a speculative gsr_refine on these workspaces left%s", "");
    std::vector<uint32_t> list((size_t)im.seg_budget);
    HIPCHK(hipMemcpyAsync(list.data(), im.seg_list[last_par ^ 1], list.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (uint32_t e : list) {
        if (e == 0xFFFFFFFFu) continue;
        out[0]++;
        const uint32_t sgi = (e >> 16) & 0xFFu, k = e >> 24;
        if (k > 1u && sgi == 0u) out[1]++;
        if ((long long)k > out[2]) out[2] = k;
    }
    return 0;
}

int gsr_debug_lean_check(const gsr_refine_args* a, long long out[5])
{
    using namespace gsr;
    const int debug = 0;
    if (!a || !out) return fail(GSR_E_INVALID, "gsr_debug_lean_check: NULL argument%s", "");
    if (!a->warm_state || (*a->warm_state & 0xFF) < 1 || (*a->warm_state & 0xFF) > 2 || !a->carry_state || !(*a->carry_state & 2))
        return fail(GSR_E_INVALID, "gsr_debug_lean_check: needs the warm_state / carry_state a speculative gsr_refine on these workspaces left%s", "");
    if (!a->pose_state || !a->means3D || !a->opacities || !a->geometry_buffer || !a->image_buffer || a->P <= 0)
        return fail(GSR_E_INVALID, "gsr_debug_lean_check: a required pointer is NULL%s", "");
    hipStream_t st = (hipStream_t)a->stream;
    int rc = select_device_of(a->pose_state);
    if (rc != GSR_OK) return rc;
    Geom g;
    char* gptr = (char*)a->geometry_buffer(a->geometry_ctx, carve_geom(nullptr, a->P, g));
    if (!gptr) return fail(GSR_E_ALLOC, "geometry buffer callback returned NULL%s", "");
    carve_geom(gptr, a->P, g);
    Img im;
    char* iptr = (char*)a->image_buffer(a->image_ctx, carve_img(nullptr, a->width, a->height, im));
    if (!iptr) return fail(GSR_E_ALLOC, "image buffer callback returned NULL%s", "");
    carve_img(iptr, a->width, a->height, im);
    const int buf = (*a->warm_state & 0xFF) - 1;          // the bounds the last forward recorded: what the next iteration would bin with
    const float* ps = a->pose_state;
    PreArgs pa = {};
    pa.P = a->P; pa.W = a->width; pa.H = a->height;
    pa.gx = (a->width + GSR_TILE - 1) / GSR_TILE; pa.gy = (a->height + GSR_TILE - 1) / GSR_TILE;
    pa.means = a->means3D; pa.opac = a->opacities; pa.mod = a->scale_modifier;
    pa.view = ps + GSR_PS_VIEW; pa.proj = ps + GSR_PS_PROJ; pa.campos = ps + GSR_PS_CAMPOS;
    pa.tanx = a->tan_fovx; pa.tany = a->tan_fovy;
    pa.fx = a->width / (2.0f * a->tan_fovx); pa.fy = a->height / (2.0f * a->tan_fovy);
    pa.cov3D_pre = g.cov3D; pa.lam = g.lam;
    pa.zb = im.zb[buf]; pa.zbc = im.zbc[buf]; pa.sbx = im.sbx; pa.sby = im.sby;
    pa.pyr_tiles = (pa.gx * pa.gy <= GSR_PYR_TILES_MAX) ? 1 : 0;
    pa.zb_mul = 1.05f; pa.zb_add = 0.05f;
    unsigned long long* d = reinterpret_cast<unsigned long long*>(im.loss_shards);      // scratch (gsr_refine clears these words when it starts)
    HIPCHK(hipMemsetAsync(d, 0, 8 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_lean_check, dim3((a->P + GSR_BLOCK - 1) / GSR_BLOCK), dim3(GSR_BLOCK), 0, st, pa, d);
    LAUNCHCHK("k_lean_check");
    unsigned long long h[5] = {0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemsetAsync(d, 0, 8 * sizeof(unsigned long long), st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 4; i++) out[i] = (long long)h[i];
    out[4] = h[4] ? (long long)(0xFFFFFFFFull - h[4]) : -1;
    return 0;
}

namespace {
struct KnnWs { uint32_t* mm; uint32_t* codes; uint32_t* codes_sorted; uint32_t* idx; uint32_t* idx_sorted; float4* sorted; float* boxes;
               char* sort_tmp; size_t sort_bytes; };
size_t carve_knn(char* base, int P, KnnWs& w)
{
    Carver c(base);
    const size_t n = P > 0 ? (size_t)P : 1;
    w.mm = c.take<uint32_t>(8);
    w.codes = c.take<uint32_t>(n); w.codes_sorted = c.take<uint32_t>(n);
    w.idx = c.take<uint32_t>(n); w.idx_sorted = c.take<uint32_t>(n);
    w.sorted = c.take<float4>(n);
    w.boxes = c.take<float>(8 * ((n + GSR_KNN_BOX - 1) / GSR_KNN_BOX));
    w.sort_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, w.sort_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (int)n);
    w.sort_tmp = c.take<char>(w.sort_bytes);
    return c.size();
}
}  // namespace

size_t gsr_knn_bytes(int P) { KnnWs w; return carve_knn(nullptr, P, w); }

// ---- the per-frame gradient mask (camera_utils.py:164-193 + the keypoint boxes of the scripts) ----
namespace {
const size_t kGmHistBytes = 3 * GSR_GM_BINS * sizeof(uint32_t);          // three histograms, then 256 B of result words, then the intensity image
struct GmCarve { uint32_t* hist; float* median; float* intensity; };
int gm_begin(int width, int height, const float* image, gsr_resize_fn workspace, void* workspace_ctx, float* intensity_out, hipStream_t st,
             gsr::GradMaskArgs& a, GmCarve& cv)
{
    const int debug = 0;
    using namespace gsr;
    if (width < 2 || height < 2) return fail(GSR_E_INVALID, "gsr_grad_mask: the reflect padding needs an image of at least 2 x 2 pixels%s", "");
    if ((long long)width * height > 0x7fffffffLL) return fail(GSR_E_INVALID, "gsr_grad_mask: image too large%s", "");
    if (!image || !workspace) return fail(GSR_E_INVALID, "gsr_grad_mask: NULL pointer%s", "");
    int rc = select_device_of(image);
    if (rc != GSR_OK) return rc;
    char* ws = (char*)workspace(workspace_ctx, gsr_grad_mask_bytes(width, height));
    if (!ws) return fail(GSR_E_ALLOC, "workspace callback returned NULL%s", "");
    cv.hist = reinterpret_cast<uint32_t*>(ws);
    cv.median = reinterpret_cast<float*>(ws + kGmHistBytes);
    cv.intensity = intensity_out ? intensity_out : reinterpret_cast<float*>(ws + kGmHistBytes + 256);
    HIPCHK(hipMemsetAsync(ws, 0, kGmHistBytes + 256, st));
    a.W = width; a.H = height; a.image = image; a.intensity = cv.intensity; a.hist = cv.hist;
    a.rank = (uint32_t)(((long long)width * height - 1) / 2);
    a.edge_threshold = 0.f; a.mask = nullptr; a.median_out = cv.median;
    hipLaunchKernelGGL(k_gradmask_intensity, dim3((width + GSR_GM_TW - 1) / GSR_GM_TW, (height + GSR_GM_TH - 1) / GSR_GM_TH), dim3(256), 0, st, a);
    LAUNCHCHK("k_gradmask_intensity");
    return 0;
}
}  // namespace

size_t gsr_grad_mask_bytes(int width, int height)
{
    return kGmHistBytes + 256 + (size_t)(width > 0 ? width : 0) * (size_t)(height > 0 ? height : 0) * sizeof(float);
}

int gsr_grad_mask(int width, int height, const float* image, float edge_threshold, const float* keypoints, int num_keypoints, int box_k,
                  uint8_t* grad_mask, float* intensity_out, float* median_out, gsr_resize_fn workspace, void* workspace_ctx, void* stream)
{
    using namespace gsr;
    const int debug = 0;
    hipStream_t st = (hipStream_t)stream;
    if (!grad_mask) return fail(GSR_E_INVALID, "gsr_grad_mask: NULL pointer%s", "");
    if (num_keypoints < 0 || box_k < 0 || (num_keypoints > 0 && !keypoints)) return fail(GSR_E_INVALID, "gsr_grad_mask: keypoints%s", "");
    GradMaskArgs a; GmCarve cv;
    int rc = gm_begin(width, height, image, workspace, workspace_ctx, intensity_out, st, a, cv);
    if (rc != 0) return rc;
    a.edge_threshold = edge_threshold; a.mask = grad_mask;
    if (median_out) a.median_out = median_out;
    const long long n = (long long)width * height;
    const int blocks = (int)std::min<long long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(k_gradmask_hist<2>, dim3(blocks), dim3(256), 0, st, a);
    LAUNCHCHK("k_gradmask_hist<2>");
    hipLaunchKernelGGL(k_gradmask_hist<3>, dim3(blocks), dim3(256), 0, st, a);
    LAUNCHCHK("k_gradmask_hist<3>");
    hipLaunchKernelGGL(k_gradmask_threshold, dim3(blocks), dim3(256), 0, st, a);
    LAUNCHCHK("k_gradmask_threshold");
    if (num_keypoints > 0) {
        hipLaunchKernelGGL(k_gradmask_boxes, dim3((num_keypoints + 3) / 4), dim3(256), 0, st, width, height, keypoints, num_keypoints, box_k / 2, grad_mask);
        LAUNCHCHK("k_gradmask_boxes");
    }
    return 0;
}

int gsr_grad_mask_replica(int width, int height, const float* image, float edge_threshold, int rows, int cols, float* grad_mask,
                          gsr_resize_fn workspace, void* workspace_ctx, void* stream)
{
    using namespace gsr;
    const int debug = 0;
    hipStream_t st = (hipStream_t)stream;
    if (!grad_mask) return fail(GSR_E_INVALID, "gsr_grad_mask_replica: NULL pointer%s", "");
    if (rows <= 0 || cols <= 0 || height / rows <= 0 || width / cols <= 0)
        return fail(GSR_E_INVALID, "gsr_grad_mask_replica: every block of the rows x cols grid needs at least one pixel (the reference's block.median() raises on an empty block)%s", "");
    GradMaskArgs a; GmCarve cv;
    int rc = gm_begin(width, height, image, workspace, workspace_ctx, nullptr, st, a, cv);
    if (rc != 0) return rc;
    HIPCHK(hipMemcpyAsync(grad_mask, cv.intensity, (size_t)width * height * sizeof(float), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_gradmask_replica, dim3(rows * cols), dim3(256), 0, st, width, height, width / cols, height / rows, cols,
                       (const float*)cv.intensity, edge_threshold, grad_mask);
    LAUNCHCHK("k_gradmask_replica");
    return 0;
}

int gsr_dist2_knn3(int P, const float* points, float* mean_dist2, gsr_resize_fn workspace, void* workspace_ctx, void* stream)
{
    using namespace gsr;
    const int debug = 0;
    hipStream_t st = (hipStream_t)stream;
    if (P < 0) return fail(GSR_E_INVALID, "P >= 0 required%s", "");
    if (P == 0) return 0;
    if (!points || !mean_dist2 || !workspace) return fail(GSR_E_INVALID, "gsr_dist2_knn3: NULL pointer%s", "");
    int rc = select_device_of(points);
    if (rc != GSR_OK) return rc;
    KnnWs w;
    char* ws = (char*)workspace(workspace_ctx, carve_knn(nullptr, P, w));
    if (!ws) return fail(GSR_E_ALLOC, "workspace callback returned NULL%s", "");
    carve_knn(ws, P, w);
    const int pblocks = (P + GSR_BLOCK - 1) / GSR_BLOCK;
    {   // min / max seeded with the origin: encoded 0.0f
        const uint32_t zero_enc = 0x80000000u;
        const uint32_t init[8] = {zero_enc, zero_enc, zero_enc, zero_enc, zero_enc, zero_enc, 0, 0};
        HIPCHK(hipMemcpyAsync(w.mm, init, sizeof(init), hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(k_knn_minmax, dim3(pblocks < 1024 ? pblocks : 1024), dim3(GSR_BLOCK), 0, st, P, points, w.mm);
    LAUNCHCHK("k_knn_minmax");
    hipLaunchKernelGGL(k_knn_morton, dim3(pblocks), dim3(GSR_BLOCK), 0, st, P, points, (const uint32_t*)w.mm, w.codes, w.idx);
    LAUNCHCHK("k_knn_morton");
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(w.sort_tmp, w.sort_bytes, (const uint32_t*)w.codes, w.codes_sorted, (const uint32_t*)w.idx,
                                              w.idx_sorted, P, 0, 30, st));
    hipLaunchKernelGGL(k_knn_gather, dim3(pblocks), dim3(GSR_BLOCK), 0, st, P, points, (const uint32_t*)w.idx_sorted, w.sorted);
    LAUNCHCHK("k_knn_gather");
    const int nboxes = (P + GSR_KNN_BOX - 1) / GSR_KNN_BOX;
    hipLaunchKernelGGL(k_knn_boxes, dim3(nboxes), dim3(GSR_KNN_BOX), 0, st, P, (const float4*)w.sorted, w.boxes);
    LAUNCHCHK("k_knn_boxes");
    hipLaunchKernelGGL(k_knn_search, dim3(pblocks), dim3(GSR_BLOCK), 0, st, P, (const float4*)w.sorted, (const float*)w.boxes, nboxes,
                       mean_dist2);
    LAUNCHCHK("k_knn_search");
    return 0;
}

size_t gsr_training_loss_bytes(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    return (size_t)width * height * 9 * sizeof(float) + 256;
}

int gsr_training_loss(int width, int height, const float* image, const float* gt_image, float lambda_dssim,
                      const float* depth, const float* pseudo_depth, float depth_weight, float* dL_dimage,
                      float* dL_ddepth, float* out, gsr_resize_fn workspace, void* workspace_ctx, void* stream)
{
    using namespace gsr;
    const int debug = 0;
    hipStream_t st = (hipStream_t)stream;
    if (width <= 0 || height <= 0) return fail(GSR_E_INVALID, "positive image size required%s", "");
    if (!image || !gt_image || !dL_dimage || !out || !workspace) return fail(GSR_E_INVALID, "gsr_training_loss: NULL pointer%s", "");
    if ((depth == nullptr) != (pseudo_depth == nullptr)) return fail(GSR_E_INVALID, "gsr_training_loss: depth and pseudo_depth go together%s", "");
    if (depth && !dL_ddepth) return fail(GSR_E_INVALID, "gsr_training_loss: dL_ddepth required with a depth term%s", "");
    int rc = select_device_of(image);
    if (rc != GSR_OK) return rc;
    const size_t N = (size_t)width * height;
    char* ws = (char*)workspace(workspace_ctx, gsr_training_loss_bytes(width, height));
    if (!ws) return fail(GSR_E_ALLOC, "workspace callback returned NULL%s", "");
    double* sums = reinterpret_cast<double*>(ws);                  // 10 doubles (256 B reserved)
    float* maps = reinterpret_cast<float*>(ws + 256);
    HIPCHK(hipMemsetAsync(sums, 0, 10 * sizeof(double), st));
    SsimArgs sa;
    sa.W = width; sa.H = height; sa.img = image; sa.gt = gt_image; sa.lambda_dssim = lambda_dssim; sa.maps = maps;
    sa.sums = sums; sa.dL_dimage = dL_dimage;
    {   // gaussian(11, 1.5) of loss_utils.py:23-25: float32 exp values, float32 sum, float32 division
        float g[11], sum = 0.f;
        for (int x = 0; x < 11; x++) { g[x] = (float)exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5)); sum += g[x]; }
        for (int x = 0; x < 11; x++) sa.w[x] = g[x] / sum;
    }
    const dim3 grid((width + GSR_SSIM_T - 1) / GSR_SSIM_T, (height + GSR_SSIM_T - 1) / GSR_SSIM_T, 3);
    hipLaunchKernelGGL(k_ssim_fwd, grid, dim3(GSR_SSIM_THREADS), 0, st, sa);
    LAUNCHCHK("k_ssim_fwd");
    PearsonArgs pa;
    pa.n = (int)N; pa.depth = depth; pa.pseudo = pseudo_depth; pa.sums = sums; pa.weight = depth_weight; pa.dL_ddepth = dL_ddepth;
    pa.out = out; pa.lambda_dssim = lambda_dssim; pa.npix3 = (int)(3 * N);
    const int eb = (int)((N + GSR_BLOCK - 1) / GSR_BLOCK);
    if (depth) {
        hipLaunchKernelGGL(k_pearson_sums, dim3(eb < 256 ? eb : 256), dim3(GSR_BLOCK), 0, st, pa);
        LAUNCHCHK("k_pearson_sums");
    }
    hipLaunchKernelGGL(k_train_loss_finish, dim3(depth ? (eb < 1024 ? eb : 1024) : 1), dim3(GSR_BLOCK), 0, st, pa, (const double*)sums);
    LAUNCHCHK("k_train_loss_finish");
    hipLaunchKernelGGL(k_ssim_bwd, grid, dim3(GSR_SSIM_THREADS), 0, st, sa);
    LAUNCHCHK("k_ssim_bwd");
    return 0;
}

int gsr_densification_stats(int P, const int* radii, const float* dL_dmean2D, float* max_radii2D,
                            float* xyz_gradient_accum, float* denom, void* stream)
{
    using namespace gsr;
    const int debug = 0;
    hipStream_t st = (hipStream_t)stream;
    if (P < 0) return fail(GSR_E_INVALID, "P >= 0 required%s", "");
    if (P == 0) return 0;
    if (!radii || !dL_dmean2D || !max_radii2D || !xyz_gradient_accum || !denom)
        return fail(GSR_E_INVALID, "gsr_densification_stats: NULL pointer%s", "");
    int rc = select_device_of(radii);
    if (rc != GSR_OK) return rc;
    hipLaunchKernelGGL(k_densification_stats, dim3((P + GSR_BLOCK - 1) / GSR_BLOCK), dim3(GSR_BLOCK), 0, st, P, radii, dL_dmean2D,
                       max_radii2D, xyz_gradient_accum, denom);
    LAUNCHCHK("k_densification_stats");
    return 0;
}

int gsr_map_from_ply_rows(int P, const float* rows, int row_floats, const int* cols, int n_rest, int activate,
                          float* means3D, float* shs, float* opacities, float* scales, float* rotations, void* stream)
{
    using namespace gsr;
    const int debug = 0;
    hipStream_t st = (hipStream_t)stream;
    if (P < 0 || row_floats <= 0) return fail(GSR_E_INVALID, "gsr_map_from_ply_rows: bad sizes%s", "");
    if (n_rest < 0 || n_rest > GSR_PLY_MAX_REST || n_rest % 3 != 0)
        return fail(GSR_E_INVALID, "gsr_map_from_ply_rows: n_rest must be a multiple of 3, at most 45%s", "");
    if (P == 0) return 0;
    if (!rows || !cols || !means3D || !shs || !opacities || !scales || !rotations)
        return fail(GSR_E_INVALID, "gsr_map_from_ply_rows: NULL pointer%s", "");
    for (int i = 0; i < 14 + n_rest; i++)
        if (cols[i] < 0 || cols[i] >= row_floats) return fail(GSR_E_INVALID, "gsr_map_from_ply_rows: column index outside the row%s", "");
    int rc = select_device_of(rows);
    if (rc != GSR_OK) return rc;
    PlyMapArgs a;
    a.P = P; a.row_floats = row_floats; a.activate = activate ? 1 : 0; a.M = 1 + n_rest / 3;
    a.rows = rows; a.means3D = means3D; a.shs = shs; a.opacities = opacities; a.scales = scales; a.rotations = rotations;
    int k = 0;
    for (int i = 0; i < 3; i++) a.c.xyz[i] = cols[k++];
    for (int i = 0; i < 3; i++) a.c.f_dc[i] = cols[k++];
    for (int i = 0; i < GSR_PLY_MAX_REST; i++) a.c.f_rest[i] = (i < n_rest) ? cols[k++] : 0;
    a.c.opacity = cols[k++];
    for (int i = 0; i < 3; i++) a.c.scale[i] = cols[k++];
    for (int i = 0; i < 4; i++) a.c.rot[i] = cols[k++];
    a.c.n_rest = n_rest;
    const size_t lds = (size_t)64 * (row_floats | 1) * sizeof(float);
    if (lds > 64 * 1024) return fail(GSR_E_INVALID, "gsr_map_from_ply_rows: rows of more than 255 floats are not supported%s", "");
    if (row_floats == 62 && a.M == 16)        // what 3DGS writes for SH degree 3 (x y z nx ny nz + 56)
        hipLaunchKernelGGL((k_map_from_ply_rows<62, 16>), dim3((P + 63) / 64), dim3(64), lds, st, a);
    else
        hipLaunchKernelGGL((k_map_from_ply_rows<0, 0>), dim3((P + 63) / 64), dim3(64), lds, st, a);
    LAUNCHCHK("k_map_from_ply_rows");
    return 0;
}

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream)
{
    (void)projmatrix;
    const int debug = 0;
    hipStream_t st = (hipStream_t)stream;
    if (P < 0) return fail(GSR_E_INVALID, "P must be >= 0%s", "");
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present) return fail(GSR_E_INVALID, "NULL pointer%s", "");
    int rc = select_device_of(means3D);
    if (rc != GSR_OK) return rc;
    hipLaunchKernelGGL(gsr::k_mark_visible, dim3((P + GSR_BLOCK - 1) / GSR_BLOCK), dim3(GSR_BLOCK), 0, st, P, means3D,
                       viewmatrix, present);
    LAUNCHCHK("k_mark_visible");
    return 0;
}

int gsr_forward_stats(int P, int width, int height, const int* radii, const char* geom_buffer, const char* img_buffer,
                      long long stats[4], void* stream)
{
    using namespace gsr;
    const int debug = 0;
    hipStream_t st = (hipStream_t)stream;
    if (P <= 0 || !radii || !geom_buffer || !img_buffer || !stats) return fail(GSR_E_INVALID, "bad stats arguments%s", "");
    int rc = select_device_of(radii);
    if (rc != GSR_OK) return rc;
    Geom g; carve_geom(const_cast<char*>(geom_buffer), P, g);
    Img im; carve_img(const_cast<char*>(img_buffer), width, height, im);
    const int gx = (width + GSR_TILE - 1) / GSR_TILE, gy = (height + GSR_TILE - 1) / GSR_TILE;
    unsigned long long* d = reinterpret_cast<unsigned long long*>(im.loss_shards);      // scratch (only the native loop uses these words)
    HIPCHK(hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_stats_gauss, dim3((P + GSR_BLOCK - 1) / GSR_BLOCK), dim3(GSR_BLOCK), 0, st, P, radii, (const ushort4*)g.rects, d);
    LAUNCHCHK("k_stats_gauss");
    hipLaunchKernelGGL(k_stats_tiles, dim3(gx * gy), dim3(GSR_BLOCK), 0, st, width, height, gx,
                       (const uint32_t*)im.n_contrib, (const uint2*)im.ranges, d);
    LAUNCHCHK("k_stats_tiles");
    unsigned long long h[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    stats[0] = (long long)h[0];
    stats[1] = (long long)h[1];   // R under the reference's bounding rule
    stats[2] = (long long)h[3];   // instances binned: sum of the tiles' list lengths (after exact tile culling / depth speculation)
    stats[3] = (long long)h[2];   // R_eff of THIS binning (max per-pixel n_contrib summed over tiles)
    return 0;
}

}  // extern "C"
