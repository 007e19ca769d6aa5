// extern "C" entry points of libgs4d.so (see include/gs4d.h) and the scratch-chunk layouts.
// Host-side orchestration only; the kernels live in the other translation units.
//
// Orchestration matched: RAST/cuda_rasterizer/rasterizer_impl.cu:198-339 (forward) and :343-444
// (backward) of the reference, RAST = submodules/depth-diff-gaussian-rasterization.  Unlike the
// reference, which launches on the legacy default stream, every launch goes to the stream the
// caller passes.  Process-wide state: the opt-in stage timer below (off by default).  Per-thread state: the error
// string and the forward's cached num_rendered mailbox (ThreadState).  One process per GPU stays independent.
#include <stdarg.h>
#include <stdio.h>

#include <stdlib.h>
#include <time.h>
#include "gs4d_internal.h"

int gs4d_launch_scan(int P, GeomState& g, unsigned long long* mailbox, uint32_t seq, int debug, hipStream_t s,
                     uint32_t* tile_cursor = nullptr, uint32_t* tile_base = nullptr, int ntiles = 0);

static thread_local char g_err[512] = "";

// Per-THREAD cached state of gs4d_rasterize_forward (nothing is shared between threads or processes): the pinned
// host mailbox the scan kernel posts num_rendered into, its sequence number, and the previous call's num_rendered
// (the guess that sizes the binning chunk before the count is known).  Released by gs4d_thread_state_release().
namespace {
struct ThreadState {
    unsigned long long* mailbox = nullptr;
    bool tried = false;
    uint32_t seq = 0;
    size_t last_R = 0;
    size_t recent_R[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the last eight counts: the speculative launch sizes for their maximum
    unsigned recent_i = 0;
    bool direct_next = false;                   // the last forward's counts say direct placement is the cheaper binning (binning.hip)
    unsigned long long binned_direct = 0, binned_passes = 0, last_adds = 0, last_rendered = 0;
    unsigned long long speculated = 0, respun = 0;  // forwards whose binning + blend were issued before the count was read / of those, re-issued
    unsigned long long fast = 0, slow = 0;      // how often the count arrived through the mailbox / the copy + sync path
    unsigned long long regrow = 0;              // forwards whose speculative binning chunk was too small (second allocation)
};
thread_local ThreadState t_state;
}  // namespace

#include <atomic>
static std::atomic<unsigned long long> g_alloc_calls{0};
void gs4d_note_alloc() { g_alloc_calls.fetch_add(1, std::memory_order_relaxed); }

void gs4d_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------- GS4D_TRACE
#include <sys/syscall.h>
#include <unistd.h>
namespace {
thread_local double t_trace_last = 0.0;
thread_local int t_trace_depth = 0;
double trace_slow_ms()
{
    static const double v = [] { const char* e = getenv("GS4D_TRACE_SLOW_US"); return (e ? atof(e) : 1000.0) * 1e-3; }();
    return v;
}
void trace_line(const char* fmt, ...)
{
    char buf[384];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    fprintf(stderr, "[gs4d-trace %d %ld %.3f] %s\n", (int)getpid(), (long)syscall(SYS_gettid), gs4d_trace_now_ms(), buf);
}
}  // namespace

bool gs4d_trace_on()
{
    static const bool v = [] { const char* e = getenv("GS4D_TRACE"); return e && e[0] && e[0] != '0'; }();
    return v;
}

double gs4d_trace_now_ms()
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec * 1e3 + (double)t.tv_nsec * 1e-6;
}

void gs4d_trace_checkpoint(const char* what)
{
    const double now = gs4d_trace_now_ms();
    if (t_trace_last > 0.0 && now - t_trace_last >= trace_slow_ms()) trace_line("SLOW %.3f ms up to: %s", now - t_trace_last, what);
    t_trace_last = now;
}

void gs4d_trace_blocked(const char* what, double t0_ms)
{
    const double now = gs4d_trace_now_ms();
    trace_line("blocked %.3f ms in %s", now - t0_ms, what);
    t_trace_last = now;
}

Gs4dTraceScope::Gs4dTraceScope(const char* n) : name(n), t0(0.0)
{
    if (!gs4d_trace_on()) return;
    t0 = gs4d_trace_now_ms();
    t_trace_last = t0;
    trace_line("%*senter %s", 2 * t_trace_depth, "", name);
    t_trace_depth++;
}

Gs4dTraceScope::~Gs4dTraceScope()
{
    if (t0 == 0.0) return;
    t_trace_depth--;
    const double now = gs4d_trace_now_ms();
    trace_line("%*sexit  %s after %.3f ms", 2 * t_trace_depth, "", name, now - t0);
    t_trace_last = now;
}

// ---------------------------------------------------------------- per-stage timing
#include <mutex>
#include <string>
#include <vector>
namespace {
struct ProfRec { std::string stage; hipEvent_t a, b; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::string g_prof_only;                                // when non-empty: the only stage that records events
std::vector<hipEvent_t> g_prof_pool;                    // resolved events, reused (event creation is slow)
hipEvent_t prof_event_locked()
{
    hipEvent_t e = nullptr;
    if (!g_prof_pool.empty()) { e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
std::vector<ProfRec> g_prof_open;                       // recorded, not yet resolved
struct ProfSum { std::string stage; double ms; long calls; };
std::vector<ProfSum> g_prof_sums;
void prof_resolve_locked()
{
    for (auto& r : g_prof_open) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            bool found = false;
            for (auto& s : g_prof_sums)
                if (s.stage == r.stage) { s.ms += ms; s.calls++; found = true; break; }
            if (!found) g_prof_sums.push_back({r.stage, ms, 1});
        }
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof_open.clear();
}
}  // namespace

Gs4dProfScope::Gs4dProfScope(const char* stage, hipStream_t s) : slot(-1), stream(s)
{
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_only.empty() && g_prof_only != stage) return;
    ProfRec r;
    r.stage = stage;
    if (!(r.a = prof_event_locked())) return;
    if (!(r.b = prof_event_locked())) { g_prof_pool.push_back(r.a); return; }
    (void)hipEventRecord(r.a, s);
    g_prof_open.push_back(r);
    slot = (int)g_prof_open.size() - 1;
}

Gs4dProfScope::~Gs4dProfScope()
{
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot < (int)g_prof_open.size()) (void)hipEventRecord(g_prof_open[slot].b, stream);
}

// ---------------------------------------------------------------- chunk layouts
GeomState GeomState::from_chunk(char*& chunk, size_t P)
{
    GeomState g;
    const size_t nblk = (P + GS4D_PRE_BLOCK - 1) / GS4D_PRE_BLOCK;
    gs4d_obtain(chunk, g.depths, P);
    gs4d_obtain(chunk, g.clamped, P * 3);
    gs4d_obtain(chunk, g.means2D, P);
    gs4d_obtain(chunk, g.cov3D, P * 6);
    gs4d_obtain(chunk, g.conic_opacity, P);
    gs4d_obtain(chunk, g.rgb, P * 3);
    gs4d_obtain(chunk, g.tiles_touched, P);
    gs4d_obtain(chunk, g.point_offsets, P);
    gs4d_obtain(chunk, g.block_sums, nblk);
    gs4d_obtain(chunk, g.block_offsets, nblk);
    gs4d_obtain(chunk, g.counters, 64);
    gs4d_obtain(chunk, g.block_area, 4 * nblk);
    return g;
}

size_t GeomState::required(size_t P)
{
    char* p = nullptr;
    GeomState::from_chunk(p, P);
    return (size_t)p + 256;
}

ImageState ImageState::from_chunk(char*& chunk, size_t npix, size_t ntiles)
{
    ImageState im;
    gs4d_obtain(chunk, im.final_T, npix);
    gs4d_obtain(chunk, im.n_contrib, npix);
    gs4d_obtain(chunk, im.ranges, ntiles);
    gs4d_obtain(chunk, im.tile_max_contrib, ntiles);
    gs4d_obtain(chunk, im.tile_cursor, ntiles * GS4D_TILE_CURSOR_STRIDE);
    gs4d_obtain(chunk, im.tile_base, ntiles + 1);
    return im;
}

size_t ImageState::required(size_t npix, size_t ntiles)
{
    char* p = nullptr;
    ImageState::from_chunk(p, npix, ntiles);
    return (size_t)p + 256;
}

BinningState BinningState::from_chunk(char*& chunk, size_t R, int passes, bool checkpoints)
{
    BinningState b;
    gs4d_obtain(chunk, b.keys_unsorted, R);
    gs4d_obtain(chunk, b.values_unsorted, R);
    gs4d_obtain(chunk, b.keys_tmp[0], R);
    gs4d_obtain(chunk, b.keys_tmp[1], R);
    gs4d_obtain(chunk, b.values_tmp[0], R);
    gs4d_obtain(chunk, b.values_tmp[1], R);
    gs4d_obtain(chunk, b.hist, (size_t)256 * gs4d_sort_blocks(R) + 256);
    // the blend's segment checkpoints (80 of the ~116 bytes per instance) come last and exist only when a backward
    // may follow: a GS4D_FORWARD_ONLY frame (render FPS, evaluation) neither writes nor reserves them
    b.seg_slots = R / GS4D_SEG_SLOT_DIV + 2;
    b.seg_owner = nullptr;
    b.ckpt = nullptr;
    if (checkpoints) {
        gs4d_obtain(chunk, b.seg_owner, b.seg_slots);
        gs4d_obtain(chunk, b.ckpt, b.seg_slots * GS4D_CKPT_FIELDS * 128);
    }
    b.sorted = passes > 0 ? (passes - 1) & 1 : 0;
    return b;
}

size_t BinningState::required(size_t R, bool checkpoints)
{
    char* p = nullptr;
    BinningState::from_chunk(p, R, 1, checkpoints);
    return (size_t)p + 256;
}

bool gs4d_sort_tile_first()
{
    static const bool v = [] { const char* e = getenv("GS4D_SORT"); return !(e && e[0] == 'l'); }();
    return v;
}

bool gs4d_forward_bins_directly()
{
    return gs4d_sort_tile_first() && (gs4d_sort_direct_forced() >= 0 ? gs4d_sort_direct_forced() == 1 : t_state.direct_next);
}

int gs4d_sort_direct_forced()
{
    static const int v = [] { const char* e = getenv("GS4D_SORT"); return !e || !e[0] ? -1 : (e[0] == 'd' ? 1 : 0); }();
    return v;
}

uint32_t gs4d_sort_segment_cap()
{
    // GS4D_SORT_SEGMENT_CAP (tests): lists longer than this take the chunked global path; at most 2048 (SEG_CAP of binning.hip)
    static const uint32_t v = [] {
        const char* e = getenv("GS4D_SORT_SEGMENT_CAP");
        long c = e ? atol(e) : 2048;
        return (uint32_t)(c < 0 ? 0 : (c > 2048 ? 2048 : c));
    }();
    return v;
}

uint32_t gs4d_sort_wave_cap()
{
    // GS4D_SORT_WAVE_CAP (tests): lists of at most this many entries are sorted by one wave each; at most 512 (WSEG_CAP
    // of binning.hip), 0 sends every list through the workgroup form
    static const uint32_t v = [] {
        const char* e = getenv("GS4D_SORT_WAVE_CAP");
        long c = e ? atol(e) : 512;
        return (uint32_t)(c < 0 ? 0 : (c > 512 ? 512 : c));
    }();
    return v;
}

// Number of ping-pong writes the sort performs: decides which half holds the sorted lists (BinningState::sorted =
// (n - 1) & 1), for the forward that runs the sort and for the backward / state view that only re-derive the layout.
static int sort_passes(int W, int H)
{
    const int gx = (W + GS4D_TILE_X - 1) / GS4D_TILE_X, gy = (H + GS4D_TILE_Y - 1) / GS4D_TILE_Y;
    const int tbits = (int)gs4d_higher_msb((uint32_t)(gx * gy));
    if (gs4d_sort_tile_first()) return (tbits + 7) / 8 + 1;      // tile passes + the per-tile depth sort
    return (32 + tbits + 7) / 8;
}

extern "C" {

const char* gs4d_last_error(void) { return g_err; }
unsigned long long gs4d_debug_alloc_calls(void) { return g_alloc_calls.load(std::memory_order_relaxed); }
int gs4d_abi_version(void) { return 13; }

void gs4d_thread_state_release(void)
{
    ThreadState& ts = t_state;
    if (ts.mailbox) (void)hipHostFree(ts.mailbox);
    ts = ThreadState();
}

void gs4d_mailbox_stats(unsigned long long* fast, unsigned long long* slow)
{
    if (fast) *fast = t_state.fast;
    if (slow) *slow = t_state.slow;
}

void gs4d_forward_stats(unsigned long long* fast, unsigned long long* slow, unsigned long long* regrow)
{
    gs4d_mailbox_stats(fast, slow);
    if (regrow) *regrow = t_state.regrow;
}

void gs4d_binning_stats(unsigned long long* direct, unsigned long long* passes, unsigned long long* last_adds,
                        unsigned long long* last_rendered)
{
    if (direct) *direct = t_state.binned_direct;
    if (passes) *passes = t_state.binned_passes;
    if (last_adds) *last_adds = t_state.last_adds;
    if (last_rendered) *last_rendered = t_state.last_rendered;
}

void gs4d_speculation_stats(unsigned long long* issued, unsigned long long* reissued)
{
    if (issued) *issued = t_state.speculated;
    if (reissued) *reissued = t_state.respun;
}

void gs4d_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
}

void gs4d_profile_select(const char* stage)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_only = stage ? stage : "";
}

void gs4d_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_resolve_locked();
    g_prof_sums.clear();
}

int gs4d_profile_read(const char* stage, double* total_ms, long* calls)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_resolve_locked();
    for (auto& s : g_prof_sums)
        if (s.stage == stage) {
            if (total_ms) *total_ms = s.ms;
            if (calls) *calls = s.calls;
            return GS4D_OK;
        }
    if (total_ms) *total_ms = 0.0;
    if (calls) *calls = 0;
    return GS4D_OK;
}

size_t gs4d_geometry_bytes(int P) { return GeomState::required((size_t)P); }
size_t gs4d_binning_bytes(int R) { return BinningState::required((size_t)R); }
size_t gs4d_binning_bytes_forward_only(int R) { return BinningState::required((size_t)R, false); }
size_t gs4d_image_bytes(int width, int height)
{
    const size_t gx = (width + GS4D_TILE_X - 1) / GS4D_TILE_X, gy = (height + GS4D_TILE_Y - 1) / GS4D_TILE_Y;
    return ImageState::required((size_t)width * height, gx * gy);
}
size_t gs4d_backward_scratch_bytes(int R)
{
    return gs4d_align_up((size_t)R * GS4D_GRAD_ROW * sizeof(float), 256) + 256 + ((size_t)R / 64 + 64) * sizeof(uint32_t) + 256;
}

// speculate != 0 (the composite render call, csrc/render.hip): binning and blend are issued BEFORE the host has read
// num_rendered, laid out and sized for the largest of the thread's last eight counts + 1/8, with the count taken from
// device memory by the kernels; the host then reads the mailbox as always and re-issues the two stages only if the count
// exceeded that capacity (the speculative kernels do nothing in that case).  The GPU no longer idles for the round trip.
// *layout_rendered = the count the binning chunk is laid out for (the backward needs it to find the arrays again).
}  // extern "C"
int gs4d_rasterize_forward_spec(gs4d_alloc_fn geom_alloc, void* geom_ctx, gs4d_alloc_fn binning_alloc, void* binning_ctx,
                                gs4d_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                                int width, int height, const float* means3D, const float* shs,
                                const float* colors_precomp, const float* opacities, const float* scales,
                                float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                                float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii, int debug_arg,
                                void* stream, int* num_rendered, int speculate, int* layout_rendered, uint8_t* visible)
{
    GS4D_TRACE_SCOPE("gs4d_rasterize_forward");
    if (layout_rendered) *layout_rendered = 0;
    hipStream_t s = (hipStream_t)stream;
    const int debug = debug_arg & 1;                // bit 1 (GS4D_FORWARD_ONLY): see include/gs4d.h
    if (num_rendered) *num_rendered = 0;
    if (P < 0 || width <= 0 || height <= 0 || !geom_alloc || !binning_alloc || !image_alloc) {
        gs4d_set_error("gs4d_rasterize_forward: bad sizes or missing allocator");
        return GS4D_EINVAL;
    }
    if (P == 0) return GS4D_OK;  // rasterize_points.cu:82
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !cam_pos || !background || !out_color || !out_depth ||
        !radii) {
        gs4d_set_error("gs4d_rasterize_forward: null required pointer");
        return GS4D_EINVAL;
    }
    if ((shs == nullptr) == (colors_precomp == nullptr)) {
        gs4d_set_error("provide exactly one of SHs or precomputed colors");
        return GS4D_EINVAL;
    }
    if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr)) {
        gs4d_set_error("provide exactly one of scale/rotation pair or precomputed 3D covariance");
        return GS4D_EINVAL;
    }
    if (shs && (M < (D + 1) * (D + 1) || D < 0 || D > 3)) {
        gs4d_set_error("SH degree %d needs %d coefficients, got M=%d", D, (D + 1) * (D + 1), M);
        return GS4D_EINVAL;
    }
    if (rotations && ((uintptr_t)rotations & 15)) {
        gs4d_set_error("rotations must be 16-byte aligned");
        return GS4D_EINVAL;
    }

    gs4d_note_alloc();
    char* chunk = (char*)geom_alloc(geom_ctx, GeomState::required((size_t)P));
    if (!chunk) { gs4d_set_error("geometry allocator returned NULL"); return GS4D_EALLOC; }
    GeomState g = GeomState::from_chunk(chunk, (size_t)P);

    const size_t gx = (width + GS4D_TILE_X - 1) / GS4D_TILE_X, gy = (height + GS4D_TILE_Y - 1) / GS4D_TILE_Y;
    gs4d_note_alloc();
    char* ichunk = (char*)image_alloc(image_ctx, ImageState::required((size_t)width * height, gx * gy));
    if (!ichunk) { gs4d_set_error("image allocator returned NULL"); return GS4D_EALLOC; }
    ImageState im = ImageState::from_chunk(ichunk, (size_t)width * height, gx * gy);

    // How this forward bins its instances: by direct placement (csrc/tile_place.h) when the counts of the previous forward
    // of this thread say it is the cheaper way (few instances, rows in a spatially coherent order: scan_block_sums_kernel;
    // or GS4D_SORT=direct), by the radix passes on the tile bits otherwise (GS4D_SORT=tile: always).  Same lists either way, bit for bit, in the same half of the binning chunk.
    const bool direct = gs4d_forward_bins_directly();
    const bool counters_cleared = (speculate & 2) != 0;     // the caller's previous launch cleared im.tile_cursor (csrc/render.hip)
    speculate &= 1;
    int rc = gs4d_launch_preprocess(P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs,
                                    cov3D_precomp, colors_precomp, viewmatrix, projmatrix, cam_pos, width, height,
                                    tan_fovx, tan_fovy, prefiltered, radii, g, debug, s,
                                    direct ? im.tile_cursor : nullptr, counters_cleared, visible);
    if (rc) return rc;
    // The one host round trip of the forward (rasterizer_impl.cu:282): num_rendered sizes the binning chunk.  The scan
    // kernel posts ONE 64-bit word (sequence | prefilter flag, num_rendered) into a pinned, coherent host mailbox and the
    // host polls it -- a few microseconds instead of a D2H copy plus a stream synchronisation -- for at most
    // GS4D_MAILBOX_SPIN_US (default 5 ms) of wall clock; after that (a long queue ahead of this call, a shared or hung
    // device) it takes the ordinary copy + synchronise path, which also reports a faulted stream.  The binning chunk is
    // requested BEFORE the wait with the previous call's size plus 1/8 (the allocator callback runs Python), again
    // afterwards only if that guess was too small.
    ThreadState& ts = t_state;
    if (!ts.tried) {
        ts.tried = true;
        void* p = nullptr;
        gs4d_note_alloc();
        if (getenv("GS4D_NO_MAILBOX") == nullptr &&
            hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocCoherent) == hipSuccess) {
            ts.mailbox = (unsigned long long*)p;
            __atomic_store_n(ts.mailbox, 0ull, __ATOMIC_RELEASE);
        } else {
            (void)hipGetLastError();
        }
    }
    ts.seq += 4;                                // bit 0 carries the prefilter flag, bit 1 "direct placement next"
    if (ts.seq == 0) ts.seq = 4;
    const uint32_t seq = ts.seq;
    rc = gs4d_launch_scan(P, g, ts.mailbox, seq, debug, s, direct ? im.tile_cursor : nullptr, im.tile_base, (int)(gx * gy));
    if (rc) return rc;

    const bool with_ckpt = !(debug_arg & GS4D_FORWARD_ONLY);
    static const bool spec_off = [] { const char* e = getenv("GS4D_NO_SPECULATION"); return e && e[0] && e[0] != '0'; }();
    speculate = speculate && !debug && !spec_off;
    char* bchunk = nullptr;
    size_t guess_R = 0;
    if (ts.last_R) {
        size_t base_R = ts.last_R;
        if (speculate)
            for (size_t r : ts.recent_R) base_R = r > base_R ? r : base_R;
        guess_R = base_R + base_R / 8 + 4096;
        gs4d_note_alloc();
        bchunk = (char*)binning_alloc(binning_ctx, BinningState::required(guess_R, with_ckpt));
        if (gs4d_trace_on()) gs4d_trace_checkpoint("binning allocator callback (speculative size)");
        if (!bchunk) {
            (void)hipStreamSynchronize(s);                                  // no post may arrive after this call has returned
            gs4d_set_error("binning allocator returned NULL");
            return GS4D_EALLOC;
        }
    }

    const float* features = colors_precomp ? colors_precomp : g.rgb;
    bool spec_issued = false;
    if (speculate && bchunk && guess_R <= 0x7fffffffu) {
        char* c = bchunk;
        BinningState bs = BinningState::from_chunk(c, guess_R, sort_passes(width, height), with_ckpt);
        rc = gs4d_launch_binning(P, (int)guess_R, width, height, radii, g, bs, im, debug, s, g.counters, direct);
        if (!rc)
            rc = gs4d_launch_blend_forward(width, height, g, bs, im, features, background, out_color, out_depth, with_ckpt, debug, s);
        if (rc) {
            (void)hipStreamSynchronize(s);                                      // no post may arrive after this call has returned
            return rc;
        }
        spec_issued = true;
        ts.speculated++;
    }

    uint32_t host_counters[4] = {0, 0, 0, 0};
    bool posted = false;
    const double t_wait0 = gs4d_trace_on() ? gs4d_trace_now_ms() : 0.0;
    if (ts.mailbox) {
        static const long spin_us = [] { const char* e = getenv("GS4D_MAILBOX_SPIN_US"); long v = e ? atol(e) : 5000; return v < 0 ? 0L : v; }();
        struct timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned spin = 0;; spin++) {
            const unsigned long long w = __atomic_load_n(ts.mailbox, __ATOMIC_ACQUIRE);   // one word: payload and sequence together
            if ((uint32_t)(w >> 32 & ~3ull) == seq) {
                host_counters[0] = (uint32_t)w;
                host_counters[1] = (uint32_t)(w >> 32) & 1u;
                host_counters[2] = (uint32_t)(w >> 32) & 2u;
                host_counters[3] = (uint32_t)__atomic_load_n(ts.mailbox + 1, __ATOMIC_RELAXED);   // written before the post
                posted = true;
                break;
            }
            __builtin_ia32_pause();
            if ((spin & 127u) == 127u) {
                struct timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000L + (t1.tv_nsec - t0.tv_nsec) / 1000L >= spin_us) break;
            }
        }
    }
    if (posted) {
        ts.fast++;
        if (gs4d_trace_on()) gs4d_trace_blocked("num_rendered mailbox wait (host polls pinned memory, no runtime call)", t_wait0);
    } else {
        ts.slow++;
        if (gs4d_trace_on() && ts.mailbox) gs4d_trace_blocked("num_rendered mailbox wait, TIMED OUT", t_wait0);
        const double t_sync0 = gs4d_trace_on() ? gs4d_trace_now_ms() : 0.0;
        GS4D_HIP_CHECK(hipMemcpyAsync(host_counters, g.counters, sizeof host_counters, hipMemcpyDeviceToHost, s));
        GS4D_HIP_CHECK(hipStreamSynchronize(s));
        if (gs4d_trace_on()) gs4d_trace_blocked("hipMemcpyAsync + hipStreamSynchronize (num_rendered fall-back)", t_sync0);
    }
    if (host_counters[1]) {
        gs4d_set_error("Point is filtered although prefiltered is set. This shouldn't happen!");
        return GS4D_EPREFILTER;
    }
    if (host_counters[0] > 0x7fffffffu) {
        gs4d_set_error("num_rendered %u overflows int", host_counters[0]);
        return GS4D_EINVAL;
    }
    const int R = (int)host_counters[0];
    ts.direct_next = host_counters[2] != 0u;
    ts.last_adds = host_counters[3];
    ts.last_rendered = host_counters[0];
    (direct ? ts.binned_direct : ts.binned_passes)++;
    if (num_rendered) *num_rendered = R;
    ts.recent_R[ts.recent_i++ & 7u] = (size_t)R;
    if (spec_issued && (size_t)R <= guess_R) {             // what is in the stream already is the result
        ts.last_R = (size_t)R;
        if (layout_rendered) *layout_rendered = (int)guess_R;
        return GS4D_OK;
    }
    if (spec_issued) ts.respun++;
    if (layout_rendered) *layout_rendered = R;

    if (!bchunk || (size_t)R > guess_R) {
        if (bchunk) ts.regrow++;
        gs4d_note_alloc();
        bchunk = (char*)binning_alloc(binning_ctx, BinningState::required((size_t)R, with_ckpt));
        if (!bchunk) { gs4d_set_error("binning allocator returned NULL"); return GS4D_EALLOC; }
    }
    ts.last_R = (size_t)R;
    BinningState b = BinningState::from_chunk(bchunk, (size_t)R, sort_passes(width, height), with_ckpt);

    rc = gs4d_launch_binning(P, R, width, height, radii, g, b, im, debug, s, nullptr, direct);
    if (rc) return rc;
    return gs4d_launch_blend_forward(width, height, g, b, im, features, background, out_color, out_depth,
                                     !(debug_arg & GS4D_FORWARD_ONLY), debug, s);
}
extern "C" {

int gs4d_rasterize_forward(gs4d_alloc_fn geom_alloc, void* geom_ctx, gs4d_alloc_fn binning_alloc, void* binning_ctx,
                           gs4d_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                           int width, int height, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           float scale_modifier, const float* rotations, const float* cov3D_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                           float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii, int debug_arg,
                           void* stream, int* num_rendered)
{
    // the binning chunk of this entry point is laid out for num_rendered exactly (gs4d_rasterize_backward and
    // gs4d_state_view_get find its arrays from that number alone): no speculative launch here
    return gs4d_rasterize_forward_spec(geom_alloc, geom_ctx, binning_alloc, binning_ctx, image_alloc, image_ctx, P, D, M, background,
                                       width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                       cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color,
                                       out_depth, radii, debug_arg, stream, num_rendered, 0, nullptr);
}

}  // extern "C"
// layout_rendered: the count the forward laid the binning chunk out for (gs4d_rasterize_forward_spec); R otherwise
int gs4d_rasterize_backward_layout(gs4d_alloc_fn scratch_alloc, void* scratch_ctx, int P, int D, int M, int R,
                                   const float* background, int width, int height, const float* means3D, const float* shs,
                                   const float* colors_precomp, const float* scales, float scale_modifier,
                                   const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                   const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                                   const int* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
                                   const float* dL_dpix, const float* dL_dpix_depth, float* dL_dmean2D, float* dL_dconic,
                                   float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
                                   float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug,
                                   void* stream, int layout_rendered, const Gs4dActBwd* act)
{
    GS4D_TRACE_SCOPE("gs4d_rasterize_backward");
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) return GS4D_OK;
    if (P < 0 || R < 0 || width <= 0 || height <= 0 || !scratch_alloc || !geom_buffer || !img_buffer ||
        (R > 0 && !binning_buffer) || !dL_dpix || !means3D || !radii) {      // dL_dpix_depth == null: zeros (csrc/render.hip only)
        gs4d_set_error("gs4d_rasterize_backward: bad sizes or null pointer");
        return GS4D_EINVAL;
    }
    if (rotations && (((uintptr_t)rotations & 15) || ((uintptr_t)dL_drot & 15))) {
        gs4d_set_error("rotations / dL_drot must be 16-byte aligned");
        return GS4D_EINVAL;
    }
    if (R == 0) return GS4D_OK;  // nothing was rendered: every gradient stays zero
    char* chunk = geom_buffer;
    GeomState g = GeomState::from_chunk(chunk, (size_t)P);
    const size_t gx = (width + GS4D_TILE_X - 1) / GS4D_TILE_X, gy = (height + GS4D_TILE_Y - 1) / GS4D_TILE_Y;
    char* ichunk = img_buffer;
    ImageState im = ImageState::from_chunk(ichunk, (size_t)width * height, gx * gy);
    char* bchunk = binning_buffer;
    if (layout_rendered < R) { gs4d_set_error("gs4d_rasterize_backward: the binning chunk is laid out for %d instances, fewer than the %d rendered", layout_rendered, R); return GS4D_EINVAL; }
    BinningState b = BinningState::from_chunk(bchunk, (size_t)layout_rendered, sort_passes(width, height));

    gs4d_note_alloc();
    float* inst_grad = (float*)scratch_alloc(scratch_ctx, gs4d_backward_scratch_bytes(R));
    if (!inst_grad) { gs4d_set_error("backward scratch allocator returned NULL"); return GS4D_EALLOC; }
    if ((uintptr_t)inst_grad & 255) { gs4d_set_error("backward scratch must be 256-byte aligned"); return GS4D_EINVAL; }

    const float* features = colors_precomp ? colors_precomp : g.rgb;
    int rc = gs4d_launch_blend_backward(width, height, R, radii, g, b, im, features, background, dL_dpix, dL_dpix_depth,
                                        inst_grad, gs4d_backward_big_count(inst_grad, R), debug, s);
    if (rc) return rc;
    const float* cov3D = cov3D_precomp ? cov3D_precomp : g.cov3D;
    return gs4d_launch_geom_backward(P, D, M, R, width, height, means3D, radii, shs, scales, rotations,
                                     scale_modifier, cov3D, viewmatrix, projmatrix, tan_fovx, tan_fovy, campos, g, b,
                                     inst_grad, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D,
                                     dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, s, act);
}
extern "C" {

int gs4d_rasterize_backward(gs4d_alloc_fn scratch_alloc, void* scratch_ctx, int P, int D, int M, int R,
                            const float* background, int width, int height, const float* means3D, const float* shs,
                            const float* colors_precomp, const float* scales, float scale_modifier,
                            const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                            const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                            const int* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
                            const float* dL_dpix, const float* dL_dpix_depth, float* dL_dmean2D, float* dL_dconic,
                            float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
                            float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug,
                            void* stream)
{
    if (P > 0 && !dL_dpix_depth) { gs4d_set_error("gs4d_rasterize_backward: bad sizes or null pointer"); return GS4D_EINVAL; }
    return gs4d_rasterize_backward_layout(scratch_alloc, scratch_ctx, P, D, M, R, background, width, height, means3D, shs,
                                          colors_precomp, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                          campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, img_buffer, dL_dpix,
                                          dL_dpix_depth, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D,
                                          dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, stream, R);
}

int gs4d_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                      uint8_t* present, void* stream)
{
    (void)projmatrix;
    if (P == 0) return GS4D_OK;
    if (P < 0 || !means3D || !viewmatrix || !present) {
        gs4d_set_error("gs4d_mark_visible: bad arguments");
        return GS4D_EINVAL;
    }
    return gs4d_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
}

int gs4d_state_view_get(int P, int R, int width, int height, char* geom_buffer, char* binning_buffer,
                        char* img_buffer, gs4d_state_view* out)
{
    if (!out || !geom_buffer || !img_buffer || P < 0 || R < 0) {
        gs4d_set_error("gs4d_state_view_get: bad arguments");
        return GS4D_EINVAL;
    }
    char* c = geom_buffer;
    GeomState g = GeomState::from_chunk(c, (size_t)P);
    const size_t gx = (width + GS4D_TILE_X - 1) / GS4D_TILE_X, gy = (height + GS4D_TILE_Y - 1) / GS4D_TILE_Y;
    c = img_buffer;
    ImageState im = ImageState::from_chunk(c, (size_t)width * height, gx * gy);
    out->depths = g.depths; out->clamped = g.clamped; out->means2D = (const float*)g.means2D; out->cov3D = g.cov3D;
    out->conic_opacity = (const float*)g.conic_opacity; out->rgb = g.rgb; out->tiles_touched = g.tiles_touched;
    out->point_offsets = g.point_offsets;
    out->ranges = (const uint32_t*)im.ranges; out->final_T = im.final_T; out->n_contrib = im.n_contrib;
    out->keys_unsorted = nullptr; out->values_unsorted = nullptr; out->keys_sorted = nullptr; out->point_list = nullptr;
    if (binning_buffer && R > 0) {
        c = binning_buffer;
        BinningState b = BinningState::from_chunk(c, (size_t)R, sort_passes(width, height));
        out->keys_unsorted = b.keys_unsorted; out->values_unsorted = b.values_unsorted;
        out->keys_sorted = b.keys_sorted(); out->point_list = b.point_list();
    }
    return GS4D_OK;
}

}  // extern "C"
