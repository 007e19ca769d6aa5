// Internal definitions shared by the HIP translation units of libgs4d.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gs4d.h"

#define GS4D_TILE_X 16
#define GS4D_TILE_Y 16
#define GS4D_TILE_PIX (GS4D_TILE_X * GS4D_TILE_Y)
#define GS4D_PRE_BLOCK 256          // Gaussians per preprocess / emit workgroup
#define GS4D_GRAD_ROW 12            // floats per per-instance gradient row (10 used, padded to 48 B)
#ifndef GS4D_SEG
#define GS4D_SEG 128                // list entries per backward segment: 128 or 256 (half of / the forward blend's batch)
#endif
#define GS4D_SEG_SLOT_DIV (GS4D_SEG / 2)   // a tile whose list starts at x owns checkpoint slots from x / GS4D_SEG_SLOT_DIV
#define GS4D_CKPT_FIELDS 5

void gs4d_set_error(const char* fmt, ...);
// every request the library makes for memory -- an allocator callback or a runtime allocation of its own -- is counted
// (gs4d_debug_alloc_calls, include/gs4d.h): the allocation-free entry point's test reads the counter around the call
void gs4d_note_alloc();

// GS4D_TRACE=1 (environment, read once): every compute entry point logs entry / exit with wall-clock times to stderr,
// every blocking runtime call inside the library (the num_rendered wait and its copy + synchronise fall-back, debug
// synchronisations) logs how long it blocked, and any stretch between two launch checkpoints of one entry point that
// exceeds GS4D_TRACE_SLOW_US (default 1000) is logged with the name of the launch it ended at -- a kernel launch that
// takes that long is a full queue that is not draining.  One line per event: "[gs4d-trace pid tid t_ms] ...".
bool gs4d_trace_on();
void gs4d_trace_checkpoint(const char* what);            // time since the thread's previous checkpoint, logged if slow
void gs4d_trace_blocked(const char* what, double t0_ms); // "blocked <ms> in <what>", always logged
double gs4d_trace_now_ms();
struct Gs4dTraceScope {
    const char* name;
    double t0;
    explicit Gs4dTraceScope(const char* n);
    ~Gs4dTraceScope();
};
#define GS4D_TRACE_SCOPE(name) Gs4dTraceScope gs4d_trace_scope_(name)

#define GS4D_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            gs4d_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return GS4D_EHIP;                                                             \
        }                                                                                 \
    } while (0)

#define GS4D_LAUNCH_CHECK(name, debug, stream)                                            \
    do {                                                                                  \
        if (gs4d_trace_on()) gs4d_trace_checkpoint("launch " name);                       \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ == hipSuccess && (debug)) e_ = hipStreamSynchronize(stream);               \
        if (e_ != hipSuccess) {                                                           \
            gs4d_set_error("kernel %s failed: %s", name, hipGetErrorString(e_));          \
            return GS4D_EHIP;                                                             \
        }                                                                                 \
    } while (0)

// Optional per-stage device timing (HIP events on the launch stream), off by default.
// See gs4d_profile_enable / gs4d_profile_read in include/gs4d.h.
struct Gs4dProfScope {
    int slot;
    hipStream_t stream;
    Gs4dProfScope(const char* stage, hipStream_t s);
    ~Gs4dProfScope();
};
#define GS4D_PROF(stage, stream) Gs4dProfScope gs4d_prof_scope_(stage, stream)

static inline size_t gs4d_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T>
static inline void gs4d_obtain(char*& chunk, T*& ptr, size_t count)
{
    size_t off = gs4d_align_up((size_t)chunk, 256);
    ptr = (T*)off;
    chunk = (char*)(ptr + count);
}

// ---- scratch layouts (opaque to callers; see gs4d_state_view_get) ----
struct GeomState {
    float* depths;            // [P] view-space z
    uint8_t* clamped;         // [P,3]
    float2* means2D;          // [P]
    float* cov3D;             // [P,6]
    float4* conic_opacity;    // [P]
    float* rgb;               // [P,3]
    uint32_t* tiles_touched;  // [P]
    uint32_t* point_offsets;  // [P] inclusive scan of tiles_touched
    uint32_t* block_sums;     // [ceil(P/256)]
    uint32_t* block_offsets;  // [ceil(P/256)] exclusive scan of block_sums
    uint32_t* counters;       // [64] counters[0] = num_rendered, counters[1] = prefilter violation flag, counters[2] = 2 if
                              // the next forward should bin by direct placement (scan_block_sums_kernel), else 0;
                              // counters[3] = the device-scope adds direct placement needs (exact / upper bound)
    uint32_t* block_area;     // [ceil(P/256)][4 waves] the device-scope adds direct placement needs for the workgroup: exact,
                              // or an upper bound from the rectangle of tiles around its Gaussians (preprocess launch)
    static GeomState from_chunk(char*& chunk, size_t P);
    static size_t required(size_t P);
};

#define GS4D_TILE_CURSOR_STRIDE 32
struct ImageState {
    float* final_T;           // [H*W]
    uint32_t* n_contrib;      // [H*W]
    uint2* ranges;            // [tiles]
    uint32_t* tile_max_contrib; // [tiles] max n_contrib over the tile's pixels (written by the forward blend)
    // GS4D_SORT=direct (binning.hip): instances per tile, counted by the preprocess launch (cleared in front of it), turned
    // into each list's write cursor by the scan launch; tile_base = the lists' first positions, [tiles] = num_rendered
    uint32_t* tile_cursor;    // [tiles][GS4D_TILE_CURSOR_STRIDE]: one counter per 128-byte line (the memory-side atomics
                              // serialise per line: 16 tiles to a line made the counting 52 us at C2 instead of ~5)
    uint32_t* tile_base;      // [tiles + 1]
    static ImageState from_chunk(char*& chunk, size_t npix, size_t ntiles);
    static size_t required(size_t npix, size_t ntiles);
};

struct BinningState {
    uint64_t* keys_unsorted;  // [R] emission order (kept intact for gs4d_state_view_get)
    uint32_t* values_unsorted;
    uint64_t* keys_tmp[2];    // radix-sort ping-pong
    uint32_t* values_tmp[2];
    uint32_t* hist;           // sort scratch: [256 digits][sort blocks] + [256] digit totals
    // Segment checkpoints of the blend (blend_fwd.hip / blend_bwd.hip): the forward leaves every pixel's state in front
    // of list positions S, 2 S, ... of its tile (S = GS4D_SEG), so that the backward can walk the S-entry segments of a long
    // list in parallel.  A tile whose list starts at x and has n > S entries owns the slots x / (S / 2) .. x / (S / 2) + (n - 1) / S
    // (checkpoints 1 .. c, then the final colour / depth record): no two tiles share a slot, nothing is counted.
    float2* ckpt;             // [seg_slots][GS4D_CKPT_FIELDS][128 lanes]: T, C0, C1, C2, depth of the lane's two pixels
    uint2* seg_owner;         // [seg_slots] (tile, segment >= 1) whose backward workgroup this slot is; x = ~0u: none
    size_t seg_slots;         // R / GS4D_SEG_SLOT_DIV + 2
    int sorted;               // which ping-pong half holds the sorted lists after `passes` passes
    uint64_t* keys_sorted() const { return keys_tmp[sorted]; }
    uint32_t* point_list() const { return values_tmp[sorted]; }
    static BinningState from_chunk(char*& chunk, size_t R, int passes, bool checkpoints = true);
    static size_t required(size_t R, bool checkpoints = true);
};

// sort tile geometry
#define GS4D_SORT_THREADS 256
#define GS4D_SORT_ITEMS 8
#define GS4D_SORT_TILE (GS4D_SORT_THREADS * GS4D_SORT_ITEMS)
static inline uint32_t gs4d_sort_blocks(size_t R) { return (uint32_t)((R + GS4D_SORT_TILE - 1) / GS4D_SORT_TILE); }

// rasterizer_impl.cu:35-50 semantics: number of bits needed for n
static inline uint32_t gs4d_higher_msb(uint32_t n)
{
    uint32_t b = 0;
    while (b < 32 && (n >> b)) b++;
    return b == 0 ? 1 : b;
}

// sort strategy of the binning stage (process-wide, read once from the environment: GS4D_SORT=lsd selects the
// six-pass LSD sort; default is tile-first, see binning.hip) and the longest per-tile list sorted inside LDS
bool gs4d_sort_tile_first();
// GS4D_SORT=direct / tile: counted placement into the tile lists (csrc/tile_place.h) always / never; unset: per forward,
// decided by what the previous forward of the thread measured.  -1 = not forced, 0 = never, 1 = always.
int gs4d_sort_direct_forced();
// csrc/api.hip: will the calling thread's next forward bin by direct placement?  (the composite call clears the per-tile
// counters in its activation launch then: gs4d_rasterize_forward_spec's `speculate` bit 1)
bool gs4d_forward_bins_directly();
// csrc/elementwise.hip: gs4d_l1_loss_forward that also writes d(g_scale[0] * loss)/da into d_a (both null: the plain forward)
int gs4d_l1_loss_forward_grad(size_t n, const float* a, const float* b, float* scratch, float* loss, const float* g_scale,
                              float* d_a, void* stream, int skip_final);     // skip_final: only the partial sums, loss is not written
// csrc/elementwise.hip: gs4d_activate_backward whose launch also finishes the L1 loss from the partial sums (l1_loss may be null)
int gs4d_activate_backward_loss(int P, const float* scales, const float* rotations, const float* opacity,
                                const float* g_scales, const float* g_rotations_n, const float* g_opacity,
                                float* d_log_scales, float* d_rotations, float* d_opacity_logit, const float* l1_partial,
                                size_t l1_n, float* l1_loss, void* stream);
// csrc/elementwise.hip: gs4d_activate_forward that also zeroes clear[0 .. clear_words)
int gs4d_activate_forward_clear(int P, const float* log_scales, const float* rotations, const float* opacity_logit,
                                float* scales, float* rotations_n, float* opacity, uint32_t* clear, size_t clear_words,
                                void* stream);
uint32_t gs4d_sort_segment_cap();
uint32_t gs4d_sort_wave_cap();
int gs4d_radix_sort_pairs(uint32_t R, int passes, int first_shift, const uint64_t* keys_in, const uint32_t* vals_in,
                          uint64_t* const keys_tmp[2], uint32_t* const vals_tmp[2], uint32_t* hist, int* where,
                          int debug, hipStream_t s, const uint32_t* R_dev = nullptr);

// ---- stage launchers (each returns GS4D_OK or an error) ----
int gs4d_launch_preprocess(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                           const float* rotations, const float* opacities, const float* shs,
                           const float* cov3D_precomp, const float* colors_precomp, const float* viewmatrix,
                           const float* projmatrix, const float* cam_pos, int W, int H, float tan_fovx,
                           float tan_fovy, int prefiltered, int* radii, GeomState& g, int debug, hipStream_t s,
                           uint32_t* tile_count = nullptr, bool tile_count_cleared = false, uint8_t* visible = nullptr);
int gs4d_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);
// csrc/api.hip: the rasterizer entry points with the speculative binning launch / an explicit binning layout (csrc/render.hip)
int gs4d_rasterize_forward_spec(gs4d_alloc_fn geom_alloc, void* geom_ctx, gs4d_alloc_fn binning_alloc, void* binning_ctx,
                                gs4d_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                                int width, int height, const float* means3D, const float* shs,
                                const float* colors_precomp, const float* opacities, const float* scales,
                                float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                                float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii, int debug_arg,
                                void* stream, int* num_rendered, int speculate, int* layout_rendered, uint8_t* visible = nullptr);
// The activation backward of the composite call (csrc/render.hip), applied by the per-Gaussian backward itself right after
// it has written a Gaussian's dL_dscale / dL_drot / dL_dopacity: what gs4d_activate_backward does in a launch of its own
// (same device function: activation_backward_point, gs4d_math.h).  l1_loss != null: workgroup 0 of the deferred-Gaussian
// launch also finishes the L1 loss (GS4D_LOSS_AFTER_BACKWARD).
struct Gs4dActBwd {
    const float* scales_a; const float* rot_raw; const float* opacity_a;
    float* d_log_scales; float* d_rot; float* d_opacity_logit;
    const float* l1_partial; size_t l1_n; float* l1_loss;
};
#define GS4D_L1_BLOCKS 512             // partial sums of the L1 loss (csrc/elementwise.hip)
int gs4d_rasterize_backward_layout(gs4d_alloc_fn scratch_alloc, void* scratch_ctx, int P, int D, int M, int R,
                                   const float* background, int width, int height, const float* means3D, const float* shs,
                                   const float* colors_precomp, const float* scales, float scale_modifier,
                                   const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                   const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                                   const int* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
                                   const float* dL_dpix, const float* dL_dpix_depth, float* dL_dmean2D, float* dL_dconic,
                                   float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
                                   float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug,
                                   void* stream, int layout_rendered, const Gs4dActBwd* act = nullptr);
int gs4d_launch_binning(int P, int R, int W, int H, const int* radii, GeomState& g, BinningState& b, ImageState& im,
                        int debug, hipStream_t s, const uint32_t* R_dev = nullptr, bool direct = false);
// the deferred-Gaussian counter of the backward scratch (layout: gs4d_backward_scratch_bytes)
static inline uint32_t* gs4d_backward_big_count(const float* inst_grad, int R)
{
    return (uint32_t*)((char*)inst_grad + gs4d_align_up((size_t)R * GS4D_GRAD_ROW * sizeof(float), 256));
}
bool gs4d_geom_backward_fills_dsh(int M, const float* shs, const float* dL_dsh);     // csrc/geom_bwd.hip
int gs4d_launch_blend_forward(int W, int H, const GeomState& g, const BinningState& b, ImageState& im,
                              const float* features, const float* bg, float* out_color, float* out_depth, int checkpoints,
                              int debug, hipStream_t s);
// clear_word (may be null): one device word the kernel also sets to zero -- the backward's deferred-Gaussian counter,
// which would otherwise cost a memset node of its own in front of the per-Gaussian kernel
int gs4d_launch_blend_backward(int W, int H, int R, const int* radii, const GeomState& g, const BinningState& b,
                               const ImageState& im, const float* features, const float* bg, const float* dL_dpix,
                               const float* dL_dpix_depth, float* inst_grad, uint32_t* clear_word, int debug, hipStream_t s);
int gs4d_launch_geom_backward(int P, int D, int M, int R, int W, int H, const float* means3D, const int* radii,
                              const float* shs, const float* scales, const float* rotations, float scale_modifier,
                              const float* cov3D, const float* viewmatrix, const float* projmatrix,
                              float tan_fovx, float tan_fovy, const float* campos, const GeomState& g,
                              const BinningState& b, const float* inst_grad, float* dL_dmean2D, float* dL_dconic,
                              float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
                              float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug,
                              hipStream_t s, const Gs4dActBwd* act = nullptr);

// ---- kernel experiments only (-DGS4D_TIMELINE, tools/build_timeline_lib.sh -> libgs4d_tl.so, tools/*_timeline.py): a
// kernel records per workgroup its start / end on the 100 MHz clock, its placement and two counters of its choice
#ifdef GS4D_TIMELINE
#define GS4D_TIMELINE_DEFINE(name)                                                                                      \
    namespace { __device__ unsigned long long g_timeline[4 * 65536]; }                                                  \
    extern "C" __attribute__((visibility("default"))) int gs4d_debug_timeline_read_##name(void* host, size_t bytes)    \
    {                                                                                                                   \
        return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timeline), bytes < sizeof(g_timeline) ? bytes : sizeof(g_timeline)); \
    }
namespace gs4d {
struct TimelineScope {
    unsigned long long* buf; unsigned long long t0; int slot; unsigned visits = 0, n = 0;
#ifdef GS4D_TL_CLOCK
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
    __device__ TimelineScope(unsigned long long* b, int slot_) : buf(b), t0(wall_clock64()), slot(slot_) {}
    __device__ ~TimelineScope()
    {
        if (threadIdx.x == 0 && slot < 65536) {
            buf[4 * slot + 0] = t0;
            buf[4 * slot + 1] = wall_clock64();
#ifdef GS4D_TL_CLOCK
            buf[4 * slot + 2] = __builtin_amdgcn_s_memtime() - c0;      // shader-clock cycles instead of the placement
#else
            buf[4 * slot + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
            buf[4 * slot + 3] = ((unsigned long long)n << 32) | visits;
        }
    }
};
}
#else
#define GS4D_TIMELINE_DEFINE(name)
#endif
