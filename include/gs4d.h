/*
 * gs4d.h -- C ABI of libgs4d.so, the MI355X (gfx950) renderer for the 4D-Gaussian-Splatting
 * render path.  Plain pointers and sizes only (no torch types); every pointer marked
 * "device" is HIP device memory on the current device; `stream` is a hipStream_t passed
 * as void* (NULL = the default stream).  Every function returns 0 on success or a negative
 * GS4D_E* code; gs4d_last_error() returns a thread-local description of the last failure.
 *
 * The entry points mirror, one to one, the C++ API that the reference's torch glue binds
 * (paths relative to the reference repo; RAST = submodules/depth-diff-gaussian-rasterization):
 *
 *   gs4d_rasterize_forward    <->  CudaRasterizer::Rasterizer::forward     RAST/cuda_rasterizer/rasterizer.h:31-54
 *   gs4d_rasterize_backward   <->  CudaRasterizer::Rasterizer::backward    RAST/cuda_rasterizer/rasterizer.h:56-87
 *   gs4d_mark_visible         <->  CudaRasterizer::Rasterizer::markVisible RAST/cuda_rasterizer/rasterizer.h:24-29
 *   gs4d_knn_mean_dist2       <->  SimpleKNN::knn                          submodules/simple-knn/simple_knn.h:15-19
 *   gs4d_alloc_fn             <->  std::function<char*(size_t)> resize lambdas, RAST/rasterize_points.cu:27-33
 *   gs4d_deform_forward/backward <-> deform_network.forward_dynamic (PyTorch in the reference):
 *                                  scene/hexplane.py:100-131,192-215 + scene/deformation.py:128-200,265-298
 *
 * The scratch buffers (geometry / binning / image) are opaque byte chunks owned by the
 * caller exactly as in the reference (they are only round-tripped from forward to
 * backward); gs4d_state_view() exposes the arrays inside them for bit-exact checks.
 */
#ifndef GS4D_H_INCLUDED
#define GS4D_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS4D_OK 0
#define GS4D_EINVAL (-1)   /* bad argument (shape / null pointer / unsupported combination) */
#define GS4D_EHIP (-2)     /* a HIP runtime call or kernel launch failed */
#define GS4D_EALLOC (-3)   /* the caller's allocator returned NULL */
#define GS4D_EPREFILTER (-4) /* `prefiltered` was set but a Gaussian failed the near-plane test (reference: device trap, auxiliary.h:156-160) */

/* Caller-provided scratch allocator: must return device memory of at least `bytes` bytes
 * (256-byte aligned) that stays valid until the caller releases it. */
typedef void* (*gs4d_alloc_fn)(void* ctx, size_t bytes);

const char* gs4d_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int gs4d_abi_version(void);

/*
 * Per-stage device timing, for bench.py's roofline figures (no counterpart in the reference, which
 * has no profiling hooks).  While enabled, every kernel stage is bracketed by HIP events on the
 * stream it is launched on; gs4d_profile_read() waits for the recorded events and returns the
 * accumulated milliseconds and launch count of one stage ("preprocess", "scan", "emit_keys",
 * "radix_sort", "tile_ranges", "blend_forward", "blend_backward", "geom_backward",
 * "hexplane_forward", "hexplane_backward", "mlp_forward", "mlp_backward", "knn", ...).
 * Every recorded event costs device idle time (a barrier packet each), so a throughput run should
 * time ONE stage: gs4d_profile_select("stage") restricts recording to it, NULL or "" selects all.
 */
void gs4d_profile_enable(int on);
void gs4d_profile_select(const char* stage);
void gs4d_profile_reset(void);
int gs4d_profile_read(const char* stage, double* total_ms, long* calls);

/*
 * Forward render.  Replaces Rasterizer::forward (rasterizer.h:31-54; implementation
 * rasterizer_impl.cu:198-339).  Argument meaning is identical:
 *   P Gaussians, D active SH degree, M SH coefficients per Gaussian (0 if shs == NULL);
 *   exactly one of (shs, colors_precomp) and one of ((scales, rotations), cov3D_precomp)
 *   is non-NULL; viewmatrix / projmatrix are the 16-float transposed matrices of
 *   scene/cameras.py:103-105.  out_color [3,H,W], out_depth [1,H,W] and radii [P] are
 *   written in full when P != 0 (every pixel by the blend, every radius by the preprocess),
 *   so the caller need not clear them; with P == 0 the call returns at once and leaves them
 *   untouched (the reference's caller hands over zeros, rasterize_points.cu:68-70).
 * *num_rendered receives the number of (Gaussian, tile) instances.  One host
 * wait on `stream` happens inside (to size the binning buffer), as in
 * rasterizer_impl.cu:282; see gs4d_thread_state_release() below for how it is done.
 * `debug`: bit 0 as in the reference (synchronise and check after every launch); bit 1
 * (GS4D_FORWARD_ONLY, no reference counterpart) promises that the buffers of this call will not be
 * handed to gs4d_rasterize_backward: the blend then leaves out the per-segment pixel checkpoints
 * (20 bytes per pixel and 128 list entries) from which the backward walks a long tile list in parallel.
 * A backward on buffers of such a call reads undefined checkpoints for tiles with more than 128 entries.
 */
#define GS4D_FORWARD_ONLY 2
#define GS4D_LOSS_AFTER_BACKWARD 4   /* gs4d_render_desc::flags only */
int gs4d_rasterize_forward(
    gs4d_alloc_fn geom_alloc, void* geom_ctx,
    gs4d_alloc_fn binning_alloc, void* binning_ctx,
    gs4d_alloc_fn image_alloc, void* image_ctx,
    int P, int D, int M,
    const float* background,          /* device [3] */
    int width, int height,
    const float* means3D,             /* device [P,3] */
    const float* shs,                 /* device [P,M,3] or NULL */
    const float* colors_precomp,      /* device [P,3] or NULL */
    const float* opacities,           /* device [P] */
    const float* scales,              /* device [P,3] or NULL */
    float scale_modifier,
    const float* rotations,           /* device [P,4] or NULL */
    const float* cov3D_precomp,       /* device [P,6] or NULL */
    const float* viewmatrix,          /* device [16] */
    const float* projmatrix,          /* device [16] */
    const float* cam_pos,             /* device [3] */
    float tan_fovx, float tan_fovy,
    int prefiltered,
    float* out_color, float* out_depth, int* radii,
    int debug,
    void* stream,
    int* num_rendered);

/*
 * Backward.  Replaces Rasterizer::backward (rasterizer.h:56-87; rasterizer_impl.cu:343-444).
 * R is the num_rendered the forward returned; geom/binning/img buffers are the chunks the
 * forward filled.  All dL_d* outputs except dL_dmean2D must be zero-initialised by the caller
 * (rasterize_points.cu:154-163; dL_dmean2D is written in full whenever P > 0 and R > 0, otherwise the call
 * returns at once and writes nothing): dL_dmean2D [P,3], dL_dconic [P,4], dL_dopacity [P],
 * dL_dcolor [P,3], dL_ddepth [P], dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3],
 * dL_dscale [P,3], dL_drot [P,4].
 * `scratch_alloc` provides the per-instance gradient rows (48 bytes x R) that replace the
 * reference's per-pixel global atomics; results are bitwise reproducible run to run.
 */
int gs4d_rasterize_backward(
    gs4d_alloc_fn scratch_alloc, void* scratch_ctx,
    int P, int D, int M, int R,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy,
    const int* radii,
    char* geom_buffer, char* binning_buffer, char* img_buffer,
    const float* dL_dpix,             /* device [3,H,W] */
    const float* dL_dpix_depth,       /* device [1,H,W] */
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
    int debug,
    void* stream);

/*
 * Per-thread cached state of gs4d_rasterize_forward (no counterpart in the reference, whose forward pays a blocking
 * cudaMemcpy instead, rasterizer_impl.cu:282): every calling thread lazily owns 64 bytes of pinned, coherent host
 * memory -- the mailbox into which the scan kernel posts num_rendered -- plus the previous call's num_rendered, used
 * to request the binning chunk before the count is known.  Nothing is shared between threads.  The host polls the
 * mailbox for at most GS4D_MAILBOX_SPIN_US microseconds (environment, default 5000) and then falls back to
 * hipMemcpyAsync + hipStreamSynchronize; GS4D_NO_MAILBOX=1 always takes that path.
 * gs4d_thread_state_release() frees the calling thread's mailbox (call it before the thread exits or the library is
 * unloaded; a later forward simply allocates a new one).  gs4d_mailbox_stats() reports how often the calling thread
 * received the count through the mailbox (*fast) and through the copy + synchronise path (*slow).
 */
void gs4d_thread_state_release(void);
void gs4d_mailbox_stats(unsigned long long* fast, unsigned long long* slow);
/* The same two counters plus *regrow: forwards of the calling thread whose speculative binning chunk (previous
 * num_rendered + 1/8) turned out too small, i.e. that called binning_alloc a second time. */
void gs4d_forward_stats(unsigned long long* fast, unsigned long long* slow, unsigned long long* regrow);
/* gs4d_render_forward calls of this thread whose binning + blend were issued before num_rendered had been read, and how
 * many of those had to be re-issued because the count exceeded the speculative capacity (no reference counterpart).
 * GS4D_NO_SPECULATION=1 switches the speculative launch off. */
void gs4d_speculation_stats(unsigned long long* issued, unsigned long long* reissued);
/* Forwards of the calling thread whose (Gaussian, tile) instances were binned by direct placement (*direct: counted per
 * tile during preprocessing and written straight into the tile lists, csrc/tile_place.h) and by the two stable radix
 * passes on the tile bits (*passes).  The lists are the same bit for bit; which one runs is decided per forward from
 * two numbers of the thread's PREVIOUS forward: num_rendered and the device-scope atomic adds direct placement needs per
 * phase (one per tile that the Gaussians of a 256-row workgroup touch together: few when the rows of the model are in a
 * spatially coherent order; exact when that forward placed directly, an upper bound from the workgroups' tile
 * rectangles otherwise).  Direct placement runs when num_rendered + 7 adds <= 2 400 000 (the measured costs of the two
 * ways on MI355X, csrc/binning.hip); a thread's first forward takes the passes.  GS4D_SORT=direct / tile forces one of
 * them, GS4D_SORT=lsd selects the six-pass sort of the whole key (no reference counterpart: the reference always runs
 * the device-wide radix sort, rasterizer_impl.cu:301-309).  *last_adds / *last_rendered: those two numbers of the
 * thread's last forward.  Any pointer may be null. */
void gs4d_binning_stats(unsigned long long* direct, unsigned long long* passes, unsigned long long* last_adds,
                        unsigned long long* last_rendered);

/* present[i] = (view-space z of means3D[i]) > 0.2 ; rasterizer_impl.cu:54-66,141-153. `present` is device uint8/bool [P]. */
int gs4d_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                      uint8_t* present, void* stream);

/* Views into the opaque scratch chunks (device pointers), for bit-exact checks against the oracle. */
typedef struct gs4d_state_view {
    /* geometry chunk, P entries each */
    const float* depths;
    const uint8_t* clamped;         /* [P,3] */
    const float* means2D;           /* [P,2] */
    const float* cov3D;             /* [P,6] */
    const float* conic_opacity;     /* [P,4] */
    const float* rgb;               /* [P,3] */
    const uint32_t* tiles_touched;
    const uint32_t* point_offsets;  /* inclusive prefix sum */
    /* binning chunk, R entries each */
    const uint64_t* keys_unsorted;
    const uint32_t* values_unsorted;
    const uint64_t* keys_sorted;
    const uint32_t* point_list;
    /* image chunk */
    const uint32_t* ranges;         /* [tiles,2] */
    const float* final_T;           /* [H*W] */
    const uint32_t* n_contrib;      /* [H*W] */
} gs4d_state_view;

int gs4d_state_view_get(int P, int R, int width, int height,
                        char* geom_buffer, char* binning_buffer, char* img_buffer, gs4d_state_view* out);

/* Sizes of the three chunks (what the allocator callbacks will be asked for). */
size_t gs4d_geometry_bytes(int P);
size_t gs4d_binning_bytes(int R);
size_t gs4d_binning_bytes_forward_only(int R);   /* a GS4D_FORWARD_ONLY call: no segment checkpoints are reserved */
size_t gs4d_image_bytes(int width, int height);
size_t gs4d_backward_scratch_bytes(int R);

/*
 * distCUDA2: mean_dist2[i] = mean of the 3 smallest squared distances from points[i] to the
 * other points.  Replaces SimpleKNN::knn (simple_knn.h:15-19, simple_knn.cu:185-221).
 * points: device [P,3] float32, mean_dist2: device [P].  Scratch comes from `alloc`.
 */
int gs4d_knn_mean_dist2(gs4d_alloc_fn alloc, void* alloc_ctx, int P, const float* points, float* mean_dist2,
                        void* stream);

/*
 * gs4d_knn_mean_dist2 for callers that cannot allocate.  NO counterpart in the reference: it exists for callers that
 * must not call hipMalloc (or any allocator) -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid allocation.  Same inputs, outputs and results (bit for bit) as gs4d_knn_mean_dist2, but the pipeline's scratch
 * (Morton keys, sort ping-pong buffers and histograms, sorted points, boxes: 52 bytes per point) lives in a file-scope
 * device array of 16 MiB inside the library, so at most gs4d_knn_static_max_points() = 262 144 points are accepted
 * (more: GS4D_EINVAL with a clear message; the reference calls distCUDA2 with 2 000 points).  The call makes no
 * hipMalloc / hipFree / allocator call; the caller still provides `points` and `mean_dist2`.  Every GPU the library
 * runs on holds its own 16 MiB copy until the process ends; the array's device address is looked up once per device
 * (hipGetSymbolAddress, on the first call with that device current).  All calls on one GPU share that array: calls that
 * overlap on one GPU (different streams or threads) RACE -- serialise them.  Nothing in the scratch has to start at
 * zero (every array is written in full before it is read), so stale contents of an earlier, larger call are harmless.
 */
int gs4d_knn_static_max_points(void);
/* Debug accessor (no reference counterpart): how many times, process-wide, the library has asked for memory so far -- a
 * caller's allocator callback or a runtime allocation of its own (the pinned num_rendered mailbox is the only one).
 * The counter does not move across gs4d_knn_mean_dist2_static (tests/test_zz_no_alloc.py). */
unsigned long long gs4d_debug_alloc_calls(void);
int gs4d_knn_mean_dist2_static(int P, const float* points, float* mean_dist2, void* stream);

/*
 * k-nearest-neighbour query of the fork's rigidity regulariser.  Replaces
 * knnquery_cuda for ONE batch segment: pointops2/src/knnquery/knnquery_cuda_kernel.cu:65-108 (launcher :111-116,
 * Python KNNQuery pointops2/functions/pointops.py:34-49, caller utils/general_utils.py:144-158).
 * xyz [n,3], new_xyz [m,3] (NULL = query the points themselves) -> idx int32 [m,nsample] (indices into xyz),
 * dist2 [m,nsample] squared distances, ascending; among points at exactly the k-th distance the lower indices are
 * kept (the reference's choice there depends on its heap's history); if n < nsample the remaining slots hold
 * (index 0, 1e10) as the reference leaves them.  1 <= nsample <= 128.  Distances are exact.
 */
int gs4d_knn_query(gs4d_alloc_fn alloc, void* alloc_ctx, int n, const float* xyz, int m, const float* new_xyz,
                   int nsample, int* idx, float* dist2, void* stream);
/* The same query; point_order [n] int32 (may be NULL) receives the indices of the n points of xyz in the order of their
 * Morton codes, which the query builds anyway: rows visited in that order touch neighbouring points' records while they
 * are still in cache (gs4d_temporal_regulation_* take it as row_order).  No counterpart in the reference. */
int gs4d_knn_query_order(gs4d_alloc_fn alloc, void* alloc_ctx, int n, const float* xyz, int m, const float* new_xyz,
                         int nsample, int* idx, float* dist2, int* point_order, void* stream);

/*
 * HexPlane deformation field.  The reference implements it in PyTorch
 * (scene/hexplane.py:100-131,192-215 gathers; scene/deformation.py:128-200,265-298 trunk, heads,
 * positional encodings; utils/graphics_utils.py:110-133 quaternion update), so there is no native
 * interface to mirror: these two entry points bind the whole of
 * deform_network.forward_dynamic(points, scales, rotations, opacity, shs, time, force, None)
 * and its backward.  Only the path that runs in the reference is supported:
 * blend_time_force = True, no_do = True (opacity passes through), prev_frames = None, net_width = 64.
 *
 * All pointers are device pointers.  Weights use the reference's nn.Linear layout [out][in]
 * row-major.  Plane l*6+i is the reference's grids[l][i] (pairs xy,xz,xt,yz,yt,zt) stored
 * CHANNELS-LAST: [height][width][C] (torch.channels_last of the reference's [1,C,H,W] shape),
 * width = resolution of the pair's first axis, height = of its second.
 */
#define GS4D_MAX_LEVELS 4
#define GS4D_MAX_TRUNK 8
typedef struct gs4d_deform_desc {
    int C;                    /* channels per plane (4..256, power of two) */
    int L;                    /* multi-resolution levels */
    int trunk_layers;         /* max(defor_depth, 1): Linear(L*C,64) then (ReLU, Linear(64,64)) ... */
    int pos_pe, sr_pe, time_pe, force_pe;   /* numbers of PE frequencies */
    int no_dx, no_ds, no_dr, no_dshs, apply_rotation, use_time, use_force;
    const float* planes[GS4D_MAX_LEVELS * 6];
    int plane_w[GS4D_MAX_LEVELS * 6];
    int plane_h[GS4D_MAX_LEVELS * 6];
    const float* aabb;        /* [2][3]: row 0 = xyz_max, row 1 = xyz_min (scene/hexplane.py:188-190) */
    const float* ft_w;        /* force_time_embed.weight [1][2] */
    const float* ft_b;        /* force_time_embed.bias [1] */
    const float* trunk_w[GS4D_MAX_TRUNK];
    const float* trunk_b[GS4D_MAX_TRUNK];
    const float* head_w1[4];  /* pos, scales, rotations, shs: first Linear [64][in_h] */
    const float* head_b1[4];
    const float* head_w2[4];  /* second Linear [out_h][64], out = 3,3,4,48 */
    const float* head_b2[4];
    int no_grad;              /* 1: no backward will follow (render_for_image): the forward skips the activations only the
                                 backward reads (the workspace then cannot be passed to gs4d_deform_backward) */
} gs4d_deform_desc;

/* Gradient destinations, same shapes as the parameters above; every one must be zero-initialised
 * by the caller (plane gradients are accumulated with float atomics). */
typedef struct gs4d_deform_grads {
    float* planes[GS4D_MAX_LEVELS * 6];
    float* ft_w;
    float* ft_b;
    float* trunk_w[GS4D_MAX_TRUNK];
    float* trunk_b[GS4D_MAX_TRUNK];
    float* head_w1[4];
    float* head_b1[4];
    float* head_w2[4];
    float* head_b2[4];
    /* Optional cache of the plane-gradient pipeline's point order (no reference counterpart; NULL: off).  The backward
     * sorts the points by plane cell only to COMBINE gradient adds before they reach memory -- its results do not depend
     * on that order -- and the points of a scene move little from one iteration to the next.  hex_order: device
     * [L * 3 * P] ints owned by the caller.  hex_order_valid = 0: the backward sorts as usual and leaves the order here;
     * 1: it skips the ranking atomics, the bin scan and the scatter and walks the points in the cached order (same P
     * and L as when it was recorded; a stale order only groups a little less well).  A scene large enough for the
     * pipeline to run in several batches (GS4D_HEX_BATCH_BYTES) keeps one [L * 3 * n] block of batch-local indices
     * per batch, batch after batch: same total size; the environment variable must not change between the recording
     * call and the calls that reuse the order. */
    int* hex_order;
    int hex_order_valid;
    int reserved_;
} gs4d_deform_grads;

/* Bytes of the activation workspace the forward fills and the backward re-reads, and of the
 * backward's own scratch. */
size_t gs4d_deform_workspace_bytes(const gs4d_deform_desc* d, int P);
size_t gs4d_deform_backward_scratch_bytes(const gs4d_deform_desc* d, int P);

/* points [P,3], scales [P,3], rotations [P,4], shs [P,16,3], time [P], force [P] ->
 * out_means3D [P,3], out_scales [P,3], out_rotations [P,4], out_shs [P,16,3].
 * rotations, shs, out_rotations and out_shs must be 16-byte aligned (float4 accesses).
 * workspace: gs4d_deform_workspace_bytes(), 256-byte aligned, kept by the caller for the backward. */
int gs4d_deform_forward(const gs4d_deform_desc* d, int P,
                        const float* points, const float* scales, const float* rotations, const float* shs,
                        const float* time, const float* force,
                        float* out_means3D, float* out_scales, float* out_rotations, float* out_shs,
                        char* workspace, void* stream);

/* dL_d(out_*) -> dL_d(points, scales, rotations, shs) (written, not accumulated) and parameter
 * gradients accumulated into `g`.  scratch: gs4d_deform_backward_scratch_bytes().
 * dL_dshs_in may be the SAME pointer as dL_dshs: the SH coefficients only receive a residual, so their input gradient
 * IS the upstream gradient and nothing is copied then (192 bytes per point less to write). */
int gs4d_deform_backward(const gs4d_deform_desc* d, const gs4d_deform_grads* g, int P,
                         const float* points, const float* scales, const float* rotations,
                         const float* time, const float* force,
                         const float* dL_dmeans3D, const float* dL_dscales, const float* dL_drotations,
                         const float* dL_dshs,
                         float* dL_dpoints, float* dL_dscales_in, float* dL_drotations_in, float* dL_dshs_in,
                         char* workspace, char* scratch, void* stream);

/* The same backward in two phases, for data-parallel callers that want to start exchanging gradients early
 * (no counterpart in the reference, which has no distributed code; gs4d_deform_backward == both phases):
 *   GS4D_DEFORM_BWD_DATA     dL_d(points, scales, rotations, shs), the plane gradients and the gradients of
 *                            force_time_embed (g->planes, g->ft_w, g->ft_b): final when this phase returns
 *   GS4D_DEFORM_BWD_WEIGHTS  the Linear weight / bias gradients (g->trunk_*, g->head_*); must follow the data
 *                            phase of the same call arguments, with `scratch` and `workspace` untouched between. */
#define GS4D_DEFORM_BWD_DATA 1
#define GS4D_DEFORM_BWD_WEIGHTS 2
int gs4d_deform_backward_phases(const gs4d_deform_desc* d, const gs4d_deform_grads* g, int P,
                                const float* points, const float* scales, const float* rotations,
                                const float* time, const float* force,
                                const float* dL_dmeans3D, const float* dL_dscales, const float* dL_drotations,
                                const float* dL_dshs,
                                float* dL_dpoints, float* dL_dscales_in, float* dL_drotations_in, float* dL_dshs_in,
                                char* workspace, char* scratch, void* stream, int phases);

/* ---- activations + photometric loss (the framework ops between the three kernels groups) ----
 * Replaces, for the hot path, the torch ops of scene/gaussian_model.py:38-43 applied at
 * gaussian_renderer/renderer.py:170-172 (scales = exp, rotations = F.normalize, opacity = sigmoid)
 * and utils/loss_utils.py:20-21 (l1_loss = mean |a - b|), each as one fused kernel per direction.
 * rotations pointers must be 16-byte aligned.  Upstream gradients of the backward may be NULL
 * (treated as zero).  The backward reads the forward OUTPUTS scales / opacity and the raw rotations. */
int gs4d_activate_forward(int P, const float* log_scales, const float* rotations, const float* opacity_logit,
                          float* scales, float* rotations_n, float* opacity, void* stream);
int gs4d_activate_backward(int P, const float* scales, const float* rotations, const float* opacity,
                           const float* g_scales, const float* g_rotations_n, const float* g_opacity,
                           float* d_log_scales, float* d_rotations, float* d_opacity_logit, void* stream);

/* loss[0] = mean |a[i] - b[i]| over n floats (deterministic two-pass reduction; scratch of
 * gs4d_l1_loss_scratch_bytes()); backward writes d_a[i] = sign(a[i]-b[i]) * g_loss[0] / n.
 * a, b and d_a must be 16-byte aligned (float4 accesses). */
size_t gs4d_l1_loss_scratch_bytes(void);
int gs4d_l1_loss_forward(size_t n, const float* a, const float* b, float* scratch, float* loss, void* stream);
int gs4d_l1_loss_backward(size_t n, const float* a, const float* b, const float* g_loss, float* d_a, void* stream);

/* ---- the whole hot path of one view in one call per direction ----
 * No single counterpart in the reference, which composes these stages in Python: gaussian_renderer/renderer.py:123-147
 * (deformation query, then scales = exp, rotations = normalize, opacity = sigmoid), :207-216 (rasterizer call) and
 * train.py:203 / utils/loss_utils.py:20-21 (L1 loss).  gs4d_render_forward = gs4d_deform_forward ->
 * gs4d_activate_forward -> gs4d_rasterize_forward [-> gs4d_l1_loss_forward]; gs4d_render_backward = the reverse chain.
 * Nothing else runs on the device, so the results are bit-identical to calling the stages one by one; what goes away
 * is the host time between them (one call, three allocator callbacks per direction instead of ~30 tensor
 * allocations and four framework-level autograd nodes).  Requirements: P > 0, SH colours (no colors_precomp /
 * cov3D_precomp), M == 16 when a deformation field is given; other cases use the per-stage entry points. */
typedef struct gs4d_render_desc {
    int P, D, M;                  /* Gaussians, active SH degree, SH coefficients per Gaussian */
    int width, height;
    float scale_modifier, tan_fovx, tan_fovy;
    int prefiltered;
    int flags;                    /* as gs4d_rasterize_forward's `debug`: bit 0 debug, GS4D_FORWARD_ONLY; and
                                     GS4D_LOSS_AFTER_BACKWARD (4, with a target): the caller promises that gs4d_render_backward
                                     follows (a training step) and reads loss[0] only after it -- the forward then leaves the
                                     loss as partial sums and a launch of the backward call finishes it, same value */
    const float* background;      /* device [3] */
    const float* viewmatrix;      /* device [16] */
    const float* projmatrix;      /* device [16] */
    const float* cam_pos;         /* device [3] */
    const float* xyz;             /* device [P,3] */
    const float* log_scales;      /* device [P,3], pre-activation */
    const float* rotations;       /* device [P,4], pre-activation, 16-byte aligned */
    const float* opacity_logit;   /* device [P] */
    const float* shs;             /* device [P,M,3], 16-byte aligned */
    const gs4d_deform_desc* deform;   /* NULL: coarse stage (renderer.py:123-125), the raw parameters are activated and rendered */
    const float* time;            /* device [P]; read only when deform != NULL */
    const float* force;           /* device [P] */
    const float* target;          /* device [3,H,W] (16-byte aligned) or NULL.  With a target the forward also writes
                                     loss[0] = mean |color - target| and the backward may start from g_loss. */
    const float* l1_grad_scale;   /* device [1] or NULL (ABI 13).  With a target: the caller already knows the factor g its
                                     backward will start from (a training step: g = the loss weight), so the forward's loss
                                     launch also leaves d(g * L1)/dcolor = g sign(color - target) / n in the arena, and a
                                     gs4d_render_backward called with g_loss == this pointer (value unchanged) starts from
                                     it without a launch of its own for the loss gradient.  Same numbers either way. */
    unsigned char* visible;       /* device [P] or NULL (ABI 13).  Written as radii[i] > 0: the visibility filter of
                                     gaussian_renderer/renderer.py:221, without a compare launch over radii afterwards. */
} gs4d_render_desc;

/* Filled by the forward, handed back to the backward unchanged.  `arena` (activations, geometry and image chunks) is
 * the first block the forward's allocator returned, `binning` the last one (the binning chunk; the allocator is asked
 * a second time for it only when the speculative size was too small): the caller keeps both alive until the backward
 * has been issued. */
typedef struct gs4d_render_state {
    char* arena;
    size_t arena_bytes;
    char* binning;
    size_t binning_bytes;
    int num_rendered;
    int layout_rendered;          /* the count the binning chunk is laid out for (>= num_rendered: the forward issues
                                     binning and blend before it has read num_rendered, sized for the largest recent count) */
    float* deferred_loss;         /* GS4D_LOSS_AFTER_BACKWARD: where the backward call writes the loss (ABI 13) */
} gs4d_render_state;

/* out_color [3,H,W], out_depth [1,H,W], radii [P]: written in full.  loss: device [1], required with a target.
 * From a thread's second call on, binning and blend are issued before the host has read num_rendered (sized for the
 * largest of the last eight counts + 1/8, the count itself read on the device) and issued again with the exact size
 * only if the count exceeded that capacity: same results, the GPU does not idle for the round trip. */
int gs4d_render_forward(const gs4d_render_desc* r, gs4d_alloc_fn alloc, void* alloc_ctx,
                        float* out_color, float* out_depth, int* radii, float* loss,
                        gs4d_render_state* state, void* stream);

/* Gradient destinations of the backward, all written in full (none needs clearing by the caller) except `deform`,
 * whose arrays follow gs4d_deform_backward's rule (zero-initialised by the caller, accumulated into). */
typedef struct gs4d_render_grads {
    float* dL_dmeans2D;           /* [P,3] screen-space gradient (the densifier's statistic) */
    float* d_xyz;                 /* [P,3] */
    float* d_log_scales;          /* [P,3] */
    float* d_rotations;           /* [P,4], 16-byte aligned */
    float* d_opacity_logit;       /* [P] */
    float* d_shs;                 /* [P,M,3], 16-byte aligned */
    const gs4d_deform_grads* deform;   /* required when the descriptor has a deformation field */
} gs4d_render_grads;

/* Optional progress callback of the backward, called on the calling thread between launches: data-parallel callers
 * start exchanging gradients that are already final while the rest of the backward is still being issued.
 *   GS4D_RENDER_STAGE_RASTER_DONE  d_shs and dL_dmeans2D are final (all launches that write them are in the stream)
 *   GS4D_RENDER_STAGE_DATA_DONE    d_xyz, d_log_scales, d_rotations, d_opacity_logit and the plane / force_time_embed
 *                                  gradients are final; only the Linear weight gradients follow */
#define GS4D_RENDER_STAGE_RASTER_DONE 1
#define GS4D_RENDER_STAGE_DATA_DONE 2
typedef void (*gs4d_render_stage_fn)(void* ctx, int stage);

/* Exactly one of dL_dcolor [3,H,W] and g_loss (device [1], the gradient of the forward's loss; needs the descriptor's
 * target and the forward's out_color) is non-NULL; dL_ddepth [1,H,W] may be NULL (zero).  radii: the forward's.
 * alloc is called once, for the backward's scratch (free again when the stream has run the call). */
int gs4d_render_backward(const gs4d_render_desc* r, const gs4d_render_state* state,
                         const float* out_color, const int* radii,
                         const float* dL_dcolor, const float* dL_ddepth, const float* g_loss,
                         gs4d_alloc_fn alloc, void* alloc_ctx, const gs4d_render_grads* grads,
                         gs4d_render_stage_fn stage_fn, void* stage_ctx, void* stream);

/* ---- optimizer step and grid regularisers (the per-iteration work right after the render path;
 * SURVEY.md section 8(f) rank 2) ----
 *
 * gs4d_adamw_step: torch.optim.AdamW as constructed at scene/gaussian_model.py:206 (decoupled weight decay,
 * no amsgrad) for a list of parameter tensors in ONE launch; the reference calls optimizer.step() at
 * train.py:328.  Per tensor the caller supplies this step's learning rate (the per-group schedules of
 * scene/gaussian_model.py:228-242), weight decay and the two bias corrections 1 - beta^step.
 * All four pointers of a tensor are dense device arrays of `numel` floats, 16-byte aligned. */
typedef struct gs4d_adamw_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    size_t numel;
    double lr;                 /* scalars are doubles, as torch keeps them: 1 - lr*wd, lr / bc1, sqrt(bc2), 1 - beta are */
    double weight_decay;       /* formed in double and rounded to float once, like torch's Python-scalar arithmetic   */
    double bias_correction1;   /* 1 - beta1^step */
    double bias_correction2;   /* 1 - beta2^step */
} gs4d_adamw_tensor;

int gs4d_adamw_step(int ntensors, const gs4d_adamw_tensor* tensors /* host array */, double beta1, double beta2,
                    double eps, void* stream);

/* HexPlane grid regularisers: scene/regulation.py:22-29 (compute_plane_smoothness) and
 * scene/gaussian_model.py:632-706 (_plane_regulation, _time_regulation, _l1_regulation, _l2_regulation,
 * compute_regulation).  A term is one plane (channels-last [H][W][C] memory of the reference's [1,C,H,W]
 * tensor), a kind and a weight; loss[0] = sum_i weight_i * reg_kind_i(plane_i):
 *   SMOOTHNESS  mean over all elements of (t[h+2] - 2 t[h+1] + t[h])^2   (second difference along H)
 *   L1          mean |1 - t|
 *   L2          mean (1 - t)^2
 * The backward ADDS g_loss[0] * weight_i * d reg / d plane to term.grad (same layout as the plane).
 * At most 32 terms per call; scratch: gs4d_plane_reg_scratch_bytes(). */
#define GS4D_REG_SMOOTHNESS 0
#define GS4D_REG_L1 1
#define GS4D_REG_L2 2
typedef struct gs4d_plane_reg_term {
    const float* plane;
    float* grad;              /* may be NULL for the forward */
    int C, W, H;
    int kind;
    float weight;
} gs4d_plane_reg_term;

size_t gs4d_plane_reg_scratch_bytes(void);
int gs4d_plane_reg_forward(int nterms, const gs4d_plane_reg_term* terms /* host array */, float* scratch, float* loss,
                           void* stream);
int gs4d_plane_reg_backward(int nterms, const gs4d_plane_reg_term* terms /* host array */, const float* g_loss,
                            void* stream);

/* ---- densify / clone / split / prune as ONE row gather ----
 * The reference edits the per-Gaussian state tensor by tensor: boolean-mask copies in prune_points / _prune_optimizer
 * (scene/gaussian_model.py:356-389), torch.cat of parameters and of zero-filled Adam moments in cat_tensors_to_optimizer /
 * densification_postfix (:391-432), once for the clones and once more for the splits followed by a prune (:434-498).
 * Here a model edit is one index vector -- output row i takes input row index[i] -- applied to every per-Gaussian
 * array (six parameters, their two Adam moments each, the statistics) in one launch:
 *   dst[i] = 0                                   if fresh[i] and the array has zero_fresh set (Adam moments of new rows)
 *          = tail[i - (n_out - n_tail)]          if i is one of the last n_tail rows and the array has a tail
 *                                                 (positions / log-scales of the split's new Gaussians)
 *          = src[index[i]]                       otherwise
 * Rows are row_words 4-byte words, moved as bit patterns (float32 or int32 alike).  index [n_out] int64 with values in
 * [0, n_src) (NOT range-checked), fresh [n_out] bytes or NULL; `arrays` is a HOST array; all other pointers device. */
typedef struct gs4d_gather_array {
    const void* src;          /* [n_src][row_words] */
    void* dst;                /* [n_out][row_words], must not overlap src */
    const void* tail;         /* [n_tail][row_words] or NULL */
    int row_words;
    int zero_fresh;
} gs4d_gather_array;
int gs4d_gather_rows(long long n_out, const long long* index, const unsigned char* fresh, int n_tail, int narrays,
                     const gs4d_gather_array* arrays, long long n_src, void* stream);

/* ---- rigidity regulariser building block (with gs4d_knn_query) ----
 * d[i][s] = || x[idx[i][s]] - x[i] ||_2: the reference's torch.norm(vel[idx] - vel[None, :, None], p=2, dim=-1)
 * (gaussian_renderer/renderer.py:163-165,194-195,201-202) without the gathered (n, k, dim) intermediates.
 * x [n, dim] float (dim 3 or 4), idx [n, k] int64 rows of neighbour indices into x (as utils/general_utils.py:144-158
 * returns them), d [n, k].  backward: g [n, k] = dL/dd; dx [n, dim] must be zero-filled (or hold a gradient to add
 * to) -- it is accumulated with float atomics; pairs with d == 0 contribute nothing (torch.norm's subgradient). */
int gs4d_neighbor_diff_norm_forward(int n, int k, int dim, const float* x, const long long* idx, float* d, void* stream);
int gs4d_neighbor_diff_norm_backward(int n, int k, int dim, const float* x, const long long* idx, const float* d,
                                     const float* g, float* dx, void* stream);
/* The same backward without atomics, given the transposed neighbour graph: order [n*k] int32 = the pair indices
 * i*k+s sorted by idx[i][s], row_start [n+1] int32 = first position in `order` of the pairs that point at each row;
 * coef_scratch [n*k] floats.  dx is ADDED to (plain stores, one wave per row): deterministic.  n*k < 2^31. */
int gs4d_neighbor_diff_norm_backward_gather(int n, int k, int dim, const float* x, const long long* idx, const float* d,
                                            const float* g, const int* order, const int* row_start, float* coef_scratch,
                                            float* dx, void* stream);

/* The transposed neighbour graph those gather backwards read, built on the device: idx [pairs] int32 with values in
 * [0, n) (pairs = n*k for one graph; the lists as gs4d_knn_query writes them) -> order [pairs] = the pair indices sorted by
 * idx[pair], ascending pair index inside one target (a stable sort: the result is unique), row_start [n+1] = first position
 * in `order` of each target's pairs (row_start[n] = pairs).  The reference scatters through autograd's index_put
 * (gaussian_renderer/renderer.py:163-165,194-203: vel[idx] / rot[idx]); there is no counterpart call.  Indices are NOT
 * range-checked (an index >= n sorts somewhere, writes nothing out of bounds).  scratch: 256-byte aligned,
 * gs4d_transpose_graph_scratch_bytes(n, pairs) bytes, free again when the stream has run the call.  1 <= pairs < 2^31. */
size_t gs4d_transpose_graph_scratch_bytes(int n, long long pairs);
int gs4d_transpose_graph(int n, long long pairs, const int* idx, int* order, int* row_start, void* scratch,
                         size_t scratch_bytes, void* stream);

/* The three kNN terms of the fine stage's temporal branch (gaussian_renderer/renderer.py:189-203) fused: forward ->
 * sums4 = (S_xyz, S_long, S_rot, S_w) with w = exp(-100 dist) (definitions in csrc/rigidity.hip; knn_xyz = S_xyz/k/n,
 * knn_long = S_long/k/n, knn_rot = S_rot/(n k) * S_w/k/n); backward -> gradients of the four fields given coef3 =
 * (dL/dS_xyz, dL/dS_long, dL/dS_rot) in DEVICE memory and the transposed neighbour graph (order, row_start as for
 * gs4d_neighbor_diff_norm_backward_gather).  vel_* [n,3], rot* [n,4] (16-byte aligned), idx [n,k] int64, dist [n,k];
 * scratch: gs4d_rigidity_terms_scratch_bytes().  The gradients are written (not accumulated), deterministic. */
size_t gs4d_rigidity_terms_scratch_bytes(void);
/* packed: [n][16] floats (16-byte aligned), filled by the forward (one 64-byte record per row) and read by the backward */
int gs4d_rigidity_terms_forward(int n, int k, const float* vel_xyz, const float* vel_long, const float* rot1,
                                const float* rot2, const long long* idx, const float* dist, float* packed, float* scratch,
                                float* sums4, void* stream);
int gs4d_rigidity_terms_backward(int n, int k, const float* packed, const long long* idx, const float* dist,
                                 const int* order, const int* row_start, const float* coef3, float* g_vel_xyz,
                                 float* g_vel_long, float* g_rot1, float* g_rot2, void* stream);
/* The same pair with the neighbour lists as int32, the type gs4d_knn_query writes them in (pointops2's knnquery returns
 * int32 too; utils/general_utils.py:158 widens them with .long()): the render path hands the query's output over without
 * a 40 -> 80 MB conversion per view. */
int gs4d_rigidity_terms_forward_i32(int n, int k, const float* vel_xyz, const float* vel_long, const float* rot1,
                                    const float* rot2, const int* idx, const float* dist, float* packed, float* scratch,
                                    float* sums4, void* stream);
int gs4d_rigidity_terms_backward_i32(int n, int k, const float* packed, const int* idx, const float* dist,
                                     const int* order, const int* row_start, const float* coef3, float* g_vel_xyz,
                                     float* g_vel_long, float* g_rot1, float* g_rot2, void* stream);

/* The whole temporal branch of the fine stage's regulariser dictionary, gaussian_renderer/renderer.py:173-203, in two
 * calls: forward -> out6 = (momentum, scales, shs, knn_xyz, knn_long, knn_rot) on the device; backward -> the gradients of
 * the thirteen inputs given g6 = the upstream gradient of out6 (device).  in13 / grad13 are HOST arrays of 13 device
 * pointers: xyz0 [n,3]; means3D final / prev / next [n,3]; scales final / prev / next [n,3] (activated, :145,:181-182);
 * shs final / prev / next [n, sh_floats] (16-byte aligned, sh_floats a multiple of 4); rotations final (activated, :146) /
 * prev / next [n,4] (16-byte aligned).  idx [n,k] int32 and dist [n,k]: the k nearest neighbours of means3D_final
 * (gs4d_knn_query; dist carries no gradient); order / row_start: the transposed neighbour graph as for
 * gs4d_rigidity_terms_backward.  packed [n][16] floats and scratch (gs4d_temporal_regulation_scratch_bytes(n), 16-byte
 * aligned) are written by the forward and read by the backward.  Gradients are written in full, deterministic.
 * row_order [n] (may be NULL = 0, 1, 2, ...): a PERMUTATION of the rows, the order the backward visits them in; the point
 * order of gs4d_knn_query_order keeps neighbouring rows' records in cache.  It changes no result (each row's sums run in
 * the same order either way). */
size_t gs4d_temporal_regulation_scratch_bytes(int n);
int gs4d_temporal_regulation_forward(int n, int k, int sh_floats, const float* const* in13, const int* idx,
                                     const float* dist, float* packed, float* scratch, float* out6, void* stream);
int gs4d_temporal_regulation_backward(int n, int k, int sh_floats, const float* const* in13, const int* idx,
                                      const float* dist, const int* order, const int* row_start, const int* row_order,
                                      const float* packed, float* scratch, const float* g6, float* const* grad13,
                                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GS4D_H_INCLUDED */
