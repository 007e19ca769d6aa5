// Composite entry points: the whole hot path of one view in ONE call per direction (include/gs4d.h,
// gs4d_render_forward / gs4d_render_backward).
//
// Path composed (reference, Python): gaussian_renderer/renderer.py:123-147,207-216 -- deformation query, the three
// activations (scene/gaussian_model.py:38-43), the rasterizer call -- optionally followed by the L1 photometric loss
// (utils/loss_utils.py:20-21, train.py:203), and the matching backward chain.  The reference issues these as ~40
// framework ops from Python; the staged product path (three autograd functions over the per-stage entry points) still
// paid ~0.5-0.9 ms of host time per step.  Here the host issues every launch of a direction from one C call: the
// caller hands over the raw parameters and gets the image (and loss) back, every intermediate lives in two arenas
// the caller's allocator provides (activations + geometry + image chunk; binning chunk) and the backward's scratch in
// a third.  Nothing new runs on the device: the stages are the per-stage entry points of this library, in the order
// the staged path calls them, so the results are bit-identical to it.
#include <string.h>

#include "gs4d_internal.h"

namespace {

struct Carver {
    char* p;
    template <typename T>
    T* take(size_t count)
    {
        T* out;
        gs4d_obtain(p, out, count);
        return out;
    }
};

struct FwdArena {
    float *m3, *sc, *rot, *shs;          // deformed (pre-activation) parameters; null without a deformation field
    char* ws;                            // deformation workspace
    float *scales_a, *rot_n, *opacity_a; // activated
    char *geom, *image;
    float* l1_scratch;
    float* l1_grad;                      // d(l1_grad_scale * L1)/dcolor, left by the forward's loss launch (desc.l1_grad_scale)
    size_t geom_bytes, image_bytes, bytes;
};

FwdArena carve_forward(const gs4d_render_desc* r, char* base)
{
    FwdArena a = {};
    Carver c{base};
    const size_t P = (size_t)r->P;
    if (r->deform) {
        a.m3 = c.take<float>(3 * P);
        a.sc = c.take<float>(3 * P);
        a.rot = c.take<float>(4 * P);
        a.shs = c.take<float>((size_t)r->M * 3 * P);
        a.ws = c.take<char>(gs4d_deform_workspace_bytes(r->deform, r->P));
    }
    a.scales_a = c.take<float>(3 * P);
    a.rot_n = c.take<float>(4 * P);
    a.opacity_a = c.take<float>(P);
    a.geom_bytes = gs4d_geometry_bytes(r->P);
    a.geom = c.take<char>(a.geom_bytes);
    a.image_bytes = gs4d_image_bytes(r->width, r->height);
    a.image = c.take<char>(a.image_bytes);
    if (r->target) a.l1_scratch = c.take<float>(gs4d_l1_loss_scratch_bytes() / sizeof(float));
    if (r->target && r->l1_grad_scale) a.l1_grad = c.take<float>((size_t)3 * r->width * r->height);
    a.bytes = (size_t)(c.p - base) + 256;
    return a;
}

struct BwdArena {
    float* zero_begin;                   // one memset covers [zero_begin, zero_end)
    float *dL_dmean3D, *dL_dcolor, *dL_ddepth, *dL_dconic, *dL_dopacity, *dL_dcov3D, *dL_dscale, *dL_drot, *zero_depth_img;
    float* zero_end;
    float* g_img;                        // dL/dcolor of the fused L1 loss
    float *d_sc, *d_rot;                 // gradients of the deformed (pre-activation) scales / rotations
    char* inst;                          // per-instance gradient rows of the rasterizer backward
    char* deform_scratch;
    size_t inst_bytes, bytes;
};

BwdArena carve_backward(const gs4d_render_desc* r, int R, bool own_depth_zeros, char* base)
{
    BwdArena a = {};
    Carver c{base};
    const size_t P = (size_t)r->P, HW = (size_t)r->width * r->height;
    a.zero_begin = c.take<float>(0);
    if (r->deform) a.dL_dmean3D = c.take<float>(3 * P);
    a.dL_dcolor = c.take<float>(3 * P);
    a.dL_ddepth = c.take<float>(P);
    a.dL_dconic = c.take<float>(4 * P);
    a.dL_dopacity = c.take<float>(P);
    a.dL_dcov3D = c.take<float>(6 * P);
    a.dL_dscale = c.take<float>(3 * P);
    a.dL_drot = c.take<float>(4 * P);
    if (own_depth_zeros) a.zero_depth_img = c.take<float>(HW);
    a.zero_end = c.take<float>(0);
    if (r->target && !r->l1_grad_scale) a.g_img = c.take<float>(3 * HW);
    if (r->deform) {
        a.d_sc = c.take<float>(3 * P);
        a.d_rot = c.take<float>(4 * P);
        a.deform_scratch = c.take<char>(gs4d_deform_backward_scratch_bytes(r->deform, r->P));
    }
    a.inst_bytes = gs4d_backward_scratch_bytes(R);
    a.inst = c.take<char>(a.inst_bytes);
    a.bytes = (size_t)(c.p - base) + 256;
    return a;
}

// allocator handed to the per-stage entry points: returns a block this call has already carved
struct FixedBlock { char* p; size_t bytes; };
void* fixed_alloc(void* ctx, size_t bytes)
{
    FixedBlock* b = (FixedBlock*)ctx;
    return bytes <= b->bytes ? b->p : nullptr;
}

struct BinningRelay { gs4d_alloc_fn fn; void* ctx; char* last; size_t last_bytes; };
void* binning_alloc(void* ctx, size_t bytes)
{
    BinningRelay* b = (BinningRelay*)ctx;
    b->last = (char*)b->fn(b->ctx, bytes);
    b->last_bytes = bytes;
    return b->last;
}

int check_desc(const gs4d_render_desc* r, const char* who)
{
    if (!r || r->P <= 0 || r->width <= 0 || r->height <= 0 || !r->xyz || !r->log_scales || !r->rotations ||
        !r->opacity_logit || !r->shs || r->M <= 0 || !r->background || !r->viewmatrix || !r->projmatrix || !r->cam_pos) {
        gs4d_set_error("%s: null descriptor field, P <= 0 or empty image (the composite call needs P > 0 and SH colours; "
                       "use the per-stage entry points otherwise)", who);
        return GS4D_EINVAL;
    }
    if (r->deform && (!r->time || !r->force)) {
        gs4d_set_error("%s: a deformation field needs the per-point time and force columns", who);
        return GS4D_EINVAL;
    }
    if (r->deform && r->M != 16) {
        gs4d_set_error("%s: the deformation field moves 16 SH coefficients per Gaussian, got M=%d", who, r->M);
        return GS4D_EINVAL;
    }
    if (((uintptr_t)r->rotations & 15) || ((uintptr_t)r->shs & 15) || (r->target && ((uintptr_t)r->target & 15))) {
        gs4d_set_error("%s: rotations, shs and target must be 16-byte aligned", who);
        return GS4D_EINVAL;
    }
    return GS4D_OK;
}

}  // namespace

extern "C" {

int gs4d_render_forward(const gs4d_render_desc* r, gs4d_alloc_fn alloc, void* alloc_ctx, float* out_color,
                        float* out_depth, int* radii, float* loss, gs4d_render_state* state, void* stream)
{
    GS4D_TRACE_SCOPE("gs4d_render_forward");
    int rc = check_desc(r, "gs4d_render_forward");
    if (rc) return rc;
    if (!alloc || !out_color || !out_depth || !radii || !state || (r->target && !loss) ||
        (r->target && ((uintptr_t)out_color & 15))) {
        gs4d_set_error("gs4d_render_forward: null allocator / output / state (or a misaligned colour image with a target)");
        return GS4D_EINVAL;
    }
    memset(state, 0, sizeof *state);
    const FwdArena lay = carve_forward(r, nullptr);
    gs4d_note_alloc();
    char* base = (char*)alloc(alloc_ctx, lay.bytes);
    if (!base) { gs4d_set_error("gs4d_render_forward: the allocator returned NULL for the activation arena"); return GS4D_EALLOC; }
    if ((uintptr_t)base & 255) { gs4d_set_error("gs4d_render_forward: the arena must be 256-byte aligned"); return GS4D_EINVAL; }
    const FwdArena a = carve_forward(r, base);
    state->arena = base;
    state->arena_bytes = lay.bytes;

    const float *m3 = r->xyz, *sc = r->log_scales, *rot = r->rotations, *shs = r->shs;
    if (r->deform) {
        rc = gs4d_deform_forward(r->deform, r->P, r->xyz, r->log_scales, r->rotations, r->shs, r->time, r->force, a.m3,
                                 a.sc, a.rot, a.shs, a.ws, stream);
        if (rc) return rc;
        m3 = a.m3; sc = a.sc; rot = a.rot; shs = a.shs;
    }
    // a forward that bins by direct placement needs its per-tile counters cleared in front of the preprocess launch: the
    // activation launch does it on the way (the image chunk is part of the arena, so its place is known already)
    uint32_t* tile_counters = nullptr;
    size_t tile_counter_words = 0;
    if (gs4d_forward_bins_directly()) {
        const size_t gx = (r->width + GS4D_TILE_X - 1) / GS4D_TILE_X, gy = (r->height + GS4D_TILE_Y - 1) / GS4D_TILE_Y;
        char* ic = a.image;
        const ImageState im = ImageState::from_chunk(ic, (size_t)r->width * r->height, gx * gy);
        tile_counters = im.tile_cursor;
        tile_counter_words = gx * gy * GS4D_TILE_CURSOR_STRIDE;
    }
    rc = gs4d_activate_forward_clear(r->P, sc, rot, r->opacity_logit, a.scales_a, a.rot_n, a.opacity_a, tile_counters,
                                     tile_counter_words, stream);
    if (rc) return rc;
    FixedBlock geom{a.geom, a.geom_bytes}, image{a.image, a.image_bytes};
    BinningRelay bin{alloc, alloc_ctx, nullptr, 0};
    int R = 0, layout_R = 0;
    // binning and blend are issued before num_rendered has been read (capacity-checked, csrc/api.hip): the state carries
    // the count the binning chunk was laid out for
    rc = gs4d_rasterize_forward_spec(fixed_alloc, &geom, binning_alloc, &bin, fixed_alloc, &image, r->P, r->D, r->M,
                                     r->background, r->width, r->height, m3, shs, nullptr, a.opacity_a, a.scales_a,
                                     r->scale_modifier, a.rot_n, nullptr, r->viewmatrix, r->projmatrix, r->cam_pos, r->tan_fovx,
                                     r->tan_fovy, r->prefiltered, out_color, out_depth, radii, r->flags, stream, &R,
                                     1 | (tile_counters ? 2 : 0), &layout_R, r->visible);
    state->binning = bin.last;
    state->binning_bytes = bin.last_bytes;
    state->num_rendered = R;
    state->layout_rendered = layout_R;
    if (rc) return rc;
    if (r->target)
    {
        const int defer = (r->flags & GS4D_LOSS_AFTER_BACKWARD) && !(r->flags & GS4D_FORWARD_ONLY);
        rc = gs4d_l1_loss_forward_grad((size_t)3 * r->width * r->height, out_color, r->target, a.l1_scratch, loss,
                                       r->l1_grad_scale, a.l1_grad, stream, defer);
        if (defer) state->deferred_loss = loss;
    }
    return rc;
}

int gs4d_render_backward(const gs4d_render_desc* r, const gs4d_render_state* st, const float* out_color,
                         const int* radii, const float* dL_dcolor, const float* dL_ddepth, const float* g_loss,
                         gs4d_alloc_fn alloc, void* alloc_ctx, const gs4d_render_grads* g, gs4d_render_stage_fn stage_fn,
                         void* stage_ctx, void* stream)
{
    GS4D_TRACE_SCOPE("gs4d_render_backward");
    hipStream_t s = (hipStream_t)stream;
    int rc = check_desc(r, "gs4d_render_backward");
    if (rc) return rc;
    if (!st || !st->arena || !alloc || !g || !radii || !g->dL_dmeans2D || !g->d_xyz || !g->d_log_scales || !g->d_rotations ||
        !g->d_opacity_logit || !g->d_shs || (r->deform && !g->deform) || ((uintptr_t)g->d_rotations & 15) ||
        ((uintptr_t)g->d_shs & 15)) {
        gs4d_set_error("gs4d_render_backward: null / misaligned state, allocator or gradient destination");
        return GS4D_EINVAL;
    }
    if ((dL_dcolor != nullptr) == (g_loss != nullptr) || (g_loss && (!r->target || !out_color))) {
        gs4d_set_error("gs4d_render_backward: pass exactly one of dL_dcolor and g_loss (g_loss needs the forward's target and colour image)");
        return GS4D_EINVAL;
    }
    if ((r->flags & GS4D_FORWARD_ONLY) || (r->deform && r->deform->no_grad)) {
        gs4d_set_error("gs4d_render_backward: the forward ran with GS4D_FORWARD_ONLY / no_grad");
        return GS4D_EINVAL;
    }
    const FwdArena f = carve_forward(r, st->arena);
    if (f.bytes > st->arena_bytes) { gs4d_set_error("gs4d_render_backward: the state does not belong to this descriptor"); return GS4D_EINVAL; }
    const int R = st->num_rendered;
    const BwdArena lay = carve_backward(r, R, false, nullptr);
    gs4d_note_alloc();
    char* base = (char*)alloc(alloc_ctx, lay.bytes);
    if (!base) { gs4d_set_error("gs4d_render_backward: the allocator returned NULL"); return GS4D_EALLOC; }
    if ((uintptr_t)base & 255) { gs4d_set_error("gs4d_render_backward: the scratch must be 256-byte aligned"); return GS4D_EINVAL; }
    const BwdArena b = carve_backward(r, R, false, base);
    const size_t P = (size_t)r->P;

    // what the rasterizer backward accumulates into / leaves untouched for culled Gaussians starts at zero
    // (R > 0: the per-Gaussian backward writes every element, zeros for the Gaussians that touched no tile, and the blend
    // backward reads a null depth gradient as zeros: no memset node in front of the backward)
    if (R == 0) GS4D_HIP_CHECK(hipMemsetAsync(b.zero_begin, 0, (size_t)((char*)b.zero_end - (char*)b.zero_begin), s));
    const float* shs_in = r->deform ? f.shs : r->shs;
    // the per-Gaussian backward writes every element of d_shs itself when its staged form applies (csrc/geom_bwd.hip)
    if (R == 0 || !gs4d_geom_backward_fills_dsh(r->M, shs_in, g->d_shs))
        GS4D_HIP_CHECK(hipMemsetAsync(g->d_shs, 0, sizeof(float) * (size_t)r->M * 3 * P, s));
    float* dL_dmean3D = r->deform ? b.dL_dmean3D : g->d_xyz;
    if (!r->deform) GS4D_HIP_CHECK(hipMemsetAsync(g->d_xyz, 0, sizeof(float) * 3 * P, s));
    if (R == 0) GS4D_HIP_CHECK(hipMemsetAsync(g->dL_dmeans2D, 0, sizeof(float) * 3 * P, s));

    const float* g_img = dL_dcolor;
    if (g_loss && r->l1_grad_scale && g_loss != r->l1_grad_scale) {
        gs4d_set_error("gs4d_render_backward: the forward left the loss gradient for desc.l1_grad_scale; g_loss must be that pointer");
        return GS4D_EINVAL;
    }
    if (g_loss && r->l1_grad_scale && g_loss == r->l1_grad_scale) {
        g_img = f.l1_grad;                                   // the forward's loss launch left it there (same arithmetic)
    } else if (g_loss) {
        rc = gs4d_l1_loss_backward((size_t)3 * r->width * r->height, out_color, r->target, g_loss, b.g_img, stream);
        if (rc) return rc;
        g_img = b.g_img;
    }
    const float *m3 = r->deform ? f.m3 : r->xyz, *shs = r->deform ? f.shs : r->shs;
    // activations: gradients w.r.t. the (deformed) pre-activation scales / rotations and the opacity logits.  With R > 0 the
    // per-Gaussian backward applies them itself, Gaussian by Gaussian (and its second launch finishes a deferred loss):
    // no launch of their own
    const float* rot_raw = r->deform ? f.rot : r->rotations;
    Gs4dActBwd act = {f.scales_a, rot_raw, f.opacity_a, r->deform ? b.d_sc : g->d_log_scales, r->deform ? b.d_rot : g->d_rotations,
                      g->d_opacity_logit, f.l1_scratch, (size_t)3 * r->width * r->height, r->target ? st->deferred_loss : nullptr};
    const bool act_folded = R > 0;
    FixedBlock inst{b.inst, b.inst_bytes};
    rc = gs4d_rasterize_backward_layout(fixed_alloc, &inst, r->P, r->D, r->M, R, r->background, r->width, r->height, m3, shs,
                                        nullptr, f.scales_a, r->scale_modifier, f.rot_n, nullptr, r->viewmatrix, r->projmatrix,
                                        r->cam_pos, r->tan_fovx, r->tan_fovy, radii, f.geom, st->binning, f.image, g_img,
                                        dL_ddepth, g->dL_dmeans2D, b.dL_dconic, b.dL_dopacity,
                                        b.dL_dcolor, b.dL_ddepth, dL_dmean3D, b.dL_dcov3D, g->d_shs, b.dL_dscale, b.dL_drot,
                                        r->flags & 1, stream, st->layout_rendered > R ? st->layout_rendered : R,
                                        act_folded ? &act : nullptr);
    if (rc) return rc;
    if (stage_fn) stage_fn(stage_ctx, GS4D_RENDER_STAGE_RASTER_DONE);
    if (!act_folded)
    rc = gs4d_activate_backward_loss(r->P, f.scales_a, rot_raw, f.opacity_a, b.dL_dscale, b.dL_drot, b.dL_dopacity,
                                     r->deform ? b.d_sc : g->d_log_scales, r->deform ? b.d_rot : g->d_rotations,
                                     g->d_opacity_logit, f.l1_scratch, (size_t)3 * r->width * r->height,
                                     r->target ? st->deferred_loss : nullptr, stream);
    if (rc) return rc;
    if (!r->deform) {
        if (stage_fn) stage_fn(stage_ctx, GS4D_RENDER_STAGE_DATA_DONE);
        return GS4D_OK;
    }
    // the SH coefficients only receive a residual: d_shs is its own input gradient (gs4d_deform_backward's alias rule)
    rc = gs4d_deform_backward_phases(r->deform, g->deform, r->P, r->xyz, r->log_scales, r->rotations, r->time, r->force,
                                     b.dL_dmean3D, b.d_sc, b.d_rot, g->d_shs, g->d_xyz, g->d_log_scales, g->d_rotations,
                                     g->d_shs, f.ws, b.deform_scratch, stream, GS4D_DEFORM_BWD_DATA);
    if (rc) return rc;
    if (stage_fn) stage_fn(stage_ctx, GS4D_RENDER_STAGE_DATA_DONE);
    return gs4d_deform_backward_phases(r->deform, g->deform, r->P, r->xyz, r->log_scales, r->rotations, r->time, r->force,
                                       b.dL_dmean3D, b.d_sc, b.d_rot, g->d_shs, g->d_xyz, g->d_log_scales, g->d_rotations,
                                       g->d_shs, f.ws, b.deform_scratch, stream, GS4D_DEFORM_BWD_WEIGHTS);
}

}  // extern "C"
