// Forward 16x16-tile alpha blend for gfx950.
//
// Behaviour matched: RAST/cuda_rasterizer/forward.cu:261-379 of the reference
// (RAST = submodules/depth-diff-gaussian-rasterization): front-to-back blending of each tile's
// depth-sorted Gaussian list, alpha = min(0.99, o * exp(power)), skip power > 0 and
// alpha < 1/255, stop a pixel when T*(1-alpha) < 1e-4; outputs colour (+T*bg), depth (no bg
// term), final T and n_contrib = list position of the pixel's last contributor.
//
// gfx950 design:
//  * one workgroup = one tile = 2 wave64; a wave owns 8 rows x 16 columns and every lane TWO
//    horizontally adjacent pixels, so the per-pair arithmetic runs on the packed-fp32 instructions
//    (v_pk_fma/mul/add_f32: both pixels per issue) -- the kernel is bound by VALU issue, not memory;
//  * a pixel that does not blend a Gaussian runs the same arithmetic with alpha = 0 (exact identity:
//    w = 0, T * (1 - 0)), so the two pixels of a lane never diverge;
//  * each batch of 256 list entries is gathered ONCE per tile into LDS as three arrays
//    (xy+conic.xy | conic.z,opacity,depth,r | g,b) so the inner loop issues two broadcast
//    ds_read_b128 for the alpha test and touches the colour words only when some lane of the
//    wave actually blends the Gaussian (wave ballot);
//  * early termination is decided per wave with ballots, per tile with one LDS word per wave;
//  * exp() is evaluated as v_exp_f32(power * log2 e): |power| <= 5.55 for every alpha that
//    survives the 1/255 test, so the relative error stays < 1e-6 (tolerance: 1e-5 abs on colour).
#include <stdlib.h>

#include "gs4d_internal.h"
#include "gs4d_math.h"

GS4D_TIMELINE_DEFINE(blend_fwd)

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

template <bool EXACT>
__device__ __forceinline__ float fast_exp(float x)
{
    if (EXACT) return expf(x);
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

#define FWD_THREADS 128
#ifndef FWD_PAIR
#define FWD_PAIR 2          // list entries per trip of the blend loop (0: the one-entry loop, kept for A/B measurements)
#endif
static_assert(GS4D_SEG == GS4D_TILE_PIX || 2 * GS4D_SEG == GS4D_TILE_PIX, "checkpoints are taken in front of the batches and, with 128-entry segments, in their middle");

// CKPT: leave the segment checkpoints the backward walks from (BinningState::ckpt / seg_owner, gs4d_internal.h): the
// state of every pixel in front of list positions GS4D_SEG, 2 GS4D_SEG, ... of the tile (in front of every batch but the
// first and, with 128-entry segments, in the middle of every batch),
// the final colour / depth sums, and one owner record per segment >= 1 that some pixel reached.
template <bool EXACT, bool CKPT>
__global__ __launch_bounds__(FWD_THREADS) void blend_forward_kernel(
    int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float2* __restrict__ means2D, const float* __restrict__ features, const float* __restrict__ depths,
    const float4* __restrict__ conic_opacity, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_max_contrib, float* __restrict__ out_color,
    float* __restrict__ out_depth, float2* __restrict__ ckpt, uint2* __restrict__ seg_owner)
{
    __shared__ float4 s_a[GS4D_TILE_PIX];   // x, y, conic.x, conic.y
    __shared__ float4 s_b[GS4D_TILE_PIX];   // conic.z, opacity, depth, r
    __shared__ float2 s_c[GS4D_TILE_PIX];   // g, b
    __shared__ float4 s_box[GS4D_TILE_PIX]; // x - hx, x + hx, y - hy, y + hy: where alpha can reach 1/255
    __shared__ int s_wave_done[2];
    __shared__ uint32_t s_wave_max[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile_x = blockIdx.x, tile_y = blockIdx.y;
    // wave -> 8 rows x 16 columns, lane -> two adjacent pixels of a row
    const int px0 = tile_x * GS4D_TILE_X + 2 * (lane & 7);
    const int py = tile_y * GS4D_TILE_Y + 8 * wave + (lane >> 3);
    const bool in0 = px0 < W && py < H, in1 = px0 + 1 < W && py < H;
    const f2 pxf = {(float)px0, (float)(px0 + 1)};
    const float pyf = (float)py;
    // the wave's pixel rectangle (pixel coordinates are integers)
    const float rx0 = (float)(tile_x * GS4D_TILE_X), rx1 = rx0 + (float)(GS4D_TILE_X - 1);
    const float ry0 = (float)(tile_y * GS4D_TILE_Y + 8 * wave), ry1 = ry0 + 7.0f;

    const uint2 range = ranges[tile_y * gx + tile_x];
    const int n_total = (int)(range.y - range.x);
#ifdef GS4D_TIMELINE
    gs4d::TimelineScope tl(g_timeline, tile_y * gx + tile_x);
    tl.n = (unsigned)max(n_total, 0);
#endif

    bool done0 = !in0, done1 = !in1;
    f2 T = {1.0f, 1.0f};
    f2 C0 = {0.f, 0.f}, C1 = {0.f, 0.f}, C2 = {0.f, 0.f}, Dp = {0.f, 0.f};
    uint32_t last0 = 0, last1 = 0;

    for (int base = 0; base < n_total; base += GS4D_TILE_PIX) {
        // tile-wide early exit: every wave publishes whether all of its pixels are finished
        const bool wave_done = __ballot(!done0 || !done1) == 0ull;
        if (lane == 0) s_wave_done[wave] = wave_done;
        __syncthreads();
        if (s_wave_done[0] & s_wave_done[1]) break;
        // checkpoint j = the state in front of list position j * GS4D_SEG
        auto checkpoint = [&](int j) {
            float2* c = ckpt + ((size_t)(range.x / (uint32_t)GS4D_SEG_SLOT_DIV) + (size_t)j - 1u) * (GS4D_CKPT_FIELDS * FWD_THREADS) + tid;
            c[0 * FWD_THREADS] = make_float2(T.x, T.y);
            c[1 * FWD_THREADS] = make_float2(C0.x, C0.y);
            c[2 * FWD_THREADS] = make_float2(C1.x, C1.y);
            c[3 * FWD_THREADS] = make_float2(C2.x, C2.y);
            c[4 * FWD_THREADS] = make_float2(Dp.x, Dp.y);
        };
        if (CKPT && base != 0) checkpoint(base / GS4D_SEG);
        // with 128-entry segments there is one more in the middle of the batch: a wave that walks the batch takes it
        // between the second and the third 64-entry group, a wave whose pixels are all finished right away (same state)
        const bool mid_ckpt = CKPT && GS4D_SEG < GS4D_TILE_PIX && n_total - base > GS4D_SEG;
        if (mid_ckpt && wave_done) checkpoint(base / GS4D_SEG + 1);

        const int n = min(GS4D_TILE_PIX, n_total - base);
        for (int e = tid; e < n; e += FWD_THREADS) {
            const uint32_t id = point_list[range.x + base + e];
            const float2 xy = means2D[id];
            const float4 co = conic_opacity[id];
            const float d = depths[id];
            const float r = features[3 * (size_t)id + 0], g = features[3 * (size_t)id + 1],
                        b = features[3 * (size_t)id + 2];
            // A pixel that skips an entry blends it with weight 0 (straight-line loop body, below): 0 * Inf or 0 * NaN
            // would poison it, where the reference's `continue` leaves it untouched.  Colours and depth are therefore
            // staged as finite values (fmin / fmax drop a NaN; identity for every finite input) -- once per staged
            // entry, not per visit; a pixel that does blend a diverged Gaussian still ends up non-finite or huge.
            auto finite = [](float v) { return fminf(fmaxf(v, -3.4028235e38f), 3.4028235e38f); };
            s_a[e] = make_float4(xy.x, xy.y, co.x, co.y);
            s_b[e] = make_float4(co.z, co.w, finite(d), finite(r));
            s_c[e] = make_float2(finite(g), finite(b));
            s_box[e] = gs4d::alpha_support_box(xy, co);
        }
        __syncthreads();

        if (!wave_done) {
            // Which entries can reach alpha >= 1/255 somewhere in this wave's 16 x 8 pixels?  One entry per lane
            // is tested against the wave's rectangle, the ballots are wave-uniform bit masks, and the blend loop
            // walks their set bits: an entry whose support misses the rectangle costs nothing at all.
            unsigned long long hits[GS4D_TILE_PIX / 64];
#pragma unroll
            for (int k = 0; k < GS4D_TILE_PIX / 64; k++) {
                const int e = 64 * k + lane;
                bool hit = false;
                if (e < n) {
                    const float4 bx = s_box[e];
                    hit = bx.x <= rx1 && bx.y >= rx0 && bx.z <= ry1 && bx.w >= ry0;
                    if (hit) {
                        const float4 ea = s_a[e], eb = s_b[e];
                        hit = !gs4d::alpha_support_misses_rect(ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, rx0, rx1, ry0, ry1);
                    }
                }
                hits[k] = __ballot(hit);
            }
            bool all_done = false;
#pragma unroll
            for (int k = 0; k < GS4D_TILE_PIX / 64; k++) {
                if (mid_ckpt && 64 * k == GS4D_SEG) checkpoint(base / GS4D_SEG + 1);
                unsigned long long m = hits[k];
#if FWD_PAIR
                // FWD_PAIR entries per trip.  A visit is ONE wave's dependent chain (LDS read -> quadratic form -> exp -> alpha ->
                // four compare / mask hops between the vector and the scalar unit -> the exit ballot: ~90 instructions that
                // took ~950 cycles, whatever else ran on the SIMD), but only four instructions of it -- through T -- depend
                // on the previous entry.  With the next set bits of the mask in one straight-line body the entries'
                // independent parts overlap; the T-dependent parts run in list order, so every pixel sees exactly the
                // arithmetic of the one-entry loop (bit-identical images, n_contrib, checkpoints).  A trip that runs out of
                // set bits repeats its last entry with nobody using it (alpha 0: the identity, as for a skipped entry).
                while (m != 0ull && !all_done) {
                    int j[FWD_PAIR];
                    bool live[FWD_PAIR];
#pragma unroll
                    for (int q = 0; q < FWD_PAIR; q++) {
                        live[q] = m != 0ull;
                        j[q] = live[q] ? 64 * k + __builtin_ctzll(m) : j[q ? q - 1 : 0];
                        m &= m - 1;                               // (0 & -1 stays 0)
                    }
                    float4 ea[FWD_PAIR], eb[FWD_PAIR];
                    float2 ec[FWD_PAIR];
#pragma unroll
                    for (int q = 0; q < FWD_PAIR; q++) { ea[q] = s_a[j[q]]; eb[q] = s_b[j[q]]; ec[q] = s_c[j[q]]; }
                    f2 alpha[FWD_PAIR], om[FWD_PAIR];
                    bool ok0[FWD_PAIR], ok1[FWD_PAIR];
#pragma unroll
                    for (int q = 0; q < FWD_PAIR; q++) {          // independent of the other entries of the trip
                        const f2 dx = ea[q].x - pxf;
                        const float dy = ea[q].y - pyf;
                        const f2 power = -0.5f * (ea[q].z * dx * dx + eb[q].x * dy * dy) - ea[q].w * dx * dy;
                        alpha[q] = f2{fminf(0.99f, eb[q].y * fast_exp<EXACT>(power.x)), fminf(0.99f, eb[q].y * fast_exp<EXACT>(power.y))};
                        ok0[q] = live[q] && !(power.x > 0.0f) && !(alpha[q].x < (1.0f / 255.0f));
                        ok1[q] = live[q] && !(power.y > 0.0f) && !(alpha[q].y < (1.0f / 255.0f));
                        om[q] = 1.0f - alpha[q];
                    }
#pragma unroll
                    for (int q = 0; q < FWD_PAIR; q++) {          // in list order: the only part that depends on T
                        const f2 test_T = T * om[q];
                        bool use0 = !done0 && ok0[q], use1 = !done1 && ok1[q];
                        if (use0 && test_T.x < 0.0001f) { done0 = true; use0 = false; }
                        if (use1 && test_T.y < 0.0001f) { done1 = true; use1 = false; }
                        const f2 al = {use0 ? alpha[q].x : 0.f, use1 ? alpha[q].y : 0.f};
                        const f2 w = al * T;
                        C0 += eb[q].w * w;
                        C1 += ec[q].x * w;
                        C2 += ec[q].y * w;
                        Dp += eb[q].z * w;
                        T.x = use0 ? test_T.x : T.x;
                        T.y = use1 ? test_T.y : T.y;
                        last0 = use0 ? (uint32_t)(base + j[q] + 1) : last0;
                        last1 = use1 ? (uint32_t)(base + j[q] + 1) : last1;
#ifdef GS4D_TIMELINE
                        tl.visits += live[q] ? 1u : 0u;
#endif
                    }
                    if (__ballot(!done0 || !done1) == 0ull) all_done = true;  // every pixel of this wave is finished
                }
#else
                while (m != 0ull && !all_done) {
                    const int j = 64 * k + __builtin_ctzll(m);
                    m &= m - 1;
                    const float4 a = s_a[j];
                    const float4 b = s_b[j];
                    const f2 dx = a.x - pxf;
                    const float dy = a.y - pyf;
                    const f2 power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                    const f2 alpha = {fminf(0.99f, b.y * fast_exp<EXACT>(power.x)), fminf(0.99f, b.y * fast_exp<EXACT>(power.y))};
                    const f2 test_T = T * (1.0f - alpha);
                    bool use0 = !done0 && !(power.x > 0.0f) && !(alpha.x < (1.0f / 255.0f));
                    bool use1 = !done1 && !(power.y > 0.0f) && !(alpha.y < (1.0f / 255.0f));
                    if (use0 && test_T.x < 0.0001f) { done0 = true; use0 = false; }
                    if (use1 && test_T.y < 0.0001f) { done1 = true; use1 = false; }
                    // No branch around an entry that no pixel of the wave blends (6 % of the visits once the exact
                    // support test has run): a pixel that skips blends with alpha 0, the exact identity on its state for
                    // finite colours / depths, and the straight-line body keeps the seven loop-carried registers in
                    // place -- with a branch the compiler copied all of them on every visit (118 -> 110 us at C2).
                    {
                        const float2 c = s_c[j];
                        const f2 al = {use0 ? alpha.x : 0.f, use1 ? alpha.y : 0.f};
                        const f2 w = al * T;
                        C0 += b.w * w;
                        C1 += c.x * w;
                        C2 += c.y * w;
                        Dp += b.z * w;
                        T.x = use0 ? test_T.x : T.x;
                        T.y = use1 ? test_T.y : T.y;
                        last0 = use0 ? (uint32_t)(base + j + 1) : last0;
                        last1 = use1 ? (uint32_t)(base + j + 1) : last1;
                    }
                    if (__ballot(!done0 || !done1) == 0ull) all_done = true;  // every pixel of this wave is finished
#ifdef GS4D_TIMELINE
                    tl.visits++;
#endif
                }
#endif
            }
        }
        __syncthreads();  // the staging arrays are rewritten by the next batch
    }

    const size_t pix = (size_t)py * W + px0;
    const size_t plane = (size_t)H * W;
    if (in0) {
        final_T[pix] = T.x;
        n_contrib[pix] = last0;
        out_color[pix] = C0.x + T.x * bg[0];
        out_color[plane + pix] = C1.x + T.x * bg[1];
        out_color[2 * plane + pix] = C2.x + T.x * bg[2];
        out_depth[pix] = Dp.x;
    }
    if (in1) {
        final_T[pix + 1] = T.y;
        n_contrib[pix + 1] = last1;
        out_color[pix + 1] = C0.y + T.y * bg[0];
        out_color[plane + pix + 1] = C1.y + T.y * bg[1];
        out_color[2 * plane + pix + 1] = C2.y + T.y * bg[2];
        out_depth[pix + 1] = Dp.y;
    }
    // per-tile maximum of n_contrib: lets the backward skip list entries no pixel reached
    uint32_t m = max(in0 ? last0 : 0u, in1 ? last1 : 0u);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_down(m, off, 64));
    if (lane == 0) s_wave_max[wave] = m;
    __syncthreads();
    const uint32_t n_live = max(s_wave_max[0], s_wave_max[1]);
    if (tid == 0) tile_max_contrib[tile_y * gx + tile_x] = n_live;
    if (CKPT && n_total > GS4D_SEG) {
        const uint32_t first = range.x / (uint32_t)GS4D_SEG_SLOT_DIV, c_t = (uint32_t)(n_total - 1) / GS4D_SEG;
        float2* c = ckpt + ((size_t)first + c_t) * (GS4D_CKPT_FIELDS * FWD_THREADS) + tid;      // the final record
        c[1 * FWD_THREADS] = make_float2(C0.x, C0.y);
        c[2 * FWD_THREADS] = make_float2(C1.x, C1.y);
        c[3 * FWD_THREADS] = make_float2(C2.x, C2.y);
        c[4 * FWD_THREADS] = make_float2(Dp.x, Dp.y);
        for (uint32_t j = 1 + tid; j <= c_t; j += FWD_THREADS)
            if (j * GS4D_SEG < n_live) seg_owner[first + j - 1] = make_uint2((uint32_t)(tile_y * gx + tile_x), j);
    }
}

}  // namespace

int gs4d_launch_blend_forward(int W, int H, const GeomState& g, const BinningState& b, ImageState& im,
                              const float* features, const float* bg, float* out_color, float* out_depth, int checkpoints,
                              int debug, hipStream_t s)
{
    GS4D_PROF("blend_forward", s);
    const int gx = (W + GS4D_TILE_X - 1) / GS4D_TILE_X, gy = (H + GS4D_TILE_Y - 1) / GS4D_TILE_Y;
    static const bool exact = getenv("GS4D_EXACT_EXP") != nullptr;  // diagnostic switch: libm-accurate exp
    auto kernel = exact ? (checkpoints ? blend_forward_kernel<true, true> : blend_forward_kernel<true, false>)
                        : (checkpoints ? blend_forward_kernel<false, true> : blend_forward_kernel<false, false>);
    hipLaunchKernelGGL(kernel, dim3(gx, gy), dim3(FWD_THREADS), 0, s, W, H, gx, im.ranges,
                       b.point_list(), g.means2D, features, g.depths, g.conic_opacity, bg, im.final_T, im.n_contrib,
                       im.tile_max_contrib, out_color, out_depth, b.ckpt, b.seg_owner);
    GS4D_LAUNCH_CHECK("blend_forward", debug, s);
    return GS4D_OK;
}
