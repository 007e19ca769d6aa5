// Backward 16x16-tile blend for gfx950.
//
// Behaviour matched: RAST/cuda_rasterizer/backward.cu:415-590 of the reference
// (RAST = submodules/depth-diff-gaussian-rasterization): back-to-front replay of every pixel's
// blend using final_T and n_contrib, producing for every Gaussian the sums over pixels of the
// gradients w.r.t. colour (3), depth (1), 2D mean (2, scaled by 0.5 W / 0.5 H), conic (3) and
// opacity (1).
//
// The reference issues 10 global float atomics per contributing (pixel, Gaussian) pair.  Here:
//  * a workgroup is two wave64; every lane owns TWO horizontally adjacent pixels, so the per-pair
//    arithmetic runs on the packed-fp32 instructions (v_pk_fma/mul/add_f32: both pixels per issue) --
//    this kernel is bound by VALU issue, not by memory;
//  * a pixel that does not blend a Gaussian is not branched around: its alpha and G are replaced by 0,
//    which turns every state update into the exact identity (T/(1-0), 0*c + 1*acc) and every gradient
//    term into 0, so the two pixels of a lane never diverge;
//  * the 10 per-Gaussian sums over the wave's 128 pixels use a transposing reduction:
//    v_permlane32_swap / v_permlane16_swap halve the lanes AND the number of live values per step
//    (28 VALU ops per Gaussian instead of 60 DPP adds), ending in three 16-lane DPP reductions;
//  * lanes 15/31/47/63 store the wave partial into LDS (one private row per wave: no LDS atomics);
//    a per-wave bit mask records which list entries were written, so nothing is zero-filled;
//  * after a batch of <= 256 list entries the two wave partials are added in fixed order and
//    the tile's partial row (12 floats, 10 used) is written to inst_grad[emission slot of the (Gaussian, tile)
//    pair], i.e. one 48-byte row per instance and NO global atomics;
//  * the per-Gaussian kernel (geom_bwd.hip) sums a Gaussian's rows -- contiguous, in emission order -- so
//    gradients are bitwise reproducible run to run.
// Rows of list entries that no pixel of the tile reached (position >= tile_max_contrib) are
// zero-filled without being evaluated.  1/(1-alpha) uses v_rcp_f32 (1 ulp) where the reference
// divides; gradient tolerance is 1e-4 relative.
#include <stdlib.h>

#include "gs4d_internal.h"
#include "gs4d_math.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

template <bool EXACT>
__device__ __forceinline__ float fast_exp(float x)
{
    if (EXACT) return expf(x);
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(moved);
}

// sum over each row of 16 lanes; valid in lanes 12..15 of the row
__device__ __forceinline__ float row16_sum(float v)
{
    v = dpp_add<0xB1>(v);             // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);             // quad_perm [2,3,0,1]
    v = dpp_add<0x114>(v);            // row_shr:4
    v = dpp_add<0x118>(v);            // row_shr:8
    return v;
}

// the same for three values at once, as fused DPP adds (left to itself the compiler pairs two of the three into
// v_pk_add_f32, which cannot take a DPP operand, and spends 5 instructions per step on moves).  A DPP read of a
// register needs two wait states after the VALU write of it: s_nop 1 in front, and within the block every value is
// touched again only after the other two.
__device__ __forceinline__ void row16_sum3(float& a, float& b, float& c)
{
#define GS4D_DPP3(ctrl)                                                                   \
    "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"          \
    "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"          \
    "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
    asm volatile("s_nop 1\n" GS4D_DPP3("quad_perm:[1,0,3,2]") GS4D_DPP3("quad_perm:[2,3,0,1]") GS4D_DPP3("row_shr:4")
                 GS4D_DPP3("row_shr:8")
                 : "+v"(a), "+v"(b), "+v"(c));
#undef GS4D_DPP3
}

// lanes 0..31 receive a[l] + a[l+32], lanes 32..63 receive b[l-32] + b[l]
__device__ __forceinline__ float fold32(float a, float b)
{
    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// per 32-lane half: its row 0 receives a[row0] + a[row1], its row 1 receives b[row0] + b[row1]
__device__ __forceinline__ float fold16(float a, float b)
{
    const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

#define NQ 10     // gradient values per (tile, Gaussian) instance
#define NSLOT 12  // LDS slots per instance: slot = 3 * (lane >> 4) + t, see the store below
#define BWD_THREADS 128
#define BWD_BATCH 128   // list entries staged per pass (LDS: 5 KB staging + 12 KB partials -> 9 workgroups per CU)
static_assert(GS4D_SEG % BWD_BATCH == 0, "a segment is a whole number of batches");

GS4D_TIMELINE_DEFINE(blend_bwd)

template <bool EXACT>
__global__ __launch_bounds__(BWD_THREADS) void blend_backward_kernel(
    int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const uint32_t* __restrict__ tile_max_contrib, const float* __restrict__ bg, const float2* __restrict__ means2D,
    const float4* __restrict__ conic_opacity, const float* __restrict__ colors, const float* __restrict__ depths,
    const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dpixel_depths, const int* __restrict__ radii, const uint32_t* __restrict__ point_offsets,
    int gy, float* __restrict__ inst_grad, uint32_t* __restrict__ clear_word, const float2* __restrict__ ckpt,
    const uint2* __restrict__ seg_owner, int seg_slots)
{
    __shared__ float4 s_a[BWD_BATCH];   // x, y, conic.x, conic.y
    __shared__ float4 s_b[BWD_BATCH];   // conic.z, opacity, depth, r
    __shared__ float2 s_c[BWD_BATCH];   // g, b
    __shared__ float4 s_box[BWD_BATCH]; // x - hx, x + hx, y - hy, y + hy: where alpha can reach 1/255
    __shared__ float s_part[2][BWD_BATCH][NSLOT];  // per-wave partial sums of the current batch (24 KB)
    __shared__ unsigned long long s_mask[2][BWD_BATCH / 64];        // which entries of the batch a wave has written
    __shared__ uint32_t s_slot[BWD_BATCH];                          // emission slot of the entry: where its row goes

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroup q < seg_slots walks the segment >= 1 that owns checkpoint slot q, if any (seg_owner, written by the forward
    // blend); workgroups seg_slots + t walk segment 0 of tile t (list positions [0, GS4D_SEG)): the lists are walked back
    // to front in GS4D_SEG-entry segments, all at once, each from the pixel state the forward left in front of the NEXT
    // segment.  With one workgroup per tile the launch lasted as long as the longest list's serial walk (every workgroup
    // is resident within microseconds): 204 us at C2 where the arithmetic of all tiles together needs half of that;
    // 256-entry segments 162 us, 128-entry segments 151 us (what is left of the tail is one segment's walk).
    // The segment workgroups come first in the grid: they belong to the long lists.
#ifdef GS4D_TIMELINE
    gs4d::TimelineScope tl(g_timeline, (int)blockIdx.x);
#endif
    const int blk = (int)blockIdx.x;
    int tile = blk - seg_slots, seg = 0;
    if (tile < 0) {
        const uint2 o = seg_owner[blk];
        if (o.x == ~0u) return;
        tile = (int)o.x;
        seg = (int)o.y;
    }
    const int tile_x = tile % gx, tile_y = tile / gx;
    if (clear_word && blockIdx.x == gridDim.x - 1 && tid == 0) *clear_word = 0u;     // the per-Gaussian kernel's deferred-Gaussian counter
    // wave -> 8 rows x 16 columns, lane -> two adjacent pixels of a row
    const int px0 = tile_x * GS4D_TILE_X + 2 * (lane & 7);
    const int py = tile_y * GS4D_TILE_Y + 8 * wave + (lane >> 3);
    const bool in0 = px0 < W && py < H, in1 = px0 + 1 < W && py < H;
    const f2 pxf = {(float)px0, (float)(px0 + 1)};
    const float pyf = (float)py;
    // the wave's pixel rectangle (pixel coordinates are integers)
    const float rx0 = (float)(tile_x * GS4D_TILE_X), rx1 = rx0 + (float)(GS4D_TILE_X - 1);
    const float ry0 = (float)(tile_y * GS4D_TILE_Y + 8 * wave), ry1 = ry0 + 7.0f;
    const uint2 range = ranges[tile];
    const int n_total = (int)(range.y - range.x);
#ifdef GS4D_TIMELINE
    tl.n = (unsigned)max(n_total, 0);
#ifdef GS4D_TL_STRIDE
    if (tile % GS4D_TL_STRIDE != 0) return;        // experiment: uncontended per-visit latency
#endif
#endif
    if (n_total <= 0) return;
    const int n_live = (int)min(tile_max_contrib[tile], (uint32_t)n_total);  // entries some pixel blended

    // A row goes to the entry's EMISSION slot (the position duplicateWithKeys gave the (Gaussian, tile) pair:
    // offsets[id - 1] + row-major index of this tile inside the Gaussian's tile rectangle), not to its sorted
    // position: all rows of one Gaussian are then contiguous, the per-Gaussian kernel streams them, and the inverse
    // permutation (a kernel over all R entries plus a dependent gather per row) is not needed at all.
    float4* __restrict__ rows = reinterpret_cast<float4*>(inst_grad);
    auto slot_of = [&](uint32_t id, float2 xy) -> uint32_t {
        int x0, y0, x1, y1;
        gs4d::tile_rect(xy.x, xy.y, radii[id], gx, gy, x0, y0, x1, y1);
        const uint32_t first = id ? point_offsets[id - 1] : 0u;
        return first + (uint32_t)((tile_y - y0) * (x1 - x0) + (tile_x - x0));
    };
    // zero-fill rows of entries nobody reached (the tile's first workgroup)
    if (seg == 0)
        for (int i = n_live + tid; i < n_total; i += BWD_THREADS) {
            const uint32_t id = point_list[range.x + i];
            float4* dst = rows + (size_t)slot_of(id, means2D[id]) * (GS4D_GRAD_ROW / 4);
            dst[0] = dst[1] = dst[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    const int seg_lo = seg * GS4D_SEG, seg_hi = min(n_live, seg_lo + GS4D_SEG);
    if (seg_hi <= seg_lo) return;

    const size_t pix = (size_t)py * W + px0;
    const size_t plane = (size_t)H * W;
    f2 T_final = {0.f, 0.f}, dLp0 = {0.f, 0.f}, dLp1 = {0.f, 0.f}, dLp2 = {0.f, 0.f}, dLpd = {0.f, 0.f};
    int lastc0 = 0, lastc1 = 0;
    if (in0) {
        T_final.x = final_Ts[pix]; lastc0 = (int)n_contrib[pix];
        dLp0.x = dL_dpixels[pix]; dLp1.x = dL_dpixels[plane + pix]; dLp2.x = dL_dpixels[2 * plane + pix];
        dLpd.x = dL_dpixel_depths ? dL_dpixel_depths[pix] : 0.f;      // null (composite call without a depth loss): zeros
    }
    if (in1) {
        T_final.y = final_Ts[pix + 1]; lastc1 = (int)n_contrib[pix + 1];
        dLp0.y = dL_dpixels[pix + 1]; dLp1.y = dL_dpixels[plane + pix + 1]; dLp2.y = dL_dpixels[2 * plane + pix + 1];
        dLpd.y = dL_dpixel_depths ? dL_dpixel_depths[pix + 1] : 0.f;
    }
    f2 T = T_final;
    f2 bg_dot_dpixel = {0.f, 0.f};
    bg_dot_dpixel += bg[0] * dLp0;
    bg_dot_dpixel += bg[1] * dLp1;
    bg_dot_dpixel += bg[2] * dLp2;
    f2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f}, accd = {0.f, 0.f}, last_alpha = {0.f, 0.f};
    if (seg_hi < n_live) {
        // Not the last segment: T in front of position seg_hi is the forward's own value, and the colour / depth blended
        // BEHIND that position (what the walk from the back would have accumulated) is what the forward still added
        // after it, seen through that T: (final sum - sum so far) / T.  Its rounding error, ~eps |final sum| / T, enters
        // the gradients multiplied by T again.  A pixel that had stopped before seg_hi has equal sums: exactly 0.
        const uint32_t first = range.x / (uint32_t)GS4D_SEG_SLOT_DIV, c_t = (uint32_t)(n_total - 1) / GS4D_SEG;
        const float2* ck = ckpt + ((size_t)first + (size_t)seg) * (GS4D_CKPT_FIELDS * BWD_THREADS) + tid;
        const float2* fin = ckpt + ((size_t)first + c_t) * (GS4D_CKPT_FIELDS * BWD_THREADS) + tid;
        const float2 cT = ck[0], c0 = ck[1 * BWD_THREADS], c1 = ck[2 * BWD_THREADS], c2 = ck[3 * BWD_THREADS], cd = ck[4 * BWD_THREADS];
        const float2 f0 = fin[1 * BWD_THREADS], f1 = fin[2 * BWD_THREADS], f2_ = fin[3 * BWD_THREADS], fd = fin[4 * BWD_THREADS];
        auto behind = [](float fin_v, float ck_v, float t) { const float d = fin_v - ck_v; return d != 0.f ? d / t : 0.f; };
        T = f2{cT.x, cT.y};
        acc0 = f2{behind(f0.x, c0.x, cT.x), behind(f0.y, c0.y, cT.y)};
        acc1 = f2{behind(f1.x, c1.x, cT.x), behind(f1.y, c1.y, cT.y)};
        acc2 = f2{behind(f2_.x, c2.x, cT.x), behind(f2_.y, c2.y, cT.y)};
        accd = f2{behind(fd.x, cd.x, cT.x), behind(fd.y, cd.y, cT.y)};
    }
    float lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_depth = 0.f;   // colour / depth of the previous evaluated entry (wave-uniform)
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);

    // batches walk the segment from the back; each covers positions [lo, lo + n)
    for (int hi = seg_hi; hi > seg_lo; hi -= BWD_BATCH) {
        const int lo = max(seg_lo, hi - BWD_BATCH);
        const int n = hi - lo;
        __syncthreads();  // previous batch's rows have been flushed
        for (int e = tid; e < n; e += BWD_THREADS) {
            const uint32_t id = point_list[range.x + lo + e];
            const float2 xy = means2D[id];
            const float4 co = conic_opacity[id];
            const float d = depths[id];
            const float r = colors[3 * (size_t)id + 0], g = colors[3 * (size_t)id + 1], b = colors[3 * (size_t)id + 2];
            s_a[e] = make_float4(xy.x, xy.y, co.x, co.y);
            s_b[e] = make_float4(co.z, co.w, d, r);
            s_c[e] = make_float2(g, b);
            s_box[e] = gs4d::alpha_support_box(xy, co);
            s_slot[e] = slot_of(id, xy);
        }
        __syncthreads();

        for (int jb = (n - 1) >> 6; jb >= 0; jb--) {
            // entries whose alpha support misses this wave's 16 x 8 pixels are never visited (see blend_fwd.hip)
            unsigned long long todo;
            {
                const int e = 64 * jb + lane;
                bool hit = false;
                if (e < n) {
                    const float4 bx = s_box[e];
                    hit = bx.x <= rx1 && bx.y >= rx0 && bx.z <= ry1 && bx.w >= ry0;
                    if (hit) {
                        const float4 ea = s_a[e], eb = s_b[e];
                        hit = !gs4d::alpha_support_misses_rect(ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, rx0, rx1, ry0, ry1);
                    }
                }
                todo = __ballot(hit);
            }
            unsigned long long written = 0ull;     // wave-uniform
            while (todo != 0ull) {
                const int jl = 63 - __builtin_clzll(todo);
                todo &= ~(1ull << jl);
                const int j = 64 * jb + jl;
                const int pos = lo + j;  // 0-based list position == the reference's `contributor` after decrement
                const float4 a = s_a[j];
                const float4 b = s_b[j];
                const f2 dx = a.x - pxf;
                const float dy = a.y - pyf;
                const f2 power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                const f2 G = {fast_exp<EXACT>(power.x), fast_exp<EXACT>(power.y)};
                const f2 alpha = {fminf(0.99f, b.y * G.x), fminf(0.99f, b.y * G.y)};
                const bool use0 = pos < lastc0 && !(power.x > 0.0f) && !(alpha.x < (1.0f / 255.0f));
                const bool use1 = pos < lastc1 && !(power.y > 0.0f) && !(alpha.y < (1.0f / 255.0f));
                if (__ballot(use0 || use1) == 0ull) continue;

                const float2 c = s_c[j];
                // a pixel that skips this entry runs the same arithmetic with alpha = G = 0 (exact identity)
                const f2 al = {use0 ? alpha.x : 0.f, use1 ? alpha.y : 0.f};
                const f2 Gm = {use0 ? G.x : 0.f, use1 ? G.y : 0.f};
                const f2 one_minus_alpha = 1.f - al;
                const f2 rcp = {__builtin_amdgcn_rcpf(one_minus_alpha.x), __builtin_amdgcn_rcpf(one_minus_alpha.y)};
                T = T * rcp;
                const f2 w = al * T;
                const f2 oml = 1.f - last_alpha;
                f2 dL_dalpha = {0.f, 0.f};
                acc0 = last_alpha * lc0 + oml * acc0;
                dL_dalpha += (b.w - acc0) * dLp0;
                acc1 = last_alpha * lc1 + oml * acc1;
                dL_dalpha += (c.x - acc1) * dLp1;
                acc2 = last_alpha * lc2 + oml * acc2;
                dL_dalpha += (c.y - acc2) * dLp2;
                accd = last_alpha * last_depth + oml * accd;
                dL_dalpha += (b.z - accd) * dLpd;
                lc0 = b.w; lc1 = c.x; lc2 = c.y; last_depth = b.z;
                dL_dalpha *= T;
                last_alpha = al;
                dL_dalpha += (-T_final * rcp) * bg_dot_dpixel;
                // Every remaining sum is a moment of q = G * dL_dalpha over the wave's pixels: with d = mean - pixel,
                //   dL_dG * dG_ddelx = -o q (cx dx + cy dy),  dL_dG * dG_ddely = -o q (cz dy + cy dx),
                //   gdx dx dL_dG = o q dx^2,  gdx dy dL_dG = o q dx dy,  gdy dy dL_dG = o q dy^2
                // (o = opacity, (cx, cy, cz) = conic: constants of the entry), so the pixels only form q, q d and q d d^T
                // and the per-entry factors are applied ONCE, at the flush, to the sums.
                const f2 q = Gm * dL_dalpha;
                const f2 qx = q * dx, qy = q * dy;
                const f2 p0 = w * dLp0, p1 = w * dLp1, p2 = w * dLp2, p3 = w * dLpd;
                const f2 p4 = qx, p5 = qy;
                const f2 p6 = qx * dx, p7 = qx * dy, p8 = qy * dy;
                const f2 p9 = q;
                // the lane's two pixels, then the transposing reduction over the wave
                const float u0 = fold32(p0.x + p0.y, p5.x + p5.y);   // lanes <32: v0, >=32: v5
                const float u1 = fold32(p1.x + p1.y, p6.x + p6.y);   //            v1        v6
                const float u2 = fold32(p2.x + p2.y, p7.x + p7.y);   //            v2        v7
                const float u3 = fold32(p3.x + p3.y, p8.x + p8.y);   //            v3        v8
                const float u4 = fold32(p4.x + p4.y, p9.x + p9.y);   //            v4        v9
                float t0 = fold16(u0, u1);                           // rows: v0 v1 v5 v6
                float t1 = fold16(u2, u3);                           // rows: v2 v3 v7 v8
                float t2 = fold16(u4, u4);                           // rows: v4 v4 v9 v9
                row16_sum3(t0, t1, t2);
                if ((lane & 15) == 15) {
                    float* p = &s_part[wave][j][3 * (lane >> 4)];
                    p[0] = t0; p[1] = t1; p[2] = t2;
                }
                written |= 1ull << (j & 63);
#ifdef GS4D_TIMELINE
                tl.visits++;
#endif
            }
            if (lane == 0) s_mask[wave][jb] = written;
        }
        __syncthreads();
        // flush: fixed-order sum of the two wave partials, one 48-byte row per list entry
        for (int e = tid; e < n; e += BWD_THREADS) {
            const bool w0 = (s_mask[0][e >> 6] >> (e & 63)) & 1ull, w1 = (s_mask[1][e >> 6] >> (e & 63)) & 1ull;
            float sl[NSLOT];
#pragma unroll
            for (int q = 0; q < NSLOT; q++) sl[q] = (w0 ? s_part[0][e][q] : 0.f) + (w1 ? s_part[1][e][q] : 0.f);
            // slots: [v0 v2 v4 | v1 v3 v4 | v5 v7 v9 | v6 v8 v9]; v4.. are the moments of q (see above)
            const float4 ea = s_a[e], eb = s_b[e];
            const float o = eb.y, cx = ea.z, cy = ea.w, cz = eb.x;
            const float Sx = sl[2], Sy = sl[6], Sxx = sl[9], Sxy = sl[7], Syy = sl[10], Sq = sl[8];
            const float mx = -o * (cx * Sx + cy * Sy), my = -o * (cz * Sy + cy * Sx);
            const float oh = -0.5f * o;
            float4* dst = rows + (size_t)s_slot[e] * (GS4D_GRAD_ROW / 4);
            dst[0] = make_float4(sl[0], sl[3], sl[1], sl[4]);
            dst[1] = make_float4(mx * ddelx_dx, my * ddely_dy, oh * Sxx, oh * Sxy);
            dst[2] = make_float4(oh * Syy, Sq, 0.f, 0.f);
        }
    }
}

}  // namespace


int gs4d_launch_blend_backward(int W, int H, int R, const int* radii, const GeomState& g, const BinningState& b,
                               const ImageState& im, const float* features, const float* bg, const float* dL_dpix,
                               const float* dL_dpix_depth, float* inst_grad, uint32_t* clear_word, int debug, hipStream_t s)
{
    GS4D_PROF("blend_backward", s);
    const int gx = (W + GS4D_TILE_X - 1) / GS4D_TILE_X, gy = (H + GS4D_TILE_Y - 1) / GS4D_TILE_Y;
    static const bool exact = getenv("GS4D_EXACT_EXP") != nullptr;  // diagnostic switch: libm-accurate exp
    (void)R;
    hipLaunchKernelGGL(exact ? blend_backward_kernel<true> : blend_backward_kernel<false>, dim3((unsigned)(gx * gy + (int)b.seg_slots)),
                       dim3(BWD_THREADS), 0, s, W, H, gx, im.ranges,
                       b.point_list(), im.tile_max_contrib, bg, g.means2D, g.conic_opacity, features, g.depths,
                       im.final_T, im.n_contrib, dL_dpix, dL_dpix_depth, radii, g.point_offsets, gy, inst_grad, clear_word,
                       b.ckpt, b.seg_owner, (int)b.seg_slots);
    GS4D_LAUNCH_CHECK("blend_backward", debug, s);
    return GS4D_OK;
}
