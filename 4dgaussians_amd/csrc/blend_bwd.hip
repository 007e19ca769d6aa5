// Backward 16x16-tile blend for gfx950.
//
// Behaviour matched: RAST/cuda_rasterizer/backward.cu:415-590 of the reference
// (RAST = submodules/depth-diff-gaussian-rasterization): back-to-front replay of every pixel's
// blend using final_T and n_contrib, producing for every Gaussian the sums over pixels of the
// gradients w.r.t. colour (3), depth (1), 2D mean (2, scaled by 0.5 W / 0.5 H), conic (3) and
// opacity (1).
//
// The reference issues 10 global float atomics per contributing (pixel, Gaussian) pair.  Here:
//  * each wave64 (one 8x8 pixel quadrant) reduces its 64 per-lane contributions with 6 DPP adds
//    per value and lane 63 stores the wave partial into LDS (one private row per wave: no
//    LDS atomics);
//  * after a batch of <= 256 list entries the four wave partials are added in fixed order and
//    the tile's partial row (12 floats, 10 used) is written to inst_grad[sorted position],
//    i.e. one coalesced 48-byte row per (tile, Gaussian) instance and NO global atomics;
//  * the per-Gaussian kernel (geom_bwd.hip) gathers a Gaussian's rows through the inverse
//    permutation in fixed order, so gradients are bitwise reproducible run to run.
// Rows of list entries that no pixel of the tile reached (position >= tile_max_contrib) are
// zero-filled without being evaluated.
#include <stdlib.h>

#include "gs4d_internal.h"

namespace {

template <bool EXACT>
__device__ __forceinline__ float fast_exp(float x)
{
    if (EXACT) return expf(x);
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false);
    return v + __int_as_float(moved);
}

// Sum over the 64 lanes of the wave; the total is valid in lane 63 only.
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0xB1>(v);             // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);             // quad_perm [2,3,0,1]
    v = dpp_add<0x114>(v);            // row_shr:4
    v = dpp_add<0x118>(v);            // row_shr:8  -> lanes 12..15 of each row hold the row sum
    v = dpp_add<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);       // row_bcast:31 into rows 2 and 3 -> lane 63 = total
    return v;
}

#define NQ 10  // gradient values per (tile, Gaussian) instance

template <bool EXACT>
__global__ __launch_bounds__(GS4D_TILE_PIX) void blend_backward_kernel(
    int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const uint32_t* __restrict__ tile_max_contrib, const float* __restrict__ bg, const float2* __restrict__ means2D,
    const float4* __restrict__ conic_opacity, const float* __restrict__ colors, const float* __restrict__ depths,
    const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dpixel_depths, float* __restrict__ inst_grad)
{
    __shared__ float4 s_a[GS4D_TILE_PIX];   // x, y, conic.x, conic.y
    __shared__ float4 s_b[GS4D_TILE_PIX];   // conic.z, opacity, depth, r
    __shared__ float2 s_c[GS4D_TILE_PIX];   // g, b
    __shared__ float s_part[4][GS4D_TILE_PIX][NQ];  // per-wave partial sums of the current batch (40 KB)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile_x = blockIdx.x, tile_y = blockIdx.y;
    const int tile = tile_y * gx + tile_x;
    const int px = tile_x * GS4D_TILE_X + ((wave & 1) << 3) + (lane & 7);
    const int py = tile_y * GS4D_TILE_Y + ((wave >> 1) << 3) + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const int n_total = (int)(range.y - range.x);
    if (n_total <= 0) return;
    const int n_live = (int)min(tile_max_contrib[tile], (uint32_t)n_total);  // entries some pixel blended

    float4* __restrict__ rows = reinterpret_cast<float4*>(inst_grad) + (size_t)range.x * (GS4D_GRAD_ROW / 4);
    // zero-fill rows of entries nobody reached
    for (int i = n_live * (GS4D_GRAD_ROW / 4) + tid; i < n_total * (GS4D_GRAD_ROW / 4); i += GS4D_TILE_PIX)
        rows[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n_live == 0) return;

    const size_t pix = (size_t)py * W + px;
    const size_t plane = (size_t)H * W;
    const float T_final = inside ? final_Ts[pix] : 0.f;
    float T = T_final;
    const int last_contributor = inside ? (int)n_contrib[pix] : 0;
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f, dLpd = 0.f;
    if (inside) {
        dLp0 = dL_dpixels[pix];
        dLp1 = dL_dpixels[plane + pix];
        dLp2 = dL_dpixels[2 * plane + pix];
        dLpd = dL_dpixel_depths[pix];
    }
    float bg_dot_dpixel = 0.f;
    bg_dot_dpixel += bg[0] * dLp0;
    bg_dot_dpixel += bg[1] * dLp1;
    bg_dot_dpixel += bg[2] * dLp2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accd = 0.f;
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_depth = 0.f;
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);

    float* part = &s_part[0][0][0];
    // batches walk the live part of the list from the back; each covers positions [lo, lo + n)
    for (int hi = n_live; hi > 0; hi -= GS4D_TILE_PIX) {
        const int lo = max(0, hi - GS4D_TILE_PIX);
        const int n = hi - lo;
        __syncthreads();  // previous batch's rows have been flushed
        for (int i = tid; i < 4 * GS4D_TILE_PIX * NQ; i += GS4D_TILE_PIX) part[i] = 0.f;
        if (tid < n) {
            const uint32_t id = point_list[range.x + lo + tid];
            const float2 xy = means2D[id];
            const float4 co = conic_opacity[id];
            const float d = depths[id];
            const float r = colors[3 * (size_t)id + 0], g = colors[3 * (size_t)id + 1], b = colors[3 * (size_t)id + 2];
            s_a[tid] = make_float4(xy.x, xy.y, co.x, co.y);
            s_b[tid] = make_float4(co.z, co.w, d, r);
            s_c[tid] = make_float2(g, b);
        }
        __syncthreads();

        for (int j = n - 1; j >= 0; j--) {
            const int pos = lo + j;  // 0-based list position == the reference's `contributor` after decrement
            const float4 a = s_a[j];
            const float4 b = s_b[j];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            const float G = fast_exp<EXACT>(power);
            const float alpha = fminf(0.99f, b.y * G);
            const bool use = pos < last_contributor && !(power > 0.0f) && !(alpha < (1.0f / 255.0f));
            if (__ballot(use) == 0ull) continue;

            float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f, v9 = 0.f;
            if (use) {
                const float2 c = s_c[j];
                const float one_minus_alpha = 1.f - alpha;
                T = T / one_minus_alpha;
                const float w = alpha * T;
                float dL_dalpha = 0.0f;
                acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0;
                lc0 = b.w;
                dL_dalpha += (b.w - acc0) * dLp0;
                v0 = w * dLp0;
                acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1;
                lc1 = c.x;
                dL_dalpha += (c.x - acc1) * dLp1;
                v1 = w * dLp1;
                acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2;
                lc2 = c.y;
                dL_dalpha += (c.y - acc2) * dLp2;
                v2 = w * dLp2;
                accd = last_alpha * last_depth + (1.f - last_alpha) * accd;
                last_depth = b.z;
                dL_dalpha += (b.z - accd) * dLpd;
                v3 = w * dLpd;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / one_minus_alpha) * bg_dot_dpixel;
                const float dL_dG = b.y * dL_dalpha;
                const float gdx = G * dx;
                const float gdy = G * dy;
                const float dG_ddelx = -gdx * a.z - gdy * a.w;
                const float dG_ddely = -gdy * b.x - gdx * a.w;
                v4 = dL_dG * dG_ddelx * ddelx_dx;
                v5 = dL_dG * dG_ddely * ddely_dy;
                v6 = -0.5f * gdx * dx * dL_dG;
                v7 = -0.5f * gdx * dy * dL_dG;
                v8 = -0.5f * gdy * dy * dL_dG;
                v9 = G * dL_dalpha;
            }
            v0 = wave_sum_to_lane63(v0);
            v1 = wave_sum_to_lane63(v1);
            v2 = wave_sum_to_lane63(v2);
            v3 = wave_sum_to_lane63(v3);
            v4 = wave_sum_to_lane63(v4);
            v5 = wave_sum_to_lane63(v5);
            v6 = wave_sum_to_lane63(v6);
            v7 = wave_sum_to_lane63(v7);
            v8 = wave_sum_to_lane63(v8);
            v9 = wave_sum_to_lane63(v9);
            if (lane == 63) {
                float* p = &s_part[wave][j][0];
                p[0] = v0; p[1] = v1; p[2] = v2; p[3] = v3; p[4] = v4;
                p[5] = v5; p[6] = v6; p[7] = v7; p[8] = v8; p[9] = v9;
            }
        }
        __syncthreads();
        // flush: fixed-order sum of the four wave partials, one 48-byte row per list entry
        if (tid < n) {
            float r_[GS4D_GRAD_ROW];
#pragma unroll
            for (int q = 0; q < NQ; q++)
                r_[q] = ((s_part[0][tid][q] + s_part[1][tid][q]) + s_part[2][tid][q]) + s_part[3][tid][q];
            r_[10] = 0.f;
            r_[11] = 0.f;
            float4* dst = rows + (size_t)(lo + tid) * (GS4D_GRAD_ROW / 4);
            dst[0] = make_float4(r_[0], r_[1], r_[2], r_[3]);
            dst[1] = make_float4(r_[4], r_[5], r_[6], r_[7]);
            dst[2] = make_float4(r_[8], r_[9], r_[10], r_[11]);
        }
    }
}

}  // namespace

int gs4d_launch_blend_backward(int W, int H, int R, const GeomState& g, const BinningState& b, const ImageState& im,
                               const float* features, const float* bg, const float* dL_dpix,
                               const float* dL_dpix_depth, float* inst_grad, int debug, hipStream_t s)
{
    (void)R;
    GS4D_PROF("blend_backward", s);
    const int gx = (W + GS4D_TILE_X - 1) / GS4D_TILE_X, gy = (H + GS4D_TILE_Y - 1) / GS4D_TILE_Y;
    static const bool exact = getenv("GS4D_EXACT_EXP") != nullptr;  // diagnostic switch: libm-accurate exp
    hipLaunchKernelGGL(exact ? blend_backward_kernel<true> : blend_backward_kernel<false>, dim3(gx, gy), dim3(GS4D_TILE_PIX), 0, s, W, H, gx, im.ranges,
                       b.point_list(), im.tile_max_contrib, bg, g.means2D, g.conic_opacity, features, g.depths,
                       im.final_T, im.n_contrib, dL_dpix, dL_dpix_depth, inst_grad);
    GS4D_LAUNCH_CHECK("blend_backward", debug, s);
    return GS4D_OK;
}
