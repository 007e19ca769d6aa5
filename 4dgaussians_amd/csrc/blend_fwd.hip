// Forward 16x16-tile alpha blend for gfx950.
//
// Behaviour matched: RAST/cuda_rasterizer/forward.cu:261-379 of the reference
// (RAST = submodules/depth-diff-gaussian-rasterization): front-to-back blending of each tile's
// depth-sorted Gaussian list, alpha = min(0.99, o * exp(power)), skip power > 0 and
// alpha < 1/255, stop a pixel when T*(1-alpha) < 1e-4; outputs colour (+T*bg), depth (no bg
// term), final T and n_contrib = list position of the pixel's last contributor.
//
// gfx950 design:
//  * one workgroup = one tile = 4 wave64; each wave owns an 8x8 pixel quadrant (better wave-level
//    coherence for early termination than 16x4 strips; results do not depend on the mapping);
//  * each batch of 256 list entries is gathered ONCE per tile into LDS as three arrays
//    (xy+conic.xy | conic.z,opacity,depth,r | g,b) so the inner loop issues two broadcast
//    ds_read_b128 for the alpha test and touches the colour words only when some lane of the
//    wave actually blends the Gaussian (wave ballot);
//  * early termination is decided per wave with ballots, per tile with one LDS word per wave;
//  * exp() is evaluated as v_exp_f32(power * log2 e): |power| <= 5.55 for every alpha that
//    survives the 1/255 test, so the relative error stays < 1e-6 (tolerance: 1e-5 abs on colour).
#include <stdlib.h>

#include "gs4d_internal.h"

namespace {

template <bool EXACT>
__device__ __forceinline__ float fast_exp(float x)
{
    if (EXACT) return expf(x);
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

template <bool EXACT>
__global__ __launch_bounds__(GS4D_TILE_PIX) void blend_forward_kernel(
    int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float2* __restrict__ means2D, const float* __restrict__ features, const float* __restrict__ depths,
    const float4* __restrict__ conic_opacity, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_max_contrib, float* __restrict__ out_color,
    float* __restrict__ out_depth)
{
    __shared__ float4 s_a[GS4D_TILE_PIX];   // x, y, conic.x, conic.y
    __shared__ float4 s_b[GS4D_TILE_PIX];   // conic.z, opacity, depth, r
    __shared__ float2 s_c[GS4D_TILE_PIX];   // g, b
    __shared__ int s_wave_done[4];
    __shared__ uint32_t s_wave_max[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile_x = blockIdx.x, tile_y = blockIdx.y;
    // wave -> 8x8 quadrant, lane -> pixel inside it
    const int px = tile_x * GS4D_TILE_X + ((wave & 1) << 3) + (lane & 7);
    const int py = tile_y * GS4D_TILE_Y + ((wave >> 1) << 3) + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;

    const uint2 range = ranges[tile_y * gx + tile_x];
    const int n_total = (int)(range.y - range.x);

    bool done = !inside;
    float T = 1.0f;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last_contributor = 0;

    for (int base = 0; base < n_total; base += GS4D_TILE_PIX) {
        // tile-wide early exit: every wave publishes whether all of its pixels are finished
        const bool wave_done = __ballot(!done) == 0ull;
        if (lane == 0) s_wave_done[wave] = wave_done;
        __syncthreads();
        if (s_wave_done[0] & s_wave_done[1] & s_wave_done[2] & s_wave_done[3]) break;

        const int n = min(GS4D_TILE_PIX, n_total - base);
        if (tid < n) {
            const uint32_t id = point_list[range.x + base + tid];
            const float2 xy = means2D[id];
            const float4 co = conic_opacity[id];
            const float d = depths[id];
            const float r = features[3 * (size_t)id + 0], g = features[3 * (size_t)id + 1],
                        b = features[3 * (size_t)id + 2];
            s_a[tid] = make_float4(xy.x, xy.y, co.x, co.y);
            s_b[tid] = make_float4(co.z, co.w, d, r);
            s_c[tid] = make_float2(g, b);
        }
        __syncthreads();

        if (!wave_done) {
            for (int j = 0; j < n; j++) {
                const float4 a = s_a[j];
                const float4 b = s_b[j];
                const float dx = a.x - pxf, dy = a.y - pyf;
                const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                const float alpha = fminf(0.99f, b.y * fast_exp<EXACT>(power));
                const float test_T = T * (1.0f - alpha);
                bool use = !done && !(power > 0.0f) && !(alpha < (1.0f / 255.0f));
                if (use && test_T < 0.0001f) {
                    done = true;
                    use = false;
                }
                if (__ballot(use) != 0ull) {
                    const float2 c = s_c[j];
                    if (use) {
                        const float w = alpha * T;
                        C0 += b.w * w;
                        C1 += c.x * w;
                        C2 += c.y * w;
                        Dp += b.z * w;
                        T = test_T;
                        last_contributor = (uint32_t)(base + j + 1);
                    }
                } else if (__ballot(!done) == 0ull) {
                    break;  // every pixel of this wave is finished
                }
            }
        }
        __syncthreads();  // the staging arrays are rewritten by the next batch
    }

    if (inside) {
        const size_t pix = (size_t)py * W + px;
        const size_t plane = (size_t)H * W;
        final_T[pix] = T;
        n_contrib[pix] = last_contributor;
        out_color[pix] = C0 + T * bg[0];
        out_color[plane + pix] = C1 + T * bg[1];
        out_color[2 * plane + pix] = C2 + T * bg[2];
        out_depth[pix] = Dp;
    }
    // per-tile maximum of n_contrib: lets the backward skip list entries no pixel reached
    uint32_t m = inside ? last_contributor : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_down(m, off, 64));
    if (lane == 0) s_wave_max[wave] = m;
    __syncthreads();
    if (tid == 0)
        tile_max_contrib[tile_y * gx + tile_x] = max(max(s_wave_max[0], s_wave_max[1]), max(s_wave_max[2], s_wave_max[3]));
}

}  // namespace

int gs4d_launch_blend_forward(int W, int H, const GeomState& g, const BinningState& b, ImageState& im,
                              const float* features, const float* bg, float* out_color, float* out_depth, int debug,
                              hipStream_t s)
{
    GS4D_PROF("blend_forward", s);
    const int gx = (W + GS4D_TILE_X - 1) / GS4D_TILE_X, gy = (H + GS4D_TILE_Y - 1) / GS4D_TILE_Y;
    static const bool exact = getenv("GS4D_EXACT_EXP") != nullptr;  // diagnostic switch: libm-accurate exp
    hipLaunchKernelGGL(exact ? blend_forward_kernel<true> : blend_forward_kernel<false>, dim3(gx, gy), dim3(GS4D_TILE_PIX), 0, s, W, H, gx, im.ranges,
                       b.point_list(), g.means2D, features, g.depths, g.conic_opacity, bg, im.final_T, im.n_contrib,
                       im.tile_max_contrib, out_color, out_depth);
    GS4D_LAUNCH_CHECK("blend_forward", debug, s);
    return GS4D_OK;
}
