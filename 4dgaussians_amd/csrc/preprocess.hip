// Per-Gaussian forward preprocess for gfx950: near cull, projection, 3D covariance, EWA
// 2D covariance + conic, screen radius, tile rectangle, SH -> RGB, and the per-workgroup
// tile-count sums that seed the instance-offset scan.
//
// Behaviour matched: RAST/cuda_rasterizer/forward.cu:20-71,74-113,118-152,155-256 and
// auxiliary.h:139-164 (RAST = submodules/depth-diff-gaussian-rasterization of the reference).
// Compiled with -ffp-contract=off (see gs4d_math.h).
//
// Layout/launch: one Gaussian per lane, 256 lanes (4 wave64) per workgroup; all outputs are
// SoA arrays in the geometry chunk so consecutive lanes write consecutive addresses.  HBM
// bound: 236 B read + 75 B written per Gaussian.
#include "gs4d_internal.h"
#include "gs4d_math.h"
#include "tile_place.h"

using namespace gs4d;

namespace {

__device__ __forceinline__ float3 sh_to_rgb(int deg, const float* sh, const float3 dir, bool* neg)
{
    float out[3];
    const float x = dir.x, y = dir.y, z = dir.z;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float v = kSH_C0 * sh[0 * 3 + c];
        if (deg > 0) {
            v = v - kSH_C1 * y * sh[1 * 3 + c] + kSH_C1 * z * sh[2 * 3 + c] - kSH_C1 * x * sh[3 * 3 + c];
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z;
                const float xy = x * y, yz = y * z, xz = x * z;
                v = v + kSH_C2[0] * xy * sh[4 * 3 + c] + kSH_C2[1] * yz * sh[5 * 3 + c] +
                    kSH_C2[2] * (2.0f * zz - xx - yy) * sh[6 * 3 + c] + kSH_C2[3] * xz * sh[7 * 3 + c] +
                    kSH_C2[4] * (xx - yy) * sh[8 * 3 + c];
                if (deg > 2) {
                    v = v + kSH_C3[0] * y * (3.0f * xx - yy) * sh[9 * 3 + c] + kSH_C3[1] * xy * z * sh[10 * 3 + c] +
                        kSH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11 * 3 + c] +
                        kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * 3 + c] +
                        kSH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13 * 3 + c] +
                        kSH_C3[5] * z * (xx - yy) * sh[14 * 3 + c] + kSH_C3[6] * x * (xx - 3.0f * yy) * sh[15 * 3 + c];
                }
            }
        }
        v += 0.5f;
        neg[c] = v < 0.0f;
        out[c] = fmaxf(v, 0.0f);
    }
    return make_float3(out[0], out[1], out[2]);
}

__global__ __launch_bounds__(GS4D_PRE_BLOCK) void preprocess_kernel(
    int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
    const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ cam_pos,
    int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y, int gx, int gy, int prefiltered,
    int* __restrict__ radii, float2* __restrict__ means2D, float* __restrict__ depths, float* __restrict__ cov3Ds,
    float* __restrict__ rgb, float4* __restrict__ conic_opacity, uint8_t* __restrict__ clamped,
    uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ block_sums, uint32_t* __restrict__ block_area,
    uint32_t* __restrict__ tile_count, uint8_t* __restrict__ visible)
{
    __shared__ float s_view[16], s_proj[16], s_cam[3];
    __shared__ uint32_t s_wave_sum[GS4D_PRE_BLOCK / 64];
    if (threadIdx.x < 16) {
        s_view[threadIdx.x] = viewmatrix[threadIdx.x];
        s_proj[threadIdx.x] = projmatrix[threadIdx.x];
    }
    if (threadIdx.x < 3) s_cam[threadIdx.x] = cam_pos[threadIdx.x];
    __syncthreads();

    const int idx = blockIdx.x * GS4D_PRE_BLOCK + threadIdx.x;
    uint32_t my_tiles = 0;
    int my_radius_i = 0;
    int my_x0 = 0, my_y0 = 0, my_x1 = 1, my_y1 = 1;
    float my_depth = 0.f;
    float2 my_mean = make_float2(0.f, 0.f);
    float4 my_conic = make_float4(0.f, 0.f, 0.f, 0.f);
    int bad = 0;
    // Every input of the Gaussian is requested up front, before the first test that could skip it: under the near-plane /
    // empty-rectangle branches the scales, the rotation, the opacity and the 48 SH floats were three memory latencies in
    // a row (17.5 us at 100 k Gaussians, where the data needs 5).  The arithmetic is unchanged.
    const int ld = min(idx, P - 1);
    const float3 p_in = make_float3(means3D[3 * ld], means3D[3 * ld + 1], means3D[3 * ld + 2]);
    float3 sc_in = make_float3(0.f, 0.f, 0.f);
    float4 q_in = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cov3D_precomp == nullptr) {                     // uniform
        sc_in = make_float3(scales[3 * ld], scales[3 * ld + 1], scales[3 * ld + 2]);
        q_in = reinterpret_cast<const float4*>(rotations)[ld];
    }
    const float op_in = opacities[ld];
    float sh_in[48];
    const bool sh_regs = colors_precomp == nullptr && M == 16 && (((uintptr_t)shs) & 15) == 0;   // uniform
    if (sh_regs) {
        const float4* __restrict__ s4 = reinterpret_cast<const float4*>(shs + (size_t)ld * 48);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const float4 t = s4[k];
            sh_in[4 * k] = t.x; sh_in[4 * k + 1] = t.y; sh_in[4 * k + 2] = t.z; sh_in[4 * k + 3] = t.w;
        }
    }
    if (idx < P) {
        const float3 p_orig = p_in;
        const float3 p_view = affine_point(p_orig, s_view);
        if (p_view.z > 0.2f) {
            const float4 p_hom = projective_point(p_orig, s_proj);
            const float p_w = 1.0f / (p_hom.w + 0.0000001f);
            const float ndc_x = p_hom.x * p_w, ndc_y = p_hom.y * p_w;

            float c3[6];
            if (cov3D_precomp != nullptr) {
#pragma unroll
                for (int k = 0; k < 6; k++) c3[k] = cov3D_precomp[6 * (size_t)idx + k];
            } else {
                const float3 sc = make_float3(scale_modifier * sc_in.x, scale_modifier * sc_in.y, scale_modifier * sc_in.z);
                const float4 q = q_in;
                const Mat3 Mm = mat3_product(scale_columns(sc), quat_columns(q));
                const Mat3 Sigma = mat3_product(mat3_transposed(Mm), Mm);
                c3[0] = Sigma.e[0][0]; c3[1] = Sigma.e[0][1]; c3[2] = Sigma.e[0][2];
                c3[3] = Sigma.e[1][1]; c3[4] = Sigma.e[1][2]; c3[5] = Sigma.e[2][2];
#pragma unroll
                for (int k = 0; k < 6; k++) cov3Ds[6 * (size_t)idx + k] = c3[k];
            }
            const Ewa e = ewa_project(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, c3, s_view);
            const float cx = e.cov.e[0][0] + 0.3f;
            const float cy = e.cov.e[0][1];
            const float cz = e.cov.e[1][1] + 0.3f;
            const float det = (cx * cz - cy * cy);
            if (det != 0.0f) {
                const float det_inv = 1.f / det;
                const float mid = 0.5f * (cx + cz);
                const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
                const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
                const float pix_x = ndc_to_pixel(ndc_x, W), pix_y = ndc_to_pixel(ndc_y, H);
                int x0, y0, x1, y1;
                tile_rect(pix_x, pix_y, (int)my_radius, gx, gy, x0, y0, x1, y1);
                const uint32_t ntiles = (uint32_t)(x1 - x0) * (uint32_t)(y1 - y0);
                if (ntiles != 0) {
                    // the colour and the per-Gaussian stores come LAST (below): the tile rectangle is all the counting of
                    // the direct placement needs, and its device-scope adds retire while the colour is evaluated
                    my_depth = p_view.z;
                    my_mean = make_float2(pix_x, pix_y);
                    my_conic = make_float4(cz * det_inv, -cy * det_inv, cx * det_inv, op_in);
                    my_radius_i = (int)my_radius;
                    my_tiles = ntiles;
                    my_x0 = x0; my_y0 = y0; my_x1 = x1; my_y1 = y1;
                }
            }
        } else if (prefiltered) {
            bad = 1;  // reference traps here (auxiliary.h:156-160); we flag and report GS4D_EPREFILTER
        }
        radii[idx] = my_radius_i;
        if (visible) visible[idx] = my_radius_i > 0;         // the caller's visibility filter (gs4d_render_desc::visible)
        tiles_touched[idx] = my_tiles;
    }

    // workgroup sum of tile counts -> block_sums[blockIdx.x]; bit 31 carries the workgroup's prefilter flag to the scan
    // kernel (a sum stays below 2^31: 256 Gaussians x tiles of the image), so no counter has to be cleared beforehand
    uint32_t v = my_tiles;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_wave_sum[threadIdx.x >> 6] = v;
    const int any_bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < GS4D_PRE_BLOCK / 64; w++) t += s_wave_sum[w];
        block_sums[blockIdx.x] = (t & 0x7fffffffu) | (any_bad ? 0x80000000u : 0u);
    }

    // The rectangle of tiles around the workgroup's Gaussians (the big splats apart): small when the rows are in a spatially
    // coherent order.  block_area = the device-scope adds direct placement would need for this workgroup per phase, at
    // most: one per tile of the rectangle or per instance of a Gaussian in it, whichever is fewer, + one per instance of a
    // big splat.  The scan launch adds them up and tells the host whether direct placement pays (binning.hip, api.hip).
    __shared__ int s_red[GS4D_PRE_BLOCK / 64][6];
    __shared__ __align__(16) uint32_t s_bins[GS4D_AGG_BINS];
    if (tile_count != nullptr) block_tile_clear(s_bins);    // uniform; the barrier inside block_tile_bbox() follows
    const bool big = my_tiles != 0u && tile_rect_is_big(my_x0, my_y0, my_x1, my_y1);
    const bool small = my_tiles != 0u && !big;
    int bx0, by0, bw, bh;
    uint32_t small_tiles = small ? my_tiles : 0u, big_tiles = big ? my_tiles : 0u;
    const bool any_small = block_tile_bbox(small, my_x0, my_y0, my_x1, my_y1, s_red, bx0, by0, bw, bh, small_tiles, big_tiles);   // barrier inside
    // (one word per wave: the exact count below needs no workgroup sum then)
    if ((threadIdx.x & 63) == 0)
        block_area[4 * blockIdx.x + (threadIdx.x >> 6)] = threadIdx.x == 0 ? (any_small ? min((uint32_t)(bw * bh), small_tiles) : 0u) + big_tiles : 0u;

    // Direct placement (tile_place.h): the (Gaussian, tile) instances are counted per tile here, so that the scan launch can
    // lay the tile lists out before the keys exist: counts of the workgroup's own Gaussians over its rectangle of tiles in
    // LDS (four adds per Gaussian), then one device-scope add per tile it touches.
    if (tile_count != nullptr) {                            // uniform
        const int tid = threadIdx.x, lane = tid & 63;
        const bool agg = any_small && bw < GS4D_AGG_SIDE && bh < GS4D_AGG_SIDE;   // uniform
        uint32_t adds = 0;                                  // device-scope adds of this lane: block_area becomes their exact number
        if (agg) {
            block_tile_counts(small, my_x0, my_y0, my_x1, my_y1, bx0, by0, bw, bh, s_bins);
            const int pitch = bw + 1, cells = pitch * bh;
            for (int b = tid; b < cells; b += GS4D_PRE_BLOCK) {
                const uint32_t c = s_bins[b];
                if (c) {
                    atomicAdd(&tile_count[(size_t)((by0 + b / pitch) * gx + bx0 + b % pitch) * GS4D_TILE_CURSOR_STRIDE], c);
                    adds++;
                }
            }
        }
        // one add per instance: the big splats (and everything, if the rows are in no spatial order), a wave per Gaussian
        unsigned long long todo = __ballot(agg ? big : my_tiles != 0u);
        while (todo) {
            const int src = __ffsll(todo) - 1;
            todo &= todo - 1ull;
            const int x0 = __shfl(my_x0, src, 64), y0 = __shfl(my_y0, src, 64), w = __shfl(my_x1, src, 64) - x0;
            const int n = (int)__shfl((int)my_tiles, src, 64);
            for (int k = lane; k < n; k += 64) {
                atomicAdd(&tile_count[(size_t)((y0 + k / w) * gx + x0 + k % w) * GS4D_TILE_CURSOR_STRIDE], 1u);
                adds++;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) adds += __shfl_xor(adds, off, 64);
        if (lane == 0) block_area[4 * blockIdx.x + (tid >> 6)] = adds;
    }

    // colour (forward.cu:161-207: same arithmetic, same order) and the per-Gaussian records of the Gaussians that touch a tile
    if (my_tiles != 0u) {
        if (colors_precomp == nullptr) {
            float3 dir = make_float3(p_in.x - s_cam[0], p_in.y - s_cam[1], p_in.z - s_cam[2]);
            const float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
            dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
            bool neg[3];
            const float3 col = sh_regs ? sh_to_rgb(D, sh_in, dir, neg) : sh_to_rgb(D, shs + (size_t)idx * M * 3, dir, neg);
            rgb[3 * (size_t)idx + 0] = col.x;
            rgb[3 * (size_t)idx + 1] = col.y;
            rgb[3 * (size_t)idx + 2] = col.z;
            clamped[3 * (size_t)idx + 0] = neg[0];
            clamped[3 * (size_t)idx + 1] = neg[1];
            clamped[3 * (size_t)idx + 2] = neg[2];
        }
        depths[idx] = my_depth;
        means2D[idx] = my_mean;
        conic_opacity[idx] = my_conic;
    }
}

__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ viewmatrix,
                                                           uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    const float3 pv = affine_point(p, viewmatrix);
    present[idx] = pv.z > 0.2f;
}

}  // namespace

int gs4d_launch_preprocess(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                           const float* rotations, const float* opacities, const float* shs,
                           const float* cov3D_precomp, const float* colors_precomp, const float* viewmatrix,
                           const float* projmatrix, const float* cam_pos, int W, int H, float tan_fovx,
                           float tan_fovy, int prefiltered, int* radii, GeomState& g, int debug, hipStream_t s,
                           uint32_t* tile_count, bool tile_count_cleared, uint8_t* visible)
{
    GS4D_PROF("preprocess", s);
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + GS4D_TILE_X - 1) / GS4D_TILE_X, gy = (H + GS4D_TILE_Y - 1) / GS4D_TILE_Y;
    const int nblk = (P + GS4D_PRE_BLOCK - 1) / GS4D_PRE_BLOCK;
    if (tile_count && !tile_count_cleared) GS4D_HIP_CHECK(hipMemsetAsync(tile_count, 0, (size_t)gx * gy * GS4D_TILE_CURSOR_STRIDE * sizeof(uint32_t), s));
    hipLaunchKernelGGL(preprocess_kernel, dim3(nblk), dim3(GS4D_PRE_BLOCK), 0, s, P, D, M, means3D, scales,
                       scale_modifier, rotations, opacities, shs, cov3D_precomp, colors_precomp, viewmatrix,
                       projmatrix, cam_pos, W, H, tan_fovx, tan_fovy, focal_x, focal_y, gx, gy, prefiltered, radii,
                       g.means2D, g.depths, g.cov3D, g.rgb, g.conic_opacity, g.clamped, g.tiles_touched,
                       g.block_sums, g.block_area, tile_count, visible);
    GS4D_LAUNCH_CHECK("preprocess", debug, s);
    return GS4D_OK;
}

int gs4d_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
    GS4D_LAUNCH_CHECK("mark_visible", 0, s);
    return GS4D_OK;
}
