// Device-side float32 math for the per-Gaussian kernels.  Translation units that include
// this header are compiled with -ffp-contract=off: radii, tile rectangles and sort keys are
// integers decided by float32 arithmetic, and the parity contract is bit-exactness
// against an un-contracted IEEE evaluation in a fixed operation order.
//
// Behaviour being matched (reference, RAST = submodules/depth-diff-gaussian-rasterization):
//   RAST/cuda_rasterizer/auxiliary.h:22-132   constants, ndc2Pix, getRect, transforms, dnormvdv
//   RAST/third_party/glm                      column-major mat3, product order of type_mat3x3.inl:486-519
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gs4d {

__device__ __constant__ const float kSH_C0 = 0.28209479177387814f;
__device__ __constant__ const float kSH_C1 = 0.4886025119029199f;
__device__ __constant__ const float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                                  -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ const float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                                  0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                                  -0.5900435899266435f};

// Column-major 3x3: e[c][r] is column c, row r.
struct Mat3 {
    float e[3][3];
};

__device__ __forceinline__ Mat3 mat3_from_columns(float a, float b, float c, float d, float e, float f, float g,
                                                  float h, float i)
{
    Mat3 m;
    m.e[0][0] = a; m.e[0][1] = b; m.e[0][2] = c;
    m.e[1][0] = d; m.e[1][1] = e; m.e[1][2] = f;
    m.e[2][0] = g; m.e[2][1] = h; m.e[2][2] = i;
    return m;
}

// out[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2], summed left to right.
__device__ __forceinline__ Mat3 mat3_product(const Mat3& A, const Mat3& B)
{
    Mat3 o;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            o.e[c][r] = A.e[0][r] * B.e[c][0] + A.e[1][r] * B.e[c][1] + A.e[2][r] * B.e[c][2];
    return o;
}

__device__ __forceinline__ Mat3 mat3_transposed(const Mat3& A)
{
    Mat3 o;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) o.e[c][r] = A.e[r][c];
    return o;
}

// The matrices arrive as 16 floats of a transposed 4x4 (scene/cameras.py:103-105), so element
// [k*4 + j] multiplies input coordinate k and feeds output coordinate j.
__device__ __forceinline__ float3 affine_point(const float3 p, const float* __restrict__ m)
{
    float3 o;
    o.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    o.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    o.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    return o;
}

__device__ __forceinline__ float4 projective_point(const float3 p, const float* __restrict__ m)
{
    float4 o;
    o.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    o.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    o.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    o.w = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
    return o;
}

// NDC -> pixel centre, evaluated in double then narrowed (auxiliary.h:41-44 uses double literals).
__device__ __forceinline__ float ndc_to_pixel(float v, int S)
{
    return (float)(((v + 1.0) * S - 1.0) * 0.5);
}

// Tile rectangle [lo, hi) of a disc of integer radius around p, clamped to the tile grid
// (auxiliary.h:46-56).  float -> int conversion saturates on this hardware (v_cvt_i32_f32).
__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int& x0, int& y0,
                                          int& x1, int& y1)
{
    x0 = min(gx, max(0, (int)((px - radius) / GS4D_TILE_X)));
    y0 = min(gy, max(0, (int)((py - radius) / GS4D_TILE_Y)));
    x1 = min(gx, max(0, (int)((px + radius + GS4D_TILE_X - 1) / GS4D_TILE_X)));
    y1 = min(gy, max(0, (int)((py + radius + GS4D_TILE_Y - 1) / GS4D_TILE_Y)));
}

// Everything the EWA projection needs in both directions: clamped view-space mean t, the
// un-clamped ratios, T = W*J, the symmetric world covariance as a matrix, W, and cov2D
// before the 0.3 low-pass (forward.cu:74-106 / backward.cu:166-194).
struct Ewa {
    float3 t;
    float txtz, tytz;
    Mat3 T, V, Wm, cov;
};

__device__ __forceinline__ Ewa ewa_project(const float3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                                           const float* __restrict__ c3, const float* __restrict__ view)
{
    Ewa o;
    float3 t = affine_point(mean, view);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    o.txtz = t.x / t.z;
    o.tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, o.txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, o.tytz)) * t.z;
    o.t = t;
    const Mat3 J = mat3_from_columns(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z),
                                     0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z),
                                     0.0f, 0.0f, 0.0f);
    o.Wm = mat3_from_columns(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    o.T = mat3_product(o.Wm, J);
    o.V = mat3_from_columns(c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]);
    const Mat3 Tt = mat3_transposed(o.T);
    const Mat3 Vt = mat3_transposed(o.V);
    const Mat3 tmp = mat3_product(Tt, Vt);
    o.cov = mat3_product(tmp, o.T);
    return o;
}

// Rotation matrix columns from the (r,x,y,z) quaternion, NOT normalised (forward.cu:127).
__device__ __forceinline__ Mat3 quat_columns(const float4 q)
{
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    return mat3_from_columns(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

__device__ __forceinline__ Mat3 scale_columns(const float3 s)
{
    return mat3_from_columns(s.x, 0.f, 0.f, 0.f, s.y, 0.f, 0.f, 0.f, s.z);
}

// Axis-aligned box around a splat outside of which alpha = opacity * exp(power) < 1/255 for certain, i.e.
// where both blend kernels skip the splat (forward.cu:343-346, backward.cu:515-518 of the reference).
// alpha >= 1/255  <=>  0.5 (cx dx^2 + 2 cy dx dy + cz dy^2) <= t,  t = ln(255 opacity); the ellipse's
// half extents are sqrt(2 t cz / det) and sqrt(2 t cx / det), det = cx cz - cy^2.  The box is inflated
// (0.1 % + 0.05 px) so float rounding in the kernels' own alpha can never disagree with it; a splat
// that cannot reach 1/255 anywhere (t < 0) gets an empty box.  Anything NaN (opacity, conic, mean) or degenerate gets
// the never-cull box: the reference's tests `power > 0` and `alpha < 1/255` are both false for NaN, so it BLENDS such
// a splat and the NaN reaches the image -- a diverged model must stay visible here too.
// Returns (x - hx, x + hx, y - hy, y + hy).
__device__ __forceinline__ float4 alpha_support_box(float2 xy, float4 conic_opacity)
{
    const float cx = conic_opacity.x, cy = conic_opacity.y, cz = conic_opacity.z;
    const float4 never_cull = make_float4(-3.0e38f, 3.0e38f, -3.0e38f, 3.0e38f);
    const float t = logf(255.0f * conic_opacity.w);
    if (t < 0.0f) return make_float4(3.0e38f, -3.0e38f, 3.0e38f, -3.0e38f);
    const float det = cx * cz - cy * cy;
    if (!(t >= 0.0f) || !(det > 0.0f)) return never_cull;                 // NaN opacity / NaN or degenerate conic
    const float k = 2.0f * t / det;
    const float hx = sqrtf(k * cz) * 1.001f + 0.05f, hy = sqrtf(k * cx) * 1.001f + 0.05f;
    const float4 box = make_float4(xy.x - hx, xy.x + hx, xy.y - hy, xy.y + hy);
    if (!(box.x <= box.y) || !(box.z <= box.w)) return never_cull;         // NaN mean or extent
    return box;
}

// Exact version of the same question for one rectangle of pixel centres [rx0, rx1] x [ry0, ry1]: true only if the
// splat PROVABLY reaches alpha >= 1/255 at no point of the rectangle (so at none of its pixels), i.e. if the minimum of
// f(d) = 0.5 (cx dx^2 + cz dy^2) + cy dx dy over d = mean - pixel exceeds t = ln(255 opacity).  f is convex: its minimum
// over the rectangle is 0 if the mean lies inside, otherwise it sits on one of the four edges, where it is a clamped
// one-dimensional parabola.  The box above is what a diagonal, elongated splat wastes most on: a third of the
// (wave, entry) visits that passed the box test at C2 blended nothing.  Conservative against float rounding: the
// rectangle is inflated by 0.05 px, the threshold by 0.1 % + 1e-4, and 4e-6 of the magnitude of f's terms is taken off
// (the kernels' own power carries ~1e-7 of them).  NaN or a non-positive-definite conic never culls.
__device__ __forceinline__ bool alpha_support_misses_rect(float mx, float my, float cx, float cy, float cz, float opacity,
                                                          float rx0, float rx1, float ry0, float ry1)
{
    if (!(cx > 0.0f && cz > 0.0f && cx * cz - cy * cy > 0.0f)) return false;
    const float t = logf(255.0f * opacity);
    if (!(t >= 0.0f)) return false;                                  // t < 0: the box is already empty; NaN: never cull
    const float xl = mx - (rx1 + 0.05f), xh = mx - (rx0 - 0.05f);    // range of dx over the rectangle
    const float yl = my - (ry1 + 0.05f), yh = my - (ry0 - 0.05f);
    if (xl <= 0.0f && xh >= 0.0f && yl <= 0.0f && yh >= 0.0f) return false;    // mean inside
    const float rcz = 1.0f / cz, rcx = 1.0f / cx;
    float best = 3.0e38f;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        {   // edge dx = X
            const float X = k ? xh : xl;
            const float Y = fminf(fmaxf(-cy * X * rcz, yl), yh);
            const float q = 0.5f * (cx * X * X + cz * Y * Y), m = cy * X * Y;
            best = fminf(best, (q + m) - 4.0e-6f * (q + fabsf(m)));
        }
        {   // edge dy = Y
            const float Y = k ? yh : yl;
            const float X = fminf(fmaxf(-cy * Y * rcx, xl), xh);
            const float q = 0.5f * (cx * X * X + cz * Y * Y), m = cy * X * Y;
            best = fminf(best, (q + m) - 4.0e-6f * (q + fabsf(m)));
        }
    }
    return best > t * 1.001f + 1.0e-4f;                              // false for NaN
}

// Activation backward of ONE Gaussian (scene/gaussian_model.py:38-43 differentiated: exp, L2-normalise, sigmoid): the body
// of activate_backward_kernel (csrc/elementwise.hip), also called by the per-Gaussian rasterizer backward when the
// composite call folds the activation backward into it (csrc/geom_bwd.hip).  Inputs: the forward OUTPUTS (scales, opacity)
// and the raw rotation; any upstream gradient array may be NULL.
__device__ __forceinline__ void activation_backward_point(int i, const float* __restrict__ scales, const float4* __restrict__ rot,
                                                          const float* __restrict__ opacity, const float* g_scales,
                                                          const float4* g_rot, const float* g_opacity,
                                                          float* __restrict__ d_log_scales, float4* __restrict__ d_rot,
                                                          float* __restrict__ d_opacity_logit)
{
#pragma clang fp contract(off)         // the same rounding in every translation unit that calls it (they differ in -ffp-contract)
#pragma unroll
    for (int a = 0; a < 3; a++)
        d_log_scales[3 * (size_t)i + a] = g_scales ? g_scales[3 * (size_t)i + a] * scales[3 * (size_t)i + a] : 0.f;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g_rot) {
        const float4 q = rot[i], g = g_rot[i];
        const float len = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        if (len > 1e-12f) {
            const float inv = 1.0f / len;
            const float4 u = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
            const float ug = u.x * g.x + u.y * g.y + u.z * g.z + u.w * g.w;
            d = make_float4((g.x - u.x * ug) * inv, (g.y - u.y * ug) * inv, (g.z - u.z * ug) * inv, (g.w - u.w * ug) * inv);
        } else {
            d = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);   // the clamp is active: q / eps
        }
    }
    d_rot[i] = d;
    const float o = opacity[i];
    d_opacity_logit[i] = g_opacity ? g_opacity[i] * o * (1.0f - o) : 0.f;
}

// sum of v over the 256 threads of the workgroup (s: 256 floats of LDS), a fixed tree: every caller gets the same rounding
__device__ __forceinline__ float block_sum_256(float v, float* s)
{
    const int tid = threadIdx.x;
    s[tid] = v;
    __syncthreads();
#pragma unroll
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) s[tid] += s[tid + off];
        __syncthreads();
    }
    return s[0];
}

// loss[0] = (sum of the GS4D_L1_BLOCKS partial sums) / n, by one 256-thread workgroup: l1_final_kernel's arithmetic
__device__ __forceinline__ void l1_finish(const float* __restrict__ partial, size_t n, float* __restrict__ loss, float* s256)
{
    float acc = 0.f;
    for (int k = threadIdx.x; k < 512; k += 256) acc += partial[k];
    const float t = block_sum_256(acc, s256);
    if (threadIdx.x == 0) loss[0] = t / (float)n;
}

}  // namespace gs4d
