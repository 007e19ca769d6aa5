// The building block of the k-nearest-neighbour rigidity regulariser:
//
//     d[i][s] = || x[idx[i][s]] - x[i] ||_2                 x: [n, dim] (dim = 3 velocities, 4 quaternion deltas)
//
// which the reference writes as  torch.norm(vel[idx] - vel[None, :, None], p=2, dim=-1)
// (gaussian_renderer/renderer.py:163-165,194-195,201-202) on idx (1, n, k) int64 from knn(), k = min(100, n).
// In PyTorch that line materialises the gathered (n, k, dim) tensor and its difference (2 x 120 MB at n = 100 k,
// k = 100, dim = 3), keeps them for autograd and scatters the gradient back through index_put: about a GB of
// traffic per term, five terms per view.  Here the forward reads idx (8 B) and writes d (4 B) per pair -- x itself
// (1.2 MB) stays in L2 -- and the backward re-reads idx, d and the upstream gradient:
//     dL/dx[j] += g[i][s] * (x[j] - x[i]) / d[i][s]          (j = idx[i][s]; float atomics, ~k adds per row of x)
//     dL/dx[i] -= sum_s g[i][s] * (x[j] - x[i]) / d[i][s]    (summed over the wave first, then one atomic per column)
// The atomic scatter runs at ~20 G float atomics/s (1.5 - 2 ms per term at n = 100 k): the neighbour graph is the
// same for every term of a view, so its transpose (pairs sorted by target) is built once and the backward becomes
// a gather, one wave per row, deterministic (gs4d_neighbor_diff_norm_backward_gather below).
// A pair with d == 0 (every point is its own nearest neighbour) contributes no gradient, as torch.norm's
// backward does at zero.  HBM bound: 12 B per pair forward, 16 B per pair backward.
#include "gs4d_internal.h"

namespace {

#define RIG_MAX_DIM 4

// one thread per (i, s) pair
template <int DIM>
__global__ __launch_bounds__(256) void neighbor_diff_norm_forward_kernel(size_t pairs, int k, int n,
                                                                         const float* __restrict__ x,
                                                                         const long long* __restrict__ idx,
                                                                         float* __restrict__ d)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= pairs) return;
    const size_t i = t / (size_t)k;
    long long j = idx[t];
    if (j < 0 || j >= n) j = (long long)i;          // a malformed index degrades to a zero distance, never a wild read
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < DIM; c++) {
        const float diff = x[(size_t)j * DIM + c] - x[i * DIM + c];
        acc += diff * diff;
    }
    d[t] = sqrtf(acc);
}

// one wave per row i: lanes stride over its k neighbours
template <int DIM>
__global__ __launch_bounds__(256) void neighbor_diff_norm_backward_kernel(int n, int k, const float* __restrict__ x,
                                                                          const long long* __restrict__ idx,
                                                                          const float* __restrict__ d,
                                                                          const float* __restrict__ g,
                                                                          float* __restrict__ dx)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;                              // whole waves leave together
    float xi[DIM], own[DIM];
#pragma unroll
    for (int c = 0; c < DIM; c++) { xi[c] = x[(size_t)i * DIM + c]; own[c] = 0.f; }
    for (int s = lane; s < k; s += 64) {
        const size_t t = (size_t)i * k + s;
        const float dist = d[t];
        long long j = idx[t];
        if (!(dist > 0.f) || j < 0 || j >= n) continue;
        const float coef = g[t] / dist;
#pragma unroll
        for (int c = 0; c < DIM; c++) {
            const float v = coef * (x[(size_t)j * DIM + c] - xi[c]);
            atomicAdd(&dx[(size_t)j * DIM + c], v);
            own[c] -= v;
        }
    }
#pragma unroll
    for (int c = 0; c < DIM; c++) {
        float v = own[c];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && v != 0.f) atomicAdd(&dx[(size_t)i * DIM + c], v);
    }
}

// ---- backward without atomics, given the transposed neighbour graph ----
// order[e]: the pair indices t = i * k + s sorted by their target idx[t]; row_start[j] .. row_start[j + 1]: the pairs
// that point AT row j.  coef[t] = g[t] / d[t] (0 where d == 0) is formed in one streaming pass, so the per-row kernel
// reads one random word per incoming pair (x itself stays in L2).
__global__ __launch_bounds__(256) void neighbor_coef_kernel(size_t pairs, const float* __restrict__ d,
                                                            const float* __restrict__ g, float* __restrict__ coef)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= pairs) return;
    const float dist = d[t];
    coef[t] = dist > 0.f ? g[t] / dist : 0.f;
}

// one wave per row j: dx[j] += sum over incoming pairs (i -> j) of coef * (x[j] - x[i])
//                              - sum over its own outgoing pairs (j -> m) of coef * (x[m] - x[j])
template <int DIM>
__global__ __launch_bounds__(256) void neighbor_diff_norm_backward_gather_kernel(
    int n, int k, const float* __restrict__ x, const long long* __restrict__ idx, const float* __restrict__ coef,
    const int* __restrict__ order, const int* __restrict__ row_start, float* __restrict__ dx)
{
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    float xj[DIM], acc[DIM];
#pragma unroll
    for (int c = 0; c < DIM; c++) { xj[c] = x[(size_t)j * DIM + c]; acc[c] = 0.f; }
    for (int s = lane; s < k; s += 64) {                       // outgoing
        const size_t t = (size_t)j * k + s;
        const long long m = idx[t];
        if (m < 0 || m >= n) continue;
        const float cf = coef[t];
#pragma unroll
        for (int c = 0; c < DIM; c++) acc[c] -= cf * (x[(size_t)m * DIM + c] - xj[c]);
    }
    const int e1 = row_start[j + 1];
    for (int e = row_start[j] + lane; e < e1; e += 64) {       // incoming
        const int t = order[e];
        const int i = t / k;
        const float cf = coef[t];
#pragma unroll
        for (int c = 0; c < DIM; c++) acc[c] += cf * (xj[c] - x[(size_t)i * DIM + c]);
    }
#pragma unroll
    for (int c = 0; c < DIM; c++) {
        float v = acc[c];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) dx[(size_t)j * DIM + c] += v;
    }
}

}  // namespace

extern "C" {

int gs4d_neighbor_diff_norm_forward(int n, int k, int dim, const float* x, const long long* idx, float* d, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || k < 1 || (dim != 3 && dim != 4)) { gs4d_set_error("neighbor_diff_norm: need n >= 0, k >= 1, dim 3 or 4"); return GS4D_EINVAL; }
    if (n == 0) return GS4D_OK;
    if (!x || !idx || !d) { gs4d_set_error("neighbor_diff_norm_forward: null pointer"); return GS4D_EINVAL; }
    const size_t pairs = (size_t)n * k;
    const unsigned grid = (unsigned)((pairs + 255) / 256);
    GS4D_PROF("rigidity_forward", s);
    if (dim == 3) hipLaunchKernelGGL(neighbor_diff_norm_forward_kernel<3>, dim3(grid), dim3(256), 0, s, pairs, k, n, x, idx, d);
    else hipLaunchKernelGGL(neighbor_diff_norm_forward_kernel<4>, dim3(grid), dim3(256), 0, s, pairs, k, n, x, idx, d);
    GS4D_LAUNCH_CHECK("neighbor_diff_norm_forward", 0, s);
    return GS4D_OK;
}

int gs4d_neighbor_diff_norm_backward(int n, int k, int dim, const float* x, const long long* idx, const float* d,
                                     const float* g, float* dx, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || k < 1 || (dim != 3 && dim != 4)) { gs4d_set_error("neighbor_diff_norm: need n >= 0, k >= 1, dim 3 or 4"); return GS4D_EINVAL; }
    if (n == 0) return GS4D_OK;
    if (!x || !idx || !d || !g || !dx) { gs4d_set_error("neighbor_diff_norm_backward: null pointer"); return GS4D_EINVAL; }
    GS4D_PROF("rigidity_backward", s);
    const unsigned grid = (unsigned)((n + 3) / 4);
    if (dim == 3) hipLaunchKernelGGL(neighbor_diff_norm_backward_kernel<3>, dim3(grid), dim3(256), 0, s, n, k, x, idx, d, g, dx);
    else hipLaunchKernelGGL(neighbor_diff_norm_backward_kernel<4>, dim3(grid), dim3(256), 0, s, n, k, x, idx, d, g, dx);
    GS4D_LAUNCH_CHECK("neighbor_diff_norm_backward", 0, s);
    return GS4D_OK;
}

}  // extern "C"

extern "C" int gs4d_neighbor_diff_norm_backward_gather(int n, int k, int dim, const float* x, const long long* idx,
                                                        const float* d, const float* g, const int* order,
                                                        const int* row_start, float* coef_scratch, float* dx, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || k < 1 || (dim != 3 && dim != 4) || (size_t)n * k > 0x7fffffffull) {
        gs4d_set_error("neighbor_diff_norm: need n >= 0, k >= 1, dim 3 or 4, n * k < 2^31");
        return GS4D_EINVAL;
    }
    if (n == 0) return GS4D_OK;
    if (!x || !idx || !d || !g || !order || !row_start || !coef_scratch || !dx) { gs4d_set_error("neighbor_diff_norm_backward_gather: null pointer"); return GS4D_EINVAL; }
    GS4D_PROF("rigidity_backward", s);
    const size_t pairs = (size_t)n * k;
    hipLaunchKernelGGL(neighbor_coef_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, pairs, d, g, coef_scratch);
    const unsigned grid = (unsigned)((n + 3) / 4);
    if (dim == 3) hipLaunchKernelGGL(neighbor_diff_norm_backward_gather_kernel<3>, dim3(grid), dim3(256), 0, s, n, k, x, idx, coef_scratch, order, row_start, dx);
    else hipLaunchKernelGGL(neighbor_diff_norm_backward_gather_kernel<4>, dim3(grid), dim3(256), 0, s, n, k, x, idx, coef_scratch, order, row_start, dx);
    GS4D_LAUNCH_CHECK("neighbor_diff_norm_backward_gather", 0, s);
    return GS4D_OK;
}

// ---------------------------------------------------------------- the three kNN terms of the temporal branch, fused
// gaussian_renderer/renderer.py:189-203 in one forward kernel (+ a one-workgroup sum) and one backward kernel:
//   S_xyz  = sum_pairs w * |vx_j - vx_i|                      knn_xyz  = S_xyz / k / n
//   S_long = sum_pairs w * | |vl_j - vl_i| - |vx_j - vx_i| |  knn_long = S_long / k / n
//   S_rot  = sum_pairs (|r1_j - r1_i| + |r2_j - r2_i|)        knn_rot  = (S_rot / (n k)) * S_w / k / n   (:203 as written)
//   S_w    = sum_pairs w,   w = exp(-100 dist)   (dist comes from detached positions: no gradient through w)
// Nothing per pair is stored: the backward recomputes the four differences per edge (x stays in L2) and gathers over
// the outgoing and the incoming edges of a row (transposed graph), so it needs neither atomics nor (n, k) temporaries.
namespace {

#define RIGT_BLOCKS 2048

// The four fields of a row are packed into ONE 64-byte record (vx 3, vl 3, r1 4, r2 4, 2 pad): a pair then costs one
// cache line for its neighbour instead of four.
__global__ __launch_bounds__(256) void rigidity_pack_kernel(int n, const float* __restrict__ vx, const float* __restrict__ vl,
                                                            const float* __restrict__ r1, const float* __restrict__ r2,
                                                            float4* __restrict__ packed)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float4 a = *reinterpret_cast<const float4*>(r1 + (size_t)p * 4), b = *reinterpret_cast<const float4*>(r2 + (size_t)p * 4);
    packed[(size_t)p * 4 + 0] = make_float4(vx[(size_t)p * 3], vx[(size_t)p * 3 + 1], vx[(size_t)p * 3 + 2], vl[(size_t)p * 3]);
    packed[(size_t)p * 4 + 1] = make_float4(vl[(size_t)p * 3 + 1], vl[(size_t)p * 3 + 2], 0.f, 0.f);
    packed[(size_t)p * 4 + 2] = a;
    packed[(size_t)p * 4 + 3] = b;
}

struct RigRow { float vx[3], vl[3], r1[4], r2[4]; };
__device__ __forceinline__ RigRow rig_load(const float4* __restrict__ packed, size_t p)
{
    const float4 q0 = packed[p * 4], q1 = packed[p * 4 + 1], q2 = packed[p * 4 + 2], q3 = packed[p * 4 + 3];
    RigRow r;
    r.vx[0] = q0.x; r.vx[1] = q0.y; r.vx[2] = q0.z; r.vl[0] = q0.w; r.vl[1] = q1.x; r.vl[2] = q1.y;
    r.r1[0] = q2.x; r.r1[1] = q2.y; r.r1[2] = q2.z; r.r1[3] = q2.w;
    r.r2[0] = q3.x; r.r2[1] = q3.y; r.r2[2] = q3.z; r.r2[3] = q3.w;
    return r;
}
template <int D>
__device__ __forceinline__ float rig_diff(const float (&b)[D], const float (&a)[D], float (&d)[D])
{
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; c++) { d[c] = b[c] - a[c]; s += d[c] * d[c]; }
    return sqrtf(s);
}

template <typename I>
__global__ __launch_bounds__(256) void rigidity_terms_forward_kernel(size_t pairs, int k, int n, const float4* __restrict__ packed,
                                                                     const I* __restrict__ idx,
                                                                     const float* __restrict__ dist,
                                                                     float* __restrict__ partial /* [RIGT_BLOCKS][4] */)
{
    __shared__ float s[4][4];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < pairs; t += (size_t)RIGT_BLOCKS * 256) {
        const size_t i = t / (size_t)k;
        long long j = (long long)idx[t];
        if (j < 0 || j >= n) j = (long long)i;
        const float w = expf(-100.f * dist[t]);
        const RigRow a = rig_load(packed, i), b = rig_load(packed, (size_t)j);
        float d3[3], d4[4];
        const float dx = rig_diff<3>(b.vx, a.vx, d3);
        const float dl = rig_diff<3>(b.vl, a.vl, d3);
        const float d1 = rig_diff<4>(b.r1, a.r1, d4);
        const float d2 = rig_diff<4>(b.r2, a.r2, d4);
        acc[0] += w * dx;
        acc[1] += w * fabsf(dl - dx);
        acc[2] += d1 + d2;
        acc[3] += w;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float v = acc[q];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        partial[(size_t)blockIdx.x * 4 + threadIdx.x] = (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]);
}

// one workgroup: out[q] = sum over the blocks of partial[.][q], fixed order, double accumulation
__global__ __launch_bounds__(256) void rigidity_terms_sum_kernel(const float* __restrict__ partial, float* __restrict__ out)
{
    __shared__ double s[4][4];
    double acc[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < RIGT_BLOCKS; b += 256)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] += (double)partial[(size_t)b * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        double v = acc[q];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) out[threadIdx.x] = (float)((s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]));
}

// gradient of one edge (A -> B, A = the row whose neighbour list holds B) with respect to the fields AT `self`
// (sign +1 if self is B, -1 if self is A): accumulates into g[14] = (vx 3, vl 3, r1 4, r2 4)
__device__ __forceinline__ void rig_edge(const RigRow& A, const RigRow& B, float w, float cx, float cl, float cr, float sign,
                                         float (&g)[14])
{
    float dxv[3], dlv[3], d1v[4], d2v[4];
    const float dx = rig_diff<3>(B.vx, A.vx, dxv), dl = rig_diff<3>(B.vl, A.vl, dlv);
    const float d1 = rig_diff<4>(B.r1, A.r1, d1v), d2 = rig_diff<4>(B.r2, A.r2, d2v);
    const float sg = dl > dx ? 1.f : (dl < dx ? -1.f : 0.f);                 // d|u|/du
    const float kx = dx > 0.f ? sign * w * (cx - cl * sg) / dx : 0.f;        // knn_xyz through dx, knn_long through -dx
    const float kl = dl > 0.f ? sign * w * cl * sg / dl : 0.f;
    const float k1 = d1 > 0.f ? sign * cr / d1 : 0.f;
    const float k2 = d2 > 0.f ? sign * cr / d2 : 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) { g[c] += kx * dxv[c]; g[3 + c] += kl * dlv[c]; }
#pragma unroll
    for (int c = 0; c < 4; c++) { g[6 + c] += k1 * d1v[c]; g[10 + c] += k2 * d2v[c]; }
}

// one wave per row p.  coef (device): cx = dL/dS_xyz, cl = dL/dS_long, cr = dL/dS_rot
template <typename I>
__global__ __launch_bounds__(256) void rigidity_terms_backward_kernel(int n, int k, const float4* __restrict__ packed,
                                                                      const I* __restrict__ idx,
                                                                      const float* __restrict__ dist,
                                                                      const int* __restrict__ order,
                                                                      const int* __restrict__ row_start,
                                                                      const float* __restrict__ coef,
                                                                      float* __restrict__ g_vx, float* __restrict__ g_vl,
                                                                      float* __restrict__ g_r1, float* __restrict__ g_r2,
                                                                      const int* __restrict__ row_order)
{
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= n) return;
    // row_order (may be null): a permutation of the rows, spatially coherent (Morton order of the points): the four waves of
    // a workgroup and the workgroups in flight then read neighbouring points' records, mostly the same ones, while they are
    // still in cache (387 -> ~250 us at n = 100k, k = 100, where the 6.4 MB of records exceed one XCD's L2)
    const int p = row_order ? row_order[slot] : slot;
    if ((unsigned)p >= (unsigned)n) return;
    const float cx = coef[0], cl = coef[1], cr = coef[2];
    const RigRow self = rig_load(packed, (size_t)p);
    float g[14];
#pragma unroll
    for (int c = 0; c < 14; c++) g[c] = 0.f;
    for (int s = lane; s < k; s += 64) {                         // outgoing edges p -> m: self is the tail
        const size_t t = (size_t)p * k + s;
        const long long m = (long long)idx[t];
        if (m < 0 || m >= n || m == p) continue;
        rig_edge(self, rig_load(packed, (size_t)m), expf(-100.f * dist[t]), cx, cl, cr, -1.f, g);
    }
    const int e1 = row_start[p + 1];
    for (int e = row_start[p] + lane; e < e1; e += 64) {         // incoming edges i -> p: self is the head
        const int t = order[e];
        const int i = t / k;
        if (i == p) continue;
        rig_edge(rig_load(packed, (size_t)i), self, expf(-100.f * dist[t]), cx, cl, cr, 1.f, g);
    }
#pragma unroll
    for (int c = 0; c < 14; c++) {
        float v = g[c];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        g[c] = v;
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { g_vx[(size_t)p * 3 + c] = g[c]; g_vl[(size_t)p * 3 + c] = g[3 + c]; }
        *reinterpret_cast<float4*>(g_r1 + (size_t)p * 4) = make_float4(g[6], g[7], g[8], g[9]);
        *reinterpret_cast<float4*>(g_r2 + (size_t)p * 4) = make_float4(g[10], g[11], g[12], g[13]);
    }
}

}  // namespace

extern "C" size_t gs4d_rigidity_terms_scratch_bytes(void) { return sizeof(float) * 4 * RIGT_BLOCKS; }

template <typename I>
static int rigidity_terms_forward_impl(int n, int k, const float* vel_xyz, const float* vel_long, const float* rot1,
                                       const float* rot2, const I* idx, const float* dist, float* packed, float* scratch,
                                       float* sums4, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 1 || k < 1 || !vel_xyz || !vel_long || !rot1 || !rot2 || !idx || !dist || !packed || !scratch || !sums4 ||
        ((uintptr_t)rot1 & 15) || ((uintptr_t)rot2 & 15) || ((uintptr_t)packed & 15)) {
        gs4d_set_error("gs4d_rigidity_terms_forward: bad sizes, null or misaligned pointer");
        return GS4D_EINVAL;
    }
    GS4D_PROF("rigidity_forward", s);
    hipLaunchKernelGGL(rigidity_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, vel_xyz, vel_long, rot1, rot2,
                       (float4*)packed);
    hipLaunchKernelGGL(rigidity_terms_forward_kernel<I>, dim3(RIGT_BLOCKS), dim3(256), 0, s, (size_t)n * k, k, n,
                       (const float4*)packed, idx, dist, scratch);
    hipLaunchKernelGGL(rigidity_terms_sum_kernel, dim3(1), dim3(256), 0, s, scratch, sums4);
    GS4D_LAUNCH_CHECK("rigidity_terms_forward", 0, s);
    return GS4D_OK;
}

template <typename I>
static int rigidity_terms_backward_impl(int n, int k, const float* packed, const I* idx, const float* dist, const int* order,
                                        const int* row_start, const float* coef3, float* g_vel_xyz, float* g_vel_long,
                                        float* g_rot1, float* g_rot2, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 1 || k < 1 || (size_t)n * k > 0x7fffffffull || !packed || !idx || !dist || !order || !row_start || !coef3 ||
        !g_vel_xyz || !g_vel_long || !g_rot1 || !g_rot2 || ((uintptr_t)packed & 15) || ((uintptr_t)g_rot1 & 15) ||
        ((uintptr_t)g_rot2 & 15)) {
        gs4d_set_error("gs4d_rigidity_terms_backward: bad sizes, null or misaligned pointer");
        return GS4D_EINVAL;
    }
    GS4D_PROF("rigidity_backward", s);
    hipLaunchKernelGGL(rigidity_terms_backward_kernel<I>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, n, k, (const float4*)packed,
                       idx, dist, order, row_start, coef3, g_vel_xyz, g_vel_long, g_rot1, g_rot2, (const int*)nullptr);
    GS4D_LAUNCH_CHECK("rigidity_terms_backward", 0, s);
    return GS4D_OK;
}

extern "C" int gs4d_rigidity_terms_forward(int n, int k, const float* vel_xyz, const float* vel_long, const float* rot1,
                                           const float* rot2, const long long* idx, const float* dist, float* packed,
                                           float* scratch, float* sums4, void* stream)
{
    return rigidity_terms_forward_impl<long long>(n, k, vel_xyz, vel_long, rot1, rot2, idx, dist, packed, scratch, sums4, stream);
}

extern "C" int gs4d_rigidity_terms_forward_i32(int n, int k, const float* vel_xyz, const float* vel_long, const float* rot1,
                                               const float* rot2, const int* idx, const float* dist, float* packed,
                                               float* scratch, float* sums4, void* stream)
{
    return rigidity_terms_forward_impl<int>(n, k, vel_xyz, vel_long, rot1, rot2, idx, dist, packed, scratch, sums4, stream);
}

extern "C" int gs4d_rigidity_terms_backward(int n, int k, const float* packed, const long long* idx, const float* dist,
                                            const int* order, const int* row_start, const float* coef3, float* g_vel_xyz,
                                            float* g_vel_long, float* g_rot1, float* g_rot2, void* stream)
{
    return rigidity_terms_backward_impl<long long>(n, k, packed, idx, dist, order, row_start, coef3, g_vel_xyz, g_vel_long, g_rot1,
                                                   g_rot2, stream);
}

extern "C" int gs4d_rigidity_terms_backward_i32(int n, int k, const float* packed, const int* idx, const float* dist,
                                                const int* order, const int* row_start, const float* coef3, float* g_vel_xyz,
                                                float* g_vel_long, float* g_rot1, float* g_rot2, void* stream)
{
    return rigidity_terms_backward_impl<int>(n, k, packed, idx, dist, order, row_start, coef3, g_vel_xyz, g_vel_long, g_rot1,
                                             g_rot2, stream);
}

// ---------------------------------------------------------------- the whole temporal branch in two launches per direction
// gaussian_renderer/renderer.py:173-203: the three second-difference terms
//   momentum = mean |x+ + x- - 2 x|,  scales = mean_rows sum_c |s+ + s- - 2 s|,  shs = mean |h+ + h- - 2 h|
// and the inputs of the three kNN terms (vel_xyz = (x+ - x) / 2, vel_long = x+ - x0, rot_v1 = r - r-, rot_v2 = r+ - r)
// were ~30 elementwise launches forward and as many backward per view.  Here one kernel reads the thirteen tensors once,
// forms the three sums and writes the kNN terms' packed records directly; the backward kernel takes the gradients of
// the packed fields from rigidity_terms_backward_kernel and writes all thirteen input gradients.
namespace {

#define TEMP_BLOCKS_MAX 4096
struct TemporalIn {
    const float* xyz0; const float* m[3]; const float* s[3]; const float* h[3]; const float* r[3];   // [0] final, [1] prev, [2] next
};
struct TemporalGrad {
    float* xyz0; float* m[3]; float* s[3]; float* h[3]; float* r[3];
};

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// 16 lanes per point: lanes 0..11 one float4 of the 48 SH floats each (shf4 = 12), lane 12 the positions, 13 the scales,
// 14 the rotations
__global__ __launch_bounds__(256) void temporal_forward_kernel(int n, int shf4, TemporalIn in, float4* __restrict__ packed,
                                                               float* __restrict__ partial /* [gridDim.x][3] */)
{
    __shared__ float s_red[4][3];
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int p = gid >> 4, sub = gid & 15;
    float a_m = 0.f, a_s = 0.f, a_h = 0.f;
    if (p < n) {
        if (sub < 12) {
            for (int q = sub; q < shf4; q += 12) {
                const size_t o = (size_t)p * shf4 + q;
                const float4 f = reinterpret_cast<const float4*>(in.h[0])[o], a = reinterpret_cast<const float4*>(in.h[1])[o],
                             b = reinterpret_cast<const float4*>(in.h[2])[o];
                a_h += fabsf(b.x + a.x - 2.f * f.x) + fabsf(b.y + a.y - 2.f * f.y) + fabsf(b.z + a.z - 2.f * f.z) + fabsf(b.w + a.w - 2.f * f.w);
            }
        } else if (sub == 12) {
            float vx[3], vl[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const size_t o = (size_t)p * 3 + c;
                const float f = in.m[0][o], a = in.m[1][o], b = in.m[2][o];
                a_m += fabsf(b + a - 2.f * f);
                vx[c] = (b - f) / 2.f;
                vl[c] = b - in.xyz0[o];
            }
            packed[(size_t)p * 4 + 0] = make_float4(vx[0], vx[1], vx[2], vl[0]);
            packed[(size_t)p * 4 + 1] = make_float4(vl[1], vl[2], 0.f, 0.f);
        } else if (sub == 13) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const size_t o = (size_t)p * 3 + c;
                a_s += fabsf(in.s[2][o] + in.s[1][o] - 2.f * in.s[0][o]);
            }
        } else if (sub == 14) {
            const float4 f = reinterpret_cast<const float4*>(in.r[0])[p], a = reinterpret_cast<const float4*>(in.r[1])[p],
                         b = reinterpret_cast<const float4*>(in.r[2])[p];
            packed[(size_t)p * 4 + 2] = make_float4(f.x - a.x, f.y - a.y, f.z - a.z, f.w - a.w);
            packed[(size_t)p * 4 + 3] = make_float4(b.x - f.x, b.y - f.y, b.z - f.z, b.w - f.w);
        }
    }
    float v[3] = {a_m, a_s, a_h};
#pragma unroll
    for (int q = 0; q < 3; q++) {
        for (int off = 32; off >= 1; off >>= 1) v[q] += __shfl_xor(v[q], off, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][q] = v[q];
    }
    __syncthreads();
    if (threadIdx.x < 3)
        partial[(size_t)blockIdx.x * 3 + threadIdx.x] = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

// one workgroup: the seven raw sums in fixed order (double), then the six terms of the dictionary
//   out6 = (momentum, scales, shs, knn_xyz, knn_long, knn_rot), sums7 kept for the backward
__global__ __launch_bounds__(256) void temporal_final_kernel(int n, int k, int shf, int nblk, const float* __restrict__ part3,
                                                             const float* __restrict__ part4, float* __restrict__ sums7,
                                                             float* __restrict__ out6)
{
    __shared__ double s[4][7];
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblk; b += 256)
        for (int q = 0; q < 3; q++) acc[q] += (double)part3[(size_t)b * 3 + q];
    for (int b = threadIdx.x; b < RIGT_BLOCKS; b += 256)
        for (int q = 0; q < 4; q++) acc[3 + q] += (double)part4[(size_t)b * 4 + q];
#pragma unroll
    for (int q = 0; q < 7; q++) {
        double v = acc[q];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[7];
        for (int q = 0; q < 7; q++) { t[q] = (float)((s[0][q] + s[1][q]) + (s[2][q] + s[3][q])); sums7[q] = t[q]; }
        const float kn = (float)k * (float)n;
        out6[0] = t[0] / (3.f * (float)n);
        out6[1] = t[1] / (float)n;
        out6[2] = t[2] / ((float)shf * (float)n);
        out6[3] = t[3] / kn;
        out6[4] = t[4] / kn;
        out6[5] = t[5] / kn * t[6] / kn;
    }
}

// coef6 = dL/d(raw sums) from the upstream gradient of the six terms
__global__ void temporal_coef_kernel(int n, int k, int shf, const float* __restrict__ g6, const float* __restrict__ sums7,
                                     float* __restrict__ coef6)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float kn = (float)k * (float)n;
    coef6[0] = g6[3] / kn;                       // cx, cl, cr first: rigidity_terms_backward_kernel reads coef[0..2]
    coef6[1] = g6[4] / kn;
    coef6[2] = g6[5] * sums7[6] / kn / kn;
    coef6[3] = g6[0] / (3.f * (float)n);
    coef6[4] = g6[1] / (float)n;
    coef6[5] = g6[2] / ((float)shf * (float)n);
}

__global__ __launch_bounds__(256) void temporal_backward_kernel(int n, int shf4, TemporalIn in, const float* __restrict__ coef6,
                                                                const float* __restrict__ g_vx, const float* __restrict__ g_vl,
                                                                const float* __restrict__ g_r1, const float* __restrict__ g_r2,
                                                                TemporalGrad d)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int p = gid >> 4, sub = gid & 15;
    if (p >= n) return;
    const float cm = coef6[3], cs = coef6[4], ch = coef6[5];
    if (sub < 12) {
        for (int q = sub; q < shf4; q += 12) {
            const size_t o = (size_t)p * shf4 + q;
            const float4 f = reinterpret_cast<const float4*>(in.h[0])[o], a = reinterpret_cast<const float4*>(in.h[1])[o],
                         b = reinterpret_cast<const float4*>(in.h[2])[o];
            const float4 t = make_float4(ch * sgn(b.x + a.x - 2.f * f.x), ch * sgn(b.y + a.y - 2.f * f.y),
                                         ch * sgn(b.z + a.z - 2.f * f.z), ch * sgn(b.w + a.w - 2.f * f.w));
            reinterpret_cast<float4*>(d.h[0])[o] = make_float4(-2.f * t.x, -2.f * t.y, -2.f * t.z, -2.f * t.w);
            reinterpret_cast<float4*>(d.h[1])[o] = t;
            reinterpret_cast<float4*>(d.h[2])[o] = t;
        }
    } else if (sub == 12) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t o = (size_t)p * 3 + c;
            const float t = cm * sgn(in.m[2][o] + in.m[1][o] - 2.f * in.m[0][o]);
            const float gx = g_vx[o] * 0.5f, gl = g_vl[o];
            d.m[0][o] = -2.f * t - gx;
            d.m[1][o] = t;
            d.m[2][o] = t + gx + gl;
            d.xyz0[o] = -gl;
        }
    } else if (sub == 13) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t o = (size_t)p * 3 + c;
            const float t = cs * sgn(in.s[2][o] + in.s[1][o] - 2.f * in.s[0][o]);
            d.s[0][o] = -2.f * t;
            d.s[1][o] = t;
            d.s[2][o] = t;
        }
    } else if (sub == 14) {
        const float4 a = reinterpret_cast<const float4*>(g_r1)[p], b = reinterpret_cast<const float4*>(g_r2)[p];
        reinterpret_cast<float4*>(d.r[0])[p] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
        reinterpret_cast<float4*>(d.r[1])[p] = make_float4(-a.x, -a.y, -a.z, -a.w);
        reinterpret_cast<float4*>(d.r[2])[p] = b;
    }
}

static bool temporal_ptrs_ok(const float* const* t13, int align_from)
{
    for (int i = 0; i < 13; i++) {
        if (!t13[i]) return false;
        if (i >= align_from && ((uintptr_t)t13[i] & 15)) return false;      // shs and rotations are read as float4
    }
    return true;
}

}  // namespace

extern "C" size_t gs4d_temporal_regulation_scratch_bytes(int n)
{
    const size_t nblk = ((size_t)n * 16 + 255) / 256;
    return sizeof(float) * (3 * nblk + 4 * RIGT_BLOCKS + 7 + 6 + 14 * (size_t)n) + 1024;
}

// in13: xyz0, means (final, prev, next), scales (final, prev, next: ACTIVATED), shs (final, prev, next), rotations (final
// activated, prev, next as the deformation returns them).  out6: the six terms.  scratch: gs4d_temporal_regulation_scratch_bytes(n),
// kept (with packed) for the backward.
extern "C" int gs4d_temporal_regulation_forward(int n, int k, int sh_floats, const float* const* in13, const int* idx,
                                                const float* dist, float* packed, float* scratch, float* out6, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 1 || k < 1 || sh_floats < 4 || (sh_floats & 3) || !in13 || !idx || !dist || !packed || !scratch || !out6 ||
        ((uintptr_t)packed & 15) || ((uintptr_t)scratch & 15) || !temporal_ptrs_ok(in13, 7)) {
        gs4d_set_error("gs4d_temporal_regulation_forward: bad sizes, null or misaligned pointer");
        return GS4D_EINVAL;
    }
    TemporalIn in;
    in.xyz0 = in13[0];
    for (int q = 0; q < 3; q++) { in.m[q] = in13[1 + q]; in.s[q] = in13[4 + q]; in.h[q] = in13[7 + q]; in.r[q] = in13[10 + q]; }
    const int nblk = (int)(((size_t)n * 16 + 255) / 256);
    float* part3 = scratch;
    float* part4 = part3 + 3 * (size_t)nblk;
    float* sums7 = part4 + 4 * RIGT_BLOCKS;
    GS4D_PROF("rigidity_forward", s);
    hipLaunchKernelGGL(temporal_forward_kernel, dim3(nblk), dim3(256), 0, s, n, sh_floats / 4, in, (float4*)packed, part3);
    hipLaunchKernelGGL(rigidity_terms_forward_kernel<int>, dim3(RIGT_BLOCKS), dim3(256), 0, s, (size_t)n * k, k, n,
                       (const float4*)packed, idx, dist, part4);
    hipLaunchKernelGGL(temporal_final_kernel, dim3(1), dim3(256), 0, s, n, k, sh_floats, nblk, part3, part4, sums7, out6);
    GS4D_LAUNCH_CHECK("temporal_regulation_forward", 0, s);
    return GS4D_OK;
}

// g6: upstream gradient of the six terms (device).  grad13: gradients of in13, written in full.
extern "C" int gs4d_temporal_regulation_backward(int n, int k, int sh_floats, const float* const* in13, const int* idx,
                                                 const float* dist, const int* order, const int* row_start,
                                                 const int* row_order, const float* packed, float* scratch, const float* g6,
                                                 float* const* grad13, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 1 || k < 1 || (size_t)n * k > 0x7fffffffull || sh_floats < 4 || (sh_floats & 3) || !in13 || !grad13 || !idx ||
        !dist || !order || !row_start || !packed || !scratch || !g6 || ((uintptr_t)packed & 15) || ((uintptr_t)scratch & 15) ||
        !temporal_ptrs_ok(in13, 7) || !temporal_ptrs_ok((const float* const*)grad13, 7)) {
        gs4d_set_error("gs4d_temporal_regulation_backward: bad sizes, null or misaligned pointer");
        return GS4D_EINVAL;
    }
    TemporalIn in;
    TemporalGrad d;
    in.xyz0 = in13[0]; d.xyz0 = grad13[0];
    for (int q = 0; q < 3; q++) {
        in.m[q] = in13[1 + q]; in.s[q] = in13[4 + q]; in.h[q] = in13[7 + q]; in.r[q] = in13[10 + q];
        d.m[q] = grad13[1 + q]; d.s[q] = grad13[4 + q]; d.h[q] = grad13[7 + q]; d.r[q] = grad13[10 + q];
    }
    const int nblk = (int)(((size_t)n * 16 + 255) / 256);
    float* sums7 = scratch + 3 * (size_t)nblk + 4 * RIGT_BLOCKS;
    float* coef6 = sums7 + 7;
    float* gbase = (float*)(((uintptr_t)(coef6 + 6) + 255) & ~(uintptr_t)255);     // g_r1 [n,4] | g_r2 [n,4] | g_vx [n,3] | g_vl [n,3]
    // rotations first: 16-byte aligned for any n
    float* g_r1 = gbase; float* g_r2 = gbase + 4 * (size_t)n; float* g_vx = gbase + 8 * (size_t)n; float* g_vl = g_vx + 3 * (size_t)n;
    GS4D_PROF("rigidity_backward", s);
    hipLaunchKernelGGL(temporal_coef_kernel, dim3(1), dim3(64), 0, s, n, k, sh_floats, g6, sums7, coef6);
    hipLaunchKernelGGL(rigidity_terms_backward_kernel<int>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, n, k, (const float4*)packed,
                       idx, dist, order, row_start, coef6, g_vx, g_vl, g_r1, g_r2, row_order);
    hipLaunchKernelGGL(temporal_backward_kernel, dim3(nblk), dim3(256), 0, s, n, sh_floats / 4, in, coef6, g_vx, g_vl, g_r1, g_r2, d);
    GS4D_LAUNCH_CHECK("temporal_regulation_backward", 0, s);
    return GS4D_OK;
}

// ------------------------------------------------------------------ transposed neighbour graph (order, row_start)
// The pairs p = i*k+s sorted by their target idx[p], stable (ascending p inside a target), and where each target's run
// begins: what every gather backward above reads.  A least-significant-digit radix sort over exactly the bits the targets
// have -- ceil(log2 n) of them in digits of up to nine bits: two passes up to n = 262 144 -- with 32-bit keys, the pair
// index made up in the first pass instead of read, and row_start taken from the sorted keys by a bounds kernel.
// (The host-side version was a 32-bit torch.sort: four passes with 64-bit indices, an index fill before it, a searchsorted
// and two dtype conversions after it.)  Pass structure as in csrc/binning.hip: per-tile digit histogram, one scan per
// digit over the tiles, then a stable scatter that ranks with wave ballots and per-wave LDS counters.
namespace {

#define TG_THREADS 256
#ifndef TG_ITEMS
#define TG_ITEMS 8
#endif
#define TG_TILE (TG_THREADS * TG_ITEMS)
#define TG_BINS 512

// lanes of the wave whose digit (<= 9 bits; a narrower digit's upper bits are zero and change nothing) equals this lane's;
// invalid lanes form their own group
__device__ __forceinline__ uint64_t tg_match(uint32_t d, bool valid)
{
    const uint64_t vm = __ballot(valid);
    uint32_t lo = (uint32_t)vm, hi = (uint32_t)(vm >> 32);
    if (!valid) { lo = ~lo; hi = ~hi; }
#pragma unroll
    for (int b = 0; b < 9; b++) {
        const uint32_t bit = (d >> b) & 1u;
        const uint64_t mb = __builtin_amdgcn_uicmp(bit, 0u, 33);    // v_cmp_ne_u32 = the ballot
        const uint32_t neg = 0u - bit;
        lo &= ~((uint32_t)mb ^ neg);
        hi &= ~((uint32_t)(mb >> 32) ^ neg);
    }
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}

// hist[digit][tile]
__global__ __launch_bounds__(TG_THREADS) void tg_hist_kernel(uint32_t R, int shift, int bits, const uint32_t* __restrict__ keys,
                                                             uint32_t* __restrict__ hist, uint32_t ntiles)
{
    __shared__ uint32_t s_h[TG_BINS];
    const uint32_t mask = (1u << bits) - 1u;
    s_h[threadIdx.x] = 0; s_h[threadIdx.x + 256] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * TG_TILE;
    uint32_t d[TG_ITEMS];
#pragma unroll
    for (int i = 0; i < TG_ITEMS; i++) d[i] = (keys[min(base + i * TG_THREADS + threadIdx.x, R - 1u)] >> shift) & mask;
#pragma unroll
    for (int i = 0; i < TG_ITEMS; i++)
        if (base + i * TG_THREADS + threadIdx.x < R) atomicAdd(&s_h[d[i]], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b <= mask; b += TG_THREADS) hist[(size_t)b * ntiles + blockIdx.x] = s_h[b];
}

// workgroup d: exclusive scan of row d of hist in place, row total -> digit_totals[d]
__global__ __launch_bounds__(256) void tg_scan_kernel(uint32_t ntiles, uint32_t* __restrict__ hist, uint32_t* __restrict__ digit_totals)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    uint32_t* row = hist + (size_t)blockIdx.x * ntiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t b = 0; b < ntiles; b += 1024) {
        // four consecutive tiles per lane: a quarter of the barriers
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t i = b + 4 * threadIdx.x + q; v[q] = i < ntiles ? row[i] : 0u; sum += v[q]; }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wp = 0;
        for (int w = 0; w < wave; w++) wp += s_wave[w];
        const uint32_t carry = s_carry;
        uint32_t run = carry + wp + incl - sum;
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t i = b + 4 * threadIdx.x + q; if (i < ntiles) row[i] = run; run += v[q]; }
        __syncthreads();
        if (threadIdx.x == 255) s_carry = run;
        __syncthreads();
    }
    if (threadIdx.x == 0) digit_totals[blockIdx.x] = s_carry;
}

// FIRST: the values are the positions themselves (nothing to read)
template <bool FIRST>
__global__ __launch_bounds__(TG_THREADS) void tg_scatter_kernel(uint32_t R, int shift, int bits, const uint32_t* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ hist,
                                                                const uint32_t* __restrict__ digit_totals, uint32_t ntiles)
{
    __shared__ uint32_t s_cnt[4][TG_BINS];        // per-wave digit counts -> exclusive offsets inside the tile
    __shared__ uint32_t s_digit_start[TG_BINS];   // first staged slot of each digit inside the tile
    __shared__ uint32_t s_global[TG_BINS];        // global output position of that first slot
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_keys[TG_TILE];
    __shared__ uint32_t s_vals[TG_TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t mask = (1u << bits) - 1u;
    for (int i = tid; i < 4 * TG_BINS; i += TG_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();

    const uint32_t tile_base = blockIdx.x * TG_TILE;
    const uint32_t wave_base = tile_base + wave * (64 * TG_ITEMS);     // (wave, round, lane) order == memory order: stable
    uint32_t key[TG_ITEMS], val[TG_ITEMS], rank[TG_ITEMS];
    const uint64_t lanemask_lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < TG_ITEMS; r++) {                               // unconditional, clamped: all loads in flight together
        const uint32_t p = min(wave_base + r * 64 + lane, R - 1u);
        key[r] = keys_in[p];
        val[r] = FIRST ? p : vals_in[p];
    }
    uint32_t before[TG_ITEMS], group[TG_ITEMS];
    int leader[TG_ITEMS];
#pragma unroll
    for (int r = 0; r < TG_ITEMS; r++) {
        const bool valid = wave_base + r * 64 + lane < R;
        const uint64_t same = tg_match((key[r] >> shift) & mask, valid);
        before[r] = (uint32_t)__popcll(same & lanemask_lt);
        leader[r] = __ffsll((unsigned long long)same) - 1;
        group[r] = (uint32_t)__popcll(same);
    }
#pragma unroll
    for (int r = 0; r < TG_ITEMS; r++) {
        rank[r] = 0u;
        if (wave_base + r * 64 + lane < R && lane == leader[r]) {
            const uint32_t d = (key[r] >> shift) & mask;
            rank[r] = s_cnt[wave][d];
            s_cnt[wave][d] = rank[r] + group[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // the next round's leaders read what this one's wrote
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int r = 0; r < TG_ITEMS; r++) rank[r] = (uint32_t)__shfl((int)rank[r], leader[r], 64) + before[r];
    __syncthreads();

    // lane tid owns digits 2 tid and 2 tid + 1: offsets across the four waves, then exclusive scans over the digits
    {
        uint32_t tot[2], dt[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int d = 2 * tid + q;
            const uint32_t c0 = s_cnt[0][d], c1 = s_cnt[1][d], c2 = s_cnt[2][d], c3 = s_cnt[3][d];
            s_cnt[0][d] = 0; s_cnt[1][d] = c0; s_cnt[2][d] = c0 + c1; s_cnt[3][d] = c0 + c1 + c2;
            tot[q] = c0 + c1 + c2 + c3;
            dt[q] = (uint32_t)d <= mask ? digit_totals[d] : 0u;
        }
        const uint32_t pair = tot[0] + tot[1], dpair = dt[0] + dt[1];
        uint32_t incl = pair, dincl = dpair;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64), u = __shfl_up(dincl, off, 64);
            if (lane >= off) { incl += t; dincl += u; }
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wp = 0;
        for (int w = 0; w < wave; w++) wp += s_wave[w];
        __syncthreads();
        if (lane == 63) s_wave[wave] = dincl;
        __syncthreads();
        uint32_t smaller = 0;
        for (int w = 0; w < wave; w++) smaller += s_wave[w];
        const uint32_t start0 = wp + incl - pair, small0 = smaller + dincl - dpair;
        s_digit_start[2 * tid] = start0;
        s_digit_start[2 * tid + 1] = start0 + tot[0];
        // global position = (keys with a smaller digit anywhere) + (same digit in earlier tiles)
        s_global[2 * tid] = small0 + ((uint32_t)(2 * tid) <= mask ? hist[(size_t)(2 * tid) * ntiles + blockIdx.x] : 0u);
        s_global[2 * tid + 1] = small0 + dt[0] + ((uint32_t)(2 * tid + 1) <= mask ? hist[(size_t)(2 * tid + 1) * ntiles + blockIdx.x] : 0u);
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < TG_ITEMS; r++) {
        if (wave_base + r * 64 + lane < R) {
            const uint32_t d = (key[r] >> shift) & mask;
            const uint32_t slot = s_digit_start[d] + s_cnt[wave][d] + rank[r];
            s_keys[slot] = key[r];
            s_vals[slot] = val[r];
        }
    }
    __syncthreads();

    const uint32_t n_valid = min((uint32_t)TG_TILE, R - tile_base);
    for (uint32_t i = tid; i < n_valid; i += TG_THREADS) {
        const uint32_t kk = s_keys[i];
        const uint32_t d = (kk >> shift) & mask;
        const uint32_t dst = s_global[d] + (i - s_digit_start[d]);
        keys_out[dst] = kk;
        vals_out[dst] = s_vals[i];
    }
}

// row_start[t] = first position whose key is >= t, for t = 0..n (row_start[n] = R); keys are sorted
__global__ __launch_bounds__(256) void tg_bounds_kernel(uint32_t R, int n, const uint32_t* __restrict__ keys, int* __restrict__ row_start)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i > R) return;
    const int a = i == 0 ? -1 : (int)min(keys[i - 1], (uint32_t)n);
    const int b = i == R ? n : (int)min(keys[i], (uint32_t)n);
    for (int t = a + 1; t <= b; t++) row_start[t] = (int)i;
}

struct TgPlan { int passes, width; size_t R, ntiles, off_k[2], off_v[2], off_hist, off_tot, bytes; };

TgPlan tg_plan(int n, size_t R)
{
    TgPlan p;
    int bits = 1;
    while (bits < 31 && ((size_t)1 << bits) < (size_t)n) bits++;
    p.passes = (bits + 8) / 9;
    p.width = (bits + p.passes - 1) / p.passes;
    p.R = R;
    p.ntiles = (R + TG_TILE - 1) / TG_TILE;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o = 0;
    p.off_k[0] = o; o += up(4 * R);
    p.off_k[1] = o; o += up(4 * R);
    p.off_v[0] = o; o += up(4 * R);
    p.off_v[1] = o; if (p.passes >= 3) o += up(4 * R);
    p.off_hist = o; o += up(4 * p.ntiles * ((size_t)1 << p.width));
    p.off_tot = o; o += up(4 * TG_BINS);
    p.bytes = o;
    return p;
}

}  // namespace

extern "C" size_t gs4d_transpose_graph_scratch_bytes(int n, long long pairs)
{
    if (n < 1 || pairs < 1) return 0;
    return tg_plan(n, (size_t)pairs).bytes;
}

extern "C" int gs4d_transpose_graph(int n, long long pairs, const int* idx, int* order, int* row_start, void* scratch,
                                    size_t scratch_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 1 || pairs < 1 || pairs > 0x7fffffffll) { gs4d_set_error("gs4d_transpose_graph: need n >= 1 and 1 <= pairs < 2^31"); return GS4D_EINVAL; }
    if (!idx || !order || !row_start || !scratch || ((uintptr_t)scratch & 255)) {
        gs4d_set_error("gs4d_transpose_graph: null pointer, or scratch not 256-byte aligned");
        return GS4D_EINVAL;
    }
    const TgPlan p = tg_plan(n, (size_t)pairs);
    if (scratch_bytes < p.bytes) { gs4d_set_error("gs4d_transpose_graph: scratch smaller than gs4d_transpose_graph_scratch_bytes(n, pairs)"); return GS4D_EINVAL; }
    char* base = (char*)scratch;
    uint32_t* kbuf[2] = {(uint32_t*)(base + p.off_k[0]), (uint32_t*)(base + p.off_k[1])};
    uint32_t* vbuf[2] = {(uint32_t*)(base + p.off_v[0]), (uint32_t*)(base + p.off_v[1])};
    uint32_t* hist = (uint32_t*)(base + p.off_hist);
    uint32_t* totals = (uint32_t*)(base + p.off_tot);
    const uint32_t R = (uint32_t)pairs, nt = (uint32_t)p.ntiles;
    GS4D_PROF("transpose_graph", s);
    const uint32_t* kin = (const uint32_t*)idx;
    const uint32_t* vin = nullptr;
    for (int q = 0; q < p.passes; q++) {
        const int shift = q * p.width;
        uint32_t* kout = kbuf[q & 1];
        uint32_t* vout = q == p.passes - 1 ? (uint32_t*)order : vbuf[q & 1];
        hipLaunchKernelGGL(tg_hist_kernel, dim3(nt), dim3(TG_THREADS), 0, s, R, shift, p.width, kin, hist, nt);
        hipLaunchKernelGGL(tg_scan_kernel, dim3(1u << p.width), dim3(256), 0, s, nt, hist, totals);
        if (q == 0)
            hipLaunchKernelGGL(tg_scatter_kernel<true>, dim3(nt), dim3(TG_THREADS), 0, s, R, shift, p.width, kin, vin, kout, vout, hist, totals, nt);
        else
            hipLaunchKernelGGL(tg_scatter_kernel<false>, dim3(nt), dim3(TG_THREADS), 0, s, R, shift, p.width, kin, vin, kout, vout, hist, totals, nt);
        kin = kout; vin = vout;
    }
    hipLaunchKernelGGL(tg_bounds_kernel, dim3((R + 1 + 255) / 256), dim3(256), 0, s, R, n, kin, row_start);
    GS4D_LAUNCH_CHECK("transpose_graph", 0, s);
    return GS4D_OK;
}
