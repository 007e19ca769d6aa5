// distCUDA2 for gfx950: mean of the three smallest squared distances from each point to the
// other points (used once, to initialise Gaussian scales).
//
// Behaviour matched: submodules/simple-knn/simple_knn.cu:45-70 (30-bit Morton code on the
// origin-including bbox), :78-117 (AABB per 1024 Morton-sorted points), :119-183 (3-NN with
// conservative box pruning), :185-221 (driver) of the reference.  The result is the exact 3-NN
// answer, so it is bit-identical to a brute-force evaluation of the same float32 expression
// (compiled with -ffp-contract=off).
//
// gfx950 design: no host round trips (the reference copies the bbox to the host twice); points
// are gathered once into Morton order as float4 so that the box scans of neighbouring lanes read
// the same addresses (served as broadcasts from L1/LDS-free), and the sort reuses the stable
// LDS radix sort of the rasterizer.
#include <float.h>

#include "gs4d_internal.h"

#define KNN_BOX 1024

namespace {

struct Box { float mn[3]; float mx[3]; };

// stage 1: per-workgroup partial bbox (init 0,0,0 as simple_knn.cu:191)
__global__ __launch_bounds__(256) void bbox_partial_kernel(int P, const float* __restrict__ pts, float* __restrict__ partial)
{
    __shared__ float s[256][6];
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = pts[3 * (size_t)i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; a++) { s[threadIdx.x][a] = mn[a]; s[threadIdx.x][3 + a] = mx[a]; }
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                s[threadIdx.x][a] = fminf(s[threadIdx.x][a], s[threadIdx.x + off][a]);
                s[threadIdx.x][3 + a] = fmaxf(s[threadIdx.x][3 + a], s[threadIdx.x + off][3 + a]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) partial[blockIdx.x * 6 + threadIdx.x] = s[0][threadIdx.x];
}

__global__ __launch_bounds__(64) void bbox_final_kernel(int n, const float* __restrict__ partial, float* __restrict__ bbox)
{
    if (threadIdx.x < 6) {
        float v = 0.f;
        for (int i = 0; i < n; i++) v = threadIdx.x < 3 ? fminf(v, partial[i * 6 + threadIdx.x]) : fmaxf(v, partial[i * 6 + threadIdx.x]);
        bbox[threadIdx.x] = v;
    }
}

__device__ __forceinline__ uint32_t spread_bits(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

// float -> uint32 conversion saturates on this hardware (negative / NaN -> 0), as on the reference's GPU
__global__ __launch_bounds__(256) void morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ bbox,
                                                     uint64_t* __restrict__ codes, uint32_t* __restrict__ ids)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float mn = bbox[a], mx = bbox[3 + a];
        c[a] = spread_bits((uint32_t)(((pts[3 * (size_t)i + a] - mn) / (mx - mn)) * ((1 << 10) - 1)));
    }
    codes[i] = (uint64_t)(c[0] | (c[1] << 1) | (c[2] << 2));
    ids[i] = (uint32_t)i;
}

// gather into Morton order + AABB of each run of 1024
__global__ __launch_bounds__(KNN_BOX) void gather_box_kernel(int P, const float* __restrict__ pts,
                                                             const uint32_t* __restrict__ order,
                                                             float4* __restrict__ sorted, Box* __restrict__ boxes)
{
    __shared__ float s[KNN_BOX][6];
    const int i = blockIdx.x * KNN_BOX + threadIdx.x;
    float p[3];
    if (i < P) {
        const uint32_t id = order[i];
        p[0] = pts[3 * (size_t)id]; p[1] = pts[3 * (size_t)id + 1]; p[2] = pts[3 * (size_t)id + 2];
        sorted[i] = make_float4(p[0], p[1], p[2], __uint_as_float(id));
#pragma unroll
        for (int a = 0; a < 3; a++) { s[threadIdx.x][a] = p[a]; s[threadIdx.x][3 + a] = p[a]; }
    } else {
#pragma unroll
        for (int a = 0; a < 3; a++) { s[threadIdx.x][a] = FLT_MAX; s[threadIdx.x][3 + a] = -FLT_MAX; }
    }
    __syncthreads();
    for (int off = KNN_BOX / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                s[threadIdx.x][a] = fminf(s[threadIdx.x][a], s[threadIdx.x + off][a]);
                s[threadIdx.x][3 + a] = fmaxf(s[threadIdx.x][3 + a], s[threadIdx.x + off][3 + a]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 3) {
        boxes[blockIdx.x].mn[threadIdx.x] = s[0][threadIdx.x];
        boxes[blockIdx.x].mx[threadIdx.x] = s[0][3 + threadIdx.x];
    }
}

__device__ __forceinline__ void keep3(const float4 ref, const float4 p, float* best)
{
    const float dx = p.x - ref.x, dy = p.y - ref.y, dz = p.z - ref.z;
    float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (best[j] > dist) {
            const float t = best[j];
            best[j] = dist;
            dist = t;
        }
    }
}

__device__ __forceinline__ float box_dist2(const Box& b, const float4 p)
{
    float d[3] = {0.f, 0.f, 0.f};
    const float q[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; a++)
        if (q[a] < b.mn[a] || q[a] > b.mx[a]) d[a] = fminf(fabsf(q[a] - b.mn[a]), fabsf(q[a] - b.mx[a]));
    return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}

__global__ __launch_bounds__(256) void mean_dist_kernel(int P, int nboxes, const float4* __restrict__ sorted,
                                                        const Box* __restrict__ boxes, float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < P;
    const float4 me = live ? sorted[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    if (live)
        for (int j = max(0, i - 3); j <= min(P - 1, i + 3); j++)
            if (j != i) keep3(me, sorted[j], best);
    const float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    for (int b = 0; b < nboxes; b++) {
        const Box box = boxes[b];
        bool want = false;
        if (live) {
            const float d = box_dist2(box, me);
            want = !(d > reject || d > best[2]);
        }
        if (__ballot(want) == 0ull) continue;
        const int lo = b * KNN_BOX, hi = min(P, (b + 1) * KNN_BOX);
        if (want)
            for (int j = lo; j < hi; j++)
                if (j != i) keep3(me, sorted[j], best);
    }
    if (live) out[__float_as_uint(me.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace

extern "C" int gs4d_knn_mean_dist2(gs4d_alloc_fn alloc, void* alloc_ctx, int P, const float* points,
                                   float* mean_dist2, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) return GS4D_OK;
    if (P < 0 || !points || !mean_dist2 || !alloc) {
        gs4d_set_error("gs4d_knn_mean_dist2: bad arguments");
        return GS4D_EINVAL;
    }
    GS4D_PROF("knn", s);
    const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    const int nred = min(1024, (P + 255) / 256);
    // carve scratch
    char* p = nullptr;
    float* partial; float* bbox; uint64_t* codes; uint32_t* ids; uint64_t* ktmp[2]; uint32_t* vtmp[2]; uint32_t* hist;
    float4* sorted; Box* boxes;
    auto carve = [&](char* base) {
        p = base;
        gs4d_obtain(p, partial, (size_t)nred * 6);
        gs4d_obtain(p, bbox, 8);
        gs4d_obtain(p, codes, (size_t)P);
        gs4d_obtain(p, ids, (size_t)P);
        gs4d_obtain(p, ktmp[0], (size_t)P);
        gs4d_obtain(p, ktmp[1], (size_t)P);
        gs4d_obtain(p, vtmp[0], (size_t)P);
        gs4d_obtain(p, vtmp[1], (size_t)P);
        gs4d_obtain(p, hist, (size_t)256 * gs4d_sort_blocks((size_t)P) + 256);
        gs4d_obtain(p, sorted, (size_t)P);
        gs4d_obtain(p, boxes, (size_t)nboxes);
    };
    carve(nullptr);
    const size_t bytes = (size_t)p + 256;
    char* base = (char*)alloc(alloc_ctx, bytes);
    if (!base) { gs4d_set_error("knn scratch allocator returned NULL"); return GS4D_EALLOC; }
    if ((uintptr_t)base & 255) { gs4d_set_error("knn scratch must be 256-byte aligned"); return GS4D_EINVAL; }
    carve(base);

    hipLaunchKernelGGL(bbox_partial_kernel, dim3(nred), dim3(256), 0, s, P, points, partial);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(64), 0, s, nred, partial, bbox);
    hipLaunchKernelGGL(morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, bbox, codes, ids);
    GS4D_LAUNCH_CHECK("knn_morton", 0, s);
    int where = 0;
    int rc = gs4d_radix_sort_pairs((uint32_t)P, 4, codes, ids, ktmp, vtmp, hist, &where, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(gather_box_kernel, dim3(nboxes), dim3(KNN_BOX), 0, s, P, points, vtmp[where], sorted, boxes);
    hipLaunchKernelGGL(mean_dist_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, nboxes, sorted, boxes, mean_dist2);
    GS4D_LAUNCH_CHECK("knn_mean_dist", 0, s);
    return GS4D_OK;
}
