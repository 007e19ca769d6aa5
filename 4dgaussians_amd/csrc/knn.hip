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

#include <mutex>

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
    // one wave: lane l folds the partial boxes l, l + 64, ..., then a shuffle tree (min / max: the order does not matter)
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < n; i += 64)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            mn[a] = fminf(mn[a], partial[i * 6 + a]);
            mx[a] = fmaxf(mx[a], partial[i * 6 + 3 + a]);
        }
#pragma unroll
    for (int a = 0; a < 3; a++)
        for (int off = 32; off >= 1; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
        }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { bbox[a] = mn[a]; bbox[3 + a] = mx[a]; }
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

// ------------------------------------------------------------------------------------------------
// k-nearest-neighbour query (k <= 128) of the fork's rigidity regulariser.
//
// Behaviour matched: pointops2/src/knnquery/knnquery_cuda_kernel.cu:65-108 (one batch segment): for every
// query the nsample smallest squared distances over ALL points, ascending, slots beyond the point count left at
// (1e10, first index).  Which of several points at EXACTLY the k-th distance the reference keeps depends on its
// heap's history (a tied point is dropped when it happens to be the root); here the lower indices are kept.  The reference is a brute force
// O(m * n) scan with a 100-entry per-thread heap in local memory.
//
// gfx950 design: the points are Morton-sorted once (same machinery as distCUDA2) into boxes of 64 with an
// AABB each and super-boxes of 64 boxes.  One wave64 answers one query: candidates are 64-bit keys
// (d2 bits << 32 | index) -- unique, and their integer order IS the reference's (distance, index) order --
// collected in a 256-entry LDS buffer; when the buffer fills, a radix select over ballots finds the k-th
// smallest key, which becomes the acceptance bound tau.  A box is scanned only if its AABB distance (a
// lower bound of every member's distance in the same float arithmetic, see box_dist2) is <= tau's distance,
// one point per lane, appended through a ballot-prefix compaction.  Queries are processed in Morton order so
// neighbouring waves touch the same boxes.  Exact: bit-identical distances, identical indices up to that tie.
#define KQ_BOX 64
// Buffer size and cut-back trigger, measured at 100 k points (k = 100): 256 / 192 -> 1.12 ms; 192 / 128 1.38, 256 / 160 1.43,
// 256 / 128 1.68, 384 / 320 1.36, 512 / 448 1.67 ms: cutting back earlier costs more selects than the tighter bound saves,
// cutting back later scans more boxes with a loose bound.  Same picture with the coarse cut-backs (KQ_COARSE): 256 / 192
// 0.87 ms; 192 / 128 0.94, 256 / 176 0.95, 256 / 160 0.99, 320 / 256 0.99 ms.
#ifndef KQ_CAP
#define KQ_CAP 256
#endif
#define KQ_SLOTS (KQ_CAP / 64)
#ifndef KQ_COARSE
#define KQ_COARSE 15                         // low distance bits a coarse cut-back does not select on
#endif
#ifndef KQ_TRIGGER
#define KQ_TRIGGER (KQ_CAP - KQ_BOX)     // candidates in the buffer beyond which it is cut back to the k best
#endif
#define KQ_MAXK 128
typedef unsigned long long u64;

__global__ __launch_bounds__(KQ_BOX) void kq_gather_box_kernel(int n, const float* __restrict__ pts,
                                                               const uint32_t* __restrict__ order,
                                                               float4* __restrict__ sorted, Box* __restrict__ boxes)
{
    const int i = blockIdx.x * KQ_BOX + threadIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (i < n) {
        const uint32_t id = order[i];
        const float p[3] = {pts[3 * (size_t)id], pts[3 * (size_t)id + 1], pts[3 * (size_t)id + 2]};
        sorted[i] = make_float4(p[0], p[1], p[2], __uint_as_float(id));
#pragma unroll
        for (int a = 0; a < 3; a++) mn[a] = mx[a] = p[a];
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
        for (int off = 32; off >= 1; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
        }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { boxes[blockIdx.x].mn[a] = mn[a]; boxes[blockIdx.x].mx[a] = mx[a]; }
    }
}

__global__ __launch_bounds__(64) void kq_super_box_kernel(int nboxes, const Box* __restrict__ boxes, Box* __restrict__ supers)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (b < nboxes) {
#pragma unroll
        for (int a = 0; a < 3; a++) { mn[a] = boxes[b].mn[a]; mx[a] = boxes[b].mx[a]; }
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
        for (int off = 32; off >= 1; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
        }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { supers[blockIdx.x].mn[a] = mn[a]; supers[blockIdx.x].mx[a] = mx[a]; }
    }
}

// One wave per query, four queries per workgroup (no workgroup barriers: the waves run independently).
__global__ __launch_bounds__(256) void kq_query_kernel(int n, int m, int k, int nboxes, int nsuper, int idbits,
                                                       const float4* __restrict__ sorted, const Box* __restrict__ boxes,
                                                       const Box* __restrict__ supers, const float* __restrict__ queries,
                                                       const uint32_t* __restrict__ qorder,   // Morton order of the queries when they ARE the points, else NULL
                                                       int* __restrict__ idx_out, float* __restrict__ d2_out)
{
    __shared__ u64 s_keys[4][KQ_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qslot = blockIdx.x * 4 + wave;
    if (qslot >= m) return;
    const int q = qorder ? (int)qorder[qslot] : qslot;
    const float qx = queries[3 * (size_t)q], qy = queries[3 * (size_t)q + 1], qz = queries[3 * (size_t)q + 2];
    const float4 qp = make_float4(qx, qy, qz, 0.f);
    u64* keys = s_keys[wave];
    const u64 lt = (1ull << lane) - 1ull;
    int cnt = 0;
    u64 tau = ~0ull;               // accept key < tau
    float tau_d2 = FLT_MAX;        // scan a box only if its AABB distance is <= tau_d2

    // k-th smallest (1-based kk <= cnt) of the cnt keys: radix select over ballots, most significant bit first.
    // Bit 63 (sign of d2) is always 0 and bits [idbits, 32) of the index half are always 0.
    // exact = false: only the upper distance bits are selected and the bound is rounded UP (low bits set), which keeps
    // every key up to that bound -- at least kk, a few more when candidates share the upper bits.  Enough to tighten
    // the acceptance bound during the scan at two thirds of the cost; the last call is exact.
    auto shrink = [&](int kk, bool exact) {
        __builtin_amdgcn_wave_barrier();
        u64 my[KQ_SLOTS]; bool ok[KQ_SLOTS];
#pragma unroll
        for (int s = 0; s < KQ_SLOTS; s++) { const int i = lane + 64 * s; ok[s] = i < cnt; my[s] = ok[s] ? keys[i] : ~0ull; }
        // `alive[s]`: the slot still matches the prefix chosen so far.  Per bit and slot that leaves one bit test, one
        // ballot and one popcount (the kernel is bound by the CU's scalar unit: ballots, popcounts and the uniform
        // bookkeeping are all scalar instructions), and the two key halves are walked as 32-bit words.
        bool alive[KQ_SLOTS];
        uint32_t hi[KQ_SLOTS], lo[KQ_SLOTS];
#pragma unroll
        for (int s = 0; s < KQ_SLOTS; s++) { alive[s] = ok[s]; hi[s] = (uint32_t)(my[s] >> 32); lo[s] = (uint32_t)my[s]; }
        uint32_t phi = 0u, plo = 0u;
        int need = kk;
        const int last = exact ? 0 : KQ_COARSE;
        for (int bit = 30; bit >= last; bit--) {              // distance half (bit 31 = sign of d2 = 0)
            const uint32_t m = 1u << bit;
            bool z[KQ_SLOTS];
            int c = 0;
#pragma unroll
            for (int s = 0; s < KQ_SLOTS; s++) { z[s] = alive[s] && !(hi[s] & m); c += __popcll(__ballot(z[s])); }
            const bool one = need > c;                         // uniform
            if (one) { need -= c; phi |= m; }
#pragma unroll
            for (int s = 0; s < KQ_SLOTS; s++) alive[s] = one ? (alive[s] && !z[s]) : z[s];
        }
        if (!exact) {
            phi |= (1u << KQ_COARSE) - 1u;
            plo = 0xffffffffu;
        } else {
            // the distance half is decided.  Unless several candidates TIE at the k-th distance and not all of them
            // are needed, the index bits need no selection: every key with that distance is kept.
            int ceq = 0;
#pragma unroll
            for (int s = 0; s < KQ_SLOTS; s++) ceq += __popcll(__ballot(alive[s]));
            if (ceq == need) {
                plo = idbits >= 32 ? 0xffffffffu : ((1u << idbits) - 1u);
            } else {
                for (int bit = idbits - 1; bit >= 0; bit--) {   // bits [idbits, 32) of the index half are always 0
                    const uint32_t m = 1u << bit;
                    bool z[KQ_SLOTS];
                    int c = 0;
#pragma unroll
                    for (int s = 0; s < KQ_SLOTS; s++) { z[s] = alive[s] && !(lo[s] & m); c += __popcll(__ballot(z[s])); }
                    const bool one = need > c;
                    if (one) { need -= c; plo |= m; }
#pragma unroll
                    for (int s = 0; s < KQ_SLOTS; s++) alive[s] = one ? (alive[s] && !z[s]) : z[s];
                }
            }
        }
        const u64 prefix = ((u64)phi << 32) | (u64)plo;
        const u64 kth = prefix;
        int base = 0;
#pragma unroll
        for (int s = 0; s < KQ_SLOTS; s++) {
            const bool keep = ok[s] && my[s] <= kth;
            const u64 mask = __ballot(keep);
            if (keep) keys[base + __popcll(mask & lt)] = my[s];
            base += __popcll(mask);
        }
        cnt = base;                // exact: == kk (keys are unique); coarse: >= kk
        tau = kth + 1ull;
        tau_d2 = __uint_as_float((uint32_t)(kth >> 32));
        __builtin_amdgcn_wave_barrier();
    };

    auto scan_box = [&](int b) {
        const int j = b * KQ_BOX + lane;
        const float4 p = sorted[min(j, n - 1)];
        const float d2 = (qx - p.x) * (qx - p.x) + (qy - p.y) * (qy - p.y) + (qz - p.z) * (qz - p.z);
        const u64 key = ((u64)__float_as_uint(d2) << 32) | (u64)__float_as_uint(p.w);
        const bool keep = j < n && d2 < 1e10f && key < tau;
        const u64 mask = __ballot(keep);
        if (keep) keys[cnt + __popcll(mask & lt)] = key;
        cnt += __popcll(mask);
        if (cnt > KQ_TRIGGER && cnt > k) {
            shrink(k, false);
            if (cnt > KQ_TRIGGER) shrink(k, true);          // many candidates inside one coarse bucket (duplicates)
        }
    };

    // seed: the boxes around the query's own place in the Morton order (or around the closest box)
    int b0;
    if (qorder) {
        b0 = qslot / KQ_BOX;
    } else {
        float best = FLT_MAX; int bestb = 0;
        for (int b = lane; b < nboxes; b += 64) {
            const float d = box_dist2(boxes[b], qp);
            if (d < best) { best = d; bestb = b; }
        }
        for (int off = 32; off >= 1; off >>= 1) {
            const float od = __shfl_xor(best, off, 64);
            const int ob = __shfl_xor(bestb, off, 64);
            if (od < best || (od == best && ob < bestb)) { best = od; bestb = ob; }
        }
        b0 = bestb;
    }
    const int nseed = (k + KQ_BOX - 1) / KQ_BOX + 2;
    int s0 = max(0, b0 - nseed / 2), s1 = min(nboxes, s0 + nseed);
    s0 = max(0, s1 - nseed);
    for (int b = s0; b < s1; b++) scan_box(b);
    if (cnt > k) shrink(k, false);

    // every other box whose AABB can still hold a better candidate
    for (int sb0 = 0; sb0 < nsuper; sb0 += 64) {
        const int sb = sb0 + lane;
        u64 shit = __ballot(sb < nsuper && !(box_dist2(supers[min(sb, nsuper - 1)], qp) > tau_d2));
        while (shit) {
            const int sl = __ffsll((long long)shit) - 1;
            shit &= shit - 1ull;
            const int b = (sb0 + sl) * 64 + lane;
            u64 bhit = __ballot(b < nboxes && (b < s0 || b >= s1) && !(box_dist2(boxes[min(b, nboxes - 1)], qp) > tau_d2));
            while (bhit) {
                const int bl = __ffsll((long long)bhit) - 1;
                bhit &= bhit - 1ull;
                const int bb = (sb0 + sl) * 64 + bl;
                // tau may have tightened since the ballot: re-test (uniform)
                if (!(box_dist2(boxes[bb], qp) > tau_d2)) scan_box(bb);
            }
        }
    }
    if (cnt > k) shrink(k, true);
    __builtin_amdgcn_wave_barrier();

    // ascending order by counting ranks (keys are unique); unfilled slots keep (1e10, 0) as in the reference
    u64 mine[2]; int rank[2];
#pragma unroll
    for (int s = 0; s < 2; s++) { const int i = lane + 64 * s; mine[s] = i < cnt ? keys[i] : ~0ull; rank[s] = 0; }
    for (int j = 0; j < cnt; j++) {
        const u64 o = keys[j];
        rank[0] += o < mine[0];
        rank[1] += o < mine[1];
    }
    int* io = idx_out + (size_t)q * k;
    float* dout = d2_out + (size_t)q * k;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int i = lane + 64 * s;
        if (i < cnt) { io[rank[s]] = (int)(uint32_t)mine[s]; dout[rank[s]] = __uint_as_float((uint32_t)(mine[s] >> 32)); }
        else if (i < k) { io[i] = 0; dout[i] = 1e10f; }
    }
}

}  // namespace

// scratch layout of the 3-NN pipeline, shared by the two entry points below
namespace {
struct KnnScratch {
    float* partial; float* bbox; uint64_t* codes; uint32_t* ids; uint64_t* ktmp[2]; uint32_t* vtmp[2]; uint32_t* hist;
    float4* sorted; Box* boxes; int nboxes, nred;
};

size_t knn_carve(char* base, int P, KnnScratch& k)
{
    k.nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    k.nred = min(1024, (P + 255) / 256);
    char* p = base;
    gs4d_obtain(p, k.partial, (size_t)k.nred * 6);
    gs4d_obtain(p, k.bbox, 8);
    gs4d_obtain(p, k.codes, (size_t)P);
    gs4d_obtain(p, k.ids, (size_t)P);
    gs4d_obtain(p, k.ktmp[0], (size_t)P);
    gs4d_obtain(p, k.ktmp[1], (size_t)P);
    gs4d_obtain(p, k.vtmp[0], (size_t)P);
    gs4d_obtain(p, k.vtmp[1], (size_t)P);
    gs4d_obtain(p, k.hist, (size_t)256 * gs4d_sort_blocks((size_t)P) + 256);
    gs4d_obtain(p, k.sorted, (size_t)P);
    gs4d_obtain(p, k.boxes, (size_t)k.nboxes);
    return (size_t)(p - base) + 256;
}

// Every array of the scratch is written in full by a kernel before another kernel reads it (bbox partials, Morton
// codes, the radix-sort histograms of each pass, sorted points, boxes): nothing has to start at zero.
int knn_run(const KnnScratch& k, int P, const float* points, float* mean_dist2, hipStream_t s)
{
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(k.nred), dim3(256), 0, s, P, points, k.partial);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(64), 0, s, k.nred, k.partial, k.bbox);
    hipLaunchKernelGGL(morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, k.bbox, k.codes, k.ids);
    GS4D_LAUNCH_CHECK("knn_morton", 0, s);
    int where = 0;
    int rc = gs4d_radix_sort_pairs((uint32_t)P, 4, 0, k.codes, k.ids, k.ktmp, k.vtmp, k.hist, &where, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(gather_box_kernel, dim3(k.nboxes), dim3(KNN_BOX), 0, s, P, points, k.vtmp[where], k.sorted, k.boxes);
    hipLaunchKernelGGL(mean_dist_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, k.nboxes, k.sorted, k.boxes, mean_dist2);
    GS4D_LAUNCH_CHECK("knn_mean_dist", 0, s);
    return GS4D_OK;
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
    KnnScratch k;
    const size_t bytes = knn_carve(nullptr, P, k);
    gs4d_note_alloc();
    char* base = (char*)alloc(alloc_ctx, bytes);
    if (!base) { gs4d_set_error("knn scratch allocator returned NULL"); return GS4D_EALLOC; }
    if ((uintptr_t)base & 255) { gs4d_set_error("knn scratch must be 256-byte aligned"); return GS4D_EINVAL; }
    knn_carve(base, P, k);
    return knn_run(k, P, points, mean_dist2, s);
}

// ---- the same computation with NO allocation anywhere (include/gs4d.h: gs4d_knn_mean_dist2_static) ----
// The pipeline's scratch (Morton keys, sort ping-pong buffers and histograms, sorted points, boxes) lives in a
// file-scope device array of fixed size.  In host code the array's NAME is only a stand-in: the device address is
// asked for once per device with hipGetSymbolAddress (each GPU holds its own copy at its own address) and only that
// pointer is handed to kernels.
#define GS4D_KNN_STATIC_MAX_POINTS 262144
#define GS4D_KNN_STATIC_BYTES (16u << 20)
#define GS4D_KNN_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[GS4D_KNN_STATIC_BYTES];
namespace {
std::mutex g_static_mu;
char* g_static_ptr[GS4D_KNN_MAX_DEVICES] = {};
}  // namespace

extern "C" int gs4d_knn_static_max_points(void) { return GS4D_KNN_STATIC_MAX_POINTS; }

extern "C" int gs4d_knn_mean_dist2_static(int P, const float* points, float* mean_dist2, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) return GS4D_OK;
    if (P < 0 || !points || !mean_dist2) {
        gs4d_set_error("gs4d_knn_mean_dist2_static: bad arguments");
        return GS4D_EINVAL;
    }
    if (P > GS4D_KNN_STATIC_MAX_POINTS) {
        gs4d_set_error("gs4d_knn_mean_dist2_static: %d points exceed the static scratch's maximum of %d "
                       "(use gs4d_knn_mean_dist2 with an allocator)", P, GS4D_KNN_STATIC_MAX_POINTS);
        return GS4D_EINVAL;
    }
    int dev = -1;
    GS4D_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= GS4D_KNN_MAX_DEVICES) { gs4d_set_error("gs4d_knn_mean_dist2_static: device %d out of range", dev); return GS4D_EINVAL; }
    char* base = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_static_mu);
        if (!g_static_ptr[dev]) {                       // first call on this device (it is the current one)
            void* q = nullptr;
            const hipError_t e = hipGetSymbolAddress(&q, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !q) {
                gs4d_set_error("gs4d_knn_mean_dist2_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
                return GS4D_EHIP;
            }
            g_static_ptr[dev] = (char*)q;
        }
        base = g_static_ptr[dev];
    }
    KnnScratch k;
    if (knn_carve(base, P, k) > GS4D_KNN_STATIC_BYTES) {      // cannot happen for P <= the stated maximum; checked all the same
        gs4d_set_error("gs4d_knn_mean_dist2_static: scratch layout for %d points exceeds %u bytes", P, GS4D_KNN_STATIC_BYTES);
        return GS4D_EINVAL;
    }
    return knn_run(k, P, points, mean_dist2, s);
}

extern "C" int gs4d_knn_query(gs4d_alloc_fn alloc, void* alloc_ctx, int n, const float* xyz, int m, const float* new_xyz,
                              int nsample, int* idx, float* dist2, void* stream)
{
    return gs4d_knn_query_order(alloc, alloc_ctx, n, xyz, m, new_xyz, nsample, idx, dist2, nullptr, stream);
}

extern "C" int gs4d_knn_query_order(gs4d_alloc_fn alloc, void* alloc_ctx, int n, const float* xyz, int m, const float* new_xyz,
                                    int nsample, int* idx, float* dist2, int* point_order, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (m == 0) return GS4D_OK;
    if (n <= 0 || m < 0 || !xyz || !idx || !dist2 || !alloc || nsample < 1 || nsample > KQ_MAXK) {
        gs4d_set_error("gs4d_knn_query: bad arguments (need n >= 1, 1 <= nsample <= %d)", KQ_MAXK);
        return GS4D_EINVAL;
    }
    GS4D_PROF("knn_query", s);
    const bool same = new_xyz == nullptr || new_xyz == xyz;
    const int nboxes = (n + KQ_BOX - 1) / KQ_BOX, nsuper = (nboxes + 63) / 64;
    const int nred = min(1024, (n + 255) / 256);
    char* p = nullptr;
    float* partial; float* bbox; uint64_t* codes; uint32_t* ids; uint64_t* ktmp[2]; uint32_t* vtmp[2]; uint32_t* hist;
    float4* sorted; Box* boxes; Box* supers;
    auto carve = [&](char* base) {
        p = base;
        gs4d_obtain(p, partial, (size_t)nred * 6);
        gs4d_obtain(p, bbox, 8);
        gs4d_obtain(p, codes, (size_t)n);
        gs4d_obtain(p, ids, (size_t)n);
        gs4d_obtain(p, ktmp[0], (size_t)n);
        gs4d_obtain(p, ktmp[1], (size_t)n);
        gs4d_obtain(p, vtmp[0], (size_t)n);
        gs4d_obtain(p, vtmp[1], (size_t)n);
        gs4d_obtain(p, hist, (size_t)256 * gs4d_sort_blocks((size_t)n) + 256);
        gs4d_obtain(p, sorted, (size_t)n);
        gs4d_obtain(p, boxes, (size_t)nboxes);
        gs4d_obtain(p, supers, (size_t)nsuper);
    };
    carve(nullptr);
    const size_t bytes = (size_t)p + 256;
    gs4d_note_alloc();
    char* base = (char*)alloc(alloc_ctx, bytes);
    if (!base) { gs4d_set_error("knn scratch allocator returned NULL"); return GS4D_EALLOC; }
    if ((uintptr_t)base & 255) { gs4d_set_error("knn scratch must be 256-byte aligned"); return GS4D_EINVAL; }
    carve(base);

    hipLaunchKernelGGL(bbox_partial_kernel, dim3(nred), dim3(256), 0, s, n, xyz, partial);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(64), 0, s, nred, partial, bbox);
    hipLaunchKernelGGL(morton_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, xyz, bbox, codes, ids);
    GS4D_LAUNCH_CHECK("knn_query_morton", 0, s);
    int where = 0;
    int rc = gs4d_radix_sort_pairs((uint32_t)n, 4, 0, codes, ids, ktmp, vtmp, hist, &where, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(kq_gather_box_kernel, dim3(nboxes), dim3(KQ_BOX), 0, s, n, xyz, vtmp[where], sorted, boxes);
    if (point_order) GS4D_HIP_CHECK(hipMemcpyAsync(point_order, vtmp[where], sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(kq_super_box_kernel, dim3(nsuper), dim3(64), 0, s, nboxes, boxes, supers);
    int idbits = 1;
    while (idbits < 32 && ((unsigned)(n - 1) >> idbits)) idbits++;
    hipLaunchKernelGGL(kq_query_kernel, dim3((m + 3) / 4), dim3(256), 0, s, n, m, nsample, nboxes, nsuper, idbits, sorted,
                       boxes, supers, same ? xyz : new_xyz, same ? vtmp[where] : (const uint32_t*)nullptr, idx, dist2);
    GS4D_LAUNCH_CHECK("knn_query", 0, s);
    return GS4D_OK;
}
