// HexPlane field for gfx950: the fused multi-resolution plane gather of the forward and the plane-gradient pipeline of the
// backward (per-point sample gradients -> counting sort by texel cell -> reduce in cell order).  Launched by deform.hip
// through deform_hexplane.h; built with -fno-slp-vectorize (see that header).
//
// Behaviour matched (reference, PyTorch):
//   scene/hexplane.py:18-19,21-47,100-131,192-215   aabb normalisation (max -> -1), bilinear border sampling with
//                                                   align_corners, product over the six planes, concatenation over levels
//   scene/deformation.py:128-141                    t' = force_time_embed([t, f])
#include <stdlib.h>
#include <string.h>

#include "deform_hexplane.h"

namespace {

// ------------------------------------------------------------------ HexPlane gather
struct PlaneTable {
    const float* p[GS4D_MAX_LEVELS * 6];
    float* g[GS4D_MAX_LEVELS * 6];
    int w[GS4D_MAX_LEVELS * 6];
    int h[GS4D_MAX_LEVELS * 6];
};

__device__ __constant__ const int kPairA[6] = {0, 0, 0, 1, 1, 2};
__device__ __constant__ const int kPairB[6] = {1, 2, 3, 2, 3, 3};

struct Corner {
    int x0, x1, y0, y1;          // texel coordinates of the four corners
    int i00, i01, i10, i11;      // texel indices (y*W + x)
    float w00, w01, w10, w11;
    float wx, wy;                // fractional parts
    float sx, sy;                // d(pixel coord)/d(normalised coord), 0 where the border clamp is active
};

__device__ __forceinline__ Corner corner_of(float u, float v, int Ww, int Hh)
{
    Corner c;
    float x = (u + 1.f) * 0.5f * (float)(Ww - 1);
    float y = (v + 1.f) * 0.5f * (float)(Hh - 1);
    c.sx = (x < 0.f || x > (float)(Ww - 1)) ? 0.f : 0.5f * (float)(Ww - 1);
    c.sy = (y < 0.f || y > (float)(Hh - 1)) ? 0.f : 0.5f * (float)(Hh - 1);
    x = fminf(fmaxf(x, 0.f), (float)(Ww - 1));
    y = fminf(fmaxf(y, 0.f), (float)(Hh - 1));
    const float x0 = floorf(x), y0 = floorf(y);
    c.wx = x - x0;
    c.wy = y - y0;
    const int x0i = (int)x0, y0i = (int)y0;
    const int x1i = min(x0i + 1, Ww - 1), y1i = min(y0i + 1, Hh - 1);
    c.x0 = x0i; c.x1 = x1i; c.y0 = y0i; c.y1 = y1i;
    c.i00 = y0i * Ww + x0i; c.i01 = y0i * Ww + x1i;
    c.i10 = y1i * Ww + x0i; c.i11 = y1i * Ww + x1i;
    c.w00 = (1.f - c.wx) * (1.f - c.wy); c.w01 = c.wx * (1.f - c.wy);
    c.w10 = (1.f - c.wx) * c.wy;         c.w11 = c.wx * c.wy;
    return c;
}

__device__ __forceinline__ void load_coords(const float* __restrict__ xyz, const float* __restrict__ aabb,
                                            const float* __restrict__ time, const float* __restrict__ force,
                                            const float* __restrict__ ft_w, const float* __restrict__ ft_b,
                                            int use_time, int use_force, int p, float pts[4])
{
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = aabb[a], hi = aabb[3 + a];
        pts[a] = (xyz[3 * (size_t)p + a] - lo) * (2.0f / (hi - lo)) - 1.0f;
    }
    const float t = use_time ? time[p] : time[p] * 0.0f;
    const float f = use_force ? force[p] : force[p] * 0.0f;
    pts[3] = t * ft_w[0] + f * ft_w[1] + ft_b[0];
}

// LPP = C/4 lanes per point, each lane owns 4 consecutive channels (float4 corner reads).
__global__ __launch_bounds__(256) void hexplane_forward_kernel(
    int P, int C, int L, int lpp_shift, PlaneTable T, const float* __restrict__ aabb, const float* __restrict__ xyz,
    const float* __restrict__ time, const float* __restrict__ force, const float* __restrict__ ft_w,
    const float* __restrict__ ft_b, int use_time, int use_force, float* __restrict__ tprime,
    float* __restrict__ feat, int ldf)
{
    const int gtid = blockIdx.x * 256 + threadIdx.x;
    const int p = gtid >> lpp_shift;
    const int sub = gtid & ((1 << lpp_shift) - 1);
    if (p >= P) return;
    float pts[4];
    load_coords(xyz, aabb, time, force, ft_w, ft_b, use_time, use_force, p, pts);
    if (sub == 0) tprime[p] = pts[3];
    for (int l = 0; l < L; l++) {
        float4 acc = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int pl = l * 6 + i;
            const Corner c = corner_of(pts[kPairA[i]], pts[kPairB[i]], T.w[pl], T.h[pl]);
            const float4* __restrict__ base = reinterpret_cast<const float4*>(T.p[pl]) + sub;
            const int q = C >> 2;
            const float4 f00 = base[(size_t)c.i00 * q], f01 = base[(size_t)c.i01 * q];
            const float4 f10 = base[(size_t)c.i10 * q], f11 = base[(size_t)c.i11 * q];
            acc.x *= f00.x * c.w00 + f01.x * c.w01 + f10.x * c.w10 + f11.x * c.w11;
            acc.y *= f00.y * c.w00 + f01.y * c.w01 + f10.y * c.w10 + f11.y * c.w11;
            acc.z *= f00.z * c.w00 + f01.z * c.w01 + f10.z * c.w10 + f11.z * c.w11;
            acc.w *= f00.w * c.w00 + f01.w * c.w01 + f10.w * c.w10 + f11.w * c.w11;
        }
        reinterpret_cast<float4*>(feat + (size_t)p * ldf + l * C)[sub] = acc;
    }
}

// The same gather with the corner arithmetic done ONCE per (point, plane): with C/4 lanes per point every lane of the
// kernel above repeats the twelve corner_of evaluations of its point (about 45 VALU instructions each against the ~20
// of the blend that uses them: the counters showed the vector ALU busy 60 % of the time at C = 32 and 75 % at C = 128,
// waves waiting 14 %).  A workgroup of 256 lanes owns 256 / (C/4) consecutive points: their normalised coordinates go
// to LDS, then the (point, plane) pairs are dealt to the lanes and each leaves its four texel indices and four weights in
// LDS, where the point's lanes read them back as two broadcast ds_read_b128.  Same arithmetic by another lane: the
// features are bit-identical.  Used when C >= 16 (four lanes per point or more).
__device__ __forceinline__ int pair_a(int i) { return (0x211000 >> (4 * i)) & 15; }      // kPairA / kPairB without a table read
__device__ __forceinline__ int pair_b(int i) { return (0x332321 >> (4 * i)) & 15; }

__global__ __launch_bounds__(256) void hexplane_forward_shared_kernel(
    int P, int C, int L, int lpp_shift, PlaneTable T, const float* __restrict__ aabb, const float* __restrict__ xyz,
    const float* __restrict__ time, const float* __restrict__ force, const float* __restrict__ ft_w,
    const float* __restrict__ ft_b, int use_time, int use_force, float* __restrict__ tprime,
    float* __restrict__ feat, int ldf)
{
    extern __shared__ float4 s_hex[];                    // [ppw] coordinates | [ppw * L * 6][2] corner sets
    const int ppw = 256 >> lpp_shift;                    // points of this workgroup
    const int np = L * 6;
    float4* s_pts = s_hex;
    float4* s_set = s_hex + ppw;
    const int p0 = blockIdx.x * ppw;
    if ((int)threadIdx.x < ppw) {
        const int p = p0 + (int)threadIdx.x;
        float pts[4] = {0.f, 0.f, 0.f, 0.f};
        if (p < P) {
            load_coords(xyz, aabb, time, force, ft_w, ft_b, use_time, use_force, p, pts);
            tprime[p] = pts[3];
        }
        s_pts[threadIdx.x] = make_float4(pts[0], pts[1], pts[2], pts[3]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ppw * np; e += 256) {
        const int lp = e / np, pl = e - lp * np, i = pl % 6;
        const float* pp = reinterpret_cast<const float*>(&s_pts[lp]);
        const Corner c = corner_of(pp[pair_a(i)], pp[pair_b(i)], T.w[pl], T.h[pl]);
        s_set[2 * e] = make_float4(__int_as_float(c.i00), __int_as_float(c.i01), __int_as_float(c.i10), __int_as_float(c.i11));
        s_set[2 * e + 1] = make_float4(c.w00, c.w01, c.w10, c.w11);
    }
    __syncthreads();
    const int lp = (int)threadIdx.x >> lpp_shift;
    const int sub = (int)threadIdx.x & ((1 << lpp_shift) - 1);
    const int p = p0 + lp;
    if (p >= P) return;
    const int q = C >> 2;
    const float4* __restrict__ mine = s_set + 2 * (size_t)lp * np;
    for (int l = 0; l < L; l++) {
        float4 acc = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int pl = l * 6 + i;
            const float4 ci = mine[2 * pl], cw = mine[2 * pl + 1];
            const float4* __restrict__ base = reinterpret_cast<const float4*>(T.p[pl]) + sub;
            const float4 f00 = base[(size_t)__float_as_int(ci.x) * q], f01 = base[(size_t)__float_as_int(ci.y) * q];
            const float4 f10 = base[(size_t)__float_as_int(ci.z) * q], f11 = base[(size_t)__float_as_int(ci.w) * q];
            acc.x *= f00.x * cw.x + f01.x * cw.y + f10.x * cw.z + f11.x * cw.w;
            acc.y *= f00.y * cw.x + f01.y * cw.y + f10.y * cw.z + f11.y * cw.w;
            acc.z *= f00.z * cw.x + f01.z * cw.y + f10.z * cw.z + f11.z * cw.w;
            acc.w *= f00.w * cw.x + f01.w * cw.y + f10.w * cw.z + f11.w * cw.w;
        }
        reinterpret_cast<float4*>(feat + (size_t)p * ldf + l * C)[sub] = acc;
    }
}

// Backward, common part: for one (point, level, channel) sample the six planes and return
// gi[i] = dL/dfeat * product of the other five samples, plus the coordinate derivatives.
struct SixSamples { float gi[6], ddx[6], ddy[6]; };

__device__ __forceinline__ SixSamples six_samples(const PlaneTable& T, int l, int C, int ch, const Corner cn[6], float g)
{
    SixSamples o;
    float v[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float* __restrict__ base = T.p[l * 6 + i] + ch;
        const float f00 = base[(size_t)cn[i].i00 * C], f01 = base[(size_t)cn[i].i01 * C];
        const float f10 = base[(size_t)cn[i].i10 * C], f11 = base[(size_t)cn[i].i11 * C];
        v[i] = f00 * cn[i].w00 + f01 * cn[i].w01 + f10 * cn[i].w10 + f11 * cn[i].w11;
        o.ddx[i] = ((f01 - f00) * (1.f - cn[i].wy) + (f11 - f10) * cn[i].wy) * cn[i].sx;
        o.ddy[i] = ((f10 - f00) * (1.f - cn[i].wx) + (f11 - f01) * cn[i].wx) * cn[i].sy;
    }
    // product of the other five planes via prefix / suffix products (no division)
    float pre[6], suf[6];
    pre[0] = 1.f;
#pragma unroll
    for (int i = 1; i < 6; i++) pre[i] = pre[i - 1] * v[i - 1];
    suf[5] = 1.f;
#pragma unroll
    for (int i = 4; i >= 0; i--) suf[i] = suf[i + 1] * v[i + 1];
#pragma unroll
    for (int i = 0; i < 6; i++) o.gi[i] = g * pre[i] * suf[i];
    return o;
}

// ---- plane gradients: bin, then reduce in cell order -------------------------------------------------
// Memory-side float atomics run at one chip-wide rate (~1.2 TB/s measured), and the direct scatter issues
// 4 corner rows per (point, plane, level): 4.8 M rows of 128 B at C2 = 250 us for the spatial planes alone.
// Neighbouring points share texels, so the adds are combined BEFORE they reach memory:
//   1. hexplane_backward_point_kernel (per point, no atomics on floats) writes gsix[p][l][plane][ch] =
//      dL/d(sample of that plane) and the coordinate gradients, and ranks the point inside its finest-level
//      cell of three "sort planes" (one int atomic each);
//   2. a counting sort per sort plane orders the points by cell: xy A-major (also groups the xt plane by x),
//      yz A-major (groups yt by y), xz B-major (groups zt by z);
//   3. hexplane_backward_reduce_kernel walks fixed runs of RUN consecutive sorted points per (plane, level),
//      accumulates the four corner sums of the current cell in registers and flushes them with four atomic
//      rows only when the cell changes: ~6x (fine level) to ~25x (coarse level, time planes) fewer atomic
//      rows.  Correctness never depends on the grouping (a flush happens on every cell change), so per-point
//      timestamps, clustered scenes and non-power-of-two multires just change how often it flushes.
#ifndef HEX_RUN
#define HEX_RUN 32
#endif
// plane i of a level is served by sort kSortOf[i]; its (A, B) coordinates are fields kSelA/kSelB of the sorted
// record (p, a, b, t'): S0 = (x, y), S1 = (y, z), S2 = (x, z)
__device__ __constant__ const int kSortOf[6] = {0, 2, 0, 1, 1, 2};
__device__ __constant__ const int kSelA[6] = {0, 0, 0, 0, 0, 1};
__device__ __constant__ const int kSelB[6] = {1, 1, 2, 1, 2, 2};

__device__ __forceinline__ int hex_key(const HexBins& B, int l, int s, int x0, int y0, int Ww, int Hh)
{
    const int sh = B.shift[l][s];
    const int xs = x0 >> sh, ys = y0 >> sh;
    const int ws = ((Ww - 1) >> sh) + 1, hs = ((Hh - 1) >> sh) + 1;
    return B.bmajor[s] ? ys * ws + xs : xs * hs + ys;
}

// C/4 lanes per point, four channels (one float4 per corner) per lane, as in the forward gather.
__global__ __launch_bounds__(256) void hexplane_backward_point_kernel(
    int P, int C, int L, int lpp_shift, PlaneTable T, HexBins B, const float* __restrict__ aabb,
    const float* __restrict__ xyz, const float* __restrict__ tprime, const float* __restrict__ dfeat, int ldf,
    float4* __restrict__ dcoord,               // [P]: (d/dx, d/dy, d/dz, d/dt') in normalised coordinates
    float* __restrict__ gsix,                  // [P][L][6][C]: dL/d(sample) of every plane
    int* __restrict__ hist,                    // [L][HEX_SORTS][nbins], zeroed by the caller
    int* __restrict__ rank, int* __restrict__ key,   // [L][HEX_SORTS][P]
    int do_rank)                               // 0: the point order is taken from the caller's cache, nothing is ranked
{
    const int lanes = 1 << lpp_shift;
    const int gtid = blockIdx.x * 256 + threadIdx.x;
    const int p = gtid >> lpp_shift;
    const int sub = gtid & (lanes - 1);
    const bool live = p < P;
    const int q4 = C >> 2;
    float pts[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float lo = aabb[a], hi = aabb[3 + a];
            pts[a] = (xyz[3 * (size_t)p + a] - lo) * (2.0f / (hi - lo)) - 1.0f;
        }
        pts[3] = tprime[p];
    }
    float dpts[4] = {0.f, 0.f, 0.f, 0.f};
    if (live && do_rank) {                      // rank the point inside its cell of every level's sort planes
        for (int ls = sub; ls < L * HEX_SORTS; ls += lanes) {    // one (level, sort) per lane of the point
            const int l = ls / HEX_SORTS, s = ls - l * HEX_SORTS;
            const int i = B.plane[s], pl = l * 6 + i;
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int k = 0; k < 6; k++)
                if (k == i) { u = pts[kPairA[k]]; v = pts[kPairB[k]]; }
            const Corner c = corner_of(u, v, T.w[pl], T.h[pl]);
            const int k = hex_key(B, l, s, c.x0, c.y0, T.w[pl], T.h[pl]);
            key[(size_t)ls * P + p] = k;
            rank[(size_t)ls * P + p] = atomicAdd(&hist[(size_t)ls * B.nbins + k], 1);
        }
    }
    if (live) {
        for (int l = 0; l < L; l++) {
            float4 v[6], dx[6], dy[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int pl = l * 6 + i;
                const Corner c = corner_of(pts[kPairA[i]], pts[kPairB[i]], T.w[pl], T.h[pl]);
                const float4* __restrict__ base = reinterpret_cast<const float4*>(T.p[pl]) + sub;
                const float4 f00 = base[(size_t)c.i00 * q4], f01 = base[(size_t)c.i01 * q4];
                const float4 f10 = base[(size_t)c.i10 * q4], f11 = base[(size_t)c.i11 * q4];
                const float ax = (1.f - c.wy) * c.sx, bx = c.wy * c.sx, ay = (1.f - c.wx) * c.sy, by = c.wx * c.sy;
#define HEX_COMP(m)                                                                                   \
                v[i].m = f00.m * c.w00 + f01.m * c.w01 + f10.m * c.w10 + f11.m * c.w11;               \
                dx[i].m = (f01.m - f00.m) * ax + (f11.m - f10.m) * bx;                                 \
                dy[i].m = (f10.m - f00.m) * ay + (f11.m - f01.m) * by;
                HEX_COMP(x) HEX_COMP(y) HEX_COMP(z) HEX_COMP(w)
#undef HEX_COMP
            }
            const float4 g = *reinterpret_cast<const float4*>(dfeat + (size_t)p * ldf + l * C + 4 * sub);
            // product of the other five planes via prefix / suffix products (no division)
            float4 pre[6], suf[6];
            pre[0] = make_float4(1.f, 1.f, 1.f, 1.f);
            suf[5] = pre[0];
#pragma unroll
            for (int i = 1; i < 6; i++) pre[i] = make_float4(pre[i - 1].x * v[i - 1].x, pre[i - 1].y * v[i - 1].y, pre[i - 1].z * v[i - 1].z, pre[i - 1].w * v[i - 1].w);
#pragma unroll
            for (int i = 4; i >= 0; i--) suf[i] = make_float4(suf[i + 1].x * v[i + 1].x, suf[i + 1].y * v[i + 1].y, suf[i + 1].z * v[i + 1].z, suf[i + 1].w * v[i + 1].w);
            float4* __restrict__ out = reinterpret_cast<float4*>(gsix + ((size_t)p * L + l) * 6 * C) + sub;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const float4 gi = make_float4(g.x * pre[i].x * suf[i].x, g.y * pre[i].y * suf[i].y, g.z * pre[i].z * suf[i].z, g.w * pre[i].w * suf[i].w);
                out[(size_t)i * q4] = gi;
                dpts[kPairA[i]] += (gi.x * dx[i].x + gi.y * dx[i].y) + (gi.z * dx[i].z + gi.w * dx[i].w);
                dpts[kPairB[i]] += (gi.x * dy[i].x + gi.y * dy[i].y) + (gi.z * dy[i].z + gi.w * dy[i].w);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
        for (int off = lanes >> 1; off >= 1; off >>= 1) dpts[a] += __shfl_xor(dpts[a], off, 64);
    if (live && sub == 0) dcoord[p] = make_float4(dpts[0], dpts[1], dpts[2], dpts[3]);
}

// The point kernel with the corner sets of a workgroup's points shared through LDS (see hexplane_forward_shared_kernel):
// per (point, plane) the four texel indices, the four weights and the four coordinate-derivative factors are computed by
// ONE lane instead of by each of the point's C/4 lanes.  Same arithmetic, bit-identical outputs.  C >= 16.
__global__ __launch_bounds__(256) void hexplane_backward_point_shared_kernel(
    int P, int C, int L, int lpp_shift, PlaneTable T, HexBins B, const float* __restrict__ aabb,
    const float* __restrict__ xyz, const float* __restrict__ tprime, const float* __restrict__ dfeat, int ldf,
    float4* __restrict__ dcoord, float* __restrict__ gsix, int* __restrict__ hist, int* __restrict__ rank,
    int* __restrict__ key, int do_rank)
{
    extern __shared__ float4 s_hex[];                    // [ppw] coordinates | [ppw * L * 6][3] corner sets
    const int lanes = 1 << lpp_shift;
    const int ppw = 256 >> lpp_shift;
    const int np = L * 6;
    float4* s_pts = s_hex;
    float4* s_set = s_hex + ppw;
    const int p0 = blockIdx.x * ppw;
    if ((int)threadIdx.x < ppw) {
        const int pp = p0 + (int)threadIdx.x;
        float c[4] = {0.f, 0.f, 0.f, 0.f};
        if (pp < P) {
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float lo = aabb[a], hi = aabb[3 + a];
                c[a] = (xyz[3 * (size_t)pp + a] - lo) * (2.0f / (hi - lo)) - 1.0f;
            }
            c[3] = tprime[pp];
        }
        s_pts[threadIdx.x] = make_float4(c[0], c[1], c[2], c[3]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ppw * np; e += 256) {
        const int lq = e / np, pl = e - lq * np, i = pl % 6;
        const float* pp = reinterpret_cast<const float*>(&s_pts[lq]);
        const Corner c = corner_of(pp[pair_a(i)], pp[pair_b(i)], T.w[pl], T.h[pl]);
        s_set[3 * e] = make_float4(__int_as_float(c.i00), __int_as_float(c.i01), __int_as_float(c.i10), __int_as_float(c.i11));
        s_set[3 * e + 1] = make_float4(c.w00, c.w01, c.w10, c.w11);
        s_set[3 * e + 2] = make_float4((1.f - c.wy) * c.sx, c.wy * c.sx, (1.f - c.wx) * c.sy, c.wx * c.sy);
    }
    __syncthreads();
    const int lp = (int)threadIdx.x >> lpp_shift;
    const int sub = (int)threadIdx.x & (lanes - 1);
    const int p = p0 + lp;
    const bool live = p < P;
    const int q4 = C >> 2;
    const float4 my = s_pts[lp];
    const float pts[4] = {my.x, my.y, my.z, my.w};
    float dpts[4] = {0.f, 0.f, 0.f, 0.f};
    if (live && do_rank) {                      // rank the point inside its cell of every level's sort planes
        for (int ls = sub; ls < L * HEX_SORTS; ls += lanes) {    // one (level, sort) per lane of the point
            const int l = ls / HEX_SORTS, s = ls - l * HEX_SORTS;
            const int i = B.plane[s], pl = l * 6 + i;
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int k = 0; k < 6; k++)
                if (k == i) { u = pts[kPairA[k]]; v = pts[kPairB[k]]; }
            const Corner c = corner_of(u, v, T.w[pl], T.h[pl]);
            const int k = hex_key(B, l, s, c.x0, c.y0, T.w[pl], T.h[pl]);
            key[(size_t)ls * P + p] = k;
            rank[(size_t)ls * P + p] = atomicAdd(&hist[(size_t)ls * B.nbins + k], 1);
        }
    }
    if (live) {
        const float4* __restrict__ mine = s_set + 3 * (size_t)lp * np;
        for (int l = 0; l < L; l++) {
            float4 v[6], dx[6], dy[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int pl = l * 6 + i;
                const float4 ci = mine[3 * pl], cw = mine[3 * pl + 1], ca = mine[3 * pl + 2];
                const float4* __restrict__ base = reinterpret_cast<const float4*>(T.p[pl]) + sub;
                const float4 f00 = base[(size_t)__float_as_int(ci.x) * q4], f01 = base[(size_t)__float_as_int(ci.y) * q4];
                const float4 f10 = base[(size_t)__float_as_int(ci.z) * q4], f11 = base[(size_t)__float_as_int(ci.w) * q4];
#define HEX_COMP(m)                                                                                   \
                v[i].m = f00.m * cw.x + f01.m * cw.y + f10.m * cw.z + f11.m * cw.w;                   \
                dx[i].m = (f01.m - f00.m) * ca.x + (f11.m - f10.m) * ca.y;                             \
                dy[i].m = (f10.m - f00.m) * ca.z + (f11.m - f01.m) * ca.w;
                HEX_COMP(x) HEX_COMP(y) HEX_COMP(z) HEX_COMP(w)
#undef HEX_COMP
            }
            const float4 g = *reinterpret_cast<const float4*>(dfeat + (size_t)p * ldf + l * C + 4 * sub);
            float4 pre[6], suf[6];
            pre[0] = make_float4(1.f, 1.f, 1.f, 1.f);
            suf[5] = pre[0];
#pragma unroll
            for (int i = 1; i < 6; i++) pre[i] = make_float4(pre[i - 1].x * v[i - 1].x, pre[i - 1].y * v[i - 1].y, pre[i - 1].z * v[i - 1].z, pre[i - 1].w * v[i - 1].w);
#pragma unroll
            for (int i = 4; i >= 0; i--) suf[i] = make_float4(suf[i + 1].x * v[i + 1].x, suf[i + 1].y * v[i + 1].y, suf[i + 1].z * v[i + 1].z, suf[i + 1].w * v[i + 1].w);
            float4* __restrict__ out = reinterpret_cast<float4*>(gsix + ((size_t)p * L + l) * 6 * C) + sub;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const float4 gi = make_float4(g.x * pre[i].x * suf[i].x, g.y * pre[i].y * suf[i].y, g.z * pre[i].z * suf[i].z, g.w * pre[i].w * suf[i].w);
                out[(size_t)i * q4] = gi;
                dpts[kPairA[i]] += (gi.x * dx[i].x + gi.y * dx[i].y) + (gi.z * dx[i].z + gi.w * dx[i].w);
                dpts[kPairB[i]] += (gi.x * dy[i].x + gi.y * dy[i].y) + (gi.z * dy[i].z + gi.w * dy[i].w);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
        for (int off = lanes >> 1; off >= 1; off >>= 1) dpts[a] += __shfl_xor(dpts[a], off, 64);
    if (live && sub == 0) dcoord[p] = make_float4(dpts[0], dpts[1], dpts[2], dpts[3]);
}

// exclusive prefix sum of each sort's histogram, in place; one workgroup of 1024 lanes per sort
__global__ __launch_bounds__(1024) void hexplane_bin_scan_kernel(int nbins, int* __restrict__ hist)
{
    __shared__ int part[16];
    __shared__ int carry_s;
    int* h = hist + (size_t)blockIdx.x * nbins;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    // tiles of 4096 bins: four consecutive bins per lane, scanned inside the wave by shuffles and across the
    // 16 waves through LDS (no per-lane arrays: a spilling kernel pays a scratch set-up on every dispatch)
    for (int t0 = 0; t0 < nbins; t0 += 4096) {
        const int b = t0 + 4 * threadIdx.x;
        int v0 = b < nbins ? h[b] : 0, v1 = b + 1 < nbins ? h[b + 1] : 0;
        int v2 = b + 2 < nbins ? h[b + 2] : 0, v3 = b + 3 < nbins ? h[b + 3] : 0;
        const int s = v0 + v1 + v2 + v3;
        int inc = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(inc, off, 64);
            if (lane >= off) inc += u;
        }
        if (lane == 63) part[wave] = inc;
        __syncthreads();
        int prefix = carry;
        for (int w = 0; w < wave; w++) prefix += part[w];
        if (threadIdx.x == 1023) carry_s = prefix + inc;
        int run = prefix + inc - s;
        if (b < nbins) h[b] = run;
        run += v0;
        if (b + 1 < nbins) h[b + 1] = run;
        run += v1;
        if (b + 2 < nbins) h[b + 2] = run;
        run += v2;
        if (b + 3 < nbins) h[b + 3] = run;
        __syncthreads();
        carry = carry_s;
    }
}

// sorted record of sort s at position start[key] + rank: (p, a, b, t') with (a, b) = (x,y), (y,z), (x,z)
__global__ __launch_bounds__(256) void hexplane_bin_scatter_kernel(int P, int L, int nbins, const float* __restrict__ aabb,
                                                                   const float* __restrict__ xyz,
                                                                   const float* __restrict__ tprime,
                                                                   const int* __restrict__ start,
                                                                   const int* __restrict__ rank,
                                                                   const int* __restrict__ key, float4* __restrict__ rec,
                                                                   int* __restrict__ order)     // may be null: [L][HEX_SORTS][P], sorted position -> point
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    float pts[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = aabb[a], hi = aabb[3 + a];
        pts[a] = (xyz[3 * (size_t)p + a] - lo) * (2.0f / (hi - lo)) - 1.0f;
    }
    const float t = tprime[p];
    const float ra[HEX_SORTS] = {pts[0], pts[1], pts[0]}, rb[HEX_SORTS] = {pts[1], pts[2], pts[2]};
    for (int l = 0; l < L; l++) {
#pragma unroll
        for (int s = 0; s < HEX_SORTS; s++) {
            const size_t ls = (size_t)l * HEX_SORTS + s;
            const int pos = start[ls * nbins + key[ls * P + p]] + rank[ls * P + p];
            rec[ls * P + pos] = make_float4(__int_as_float(p), ra[s], rb[s], t);
            if (order) order[ls * P + pos] = p;
        }
    }
}

// The sorted records from a cached order (gs4d_deform_grads::hex_order): position -> point as recorded by an earlier
// backward, coordinates as they are now.
__global__ __launch_bounds__(256) void hexplane_rec_from_order_kernel(int P, int L, const float* __restrict__ aabb,
                                                                      const float* __restrict__ xyz,
                                                                      const float* __restrict__ tprime,
                                                                      const int* __restrict__ order, float4* __restrict__ rec)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)L * HEX_SORTS * P) return;
    const int s = (int)((e / P) % HEX_SORTS);
    const int p = min(max(order[e], 0), P - 1);
    float pts[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = aabb[a], hi = aabb[3 + a];
        pts[a] = (xyz[3 * (size_t)p + a] - lo) * (2.0f / (hi - lo)) - 1.0f;
    }
    const float ra = s == 1 ? pts[1] : pts[0], rb = s == 0 ? pts[1] : pts[2];
    rec[e] = make_float4(__int_as_float(p), ra, rb, tprime[p]);
}

// blockIdx.y = level * 6 + plane.  A group of min(C, 64) lanes (one per channel) walks HEX_RPW * HEX_RUN
// consecutive sorted points, a workgroup 256 / lanes such groups (consecutive ranges).  Per step of HEX_RUN
// points the corners are computed once per entry (one lane each, next step's records already in flight) into
// LDS and the HEX_RUN gradient rows of the step are gathered together.  A group keeps the four corner sums of
// its current cell in registers across steps, steps by one texel along the sort's minor axis keep the two
// shared corners, and memory is touched only when the cell really changes.  What a group still holds at its
// end is merged with its neighbours' through LDS (plain stores: measured, ds_add_f32 tiles cost more than the
// memory atomics they saved), so a cell spanning the whole workgroup -- every time plane, whose points share
// two texel rows -- reaches memory once per workgroup.  Same-line memory-side atomics serialise at ~0.5 us
// each: the 128 hot rows of the coarse time planes alone cost 57 us when every run flushed to memory.
#ifndef HEX_RPW
#define HEX_RPW 1          // steps per group: 4 -> 1 in round 3 (the walk of a group is a serial chain: 47 us at ANY point count below 100 k with four steps; the extra end-of-group merges cost less than the shorter chain saves: 135 -> 127 us at C2, neutral at C5)
#endif
#define HEX_GATHER 16
template <int lanes_shift>
__global__ __launch_bounds__(256) void hexplane_backward_reduce_kernel(int P, int C, int L, PlaneTable T,
                                                                       const float4* __restrict__ rec,
                                                                       const float* __restrict__ gsix, int ft_parts,
                                                                       const float* __restrict__ ft_partial,
                                                                       float* __restrict__ d_ft_w, float* __restrict__ d_ft_b)
{
    extern __shared__ float4 s_mem[];                       // corner weights | corner indices | end-of-group partials
    if (ft_parts && blockIdx.x == 0 && blockIdx.y == 0) {
        // One workgroup first does what ft_embed_final_kernel (csrc/deform.hip) did in a launch of its own, thread for thread:
        // the fixed-order sum of hexplane_coord_reduce_kernel's partial sums onto d(ft_w), d(ft_b) (t' = w0 t + w1 f + b).
        float (*s_ft)[3] = reinterpret_cast<float (*)[3]>(s_mem);      // 3 KB of the staging area, before its first use
        float a[3] = {0.f, 0.f, 0.f};
        for (int q = threadIdx.x; q < ft_parts; q += 256)
            for (int k = 0; k < 3; k++) a[k] += ft_partial[3 * (size_t)q + k];
        for (int k = 0; k < 3; k++) s_ft[threadIdx.x][k] = a[k];
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) {
            if ((int)threadIdx.x < off)
                for (int k = 0; k < 3; k++) s_ft[threadIdx.x][k] += s_ft[threadIdx.x + off][k];
            __syncthreads();
        }
        if (threadIdx.x == 0) { d_ft_w[0] += s_ft[0][0]; d_ft_w[1] += s_ft[0][1]; d_ft_b[0] += s_ft[0][2]; }
        __syncthreads();
    }
    const int lanes = 1 << lanes_shift;
    const int pl = blockIdx.y, l = pl / 6, i = pl % 6;
    const int groups = 256 >> lanes_shift;
    const int step_entries = groups * HEX_RUN;
    const int grp = threadIdx.x >> lanes_shift;
    const int sub = threadIdx.x & (lanes - 1);
    const float4* __restrict__ R = rec + ((size_t)l * HEX_SORTS + kSortOf[i]) * P;
    const size_t wg_e0 = (size_t)blockIdx.x * step_entries * HEX_RPW;
    if (wg_e0 >= (size_t)P) return;                         // uniform
    const int Ww = T.w[pl], Hh = T.h[pl];
    const int selA = kSelA[i], selB = kSelB[i];
    float4* s_w = s_mem;
    int4* s_i = reinterpret_cast<int4*>(s_mem + step_entries);
    float* s_part = reinterpret_cast<float*>(s_mem + 2 * step_entries);   // [groups][4][lanes]
    int* s_cell = reinterpret_cast<int*>(s_part + 4 * 256);               // [groups][3]
    float* __restrict__ G = T.g[pl];
    const float* __restrict__ gbase = gsix + ((size_t)l * 6 + i) * C;
    const size_t pstride = (size_t)L * 6 * C;
    // staging slot k of a step belongs to group k / HEX_RUN, entry k % HEX_RUN of that group's step
    constexpr int per_thread = lanes_shift >= 5 ? 1 : (32 >> lanes_shift);   // staging slots per thread: 1 (lanes >= 32) .. 8 (lanes = 4)
    auto entry_of = [&](int it, int k) { return wg_e0 + ((size_t)(k / HEX_RUN) * HEX_RPW + it) * HEX_RUN + (k % HEX_RUN); };

    for (int ch0 = 0; ch0 < C; ch0 += lanes) {
        const int ch = ch0 + sub;
        float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
        int cell = -1, dx = 0, dy = 0;
        auto flush = [&]() {
            float* g0 = G + ((size_t)(cell >> 16) * Ww + (cell & 0xFFFF)) * C + ch;
            unsafeAtomicAdd(g0, a00);
            unsafeAtomicAdd(g0 + (size_t)dx * C, a01);
            unsafeAtomicAdd(g0 + (size_t)dy * Ww * C, a10);
            unsafeAtomicAdd(g0 + ((size_t)dy * Ww + dx) * C, a11);
        };
        float4 rn[per_thread];
#pragma unroll
        for (int j = 0; j < per_thread; j++)
            {
                const int k = threadIdx.x + 256 * j;
                rn[j] = R[min(entry_of(0, min(k, step_entries - 1)), (size_t)P - 1)];
            }
        for (int it = 0; it < HEX_RPW; it++) {
            __syncthreads();                                // the previous step's readers are done
#pragma unroll
            for (int j = 0; j < per_thread; j++)
                {
                    const int k = threadIdx.x + 256 * j;
                    if (k < step_entries) {
                        const float f[3] = {rn[j].y, rn[j].z, rn[j].w};
                        const Corner c = corner_of(f[selA], f[selB], Ww, Hh);
                        s_w[k] = make_float4(c.w00, c.w01, c.w10, c.w11);
                        s_i[k] = make_int4(__float_as_int(rn[j].x), c.x0 | (c.y0 << 16), c.x1 - c.x0, c.y1 - c.y0);
                    }
                }
            __syncthreads();
            if (it + 1 < HEX_RPW) {
#pragma unroll
                for (int j = 0; j < per_thread; j++)
                    {
                        const int k = threadIdx.x + 256 * j;
                        rn[j] = R[min(entry_of(it + 1, min(k, step_entries - 1)), (size_t)P - 1)];
                    }
            }
            const size_t e0 = wg_e0 + ((size_t)grp * HEX_RPW + it) * HEX_RUN;
            if (e0 >= (size_t)P) continue;
            const int n = (int)min((size_t)HEX_RUN, (size_t)P - e0);
            const float4* __restrict__ myw = s_w + grp * HEX_RUN;
            const int4* __restrict__ myi = s_i + grp * HEX_RUN;
            for (int e = 0; e < n; e += HEX_GATHER) {
                float g[HEX_GATHER];
#pragma unroll
                for (int u = 0; u < HEX_GATHER; u++) g[u] = gbase[(size_t)myi[min(e + u, n - 1)].x * pstride + ch];
#pragma unroll
                for (int u = 0; u < HEX_GATHER; u++) {
                    if (e + u >= n) break;
                    const int4 ci = myi[e + u];
                    const float4 cw = myw[e + u];
                    if (ci.y != cell || ci.z != dx || ci.w != dy) {
                        if (cell >= 0 && ci.z == dx && ci.w == dy && ci.y == cell + 0x10000 && dy == 1) {
                            float* g0 = G + ((size_t)(cell >> 16) * Ww + (cell & 0xFFFF)) * C + ch;
                            unsafeAtomicAdd(g0, a00);
                            unsafeAtomicAdd(g0 + (size_t)dx * C, a01);
                            a00 = a10; a01 = a11; a10 = a11 = 0.f;
                        } else if (cell >= 0 && ci.z == dx && ci.w == dy && ci.y == cell + 1 && dx == 1) {
                            float* g0 = G + ((size_t)(cell >> 16) * Ww + (cell & 0xFFFF)) * C + ch;
                            unsafeAtomicAdd(g0, a00);
                            unsafeAtomicAdd(g0 + (size_t)dy * Ww * C, a10);
                            a00 = a01; a10 = a11; a01 = a11 = 0.f;
                        } else {
                            if (cell >= 0) flush();
                            a00 = a01 = a10 = a11 = 0.f;
                        }
                        cell = ci.y; dx = ci.z; dy = ci.w;
                    }
                    a00 += g[u] * cw.x; a01 += g[u] * cw.y; a10 += g[u] * cw.z; a11 += g[u] * cw.w;
                }
            }
        }
        // merge what the groups still hold, in group (= sorted) order, by the lanes of group 0
        s_part[(grp * 4 + 0) * lanes + sub] = a00; s_part[(grp * 4 + 1) * lanes + sub] = a01;
        s_part[(grp * 4 + 2) * lanes + sub] = a10; s_part[(grp * 4 + 3) * lanes + sub] = a11;
        if (sub == 0) { s_cell[grp * 3] = cell; s_cell[grp * 3 + 1] = dx; s_cell[grp * 3 + 2] = dy; }
        __syncthreads();
        if (grp == 0) {
            a00 = a01 = a10 = a11 = 0.f;
            cell = -1;
            for (int r = 0; r < groups; r++) {
                const int rc = s_cell[r * 3], rdx = s_cell[r * 3 + 1], rdy = s_cell[r * 3 + 2];
                if (rc < 0) continue;
                if (rc != cell || rdx != dx || rdy != dy) {
                    if (cell >= 0) flush();
                    a00 = a01 = a10 = a11 = 0.f;
                    cell = rc; dx = rdx; dy = rdy;
                }
                a00 += s_part[(r * 4 + 0) * lanes + sub]; a01 += s_part[(r * 4 + 1) * lanes + sub];
                a10 += s_part[(r * 4 + 2) * lanes + sub]; a11 += s_part[(r * 4 + 3) * lanes + sub];
            }
            if (cell >= 0) flush();
        }
        __syncthreads();
    }
}

// dL_dpoints += (coordinate gradient through the planes) * d(normalised)/d(world); the t' part is
// reduced per workgroup for the time/force embedding gradient (fixed order: deterministic given its inputs).
__global__ __launch_bounds__(256) void hexplane_coord_reduce_kernel(int P, const float4* __restrict__ part,
                                                                    const float* __restrict__ aabb,
                                                                    const float* __restrict__ time,
                                                                    const float* __restrict__ force, int use_time,
                                                                    int use_force, float* __restrict__ dL_dpoints,
                                                                    float* __restrict__ ft_partial)
{
    __shared__ float s[256][3];
    const int p = blockIdx.x * 256 + threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (p < P) {
        const float4 u = part[p];
        const float d[3] = {u.x, u.y, u.z};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float lo = aabb[a], hi = aabb[3 + a];
            dL_dpoints[3 * (size_t)p + a] += d[a] * (2.0f / (hi - lo));
        }
        const float g = u.w;
        a0 = g * (use_time ? time[p] : 0.f);
        a1 = g * (use_force ? force[p] : 0.f);
        a2 = g;
    }
    s[threadIdx.x][0] = a0; s[threadIdx.x][1] = a1; s[threadIdx.x][2] = a2;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 3; k++) s[threadIdx.x][k] += s[threadIdx.x + off][k];
        __syncthreads();
    }
    if (threadIdx.x < 3) ft_partial[3 * (size_t)blockIdx.x + threadIdx.x] = s[0][threadIdx.x];
}

// hexplane_rec_from_order_kernel and hexplane_coord_reduce_kernel in one launch (one batch, cached point order: 15 of 16
// iterations): thread t does entry t of the records and, in the first ceil(P / 256) workgroups, point t of the per-point
// tail -- the same arithmetic as the two kernels, thread for thread.
__global__ __launch_bounds__(256) void hexplane_rec_coord_kernel(int P, int L, const float* __restrict__ aabb,
                                                                 const float* __restrict__ xyz, const float* __restrict__ tprime,
                                                                 const int* __restrict__ order, float4* __restrict__ rec,
                                                                 const float4* __restrict__ part, const float* __restrict__ time,
                                                                 const float* __restrict__ force, int use_time, int use_force,
                                                                 float* __restrict__ dL_dpoints, float* __restrict__ ft_partial)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e < (size_t)L * HEX_SORTS * P) {
        const int s = (int)((e / P) % HEX_SORTS);
        const int p = min(max(order[e], 0), P - 1);
        float pts[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float lo = aabb[a], hi = aabb[3 + a];
            pts[a] = (xyz[3 * (size_t)p + a] - lo) * (2.0f / (hi - lo)) - 1.0f;
        }
        const float ra = s == 1 ? pts[1] : pts[0], rb = s == 0 ? pts[1] : pts[2];
        rec[e] = make_float4(__int_as_float(p), ra, rb, tprime[p]);
    }
    if ((size_t)blockIdx.x * 256 >= (size_t)P) return;      // uniform: the per-point tail belongs to the first workgroups
    __shared__ float s[256][3];
    const int p = blockIdx.x * 256 + threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (p < P) {
        const float4 u = part[p];
        const float d[3] = {u.x, u.y, u.z};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float lo = aabb[a], hi = aabb[3 + a];
            dL_dpoints[3 * (size_t)p + a] += d[a] * (2.0f / (hi - lo));
        }
        const float g = u.w;
        a0 = g * (use_time ? time[p] : 0.f);
        a1 = g * (use_force ? force[p] : 0.f);
        a2 = g;
    }
    s[threadIdx.x][0] = a0; s[threadIdx.x][1] = a1; s[threadIdx.x][2] = a2;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 3; k++) s[threadIdx.x][k] += s[threadIdx.x + off][k];
        __syncthreads();
    }
    if (threadIdx.x < 3) ft_partial[3 * (size_t)blockIdx.x + threadIdx.x] = s[0][threadIdx.x];
}

}  // namespace

// GS4D_HEX_CORNERS=private (tests / experiments): every lane computes its point's corner sets itself, as before round 3
static bool hex_private_corners()
{
    static const bool v = [] { const char* e = getenv("GS4D_HEX_CORNERS"); return e && e[0] == 'p'; }();
    return v;
}

HexBins gs4d_hex_bins(const gs4d_deform_desc* d)
{
    HexBins B = {};
    const int planes[HEX_SORTS] = {0, 3, 1};
    const int bmajor[HEX_SORTS] = {0, 0, 1};
    B.nbins = 1;
    for (int s = 0; s < HEX_SORTS; s++) {
        B.plane[s] = planes[s]; B.bmajor[s] = bmajor[s];
        for (int lv = 0; lv < d->L; lv++) {
            const int pl = lv * 6 + planes[s];
            int sh = 0;
            while ((long long)(((d->plane_w[pl] - 1) >> sh) + 1) * (((d->plane_h[pl] - 1) >> sh) + 1) > HEX_MAX_BINS) sh++;
            B.shift[lv][s] = sh;
            const int nb = (((d->plane_w[pl] - 1) >> sh) + 1) * (((d->plane_h[pl] - 1) >> sh) + 1);
            B.nbins = nb > B.nbins ? nb : B.nbins;
        }
    }
    return B;
}

// The plane-gradient pipeline (point kernel -> counting sort -> reduce) runs over batches of points so that its
// per-sample scratch (gsix: 6 L C floats per point, 6 KB at C = 128) stays bounded: <= GS4D_HEX_BATCH_BYTES
// (default 2 GiB) however many points the scene has.  A 5 M-point, C = 128 scene needed 31 GB before.
size_t gs4d_hex_batch_points(const gs4d_deform_desc* d, size_t P)
{
    static const size_t limit = [] {
        const char* e = getenv("GS4D_HEX_BATCH_BYTES");
        const long long v = e ? atoll(e) : 0;
        return v > 0 ? (size_t)v : ((size_t)2 << 30);
    }();
    size_t n = limit / ((size_t)d->L * 6 * d->C * sizeof(float));
    n = n / 256 * 256;
    if (n < 256) n = 256;
    return n < P ? n : P;
}

int gs4d_hex_launch_forward(const gs4d_deform_desc* d, int P, const float* points, const float* time, const float* force,
                            float* tprime, float* feat, int ldf, hipStream_t s)
{
    PlaneTable T = {};
    for (int i = 0; i < d->L * 6; i++) {
        if (!d->planes[i] || ((uintptr_t)d->planes[i] & 15)) { gs4d_set_error("deform: plane %d null or not 16-byte aligned", i); return GS4D_EINVAL; }
        T.p[i] = d->planes[i]; T.w[i] = d->plane_w[i]; T.h[i] = d->plane_h[i];
    }
    int lpp_shift = 0;
    while ((1 << lpp_shift) < d->C / 4) lpp_shift++;
    const size_t threads = (size_t)P << lpp_shift;
    if (lpp_shift >= 2 && !hex_private_corners()) {      // four lanes per point or more: corner sets shared through LDS
        const int ppw = 256 >> lpp_shift;
        const size_t lds = sizeof(float4) * ((size_t)ppw + 2 * (size_t)ppw * d->L * 6);      // <= 49 KB (64 points, 4 levels)
        hipLaunchKernelGGL(hexplane_forward_shared_kernel, dim3((unsigned)(((size_t)P + ppw - 1) / ppw)), dim3(256), lds, s, P,
                           d->C, d->L, lpp_shift, T, d->aabb, points, time, force, d->ft_w, d->ft_b, d->use_time,
                           d->use_force, tprime, feat, ldf);
    } else {
        hipLaunchKernelGGL(hexplane_forward_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, P, d->C,
                           d->L, lpp_shift, T, d->aabb, points, time, force, d->ft_w, d->ft_b, d->use_time,
                           d->use_force, tprime, feat, ldf);
    }
    GS4D_LAUNCH_CHECK("hexplane_forward", 0, s);
    return GS4D_OK;
}

int gs4d_hex_launch_backward(const gs4d_deform_desc* d, const gs4d_deform_grads* g, int P, const float* points,
                             const float* tprime, const float* dfeat, int ldf, float* dxyz_hex, float* gsix, int* hist,
                             int* rank, int* key, float4* rec, const float* time, const float* force, float* dL_dpoints,
                             float* dtprime, int* ft_done, hipStream_t s)
{
    PlaneTable T = {};
    for (int i = 0; i < d->L * 6; i++) {
        if (!g->planes[i]) { gs4d_set_error("deform backward: plane gradient %d is null", i); return GS4D_EINVAL; }
        T.p[i] = d->planes[i]; T.g[i] = g->planes[i]; T.w[i] = d->plane_w[i]; T.h[i] = d->plane_h[i];
    }
    int lanes_shift = 0;
    while ((1 << lanes_shift) < (d->C < 64 ? d->C : 64)) lanes_shift++;
    const HexBins B = gs4d_hex_bins(d);
    int lpp_shift = 0;
    while ((1 << lpp_shift) < d->C / 4) lpp_shift++;
    const size_t HB = gs4d_hex_batch_points(d, (size_t)P);
    const size_t hist_bytes = sizeof(int) * (size_t)d->L * HEX_SORTS * B.nbins;
    const int step_entries = (256 >> lanes_shift) * HEX_RUN;
    const size_t per_wg = (size_t)step_entries * HEX_RPW;
    const size_t lds = 2 * sizeof(float4) * step_entries + sizeof(float) * 4 * 256 + sizeof(int) * 3 * 64;
    void (*kern)(int, int, int, PlaneTable, const float4*, const float*, int, const float*, float*, float*) =
        lanes_shift == 2 ? hexplane_backward_reduce_kernel<2> : lanes_shift == 3 ? hexplane_backward_reduce_kernel<3> :
        lanes_shift == 4 ? hexplane_backward_reduce_kernel<4> : lanes_shift == 5 ? hexplane_backward_reduce_kernel<5> :
                           hexplane_backward_reduce_kernel<6>;
    static bool point_attr = false;
    if (!point_attr) {
        GS4D_HIP_CHECK(hipFuncSetAttribute((const void*)hexplane_backward_point_shared_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        point_attr = true;
    }
    static bool reduce_attr[8] = {};
    if (!reduce_attr[lanes_shift]) {
        GS4D_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        reduce_attr[lanes_shift] = true;
    }
    // batches of HB points (one batch unless the scene is large, see hex_batch_points): the sorted records and
    // gsix carry batch-local point indices, the plane gradients accumulate over the batches
    // the caller's cached order holds one [L][HEX_SORTS][n] block of batch-local indices per batch, batch after batch
    // (the batch size follows from the descriptor and P alone, so a recorded order and a later pass agree on it)
    const bool reuse_order = g->hex_order && g->hex_order_valid;
    const bool record = g->hex_order && !g->hex_order_valid;
    const int nred = (P + 255) / 256;
    const bool fold_ft = (size_t)P <= HB && g->ft_w && g->ft_b && (size_t)lds >= sizeof(float) * 3 * 256;   // one batch
    for (size_t p0 = 0; p0 < (size_t)P; p0 += HB) {
        const int n = (int)(((size_t)P - p0) < HB ? ((size_t)P - p0) : HB);
        if (p0 > 0) GS4D_HIP_CHECK(hipMemsetAsync(hist, 0, hist_bytes, s));     // the first batch's zeros come from pe_backward
        const float* pts_b = points + 3 * p0;
        const float* tp_b = tprime + p0;
        const size_t threads = (size_t)n << lpp_shift;
        if (lpp_shift >= 2 && !hex_private_corners()) {
            const int ppw = 256 >> lpp_shift;
            const size_t plds = sizeof(float4) * ((size_t)ppw + 3 * (size_t)ppw * d->L * 6);   // <= 74 KB (64 points, 4 levels)
            hipLaunchKernelGGL(hexplane_backward_point_shared_kernel, dim3((unsigned)((n + ppw - 1) / ppw)), dim3(256), plds, s, n,
                               d->C, d->L, lpp_shift, T, B, d->aabb, pts_b, tp_b, dfeat + p0 * (size_t)ldf, ldf,
                               (float4*)dxyz_hex + p0, gsix, hist, rank, key, reuse_order ? 0 : 1);
        } else {
            hipLaunchKernelGGL(hexplane_backward_point_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, n,
                               d->C, d->L, lpp_shift, T, B, d->aabb, pts_b, tp_b, dfeat + p0 * (size_t)ldf, ldf,
                               (float4*)dxyz_hex + p0, gsix, hist, rank, key, reuse_order ? 0 : 1);
        }
        if (fold_ft && !reuse_order) {
            // one batch: the per-point tail runs here, in front of the reduce launch, whose first workgroup then also adds up
            // the t' partial sums (no ft_embed_final launch)
            hipLaunchKernelGGL(hexplane_coord_reduce_kernel, dim3(nred), dim3(256), 0, s, P, (const float4*)dxyz_hex, d->aabb,
                               time, force, d->use_time, d->use_force, dL_dpoints, dtprime);
        }
        if (reuse_order && fold_ft) {
            // ... and with the cached point order it shares the launch that rebuilds the records
            const size_t ne = (size_t)d->L * HEX_SORTS * n;
            hipLaunchKernelGGL(hexplane_rec_coord_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, n, d->L,
                               d->aabb, pts_b, tp_b, g->hex_order + (size_t)d->L * HEX_SORTS * p0, rec, (const float4*)dxyz_hex,
                               time, force, d->use_time, d->use_force, dL_dpoints, dtprime);
        } else if (reuse_order) {
            const size_t ne = (size_t)d->L * HEX_SORTS * n;
            hipLaunchKernelGGL(hexplane_rec_from_order_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, n, d->L,
                               d->aabb, pts_b, tp_b, g->hex_order + (size_t)d->L * HEX_SORTS * p0, rec);
        } else {
            hipLaunchKernelGGL(hexplane_bin_scan_kernel, dim3(HEX_SORTS * d->L), dim3(1024), 0, s, B.nbins, hist);
            hipLaunchKernelGGL(hexplane_bin_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d->L, B.nbins, d->aabb,
                               pts_b, tp_b, hist, rank, key, rec, record ? g->hex_order + (size_t)d->L * HEX_SORTS * p0 : nullptr);
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)(((size_t)n + per_wg - 1) / per_wg), 6 * d->L), dim3(256), lds, s, n, d->C,
                           d->L, T, rec, gsix, fold_ft ? nred : 0, dtprime, g->ft_w, g->ft_b);
    }
    GS4D_LAUNCH_CHECK("hexplane_backward", 0, s);
    if (!fold_ft) {
        hipLaunchKernelGGL(hexplane_coord_reduce_kernel, dim3(nred), dim3(256), 0, s, P, (const float4*)dxyz_hex, d->aabb,
                           time, force, d->use_time, d->use_force, dL_dpoints, dtprime);
        GS4D_LAUNCH_CHECK("hexplane_coord_reduce", 0, s);
    }
    *ft_done = fold_ft ? 1 : 0;
    return GS4D_OK;
}
