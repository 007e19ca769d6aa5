// Shared by preprocess.hip (counting) and binning.hip (placement): the pieces of GS4D_SORT=direct that both launches need.
//
// Direct placement replaces the two stable radix passes on the tile bits: the (Gaussian, tile) instances are COUNTED per
// tile while the Gaussians are preprocessed, the scan launch turns the counts into the tile lists' positions, and the
// emission launch writes every instance straight into its tile's list.  Device-scope atomics are the price: the chip
// retires ~24 scattered 32-bit atomic adds per nanosecond whatever their addresses (tools/ubench/atomic_rate.hip: 683 k
// adds = 28 us, returning or not, on 2 500 padded counters or on 640 k), so one add per instance and phase (2 x 28 us at
// C2) would cost more than the passes it replaces (41 us).  A workgroup therefore combines its own instances first: its
// 256 Gaussians are neighbours in space when the model keeps its rows in Morton order, their tiles fall into a small
// rectangle of the tile grid (C2: 206 tiles on average, 146 of them touched), and counting inside that rectangle in LDS
// leaves one device-scope add per DISTINCT tile of the workgroup (57 k per phase instead of 768 k).  The LDS counts come
// from a 2-D difference array: four LDS adds per Gaussian (+1 / -1 at the corners of its rectangle) and a prefix sum
// along x, then y -- one LDS add per INSTANCE ran at the ~0.3 lanes per clock of conflicting ds_add (14 us in the
// preprocess launch).  A splat wider or taller than GS4D_BIG_SIDE tiles stays out of the rectangle (one of them would
// blow it up for its whole workgroup) and takes one device-scope add per instance: its instances are distinct tiles
// anyway.  A workgroup whose rectangle is still wider or taller than 63 tiles (rows in no spatial order) falls back to one
// device-scope add per instance for all its Gaussians.
#pragma once
#include "gs4d_internal.h"
#include <limits.h>

#define GS4D_AGG_SIDE 64           // the rectangle + one guard row / column fits 64 x 64 cells (16 KB)
#define GS4D_AGG_BINS (GS4D_AGG_SIDE * GS4D_AGG_SIDE)
#define GS4D_BIG_SIDE 24           // a splat over more tiles than this in x or y is placed instance by instance (12: +2.6 us at C2, 48: same)

namespace gs4d {

// inclusive sum over the 64 lanes (row_shr 1 / 2 / 4 / 8 inside the rows of 16, then row_bcast 15 and 31)
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}

__device__ __forceinline__ bool tile_rect_is_big(int x0, int y0, int x1, int y1)
{
    return x1 - x0 > GS4D_BIG_SIDE || y1 - y0 > GS4D_BIG_SIDE;
}

// Rectangle [bx0, bx0 + bw) x [by0, by0 + bh) of tiles around the rectangles of the workgroup's lanes (has = the lane's
// Gaussian touches tiles [x0, x1) x [y0, y1) and is not big); false if no lane has one.  sum_a / sum_b: two per-lane
// numbers summed over the workgroup on the way (in: the lane's, out: the workgroup's).  One workgroup barrier.
__device__ __forceinline__ bool block_tile_bbox(bool has, int x0, int y0, int x1, int y1, int (*s_red)[6], int& bx0, int& by0,
                                                int& bw, int& bh, uint32_t& sum_a, uint32_t& sum_b)
{
    int a = has ? x0 : INT_MAX, b = has ? y0 : INT_MAX, c = has ? x1 : 0, d = has ? y1 : 0;
    uint32_t e = sum_a, f = sum_b;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a = min(a, __shfl_xor(a, off, 64));
        b = min(b, __shfl_xor(b, off, 64));
        c = max(c, __shfl_xor(c, off, 64));
        d = max(d, __shfl_xor(d, off, 64));
        e += __shfl_xor(e, off, 64);
        f += __shfl_xor(f, off, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_red[wave][0] = a; s_red[wave][1] = b; s_red[wave][2] = c; s_red[wave][3] = d; s_red[wave][4] = (int)e; s_red[wave][5] = (int)f;
    }
    __syncthreads();
    a = INT_MAX; b = INT_MAX; c = 0; d = 0; e = 0; f = 0;
#pragma unroll
    for (int w = 0; w < GS4D_PRE_BLOCK / 64; w++) {
        a = min(a, s_red[w][0]); b = min(b, s_red[w][1]); c = max(c, s_red[w][2]); d = max(d, s_red[w][3]);
        e += (uint32_t)s_red[w][4]; f += (uint32_t)s_red[w][5];
    }
    sum_a = e; sum_b = f;
    if (c == 0) { bx0 = by0 = 0; bw = bh = 1; return false; }
    bx0 = a; by0 = b; bw = c - a; bh = d - b;
    return true;
}

// Clears the cells block_tile_counts() works on; call it early (the rectangle is not known yet, so all of them: 16 words
// per lane) and in front of a workgroup barrier.
__device__ __forceinline__ void block_tile_clear(uint32_t* s_bins)
{
    uint4* z = reinterpret_cast<uint4*>(s_bins);
#pragma unroll
    for (int b = 0; b < GS4D_AGG_BINS / 4 / GS4D_PRE_BLOCK; b++) z[b * GS4D_PRE_BLOCK + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
}

// Gaussians of the workgroup per tile of its rectangle (bw, bh <= GS4D_AGG_SIDE - 1): on return s_bins[y * (bw + 1) + x]
// = number of lanes whose rectangle holds tile (bx0 + x, by0 + y); the guard column x = bw and row y = bh hold 0.
// s_bins: GS4D_AGG_BINS words, ALL ZERO on entry and that visible to the workgroup (block_tile_clear() in front of the
// barrier inside block_tile_bbox()).  2-D difference array (unsigned wrap-around stands in for the -1), prefix along x
// with one wave per row, along y with one wave per column.  Ends with a workgroup barrier.
__device__ __forceinline__ void block_tile_counts(bool has, int x0, int y0, int x1, int y1, int bx0, int by0, int bw, int bh,
                                                  uint32_t* s_bins)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pitch = bw + 1;
    if (has) {
        const int ax = x0 - bx0, ay = y0 - by0, cx = x1 - bx0, cy = y1 - by0;
        atomicAdd(&s_bins[ay * pitch + ax], 1u);
        atomicAdd(&s_bins[ay * pitch + cx], ~0u);
        atomicAdd(&s_bins[cy * pitch + ax], ~0u);
        atomicAdd(&s_bins[cy * pitch + cx], 1u);
    }
    __syncthreads();
    for (int y = wave; y <= bh; y += GS4D_PRE_BLOCK / 64) {
        const uint32_t v = wave_inclusive_sum(lane < pitch ? s_bins[y * pitch + lane] : 0u);
        if (lane < pitch) s_bins[y * pitch + lane] = v;
    }
    __syncthreads();
    for (int x = wave; x <= bw; x += GS4D_PRE_BLOCK / 64) {
        const uint32_t v = wave_inclusive_sum(lane <= bh ? s_bins[lane * pitch + x] : 0u);
        if (lane <= bh) s_bins[lane * pitch + x] = v;
    }
    __syncthreads();
}

// First Gaussian g of the workgroup with s_incl[g] > slot (s_incl = inclusive scan of the 256 tile counts): eight
// dependent LDS reads, branch-free.
__device__ __forceinline__ int slot_owner(const uint32_t* s_incl, uint32_t slot)
{
    int pos = 0;
#pragma unroll
    for (int step = GS4D_PRE_BLOCK / 2; step >= 1; step >>= 1)
        if (s_incl[pos + step - 1] <= slot) pos += step;
    return pos;
}

}  // namespace gs4d
