// Binning stage for gfx950: instance-offset scan, balanced tile-key emission, stable LSD radix
// sort of (tile | depth) keys with wave-level match ranking in LDS, and tile ranges.
//
// Behaviour matched (reference, RAST = submodules/depth-diff-gaussian-rasterization):
//   RAST/cuda_rasterizer/rasterizer_impl.cu:278     inclusive scan of tiles_touched
//   RAST/cuda_rasterizer/rasterizer_impl.cu:70-111  duplicateWithKeys (key = tile<<32 | depth bits, y outer / x inner)
//   RAST/cuda_rasterizer/rasterizer_impl.cu:301-309 stable radix sort on the low 32+bit key bits
//   RAST/cuda_rasterizer/rasterizer_impl.cu:311,116-138 zeroed ranges + identifyTileRanges
// All outputs here are integers and must be bit-exact against the oracle.
#include "gs4d_internal.h"
#include "gs4d_math.h"
#include "tile_place.h"

using namespace gs4d;

GS4D_TIMELINE_DEFINE(binning)

namespace {

// Lanes of the wave whose 8-bit digit equals this lane's (invalid lanes form their own group), as two 32-bit halves.
// Per bit: one ballot, and same &= ~(ballot ^ -bit) on each half (xnor + and) -- written out because the compiler's
// version of `bit ? m : ~m` on a 64-bit mask costs 11 instructions per bit instead of 7.
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid)
{
    const uint64_t vm = __ballot(valid);
    uint32_t lo = (uint32_t)vm, hi = (uint32_t)(vm >> 32);
    if (!valid) { lo = ~lo; hi = ~hi; }
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const uint32_t bit = (d >> b) & 1u;
        const uint64_t mb = __builtin_amdgcn_uicmp(bit, 0u, 33);    // v_cmp_ne_u32: the ballot itself, nothing around it
        const uint32_t neg = 0u - bit;
        lo &= ~((uint32_t)mb ^ neg);
        hi &= ~((uint32_t)(mb >> 32) ^ neg);
    }
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}

// The number of instances of a launch: known to the host (R_dev == nullptr, R_host exact), or still on the device when the
// forward issued this launch SPECULATIVELY, sized for at most R_host instances, before it had read num_rendered (api.hip):
// then the count is counters[0], written by scan_block_sums_kernel earlier in the stream.  More instances than the launch
// was sized for -> GS4D_R_OVERFLOW: the kernel does nothing (the host re-issues the stage with the exact size).
#define GS4D_R_OVERFLOW 0xffffffffu
__device__ __forceinline__ uint32_t resolve_R(uint32_t R_host, const uint32_t* __restrict__ R_dev)
{
    if (!R_dev) return R_host;
    const uint32_t r = *R_dev;
    return r > R_host ? GS4D_R_OVERFLOW : r;
}

// ------------------------------------------------------------------ block-sum scan (1 workgroup; direct placement: a second one lays out the tile lists)
// block_offsets = exclusive scan of block_sums (bits 0..30; bit 31 is the workgroup's prefilter flag); counters[0] =
// total, counters[1] = OR of the flags (both words are written here: nothing clears them).  n is ceil(P/256)
// (391 at P = 100k, 19.5k at P = 5M), so one 1024-lane workgroup looping over chunks is enough.
// mailbox (may be null): one 64-bit word of pinned, coherent host memory; the kernel posts (seq | prefilter flag) << 32
// | total with ONE system-scope release store -- payload and sequence number cannot be seen apart -- and the host,
// polling that word, learns num_rendered without a stream synchronisation (seq is a multiple of 4, never 0; bit 0 = the
// prefilter flag, bit 1 = "bin the next forward by direct placement", see below).
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(int n, const uint32_t* __restrict__ block_sums,
                                                               uint32_t* __restrict__ block_offsets,
                                                               uint32_t* __restrict__ counters,
                                                               unsigned long long* __restrict__ mailbox, uint32_t seq,
                                                               const uint32_t* __restrict__ block_area,
                                                               uint32_t* __restrict__ tile_cursor,
                                                               uint32_t* __restrict__ tile_base, int ntiles)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_area[16];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    if (blockIdx.x == 1) {
        // Direct placement, second workgroup of the launch (beside the block sums, not behind them): tile_cursor holds the
        // instances per tile (counted by the preprocess launch); their exclusive scan is where each tile's list starts.
        // tile_base keeps it (the ranges), tile_cursor becomes the list's write cursor.  Four consecutive tiles per lane.
        __syncthreads();
        for (int base = 0; base < ntiles; base += 4096) {
            const int i0 = base + 4 * (int)threadIdx.x;
            uint32_t c[4];
#pragma unroll
            for (int q = 0; q < 4; q++) c[q] = tile_cursor[(size_t)min(i0 + q, ntiles - 1) * GS4D_TILE_CURSOR_STRIDE];
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (i0 + q >= ntiles) c[q] = 0u;
            const uint32_t v = c[0] + c[1] + c[2] + c[3];
            uint32_t incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            if (lane == 63) s_wave[wave] = incl;
            __syncthreads();
            uint32_t wave_prefix = 0;
            for (int w = 0; w < wave; w++) wave_prefix += s_wave[w];
            const uint32_t carry = s_carry;
            uint32_t run = carry + wave_prefix + incl - v;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (i0 + q < ntiles) {
                    tile_base[i0 + q] = run;
                    tile_cursor[(size_t)(i0 + q) * GS4D_TILE_CURSOR_STRIDE] = run;
                }
                run += c[q];
            }
            __syncthreads();
            if (threadIdx.x == 1023) s_carry = carry + wave_prefix + incl;
            __syncthreads();
        }
        if (threadIdx.x == 0) tile_base[ntiles] = s_carry;
        return;
    }
    __syncthreads();
    int flag = 0;
    uint32_t area = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t raw = i < n ? block_sums[i] : 0u;
        if (i < n) {
            const uint4 a4 = reinterpret_cast<const uint4*>(block_area)[i];     // the workgroup's four per-wave words
            area += a4.x + a4.y + a4.z + a4.w;
        }
        const uint32_t v = raw & 0x7fffffffu;
        flag |= (int)(raw >> 31);
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wave_prefix = 0;
        for (int w = 0; w < wave; w++) wave_prefix += s_wave[w];
        const uint32_t carry = s_carry;
        if (i < n) block_offsets[i] = carry + wave_prefix + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + wave_prefix + incl;
        __syncthreads();
    }
    // block_area summed = the device-scope adds direct placement needs per phase: their exact number when this forward
    // bins by direct placement (tile_cursor != null), an upper bound from the workgroups' tile rectangles otherwise (the
    // adds that happen are ~0.6 of it at C2 and C4).  Whether the NEXT forward of the thread bins by direct placement
    // (api.hip) follows the measured cost of the two ways on this chip (preprocess + scan + emission + sort at C2 / C3 / C4,
    // DESIGN.md section 4): passes 71.5 us + 52 ps per instance, direct placement 25 us + 75 ps per instance + 152 ps per
    // add -- the passes pay six more latency-bound launches, placement pays per instance and per add: break-even at
    // num_rendered + 7 adds = 2.1 M with those figures, 2.4 M after the lists' (depth, index) words became one store:
    //     direct placement  <=>  num_rendered + 7 adds <= 2 400 000.
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) area += __shfl_xor(area, off, 64);
    if (lane == 0) s_area[wave] = area;
    const int any_flag = __syncthreads_or(flag);
    if (threadIdx.x == 0) {
        uint32_t area_sum = 0;
        for (int w = 0; w < 16; w++) area_sum += s_area[w];
        const unsigned long long adds = tile_cursor ? (unsigned long long)area_sum : (unsigned long long)area_sum * 3ull / 5ull;
        const uint32_t compact = (s_carry != 0u && (unsigned long long)s_carry + 7ull * adds <= 2400000ull) ? 2u : 0u;
        counters[0] = s_carry;
        counters[1] = any_flag ? 1u : 0u;
        counters[2] = compact;
        counters[3] = area_sum;
        if (mailbox) {
            mailbox[1] = (unsigned long long)area_sum;       // diagnostics (gs4d_binning_stats); ordered before the post below
            const unsigned long long word = ((unsigned long long)(seq | compact | (any_flag ? 1u : 0u)) << 32) | (unsigned long long)s_carry;
            __hip_atomic_store(mailbox, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------ key emission
// One workgroup per 256 consecutive Gaussians.  The workgroup scans its 256 tile counts in LDS
// (giving point_offsets for free) and then walks its output slots with all 256 lanes: slot s is
// mapped back to its Gaussian by a binary search over the LDS offsets, so a Gaussian that covers
// thousands of tiles costs the same per slot as one that covers four, and every store is
// coalesced.  Emission order inside a Gaussian is y outer, x inner, as the reference.
__global__ __launch_bounds__(GS4D_PRE_BLOCK) void emit_keys_kernel(
    int P, int gx, int gy, const float2* __restrict__ means2D, const float* __restrict__ depths,
    const int* __restrict__ radii, const uint32_t* __restrict__ tiles_touched,
    const uint32_t* __restrict__ block_offsets, uint32_t* __restrict__ point_offsets,
    uint64_t* __restrict__ keys, uint32_t* __restrict__ values, uint2* __restrict__ ranges, int ntiles,
    uint2* __restrict__ seg_owner, int seg_slots, uint32_t cap, const uint32_t* __restrict__ R_dev)
{
    // the tile ranges start as (0, 0) (empty tile) and the backward's segment slots as "nobody"; this launch is the first
    // of the binning stage and has lanes to spare
    for (int t = blockIdx.x * GS4D_PRE_BLOCK + threadIdx.x; t < ntiles; t += gridDim.x * GS4D_PRE_BLOCK)
        ranges[t] = make_uint2(0u, 0u);
    for (int t = blockIdx.x * GS4D_PRE_BLOCK + threadIdx.x; t < seg_slots; t += gridDim.x * GS4D_PRE_BLOCK)
        seg_owner[t] = make_uint2(~0u, 0u);
    // speculative launch with more instances than it was sized for: the tiles stay empty, nothing is emitted
    if (R_dev && resolve_R(cap, R_dev) == GS4D_R_OVERFLOW) return;
    __shared__ uint32_t s_incl[GS4D_PRE_BLOCK];   // inclusive scan of counts inside the workgroup
    __shared__ uint32_t s_wave[GS4D_PRE_BLOCK / 64];
    __shared__ int s_x0[GS4D_PRE_BLOCK], s_y0[GS4D_PRE_BLOCK], s_w[GS4D_PRE_BLOCK];
    __shared__ uint32_t s_depth[GS4D_PRE_BLOCK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = blockIdx.x * GS4D_PRE_BLOCK + tid;
    uint32_t cnt = 0;
    {   // the four inputs requested together (clamped index), not one latency after the other under the tests
        const int ld = min(idx, P - 1);
        const uint32_t c_in = tiles_touched[ld];
        const float2 m = means2D[ld];
        const int r_in = radii[ld];
        const float d_in = depths[ld];
        if (idx < P) {
            cnt = c_in;
            if (cnt) {
                int x0, y0, x1, y1;
                tile_rect(m.x, m.y, r_in, gx, gy, x0, y0, x1, y1);
                s_x0[tid] = x0; s_y0[tid] = y0; s_w[tid] = x1 - x0;
                s_depth[tid] = __float_as_uint(d_in);
            }
        }
    }
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wave_prefix = 0;
    for (int w = 0; w < wave; w++) wave_prefix += s_wave[w];
    incl += wave_prefix;
    s_incl[tid] = incl;
    const uint32_t base = block_offsets[blockIdx.x];
    if (idx < P) point_offsets[idx] = base + incl;
    __syncthreads();

    const uint32_t total = s_incl[GS4D_PRE_BLOCK - 1];
    // Two slots per trip, each mapped back to its Gaussian by a branch-free eight-step search (first g with
    // s_incl[g] > slot): a search is a chain of eight dependent LDS reads, two independent chains overlap.
    for (uint32_t s0 = tid; s0 < total; s0 += 2 * GS4D_PRE_BLOCK) {
        uint32_t sv[2] = {s0, min(s0 + (uint32_t)GS4D_PRE_BLOCK, total - 1u)};
        const bool second = s0 + (uint32_t)GS4D_PRE_BLOCK < total;
        int pos[2] = {0, 0};
#pragma unroll
        for (int step = GS4D_PRE_BLOCK / 2; step >= 1; step >>= 1) {
#pragma unroll
            for (int q = 0; q < 2; q++)
                if (s_incl[pos[q] + step - 1] <= sv[q]) pos[q] += step;
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (q == 1 && !second) break;
            const int g = pos[q];
            const uint32_t s = sv[q];
            const uint32_t excl = g ? s_incl[g - 1] : 0u;
            const uint32_t k = s - excl;
            const int w = s_w[g];
            const int ty = s_y0[g] + (int)(k / (uint32_t)w);
            const int tx = s_x0[g] + (int)(k % (uint32_t)w);
            const uint64_t key = ((uint64_t)(uint32_t)(ty * gx + tx) << 32) | (uint64_t)s_depth[g];
            keys[base + s] = key;
            values[base + s] = (uint32_t)(blockIdx.x * GS4D_PRE_BLOCK + g);
        }
    }
}

// ------------------------------------------------------------------ key emission with direct placement (GS4D_SORT=direct)
// Same emission as emit_keys_kernel (keys / values in emission order are still written: gs4d_state_view_get shows them),
// and every instance also goes straight into its tile's list: [tile_base[t], tile_base[t + 1]) of keys_tile, laid out
// by the scan launch from the counts of the preprocess launch, as ONE 8-byte word (depth bits low, Gaussian index high:
// the tile is the list itself; a second scattered store for the index cost 12 MB of partial-sector writes at C2 --
// segment_depth_sort_kernel<true> reads this form).  A list is filled in arbitrary
// order -- the per-tile sort that follows orders it by depth, then Gaussian index.  Positions are handed out in two
// levels (tile_place.h): the workgroup counts its Gaussians per tile of its rectangle in LDS, takes one run of positions
// per tile from the tile's cursor (one device-scope add per tile it touches) and leaves the run's start in the tile's LDS
// cell; an instance then takes the next position of its cell (ds_add_rtn).
__global__ __launch_bounds__(GS4D_PRE_BLOCK) void emit_place_kernel(
    int P, int gx, int gy, const float2* __restrict__ means2D, const float* __restrict__ depths,
    const int* __restrict__ radii, const uint32_t* __restrict__ tiles_touched,
    const uint32_t* __restrict__ block_offsets, uint32_t* __restrict__ point_offsets,
    uint64_t* __restrict__ keys, uint32_t* __restrict__ values, uint2* __restrict__ ranges, int ntiles,
    uint2* __restrict__ seg_owner, int seg_slots, uint32_t cap, const uint32_t* __restrict__ R_dev,
    uint32_t* __restrict__ tile_cursor, const uint32_t* __restrict__ tile_base, uint64_t* __restrict__ keys_tile)
{
    // speculative launch with more instances than it was sized for: the tiles stay empty, nothing is emitted (the cursors
    // stay where the scan launch put them: the re-issued launch finds them)
    const bool overflow = R_dev && resolve_R(cap, R_dev) == GS4D_R_OVERFLOW;
    for (int t = blockIdx.x * GS4D_PRE_BLOCK + threadIdx.x; t < ntiles; t += gridDim.x * GS4D_PRE_BLOCK)
    {
        const uint32_t b0 = tile_base[t], b1 = tile_base[t + 1];
        ranges[t] = (overflow || b0 == b1) ? make_uint2(0u, 0u) : make_uint2(b0, b1);   // an empty tile is (0, 0) (rasterizer_impl.cu:311)
    }
    for (int t = blockIdx.x * GS4D_PRE_BLOCK + threadIdx.x; t < seg_slots; t += gridDim.x * GS4D_PRE_BLOCK)
        seg_owner[t] = make_uint2(~0u, 0u);
    if (overflow) return;
    __shared__ uint32_t s_incl[GS4D_PRE_BLOCK];
    __shared__ uint32_t s_wave[GS4D_PRE_BLOCK / 64];
    __shared__ int s_x0[GS4D_PRE_BLOCK], s_y0[GS4D_PRE_BLOCK], s_w[GS4D_PRE_BLOCK];
    __shared__ uint32_t s_depth[GS4D_PRE_BLOCK];
    __shared__ __align__(16) uint32_t s_bins[GS4D_AGG_BINS];
    __shared__ int s_red[GS4D_PRE_BLOCK / 64][6];
    __shared__ uint8_t s_big[GS4D_PRE_BLOCK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = blockIdx.x * GS4D_PRE_BLOCK + tid;
    uint32_t cnt = 0;
    int x0 = 0, y0 = 0, x1 = 1, y1 = 1;
    {
        const int ld = min(idx, P - 1);
        const uint32_t c_in = tiles_touched[ld];
        const float2 m = means2D[ld];
        const int r_in = radii[ld];
        const float d_in = depths[ld];
        if (idx < P) {
            cnt = c_in;
            if (cnt) {
                tile_rect(m.x, m.y, r_in, gx, gy, x0, y0, x1, y1);
                s_x0[tid] = x0; s_y0[tid] = y0; s_w[tid] = x1 - x0;
                s_depth[tid] = __float_as_uint(d_in);
            }
        }
    }
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    block_tile_clear(s_bins);                                // in front of the barrier inside block_tile_bbox()
    // the same split as in the preprocess launch: big splats apart, the others inside the workgroup's rectangle
    const bool big = cnt != 0u && tile_rect_is_big(x0, y0, x1, y1);
    const bool small = cnt != 0u && !big;
    int bx0, by0, bw, bh;
    uint32_t unused_a = 0, unused_b = 0;
    const bool any_small = block_tile_bbox(small, x0, y0, x1, y1, s_red, bx0, by0, bw, bh, unused_a, unused_b);   // barrier inside
    uint32_t wave_prefix = 0;
    for (int w = 0; w < wave; w++) wave_prefix += s_wave[w];
    incl += wave_prefix;
    s_incl[tid] = incl;
    const uint32_t base = block_offsets[blockIdx.x];
    if (idx < P) point_offsets[idx] = base + incl;
    const bool agg = any_small && bw < GS4D_AGG_SIDE && bh < GS4D_AGG_SIDE;   // uniform
    s_big[tid] = (uint8_t)(agg ? big : true);                // 1: the Gaussian's instances take their position from the tile's cursor itself
    const int pitch = bw + 1;
    if (agg) {
        block_tile_counts(small, x0, y0, x1, y1, bx0, by0, bw, bh, s_bins);                    // ends with a barrier
        const int cells = pitch * bh;
        for (int b = tid; b < cells; b += GS4D_PRE_BLOCK) {
            const uint32_t c = s_bins[b];
            if (c) s_bins[b] = atomicAdd(&tile_cursor[(size_t)((by0 + b / pitch) * gx + bx0 + b % pitch) * GS4D_TILE_CURSOR_STRIDE], c);
        }
    }
    __syncthreads();                                         // s_incl, s_big and the cells' run starts

    const uint32_t total = s_incl[GS4D_PRE_BLOCK - 1];
    for (uint32_t s0 = tid; s0 < total; s0 += 2 * GS4D_PRE_BLOCK) {
        const uint32_t sv[2] = {s0, min(s0 + (uint32_t)GS4D_PRE_BLOCK, total - 1u)};
        const bool second = s0 + (uint32_t)GS4D_PRE_BLOCK < total;
        const int g[2] = {slot_owner(s_incl, sv[0]), slot_owner(s_incl, sv[1])};   // two searches side by side
        uint64_t key[2];
        uint32_t val[2], at[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t k = sv[q] - (g[q] ? s_incl[g[q] - 1] : 0u);
            const int w = s_w[g[q]];
            const int ty = s_y0[g[q]] + (int)(k / (uint32_t)w), tx = s_x0[g[q]] + (int)(k % (uint32_t)w);
            const uint32_t tile = (uint32_t)(ty * gx + tx);
            key[q] = ((uint64_t)tile << 32) | (uint64_t)s_depth[g[q]];
            val[q] = (uint32_t)(blockIdx.x * GS4D_PRE_BLOCK + g[q]);
            at[q] = 0u;
            if (q == 0 || second)
                at[q] = s_big[g[q]] ? atomicAdd(&tile_cursor[(size_t)tile * GS4D_TILE_CURSOR_STRIDE], 1u)
                                    : atomicAdd(&s_bins[(ty - by0) * pitch + (tx - bx0)], 1u);
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (q == 1 && !second) break;
            keys[base + sv[q]] = key[q];
            values[base + sv[q]] = val[q];
            keys_tile[at[q]] = ((uint64_t)val[q] << 32) | (key[q] & 0xffffffffull);   // (depth bits, Gaussian): ONE scattered store
        }
    }
}

// ------------------------------------------------------------------ radix sort, one 8-bit digit per pass
// Pass structure (all kernels on one stream, no inter-workgroup communication inside a launch):
//   sort_hist_kernel    : per-tile 256-bin digit histogram -> hist[digit][tile]
//   sort_scan_kernel    : workgroup d scans row d of hist in place (exclusive) and writes the row total
//   sort_scatter_kernel : re-reads its tile, ranks keys stably (wave-level match + per-wave LDS
//                         counters), stages the tile sorted by digit in LDS and writes runs out.
// A tile is 2048 keys: wave w owns keys [512w, 512w+512), lane l handles key 512w + 64r + l in
// round r, so (wave, round, lane) order == memory order, which is what makes the sort stable.

__global__ __launch_bounds__(GS4D_SORT_THREADS) void sort_hist_kernel(uint32_t R_host, int shift,
                                                                      const uint64_t* __restrict__ keys,
                                                                      uint32_t* __restrict__ hist, uint32_t ntiles,
                                                                      const uint32_t* __restrict__ R_dev)
{
    const uint32_t R = resolve_R(R_host, R_dev);
    if (R == GS4D_R_OVERFLOW || R == 0u) return;           // (speculative launches only: the host never launches with R = 0)
    __shared__ uint32_t s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * GS4D_SORT_TILE;
    uint32_t d[GS4D_SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < GS4D_SORT_ITEMS; i++)          // unconditional, clamped: all loads in flight together
        d[i] = (uint32_t)(keys[min(base + i * GS4D_SORT_THREADS + threadIdx.x, R - 1u)] >> shift) & 255u;
#pragma unroll
    for (int i = 0; i < GS4D_SORT_ITEMS; i++)
        if (base + i * GS4D_SORT_THREADS + threadIdx.x < R) atomicAdd(&s_h[d[i]], 1u);
    __syncthreads();
    hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = s_h[threadIdx.x];
}

__global__ __launch_bounds__(256) void sort_scan_kernel(uint32_t ntiles, uint32_t* __restrict__ hist,
                                                        uint32_t* __restrict__ digit_totals)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    uint32_t* row = hist + (size_t)blockIdx.x * ntiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t b = 0; b < ntiles; b += 256) {
        const uint32_t i = b + threadIdx.x;
        const uint32_t v = i < ntiles ? row[i] : 0u;
        uint32_t incl = v;
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
        if (i < ntiles) row[i] = carry + wp + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = carry + wp + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) digit_totals[blockIdx.x] = s_carry;
}

__global__ __launch_bounds__(GS4D_SORT_THREADS) void sort_scatter_kernel(
    uint32_t R_host, int shift, const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ hist,
    const uint32_t* __restrict__ digit_totals, uint32_t ntiles, const uint32_t* __restrict__ R_dev)
{
    const uint32_t R = resolve_R(R_host, R_dev);
    if (R == GS4D_R_OVERFLOW || blockIdx.x * GS4D_SORT_TILE >= R) return;   // uniform: a tile beyond the last instance
    __shared__ uint32_t s_cnt[4][256];        // per-wave digit counts -> exclusive offsets inside the tile
    __shared__ uint32_t s_digit_start[256];   // first staged slot of each digit inside the tile
    __shared__ uint32_t s_global[256];        // global output position of that first slot
    __shared__ uint32_t s_wave[4];
    __shared__ uint64_t s_keys[GS4D_SORT_TILE];
    __shared__ uint32_t s_vals[GS4D_SORT_TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * 256; i += GS4D_SORT_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();

    const uint32_t tile_base = blockIdx.x * GS4D_SORT_TILE;
    const uint32_t wave_base = tile_base + wave * (64 * GS4D_SORT_ITEMS);
    uint64_t key[GS4D_SORT_ITEMS];
    uint32_t val[GS4D_SORT_ITEMS];
    uint32_t rank[GS4D_SORT_ITEMS];   // rank among equal digits inside this wave's keys
    const uint64_t lanemask_lt = (1ull << lane) - 1ull;
    // loads unconditional with clamped addresses (a load under a branch is waited for before the next one is issued:
    // the eight key / value pairs of a lane then cost eight memory latencies in a row)
#pragma unroll
    for (int r = 0; r < GS4D_SORT_ITEMS; r++) {
        const uint32_t k = min(wave_base + r * 64 + lane, R - 1u);
        key[r] = keys_in[k];
        val[r] = vals_in[k];
    }
    // ranking as in the per-tile sort: the register-only part of every round first, then ONE chain through the LDS
    // counters, then the shuffles
    uint32_t before[GS4D_SORT_ITEMS], group[GS4D_SORT_ITEMS];
    int leader[GS4D_SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < GS4D_SORT_ITEMS; r++) {
        const bool valid = wave_base + r * 64 + lane < R;
        if (!valid) key[r] = ~0ull;
        const uint64_t same = match_digit((uint32_t)(key[r] >> shift) & 255u, valid);   // lanes holding the same digit
        before[r] = (uint32_t)__popcll(same & lanemask_lt);
        leader[r] = __ffsll((unsigned long long)same) - 1;
        group[r] = (uint32_t)__popcll(same);
    }
#pragma unroll
    for (int r = 0; r < GS4D_SORT_ITEMS; r++) {
        rank[r] = 0u;
        if (wave_base + r * 64 + lane < R && lane == leader[r]) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
            rank[r] = s_cnt[wave][d];
            s_cnt[wave][d] = rank[r] + group[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next round's leaders read what this one's wrote
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int r = 0; r < GS4D_SORT_ITEMS; r++) rank[r] = (uint32_t)__shfl((int)rank[r], leader[r], 64) + before[r];
    __syncthreads();

    // per digit: exclusive offsets across the 4 waves, tile total, and exclusive scan over digits
    {
        const uint32_t c0 = s_cnt[0][tid], c1 = s_cnt[1][tid], c2 = s_cnt[2][tid], c3 = s_cnt[3][tid];
        s_cnt[0][tid] = 0; s_cnt[1][tid] = c0; s_cnt[2][tid] = c0 + c1; s_cnt[3][tid] = c0 + c1 + c2;
        const uint32_t total = c0 + c1 + c2 + c3;
        uint32_t incl = total;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wp = 0;
        for (int w = 0; w < wave; w++) wp += s_wave[w];
        s_digit_start[tid] = wp + incl - total;
        // global position = (keys with a smaller digit anywhere) + (same digit in earlier tiles)
        uint32_t smaller = 0;
        // exclusive scan of the 256 digit totals, recomputed per workgroup (256 adds per lane worst case
        // would be wasteful; do it with the same wave-scan)
        const uint32_t dt = digit_totals[tid];
        uint32_t dincl = dt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(dincl, off, 64);
            if (lane >= off) dincl += t;
        }
        __syncthreads();
        if (lane == 63) s_wave[wave] = dincl;
        __syncthreads();
        for (int w = 0; w < wave; w++) smaller += s_wave[w];
        smaller += dincl - dt;
        s_global[tid] = smaller + hist[(size_t)tid * ntiles + blockIdx.x];
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < GS4D_SORT_ITEMS; r++) {
        const uint32_t k = wave_base + r * 64 + lane;
        if (k < R) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
            const uint32_t slot = s_digit_start[d] + s_cnt[wave][d] + rank[r];
            s_keys[slot] = key[r];
            s_vals[slot] = val[r];
        }
    }
    __syncthreads();

    const uint32_t n_valid = min((uint32_t)GS4D_SORT_TILE, R - tile_base);
    for (uint32_t i = tid; i < n_valid; i += GS4D_SORT_THREADS) {
        const uint64_t kk = s_keys[i];
        const uint32_t d = (uint32_t)(kk >> shift) & 255u;
        const uint32_t dst = s_global[d] + (i - s_digit_start[d]);
        keys_out[dst] = kk;
        vals_out[dst] = s_vals[i];
    }
}

// ------------------------------------------------------------------ tile-first sort: per-tile depth sort in LDS
// The LSD sort above moves every key through HBM six times (4 depth + 2 tile digits at C1-C4).  Tile-first order does
// the same job with three trips: (1) the two stable passes on the TILE bits group the keys by tile, keeping emission
// order (= ascending Gaussian index) inside a tile; (2) tile_bounds_kernel finds each tile's [begin, end); (3) one
// workgroup per tile sorts its list by the 32 depth bits -- four stable 8-bit passes, ranked exactly like
// sort_scatter_kernel (per-wave contiguous ownership, wave match-ballot, per-wave digit counters) -- with the list held
// in LDS between passes, and writes it out once.  Stable on both levels, hence the same permutation as
// the LSD sort, bit for bit.  Lists longer than SEG_CAP (rare: > 2048 Gaussians on one 16x16 tile) take the same
// passes chunk by chunk through the two global buffers (slow, correct).
#define SEG_CAP 2048

__global__ __launch_bounds__(256) void tile_bounds_kernel(uint32_t R_host, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ R_dev)
{
    const uint32_t R = resolve_R(R_host, R_dev);
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (R == GS4D_R_OVERFLOW || idx >= R) return;
    const uint32_t tile = (uint32_t)(keys[idx] >> 32);
    if (idx == 0)
        ranges[tile].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (tile != prev) {
            ranges[prev].y = idx;
            ranges[tile].x = idx;
        }
    }
    if (idx == R - 1) ranges[tile].y = R;
}

// The list lives in LDS between passes (two key / value buffers + the ranks); loops over rounds are dynamic, so the
// kernel needs few registers and occupancy is set by its 46 KB of LDS (3 workgroups per CU).
struct SegShared {
    uint32_t cnt[4][256];        // per-wave digit counts -> exclusive offsets of the wave inside the chunk's digit run
    uint32_t digit_start[256];   // first slot of each digit inside the chunk (fast path) / running global base (long lists)
    uint32_t chunk_total[256];
    uint32_t wave[4];
    uint32_t k[1][SEG_CAP];
    uint32_t v[1][SEG_CAP];
    uint16_t rank[SEG_CAP];
};

// Ranks the n items of key[] (item (wave, round, lane) = position wave * 64 * rounds + round * 64 + lane):
// on return S.rank[pos] = number of earlier items of the SAME WAVE with the same digit, S.cnt[w][d] = exclusive offset
// of wave w among the chunk's items with digit d, S.chunk_total[d] = their number, and, if scan_digits,
// S.digit_start[d] = exclusive scan of chunk_total.
__device__ __forceinline__ void seg_rank_chunk(SegShared& S, const uint32_t* key, int rounds, int n, int shift, bool scan_digits)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lanemask_lt = (1ull << lane) - 1ull;
    for (int i = tid; i < 4 * 256; i += 256) (&S.cnt[0][0])[i] = 0;
    __syncthreads();
    for (int r = 0; r < rounds; r++) {
        const int pos = wave * 64 * rounds + r * 64 + lane;
        const bool valid = pos < n;
        const uint32_t d = valid ? (key[pos] >> shift) & 255u : 0u;
        const uint64_t same = match_digit(d, valid);
        const uint32_t before = (uint32_t)__popcll(same & lanemask_lt);
        const int leader = __ffsll((unsigned long long)same) - 1;
        uint32_t prev = 0;
        if (valid && lane == leader) {
            prev = S.cnt[wave][d];
            S.cnt[wave][d] = prev + (uint32_t)__popcll(same);
        }
        prev = __shfl(prev, leader, 64);
        if (valid) S.rank[pos] = (uint16_t)(prev + before);
    }
    __syncthreads();
    {
        const uint32_t c0 = S.cnt[0][tid], c1 = S.cnt[1][tid], c2 = S.cnt[2][tid], c3 = S.cnt[3][tid];
        S.cnt[0][tid] = 0; S.cnt[1][tid] = c0; S.cnt[2][tid] = c0 + c1; S.cnt[3][tid] = c0 + c1 + c2;
        const uint32_t total = c0 + c1 + c2 + c3;
        S.chunk_total[tid] = total;
        if (scan_digits) {
            uint32_t incl = total;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            if (lane == 63) S.wave[wave] = incl;
            __syncthreads();
            uint32_t wp = 0;
            for (int w = 0; w < wave; w++) wp += S.wave[w];
            S.digit_start[tid] = wp + incl - total;
        }
    }
    __syncthreads();
}

// Short lists -- most of them: a tile's list averages a few hundred entries -- are sorted by ONE wave each, four lists per
// workgroup, with no workgroup barrier at all: the workgroup form above pays ~7 barriers per pass whatever the length of the
// list, and its 43 KB of LDS per LIST kept only 3 lists per compute unit in flight.  Same passes and the same match-ballot
// ranking (~1.2 instructions per key and pass; two other formulations are recorded in DESIGN.md: digit histograms by LDS
// atomics ran into the ~0.3 lanes per clock of ds_add, and lane-owned blocks with 4-bit digits and DPP scans needed 3.5x
// the instructions).  The wave keeps its keys, values and ranks in REGISTERS (entry r * 64 + lane in slot r, R = 2 / 4 / 8
// slots), so LDS holds one copy of the list and the 256 digit counters: 5 KB per list of up to 512 entries.  A pass
// whose digit is the same for every key (the sign / exponent byte of one tile's depths, nearly always) is skipped.
#define WSEG_CAP 512               // entries a single wave sorts on its own
struct WaveSeg {
    uint32_t k[WSEG_CAP];
    uint32_t v[WSEG_CAP];
    uint32_t cnt[256];
};
union SegUnion {
    SegShared wg;
    WaveSeg wv[4];
};

// LDS traffic of one wave is executed in program order; this only keeps the compiler from moving it across
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// begin, n (1 <= n <= 64 R) and tile are wave-uniform.  Loads are unconditional with clamped addresses (a load under a
// branch makes the compiler wait for every load before it issues the next one).
// Position of the list entry (key kp, Gaussian vp) found at `pos` of the depth-sorted list sk / sv (n entries) once every
// run of EQUAL depths is ordered by Gaussian index: the stable sort of the reference leaves such a run in emission order
// (ascending index); the direct placement (GS4D_SORT=direct) hands the lists over in arbitrary order, so the order inside
// a run is restored here.  A Gaussian is at most once in a tile's list: the indices of a run are distinct.
template <typename K>
__device__ __forceinline__ uint32_t tie_position(const K* sk, const uint32_t* sv, int n, int pos, uint32_t kp, uint32_t vp)
{
    const bool tie = (pos > 0 && (uint32_t)sk[pos - 1] == kp) || (pos + 1 < n && (uint32_t)sk[pos + 1] == kp);
    if (!tie) return (uint32_t)pos;
    int first = pos;
    while (first > 0 && (uint32_t)sk[first - 1] == kp) first--;
    uint32_t smaller = 0;
    for (int q = first; q < n && (uint32_t)sk[q] == kp; q++) smaller += sv[q] < vp ? 1u : 0u;
    return (uint32_t)first + smaller;
}

template <int R, bool TIES>
__device__ __forceinline__ void wave_sort_list(WaveSeg& S, uint32_t begin, int n, uint32_t tile, const uint64_t* __restrict__ keys_a,
                                               const uint32_t* __restrict__ vals_a, uint64_t* __restrict__ keys_b,
                                               uint32_t* __restrict__ vals_b)
{
    const int lane = threadIdx.x & 63;
    const uint64_t lanemask_lt = (1ull << lane) - 1ull;
    uint32_t k[R], v[R], rank[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int pos = min(r * 64 + lane, n - 1);
        if (TIES) {                                           // direct placement's lists: (depth bits, Gaussian) in one word
            const uint64_t kv = keys_a[begin + pos];
            k[r] = (uint32_t)kv;
            v[r] = (uint32_t)(kv >> 32);
        } else {
            k[r] = (uint32_t)keys_a[begin + pos];
            v[r] = vals_a[begin + pos];
        }
    }
    const uint32_t key0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k[0]);
    uint32_t diff = 0;
#pragma unroll
    for (int r = 0; r < R; r++) diff |= k[r] ^ key0;          // clamped duplicates of the last entry change nothing
    uint32_t live = 0;                                        // bit p: byte p differs somewhere in the list (wave-uniform)
#pragma unroll
    for (int p = 0; p < 4; p++)
        if (__ballot(((diff >> (8 * p)) & 255u) != 0u) != 0ull) live |= 1u << p;
    uint4* c4 = reinterpret_cast<uint4*>(S.cnt);
    for (int pass = 0; pass < 4; pass++) {
        if (!((live >> pass) & 1u)) continue;                 // one digit for every key: the pass is the identity
        const int shift = 8 * pass;
        wave_sync();
        c4[lane] = make_uint4(0u, 0u, 0u, 0u);
        wave_sync();
        // ranking in two sweeps: first the register-only part of every round (the nine ballots and what follows from
        // them), then the chain through the LDS counters -- one read-modify-write per round, issued back to back -- and
        // last the shuffles; with everything inside one loop a round cost three LDS round trips in sequence
        uint32_t before[R], group[R];
        int leader[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool valid = r * 64 + lane < n;
            const uint64_t same = match_digit((k[r] >> shift) & 255u, valid);
            before[r] = (uint32_t)__popcll(same & lanemask_lt);
            leader[r] = __ffsll((unsigned long long)same) - 1;
            group[r] = (uint32_t)__popcll(same);
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            rank[r] = 0u;
            if (r * 64 + lane < n && lane == leader[r]) {
                const uint32_t d = (k[r] >> shift) & 255u;
                rank[r] = S.cnt[d];
                S.cnt[d] = rank[r] + group[r];
            }
            wave_sync();                                      // the next round's leaders read what this one's wrote
        }
#pragma unroll
        for (int r = 0; r < R; r++) rank[r] = (uint32_t)__shfl((int)rank[r], leader[r], 64) + before[r];
        {
            const uint4 c = c4[lane];                         // digits 4 lane .. 4 lane + 3
            const uint32_t total = c.x + c.y + c.z + c.w;
            const uint32_t excl = wave_inclusive_sum(total) - total;
            c4[lane] = make_uint4(excl, excl + c.x, excl + c.x + c.y, excl + c.x + c.y + c.z);
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < R; r++)
            if (r * 64 + lane < n) {
                const uint32_t slot = S.cnt[(k[r] >> shift) & 255u] + rank[r];
                S.k[slot] = k[r];
                S.v[slot] = v[r];
            }
        wave_sync();
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int pos = min(r * 64 + lane, n - 1);
            k[r] = S.k[pos];
            v[r] = S.v[pos];
        }
    }
    const uint64_t hi = (uint64_t)tile << 32;
    if (TIES) {
        // does ANY entry of the list equal its predecessor?  (entry pos - 1 is the previous lane's slot r: one DPP move,
        // lane 0 takes lane 63's slot r - 1.)  A handful of lists per frame do; the others leave through the plain stores.
        bool tie = false;
#pragma unroll
        for (int r = 0; r < R; r++) {
            uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k[r], 0x138, 0xf, 0xf, false);   // wave_shr:1
            if (r > 0) {
                const uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)k[r - 1], 63);
                if (lane == 0) prev = carry;
            }
            const int pos = r * 64 + lane;
            tie |= pos > 0 && pos < n && prev == k[r];
        }
        if (__ballot(tie) != 0ull) {
            if (!live) {                                      // no pass ran (every depth the same): the list is not in LDS yet
                wave_sync();
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (r * 64 + lane < n) { S.k[r * 64 + lane] = k[r]; S.v[r * 64 + lane] = v[r]; }
                wave_sync();
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int pos = r * 64 + lane;
                if (pos < n) {
                    const uint32_t at = tie_position(S.k, S.v, n, pos, k[r], v[r]);
                    keys_b[begin + at] = hi | (uint64_t)k[r];
                    vals_b[begin + at] = v[r];
                }
            }
            return;
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int pos = r * 64 + lane;
        if (pos < n) {
            keys_b[begin + pos] = hi | (uint64_t)k[r];
            vals_b[begin + pos] = v[r];
        }
    }
}

// Workgroup form for lists of up to SEG_CAP entries: the same register-resident scheme spread over four waves.  Wave w owns
// the entries [w 64 R, (w + 1) 64 R) (R = 4 or 8 rounds of 64), ranks them against its own 256 counters (wave-private: no
// barrier), then the counters of the four waves are combined per digit and scanned (thread = digit), and every entry goes to
// digit_start[d] + offset of its wave + rank.  Four barriers per pass.
template <int R, bool TIES>
__device__ __forceinline__ void wg_sort_list(SegShared& S, uint32_t begin, int n, uint32_t tile, const uint64_t* __restrict__ keys_a,
                                             const uint32_t* __restrict__ vals_a, uint64_t* __restrict__ keys_b,
                                             uint32_t* __restrict__ vals_b)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lanemask_lt = (1ull << lane) - 1ull;
    const int base = wave * 64 * R;
    uint32_t* sk = S.k[0];
    uint32_t* sv = S.v[0];
    uint32_t k[R], v[R], rank[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int pos = min(base + r * 64 + lane, n - 1);
        if (TIES) {
            const uint64_t kv = keys_a[begin + pos];
            k[r] = (uint32_t)kv;
            v[r] = (uint32_t)(kv >> 32);
        } else {
            k[r] = (uint32_t)keys_a[begin + pos];
            v[r] = vals_a[begin + pos];
        }
    }
    const uint32_t key0 = (uint32_t)keys_a[begin];
    uint32_t diff = 0;
#pragma unroll
    for (int r = 0; r < R; r++) diff |= k[r] ^ key0;          // clamped duplicates of the last entry change nothing
    uint32_t live = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) live |= __syncthreads_or((int)((diff >> (8 * p)) & 255u)) ? 1u << p : 0u;
    uint4* c4 = reinterpret_cast<uint4*>(S.cnt[wave]);
    for (int pass = 0; pass < 4; pass++) {
        if (!((live >> pass) & 1u)) continue;                 // one digit for every key: the pass is the identity
        const int shift = 8 * pass;
        c4[lane] = make_uint4(0u, 0u, 0u, 0u);                // the wave's own counters (read by all before the last barrier)
        uint32_t before[R], group[R];
        int leader[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool valid = base + r * 64 + lane < n;
            const uint64_t same = match_digit((k[r] >> shift) & 255u, valid);
            before[r] = (uint32_t)__popcll(same & lanemask_lt);
            leader[r] = __ffsll((unsigned long long)same) - 1;
            group[r] = (uint32_t)__popcll(same);
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < R; r++) {
            rank[r] = 0u;
            if (base + r * 64 + lane < n && lane == leader[r]) {
                const uint32_t d = (k[r] >> shift) & 255u;
                rank[r] = S.cnt[wave][d];
                S.cnt[wave][d] = rank[r] + group[r];
            }
            wave_sync();                                      // the next round's leaders read what this one's wrote
        }
#pragma unroll
        for (int r = 0; r < R; r++) rank[r] = (uint32_t)__shfl((int)rank[r], leader[r], 64) + before[r];
        __syncthreads();
        {
            const uint32_t c0 = S.cnt[0][tid], c1 = S.cnt[1][tid], c2 = S.cnt[2][tid], c3 = S.cnt[3][tid];
            S.cnt[0][tid] = 0; S.cnt[1][tid] = c0; S.cnt[2][tid] = c0 + c1; S.cnt[3][tid] = c0 + c1 + c2;
            const uint32_t total = c0 + c1 + c2 + c3;
            const uint32_t incl = wave_inclusive_sum(total);
            if (lane == 63) S.wave[wave] = incl;
            __syncthreads();
            uint32_t wp = 0;
            for (int w = 0; w < wave; w++) wp += S.wave[w];
            S.digit_start[tid] = wp + incl - total;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++)
            if (base + r * 64 + lane < n) {
                const uint32_t d = (k[r] >> shift) & 255u;
                const uint32_t slot = S.digit_start[d] + S.cnt[wave][d] + rank[r];
                sk[slot] = k[r];
                sv[slot] = v[r];
            }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int pos = min(base + r * 64 + lane, n - 1);
            k[r] = sk[pos];
            v[r] = sv[pos];
        }
    }
    const uint64_t hi = (uint64_t)tile << 32;
    if (TIES) {                                               // runs of equal depths: by Gaussian index (tie_position)
        if (!live) {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (base + r * 64 + lane < n) { sk[base + r * 64 + lane] = k[r]; sv[base + r * 64 + lane] = v[r]; }
            __syncthreads();
        }
        int tie = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int pos = base + r * 64 + lane;
            tie |= pos > 0 && pos < n && sk[pos - 1] == k[r];
        }
        if (__syncthreads_or(tie)) {                          // rare; the other lists leave through the plain stores
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int pos = base + r * 64 + lane;
                if (pos < n) {
                    const uint32_t at = tie_position(sk, sv, n, pos, k[r], v[r]);
                    keys_b[begin + at] = hi | (uint64_t)k[r];
                    vals_b[begin + at] = v[r];
                }
            }
            return;
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int pos = base + r * 64 + lane;
        if (pos < n) {
            keys_b[begin + pos] = hi | (uint64_t)k[r];
            vals_b[begin + pos] = v[r];
        }
    }
}

// Grid: blocks [0, ntiles) take the lists LONGER than wave_cap, one workgroup per list (they start first: they are the
// long poles); blocks [ntiles, ntiles + ceil(ntiles / 4)) take the lists of at most wave_cap entries, one wave per list.
template <bool TIES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void segment_depth_sort_kernel(const uint2* __restrict__ ranges,
                                                                 uint64_t* __restrict__ keys_a, uint32_t* __restrict__ vals_a,
                                                                 uint64_t* __restrict__ keys_b, uint32_t* __restrict__ vals_b,
                                                                 uint32_t cap, uint32_t wave_cap, uint32_t ntiles)
{
    __shared__ SegUnion U;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef GS4D_TIMELINE
    gs4d::TimelineScope tl(g_timeline, (int)blockIdx.x);
#endif
    if (blockIdx.x >= ntiles) {
        const uint32_t t = (blockIdx.x - ntiles) * 4u + (uint32_t)wave;
        if (t >= ntiles) return;
        const uint2 range = ranges[t];
        const uint32_t begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)range.x);
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(range.y - range.x));
#ifdef GS4D_TIMELINE
        tl.n = n; tl.visits = 1;
#endif
        if (n == 0u || n > wave_cap) return;
        WaveSeg& Sw = U.wv[wave];
        if (n <= 128u) wave_sort_list<2, TIES>(Sw, begin, (int)n, t, keys_a, vals_a, keys_b, vals_b);
        else if (n <= 256u) wave_sort_list<4, TIES>(Sw, begin, (int)n, t, keys_a, vals_a, keys_b, vals_b);
        else wave_sort_list<8, TIES>(Sw, begin, (int)n, t, keys_a, vals_a, keys_b, vals_b);
        return;
    }
    SegShared& S = U.wg;
    const uint32_t tile = blockIdx.x;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
#ifdef GS4D_TIMELINE
    tl.n = (unsigned)max(n, 0);
#endif
    if (n <= 0 || (uint32_t)n <= wave_cap) return;
    const uint64_t hi = (uint64_t)tile << 32;

    if ((uint32_t)n <= cap) {
        // ---- the list fits (cap <= SEG_CAP = 2048)
        if (n <= 1024) wg_sort_list<4, TIES>(S, range.x, n, tile, keys_a, vals_a, keys_b, vals_b);
        else wg_sort_list<8, TIES>(S, range.x, n, tile, keys_a, vals_a, keys_b, vals_b);
        return;
    }

    // ---- long list: the same four passes, chunk by chunk, ping-pong between the two global buffers
    // (a -> b -> a -> b -> a), then a copy a -> b so that the result lands where the fast path puts it
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 8 * pass;
        const uint64_t* kin = ((pass & 1) ? keys_b : keys_a) + range.x;
        const uint32_t* vin = ((pass & 1) ? vals_b : vals_a) + range.x;
        uint64_t* kout = ((pass & 1) ? keys_a : keys_b) + range.x;
        uint32_t* vout = ((pass & 1) ? vals_a : vals_b) + range.x;
        // digit totals of the whole list -> running base of every digit
        S.chunk_total[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 256) atomicAdd(&S.chunk_total[((uint32_t)kin[i] >> shift) & 255u], 1u);
        __syncthreads();
        {
            const uint32_t total = S.chunk_total[tid];
            uint32_t incl = total;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            if (lane == 63) S.wave[wave] = incl;
            __syncthreads();
            uint32_t wp = 0;
            for (int w = 0; w < wave; w++) wp += S.wave[w];
            S.digit_start[tid] = wp + incl - total;
        }
        __syncthreads();
        for (int c0 = 0; c0 < n; c0 += SEG_CAP) {
            const int cn = min(SEG_CAP, n - c0);
            const int rounds = (cn + 255) >> 8;
            for (int i = tid; i < cn; i += 256) {
                S.k[0][i] = (uint32_t)kin[c0 + i];
                S.v[0][i] = (TIES && pass == 0) ? (uint32_t)(kin[c0 + i] >> 32) : vin[c0 + i];   // pass 0 of direct placement: packed lists
            }
            __syncthreads();
            seg_rank_chunk(S, S.k[0], rounds, cn, shift, false);
            for (int r = 0; r < rounds; r++) {
                const int pos = wave * 64 * rounds + r * 64 + lane;
                if (pos < cn) {
                    const uint32_t kk = S.k[0][pos];
                    const uint32_t d = (kk >> shift) & 255u;
                    const uint32_t dst = S.digit_start[d] + S.cnt[wave][d] + S.rank[pos];
                    kout[dst] = hi | (uint64_t)kk;
                    vout[dst] = S.v[0][pos];
                }
            }
            __syncthreads();
            S.digit_start[tid] += S.chunk_total[tid];
            __syncthreads();
        }
        __threadfence();                                 // this workgroup re-reads what it wrote in the next pass
        __syncthreads();
    }
    for (int i = tid; i < n; i += 256) {
        const uint64_t kk = keys_a[range.x + i];
        const uint32_t vv = vals_a[range.x + i];
        const uint32_t at = TIES ? tie_position(keys_a + range.x, vals_a + range.x, n, i, (uint32_t)kk, vv) : (uint32_t)i;
        keys_b[range.x + at] = kk;
        vals_b[range.x + at] = vv;
    }
}

// ------------------------------------------------------------------ tile ranges (six-pass LSD order only)
// ranges[tile] = [first, last+1) in the sorted list (zeroed beforehand by emit_keys_kernel)
__global__ __launch_bounds__(256) void tile_ranges_kernel(uint32_t R_host, const uint64_t* __restrict__ keys_sorted,
                                                          uint2* __restrict__ ranges, const uint32_t* __restrict__ R_dev)
{
    const uint32_t R = resolve_R(R_host, R_dev);
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (R == GS4D_R_OVERFLOW || idx >= R) return;
    const uint32_t tile = (uint32_t)(keys_sorted[idx] >> 32);
    if (idx == 0)
        ranges[tile].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys_sorted[idx - 1] >> 32);
        if (tile != prev) {
            ranges[prev].y = idx;
            ranges[tile].x = idx;
        }
    }
    if (idx == R - 1) ranges[tile].y = R;
}

}  // namespace

// Stable LSD radix sort of (u64 key, u32 value) pairs on key bits [first_shift, first_shift + 8*passes).  The input
// arrays are left intact; the result lands in keys_tmp[*where] / vals_tmp[*where].
// hist must hold 256 * gs4d_sort_blocks(R) + 256 words.
int gs4d_radix_sort_pairs(uint32_t R, int passes, int first_shift, const uint64_t* keys_in, const uint32_t* vals_in,
                          uint64_t* const keys_tmp[2], uint32_t* const vals_tmp[2], uint32_t* hist, int* where,
                          int debug, hipStream_t s, const uint32_t* R_dev)
{
    const uint32_t ntiles = gs4d_sort_blocks((size_t)R);
    uint32_t* digit_totals = hist + (size_t)256 * ntiles;
    const uint64_t* kin = keys_in;
    const uint32_t* vin = vals_in;
    *where = 0;
    for (int p = 0; p < passes; p++) {
        uint64_t* kout = keys_tmp[p & 1];
        uint32_t* vout = vals_tmp[p & 1];
        hipLaunchKernelGGL(sort_hist_kernel, dim3(ntiles), dim3(GS4D_SORT_THREADS), 0, s, R, first_shift + 8 * p, kin, hist, ntiles, R_dev);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(256), dim3(256), 0, s, ntiles, hist, digit_totals);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(ntiles), dim3(GS4D_SORT_THREADS), 0, s, R, first_shift + 8 * p, kin, vin, kout,
                           vout, hist, digit_totals, ntiles, R_dev);
        GS4D_LAUNCH_CHECK("radix_sort_pass", debug, s);
        kin = kout;
        vin = vout;
        *where = p & 1;
    }
    return GS4D_OK;
}

int gs4d_launch_scan(int P, GeomState& g, unsigned long long* mailbox, uint32_t seq, int debug, hipStream_t s,
                     uint32_t* tile_cursor, uint32_t* tile_base, int ntiles)   // tile_cursor != null: direct placement
{
    GS4D_PROF("scan", s);
    const int nblk = (P + GS4D_PRE_BLOCK - 1) / GS4D_PRE_BLOCK;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(tile_cursor ? 2 : 1), dim3(1024), 0, s, nblk, g.block_sums, g.block_offsets,
                       g.counters, mailbox, seq, g.block_area, tile_cursor, tile_base, ntiles);
    GS4D_LAUNCH_CHECK("scan_block_sums", debug, s);
    return GS4D_OK;
}

// R_dev == nullptr: R is num_rendered.  Otherwise the speculative form: R is the capacity the binning chunk was laid out
// for, the kernels take the count from *R_dev (GeomState::counters) and do nothing if it exceeds R.
int gs4d_launch_binning(int P, int R, int W, int H, const int* radii, GeomState& g, BinningState& b, ImageState& im,
                        int debug, hipStream_t s, const uint32_t* R_dev, bool direct)
{
    const int gx = (W + GS4D_TILE_X - 1) / GS4D_TILE_X, gy = (H + GS4D_TILE_Y - 1) / GS4D_TILE_Y;
    const int nblk = (P + GS4D_PRE_BLOCK - 1) / GS4D_PRE_BLOCK;
    const int half_in = b.sorted ^ 1;                       // direct placement fills this half, the per-tile sort the other
    {
    GS4D_PROF("emit_keys", s);
    if (direct)
        hipLaunchKernelGGL(emit_place_kernel, dim3(nblk), dim3(GS4D_PRE_BLOCK), 0, s, P, gx, gy, g.means2D, g.depths,
                           radii, g.tiles_touched, g.block_offsets, g.point_offsets, b.keys_unsorted, b.values_unsorted,
                           im.ranges, gx * gy, b.seg_owner, b.seg_owner ? (int)b.seg_slots : 0, (uint32_t)R, R_dev,
                           im.tile_cursor, im.tile_base, b.keys_tmp[half_in]);
    else
    hipLaunchKernelGGL(emit_keys_kernel, dim3(nblk), dim3(GS4D_PRE_BLOCK), 0, s, P, gx, gy, g.means2D, g.depths,
                       radii, g.tiles_touched, g.block_offsets, g.point_offsets, b.keys_unsorted, b.values_unsorted,
                       im.ranges, gx * gy, b.seg_owner, b.seg_owner ? (int)b.seg_slots : 0,    // no slots on a forward-only call
                       (uint32_t)R, R_dev);
    GS4D_LAUNCH_CHECK("emit_keys", debug, s);
    }

    if (R <= 0) return GS4D_OK;

    const uint32_t tbits = gs4d_higher_msb((uint32_t)(gx * gy));
    int where = 0;
    int rc;
    if (direct) {
        // The tile lists are in place, in arbitrary order inside a list: one launch sorts each by depth, then Gaussian
        // index (the order the reference's stable sort leaves), into the half the radix passes would have ended in (the
        // backward and the state view find the lists without knowing how they were binned).  A speculative launch that
        // overflowed left every range empty.
        GS4D_PROF("radix_sort", s);
        const uint32_t nt = (uint32_t)(gx * gy), cap = gs4d_sort_segment_cap();
        const uint32_t wave_cap = cap < gs4d_sort_wave_cap() ? cap : gs4d_sort_wave_cap();
        hipLaunchKernelGGL(segment_depth_sort_kernel<true>, dim3(nt + (nt + 3) / 4), dim3(256), 0, s, im.ranges, b.keys_tmp[half_in],
                           b.values_tmp[half_in], b.keys_tmp[b.sorted], b.values_tmp[b.sorted], cap, wave_cap, nt);
        GS4D_LAUNCH_CHECK("segment_depth_sort", debug, s);
        where = b.sorted;
    } else if (gs4d_sort_tile_first()) {
        GS4D_PROF("radix_sort", s);
        const int tpasses = (int)((tbits + 7) / 8);
        rc = gs4d_radix_sort_pairs((uint32_t)R, tpasses, 32, b.keys_unsorted, b.values_unsorted, b.keys_tmp, b.values_tmp,
                                   b.hist, &where, debug, s, R_dev);
        if (rc) return rc;
        hipLaunchKernelGGL(tile_bounds_kernel, dim3(((uint32_t)R + 255) / 256), dim3(256), 0, s, (uint32_t)R,
                           b.keys_tmp[where], im.ranges, R_dev);
        const uint32_t nt = (uint32_t)(gx * gy), cap = gs4d_sort_segment_cap();
        const uint32_t wave_cap = cap < gs4d_sort_wave_cap() ? cap : gs4d_sort_wave_cap();
        hipLaunchKernelGGL(segment_depth_sort_kernel<false>, dim3(nt + (nt + 3) / 4), dim3(256), 0, s, im.ranges, b.keys_tmp[where],
                           b.values_tmp[where], b.keys_tmp[where ^ 1], b.values_tmp[where ^ 1], cap, wave_cap, nt);
        GS4D_LAUNCH_CHECK("segment_depth_sort", debug, s);
        where ^= 1;
    } else {
        GS4D_PROF("radix_sort", s);
        const int passes = (int)((32 + tbits + 7) / 8);
        rc = gs4d_radix_sort_pairs((uint32_t)R, passes, 0, b.keys_unsorted, b.values_unsorted, b.keys_tmp, b.values_tmp,
                                   b.hist, &where, debug, s, R_dev);
        if (rc) return rc;
    }
    if (where != b.sorted) { gs4d_set_error("binning: sorted lists ended in buffer %d, layout expects %d", where, b.sorted); return GS4D_EINVAL; }
    // ranges: already written by tile_bounds_kernel in tile-first order, by emit_place_kernel in direct placement
    if (!direct && !gs4d_sort_tile_first()) {
        GS4D_PROF("tile_ranges", s);
        hipLaunchKernelGGL(tile_ranges_kernel, dim3(((uint32_t)R + 255) / 256), dim3(256), 0, s, (uint32_t)R, b.keys_sorted(),
                           im.ranges, R_dev);
        GS4D_LAUNCH_CHECK("tile_ranges", debug, s);
    }
    return GS4D_OK;
}
