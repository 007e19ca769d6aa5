// HexPlane deformation field for gfx950: fused multi-resolution plane gathers, positional
// encodings, and the trunk / head MLPs on the f32 matrix cores (v_mfma_f32_32x32x2_f32), with
// the matching backward.
//
// Behaviour matched (reference, PyTorch):
//   scene/hexplane.py:18-19,21-47,100-131,192-215   aabb normalisation (max -> -1), bilinear border
//                                                   sampling with align_corners, product over the six
//                                                   planes, concatenation over levels
//   scene/deformation.py:128-141                    t' = force_time_embed([t, f]); trunk feature_out2
//   scene/deformation.py:147-200                    heads: ReLU, Linear, ReLU, Linear on
//                                                   [hidden, pe_tail, time_emb, force_emb]; residual adds
//   scene/deformation.py:291-298                    poc_fre layout [x, sin (dim-major), cos (dim-major)]
//   utils/graphics_utils.py:110-133                 quaternion product + normalisation
//
// gfx950 design (DESIGN.md "deformation field"):
//  * planes are channels-last, so one bilinear corner is one contiguous C*4-byte row: the gather
//    kernel gives C/4 lanes to a point and reads each corner as float4 per lane (all 12 planes
//    of a point in one kernel instead of the reference's 12 grid_sample launches);
//  * all per-point activations live in ONE row-major matrix Z = [hidden | xyz PE | scale PE |
//    rot PE | time emb | force emb]; every head reads its inputs as column segments of Z, so no
//    torch.cat copies exist;
//  * every Linear is a 128-row x 64-column workgroup tile on the exact-f32 MFMA (bit-for-bit an
//    fmaf chain, so parity with the fp32 reference holds to ~1e-6): A streamed from global memory
//    into registers, B resident in LDS (see gemm_rows_kernel);
//  * weight gradients reduce over the points with the same MFMA (K = points), one partial slab
//    per 512-point chunk and a fixed-order slab sum (deterministic);
//  * plane gradients: points are binned by texel cell (counting sort), their adds combined in
//    registers per cell, and only the combined rows go to memory with float atomics (the only
//    non-deterministic sums in the library).
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "gs4d_internal.h"
#include "deform_hexplane.h"

#define WIDTH 64            // net_width of every supported config
#define NHEAD 4             // pos, scales, rotations, shs
#define OUT_LD 72           // row pitch of the heads' raw output matrix O: every head's columns start on a multiple of 8
#define OC_X 0              //   pos 0..2 | scales 8..10 | rotations 16..19 | shs 24..71   (the gaps are zero in dO)
#define OC_S 8
#define OC_R 16
#define OC_SH 24
#define ROWS_PER_WG 128
#define GEMM_CUS 256
#define DW_ROWS_PER_WG 512

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct Layout {
    int F;        // L*C
    int oX, oS, oR, oT, oF, Kz, ldz;
    int e;        // time_emb + force_emb width
    int head_in[NHEAD];    // input widths of the heads' first Linear
    int head_seg[NHEAD];   // begin of the head's private PE segment in Z (-1: none)
    int head_seglen[NHEAD];
    int head_out[NHEAD];   // 3,3,4,48
    int head_ocol[NHEAD];  // column of the head's outputs in O
};

static Layout make_layout(const gs4d_deform_desc* d)
{
    Layout l;
    l.F = d->L * d->C;
    // every segment owns a multiple of 8 columns (zero padding) so the GEMM can fetch float4 K blocks
    l.oX = WIDTH;
    l.oS = l.oX + (6 * d->pos_pe + 7) / 8 * 8;
    l.oR = l.oS + (6 * d->sr_pe + 7) / 8 * 8;
    l.oT = l.oR + (8 * d->sr_pe + 7) / 8 * 8;
    l.oF = l.oT + 1 + 2 * d->time_pe;
    l.Kz = l.oF + 1 + 2 * d->force_pe;
    l.ldz = (l.Kz + 7) / 8 * 8;
    l.e = l.Kz - l.oT;
    const int seg[NHEAD] = {l.oX, l.oS, l.oR, -1};
    const int seglen[NHEAD] = {6 * d->pos_pe, 6 * d->sr_pe, 8 * d->sr_pe, 0};
    const int outs[NHEAD] = {3, 3, 4, 48};
    const int ocol[NHEAD] = {OC_X, OC_S, OC_R, OC_SH};
    for (int h = 0; h < NHEAD; h++) {
        l.head_seg[h] = seg[h];
        l.head_seglen[h] = seglen[h];
        l.head_in[h] = WIDTH + seglen[h] + l.e;
        l.head_out[h] = outs[h];
        l.head_ocol[h] = ocol[h];
    }
    return l;
}

struct Workspace {
    float* tprime;                 // [P]
    float* feat;                   // [P][F]
    float* trunk[GS4D_MAX_TRUNK];  // [P][64] pre-activation outputs of trunk layers 0..n-2
    float* Z;                      // [P][ldz]
    float* H1;                     // [P][256]
    float* O;                      // [P][64]
    unsigned* hmask;               // [P][10] ReLU sign bits of H1 and of the hidden columns (fused heads, deform_heads.h)
};

static Workspace carve_workspace(char* base, const gs4d_deform_desc* d, const Layout& l, size_t P, size_t* bytes)
{
    Workspace w;
    char* p = base;
    gs4d_obtain(p, w.tprime, P);
    gs4d_obtain(p, w.feat, P * l.F);
    for (int i = 0; i < GS4D_MAX_TRUNK; i++) w.trunk[i] = nullptr;
    for (int i = 0; i + 1 < d->trunk_layers; i++) gs4d_obtain(p, w.trunk[i], P * WIDTH);
    gs4d_obtain(p, w.Z, P * l.ldz);
    gs4d_obtain(p, w.H1, P * NHEAD * WIDTH);
    gs4d_obtain(p, w.O, P * OUT_LD);
    gs4d_obtain(p, w.hmask, P * 10);
    if (bytes) *bytes = (size_t)(p - base) + 256;
    return w;
}

// ------------------------------------------------------------------ positional encodings
struct PeArgs {
    int P, ldz, oX, oS, oR, oT, oF, Kz, pos_pe, sr_pe, time_pe, force_pe, use_time, use_force;
};

// A slot is one sin/cos PAIR of one input component and frequency (one range reduction serves both columns), a raw
// time / force column, or a zero padding column.  A workgroup owns PE_PTS consecutive points: it builds a table of the
// slots once (component, frequency, destination columns -- the only place that divides), stages the points' twelve
// inputs, then the (point, slot) items, dealt densely to the lanes, fill an LDS tile of the PE columns [oX, ldz), which leaves as float4 rows.
// The first version was one lane per (point, slot) with a 64-bit division by the slot count, two more by the PE
// count per lane and 4-byte stores strided over 704-byte rows: 34 us for 45 MB at C2 whatever the sincos cost.
#define PE_PTS 32
#define PE_MAX_SLOTS 256
__global__ __launch_bounds__(256) void pe_forward_kernel(PeArgs a, const float* __restrict__ xyz,
                                                         const float* __restrict__ scales,
                                                         const float* __restrict__ rots, const float* __restrict__ time,
                                                         const float* __restrict__ force, float* __restrict__ Z)
{
    extern __shared__ float s_tile[];                   // [PE_PTS][ldz - oX]
    __shared__ int4 s_tab[PE_MAX_SLOTS];                // kind (0 pair, 1 raw, 2 zero), input component, column a, column b
    __shared__ float2 s_fm[PE_MAX_SLOTS];               // frequency, keep factor (0 for a disabled time / force input)
    __shared__ float s_in[PE_PTS][12];                  // xyz 0..2 | scales 3..5 | rotations 6..9 | time 10 | force 11
    const int n_x = 3 * a.pos_pe, n_s = 3 * a.sr_pe, n_r = 4 * a.sr_pe;
    const int n_t = 1 + a.time_pe, n_f = 1 + a.force_pe;
    const int pad_x = a.oS - a.oX - 2 * n_x, pad_s = a.oR - a.oS - 2 * n_s, pad_r = a.oT - a.oR - 2 * n_r, pad_e = a.ldz - a.Kz;
    const int nslot = n_x + n_s + n_r + n_t + n_f + pad_x + pad_s + pad_r + pad_e;
    const int pitch = a.ldz - a.oX;                     // multiple of 8
    const size_t p0 = (size_t)blockIdx.x * PE_PTS;
    const int npts = (int)min((size_t)PE_PTS, (size_t)a.P - p0);
    for (int j0 = threadIdx.x; j0 < nslot; j0 += 256) {
        int j = j0, kind = 2, comp = 0, ca = 0, cb = 0;
        float freq = 1.f, mul = 1.f;
        if (j < n_x + n_s + n_r) {
            int dim, npe, c0, base;
            if (j < n_x) { dim = 3; npe = a.pos_pe; c0 = a.oX; base = 0; }
            else if (j < n_x + n_s) { j -= n_x; dim = 3; npe = a.sr_pe; c0 = a.oS; base = 3; }
            else { j -= n_x + n_s; dim = 4; npe = a.sr_pe; c0 = a.oR; base = 6; }
            kind = 0; comp = base + j / npe; freq = (float)(1 << (j % npe)); ca = c0 + j; cb = c0 + dim * npe + j;
        } else if (j - (n_x + n_s + n_r) < n_t + n_f) {
            j -= n_x + n_s + n_r;
            const bool is_force = j >= n_t;
            if (is_force) j -= n_t;
            const int npe = is_force ? a.force_pe : a.time_pe, c0 = is_force ? a.oF : a.oT;
            comp = is_force ? 11 : 10;
            mul = (is_force ? a.use_force : a.use_time) ? 1.0f : 0.0f;
            if (j == 0) { kind = 1; ca = c0; }
            else { kind = 0; freq = (float)(1 << (j - 1)); ca = c0 + j; cb = c0 + npe + j; }
        } else {
            j -= n_x + n_s + n_r + n_t + n_f;
            if (j < pad_x) ca = a.oX + 2 * n_x + j;
            else if (j < pad_x + pad_s) ca = a.oS + 2 * n_s + (j - pad_x);
            else if (j < pad_x + pad_s + pad_r) ca = a.oR + 2 * n_r + (j - pad_x - pad_s);
            else ca = a.Kz + (j - pad_x - pad_s - pad_r);
        }
        s_tab[j0] = make_int4(kind, comp, ca - a.oX, cb - a.oX);
        s_fm[j0] = make_float2(freq, mul);
    }
    for (int e = threadIdx.x; e < npts * 12; e += 256) {
        const int pl = e / 12, c = e - pl * 12;
        const size_t p = p0 + pl;
        s_in[pl][c] = c < 3 ? xyz[3 * p + c] : c < 6 ? scales[3 * p + c - 3] : c < 10 ? rots[4 * p + c - 6] : c == 10 ? time[p] : force[p];
    }
    __syncthreads();
    // (point, slot) items dealt densely to the lanes; item / nslot by a multiply-high (exact for these small ranges)
    const unsigned rcp = (unsigned)((0x100000000ull + (unsigned)nslot - 1) / (unsigned)nslot);
    for (int item = threadIdx.x; item < npts * nslot; item += 256) {
        const int pl = (int)__umulhi((unsigned)item, rcp), j = item - pl * nslot;
        float* __restrict__ z = s_tile + pl * pitch;
        const int4 t = s_tab[j];
        const float2 fm = s_fm[j];
        const float x = s_in[pl][t.y];
        if (t.x == 0) {
            float sn, cs;
            sincosf(x * fm.x, &sn, &cs);
            z[t.z] = sn * fm.y;                         // x * 0.0f for a disabled input: a NaN time stays visible
            z[t.w] = cs * fm.y;
        } else {
            z[t.z] = t.x == 1 ? x * fm.y : 0.f;
        }
    }
    __syncthreads();
    const int q4 = pitch >> 2;                          // float4 per row of the tile
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(s_tile);
    for (int e = threadIdx.x; e < npts * q4; e += 256) {
        const int pl = e / q4, c = e - pl * q4;
        reinterpret_cast<float4*>(Z + (p0 + pl) * a.ldz + a.oX)[c] = t4[e];
    }
}

// Backward of the xyz / scale / rotation encodings, same shape as the forward: a workgroup loads the dZ columns
// [oX, oT) of PE_PTS points into an LDS tile (float4 rows), one lane per (point, (component, frequency) pair) turns
// the pair's two gradients into its contribution g_sin cos(x f) f - g_cos sin(x f) f (sines / cosines recomputed:
// re-reading them from Z measured slower), written over the pair's sine column, and ten lanes per point add up their
// component's frequencies.  The first version had one lane per (point, component) with a 64-bit division, loops of 10
// and 2 frequencies side by side in a wave and 4-byte reads strided over the rows: 23 us at C2.
__global__ __launch_bounds__(256) void pe_backward_kernel(PeArgs a, const float* __restrict__ xyz,
                                                          const float* __restrict__ scales,
                                                          const float* __restrict__ rots, const float* __restrict__ dZ,
                                                          float* __restrict__ dxyz, float* __restrict__ dscales,
                                                          float* __restrict__ drots, int* __restrict__ clear, int nclear,
                                                          int apply_mask)    // 1: dZ is unmasked (fused heads): the ReLU mask of a
                                                                             // column is the sign of its recomputed sine / cosine
{
    extern __shared__ float s_tile[];                   // [PE_PTS][oT - oX]
    __shared__ int4 s_tab[PE_MAX_SLOTS];                // input component, sine column, cosine column (relative to oX), -
    __shared__ float s_fr[PE_MAX_SLOTS];
    __shared__ float s_in[PE_PTS][10];                  // xyz 0..2 | scales 3..5 | rotations 6..9
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    // the cell histograms of the plane backward (next launch in the stream) start at zero: cleared here, by lanes this
    // launch has to spare, instead of by a memset node of its own
    for (size_t i = gid; i < (size_t)nclear; i += (size_t)gridDim.x * 256) clear[i] = 0;
    const int n_x = 3 * a.pos_pe, n_s = 3 * a.sr_pe, n_r = 4 * a.sr_pe;
    const int npair = n_x + n_s + n_r;
    const int pitch = a.oT - a.oX;                      // multiple of 8
    const size_t p0 = (size_t)blockIdx.x * PE_PTS;
    if (p0 >= (size_t)a.P) return;                      // uniform (grid rounded up)
    const int npts = (int)min((size_t)PE_PTS, (size_t)a.P - p0);
    for (int j0 = threadIdx.x; j0 < npair; j0 += 256) {
        int j = j0, dim, npe, c0, base;
        if (j < n_x) { dim = 3; npe = a.pos_pe; c0 = a.oX; base = 0; }
        else if (j < n_x + n_s) { j -= n_x; dim = 3; npe = a.sr_pe; c0 = a.oS; base = 3; }
        else { j -= n_x + n_s; dim = 4; npe = a.sr_pe; c0 = a.oR; base = 6; }
        s_tab[j0] = make_int4(base + j / npe, c0 + j - a.oX, c0 + dim * npe + j - a.oX, 0);
        s_fr[j0] = (float)(1 << (j % npe));
    }
    for (int e = threadIdx.x; e < npts * 10; e += 256) {
        const int pl = e / 10, c = e - pl * 10;
        const size_t p = p0 + pl;
        s_in[pl][c] = c < 3 ? xyz[3 * p + c] : c < 6 ? scales[3 * p + c - 3] : rots[4 * p + c - 6];
    }
    const int q4 = pitch >> 2;
    float4* __restrict__ t4 = reinterpret_cast<float4*>(s_tile);
    for (int e = threadIdx.x; e < npts * q4; e += 256) {
        const int pl = e / q4, c = e - pl * q4;
        t4[e] = reinterpret_cast<const float4*>(dZ + (p0 + pl) * a.ldz + a.oX)[c];
    }
    __syncthreads();
    const unsigned rcp = (unsigned)((0x100000000ull + (unsigned)npair - 1) / (unsigned)npair);
    for (int item = threadIdx.x; item < npts * npair; item += 256) {
        const int pl = (int)__umulhi((unsigned)item, rcp), j = item - pl * npair;
        float* __restrict__ g = s_tile + pl * pitch;
        const int4 t = s_tab[j];
        const float fr = s_fr[j];
        float sn, cs;
        sincosf(s_in[pl][t.x] * fr, &sn, &cs);
        const float gs_ = (!apply_mask || sn > 0.f) ? g[t.y] : 0.f;
        const float gc_ = (!apply_mask || cs > 0.f) ? g[t.z] : 0.f;
        float acc = gs_ * cs * fr;
        acc -= gc_ * sn * fr;
        g[t.y] = acc;                                   // nobody else reads this pair's columns
    }
    __syncthreads();
    for (int e = threadIdx.x; e < npts * 10; e += 256) {
        const int pl = e / 10, c = e - pl * 10;
        const size_t p = p0 + pl;
        const int npe = c < 3 ? a.pos_pe : a.sr_pe;
        const int col = c < 3 ? c * npe : c < 6 ? (a.oS - a.oX) + (c - 3) * npe : (a.oR - a.oX) + (c - 6) * npe;
        const float* __restrict__ g = s_tile + pl * pitch + col;
        float acc = 0.f;
        for (int k = 0; k < npe; k++) acc += g[k];      // same order as the per-component loop of the first version
        float* dst = c < 3 ? dxyz + 3 * p + c : c < 6 ? dscales + 3 * p + c - 3 : drots + 4 * p + c - 6;
        *dst += acc;
    }
}

// ------------------------------------------------------------------ MFMA row-block GEMM
// Y[rows, ycol0 + j] = mask( sum_k actA(A[rows, acol(k)]) * B[k][j] + bias[j] ), j < N <= 64.
//
// A's K columns are up to three segments of the source row.  Every segment starts on a 4-float boundary and
// owns pad8(len) columns of the row (the layouts in this file pad with zeros), so a lane can fetch its
// operand as one float4 per 8-wide K block.  B rows follow the TRUE column order (no padding); the staged
// copy in LDS has zero rows where A has padding.
struct GemmJob {
    const float* A; int lda;
    int a_nseg; int a_beg[3]; int a_len[3];        // source columns of A, concatenated; K = sum(a_len)
    int K; int relu_a;
    int b_nseg; const float* B[4]; int b_ld[4]; int b_rows[4]; int b_col0[4];
    int b_transposed;                              // 1: B[k][j] = Bsrc[(b_col0 + j) * ld + k_local] (nn.Linear weight [N][K])
                                                   // 0: B[k][j] = Bsrc[k_local * ld + b_col0 + j]
    const float* bias; int N;
    float* Y; int ldy; int ycol0;
    const float* mask; int ldm; int mcol0;         // optional: multiply by (mask > 0)
    // FUSE variant (N == 64): Y2[rows, y2col0 + j] = relu(Y) W2^T + bias2, j < N2 <= 64, W2 = nn.Linear weight [N2][64];
    // Y itself is stored only if store_y
    const float* W2; const float* bias2; int N2; float* Y2; int ldy2; int y2col0; int store_y;
};
struct GemmJobs { GemmJob j[8]; int wg_begin[9]; };   // job i owns workgroups [wg_begin[i], wg_begin[i+1])

static inline __host__ __device__ int pad8(int v) { return (v + 7) & ~7; }
static inline __host__ __device__ int pad4(int v) { return (v + 3) & ~3; }
static inline __host__ __device__ int gemm_kp(const GemmJob& J)
{
    int kp = 0;
    for (int s = 0; s < J.a_nseg; s++) kp += pad8(J.a_len[s]);
    return kp;
}

#define GEMM_GB 8     // 8-wide K blocks per register group (one group is in flight while the previous one feeds the MFMAs)
#define GEMM_MAX_KB 40

// Design: the workgroup stages B (<= 320 x 64) in LDS once and then walks row blocks of 128 (persistent grid);
// each wave owns 32 rows and streams its A operand straight from global memory into registers: with
// v_mfma_f32_32x32x2_f32 lane (i, h) supplies A[i][k] for one k per instruction, and since the sum over k
// may run in any order as long as A and B agree, lane (i, h) takes k = 8m + 4h + q (q = 0..3) of block m --
// one 16-byte load per block.  B is staged as sB[m][h][j][4] (the four k values lane (j, h) needs from block
// m are contiguous and neighbouring lanes are 16 bytes apart), so a lane fetches them with one
// conflict-free ds_read_b128 per column tile.  No LDS round trip for A and no barrier in the main loop;
// the next group of blocks (or the next row block) is loaded while the current one is multiplied.
// RELU / MASK are compile-time so the MFMA loop and the epilogue are branch-free; both column tiles are
// always multiplied (columns >= N are zero in LDS).
// Measured (C2, rocprofv3): every launch moves 2.2-3.6 TB/s of its own operands -- these layer-by-layer
// products are bound by HBM traffic, not by the matrix cores (MFMA busy ~46 %).
// FUSE: a head's two Linears in one pass over the rows.  The first layer's 32 x 64 accumulator tile of a wave is
// biased, (optionally stored,) rectified and transposed through a small per-wave LDS tile -- 32 columns at a
// time -- into the A-operand layout, multiplied with the second weight matrix (resident in LDS next to the
// first), and only the second layer's <= 64 outputs per row are written: the 256-float H1 row is not re-read
// (and, without a backward to feed, not even written).
#define FUSE_TP 36          // row pitch of the transpose tile (floats)
template <bool RELU, bool MASK, bool FUSE>
__global__ __launch_bounds__(256) void gemm_rows_kernel(int P, GemmJobs jobs)
{
    extern __shared__ float smem[];
    __shared__ int s_acol[GEMM_MAX_KB];
    int job_v = 0;
    while (job_v < 7 && (int)blockIdx.x >= jobs.wg_begin[job_v + 1]) job_v++;
    const int job = __builtin_amdgcn_readfirstlane(job_v);     // provably uniform: the descriptor is read with scalar loads
    const GemmJob& J = jobs.j[job];
    const int wg = blockIdx.x - jobs.wg_begin[job], nwg = jobs.wg_begin[job + 1] - jobs.wg_begin[job];
    const int Kp = gemm_kp(J);
    const int Kb = Kp >> 3;
    const int N = J.N;
    float* sB = smem;                                   // [Kb][2 (k half)][64 (column)][4 (k)]
    float* sB2 = sB + Kp * 64;                          // FUSE: [8][2][64][4]
    float* sT = sB2 + 64 * 64;                          // FUSE: [4 waves][32][FUSE_TP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (FUSE) {                                         // B2[k][j] = W2[j][k], zero for j >= N2
        for (int e = tid; e < 64 * 64; e += 256) {
            const int j = e >> 6, k = e & 63;
            sB2[(((k >> 3) * 2 + ((k >> 2) & 1)) * 64 + j) * 4 + (k & 3)] = j < J.N2 ? J.W2[(size_t)j * 64 + k] : 0.f;
        }
    }

    // ---- stage B (zero rows for A's padding columns) and the A column of every K block.  All of a thread's
    // loads are issued before its first LDS store, so staging costs about one memory round trip.
    {
        float v[GEMM_MAX_KB / 4][8];                    // Kp <= 320: at most 10 rows of k per thread
        bool kval[GEMM_MAX_KB / 4];
#pragma unroll
        for (int it = 0; it < GEMM_MAX_KB / 4; it++) {
            const int kp = (tid & 31) + 32 * it;
            kval[it] = false;
            if (kp >= Kp) continue;
            // padded k -> (segment, local k) -> true k
            int sg = 0, kl = kp, ktrue = 0;
            while (sg < J.a_nseg - 1 && kl >= pad8(J.a_len[sg])) { kl -= pad8(J.a_len[sg]); ktrue += J.a_len[sg]; sg++; }
            kval[it] = kl < J.a_len[sg];
            ktrue += min(kl, J.a_len[sg] - 1);
            if ((kp & 7) == 0 && tid < 32) s_acol[kp >> 3] = J.a_beg[sg] + kl;
            int bs = 0, bl = ktrue;
            while (bs < J.b_nseg - 1 && bl >= J.b_rows[bs]) { bl -= J.b_rows[bs]; bs++; }
            const float* src = J.B[bs];
            const int ld = J.b_ld[bs], c0 = J.b_col0[bs];
#pragma unroll
            for (int jj = 0; jj < 8; jj++) {
                const int j = min((tid >> 5) + 8 * jj, N - 1);
                v[it][jj] = J.b_transposed ? src[(size_t)(c0 + j) * ld + bl] : src[(size_t)bl * ld + c0 + j];
            }
        }
#pragma unroll
        for (int it = 0; it < GEMM_MAX_KB / 4; it++) {
            const int kp = (tid & 31) + 32 * it;
            if (kp >= Kp) continue;
#pragma unroll
            for (int jj = 0; jj < 8; jj++) {
                const int j = (tid >> 5) + 8 * jj;
                sB[(((kp >> 3) * 2 + ((kp >> 2) & 1)) * 64 + j) * 4 + (kp & 3)] = (kval[it] && j < N) ? v[it][jj] : 0.f;
            }
        }
    }
    __syncthreads();

    const int i = lane & 31, khalf = lane >> 5, col = i;
    const int nrb = (P + ROWS_PER_WG - 1) / ROWS_PER_WG;
    const int NG = (Kb + GEMM_GB - 1) / GEMM_GB;
    const int my_rb = nrb > wg ? (nrb - 1 - wg) / nwg + 1 : 0;
    const int T = my_rb * NG;                           // flattened (row block, group) steps of this workgroup
    const float* __restrict__ Abase = J.A;
    const int lda = J.lda;

    auto load_group = [&](int t, float4 (&dst)[GEMM_GB]) {
        const int rb = wg + (t / NG) * nwg, g = t % NG;
        const int row = min(rb * ROWS_PER_WG + wave * 32 + i, P - 1);
        const float* arow = Abase + (size_t)row * lda + 4 * khalf;
#pragma unroll
        for (int b = 0; b < GEMM_GB; b++) {
            const int m = min(g * GEMM_GB + b, Kb - 1);  // clamped: the surplus blocks of the last group are never multiplied
            dst[b] = *reinterpret_cast<const float4*>(arow + s_acol[m]);
        }
    };

    f32x16 acc0 = {0}, acc1 = {0};
    const float bias0 = (J.bias && col < N) ? J.bias[col] : 0.f;
    const float bias1 = (J.bias && col + 32 < N) ? J.bias[col + 32] : 0.f;
    float* __restrict__ Y = J.Y + J.ycol0;
    const int ldy = J.ldy;
    const float* __restrict__ Mk = MASK ? J.mask + J.mcol0 : nullptr;
    const int ldm = J.ldm;
    const float4* sB4 = reinterpret_cast<const float4*>(sB) + khalf * 64 + col;   // + m * 128 (+ 32 for the second tile)

    auto compute_group = [&](int t, const float4 (&src)[GEMM_GB]) {
        const int rb = wg + (t / NG) * nwg, g = t % NG;
        // MASK: the epilogue's 32 mask values per lane are requested BEFORE the row block's last group is multiplied,
        // so their memory latency hides behind the MFMAs (with one wave per SIMD nothing else would cover it)
        float m0[16], m1[16];
        if (MASK && !FUSE && g == NG - 1) {
            const int mrow0 = rb * ROWS_PER_WG + wave * 32 + 4 * khalf;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = min(mrow0 + (r & 3) + 8 * (r >> 2), P - 1);
                const float* mrow = Mk + (size_t)row * ldm;
                m0[r] = mrow[min(col, N - 1)];
                m1[r] = mrow[min(col + 32, N - 1)];
            }
        }
#pragma unroll
        for (int b = 0; b < GEMM_GB; b++) {
            const int m = g * GEMM_GB + b;
            if (m >= Kb) break;
            const float4 b0 = sB4[m * 128], b1 = sB4[m * 128 + 32];
            float av[4] = {src[b].x, src[b].y, src[b].z, src[b].w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (RELU) av[q] = fmaxf(av[q], 0.f);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv0[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv1[q], acc1, 0, 0, 0);
            }
        }
        if (g == NG - 1 && FUSE) {
            const int row0 = rb * ROWS_PER_WG + wave * 32 + 4 * khalf;
            float* Tw = sT + wave * (32 * FUSE_TP);
            const float4* sB24 = reinterpret_cast<const float4*>(sB2) + khalf * 64 + col;
            f32x16 c0 = {0}, c1 = {0};
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const float bias = half ? bias1 : bias0;
                __builtin_amdgcn_wave_barrier();            // the previous half's readers are done (same wave)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int rl = (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    const float v = (half ? acc1[r] : acc0[r]) + bias;
                    if (J.store_y && row0 - 4 * khalf + rl < P) Y[(size_t)(row0 - 4 * khalf + rl) * ldy + col + 32 * half] = v;
                    Tw[rl * FUSE_TP + col] = fmaxf(v, 0.f);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const float4 a4 = *reinterpret_cast<const float4*>(Tw + i * FUSE_TP + 8 * m + 4 * khalf);
                    const float4 b0 = sB24[(half * 4 + m) * 128], b1 = sB24[(half * 4 + m) * 128 + 32];
                    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
                    const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv0[q], c0, 0, 0, 0);
                        if (J.N2 > 32) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv1[q], c1, 0, 0, 0);
                    }
                }
            }
            const int N2 = J.N2;
            const float b20 = col < N2 ? J.bias2[col] : 0.f, b21 = col + 32 < N2 ? J.bias2[col + 32] : 0.f;
            float* __restrict__ Y2 = J.Y2 + J.y2col0;
            const int ldy2 = J.ldy2;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = row0 + (r & 3) + 8 * (r >> 2);
                if (row >= P) continue;
                if (col < N2) Y2[(size_t)row * ldy2 + col] = c0[r] + b20;
                if (col + 32 < N2) Y2[(size_t)row * ldy2 + col + 32] = c1[r] + b21;
            }
            acc0 = (f32x16){0};
            acc1 = (f32x16){0};
        }
        if (g == NG - 1 && !FUSE) {
            // epilogue: accumulator register r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
            const int row0 = rb * ROWS_PER_WG + wave * 32 + 4 * khalf;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = row0 + (r & 3) + 8 * (r >> 2);
                if (row >= P) continue;
                float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
                if (MASK) { v0 = m0[r] > 0.f ? v0 : 0.f; v1 = m1[r] > 0.f ? v1 : 0.f; }
                if (col < N) Y[(size_t)row * ldy + col] = v0;
                if (col + 32 < N) Y[(size_t)row * ldy + col + 32] = v1;
            }
            acc0 = (f32x16){0};
            acc1 = (f32x16){0};
        }
    };

    // two register buffers: the next group's loads are in flight while the current one is multiplied
    float4 buf0[GEMM_GB], buf1[GEMM_GB];
    if (T > 0) load_group(0, buf0);
    for (int t = 0; t < T; t += 2) {
        if (t + 1 < T) load_group(t + 1, buf1);
        compute_group(t, buf0);
        if (t + 2 < T) load_group(t + 2, buf0);
        if (t + 1 < T) compute_group(t + 1, buf1);
    }
}

// The same product for the shape every trunk layer but a wide first one has, in both directions: K = 64 contiguous
// columns of A, N <= 64.  A wave's step is one 32-row tile: the NEXT tile's eight float4 of A and (MASK) this tile's
// 32 mask values are requested first, unconditionally (clamped rows: the compiler can then count the outstanding loads
// instead of draining them), the 64 MFMAs run without a branch, the epilogue follows.  In the general kernel above the
// row-block / group bookkeeping sits inside the MFMA loop as run-time tests and costs ~220 register copies and a
// full wait per group; at this shape that was 25 us per launch at C2 for 10 us of traffic.
template <bool RELU, bool MASK>
__global__ __launch_bounds__(256) void gemm64_rows_kernel(int P, GemmJobs jobs)
{
    __shared__ float4 sB4[8 * 2 * 64];                  // [m][h][j] -> the four k = 8m + 4h + q of column j
    int job_v = 0;
    while (job_v < 7 && (int)blockIdx.x >= jobs.wg_begin[job_v + 1]) job_v++;
    const int job = __builtin_amdgcn_readfirstlane(job_v);
    const GemmJob& J = jobs.j[job];
    const int wg = blockIdx.x - jobs.wg_begin[job], nwg = jobs.wg_begin[job + 1] - jobs.wg_begin[job];
    const int N = J.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {   // stage B: thread -> (column j, eight consecutive k); zero columns past N
        const float* src = J.B[0];
        const int ld = J.b_ld[0], c0 = J.b_col0[0];
        for (int e = tid; e < 64 * 8; e += 256) {
            const int j = e & 63, kb = e >> 6;          // k = 8 kb .. 8 kb + 7
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int k = 8 * kb + q, jc = min(j, N - 1);
                v[q] = J.b_transposed ? src[(size_t)(c0 + jc) * ld + k] : src[(size_t)k * ld + c0 + jc];
                if (j >= N) v[q] = 0.f;
            }
            sB4[(kb * 2 + 0) * 64 + j] = make_float4(v[0], v[1], v[2], v[3]);
            sB4[(kb * 2 + 1) * 64 + j] = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
    __syncthreads();
    const int i = lane & 31, khalf = lane >> 5, col = i;
    const int ntile = (P + 31) >> 5;                    // 32-row tiles; wave w of workgroup g takes tiles 4 (g + s nwg) + w
    const int tstride = 4 * nwg;
    int tile = 4 * wg + wave;
    if (tile >= ntile) return;                          // no barrier below
    const float* __restrict__ Ab = J.A + J.a_beg[0] + 4 * khalf;
    const int lda = J.lda;
    const float bias0 = (J.bias && col < N) ? J.bias[col] : 0.f;
    const float bias1 = (J.bias && col + 32 < N) ? J.bias[col + 32] : 0.f;
    float* __restrict__ Y = J.Y + J.ycol0;
    const int ldy = J.ldy;
    const float* __restrict__ Mk = MASK ? J.mask + J.mcol0 : nullptr;
    const int ldm = J.ldm;
    const float4* bcol = sB4 + khalf * 64 + col;
    const int c0m = min(col, N - 1), c1m = min(col + 32, N - 1);

    float4 a[8], an[8];
    {
        const float* arow = Ab + (size_t)min(32 * tile + i, P - 1) * lda;
#pragma unroll
        for (int m = 0; m < 8; m++) a[m] = *reinterpret_cast<const float4*>(arow + 8 * m);
    }
    for (; tile < ntile; tile += tstride) {
        {   // next tile's A (clamped to the last tile: re-read, never used)
            const float* arow = Ab + (size_t)min(32 * min(tile + tstride, ntile - 1) + i, P - 1) * lda;
#pragma unroll
            for (int m = 0; m < 8; m++) an[m] = *reinterpret_cast<const float4*>(arow + 8 * m);
        }
        const int row0 = 32 * tile + 4 * khalf;
        float m0[16], m1[16];
        if (MASK) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float* mrow = Mk + (size_t)min(row0 + (r & 3) + 8 * (r >> 2), P - 1) * ldm;
                m0[r] = mrow[c0m];
                m1[r] = mrow[c1m];
            }
        }
        f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const float4 b0 = bcol[m * 128], b1 = bcol[m * 128 + 32];
            float av[4] = {a[m].x, a[m].y, a[m].z, a[m].w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (RELU) av[q] = fmaxf(av[q], 0.f);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv0[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv1[q], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
            if (MASK) { v0 = m0[r] > 0.f ? v0 : 0.f; v1 = m1[r] > 0.f ? v1 : 0.f; }
            if (row < P && col < N) Y[(size_t)row * ldy + col] = v0;
            if (row < P && col + 32 < N) Y[(size_t)row * ldy + col + 32] = v1;
        }
#pragma unroll
        for (int m = 0; m < 8; m++) a[m] = an[m];
    }
}


// ------------------------------------------------------------------ weight gradients (reduce over points)
// slab[chunk][n][k] = sum over the chunk's rows of G[row][gcol0+n] * actX(X[row][xcol(k)]);  slab[chunk][n][Kp] = sum of G[row][gcol0+n]
struct DwJob {
    const float* G; int ldg; int gcol0; int N;
    const float* X; int ldx; int x_nseg; int x_beg[3]; int x_len[3]; int K; int relu_x;
    int Kp;                              // K with every segment padded to 4 columns (the staged / slab k space; the source pads with zeros)
    float* slabs; int kblocks;           // kblocks = ceil(Kp/32) k-blocks of MFMA tiles; slab row pitch = 32*ceil((Kp+1)/32):
                                         // column Kp of a slab row holds the bias partial (column sums of G, formed while G is staged)
    float* dW; float* db;                // final destinations [N][K], [N]
};
// One job may be BLOCK-DIAGONAL: the second Linears of the three small heads (3 / 3 / 4 outputs over 64 inputs each) run as
// one job G = dO[:, 0..23] x X = H1[:, 0..191] -- six tiles in one workgroup per chunk instead of three workgroups of two
// tiles each, every one of which paid the full per-step staging and barrier cost -- and the reduce picks head h's rows
// [n0, n0 + n) x columns [k0, k0 + 64) out of the slab.
struct DwSplit { int job; int n0[3], n[3], k0[3], K; float* dW[3]; float* db[3]; };   // job < 0: none
struct DwJobs { DwJob j[16]; int njobs, nz, nchunks; DwSplit sp; };

// One workgroup = one (chunk of DW_ROWS_PER_WG point rows, job, slice of <= 6 k-blocks).  The rows are staged
// through LDS in sub-blocks of DW_RB (every element of G and X is fetched from memory exactly once per
// workgroup, coalesced, ReLU applied at the store), double
// buffered, and the (n-block, k-block) output tiles are dealt round-robin to the four waves; both MFMA
// operands are plain ds_read_b32 (row pitches are 32 mod 64 floats, so the two k halves of a wave use
// disjoint banks).  The earlier version fetched operands per wave straight from global memory: every G block
// was read by two waves and every X block by two more, 1.8 GB of L2 traffic per step for 0.4 GB of data.
#define DW_RB 16
#define DW_DEPTH 2              // sub-blocks of global loads in flight per thread (even: the LDS double buffer follows the parity)
#define DW_NG (DW_RB / 16)       // float4s of G per thread and sub-block
#define DW_NX (3 * DW_RB / 16)   // float4s of X
#define DW_GP 96
#define DW_XP 224
#define DW_XCOLS 192
// The MFMAs of one staged sub-block for a wave that owns CNT tiles (same n-block, CNT k-blocks): every LDS operand of
// the sub-block is requested first, then the MFMAs issue back to back.  Written with run-time `if (cnt > 1)` tests
// inside the step loop, the compiler emitted ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma for every single product: the LDS
// latency was exposed 8..24 times per sub-block and a job of 3 tiles took as long as one of 10.
template <int CNT>
__device__ __forceinline__ void dw_mfma_subblock(const float* __restrict__ ga, const float* __restrict__ xb, int kstep,
                                                 f32x16 (&acc)[3])
{
    float a[DW_RB / 2], b[CNT][DW_RB / 2];
#pragma unroll
    for (int st = 0; st < DW_RB / 2; st++) {
        a[st] = ga[2 * st * DW_GP];
#pragma unroll
        for (int c = 0; c < CNT; c++) b[c][st] = xb[2 * st * DW_XP + c * kstep];
    }
#pragma unroll
    for (int st = 0; st < DW_RB / 2; st++) {
#pragma unroll
        for (int c = 0; c < CNT; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], b[c][st], acc[c], 0, 0, 0);
    }
}

__global__ __launch_bounds__(256, 3) void dw_partial_kernel(int P, DwJobs jobs)
{
    __shared__ float sG[2][DW_RB * DW_GP];
    __shared__ float sX[2][DW_RB * DW_XP];
    // the column table is only read during set-up and lives in staging buffer 1, which is first written after the
    // prologue's barrier (40 960 bytes of LDS in all; four workgroups per CU would fit, but not their registers)
    int* s_xcol = reinterpret_cast<int*>(&sX[1][0]);
    // 1-D grid, XCD-aware: consecutive workgroup ids go to different XCDs (each with its own L2), so the
    // jobs of one chunk -- which re-read the same rows of Z / dH1 / H1 / dO -- are placed 8 ids apart:
    // same XCD, adjacent in dispatch order.
    const int id = blockIdx.x;
    const int rest = id >> 3;
    const int job_v = rest % jobs.njobs;
    const int rest2 = rest / jobs.njobs;
    const int zblk = rest2 % jobs.nz;
    const int chunk = (rest2 / jobs.nz) * 8 + (id & 7);
    if (chunk >= jobs.nchunks) return;
    const DwJob& J = jobs.j[__builtin_amdgcn_readfirstlane(job_v)];
    const int kb_first = zblk * 6;
    const int nkb = min(6, J.kblocks - kb_first);
    if (nkb <= 0) return;                                   // uniform
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int nn = (J.N + 31) >> 5;                         // 1 or 2 n-blocks
    const int ntiles = nn * nkb;                            // <= 12; wave w owns tiles w, w+4, w+8 (all with n-block w % nn)
    const int vw = (wave + chunk) & 3;                      // rotate the tile deal so the 3-tile waves do not all sit on one SIMD
    const int cnt = __builtin_amdgcn_readfirstlane((ntiles - vw + 3) >> 2);
    const int nb = vw % nn;
    const int rbeg = chunk * DW_ROWS_PER_WG, rend = min(P, rbeg + DW_ROWS_PER_WG);

    // staged 4-column group -> source column of X.  Segments begin on 4-float boundaries and every segment but
    // the last is a multiple of 4 long (checked on the host), so a group is one aligned float4 of the source row.
    // Nothing is patched: a staged column at or past K (the source row is padded, so the over-read stays inside
    // it; -1: the group is entirely past Kp and is not loaded at all) only ever meets slab columns that
    // dw_reduce_kernel does not read -- every output (n, k) is a sum of products of column k alone.  The same
    // holds for columns of G at or past N.  The staging used to zero / one-fill per component; that VALU work
    // cost as much issue time as the MFMAs (a job of 3 tiles and one of 10 both took 35-45 us on their own).
    for (int c = tid; c < DW_XCOLS / 4; c += 256) {
        const int k = kb_first * 32 + 4 * c;
        int src = -1;
        if (4 * c < nkb * 32 && k < J.Kp) {
            int kl = k, sg = 0;
            while (sg < J.x_nseg - 1 && kl >= pad4(J.x_len[sg])) { kl -= pad4(J.x_len[sg]); sg++; }
            src = J.x_beg[sg] + kl;
        }
        s_xcol[c] = src;
    }
    __syncthreads();
    // fixed staging assignments of this thread: one float4 of G (16 rows x 16 groups), three of X (16 x 48)
    int xoff[DW_NX], xrow[DW_NX], xdst[DW_NX];
    bool xlive[DW_NX];
#pragma unroll
    for (int i = 0; i < DW_NX; i++) {
        const int e = tid + 256 * i;
        const int row = e / (DW_XCOLS / 4), grp = e - row * (DW_XCOLS / 4);
        xrow[i] = row;
        xoff[i] = s_xcol[grp];
        xdst[i] = row * DW_XP + 4 * grp;
        xlive[i] = 4 * grp < nkb * 32;                      // groups of k-blocks this slice does not own are never read
    }
    const int ggrp = tid & 15, grow = tid >> 4;             // G: row grow, columns 4*ggrp .. +3
    const float* __restrict__ Gp = J.G + J.gcol0 + 4 * min(ggrp, (J.N - 1) >> 2);
    const float* __restrict__ Xp = J.X;
    const int ldg = J.ldg, ldx = J.ldx;
    const bool relu = J.relu_x != 0;

    float4 gvs[DW_DEPTH][DW_NG], xvs[DW_DEPTH][DW_NX];     // register ring: loads run DW_DEPTH sub-blocks ahead of the MFMAs
    // Every fetch issues the same DW_NG + DW_NX loads, unconditionally (a slot with nothing to stage re-reads the
    // slice's first group, rows past the chunk's end are clamped): with loads under branches the compiler cannot
    // count what is outstanding and put s_waitcnt vmcnt(0) in front of every LDS store, i.e. each step also waited
    // for the loads issued one step earlier for two steps later -- the ring was one step deep, not DW_DEPTH.
    int xsrc[DW_NX];
#pragma unroll
    for (int i = 0; i < DW_NX; i++) xsrc[i] = (xlive[i] && xoff[i] >= 0) ? xoff[i] : s_xcol[0];
    auto fetch = [&](float4 (&gv)[DW_NG], float4 (&xv)[DW_NX], int r0) {
#pragma unroll
        for (int i = 0; i < DW_NG; i++)
            gv[i] = *reinterpret_cast<const float4*>(Gp + (size_t)min(r0 + grow + 16 * i, rend - 1) * ldg);
#pragma unroll
        for (int i = 0; i < DW_NX; i++)
            xv[i] = *reinterpret_cast<const float4*>(Xp + (size_t)min(r0 + xrow[i], rend - 1) * ldx + xsrc[i]);
    };
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);          // bias partial: column sums of this thread's rows of G
    auto stash = [&](const float4 (&gv)[DW_NG], const float4 (&xv)[DW_NX], int r0, int buf) {
#pragma unroll
        for (int i = 0; i < DW_NG; i++) {
            float4 g4 = gv[i];
            if (r0 + grow + 16 * i >= rend) g4 = make_float4(0.f, 0.f, 0.f, 0.f);   // rows past the end were clamped, not skipped
            bsum.x += g4.x; bsum.y += g4.y; bsum.z += g4.z; bsum.w += g4.w;
            *reinterpret_cast<float4*>(&sG[buf][(grow + 16 * i) * DW_GP + 4 * ggrp]) = g4;
        }
#pragma unroll
        for (int i = 0; i < DW_NX; i++) {
            if (!xlive[i]) continue;
            float4 v = xv[i];
            if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));   // uniform per job
            *reinterpret_cast<float4*>(&sX[buf][xdst[i]]) = v;
        }
    };

    const float* ga0 = &sG[0][half * DW_GP + nb * 32 + l31];
    const float* xb0 = &sX[0][half * DW_XP + (vw / nn) * 32 + l31];
    const int kstep = (4 / nn) * 32;                        // k-block distance between the wave's consecutive tiles
    const int pitch = 32 * ((J.Kp + 1 + 31) >> 5);
    float* slab = J.slabs + (size_t)chunk * 64 * pitch;
    // The whole row loop is instantiated per tile count of the wave (0..3, wave-uniform) and selected ONCE: a run-time
    // count inside the loop made the compiler shuttle all 48 accumulator registers between the branches' copies on
    // every step.
    auto run_rows = [&](auto tag) {
        constexpr int CNT = decltype(tag)::value;
        f32x16 acc[3] = {{0}, {0}, {0}};
#pragma unroll
        for (int i = 0; i < DW_DEPTH; i++) fetch(gvs[i], xvs[i], rbeg + i * DW_RB);
        stash(gvs[0], xvs[0], rbeg, 0);
        __syncthreads();
        fetch(gvs[0], xvs[0], rbeg + DW_DEPTH * DW_RB);
        for (int r = rbeg; r < rend; r += DW_DEPTH * DW_RB) {
#pragma unroll
            for (int u = 0; u < DW_DEPTH; u++) {
                const int rr = r + u * DW_RB;
                if (rr >= rend) break;                      // uniform
                if constexpr (CNT > 0)
                    dw_mfma_subblock<CNT>(ga0 + (u & 1) * (DW_RB * DW_GP), xb0 + (u & 1) * (DW_RB * DW_XP), kstep, acc);
                // ring slot (u+1) holds sub-block rr + DW_RB (fetched DW_DEPTH steps ago): into the other LDS buffer,
                // then the slot is refilled with the sub-block DW_DEPTH further on
                if (rr + DW_RB < rend) stash(gvs[(u + 1) % DW_DEPTH], xvs[(u + 1) % DW_DEPTH], rr + DW_RB, (u + 1) & 1);
                fetch(gvs[(u + 1) % DW_DEPTH], xvs[(u + 1) % DW_DEPTH], rr + (DW_DEPTH + 1) * DW_RB);   // past the end: clamped rows, never staged
                __syncthreads();
            }
        }
#pragma unroll
        for (int t = 0; t < CNT; t++) {
            const int kb = kb_first + vw / nn + t * (4 / nn);
            if (kb * 32 + l31 == J.Kp) continue;            // that column is the bias partial's (written below)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                slab[(size_t)n * pitch + kb * 32 + l31] = acc[t][r];
            }
        }
    };
    if (cnt >= 3) run_rows(std::integral_constant<int, 3>{});
    else if (cnt == 2) run_rows(std::integral_constant<int, 2>{});
    else if (cnt == 1) run_rows(std::integral_constant<int, 1>{});
    else run_rows(std::integral_constant<int, 0>{});
    // bias partial: the 16 threads that staged the rows of one column group add up through LDS (fixed order)
    if (zblk == 0) {
        __syncthreads();                                    // the staging buffers are free
        float4* red = reinterpret_cast<float4*>(&sX[0][0]);
        red[grow * 16 + ggrp] = bsum;
        __syncthreads();
        if (tid < 16) {
            float4 t4 = red[tid];
            for (int g = 1; g < 16; g++) {
                const float4 u = red[g * 16 + tid];
                t4.x += u.x; t4.y += u.y; t4.z += u.z; t4.w += u.w;
            }
            const float o4[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (4 * tid + q < J.N) slab[(size_t)(4 * tid + q) * pitch + J.Kp] = o4[q];
        }
    }
}

__global__ __launch_bounds__(256) void dw_reduce_kernel(int nchunks, DwJobs jobs)
{
    __shared__ float part[4][64];
    const DwJob& J = jobs.j[blockIdx.y];
    const int pitch = 32 * ((J.Kp + 1 + 31) >> 5);
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;          // output of the block, chunk quarter
    const int idx = blockIdx.x * 64 + o;
    const bool split = (int)blockIdx.y == jobs.sp.job;             // uniform
    const DwSplit& S = jobs.sp;
    const int total = split ? (S.n[0] + S.n[1] + S.n[2]) * (S.K + 1) : J.N * (J.K + 1);
    if (blockIdx.x * 64 >= total) return;                          // uniform
    int n, k, kpad = J.Kp, Kout = J.K, nout = 0, part_h = 0;       // kpad: the bias column unless k < Kout
    if (split) {
        int r = min(idx, total - 1) / (S.K + 1);                   // output row among the heads' rows, head after head
        k = min(idx, total - 1) % (S.K + 1);
        while (part_h < 2 && r >= S.n[part_h]) { r -= S.n[part_h]; part_h++; }
        nout = r; n = S.n0[part_h] + r; Kout = S.K;
        if (k < Kout) kpad = S.k0[part_h] + k;
    } else {
        n = min(idx, total - 1) / (J.K + 1); k = min(idx, total - 1) % (J.K + 1);
        nout = n;
        if (k < J.K) {
            int kl = k, sg = 0;
            kpad = 0;
            while (sg < J.x_nseg - 1 && kl >= J.x_len[sg]) { kl -= J.x_len[sg]; kpad += pad4(J.x_len[sg]); sg++; }
            kpad += kl;
        }
    }
    const int per = (nchunks + 3) / 4;
    const int cbeg = q * per, cend = min(nchunks, cbeg + per);
    const float* __restrict__ src = J.slabs + (size_t)n * pitch + kpad;
    const size_t cstride = (size_t)64 * pitch;
    float s = 0.f;
    for (int c = cbeg; c < cend; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[(size_t)min(c + u, cend - 1) * cstride];
#pragma unroll
        for (int u = 0; u < 8; u++) s += (c + u < cend) ? v[u] : 0.f;
    }
    part[q][o] = s;
    __syncthreads();
    if (q == 0 && idx < total) {
        const float t = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
        float* db = split ? S.db[part_h] : J.db;
        float* dW = split ? S.dW[part_h] : J.dW;
        if (k == Kout) db[nout] += t;
        else dW[(size_t)nout * Kout + k] += t;
    }
}

// ------------------------------------------------------------------ residual outputs
struct OutArgs { int P, no_dx, no_ds, no_dr, no_dshs, apply_rotation; };

// 16 lanes per point: lanes 0..11 one float4 of the 48 SH outputs each, lane 12 xyz, 13 scales, 14 the rotation
__global__ __launch_bounds__(256) void outputs_forward_kernel(OutArgs a, const float* __restrict__ O,
                                                              const float* __restrict__ xyz,
                                                              const float* __restrict__ scales,
                                                              const float* __restrict__ rots,
                                                              const float* __restrict__ shs, float* __restrict__ o_xyz,
                                                              float* __restrict__ o_scales, float* __restrict__ o_rots,
                                                              float* __restrict__ o_shs)
{
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t p = gid >> 4;
    if (p >= (size_t)a.P) return;
    const int c = (int)(gid & 15);
    const float* o = O + p * OUT_LD;
    if (c < 12) {
        float4 v = reinterpret_cast<const float4*>(shs + 48 * p)[c];
        if (!a.no_dshs) {
            const float4 dv = reinterpret_cast<const float4*>(o + OC_SH)[c];
            v.x += dv.x; v.y += dv.y; v.z += dv.z; v.w += dv.w;
        }
        reinterpret_cast<float4*>(o_shs + 48 * p)[c] = v;
    } else if (c == 12) {
#pragma unroll
        for (int k = 0; k < 3; k++) o_xyz[3 * p + k] = a.no_dx ? xyz[3 * p + k] : xyz[3 * p + k] + o[OC_X + k];
    } else if (c == 13) {
#pragma unroll
        for (int k = 0; k < 3; k++) o_scales[3 * p + k] = a.no_ds ? scales[3 * p + k] : scales[3 * p + k] + o[OC_S + k];
    } else if (c == 14) {
        const float4 q1 = reinterpret_cast<const float4*>(rots)[p];
        float4 r;
        if (a.no_dr) {
            r = q1;
        } else {
            const float4 q2 = *reinterpret_cast<const float4*>(o + OC_R);
            if (!a.apply_rotation) {
                r = make_float4(q1.x + q2.x, q1.y + q2.y, q1.z + q2.z, q1.w + q2.w);
            } else {
                float q[4];
                q[0] = q1.x * q2.x - q1.y * q2.y - q1.z * q2.z - q1.w * q2.w;
                q[1] = q1.x * q2.y + q1.y * q2.x + q1.z * q2.w - q1.w * q2.z;
                q[2] = q1.x * q2.z - q1.y * q2.w + q1.z * q2.x + q1.w * q2.y;
                q[3] = q1.x * q2.w + q1.y * q2.z - q1.z * q2.y + q1.w * q2.x;
                const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                r = make_float4(q[0] / nrm, q[1] / nrm, q[2] / nrm, q[3] / nrm);
            }
        }
        reinterpret_cast<float4*>(o_rots)[p] = r;
    }
}

// lane per point: dL/d(outputs) -> dO (58 used columns, rest zero) and the pass-through input gradients
// (a 16-lanes-per-point float4 version like the forward measured slower: 40 us vs 29 us)
__global__ __launch_bounds__(256) void outputs_backward_kernel(OutArgs a, const float* __restrict__ O,
                                                               const float* __restrict__ rots,
                                                               const float* __restrict__ g_xyz,
                                                               const float* __restrict__ g_scales,
                                                               const float* __restrict__ g_rots,
                                                               const float* __restrict__ g_shs, float* __restrict__ dO,
                                                               float* __restrict__ dxyz, float* __restrict__ dscales,
                                                               float* __restrict__ drots, float* __restrict__ dshs)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (size_t)a.P) return;
    float* d = dO + p * OUT_LD;
    const float* o = O + p * OUT_LD;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float g = g_xyz[3 * p + c];
        dxyz[3 * p + c] = g;
        d[OC_X + c] = a.no_dx ? 0.f : g;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float g = g_scales[3 * p + c];
        dscales[3 * p + c] = g;
        d[OC_S + c] = a.no_ds ? 0.f : g;
    }
    const float4 gq = reinterpret_cast<const float4*>(g_rots)[p];
    float dq1[4], dq2[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.no_dr) {
        dq1[0] = gq.x; dq1[1] = gq.y; dq1[2] = gq.z; dq1[3] = gq.w;
    } else if (!a.apply_rotation) {
        dq1[0] = dq2[0] = gq.x; dq1[1] = dq2[1] = gq.y; dq1[2] = dq2[2] = gq.z; dq1[3] = dq2[3] = gq.w;
    } else {
        const float4 q1 = reinterpret_cast<const float4*>(rots)[p];
        const float4 q2 = make_float4(o[OC_R], o[OC_R + 1], o[OC_R + 2], o[OC_R + 3]);
        float q[4];
        q[0] = q1.x * q2.x - q1.y * q2.y - q1.z * q2.z - q1.w * q2.w;
        q[1] = q1.x * q2.y + q1.y * q2.x + q1.z * q2.w - q1.w * q2.z;
        q[2] = q1.x * q2.z - q1.y * q2.w + q1.z * q2.x + q1.w * q2.y;
        q[3] = q1.x * q2.w + q1.y * q2.z - q1.z * q2.y + q1.w * q2.x;
        const float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        const float inv = 1.0f / sqrtf(n2);
        const float g[4] = {gq.x, gq.y, gq.z, gq.w};
        const float dot = (g[0] * q[0] + g[1] * q[1] + g[2] * q[2] + g[3] * q[3]) * inv * inv * inv;
        float dqv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) dqv[k] = g[k] * inv - q[k] * dot;   // through q / |q|
        // through the quaternion product (w,x,y,z) = q1 * q2
        dq1[0] = dqv[0] * q2.x + dqv[1] * q2.y + dqv[2] * q2.z + dqv[3] * q2.w;
        dq1[1] = -dqv[0] * q2.y + dqv[1] * q2.x - dqv[2] * q2.w + dqv[3] * q2.z;
        dq1[2] = -dqv[0] * q2.z + dqv[1] * q2.w + dqv[2] * q2.x - dqv[3] * q2.y;
        dq1[3] = -dqv[0] * q2.w - dqv[1] * q2.z + dqv[2] * q2.y + dqv[3] * q2.x;
        dq2[0] = dqv[0] * q1.x + dqv[1] * q1.y + dqv[2] * q1.z + dqv[3] * q1.w;
        dq2[1] = -dqv[0] * q1.y + dqv[1] * q1.x + dqv[2] * q1.w - dqv[3] * q1.z;
        dq2[2] = -dqv[0] * q1.z - dqv[1] * q1.w + dqv[2] * q1.x + dqv[3] * q1.y;
        dq2[3] = -dqv[0] * q1.w + dqv[1] * q1.z - dqv[2] * q1.y + dqv[3] * q1.x;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        drots[4 * p + k] = dq1[k];
        d[OC_R + k] = dq2[k];
    }
    for (int k = 0; k < 48; k++) {
        const float g = g_shs[48 * p + k];
        dshs[48 * p + k] = g;
        d[OC_SH + k] = a.no_dshs ? 0.f : g;
    }
#pragma unroll
    for (int k = 3; k < 8; k++) { d[OC_X + k] = 0.f; d[OC_S + k] = 0.f; }
#pragma unroll
    for (int k = 4; k < 8; k++) d[OC_R + k] = 0.f;
}

// d(ft_w), d(ft_b) from the per-workgroup partial sums of hexplane_coord_reduce_kernel
// (t' = w0*t + w1*f + b): single workgroup, fixed-order sum
__global__ __launch_bounds__(256) void ft_embed_final_kernel(int nparts, const float* __restrict__ ft_partial,
                                                             float* __restrict__ d_w, float* __restrict__ d_b)
{
    __shared__ float s[256][3];
    float a[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nparts; i += 256)
        for (int k = 0; k < 3; k++) a[k] += ft_partial[3 * (size_t)i + k];
    for (int k = 0; k < 3; k++) s[threadIdx.x][k] = a[k];
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 3; k++) s[threadIdx.x][k] += s[threadIdx.x + off][k];
        __syncthreads();
    }
    if (threadIdx.x == 0) { d_w[0] += s[0][0]; d_w[1] += s[0][1]; d_b[0] += s[0][2]; }
}

#include "deform_heads.h"
#include "deform_heads_bf16.h"

// ---- host side of the fused heads (deform_heads.h)
static bool heads_layered_forced()
{
    static const bool v = [] { const char* e = getenv("GS4D_DEFORM_HEADS"); return e && e[0] == 'l'; }();   // "layered": round 1's path
    return v;
}

static void heads_common(HeadsArgs& a, const gs4d_deform_desc* d, const Layout& l, int P)
{
    a = {};
    a.P = P; a.ldz = l.ldz; a.nunits = (P + 31) / 32;
    for (int h = 0; h < NHEAD; h++) {
        a.W1[h] = d->head_w1[h]; a.b1[h] = d->head_b1[h]; a.W2[h] = d->head_w2[h]; a.b2[h] = d->head_b2[h];
        a.in_w[h] = l.head_in[h];
        a.seg_beg[h] = l.head_seg[h] < 0 ? 0 : l.head_seg[h];
        a.seg_len[h] = l.head_seglen[h];
        a.seg_blocks[h] = pad8(l.head_seglen[h]) / 8;
    }
    a.oT = l.oT; a.e = l.e; a.tf_blocks = pad8(l.e) / 8;
    a.no_dx = d->no_dx; a.no_ds = d->no_ds; a.no_dr = d->no_dr; a.no_dshs = d->no_dshs; a.apply_rotation = d->apply_rotation;
    a.pos_pe = d->pos_pe; a.sr_pe = d->sr_pe;
}

#define HEADS_LDS_LIMIT (160 * 1024 - 4096)      // dynamic LDS a fused-heads kernel may ask for (the backward also has a 2 KB static table)

// forward layout (floats): W1 of the four heads | W2 pos, scales, rotations (512 each) | W2 shs (3072) | b1 (256) | b2 (112)
static size_t heads_forward_layout(HeadsArgs& a)
{
    int off = 0;
    for (int h = 0; h < NHEAD; h++) { a.w1_off[h] = off; off += (8 + a.seg_blocks[h] + a.tf_blocks) * 8 * WIDTH; }
    for (int h = 0; h < 3; h++) { a.w2_off[h] = off; off += 8 * 2 * 8 * 4; }
    a.w2_off[3] = off; off += 8 * 2 * 48 * 4;
    a.b1_off = off; off += NHEAD * WIDTH;
    a.b2_off = off; off += 64 + 48;
    return (size_t)off * sizeof(float);
}

// GS4D_MFMA=bf16x3 (environment, read once): the opt-in split-operand kernels of deform_heads_bf16.h; anything else,
// or nothing, is the exact-f32 default
static bool mfma_bf16x3()
{
    static const bool v = [] { const char* e = getenv("GS4D_MFMA"); return e && (e[0] == 'b' || e[0] == 'B'); }();
    return v;
}

// the same regions with every group of K blocks (hidden / segment / embedding) padded to whole K PAIRS
static size_t heads_forward_layout_bf16(HeadsArgs& a)
{
    int off = 0;
    for (int h = 0; h < NHEAD; h++) { a.w1_off[h] = off; off += (8 + hb_even(a.seg_blocks[h]) + hb_even(a.tf_blocks)) * 8 * WIDTH; }
    for (int h = 0; h < 3; h++) { a.w2_off[h] = off; off += 8 * 2 * 8 * 4; }
    a.w2_off[3] = off; off += 8 * 2 * 48 * 4;
    a.b1_off = off; off += NHEAD * WIDTH;
    a.b2_off = off; off += 64 + 48;
    return (size_t)off * sizeof(float);
}

// backward layout (floats): W2^T of the four heads (1, 1, 1, 6 output blocks of 512) | W1^T hidden parts (4 x 4096) |
// W1^T segment parts: pos (64 rows), scales (32 rows), rotations (32 rows)
static size_t heads_backward_layout(HeadsArgs& a)
{
    int off = 0;
    for (int h = 0; h < NHEAD; h++) { a.w2_off[h] = off; off += (h == 3 ? 6 : 1) * 8 * WIDTH; }
    for (int h = 0; h < NHEAD; h++) { a.w1_off[h] = off; off += WIDTH * WIDTH; }
    a.seg_off[0] = off; off += WIDTH * 64;
    a.seg_off[1] = off; off += WIDTH * 32;
    a.seg_off[2] = off; off += WIDTH * 32;
    return (size_t)off * sizeof(float);
}

static bool heads_fused_ok(const gs4d_deform_desc* d, const Layout& l)
{
    if (heads_layered_forced()) return false;
    HeadsArgs a;
    heads_common(a, d, l, 1);
    for (int h = 0; h < NHEAD; h++)
        if (a.seg_blocks[h] > HF_MAX_SEG_BLOCKS) return false;
    if (a.tf_blocks > HF_MAX_TF_BLOCKS) return false;
    if (a.seg_len[1] > 32 || a.seg_len[2] > 32) return false;      // backward: one 32-row tile per small segment
    if (mfma_bf16x3() && heads_forward_layout_bf16(a) > HEADS_LDS_LIMIT) return false;
    return heads_forward_layout(a) <= HEADS_LDS_LIMIT && heads_backward_layout(a) <= HEADS_LDS_LIMIT;
}

static int heads_debug_units(int nunits)
{
    static const int v = [] { const char* e = getenv("GS4D_HF_DEBUG_UNITS"); return e ? atoi(e) : -1; }();   // experiments only
    return v >= 0 && v < nunits ? v : nunits;
}

static int heads_grid(int nunits)
{
    // one persistent workgroup per CU (its weights fill most of the LDS); never more workgroups than units
    return nunits < GEMM_CUS ? nunits : GEMM_CUS;
}

static int launch_heads_forward(HeadsArgs& a, hipStream_t s)
{
    const bool split = mfma_bf16x3();
    const size_t lds = split ? heads_forward_layout_bf16(a) : heads_forward_layout(a);
    auto kern = split ? heads_forward_bf16x3_kernel : heads_forward_kernel;
    static size_t attr[2] = {0, 0};              // the largest dynamic LDS size announced so far, per kernel
    if (lds > attr[split]) {
        GS4D_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr[split] = lds;
    }
    const int grid = heads_grid(a.nunits);
    a.nunits = heads_debug_units(a.nunits);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(HFF_THREADS), lds, s, a);
    GS4D_LAUNCH_CHECK("deform_heads_forward", 0, s);
    return GS4D_OK;
}

static int launch_heads_backward(HeadsArgs& a, hipStream_t s)
{
    const size_t lds = heads_backward_layout(a);
    static size_t attr = 0;
    if (lds > attr) {
        GS4D_HIP_CHECK(hipFuncSetAttribute((const void*)heads_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    const int grid = heads_grid(a.nunits);
    a.nunits = heads_debug_units(a.nunits);
    hipLaunchKernelGGL(heads_backward_kernel, dim3(grid), dim3(HF_THREADS), lds, s, a);
    GS4D_LAUNCH_CHECK("deform_heads_backward", 0, s);
    return GS4D_OK;
}

static int validate(const gs4d_deform_desc* d, int P)
{
    if (!d || P < 0) { gs4d_set_error("deform: null descriptor or negative P"); return GS4D_EINVAL; }
    if (d->C < 4 || d->C > 256 || (d->C & (d->C - 1))) { gs4d_set_error("deform: plane channels must be a power of two in [4,256], got %d", d->C); return GS4D_EINVAL; }
    if ((d->L * d->C) % 8) { gs4d_set_error("deform: levels * channels must be a multiple of 8, got %d", d->L * d->C); return GS4D_EINVAL; }
    if (d->L < 1 || d->L > GS4D_MAX_LEVELS) { gs4d_set_error("deform: levels must be 1..%d", GS4D_MAX_LEVELS); return GS4D_EINVAL; }
    if (d->trunk_layers < 1 || d->trunk_layers > GS4D_MAX_TRUNK) { gs4d_set_error("deform: trunk_layers must be 1..%d", GS4D_MAX_TRUNK); return GS4D_EINVAL; }
    if (d->pos_pe < 0 || d->pos_pe > 10 || d->sr_pe < 0 || d->sr_pe > 8 || d->time_pe < 0 || d->time_pe > 15 || d->force_pe < 0 || d->force_pe > 15) {
        gs4d_set_error("deform: unsupported PE sizes (one head input must fit 64 + 60 columns per segment)");
        return GS4D_EINVAL;
    }
    return GS4D_OK;
}

static size_t gemm_lds_bytes(int Kp) { return (size_t)Kp * 64 * sizeof(float); }

static int launch_gemm(int P, GemmJobs& jobs, int njobs, hipStream_t s)
{
    int Kpmax = 0;
    for (int i = 0; i < njobs; i++) {
        const GemmJob& j = jobs.j[i];
        const int kp = gemm_kp(j);
        Kpmax = kp > Kpmax ? kp : Kpmax;
        bool ok = ((uintptr_t)j.A & 15) == 0 && (j.lda & 3) == 0 && kp <= 8 * GEMM_MAX_KB && j.N >= 1 && j.N <= 64;
        for (int sg = 0; sg < j.a_nseg; sg++) ok = ok && (j.a_beg[sg] & 3) == 0 && j.a_beg[sg] + pad8(j.a_len[sg]) <= j.lda;
        if (!ok) { gs4d_set_error("deform gemm: operand %d violates the 16-byte / pad-8 layout rules", i); return GS4D_EINVAL; }
    }
    // RELU, MASK and FUSE are per-launch properties (every job of a launch agrees)
    const bool relu = jobs.j[0].relu_a != 0, mask = jobs.j[0].mask != nullptr, fuse = jobs.j[0].W2 != nullptr;
    for (int i = 0; i < njobs; i++) {
        const GemmJob& j = jobs.j[i];
        if ((j.relu_a != 0) != relu || (j.mask != nullptr) != mask || (j.W2 != nullptr) != fuse) {
            gs4d_set_error("deform gemm: jobs of one launch must agree on ReLU, mask and fusion");
            return GS4D_EINVAL;
        }
        if (fuse && (!relu || mask || j.N != WIDTH || !j.bias2 || !j.Y2 || j.N2 < 1 || j.N2 > 64)) {
            gs4d_set_error("deform gemm: bad fused job %d", i);
            return GS4D_EINVAL;
        }
    }
    // every job K = 64 contiguous columns: the dedicated kernel (GS4D_DEFORM_GEMM=general keeps the general one)
    bool k64 = !fuse;
    for (int i = 0; i < njobs; i++) k64 = k64 && jobs.j[i].a_nseg == 1 && jobs.j[i].a_len[0] == 64 && jobs.j[i].b_nseg == 1;
    static const bool force_general = [] { const char* e = getenv("GS4D_DEFORM_GEMM"); return e && !strcmp(e, "general"); }();
    if (k64 && !force_general) {
        void (*k)(int, GemmJobs) = relu ? (mask ? gemm64_rows_kernel<true, true> : gemm64_rows_kernel<true, false>)
                                        : (mask ? gemm64_rows_kernel<false, true> : gemm64_rows_kernel<false, false>);
        // persistent grid: 2 workgroups (8 waves: the kernel holds 172-224 VGPRs) per CU shared between the jobs, at most one wave per 32-row tile
        const int ntile = (P + 31) / 32;
        int per = (GEMM_CUS * 2 + njobs - 1) / njobs;
        if (per > (ntile + 3) / 4) per = (ntile + 3) / 4;
        if (per < 1) per = 1;
        jobs.wg_begin[0] = 0;
        for (int i = 0; i < 8; i++) jobs.wg_begin[i + 1] = jobs.wg_begin[i] + (i < njobs ? per : 0);
        hipLaunchKernelGGL(k, dim3(jobs.wg_begin[8]), dim3(256), 0, s, P, jobs);
        GS4D_LAUNCH_CHECK("deform_gemm64", 0, s);
        return GS4D_OK;
    }
    const size_t lds = gemm_lds_bytes(Kpmax) + (fuse ? sizeof(float) * (64 * 64 + 4 * 32 * FUSE_TP) : 0);
    void (*kern)(int, GemmJobs) = fuse ? gemm_rows_kernel<true, false, true>
                                  : relu ? (mask ? gemm_rows_kernel<true, true, false> : gemm_rows_kernel<true, false, false>)
                                         : (mask ? gemm_rows_kernel<false, true, false> : gemm_rows_kernel<false, false, false>);
    static bool attr_set[5] = {false, false, false, false, false};
    const int variant = fuse ? 4 : relu * 2 + mask;
    if (!attr_set[variant]) {
        GS4D_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
        attr_set[variant] = true;
    }
    // persistent 1-D grid, ~4 workgroups per CU in total (mild oversubscription measured faster than an exactly
    // resident grid), shared between the jobs in proportion to their MFMA work per row block
    const int nrb = (P + ROWS_PER_WG - 1) / ROWS_PER_WG;
    const int total = GEMM_CUS * 4;
    double wsum = 0, w[8];
    for (int i = 0; i < njobs; i++) {
        w[i] = (gemm_kp(jobs.j[i]) / 8) * 2 + 6 + (fuse ? 8 * (jobs.j[i].N2 > 32 ? 2 : 1) + 4 : 0);
        wsum += w[i];
    }
    jobs.wg_begin[0] = 0;
    for (int i = 0; i < 8; i++) {
        int n = 0;
        if (i < njobs) {
            n = (int)(total * w[i] / wsum);
            n = n < 1 ? 1 : (n > nrb ? nrb : n);
        }
        jobs.wg_begin[i + 1] = jobs.wg_begin[i] + n;
    }
    hipLaunchKernelGGL(kern, dim3(jobs.wg_begin[8]), dim3(256), lds, s, P, jobs);
    GS4D_LAUNCH_CHECK("deform_gemm", 0, s);
    return GS4D_OK;
}

static GemmJob simple_job(const float* A, int lda, int acol0, int K, int relu_a, const float* W, int ldw, int wcol0,
                          int transposed, const float* bias, int N, float* Y, int ldy, int ycol0)
{
    GemmJob j = {};
    j.A = A; j.lda = lda; j.a_nseg = 1; j.a_beg[0] = acol0; j.a_len[0] = K; j.K = K; j.relu_a = relu_a;
    j.b_nseg = 1; j.B[0] = W; j.b_ld[0] = ldw; j.b_rows[0] = K; j.b_col0[0] = wcol0; j.b_transposed = transposed;
    j.bias = bias; j.N = N; j.Y = Y; j.ldy = ldy; j.ycol0 = ycol0; j.mask = nullptr;
    return j;
}

}  // namespace

extern "C" {

size_t gs4d_deform_workspace_bytes(const gs4d_deform_desc* d, int P)
{
    if (validate(d, P)) return 0;
    const Layout l = make_layout(d);
    size_t bytes = 0;
    carve_workspace(nullptr, d, l, (size_t)P, &bytes);
    return bytes;
}

static void carve_scratch(char* base, const gs4d_deform_desc* d, const Layout& l, size_t P, float** dO, float** dH1,
                          float** dZ, float** dT, float** dfeat, float** dxyz_hex, float** dtprime, float** gsix,
                          int** hist, int** rank, int** key, float4** rec, float** slabs, size_t* slab_floats_per_job, size_t* bytes)
{
    char* p = base;
    gs4d_obtain(p, *dO, P * OUT_LD);
    gs4d_obtain(p, *dH1, P * NHEAD * WIDTH);
    gs4d_obtain(p, *dZ, P * l.ldz);
    gs4d_obtain(p, *dT, P * WIDTH * (size_t)(d->trunk_layers > 1 ? d->trunk_layers - 1 : 1));   // one buffer per inner trunk layer (all stay live for the single weight-gradient launch)
    gs4d_obtain(p, *dfeat, P * l.F);
    gs4d_obtain(p, *dxyz_hex, P * 4);                  // (dx,dy,dz,dt') per point, normalised coordinates
    gs4d_obtain(p, *dtprime, 3 * ((P + 255) / 256) + 3);   // per-workgroup partial sums for the t' embedding
    const size_t HB = gs4d_hex_batch_points(d, P);
    gs4d_obtain(p, *gsix, HB * d->L * 6 * d->C);           // per-plane sample gradients of one batch, point kernel -> reduce kernel
    gs4d_obtain(p, *hist, (size_t)d->L * HEX_SORTS * gs4d_hex_bins(d).nbins);
    gs4d_obtain(p, *rank, (size_t)d->L * HEX_SORTS * HB);
    gs4d_obtain(p, *key, (size_t)d->L * HEX_SORTS * HB);
    gs4d_obtain(p, *rec, (size_t)d->L * HEX_SORTS * HB);
    const size_t nchunks = (P + DW_ROWS_PER_WG - 1) / DW_ROWS_PER_WG;
    int kmax = l.F > l.head_in[0] ? l.F : l.head_in[0];
    for (int h = 0; h < NHEAD; h++) kmax = l.head_in[h] > kmax ? l.head_in[h] : kmax;
    kmax = 3 * WIDTH > kmax ? 3 * WIDTH : kmax;                        // the block-diagonal job of the small heads (DwSplit)
    const size_t pitch = (size_t)((kmax + 12 + 1 + 31) / 32) * 32;     // + the 4-column padding of up to three segments
    *slab_floats_per_job = nchunks * 64 * pitch;
    gs4d_obtain(p, *slabs, *slab_floats_per_job * (2 * NHEAD + GS4D_MAX_TRUNK));
    if (bytes) *bytes = (size_t)(p - base) + 256;
}

size_t gs4d_deform_backward_scratch_bytes(const gs4d_deform_desc* d, int P)
{
    if (validate(d, P)) return 0;
    const Layout l = make_layout(d);
    float *a, *b, *c, *e, *f, *g, *h, *gt, *s;
    int *hi, *rk, *ky;
    float4* rc;
    size_t per = 0, bytes = 0;
    carve_scratch(nullptr, d, l, (size_t)P, &a, &b, &c, &e, &f, &g, &h, &gt, &hi, &rk, &ky, &rc, &s, &per, &bytes);
    return bytes;
}

int gs4d_deform_forward(const gs4d_deform_desc* d, int P, const float* points, const float* scales,
                        const float* rotations, const float* shs, const float* time, const float* force,
                        float* out_means3D, float* out_scales, float* out_rotations, float* out_shs, char* workspace,
                        void* stream)
{
    GS4D_TRACE_SCOPE("gs4d_deform_forward");
    hipStream_t s = (hipStream_t)stream;
    int rc = validate(d, P);
    if (rc) return rc;
    if (P == 0) return GS4D_OK;
    if (!points || !scales || !rotations || !shs || !time || !force || !out_means3D || !out_scales || !out_rotations ||
        !out_shs || !workspace || ((uintptr_t)workspace & 255) || ((uintptr_t)rotations & 15) || ((uintptr_t)shs & 15) ||
        ((uintptr_t)out_shs & 15) || ((uintptr_t)out_rotations & 15)) {
        gs4d_set_error("gs4d_deform_forward: null or misaligned pointer");
        return GS4D_EINVAL;
    }
    const Layout l = make_layout(d);
    Workspace w = carve_workspace(workspace, d, l, (size_t)P, nullptr);

    {
        GS4D_PROF("hexplane_forward", s);
        rc = gs4d_hex_launch_forward(d, P, points, time, force, w.tprime, w.feat, l.F, s);
        if (rc) return rc;
    }
    GS4D_PROF("mlp_forward", s);
    PeArgs pa = {P, l.ldz, l.oX, l.oS, l.oR, l.oT, l.oF, l.Kz, d->pos_pe, d->sr_pe, d->time_pe, d->force_pe,
                 d->use_time, d->use_force};
    {
        const size_t lds = sizeof(float) * PE_PTS * (size_t)(l.ldz - l.oX);     // 14 KB at the mutant shape
        if (lds > 48 * 1024 || l.ldz - l.oX > PE_MAX_SLOTS) { gs4d_set_error("deform forward: %d positional-encoding columns exceed the LDS tile", l.ldz - l.oX); return GS4D_EINVAL; }
        hipLaunchKernelGGL(pe_forward_kernel, dim3((unsigned)(((size_t)P + PE_PTS - 1) / PE_PTS)), dim3(256), lds, s, pa,
                           points, scales, rotations, time, force, w.Z);
        GS4D_LAUNCH_CHECK("pe_forward", 0, s);
    }
    // trunk
    for (int i = 0; i < d->trunk_layers; i++) {
        GemmJobs jobs = {};
        const bool last = i + 1 == d->trunk_layers;
        const float* A = i == 0 ? w.feat : w.trunk[i - 1];
        const int K = i == 0 ? l.F : WIDTH;
        jobs.j[0] = simple_job(A, K, 0, K, i > 0, d->trunk_w[i], K, 0, 1, d->trunk_b[i], WIDTH,
                               last ? w.Z : w.trunk[i], last ? l.ldz : WIDTH, 0);
        rc = launch_gemm(P, jobs, 1, s);
        if (rc) return rc;
    }
    if (heads_fused_ok(d, l)) {
        // all four heads, both Linears and the residual / quaternion outputs in ONE pass over the points (deform_heads.h)
        HeadsArgs ha;
        heads_common(ha, d, l, P);
        ha.Z = w.Z; ha.H1 = w.H1; ha.O = w.O; ha.hmask = w.hmask; ha.store = d->no_grad ? 0 : 1;
        ha.xyz = points; ha.scales = scales; ha.rots = rotations; ha.shs = shs;
        ha.o_xyz = out_means3D; ha.o_scales = out_scales; ha.o_rots = out_rotations; ha.o_shs = out_shs;
        return launch_heads_forward(ha, s);
    }
    // heads, both Linears in one pass: H1_h = relu([hidden | pe_h | tf]) W1_h^T + b1_h (stored only when a backward
    // will read it), O[:, ocol_h ..] = relu(H1_h) W2_h^T + b2_h
    {
        GemmJobs jobs = {};
        for (int h = 0; h < NHEAD; h++) {
            GemmJob& j = jobs.j[h];
            j = simple_job(w.Z, l.ldz, 0, l.head_in[h], 1, d->head_w1[h], l.head_in[h], 0, 1, d->head_b1[h], WIDTH,
                           w.H1, NHEAD * WIDTH, h * WIDTH);
            j.a_nseg = 0;
            j.a_beg[j.a_nseg] = 0; j.a_len[j.a_nseg++] = WIDTH;
            if (l.head_seglen[h] > 0) { j.a_beg[j.a_nseg] = l.head_seg[h]; j.a_len[j.a_nseg++] = l.head_seglen[h]; }
            j.a_beg[j.a_nseg] = l.oT; j.a_len[j.a_nseg++] = l.e;
            j.W2 = d->head_w2[h]; j.bias2 = d->head_b2[h]; j.N2 = l.head_out[h];
            j.Y2 = w.O; j.ldy2 = OUT_LD; j.y2col0 = l.head_ocol[h];
            j.store_y = d->no_grad ? 0 : 1;
        }
        rc = launch_gemm(P, jobs, NHEAD, s);
        if (rc) return rc;
    }
    OutArgs oa = {P, d->no_dx, d->no_ds, d->no_dr, d->no_dshs, d->apply_rotation};
    {
        const size_t threads = (size_t)P * 16;
        hipLaunchKernelGGL(outputs_forward_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, oa, w.O,
                           points, scales, rotations, shs, out_means3D, out_scales, out_rotations, out_shs);
        GS4D_LAUNCH_CHECK("deform_outputs_forward", 0, s);
    }
    return GS4D_OK;
}

int gs4d_deform_backward_phases(const gs4d_deform_desc* d, const gs4d_deform_grads* g, int P, const float* points,
                                const float* scales, const float* rotations, const float* time, const float* force,
                                const float* dL_dmeans3D, const float* dL_dscales, const float* dL_drotations,
                                const float* dL_dshs, float* dL_dpoints, float* dL_dscales_in, float* dL_drotations_in,
                                float* dL_dshs_in, char* workspace, char* scratch, void* stream, int phases)
{
    GS4D_TRACE_SCOPE("gs4d_deform_backward_phases");
    hipStream_t s = (hipStream_t)stream;
    int rc = validate(d, P);
    if (rc) return rc;
    if (P == 0) return GS4D_OK;
    if (phases < 1 || phases > (GS4D_DEFORM_BWD_DATA | GS4D_DEFORM_BWD_WEIGHTS)) {
        gs4d_set_error("gs4d_deform_backward_phases: phases must be a combination of GS4D_DEFORM_BWD_DATA and _WEIGHTS");
        return GS4D_EINVAL;
    }
    const bool do_data = (phases & GS4D_DEFORM_BWD_DATA) != 0, do_weights = (phases & GS4D_DEFORM_BWD_WEIGHTS) != 0;
    if (!g || !points || !scales || !rotations || !time || !force || !dL_dmeans3D || !dL_dscales || !dL_drotations ||
        !dL_dshs || !dL_dpoints || !dL_dscales_in || !dL_drotations_in || !dL_dshs_in || !workspace || !scratch ||
        ((uintptr_t)workspace & 255) || ((uintptr_t)scratch & 255) || ((uintptr_t)rotations & 15) ||
        ((uintptr_t)dL_drotations & 15) || ((uintptr_t)dL_dshs & 15) || ((uintptr_t)dL_dshs_in & 15) ||
        ((uintptr_t)dL_drotations_in & 15)) {
        gs4d_set_error("gs4d_deform_backward: null or misaligned pointer");
        return GS4D_EINVAL;
    }
    const Layout l = make_layout(d);
    Workspace w = carve_workspace(workspace, d, l, (size_t)P, nullptr);
    float *dO, *dH1, *dZ, *dT, *dfeat, *dxyz_hex, *dtprime, *gsix, *slabs;
    int *hist, *rank, *key;
    float4* rec;
    size_t slab_per_job = 0;
    carve_scratch(scratch, d, l, (size_t)P, &dO, &dH1, &dZ, &dT, &dfeat, &dxyz_hex, &dtprime, &gsix, &hist, &rank,
                  &key, &rec, &slabs, &slab_per_job, nullptr);
    const int nchunks = (P + DW_ROWS_PER_WG - 1) / DW_ROWS_PER_WG;

    // Order: everything the INPUT and PLANE gradients need first (data phase), the weight-gradient products
    // last (weights phase), so that a data-parallel caller can start exchanging the former while the latter run.
    Gs4dProfScope* mlp_scope = do_data ? new Gs4dProfScope("mlp_backward", s) : nullptr;
    struct ScopeGuard { Gs4dProfScope*& p; ~ScopeGuard() { delete p; } } mlp_guard{mlp_scope};
    OutArgs oa = {P, d->no_dx, d->no_ds, d->no_dr, d->no_dshs, d->apply_rotation};
    const bool fused_heads = heads_fused_ok(d, l);
    if (do_data && fused_heads) {
        // outputs backward, dH1, dZ (hidden columns) and the PE segments' input gradients in ONE pass (deform_heads.h);
        // it also clears the plane backward's cell histograms
        HeadsArgs ha;
        heads_common(ha, d, l, P);
        ha.Z = w.Z; ha.H1 = w.H1; ha.O = w.O; ha.hmask = w.hmask;
        ha.xyz = points; ha.scales = scales; ha.rots = rotations;
        ha.g_xyz = dL_dmeans3D; ha.g_scales = dL_dscales; ha.g_rots = dL_drotations; ha.g_shs = dL_dshs;
        ha.dO = dO; ha.dH1 = dH1; ha.dZ = dZ;
        ha.dxyz = dL_dpoints; ha.dscales = dL_dscales_in; ha.drots = dL_drotations_in; ha.dshs = dL_dshs_in;
        rc = launch_heads_backward(ha, s);
        if (rc) return rc;
    }
    if (do_data && !fused_heads) {
        hipLaunchKernelGGL(outputs_backward_kernel, dim3((P + 255) / 256), dim3(256), 0, s, oa, w.O, rotations,
                           dL_dmeans3D, dL_dscales, dL_drotations, dL_dshs, dO, dL_dpoints, dL_dscales_in,
                           dL_drotations_in, dL_dshs_in);
        GS4D_LAUNCH_CHECK("deform_outputs_backward", 0, s);
    }

    // dH1_h = (dO_h W2_h) * (H1_h > 0)
    if (do_data && !fused_heads) {
        GemmJobs jobs = {};
        for (int h = 0; h < NHEAD; h++) {
            GemmJob& j = jobs.j[h];
            j = simple_job(dO, OUT_LD, l.head_ocol[h], l.head_out[h], 0, d->head_w2[h], WIDTH, 0, 0, nullptr, WIDTH, dH1,
                           NHEAD * WIDTH, h * WIDTH);
            j.mask = w.H1; j.ldm = NHEAD * WIDTH; j.mcol0 = h * WIDTH;
        }
        rc = launch_gemm(P, jobs, NHEAD, s);
        if (rc) return rc;
    }
    // dZ = (sum_h dH1_h W1_h scattered to Z columns) * (Z > 0): the hidden columns and the three PE segments
    if (do_data && !fused_heads) {
        GemmJobs jobs = {};
        int nj = 0;
        auto all_heads = [&](int zcol0, int ncols, int wcol_of_head[NHEAD]) {
            GemmJob& j = jobs.j[nj++];
            j = {};
            j.A = dH1; j.lda = NHEAD * WIDTH; j.a_nseg = 1; j.a_beg[0] = 0; j.a_len[0] = NHEAD * WIDTH;
            j.K = NHEAD * WIDTH; j.relu_a = 0; j.b_nseg = NHEAD; j.b_transposed = 0;
            for (int h = 0; h < NHEAD; h++) { j.B[h] = d->head_w1[h]; j.b_ld[h] = l.head_in[h]; j.b_rows[h] = WIDTH; j.b_col0[h] = wcol_of_head[h]; }
            j.bias = nullptr; j.N = ncols; j.Y = dZ; j.ldy = l.ldz; j.ycol0 = zcol0;
            j.mask = w.Z; j.ldm = l.ldz; j.mcol0 = zcol0;
        };
        int hid[NHEAD] = {0, 0, 0, 0};
        all_heads(0, WIDTH, hid);
        // the time / force embedding columns of dZ are never consumed (time and force are inputs, not
        // parameters, and t' reaches the planes through force_time_embed, not through these columns): not computed
        for (int h = 0; h < 3; h++) {
            if (l.head_seglen[h] == 0) continue;
            GemmJob& j = jobs.j[nj++];
            j = simple_job(dH1, NHEAD * WIDTH, h * WIDTH, WIDTH, 0, d->head_w1[h], l.head_in[h], WIDTH, 0, nullptr,
                           l.head_seglen[h], dZ, l.ldz, l.head_seg[h]);
            j.mask = w.Z; j.ldm = l.ldz; j.mcol0 = l.head_seg[h];
        }
        rc = launch_gemm(P, jobs, nj, s);
        if (rc) return rc;
    }
    // trunk backward: dT chain, ending in dfeat.  Every weight-gradient product of the network (trunk layers,
    // first and second Linear of each head) is collected into ONE dw_partial / dw_reduce launch pair below.
    DwJobs dj = {};
    dj.sp.job = -1;
    int ndw = 0, kbmax = 0, nkmax = 0;
    auto add_dw = [&](const DwJob& q) {
        kbmax = q.kblocks > kbmax ? q.kblocks : kbmax;
        nkmax = q.N * (q.K + 1) > nkmax ? q.N * (q.K + 1) : nkmax;
        dj.j[ndw++] = q;
    };
    const float* dcur = dZ;  // gradient w.r.t. the output of trunk layer i (columns 0..63)
    int dcur_ld = l.ldz;
    for (int i = d->trunk_layers - 1; i >= 0; i--) {
        GemmJobs jobs = {};
        const int K = i == 0 ? l.F : WIDTH;
        float* dst = i == 0 ? dfeat : dT + (size_t)(i - 1) * P * WIDTH;
        GemmJob& j = jobs.j[0];
        j = simple_job(dcur, dcur_ld, 0, WIDTH, 0, d->trunk_w[i], K, 0, 0, nullptr, K > 64 ? 64 : K, dst, K, 0);
        if (i > 0) { j.mask = w.trunk[i - 1]; j.ldm = WIDTH; j.mcol0 = 0; }
        int nj = 1;
        for (int c0 = 64; c0 < K; c0 += 64) {   // wide first layer: more 64-column groups
            GemmJob& jj = jobs.j[nj++];
            jj = j;
            jj.b_col0[0] = c0; jj.ycol0 = c0; jj.N = K - c0 > 64 ? 64 : K - c0;
        }
        if (do_data) {
            rc = launch_gemm(P, jobs, nj, s);
            if (rc) return rc;
        }
        DwJob q = {};
        q.G = dcur; q.ldg = dcur_ld; q.gcol0 = 0; q.N = WIDTH;
        q.X = i == 0 ? w.feat : w.trunk[i - 1]; q.ldx = K; q.x_nseg = 1; q.x_beg[0] = 0; q.x_len[0] = K; q.K = K;
        q.relu_x = i > 0;
        q.Kp = K; q.kblocks = (K + 31) / 32; q.slabs = slabs + slab_per_job * (2 * NHEAD + i);
        q.dW = g->trunk_w[i]; q.db = g->trunk_b[i];
        add_dw(q);
        dcur = dst;
        dcur_ld = K;
    }
    for (int h = 0; h < NHEAD; h++) {
        DwJob a = {};
        a.G = dH1; a.ldg = NHEAD * WIDTH; a.gcol0 = h * WIDTH; a.N = WIDTH;
        a.X = w.Z; a.ldx = l.ldz; a.x_nseg = 0;
        a.x_beg[a.x_nseg] = 0; a.x_len[a.x_nseg++] = WIDTH;
        if (l.head_seglen[h] > 0) { a.x_beg[a.x_nseg] = l.head_seg[h]; a.x_len[a.x_nseg++] = l.head_seglen[h]; }
        a.x_beg[a.x_nseg] = l.oT; a.x_len[a.x_nseg++] = l.e;
        a.K = l.head_in[h]; a.relu_x = 1;
        a.Kp = 0;
        for (int sg = 0; sg < a.x_nseg; sg++) a.Kp += pad4(a.x_len[sg]);
        a.kblocks = (a.Kp + 31) / 32;
        a.slabs = slabs + slab_per_job * h; a.dW = g->head_w1[h]; a.db = g->head_b1[h];
        add_dw(a);
        if (h < 3) continue;                 // the small heads' second Linears: one block-diagonal job, below
        DwJob b = {};
        b.G = dO; b.ldg = OUT_LD; b.gcol0 = l.head_ocol[h]; b.N = l.head_out[h];
        if (fused_heads && !d->no_dshs) { b.G = dL_dshs; b.ldg = 48; b.gcol0 = 0; }   // dO's SH columns ARE the upstream gradient: not copied
        b.X = w.H1; b.ldx = NHEAD * WIDTH; b.x_nseg = 1; b.x_beg[0] = h * WIDTH; b.x_len[0] = WIDTH; b.K = WIDTH;
        b.relu_x = 1; b.Kp = WIDTH; b.kblocks = (WIDTH + 31) / 32;
        b.slabs = slabs + slab_per_job * (NHEAD + h); b.dW = g->head_w2[h]; b.db = g->head_b2[h];
        add_dw(b);
    }
    {
        DwJob b = {};
        b.G = dO; b.ldg = OUT_LD; b.gcol0 = 0; b.N = OC_R + 4;                          // columns 0..19: pos | scales | rotations
        b.X = w.H1; b.ldx = NHEAD * WIDTH; b.x_nseg = 1; b.x_beg[0] = 0; b.x_len[0] = 3 * WIDTH; b.K = 3 * WIDTH;
        b.relu_x = 1; b.Kp = 3 * WIDTH; b.kblocks = (3 * WIDTH + 31) / 32;
        b.slabs = slabs + slab_per_job * NHEAD; b.dW = nullptr; b.db = nullptr;        // destinations: dj.sp
        dj.sp.job = ndw;
        dj.sp.K = WIDTH;
        for (int h = 0; h < 3; h++) {
            dj.sp.n0[h] = l.head_ocol[h]; dj.sp.n[h] = l.head_out[h]; dj.sp.k0[h] = h * WIDTH;
            dj.sp.dW[h] = g->head_w2[h]; dj.sp.db[h] = g->head_b2[h];
        }
        nkmax = (3 + 3 + 4) * (WIDTH + 1) > nkmax ? (3 + 3 + 4) * (WIDTH + 1) : nkmax;
        kbmax = b.kblocks > kbmax ? b.kblocks : kbmax;
        dj.j[ndw++] = b;
    }
    // PE tails -> inputs (accumulates onto the pass-through gradients written by outputs_backward)
    PeArgs pa = {P, l.ldz, l.oX, l.oS, l.oR, l.oT, l.oF, l.Kz, d->pos_pe, d->sr_pe, d->time_pe, d->force_pe,
                 d->use_time, d->use_force};
    if (do_data) {
        const size_t lds = sizeof(float) * PE_PTS * (size_t)(l.oT - l.oX);
        hipLaunchKernelGGL(pe_backward_kernel, dim3((unsigned)(((size_t)P + PE_PTS - 1) / PE_PTS)), dim3(256), lds, s, pa, points,
                           scales, rotations, dZ, dL_dpoints, dL_dscales_in, dL_drotations_in, hist,
                           d->L * HEX_SORTS * gs4d_hex_bins(d).nbins, fused_heads ? 1 : 0);
        GS4D_LAUNCH_CHECK("pe_backward", 0, s);
    }
    delete mlp_scope;
    mlp_scope = nullptr;
    // planes, xyz through the planes, t'
    if (do_data) {
        GS4D_PROF("hexplane_backward", s);
        int ft_done = 0;
        const int hrc = gs4d_hex_launch_backward(d, g, P, points, w.tprime, dfeat, l.F, dxyz_hex, gsix, hist, rank, key, rec, time,
                                                 force, dL_dpoints, dtprime, &ft_done, s);
        if (hrc) return hrc;
        if (!ft_done) {                                      // several batches: the sum over the partial sums is a launch of its own
            const int nred = (P + 255) / 256;
            hipLaunchKernelGGL(ft_embed_final_kernel, dim3(1), dim3(256), 0, s, nred, dtprime, g->ft_w, g->ft_b);
        }
        GS4D_LAUNCH_CHECK("deform_tail_backward", 0, s);
    }
    // every weight-gradient product of the network (reads dO, dH1, dZ, dT and the forward activations)
    if (do_weights) {
        GS4D_PROF("weight_grads", s);
        dj.njobs = ndw; dj.nz = (kbmax + 5) / 6; dj.nchunks = nchunks;
        const int grid = (nchunks + 7) / 8 * 8 * ndw * dj.nz;
        hipLaunchKernelGGL(dw_partial_kernel, dim3(grid), dim3(256), 0, s, P, dj);
        hipLaunchKernelGGL(dw_reduce_kernel, dim3((nkmax + 63) / 64, ndw), dim3(256), 0, s, nchunks, dj);
        GS4D_LAUNCH_CHECK("deform_dw", 0, s);
    }
    return GS4D_OK;
}

int gs4d_deform_backward(const gs4d_deform_desc* d, const gs4d_deform_grads* g, int P, const float* points,
                         const float* scales, const float* rotations, const float* time, const float* force,
                         const float* dL_dmeans3D, const float* dL_dscales, const float* dL_drotations,
                         const float* dL_dshs, float* dL_dpoints, float* dL_dscales_in, float* dL_drotations_in,
                         float* dL_dshs_in, char* workspace, char* scratch, void* stream)
{
    return gs4d_deform_backward_phases(d, g, P, points, scales, rotations, time, force, dL_dmeans3D, dL_dscales,
                                       dL_drotations, dL_dshs, dL_dpoints, dL_dscales_in, dL_drotations_in, dL_dshs_in,
                                       workspace, scratch, stream, GS4D_DEFORM_BWD_DATA | GS4D_DEFORM_BWD_WEIGHTS);
}

}  // extern "C"
