// Fused deformation heads for gfx950 -- included by deform.hip (inside its anonymous namespace).
//
// Behaviour matched (reference, PyTorch): scene/deformation.py:157-197 -- for each of the four heads
// (pos, scales, rotations, shs)   O_h = Linear2_h(relu(Linear1_h(relu([hidden | pe_h | time_emb | force_emb]))))
// followed by the residual adds / the quaternion update (utils/graphics_utils.py:110-133), and their backward.
//
// Round 1 ran this as one GEMM launch per layer group (gemm_rows_kernel): every head re-read its slice of Z from
// memory (154 MB at C2), the heads' raw outputs O made a round trip through memory to a separate outputs kernel, and
// the backward was four launches (outputs_backward, dH1, dZ, pe_backward) with dH1 / dZ / dO re-read in between.
// Here a wave owns 32 POINTS and keeps them in the N dimension of v_mfma_f32_32x32x2_f32 for the whole chain:
//   D[feature][point] += W[feature][k] * X^T[k][point]        (A operand = weights from LDS, B operand = activations)
// The accumulator of one layer -- lane (point j, half h) holds features 8m + 4h + q of ITS point in register 4m + q --
// is exactly the B operand the next layer needs (lane (j, h) supplies k = 8m + 4h + q in step (m, q)), so activations
// never leave the registers: no LDS transposes, no barriers, no intermediate round trips.  All four heads run on one
// load of the point's Z row; the epilogue writes the final outputs (forward) / the input gradients (backward).
// Weights live in LDS for the lifetime of a persistent workgroup (8 waves = 2 per SIMD; ~125 KB forward, ~106 KB
// backward at the mutant shape), staged in the layout [k block m][half h][row i][4 k] that makes the A operand of four
// consecutive steps one conflict-free ds_read_b128 (the same layout gemm_rows_kernel uses for its B operand).
// The f32 MFMA is exact fp32 (an fmaf chain), and the k order equals gemm_rows_kernel's, so the results match the
// layer-by-layer path.
#pragma once

#define HF_THREADS 512
#ifndef HFF_THREADS
#define HFF_THREADS 768          // forward: 152 VGPRs -> three waves per SIMD
#endif
#define HF_WAVES (HF_THREADS / 64)
#define HF_MAX_SEG_BLOCKS 8
#define HF_MAX_TF_BLOCKS 4

struct HeadsArgs {
    int P, ldz, nunits;
    const float* Z;              // [P][ldz]: hidden | pe segments | time / force embedding (pads are zero)
    float* H1;                   // [P][256] pre-activation outputs of the heads' first Linear (forward: written if store)
    float* O;                    // [P][OUT_LD]: only the rotation head's raw output (columns OC_R..+3) is kept
    unsigned* hmask;             // [P][10]: ReLU sign bits, written by the forward when store: word 2*head + half = (H1 > 0) of
                                 // the half's 32 units (bit 4m + q <-> unit 8m + 4h + q), words 8 + half = (hidden > 0)
    int store;                   // forward: 1 = a backward will follow (write H1 and O's rotation columns)
    const float* W1[NHEAD]; const float* b1[NHEAD]; const float* W2[NHEAD]; const float* b2[NHEAD];
    int in_w[NHEAD];             // true input width of the head's first Linear (row pitch of W1)
    int seg_beg[NHEAD], seg_len[NHEAD], seg_blocks[NHEAD];
    int oT, e, tf_blocks;
    int w1_off[NHEAD];           // float offsets into the staged weights (forward: W1; backward: see heads_backward_kernel)
    int w2_off[NHEAD];
    int b1_off, b2_off;
    int no_dx, no_ds, no_dr, no_dshs, apply_rotation;
    // forward epilogue
    const float* xyz; const float* scales; const float* rots; const float* shs;
    float* o_xyz; float* o_scales; float* o_rots; float* o_shs;
    // backward
    const float* g_xyz; const float* g_scales; const float* g_rots; const float* g_shs;
    float* dO; float* dH1; float* dZ;        // [P][OUT_LD], [P][256], [P][ldz] (hidden columns only)
    float* dxyz; float* dscales; float* drots; float* dshs;
    int pos_pe, sr_pe;
    int seg_off[3];              // backward: float offsets of the staged W1^T segment parts (pos, scales, rotations)
};

__device__ __forceinline__ float4 hf_relu4(float4 v)
{
    return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}

// one K block (8 k values: this lane supplies 4) into two 32-row output tiles
__device__ __forceinline__ void hf_mm2(f32x16& c0, f32x16& c1, const float4 a0, const float4 a1, const float4 x)
{
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, x.x, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, x.x, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, x.y, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, x.y, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, x.z, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, x.z, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, x.w, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, x.w, c1, 0, 0, 0);
}

__device__ __forceinline__ void hf_mm1(f32x16& c0, const float4 a0, const float4 x)
{
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, x.x, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, x.y, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, x.z, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, x.w, c0, 0, 0, 0);
}

// registers 4m'..4m'+3 of a 32-feature accumulator tile (m' = 0..3) as one float4: features 8m' + 4h + (0..3)
__device__ __forceinline__ float4 hf_quad(const f32x16& c, int mq)
{
    return make_float4(c[4 * mq], c[4 * mq + 1], c[4 * mq + 2], c[4 * mq + 3]);
}

// LDS position (float index) of element (k, row) in a staged operand of `rows` rows
__device__ __forceinline__ int hf_slot(int k, int row, int rows)
{
    return ((((k >> 3) * 2 + ((k >> 2) & 1)) * rows + row) << 2) + (k & 3);
}

// padded k of a head's first-layer input -> true column of W1 (or -1 for a padding column)
__device__ __forceinline__ int hf_true_col(const HeadsArgs& a, int hd, int kp)
{
    if (kp < WIDTH) return kp;
    kp -= WIDTH;
    const int sp = a.seg_blocks[hd] * 8;
    if (kp < sp) return kp < a.seg_len[hd] ? WIDTH + kp : -1;
    kp -= sp;
    return kp < a.e ? WIDTH + a.seg_len[hd] + kp : -1;
}

// ------------------------------------------------------------------ forward
// LDS: W1 of every head as [Kp_h / 8][2][64][4] | W2 of pos / scales / rotations as [8][2][8][4] | W2 of shs as
// [8][2][48][4] | b1 [4][64] | b2 [64] (pos 0..2, scales 8..10, rotations 16..19, shs at OC_SH: the columns of O)
__global__ __launch_bounds__(HFF_THREADS) void heads_forward_kernel(HeadsArgs a)
{
    extern __shared__ float smem[];
    __shared__ int s_next_item;
    int* s_next = &s_next_item;
    const int tid = threadIdx.x;
    if (tid == 0) s_next_item = 0;
    // Weight staging.  EVERY global load of the prologue -- W1 and W2 of the four heads, the biases -- is issued before
    // the first LDS store, with clamped addresses instead of loads under a condition (one memory round trip instead of
    // one per head and per W2 loop trip); the padding columns of the W1 region are zeroed while the loads are in flight.
    {
        constexpr int NV = (WIDTH + HF_MAX_SEG_BLOCKS * 8 + HF_MAX_TF_BLOCKS * 8) / 8;      // W1 columns per thread
        const int j1 = min(tid >> 3, WIDTH - 1), c0 = tid & 7;  // row (H1 unit) and first column of this thread (threads >= 512 idle for W1)
        float w1v[NHEAD][NV];
#pragma unroll
        for (int hd = 0; hd < NHEAD; hd++) {
            const float* __restrict__ W = a.W1[hd] + (size_t)j1 * a.in_w[hd];
            const int last = a.in_w[hd] - 1;
#pragma unroll
            for (int it = 0; it < NV; it++) w1v[hd][it] = W[min(c0 + 8 * it, last)];
        }
        // W2: pos / scales / rotations as 8 x 64 (rows past the head's outputs are zero), shs as 48 x 64
        float w2s[3], w2h[(48 * WIDTH + HFF_THREADS - 1) / HFF_THREADS];
        const int e2 = min(tid, 8 * WIDTH - 1), j2 = e2 >> 6, k2 = e2 & 63;
#pragma unroll
        for (int hd = 0; hd < 3; hd++) w2s[hd] = a.W2[hd][(size_t)min(j2, (hd == 2 ? 4 : 3) - 1) * WIDTH + k2];
#pragma unroll
        for (int q = 0; q < (int)(sizeof(w2h) / sizeof(float)); q++) w2h[q] = a.W2[3][min(tid + q * HFF_THREADS, 48 * WIDTH - 1)];
        const float b1v = a.b1[min(tid >> 6, NHEAD - 1)][tid & 63];
        float b2v = 0.f;
        {
            const int e = tid;       // the columns of O: pos 0..2, scales 8..10, rotations 16..19; then the 48 shs outputs at 64..111
            const int hd = e < OC_S ? 0 : (e < OC_R ? 1 : 2), off = e - (hd == 0 ? OC_X : (hd == 1 ? OC_S : OC_R));
            const bool small = e < 64 && off >= 0 && off < (hd == 2 ? 4 : 3);
            const bool big = e >= 64 && e < 64 + 48;
            const float* src = big ? a.b2[3] + (e - 64) : a.b2[hd] + (small ? off : 0);
            b2v = *src;
            if (!small && !big) b2v = 0.f;
        }
        {
            const int total4 = (a.w1_off[NHEAD - 1] + (8 + a.seg_blocks[NHEAD - 1] + a.tf_blocks) * 8 * WIDTH) >> 2;
            float4* z4 = reinterpret_cast<float4*>(smem);
            for (int e = tid; e < total4; e += HFF_THREADS) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int hd = 0; hd < NHEAD; hd++) {
            float* dst = smem + a.w1_off[hd];
            const int ld = a.in_w[hd];
            const int tf0 = WIDTH + a.seg_len[hd], tfp = WIDTH + a.seg_blocks[hd] * 8;   // first tf column: true / padded
#pragma unroll
            for (int it = 0; it < NV; it++) {
                const int c = c0 + 8 * it;
                if (c < ld && tid < 8 * WIDTH) dst[hf_slot(c < tf0 ? c : c - tf0 + tfp, j1, WIDTH)] = w1v[hd][it];
            }
        }
        if (tid < 8 * WIDTH) {
#pragma unroll
            for (int hd = 0; hd < 3; hd++) smem[a.w2_off[hd] + hf_slot(k2, j2, 8)] = j2 < (hd == 2 ? 4 : 3) ? w2s[hd] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < (int)(sizeof(w2h) / sizeof(float)); q++) {
            const int e = tid + q * HFF_THREADS;
            if (e < 48 * WIDTH) smem[a.w2_off[3] + hf_slot(e & 63, e >> 6, 48)] = w2h[q];
        }
        if (tid < NHEAD * WIDTH) smem[a.b1_off + tid] = b1v;
        if (tid < 64 + 48) smem[a.b2_off + tid] = b2v;
    }
    __syncthreads();

    const int lane = tid & 63, i = lane & 31, h = lane >> 5;
    const float4* sb1 = reinterpret_cast<const float4*>(smem + a.b1_off);
    const float* sb2 = smem + a.b2_off;
    // Work items are (32-point unit, head) pairs handed out dynamically inside the workgroup: whole units quantise
    // badly (3125 units on 1024 SIMDs at C2: 3 or 4 each), a quarter of a unit does not.  The workgroup owns units
    // blockIdx.x, blockIdx.x + gridDim.x, ...; the hidden / embedding columns are re-read per item (L1 / L2 hits).
    const int my_units = a.nunits > (int)blockIdx.x ? (a.nunits - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nitems = my_units * NHEAD;
    for (;;) {
        int item = 0;
        if (lane == 0) item = atomicAdd(s_next, 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= nitems) break;
        const int u = (int)blockIdx.x + (int)gridDim.x * (item >> 2);
        const int hd = item & 3;
        const int prow = u * 32 + i;
        const bool live = prow < a.P;
        const size_t row = (size_t)min(prow, a.P - 1);
        const float* __restrict__ z = a.Z + row * a.ldz + 4 * h;
        float4 xh[8], xt[HF_MAX_TF_BLOCKS], xs[HF_MAX_SEG_BLOCKS];
        const int nseg = a.seg_blocks[hd];
        unsigned hid_bits = 0u;
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const float4 v = *reinterpret_cast<const float4*>(z + 8 * m);
            hid_bits |= (v.x > 0.f ? 1u : 0u) << (4 * m) | (v.y > 0.f ? 2u : 0u) << (4 * m) | (v.z > 0.f ? 4u : 0u) << (4 * m) | (v.w > 0.f ? 8u : 0u) << (4 * m);
            xh[m] = hf_relu4(v);
        }
#pragma unroll
        for (int m = 0; m < HF_MAX_SEG_BLOCKS; m++)
            if (m < nseg) xs[m] = hf_relu4(*reinterpret_cast<const float4*>(z + a.seg_beg[hd] + 8 * m));
#pragma unroll
        for (int m = 0; m < HF_MAX_TF_BLOCKS; m++)
            if (m < a.tf_blocks) xt[m] = hf_relu4(*reinterpret_cast<const float4*>(z + a.oT + 8 * m));
        if (hd == 0 && a.store && live) a.hmask[row * 10 + 8 + h] = hid_bits;

        f32x16 c0 = {0}, c1 = {0};
        const float4* w = reinterpret_cast<const float4*>(smem + a.w1_off[hd]) + h * 64 + i;
#pragma unroll
        for (int m = 0; m < 8; m++) hf_mm2(c0, c1, w[m * 128], w[m * 128 + 32], xh[m]);
        w += 8 * 128;
#pragma unroll
        for (int m = 0; m < HF_MAX_SEG_BLOCKS; m++)
            if (m < nseg) hf_mm2(c0, c1, w[m * 128], w[m * 128 + 32], xs[m]);
        w += nseg * 128;
#pragma unroll
        for (int m = 0; m < HF_MAX_TF_BLOCKS; m++)
            if (m < a.tf_blocks) hf_mm2(c0, c1, w[m * 128], w[m * 128 + 32], xt[m]);
        // bias, optional store of the pre-activation (the weight-gradient operand) and of its sign bits, ReLU
        float4 xr[8];
        float* __restrict__ h1row = a.H1 + row * (NHEAD * WIDTH) + hd * WIDTH + 4 * h;
        unsigned bits = 0u;
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const float4 b = sb1[hd * 16 + 2 * m + h];
            const float4 q = hf_quad(m < 4 ? c0 : c1, m & 3);
            const float4 v = make_float4(q.x + b.x, q.y + b.y, q.z + b.z, q.w + b.w);
            if (a.store && live) *reinterpret_cast<float4*>(h1row + 8 * m) = v;
            bits |= (v.x > 0.f ? 1u : 0u) << (4 * m) | (v.y > 0.f ? 2u : 0u) << (4 * m) | (v.z > 0.f ? 4u : 0u) << (4 * m) | (v.w > 0.f ? 8u : 0u) << (4 * m);
            xr[m] = hf_relu4(v);
        }
        if (a.store && live) a.hmask[row * 10 + 2 * hd + h] = bits;   // the backward's ReLU mask: 4 bytes instead of 128
        if (hd < 3) {
            // The 3- or 4-wide second Linear of the small heads on the vector ALUs: on the matrix instruction it was one
            // 32-row tile per K block with 3 or 4 live rows -- 32 MFMAs per item -- and the exact-f32 MFMA shares the f32
            // ALUs with vector instructions (tools/ubench/mfma_valu_overlap.hip), so nothing hides it; here it is 128
            // FMAs.  Every lane of a half reads the same weight quad (LDS broadcast), the two halves' partial sums
            // (k = 8m + 4h + q) meet through one cross-half exchange per output.  (Measured: 109 -> 107 us at C2 -- the
            // kernel is bound by memory latency at three waves per SIMD, not by ALU cycles.)
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            const float4* w2 = reinterpret_cast<const float4*>(smem + a.w2_off[hd]) + h * 8;
#pragma unroll
            for (int m = 0; m < 8; m++) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float4 wv = w2[m * 16 + c];
                    o[c] = fmaf(wv.w, xr[m].w, fmaf(wv.z, xr[m].z, fmaf(wv.y, xr[m].y, fmaf(wv.x, xr[m].x, o[c]))));
                }
            }
#pragma unroll
            for (int c = 0; c < 4; c++) o[c] += __shfl_xor(o[c], 32, 64);
            if (h == 0 && live) {       // features 0..3 of the point sit in registers 0..3 of the h = 0 lane
                if (hd == 0) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float x = a.xyz[3 * row + c];
                        a.o_xyz[3 * row + c] = a.no_dx ? x : x + (o[c] + sb2[OC_X + c]);
                    }
                } else if (hd == 1) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float x = a.scales[3 * row + c];
                        a.o_scales[3 * row + c] = a.no_ds ? x : x + (o[c] + sb2[OC_S + c]);
                    }
                } else {
                    const float4 q1 = reinterpret_cast<const float4*>(a.rots)[row];
                    const float4 q2 = make_float4(o[0] + sb2[OC_R], o[1] + sb2[OC_R + 1], o[2] + sb2[OC_R + 2], o[3] + sb2[OC_R + 3]);
                    if (a.store) *reinterpret_cast<float4*>(a.O + row * OUT_LD + OC_R) = q2;
                    float4 r;
                    if (a.no_dr) {
                        r = q1;
                    } else if (!a.apply_rotation) {
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
                    reinterpret_cast<float4*>(a.o_rots)[row] = r;
                }
            }
        } else {
            f32x16 o0 = {0}, o1 = {0};
            const float4* w2 = reinterpret_cast<const float4*>(smem + a.w2_off[3]) + h * 48;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const float4 wa = w2[m * 96 + i];
                float4 wb = w2[m * 96 + 32 + (i & 15)];
                if (i >= 16) wb = make_float4(0.f, 0.f, 0.f, 0.f);
                hf_mm2(o0, o1, wa, wb, xr[m]);
            }
            if (live) {
                const float4* __restrict__ sin4 = reinterpret_cast<const float4*>(a.shs + 48 * row + 4 * h);
                float4* __restrict__ sout4 = reinterpret_cast<float4*>(a.o_shs + 48 * row + 4 * h);
                const float4* b4 = reinterpret_cast<const float4*>(sb2 + 64) + h;
#pragma unroll
                for (int m = 0; m < 6; m++) {
                    float4 s = sin4[2 * m];
                    if (!a.no_dshs) {
                        const float4 q = hf_quad(m < 4 ? o0 : o1, m & 3);
                        const float4 b = b4[2 * m];
                        s.x += q.x + b.x; s.y += q.y + b.y; s.z += q.z + b.z; s.w += q.w + b.w;
                    }
                    sout4[2 * m] = s;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ backward (data phase)
// Per point: dO from the upstream gradients (residual adds / quaternion backward), dH1_h = (W2_h^T dO_h) * (H1_h > 0),
// dZ[hidden] = (sum_h W1_h[:, hidden]^T dH1_h) * (Z > 0), the PE segments' dZ (unmasked, for pe_backward_kernel) and the
// pass-through gradients.  Written for the kernels that follow: dO, dH1 (weight gradients), dZ's hidden columns (trunk
// backward + weight gradients).
// LDS: W2^T of each head as [out blocks][2][64][4] (A[i = H1 unit][k = output]); W1^T hidden part of each head as
// [8][2][64][4] (A[i = hidden column][k = H1 unit]); W1^T segment part of pos [8][2][64][4], scales / rotations
// [8][2][32][4] (A[i = segment column][k = H1 unit], zero rows past the segment).
__global__ __launch_bounds__(HF_THREADS) void heads_backward_kernel(HeadsArgs a)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x;

    // Weight staging: every global load of the prologue is issued before the first LDS store, with clamped addresses
    // instead of loads under a condition (see heads_forward_kernel): one memory round trip instead of ~a dozen.
    {
        float w2s[3], w2h[6], w1h[NHEAD][8], sp[8], ss[2][4];
        const int k8 = tid >> 6, u8 = tid & 63;                     // small heads' W2^T: k = output (0..7), unit
#pragma unroll
        for (int hd = 0; hd < 3; hd++) w2s[hd] = a.W2[hd][(size_t)min(k8, (hd == 2 ? 4 : 3) - 1) * WIDTH + u8];
#pragma unroll
        for (int q = 0; q < 6; q++) w2h[q] = a.W2[3][tid + q * HF_THREADS];                 // 48 x 64, k = e >> 6, unit = e & 63
#pragma unroll
        for (int hd = 0; hd < NHEAD; hd++) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int e = tid + q * HF_THREADS;                 // unit = e >> 6, hidden column = e & 63
                w1h[hd][q] = a.W1[hd][(size_t)(e >> 6) * a.in_w[hd] + (e & 63)];
            }
        }
        {
            const int lastc = max(a.seg_len[0] - 1, 0);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int e = tid + q * HF_THREADS;                 // pos segment, 64 rows: unit = e >> 6, column = e & 63
                sp[q] = a.W1[0][(size_t)(e >> 6) * a.in_w[0] + WIDTH + min(e & 63, lastc)];
            }
        }
#pragma unroll
        for (int hd = 1; hd < 3; hd++) {
            const int lastc = max(a.seg_len[hd] - 1, 0);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int e = tid + q * HF_THREADS;                 // scales / rotations segment, 32 rows: unit = e >> 5, column = e & 31
                ss[hd - 1][q] = a.W1[hd][(size_t)(e >> 5) * a.in_w[hd] + WIDTH + min(e & 31, lastc)];
            }
        }
#pragma unroll
        for (int hd = 0; hd < 3; hd++)                              // A[i = unit][k = output] = W2[k][i]
            smem[a.w2_off[hd] + hf_slot(k8, u8, WIDTH)] = k8 < (hd == 2 ? 4 : 3) ? w2s[hd] : 0.f;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const int e = tid + q * HF_THREADS;
            smem[a.w2_off[3] + hf_slot(e >> 6, e & 63, WIDTH)] = w2h[q];
        }
#pragma unroll
        for (int hd = 0; hd < NHEAD; hd++) {                        // A[i = hidden column][k = unit] = W1[unit][i]
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int e = tid + q * HF_THREADS;
                smem[a.w1_off[hd] + hf_slot(e >> 6, e & 63, WIDTH)] = w1h[hd][q];
            }
        }
        if (a.seg_len[0] > 0) {                                     // A[i = segment column][k = unit] = W1[unit][64 + i]
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int e = tid + q * HF_THREADS;
                smem[a.seg_off[0] + hf_slot(e >> 6, e & 63, 64)] = (e & 63) < a.seg_len[0] ? sp[q] : 0.f;
            }
        }
#pragma unroll
        for (int hd = 1; hd < 3; hd++) {
            if (a.seg_len[hd] > 0) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int e = tid + q * HF_THREADS;
                    smem[a.seg_off[hd] + hf_slot(e >> 5, e & 31, 32)] = (e & 31) < a.seg_len[hd] ? ss[hd - 1][q] : 0.f;
                }
            }
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int u = blockIdx.x + gridDim.x * wave; u < a.nunits; u += gridDim.x * HF_WAVES) {
        const int prow = u * 32 + i;
        const bool live = prow < a.P;
        const size_t row = (size_t)min(prow, a.P - 1);
        unsigned mbits[5];                                     // ReLU sign bits of the four heads' H1 and of the hidden columns
#pragma unroll
        for (int k = 0; k < 5; k++) mbits[k] = a.hmask[row * 10 + 2 * k + h];
        // ---- dO of the four heads (B operands: k = output index 8m + 4h + q)
        float4 dpos = zero4, dsc = zero4, drot = zero4;
        float gx[3] = {0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f};
        float dq1[4] = {0.f, 0.f, 0.f, 0.f};
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) { gx[c] = a.g_xyz[3 * row + c]; gs[c] = a.g_scales[3 * row + c]; }
            if (!a.no_dx) dpos = make_float4(gx[0], gx[1], gx[2], 0.f);
            if (!a.no_ds) dsc = make_float4(gs[0], gs[1], gs[2], 0.f);
            const float4 gq = reinterpret_cast<const float4*>(a.g_rots)[row];
            float dq2[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.no_dr) {
                dq1[0] = gq.x; dq1[1] = gq.y; dq1[2] = gq.z; dq1[3] = gq.w;
            } else if (!a.apply_rotation) {
                dq1[0] = dq2[0] = gq.x; dq1[1] = dq2[1] = gq.y; dq1[2] = dq2[2] = gq.z; dq1[3] = dq2[3] = gq.w;
            } else {
                const float4 q1 = reinterpret_cast<const float4*>(a.rots)[row];
                const float4 q2 = *reinterpret_cast<const float4*>(a.O + row * OUT_LD + OC_R);
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
                dq1[0] = dqv[0] * q2.x + dqv[1] * q2.y + dqv[2] * q2.z + dqv[3] * q2.w;
                dq1[1] = -dqv[0] * q2.y + dqv[1] * q2.x - dqv[2] * q2.w + dqv[3] * q2.z;
                dq1[2] = -dqv[0] * q2.z + dqv[1] * q2.w + dqv[2] * q2.x - dqv[3] * q2.y;
                dq1[3] = -dqv[0] * q2.w - dqv[1] * q2.z + dqv[2] * q2.y + dqv[3] * q2.x;
                dq2[0] = dqv[0] * q1.x + dqv[1] * q1.y + dqv[2] * q1.z + dqv[3] * q1.w;
                dq2[1] = -dqv[0] * q1.y + dqv[1] * q1.x + dqv[2] * q1.w - dqv[3] * q1.z;
                dq2[2] = -dqv[0] * q1.z - dqv[1] * q1.w + dqv[2] * q1.x + dqv[3] * q1.y;
                dq2[3] = -dqv[0] * q1.w + dqv[1] * q1.z - dqv[2] * q1.y + dqv[3] * q1.x;
            }
            drot = make_float4(dq2[0], dq2[1], dq2[2], dq2[3]);
            if (live) {                                        // dO rows for the weight-gradient kernel
                float* __restrict__ d = a.dO + row * OUT_LD;
                *reinterpret_cast<float4*>(d + OC_X) = dpos;
                *reinterpret_cast<float4*>(d + OC_S) = dsc;
                *reinterpret_cast<float4*>(d + OC_R) = drot;
            }
        }
        // ---- heads: dH1, then its contributions to dZ.  A head's PE segment is finished (propagated to the inputs)
        // right after the head, so only the hidden accumulators stay live across heads.
        f32x16 zh0 = {0}, zh1 = {0};                           // dZ hidden columns (sum over heads)
        // (a rolled loop: unrolled, the scheduler hoists the later heads' loads and the kernel spills)
#pragma unroll 1
        for (int hd = 0; hd < NHEAD; hd++) {
            f32x16 c0 = {0}, c1 = {0};
            const float4* w2 = reinterpret_cast<const float4*>(smem + a.w2_off[hd]) + h * 64 + i;
            if (hd < 3) {
                const float4 dsmall = hd == 0 ? dpos : (hd == 1 ? dsc : drot);
                hf_mm2(c0, c1, w2[0], w2[32], dsmall);
            } else {
                const float4* __restrict__ g4 = reinterpret_cast<const float4*>(a.g_shs + 48 * row + 4 * h);
                float4* __restrict__ p4 = reinterpret_cast<float4*>(a.dshs + 48 * row + 4 * h);
                float4* __restrict__ o4 = reinterpret_cast<float4*>(a.dO + row * OUT_LD + OC_SH + 4 * h);
                const bool copy_through = a.dshs != a.g_shs;   // the caller may pass the upstream gradient itself as dL_dshs_in
                float4 dsh[6];
#pragma unroll
                for (int m = 0; m < 6; m++) {
                    const float4 g = g4[2 * m];
                    dsh[m] = a.no_dshs ? zero4 : g;
                    if (live && copy_through) p4[2 * m] = g;   // pass-through gradient of the SH input
                    if (live && a.no_dshs) o4[2 * m] = zero4;  // dO's SH columns: the weight-gradient kernel reads g_shs itself otherwise
                }
#pragma unroll
                for (int m = 0; m < 6; m++) hf_mm2(c0, c1, w2[m * 128], w2[m * 128 + 32], dsh[m]);
            }
            float4 dh[8];
            float* __restrict__ dh1row = a.dH1 + row * (NHEAD * WIDTH) + hd * WIDTH + 4 * h;
            const unsigned hb = hd == 0 ? mbits[0] : (hd == 1 ? mbits[1] : (hd == 2 ? mbits[2] : mbits[3]));
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const unsigned b = hb >> (4 * m);
                const float4 q = hf_quad(m < 4 ? c0 : c1, m & 3);
                dh[m] = make_float4((b & 1u) ? q.x : 0.f, (b & 2u) ? q.y : 0.f, (b & 4u) ? q.z : 0.f, (b & 8u) ? q.w : 0.f);
                if (live) *reinterpret_cast<float4*>(dh1row + 8 * m) = dh[m];
            }
            const float4* w1 = reinterpret_cast<const float4*>(smem + a.w1_off[hd]) + h * 64 + i;
#pragma unroll
            for (int m = 0; m < 8; m++) hf_mm2(zh0, zh1, w1[m * 128], w1[m * 128 + 32], dh[m]);
            // the head's PE segment of dZ, UNMASKED (pe_backward_kernel recomputes the sines / cosines for the derivative
            // anyway and takes the ReLU mask from their signs); propagating it to the inputs in here -- 48 columns with
            // per-column index arithmetic and partner reads -- cost more registers than the kernel has
            if (hd == 0 && a.seg_len[0] > 0) {
                f32x16 zp0 = {0}, zp1 = {0};
                const float4* ws = reinterpret_cast<const float4*>(smem + a.seg_off[0]) + h * 64 + i;
#pragma unroll
                for (int m = 0; m < 8; m++) hf_mm2(zp0, zp1, ws[m * 128], ws[m * 128 + 32], dh[m]);
                float* __restrict__ dzs = a.dZ + row * a.ldz + a.seg_beg[0] + 4 * h;
#pragma unroll
                for (int m = 0; m < 8; m++)
                    if (live && m < a.seg_blocks[0]) *reinterpret_cast<float4*>(dzs + 8 * m) = hf_quad(m < 4 ? zp0 : zp1, m & 3);
            } else if ((hd == 1 || hd == 2) && a.seg_len[hd] > 0) {
                f32x16 zs = {0};
                const float4* ws = reinterpret_cast<const float4*>(smem + a.seg_off[hd]) + h * 32 + i;
#pragma unroll
                for (int m = 0; m < 8; m++) hf_mm1(zs, ws[m * 64], dh[m]);
                float* __restrict__ dzs = a.dZ + row * a.ldz + a.seg_beg[hd] + 4 * h;
#pragma unroll
                for (int m = 0; m < 4; m++)
                    if (live && m < a.seg_blocks[hd]) *reinterpret_cast<float4*>(dzs + 8 * m) = hf_quad(zs, m);
            }
        }
        // ---- dZ hidden columns: ReLU mask, stored for the trunk backward and the weight gradients
        {
            float* __restrict__ dz = a.dZ + row * a.ldz + 4 * h;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const unsigned b = mbits[4] >> (4 * m);
                const float4 q = hf_quad(m < 4 ? zh0 : zh1, m & 3);
                if (live)
                    *reinterpret_cast<float4*>(dz + 8 * m) =
                        make_float4((b & 1u) ? q.x : 0.f, (b & 2u) ? q.y : 0.f, (b & 4u) ? q.z : 0.f, (b & 8u) ? q.w : 0.f);
            }
        }
        // ---- pass-through input gradients (pe_backward_kernel adds the PE segments' part)
        if (h == 0 && live) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                a.dxyz[3 * row + c] = gx[c];
                a.dscales[3 * row + c] = gs[c];
            }
            reinterpret_cast<float4*>(a.drots)[row] = make_float4(dq1[0], dq1[1], dq1[2], dq1[3]);
        }
    }
}
