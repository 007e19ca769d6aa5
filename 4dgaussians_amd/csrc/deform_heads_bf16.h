// Split-operand ("bf16 x 3") forms of the fused deformation heads -- included by deform.hip after deform_heads.h.
// OPT-IN (GS4D_MFMA=bf16x3); the exact-f32 kernels of deform_heads.h stay the default and the parity path.
//
// Why it exists.  The reference runs these GEMMs under torch.set_float32_matmul_precision('high') (train.py:383): TF32 or
// bf16 x 3 on the hardware it was written for.  gfx950 has no TF32 matrix instruction, and its exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32) runs at the f32 VECTOR rate on the vector ALUs -- measured with tools/ubench/
// mfma_valu_overlap.hip: a wave of f32 MFMAs and a wave of v_fma_f32 on one SIMD take the SUM of their times, they do
// not overlap -- so the exact kernels are bound by MFMA cycles + VALU cycles.  Here every f32 operand is split into two
// bf16 values, x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits together), and a product becomes
//     a b  ~=  a_hi b_hi + a_hi b_lo + a_lo b_hi            (the dropped a_lo b_lo term is ~2^-16 of the product)
// on v_mfma_f32_32x32x16_bf16 with f32 accumulation: 3 instructions of 32 cycles per 16 k instead of 8 of 64 cycles,
// 5.3x fewer matrix cycles, and they DO overlap with the vector work of other waves.  Error per dot product
// ~1e-5 of sum |a b| (TF32: ~5e-4); tests/test_gpu_deformation.py::test_bf16x3_mode reports it against the goldens.
//
// Register / operand mapping.  The exact kernels give lane (point j, half h) the k values 8m + 4h + q of K block m as one
// float4.  Two consecutive K blocks (16 k) are one bf16 MFMA: the lane's 8 values fill the instruction's k slots
// 8h .. 8h + 7 -- slot s = 4 (m & 1) + q -- for the A operand (weights, from LDS) and the B operand (activations) alike;
// the sum over k does not care which slot carries which k.  The accumulator layout is that of the f32 instruction, so a
// layer's output is still the next layer's B operand without leaving the registers.  Weights are split once per
// workgroup while they are staged: LDS holds [K pair][hi | lo][half][row][8 x bf16], 4 bytes per weight as before.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

struct HbSplit { bf16x8 hi, lo; };

__device__ __forceinline__ HbSplit hb_split(const float4 a, const float4 b)
{
    const f32x8 v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    HbSplit s;
    s.hi = __builtin_convertvector(v, bf16x8);
    const f32x8 back = __builtin_convertvector(s.hi, f32x8);
    s.lo = __builtin_convertvector(v - back, bf16x8);
    return s;
}

// one K pair into one 32-row tile: the two cross terms first, the leading term last
__device__ __forceinline__ void hb_mm(f32x16& c, const bf16x8 ah, const bf16x8 al, const HbSplit& x)
{
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, x.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, x.lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, x.hi, c, 0, 0, 0);
}

// staged bf16 element (in 2-byte units) of weight (row r, padded k kp) in a region of R rows; part 0 = hi, 1 = lo
__device__ __forceinline__ int hb_slot(int kp, int r, int R, int part)
{
    const int m = kp >> 3, hh = (kp >> 2) & 1, q = kp & 3;
    return (((((m >> 1) * 2 + part) * 2 + hh) * R + r) << 3) + ((m & 1) << 2) + q;
}

__device__ __forceinline__ void hb_store(__bf16* dst, int kp, int r, int R, float w)
{
    const __bf16 hi = (__bf16)w;
    dst[hb_slot(kp, r, R, 0)] = hi;
    dst[hb_slot(kp, r, R, 1)] = (__bf16)(w - (float)hi);
}

__host__ __device__ __forceinline__ int hb_even(int blocks) { return (blocks + 1) & ~1; }

// ------------------------------------------------------------------ forward
// LDS (float offsets as heads_forward_layout_bf16 lays them out): W1 of every head as [pairs][2][2][64][8 bf16] | W2 of
// pos / scales / rotations as [4][2][2][8][8] | W2 of shs as [4][2][2][48][8] | b1 [4][64] | b2 [64 + 48] (f32)
__global__ __launch_bounds__(HFF_THREADS) void heads_forward_bf16x3_kernel(HeadsArgs a)
{
    extern __shared__ float smem[];
    __shared__ int s_next_item;
    int* s_next = &s_next_item;
    const int tid = threadIdx.x;
    if (tid == 0) s_next_item = 0;
    {
        // every global load of the prologue before the first LDS store (clamped addresses, no load under a condition)
        constexpr int NV = (WIDTH + HF_MAX_SEG_BLOCKS * 8 + HF_MAX_TF_BLOCKS * 8) / 8;
        const int j1 = min(tid >> 3, WIDTH - 1), c0 = tid & 7;
        float w1v[NHEAD][NV];
#pragma unroll
        for (int hd = 0; hd < NHEAD; hd++) {
            const float* __restrict__ W = a.W1[hd] + (size_t)j1 * a.in_w[hd];
            const int last = a.in_w[hd] - 1;
#pragma unroll
            for (int it = 0; it < NV; it++) w1v[hd][it] = W[min(c0 + 8 * it, last)];
        }
        float w2s[3], w2h[(48 * WIDTH + HFF_THREADS - 1) / HFF_THREADS];
        const int e2 = min(tid, 8 * WIDTH - 1), j2 = e2 >> 6, k2 = e2 & 63;
#pragma unroll
        for (int hd = 0; hd < 3; hd++) w2s[hd] = a.W2[hd][(size_t)min(j2, (hd == 2 ? 4 : 3) - 1) * WIDTH + k2];
#pragma unroll
        for (int q = 0; q < (int)(sizeof(w2h) / sizeof(float)); q++) w2h[q] = a.W2[3][min(tid + q * HFF_THREADS, 48 * WIDTH - 1)];
        const float b1v = a.b1[min(tid >> 6, NHEAD - 1)][tid & 63];
        float b2v = 0.f;
        {
            const int e = tid;
            const int hd = e < OC_S ? 0 : (e < OC_R ? 1 : 2), off = e - (hd == 0 ? OC_X : (hd == 1 ? OC_S : OC_R));
            const bool small = e < 64 && off >= 0 && off < (hd == 2 ? 4 : 3);
            const bool big = e >= 64 && e < 64 + 48;
            const float* src = big ? a.b2[3] + (e - 64) : a.b2[hd] + (small ? off : 0);
            b2v = *src;
            if (!small && !big) b2v = 0.f;
        }
        {   // padding columns (and the rows of the small heads' W2 past their outputs) are zero
            const int total4 = a.b1_off >> 2;
            float4* z4 = reinterpret_cast<float4*>(smem);
            for (int e = tid; e < total4; e += HFF_THREADS) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int hd = 0; hd < NHEAD; hd++) {
            __bf16* dst = reinterpret_cast<__bf16*>(smem + a.w1_off[hd]);
            const int ld = a.in_w[hd];
            const int tf0 = WIDTH + a.seg_len[hd], tfp = WIDTH + hb_even(a.seg_blocks[hd]) * 8;
#pragma unroll
            for (int it = 0; it < NV; it++) {
                const int c = c0 + 8 * it;
                if (c < ld && tid < 8 * WIDTH) hb_store(dst, c < tf0 ? c : c - tf0 + tfp, j1, WIDTH, w1v[hd][it]);
            }
        }
        if (tid < 8 * WIDTH) {
#pragma unroll
            for (int hd = 0; hd < 3; hd++)
                if (j2 < (hd == 2 ? 4 : 3)) hb_store(reinterpret_cast<__bf16*>(smem + a.w2_off[hd]), k2, j2, 8, w2s[hd]);
        }
#pragma unroll
        for (int q = 0; q < (int)(sizeof(w2h) / sizeof(float)); q++) {
            const int e = tid + q * HFF_THREADS;
            if (e < 48 * WIDTH) hb_store(reinterpret_cast<__bf16*>(smem + a.w2_off[3]), e & 63, e >> 6, 48, w2h[q]);
        }
        if (tid < NHEAD * WIDTH) smem[a.b1_off + tid] = b1v;
        if (tid < 64 + 48) smem[a.b2_off + tid] = b2v;
    }
    __syncthreads();

    const int lane = tid & 63, i = lane & 31, h = lane >> 5;
    const float4* sb1 = reinterpret_cast<const float4*>(smem + a.b1_off);
    const float* sb2 = smem + a.b2_off;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
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
            xs[m] = m < nseg ? hf_relu4(*reinterpret_cast<const float4*>(z + a.seg_beg[hd] + 8 * m)) : zero4;
#pragma unroll
        for (int m = 0; m < HF_MAX_TF_BLOCKS; m++)
            xt[m] = m < a.tf_blocks ? hf_relu4(*reinterpret_cast<const float4*>(z + a.oT + 8 * m)) : zero4;
        if (hd == 0 && a.store && live) a.hmask[row * 10 + 8 + h] = hid_bits;

        f32x16 c0 = {0}, c1 = {0};
        // 16-byte units: ((pair * 2 + part) * 2 + half) * 64 + row
        const bf16x8* w = reinterpret_cast<const bf16x8*>(smem + a.w1_off[hd]) + h * 64 + i;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const HbSplit x = hb_split(xh[2 * p], xh[2 * p + 1]);
            hb_mm(c0, w[(p * 4 + 0) * 64], w[(p * 4 + 2) * 64], x);
            hb_mm(c1, w[(p * 4 + 0) * 64 + 32], w[(p * 4 + 2) * 64 + 32], x);
        }
        w += 4 * 4 * 64;
        const int segp = hb_even(nseg) >> 1;
#pragma unroll
        for (int p = 0; p < HF_MAX_SEG_BLOCKS / 2; p++)
            if (p < segp) {
                const HbSplit x = hb_split(xs[2 * p], xs[2 * p + 1]);
                hb_mm(c0, w[(p * 4 + 0) * 64], w[(p * 4 + 2) * 64], x);
                hb_mm(c1, w[(p * 4 + 0) * 64 + 32], w[(p * 4 + 2) * 64 + 32], x);
            }
        w += segp * 4 * 64;
        const int tfp = hb_even(a.tf_blocks) >> 1;
#pragma unroll
        for (int p = 0; p < HF_MAX_TF_BLOCKS / 2; p++)
            if (p < tfp) {
                const HbSplit x = hb_split(xt[2 * p], xt[2 * p + 1]);
                hb_mm(c0, w[(p * 4 + 0) * 64], w[(p * 4 + 2) * 64], x);
                hb_mm(c1, w[(p * 4 + 0) * 64 + 32], w[(p * 4 + 2) * 64 + 32], x);
            }
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
        if (a.store && live) a.hmask[row * 10 + 2 * hd + h] = bits;
        if (hd < 3) {
            f32x16 o = {0};
            // [pair][part][half][8 rows]: lanes of rows >= 8 multiply zeros
            const bf16x8* w2 = reinterpret_cast<const bf16x8*>(smem + a.w2_off[hd]) + h * 8 + (i & 7);
            const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int p = 0; p < 4; p++) {
                bf16x8 wh = w2[(p * 4 + 0) * 8], wl = w2[(p * 4 + 2) * 8];
                if (i >= 8) { wh = z8; wl = z8; }
                hb_mm(o, wh, wl, hb_split(xr[2 * p], xr[2 * p + 1]));
            }
            if (h == 0 && live) {
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
            // [pair][part][half][48 rows]
            const bf16x8* w2 = reinterpret_cast<const bf16x8*>(smem + a.w2_off[3]) + h * 48;
            const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const HbSplit x = hb_split(xr[2 * p], xr[2 * p + 1]);
                hb_mm(o0, w2[(p * 4 + 0) * 48 + i], w2[(p * 4 + 2) * 48 + i], x);
                bf16x8 wh = w2[(p * 4 + 0) * 48 + 32 + (i & 15)], wl = w2[(p * 4 + 2) * 48 + 32 + (i & 15)];
                if (i >= 16) { wh = z8; wl = z8; }
                hb_mm(o1, wh, wl, x);
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
