// The two per-iteration pieces that follow the render path:
//
//  * AdamW over every parameter tensor in ONE launch.  Behaviour matched: torch.optim.AdamW as the
//    reference constructs it (scene/gaussian_model.py:206: AdamW(l, lr=1e-4, eps=1e-8) -> betas (0.9, 0.999),
//    decoupled weight decay 0.01, no amsgrad), eight parameter groups with their own learning rates
//    (gaussian_model.py:195-204) updated per iteration (:228-242).  Per element, in torch's order:
//        p *= 1 - lr * wd;  m = lerp(m, g, 1 - b1);  v = b2 * v + (1 - b2) * g * g;
//        p += (-(lr / bc1) * m) / (sqrt(v) / sqrt(bc2) + eps)
//    with the scalars (1 - lr*wd, 1 - b1, 1 - b2, lr / bc1, sqrt(bc2)) formed in double and rounded to float once,
//    as torch's Python-scalar arithmetic does; compiled with -ffp-contract=off.
//    The reference runs torch's foreach implementation: ~10 elementwise launches per parameter group;
//    here a step is one streaming pass (read p, g, m, v; write p, m, v = 28 B per element: HBM bound).
//
//  * The HexPlane grid regularisers (scene/regulation.py:22-29 compute_plane_smoothness,
//    scene/gaussian_model.py:632-706 _plane/_time/_l1/_l2_regulation and compute_regulation): for a plane
//    t[1, C, H, W], smoothness = mean over all elements of (t[h+2] - 2 t[h+1] + t[h])^2 (second difference
//    along H, the pair's second axis), l1 = mean |1 - t|, l2 = mean (1 - t)^2; the total is the weighted
//    sum over the listed planes.  Planes are channels-last here ([H][W][C] in memory), so a step along H
//    is a step of W*C floats and every access is a coalesced stream.  Forward: per-workgroup partial sums
//    in fixed order, then one workgroup adds them (deterministic).  Backward: the analytic gradient is
//    ADDED to the caller's plane-gradient buffers (they already hold the render path's gradients).
#include "gs4d_internal.h"

namespace {

#define ADAMW_MAX_TENSORS 48
struct AdamwBatch {
    float* p[ADAMW_MAX_TENSORS];
    const float* g[ADAMW_MAX_TENSORS];
    float* m[ADAMW_MAX_TENSORS];
    float* v[ADAMW_MAX_TENSORS];
    unsigned long long n[ADAMW_MAX_TENSORS];
    float decay[ADAMW_MAX_TENSORS];       // 1 - lr * weight_decay
    float neg_step_size[ADAMW_MAX_TENSORS];   // -lr / bias_correction1
    float bc2_sqrt[ADAMW_MAX_TENSORS];
};

// grid.y = tensor, grid.x strides over its elements; float4 body + scalar tail (pointers are 16-byte aligned)
__global__ __launch_bounds__(256) void adamw_kernel(AdamwBatch B, float w1, float beta2, float w2, float eps)
{
    const int t = blockIdx.y;
    const size_t n = (size_t)B.n[t];
    float* __restrict__ p = B.p[t];
    const float* __restrict__ g = B.g[t];
    float* __restrict__ m = B.m[t];
    float* __restrict__ v = B.v[t];
    const float decay = B.decay[t], nstep = B.neg_step_size[t], bc2s = B.bc2_sqrt[t];
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        pp *= decay;
        mm = mm + w1 * (gg - mm);                       // lerp(m, g, 1 - beta1)
        vv = vv * beta2;
        vv = vv + (w2 * gg) * gg;                       // addcmul_(g, g, value = 1 - beta2)
        const float denom = sqrtf(vv) / bc2s + eps;
        pp = pp + (nstep * mm) / denom;                 // addcdiv_(m, denom, value = -step_size)
    };
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0) {
        const size_t i = 4 * n4 + threadIdx.x;
        if (i < n) {
            float pp = p[i], mm = m[i], vv = v[i];
            upd(pp, g[i], mm, vv);
            p[i] = pp; m[i] = mm; v[i] = vv;
        }
    }
}

// ---------------------------------------------------------------- plane regularisers
#define REG_MAX_TERMS 32
#define REG_BLOCKS 256          // partial sums per term
struct RegTerms {
    const float* plane[REG_MAX_TERMS];
    float* grad[REG_MAX_TERMS];
    int C[REG_MAX_TERMS], W[REG_MAX_TERMS], H[REG_MAX_TERMS];
    int kind[REG_MAX_TERMS];             // GS4D_REG_*
    float weight[REG_MAX_TERMS];
};

__device__ __forceinline__ float block_sum(float v, float* s)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s[0] + s[1]) + (s[2] + s[3]);
}

// grid = (REG_BLOCKS, nterms): partial[term][block] = sum over the block's elements of the term's integrand
__global__ __launch_bounds__(256) void plane_reg_partial_kernel(RegTerms T, float* __restrict__ partial)
{
    __shared__ float s[4];
    const int t = blockIdx.y;
    const float* __restrict__ x = T.plane[t];
    const size_t row = (size_t)T.W[t] * T.C[t];
    const int H = T.H[t], kind = T.kind[t];
    float acc = 0.f;
    if (kind == GS4D_REG_SMOOTHNESS) {
        const size_t n = H > 2 ? row * (size_t)(H - 2) : 0;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)REG_BLOCKS * 256) {
            const float d2 = (x[i + 2 * row] - x[i + row]) - (x[i + row] - x[i]);
            acc += d2 * d2;
        }
    } else {
        const size_t n = row * (size_t)H;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)REG_BLOCKS * 256) {
            const float d = 1.0f - x[i];
            acc += kind == GS4D_REG_L1 ? fabsf(d) : d * d;
        }
    }
    const float tot = block_sum(acc, s);
    if (threadIdx.x == 0) partial[(size_t)t * REG_BLOCKS + blockIdx.x] = tot;
}

// one workgroup: loss = sum_t weight_t * (sum of the term's partials) / count_t, fixed order
__global__ __launch_bounds__(256) void plane_reg_final_kernel(int nterms, RegTerms T, const float* __restrict__ partial,
                                                              float* __restrict__ loss)
{
    __shared__ float s[4];
    float total = 0.f;
    for (int t = 0; t < nterms; t++) {
        const float v = partial[(size_t)t * REG_BLOCKS + threadIdx.x];     // REG_BLOCKS == blockDim
        __syncthreads();
        const float sum = block_sum(v, s);
        const size_t row = (size_t)T.W[t] * T.C[t];
        const size_t cnt = T.kind[t] == GS4D_REG_SMOOTHNESS ? (T.H[t] > 2 ? row * (size_t)(T.H[t] - 2) : 0) : row * (size_t)T.H[t];
        if (cnt) total += T.weight[t] * (sum / (float)cnt);
    }
    if (threadIdx.x == 0) loss[0] = total;
}

// grad += g_loss * sum over the terms of one plane of weight_t * d(term_t)/d(plane); grid = (blocks, nterms).
// Several terms may share a plane (a time plane carries smoothness + l1 + l2): only the first term of a
// gradient buffer runs, and it adds every term of that buffer in one read-modify-write per element.
__global__ __launch_bounds__(256) void plane_reg_backward_kernel(int nterms, RegTerms T, const float* __restrict__ g_loss)
{
    const int t = blockIdx.y;
    float* __restrict__ g = T.grad[t];
    for (int u = 0; u < t; u++)
        if (T.grad[u] == g) return;                     // an earlier term owns this buffer (uniform)
    const float* __restrict__ x = T.plane[t];
    const size_t row = (size_t)T.W[t] * T.C[t];
    const int H = T.H[t];
    const size_t n = row * (size_t)H;
    const float gl = g_loss[0];
    float k_s = 0.f, k_1 = 0.f, k_2 = 0.f;              // summed coefficients of the three kinds on this plane
    for (int u = t; u < nterms; u++) {
        if (T.grad[u] != g) continue;
        if (T.kind[u] == GS4D_REG_SMOOTHNESS) { if (H > 2) k_s += gl * T.weight[u] * 2.0f / (float)(row * (size_t)(H - 2)); }
        else if (T.kind[u] == GS4D_REG_L1) k_1 += gl * T.weight[u] / (float)n;
        else k_2 += gl * T.weight[u] / (float)n;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float acc = 0.f;
        if (k_s != 0.f) {
            // s[h] = x[h+2] - 2 x[h+1] + x[h] for h in [0, H-2); d/dx[h] = 2/N (s[h-2] - 2 s[h-1] + s[h])
            const int h = (int)(i / row);
            auto sd = [&](int hh) -> float {          // second difference at row hh (0 outside its range)
                if (hh < 0 || hh > H - 3) return 0.f;
                const size_t j = i - (size_t)(h - hh) * row;
                return (x[j + 2 * row] - x[j + row]) - (x[j + row] - x[j]);
            };
            acc += k_s * ((sd(h - 2) - sd(h - 1)) - (sd(h - 1) - sd(h)));
        }
        if (k_1 != 0.f || k_2 != 0.f) {
            const float d = 1.0f - x[i];
            acc += (d > 0.f ? -k_1 : (d < 0.f ? k_1 : 0.f)) - 2.0f * k_2 * d;
        }
        g[i] += acc;
    }
}

static int fill_terms(int nterms, const gs4d_plane_reg_term* terms, RegTerms& T, bool need_grad)
{
    if (nterms < 0 || nterms > REG_MAX_TERMS || (nterms > 0 && !terms)) {
        gs4d_set_error("plane regulariser: between 0 and %d terms are supported, got %d", REG_MAX_TERMS, nterms);
        return GS4D_EINVAL;
    }
    for (int i = 0; i < nterms; i++) {
        const gs4d_plane_reg_term& t = terms[i];
        if (!t.plane || (need_grad && !t.grad) || t.C < 1 || t.W < 1 || t.H < 1 || t.kind < GS4D_REG_SMOOTHNESS || t.kind > GS4D_REG_L2) {
            gs4d_set_error("plane regulariser: term %d has a null pointer, an empty shape or an unknown kind", i);
            return GS4D_EINVAL;
        }
        T.plane[i] = t.plane; T.grad[i] = t.grad; T.C[i] = t.C; T.W[i] = t.W; T.H[i] = t.H; T.kind[i] = t.kind; T.weight[i] = t.weight;
    }
    return GS4D_OK;
}


// ---------------------------------------------------------------- row gather (densify / prune surgery)
#define GATHER_MAX_ARRAYS 32
struct GatherBatch {
    const uint32_t* src[GATHER_MAX_ARRAYS];
    uint32_t* dst[GATHER_MAX_ARRAYS];
    const uint32_t* tail[GATHER_MAX_ARRAYS];
    int row_words[GATHER_MAX_ARRAYS];
    int zero_fresh[GATHER_MAX_ARRAYS];
};

// grid.y = array; grid.x strides over its n_out * row_words 4-byte words: dst[i][w] = fresh[i] && zero_fresh ? 0 :
// i >= n_out - n_tail && tail ? tail[i - (n_out - n_tail)][w] : src[index[i]][w].  Words are moved as bit patterns.
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherBatch B, long long n_out, int n_tail,
                                                          const long long* __restrict__ index,
                                                          const unsigned char* __restrict__ fresh)
{
    const int a = blockIdx.y;
    const int rw = B.row_words[a];
    const uint32_t* __restrict__ src = B.src[a];
    uint32_t* __restrict__ dst = B.dst[a];
    const uint32_t* __restrict__ tail = B.tail[a];
    const bool zf = B.zero_fresh[a] != 0;
    const long long total = n_out * rw, tail0 = n_out - n_tail;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e / rw;
        const int w = (int)(e - i * rw);
        uint32_t v;
        if (zf && fresh && fresh[i]) v = 0u;
        else if (tail && i >= tail0) v = tail[(i - tail0) * rw + w];
        else v = src[index[i] * rw + w];
        dst[e] = v;
    }
}

}  // namespace

extern "C" {

int gs4d_adamw_step(int ntensors, const gs4d_adamw_tensor* tensors, double beta1, double beta2, double eps, void* stream)
{
    GS4D_TRACE_SCOPE("gs4d_adamw_step");
    hipStream_t s = (hipStream_t)stream;
    if (ntensors < 0 || (ntensors > 0 && !tensors)) { gs4d_set_error("gs4d_adamw_step: bad tensor list"); return GS4D_EINVAL; }
    GS4D_PROF("adamw", s);
    for (int base = 0; base < ntensors; base += ADAMW_MAX_TENSORS) {
        AdamwBatch B = {};
        const int cnt = ntensors - base < ADAMW_MAX_TENSORS ? ntensors - base : ADAMW_MAX_TENSORS;
        size_t nmax = 0;
        for (int i = 0; i < cnt; i++) {
            const gs4d_adamw_tensor& t = tensors[base + i];
            if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq || (((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15)) {
                gs4d_set_error("gs4d_adamw_step: tensor %d has a null or not 16-byte aligned pointer", base + i);
                return GS4D_EINVAL;
            }
            if (!(t.bias_correction1 > 0.0) || !(t.bias_correction2 > 0.0)) {
                gs4d_set_error("gs4d_adamw_step: tensor %d: bias corrections must be positive (step >= 1)", base + i);
                return GS4D_EINVAL;
            }
            B.p[i] = t.param; B.g[i] = t.grad; B.m[i] = t.exp_avg; B.v[i] = t.exp_avg_sq; B.n[i] = t.numel;
            B.decay[i] = (float)(1.0 - t.lr * t.weight_decay);
            B.neg_step_size[i] = (float)(-(t.lr / t.bias_correction1));
            B.bc2_sqrt[i] = (float)sqrt(t.bias_correction2);
            nmax = t.numel > nmax ? t.numel : nmax;
        }
        if (nmax == 0) continue;
        size_t gx = (nmax / 4 + 255) / 256;
        gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
        hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)gx, cnt), dim3(256), 0, s, B, (float)(1.0 - beta1), (float)beta2,
                           (float)(1.0 - beta2), (float)eps);
        GS4D_LAUNCH_CHECK("adamw", 0, s);
    }
    return GS4D_OK;
}

int gs4d_gather_rows(long long n_out, const long long* index, const unsigned char* fresh, int n_tail, int narrays,
                     const gs4d_gather_array* arrays, long long n_src, void* stream)
{
    GS4D_TRACE_SCOPE("gs4d_gather_rows");
    hipStream_t s = (hipStream_t)stream;
    if (n_out < 0 || n_tail < 0 || n_tail > n_out || narrays < 0 || (narrays > 0 && !arrays) || (n_out > 0 && !index) || n_src < 0) {
        gs4d_set_error("gs4d_gather_rows: bad sizes or null pointer");
        return GS4D_EINVAL;
    }
    if (n_out == 0 || narrays == 0) return GS4D_OK;
    for (int base = 0; base < narrays; base += GATHER_MAX_ARRAYS) {
        GatherBatch B = {};
        const int cnt = narrays - base < GATHER_MAX_ARRAYS ? narrays - base : GATHER_MAX_ARRAYS;
        long long wmax = 0;
        for (int i = 0; i < cnt; i++) {
            const gs4d_gather_array& a = arrays[base + i];
            if (!a.dst || a.row_words <= 0 || (!a.src && n_src > 0)) {
                gs4d_set_error("gs4d_gather_rows: array %d has a null pointer or an empty row", base + i);
                return GS4D_EINVAL;
            }
            B.src[i] = (const uint32_t*)a.src; B.dst[i] = (uint32_t*)a.dst; B.tail[i] = (const uint32_t*)a.tail;
            B.row_words[i] = a.row_words; B.zero_fresh[i] = a.zero_fresh;
            wmax = n_out * a.row_words > wmax ? n_out * a.row_words : wmax;
        }
        long long gx = (wmax + 255) / 256;
        gx = gx < 1 ? 1 : (gx > 4096 ? 4096 : gx);
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)gx, cnt), dim3(256), 0, s, B, n_out, n_tail, index, fresh);
        GS4D_LAUNCH_CHECK("gather_rows", 0, s);
    }
    return GS4D_OK;
}

size_t gs4d_plane_reg_scratch_bytes(void) { return sizeof(float) * REG_MAX_TERMS * REG_BLOCKS; }

int gs4d_plane_reg_forward(int nterms, const gs4d_plane_reg_term* terms, float* scratch, float* loss, void* stream)
{
    GS4D_TRACE_SCOPE("gs4d_plane_reg_forward");
    hipStream_t s = (hipStream_t)stream;
    RegTerms T = {};
    int rc = fill_terms(nterms, terms, T, false);
    if (rc) return rc;
    if (!scratch || !loss) { gs4d_set_error("gs4d_plane_reg_forward: null scratch or loss pointer"); return GS4D_EINVAL; }
    GS4D_PROF("plane_reg_forward", s);
    if (nterms > 0) hipLaunchKernelGGL(plane_reg_partial_kernel, dim3(REG_BLOCKS, nterms), dim3(256), 0, s, T, scratch);
    hipLaunchKernelGGL(plane_reg_final_kernel, dim3(1), dim3(256), 0, s, nterms, T, scratch, loss);
    GS4D_LAUNCH_CHECK("plane_reg_forward", 0, s);
    return GS4D_OK;
}

int gs4d_plane_reg_backward(int nterms, const gs4d_plane_reg_term* terms, const float* g_loss, void* stream)
{
    GS4D_TRACE_SCOPE("gs4d_plane_reg_backward");
    hipStream_t s = (hipStream_t)stream;
    RegTerms T = {};
    int rc = fill_terms(nterms, terms, T, true);
    if (rc) return rc;
    if (!g_loss) { gs4d_set_error("gs4d_plane_reg_backward: null upstream gradient"); return GS4D_EINVAL; }
    if (nterms == 0) return GS4D_OK;
    GS4D_PROF("plane_reg_backward", s);
    hipLaunchKernelGGL(plane_reg_backward_kernel, dim3(512, nterms), dim3(256), 0, s, nterms, T, g_loss);
    GS4D_LAUNCH_CHECK("plane_reg_backward", 0, s);
    return GS4D_OK;
}

}  // extern "C"
