// Fused per-Gaussian activations and the L1 photometric loss for gfx950.
//
// Behaviour matched:
//  * scene/gaussian_model.py:38-43 + gaussian_renderer/renderer.py:170-172 of the reference: the
//    three activations applied between the deformation field and the rasterizer,
//    scales = exp(s), rotations = q / max(||q||_2, 1e-12) (torch.nn.functional.normalize),
//    opacity = sigmoid(o);
//  * utils/loss_utils.py:20-21: l1_loss = mean(|a - b|).
//
// Why fused: as separate framework ops these are ~25 launches of 2-5 us kernels per training
// step, each followed by a dispatch gap of similar length; here they are one kernel per direction
// (HBM-bound, 11 floats in / 11 floats out per Gaussian) and two kernels for the loss.
//
// The loss reduction is deterministic: fixed grid, fixed-order tree per workgroup, fixed-order
// final pass (no float atomics).
#include "gs4d_internal.h"
#include "gs4d_math.h"

using namespace gs4d;

#define L1_BLOCKS GS4D_L1_BLOCKS

namespace {

__global__ __launch_bounds__(256) void activate_forward_kernel(int P, const float* __restrict__ log_scales,
                                                               const float4* __restrict__ rot,
                                                               const float* __restrict__ opacity_logit,
                                                               float* __restrict__ scales, float4* __restrict__ rot_n,
                                                               float* __restrict__ opacity, uint32_t* __restrict__ clear,
                                                               unsigned clear_words)
{
    // the composite render call (csrc/render.hip) lets this launch, the one in front of the preprocess launch, clear the
    // per-tile counters of the binning by direct placement: a memset node of its own costs 5 us
    for (unsigned w = blockIdx.x * 256u + threadIdx.x; w < clear_words; w += gridDim.x * 256u) clear[w] = 0u;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
#pragma unroll
    for (int a = 0; a < 3; a++) scales[3 * (size_t)i + a] = expf(log_scales[3 * (size_t)i + a]);
    const float4 q = rot[i];
    const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    rot_n[i] = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
    opacity[i] = 1.0f / (1.0f + expf(-opacity_logit[i]));
}


// inputs: the forward OUTPUTS (scales, opacity) and the raw rotation; any upstream gradient may be NULL.
// l1_loss != null (a training step of the composite call, GS4D_LOSS_AFTER_BACKWARD): workgroup 0 first finishes the L1
// loss of the forward -- l1_final_kernel's sum over the partial sums, thread for thread -- which then needs no launch.
__global__ __launch_bounds__(256) void activate_backward_kernel(int P, const float* __restrict__ scales,
                                                                const float4* __restrict__ rot,
                                                                const float* __restrict__ opacity,
                                                                const float* __restrict__ g_scales,
                                                                const float4* __restrict__ g_rot,
                                                                const float* __restrict__ g_opacity,
                                                                float* __restrict__ d_log_scales, float4* __restrict__ d_rot,
                                                                float* __restrict__ d_opacity_logit,
                                                                const float* __restrict__ l1_partial, size_t l1_n,
                                                                float* __restrict__ l1_loss)
{
    if (l1_loss && blockIdx.x == 0) {                        // uniform for the workgroup
        __shared__ float s_l1[256];
        l1_finish(l1_partial, l1_n, l1_loss, s_l1);
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    activation_backward_point(i, scales, rot, opacity, g_scales, g_rot, g_opacity, d_log_scales, d_rot, d_opacity_logit);
}


// g_scale / d_a (both or neither): the launch also writes d(g * mean |a - b|)/da = g sign(a - b) / n, element for element
// what l1_backward_kernel computes (a training step knows g before its forward: csrc/render.hip)
template <bool GRAD>
__global__ __launch_bounds__(256) void l1_partial_kernel(size_t n, const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ partial, const float* __restrict__ g_scale,
                                                         float* __restrict__ d_a)
{
    __shared__ float s[256];
    const float k = GRAD ? g_scale[0] / (float)n : 0.f;
    auto sgn = [k](float d) { return d > 0.f ? k : (d < 0.f ? -k : 0.f); };
    float4* __restrict__ d4 = reinterpret_cast<float4*>(d_a);
    // float4 rows, two per trip, four independent partial sums per lane (fixed order: deterministic); the callers' tensors
    // are 16-byte aligned (checked by the entry point), the last n % 4 elements go to one lane
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    const size_t n4 = n >> 2, stride = (size_t)L1_BLOCKS * 256;
    const float4* __restrict__ a4 = reinterpret_cast<const float4*>(a);
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(b);
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const float4 x0 = a4[i], y0 = b4[i], x1 = a4[i + stride], y1 = b4[i + stride];
        if (GRAD) {
            d4[i] = make_float4(sgn(x0.x - y0.x), sgn(x0.y - y0.y), sgn(x0.z - y0.z), sgn(x0.w - y0.w));
            d4[i + stride] = make_float4(sgn(x1.x - y1.x), sgn(x1.y - y1.y), sgn(x1.z - y1.z), sgn(x1.w - y1.w));
        }
        acc0 += fabsf(x0.x - y0.x) + fabsf(x1.x - y1.x);
        acc1 += fabsf(x0.y - y0.y) + fabsf(x1.y - y1.y);
        acc2 += fabsf(x0.z - y0.z) + fabsf(x1.z - y1.z);
        acc3 += fabsf(x0.w - y0.w) + fabsf(x1.w - y1.w);
    }
    if (i < n4) {
        const float4 x0 = a4[i], y0 = b4[i];
        if (GRAD) d4[i] = make_float4(sgn(x0.x - y0.x), sgn(x0.y - y0.y), sgn(x0.z - y0.z), sgn(x0.w - y0.w));
        acc0 += fabsf(x0.x - y0.x); acc1 += fabsf(x0.y - y0.y); acc2 += fabsf(x0.z - y0.z); acc3 += fabsf(x0.w - y0.w);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t t = n4 << 2; t < n; t++) {
            acc0 += fabsf(a[t] - b[t]);
            if (GRAD) d_a[t] = sgn(a[t] - b[t]);
        }
    const float t = block_sum_256((acc0 + acc1) + (acc2 + acc3), s);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void l1_final_kernel(size_t n, const float* __restrict__ partial, float* __restrict__ loss)
{
    __shared__ float s[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < L1_BLOCKS; i += 256) acc += partial[i];
    const float t = block_sum_256(acc, s);
    if (threadIdx.x == 0) loss[0] = t / (float)n;
}

// d|a-b|/da = sign(a-b) (0 at equality, as torch.abs), scaled by the upstream scalar gradient / n
__global__ __launch_bounds__(256) void l1_backward_kernel(size_t n, const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ g_loss, float* __restrict__ d_a)
{
    const float k = g_loss[0] / (float)n;
    auto sgn = [k](float d) { return d > 0.f ? k : (d < 0.f ? -k : 0.f); };
    const size_t n4 = n >> 2;
    const float4* __restrict__ a4 = reinterpret_cast<const float4*>(a);
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(b);
    float4* __restrict__ d4 = reinterpret_cast<float4*>(d_a);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = a4[i], y = b4[i];
        d4[i] = make_float4(sgn(x.x - y.x), sgn(x.y - y.y), sgn(x.z - y.z), sgn(x.w - y.w));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t t = n4 << 2; t < n; t++) d_a[t] = sgn(a[t] - b[t]);
}

}  // namespace

extern "C" {

int gs4d_activate_forward(int P, const float* log_scales, const float* rotations, const float* opacity_logit,
                          float* scales, float* rotations_n, float* opacity, void* stream)
{
    return gs4d_activate_forward_clear(P, log_scales, rotations, opacity_logit, scales, rotations_n, opacity, nullptr, 0, stream);
}

}  // extern "C"
int gs4d_activate_forward_clear(int P, const float* log_scales, const float* rotations, const float* opacity_logit,
                                float* scales, float* rotations_n, float* opacity, uint32_t* clear, size_t clear_words,
                                void* stream)
{
    GS4D_TRACE_SCOPE("gs4d_activate_forward");
    if (P == 0) return GS4D_OK;
    if (P < 0 || !log_scales || !rotations || !opacity_logit || !scales || !rotations_n || !opacity ||
        ((uintptr_t)rotations & 15) || ((uintptr_t)rotations_n & 15)) {
        gs4d_set_error("gs4d_activate_forward: null, negative or misaligned argument (rotations need 16-byte alignment)");
        return GS4D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    GS4D_PROF("activate_forward", s);
    if (clear_words > 0xffffffffull) { gs4d_set_error("gs4d_activate_forward: clear range too large"); return GS4D_EINVAL; }
    hipLaunchKernelGGL(activate_forward_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, log_scales,
                       (const float4*)rotations, opacity_logit, scales, (float4*)rotations_n, opacity, clear, (unsigned)clear_words);
    GS4D_LAUNCH_CHECK("activate_forward", 0, s);
    return GS4D_OK;
}
extern "C" {

int gs4d_activate_backward(int P, const float* scales, const float* rotations, const float* opacity,
                           const float* g_scales, const float* g_rotations_n, const float* g_opacity,
                           float* d_log_scales, float* d_rotations, float* d_opacity_logit, void* stream)
{
    return gs4d_activate_backward_loss(P, scales, rotations, opacity, g_scales, g_rotations_n, g_opacity, d_log_scales, d_rotations,
                                       d_opacity_logit, nullptr, 0, nullptr, stream);
}

}  // extern "C"
int gs4d_activate_backward_loss(int P, const float* scales, const float* rotations, const float* opacity,
                                const float* g_scales, const float* g_rotations_n, const float* g_opacity,
                                float* d_log_scales, float* d_rotations, float* d_opacity_logit, const float* l1_partial,
                                size_t l1_n, float* l1_loss, void* stream)
{
    GS4D_TRACE_SCOPE("gs4d_activate_backward");
    if (P == 0) return GS4D_OK;
    if (P < 0 || !scales || !rotations || !opacity || !d_log_scales || !d_rotations || !d_opacity_logit ||
        ((uintptr_t)rotations & 15) || ((uintptr_t)g_rotations_n & 15) || ((uintptr_t)d_rotations & 15)) {
        gs4d_set_error("gs4d_activate_backward: null, negative or misaligned argument (rotations need 16-byte alignment)");
        return GS4D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    GS4D_PROF("activate_backward", s);
    hipLaunchKernelGGL(activate_backward_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, scales,
                       (const float4*)rotations, opacity, g_scales, (const float4*)g_rotations_n, g_opacity,
                       d_log_scales, (float4*)d_rotations, d_opacity_logit, l1_partial, l1_n, l1_loss);
    GS4D_LAUNCH_CHECK("activate_backward", 0, s);
    return GS4D_OK;
}
extern "C" {

size_t gs4d_l1_loss_scratch_bytes(void) { return (size_t)L1_BLOCKS * sizeof(float); }

int gs4d_l1_loss_forward(size_t n, const float* a, const float* b, float* scratch, float* loss, void* stream)
{
    return gs4d_l1_loss_forward_grad(n, a, b, scratch, loss, nullptr, nullptr, stream, 0);
}

}  // extern "C"
int gs4d_l1_loss_forward_grad(size_t n, const float* a, const float* b, float* scratch, float* loss, const float* g_scale,
                              float* d_a, void* stream, int skip_final)
{
    GS4D_TRACE_SCOPE("gs4d_l1_loss_forward");
    if ((g_scale != nullptr) != (d_a != nullptr) || ((uintptr_t)d_a & 15)) {
        gs4d_set_error("gs4d_l1_loss_forward: the gradient output needs its scale (and 16-byte alignment)");
        return GS4D_EINVAL;
    }
    if (n == 0 || !a || !b || !scratch || !loss) {
        gs4d_set_error("gs4d_l1_loss_forward: empty input or null pointer");
        return GS4D_EINVAL;
    }
    if (((uintptr_t)a | (uintptr_t)b) & 15) {
        gs4d_set_error("gs4d_l1_loss_forward: a and b must be 16-byte aligned");
        return GS4D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    GS4D_PROF("l1_loss_forward", s);
    if (d_a)
        hipLaunchKernelGGL(l1_partial_kernel<true>, dim3(L1_BLOCKS), dim3(256), 0, s, n, a, b, scratch, g_scale, d_a);
    else
        hipLaunchKernelGGL(l1_partial_kernel<false>, dim3(L1_BLOCKS), dim3(256), 0, s, n, a, b, scratch, g_scale, d_a);
    if (!skip_final)                                          // else: activate_backward_kernel finishes the sum (csrc/render.hip)
        hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, s, n, scratch, loss);
    GS4D_LAUNCH_CHECK("l1_loss_forward", 0, s);
    return GS4D_OK;
}
extern "C" {

int gs4d_l1_loss_backward(size_t n, const float* a, const float* b, const float* g_loss, float* d_a, void* stream)
{
    GS4D_TRACE_SCOPE("gs4d_l1_loss_backward");
    if (n == 0 || !a || !b || !g_loss || !d_a) {
        gs4d_set_error("gs4d_l1_loss_backward: empty input or null pointer");
        return GS4D_EINVAL;
    }
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)d_a) & 15) {
        gs4d_set_error("gs4d_l1_loss_backward: a, b and d_a must be 16-byte aligned");
        return GS4D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    GS4D_PROF("l1_loss_backward", s);
    const int blocks = (int)((n + 1023) / 1024 < 2048 ? (n + 1023) / 1024 : 2048);
    hipLaunchKernelGGL(l1_backward_kernel, dim3(blocks), dim3(256), 0, s, n, a, b, g_loss, d_a);
    GS4D_LAUNCH_CHECK("l1_loss_backward", 0, s);
    return GS4D_OK;
}

}  // extern "C"
