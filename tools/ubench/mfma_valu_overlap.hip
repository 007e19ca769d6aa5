// Micro-benchmark (experiments only, not part of the library): do f32 VALU instructions of one wave overlap with
// v_mfma_f32_32x32x2_f32 of another wave on the same SIMD?  Workgroups of 512 threads = 2 waves per SIMD on every CU;
// mode 0: every wave runs MFMAs, mode 1: every wave runs v_fma_f32 chains, mode 2: even waves MFMA / odd waves VALU
// (each SIMD gets one of each), mode 3: like 2 with v_pk_fma_f32, mode 4: even MFMA / odd idle, mode 5: even idle / odd VALU.
// Prints the time per launch; per-wave work is the same in every mode.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k(int mode, int iters, float* out)
{
    const int wave = threadIdx.x >> 6;              // waves 0..7: SIMD = wave & 3, so waves w and w + 4 share a SIMD
    const bool first = wave < 4;
    bool do_mfma = false, do_valu = false, pk = false;
    switch (mode) {
        case 0: do_mfma = true; break;
        case 1: do_valu = true; break;
        case 2: do_mfma = first; do_valu = !first; break;
        case 3: do_mfma = first; do_valu = !first; pk = true; break;
        case 4: do_mfma = first; break;
        case 5: do_valu = !first; break;
    }
    float r = 0.f;
    if (do_mfma) {
        f32x16 c0 = {0}, c1 = {0};
        float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
            }
        }
        r = c0[0] + c1[5];
    } else if (do_valu) {
        // 16 MFMAs per iteration = 1024 cycles; the same time in v_fma_f32 at 4 cycles each = 256 instructions
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = threadIdx.x * 1e-3f + u;
        const float m = 1.0000001f, ad = 1e-7f;
        if (!pk) {
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int u = 0; u < 32; u++) {
#pragma unroll
                    for (int q = 0; q < 8; q++) x[q] = __builtin_fmaf(x[q], m, ad);
                }
            }
        } else {
            f2 y[4];
#pragma unroll
            for (int q = 0; q < 4; q++) y[q] = f2{x[2 * q], x[2 * q + 1]};
            const f2 m2 = {m, m}, a2 = {ad, ad};
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int u = 0; u < 32; u++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) y[q] = __builtin_elementwise_fma(y[q], m2, a2);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) { x[2 * q] = y[q].x; x[2 * q + 1] = y[q].y; }
        }
        r = x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7];
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}

int main()
{
    float* d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    for (int mode = 0; mode < 6; mode++) {
        k<<<256, 512>>>(mode, iters, d);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; rep++) k<<<256, 512>>>(mode, iters, d);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d: %.1f us per launch (%d iterations of 16 MFMAs or 256 v_fma / 128 v_pk_fma per wave)\n", mode, ms / 5 * 1e3, iters);
    }
    return 0;
}
