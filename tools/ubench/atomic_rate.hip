// Micro-benchmark (experiments only, not part of the library): rate of device-scope 32-bit atomic adds from the whole chip
// as a function of how many distinct counters they go to and how the counters are laid out.  N adds per launch, lane i of
// the grid adds to counter hash(i) % A placed at word (counter * stride); returning (the lane then stores the result) or
// not.  Prints microseconds per launch.  Question behind it: what does counting 683 k (Gaussian, tile) instances into
// 2 500 per-tile counters cost (binning by direct placement)?
//   usage: atomic_rate <N adds> <A counters> <stride words> <returning 0/1> [runs of consecutive lanes on one counter]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k(uint32_t n, uint32_t a, uint32_t stride, int returning, uint32_t run,
                                         uint32_t* __restrict__ counters, uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t h = (i / run) * 2654435761u;
    h ^= h >> 15;
    const uint32_t c = h % a;
    if (returning)
        out[i] = atomicAdd(&counters[(size_t)c * stride], 1u);
    else
        atomicAdd(&counters[(size_t)c * stride], 1u);
}

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 683000u;
    const uint32_t a = argc > 2 ? (uint32_t)atol(argv[2]) : 2500u;
    const uint32_t stride = argc > 3 ? (uint32_t)atol(argv[3]) : 1u;
    const int returning = argc > 4 ? atoi(argv[4]) : 0;
    const uint32_t run = argc > 5 ? (uint32_t)atol(argv[5]) : 1u;
    uint32_t *counters, *out;
    if (hipMalloc(&counters, (size_t)a * stride * 4) != hipSuccess || hipMalloc(&out, (size_t)n * 4) != hipSuccess) return 1;
    (void)hipMemset(counters, 0, (size_t)a * stride * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int reps = 20;
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, n, a, stride, returning, run, counters, out);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, n, a, stride, returning, run, counters, out);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return 2;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("adds %u counters %u stride %u returning %d run %u : %.2f us per launch (%.1f adds/ns)\n", n, a, stride, returning, run,
           1e3f * ms / reps, n / (1e6f * ms / reps));
    return 0;
}
