// Micro-benchmark: sustained issue rate of v_xor_b32 + v_bcnt_u32_b32 (the Hamming inner loop) and of v_fma_f32,
// to pin the integer-VALU roof the Hamming kernel is measured against.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_int(unsigned* out, int iters, unsigned seed)
{
    unsigned q[16], acc[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) q[i] = seed * (threadIdx.x + 1) + i * 2654435761u;
    unsigned a = seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = 0; w < 16; ++w) acc[u] += __builtin_popcount(q[w] ^ (a + u * 7 + w));
        a = a * 1664525u + 1013904223u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float seed)
{
    float acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = seed + i;
    float a = seed * 0.999f, b = 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    unsigned* out; (void)hipMalloc(&out, 2048 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 2048;
    for (int rep = 0; rep < 2; ++rep) {
        const int iters = 20000;
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_int, dim3(grid), dim3(256), 0, 0, out, iters, 12345u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double laneops = (double)grid * 256 * iters * 64.0 * 2.0;     // xor + bcnt per word
        printf("v_xor + v_bcnt_u32 (add fused): %7.2f T lane-op/s  (%.1f ms)\n", laneops / (ms * 1e-3) / 1e12, ms);
    }
    for (int rep = 0; rep < 2; ++rep) {
        const int iters = 20000;
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, (float*)out, iters, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double laneops = (double)grid * 256 * iters * 128.0;
        printf("v_fma_f32: %7.2f T lane-op/s = %.1f TFLOP/s  (%.1f ms)\n", laneops / (ms * 1e-3) / 1e12, 2 * laneops / (ms * 1e-3) / 1e12, ms);
    }
    return 0;
}
