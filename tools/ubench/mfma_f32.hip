// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate for different wave/accumulator shapes.
// Usage: mfma_f32   (prints TFLOP/s per configuration)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, int iters, float seed)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = seed + i + r;
    float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-9f;
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int WPS>
void run(const char* name, int wg_per_cu)
{
    float* out; hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
    const int iters = 20000 / NACC;
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, WPS>), dim3(grid), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, WPS>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 /*waves*/ * iters * 16.0 * NACC * (2.0 * 32 * 32 * 2);
    printf("%-40s %8.2f TFLOP/s  (%.2f ms)\n", name, flops / (ms * 1e-3) / 1e12, ms);
    hipFree(out);
}

// sustained run with many distinct, data-like operands (DVFS check): 64 A and 64 B registers cycled
template <int WPS>
__global__ __launch_bounds__(256, WPS) void klong(float* out, const float* in, int iters)
{
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[32], b[64];
    for (int i = 0; i < 32; ++i) a[i] = in[(i * 256 + threadIdx.x) & 16383];
    for (int i = 0; i < 64; ++i) b[i] = in[(i * 256 + threadIdx.x + 8192) & 16383];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u + 32], acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) a[i] = a[i] * 0.999f + 0.001f * b[i];
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

void run_long(const char* name, bool random)
{
    float *out, *in; hipMalloc(&out, 256 * 256 * 16 * sizeof(float)); hipMalloc(&in, 16384 * 4);
    float h[16384]; unsigned x = 12345;
    for (int i = 0; i < 16384; ++i) { x = x * 1664525u + 1013904223u; h[i] = random ? (float)((x >> 8) & 255) : 0.f; }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 512;
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = 40000;
        hipEventRecord(e0);
        hipLaunchKernelGGL((klong<2>), dim3(grid), dim3(256), 0, 0, out, in, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * 64.0 * (2.0 * 32 * 32 * 2);
        printf("%-40s %8.2f TFLOP/s  (%.1f ms)\n", name, flops / (ms * 1e-3) / 1e12, ms);
    }
}

int main()
{
    run_long("sustained, zero operands, 2 w/SIMD", false);
    run_long("sustained, integer-valued random operands", true);
    run<1, 1>("1 wave/SIMD, 1 acc (dependent chain)", 1);
    run<2, 1>("1 wave/SIMD, 2 accs alternating", 1);
    run<4, 1>("1 wave/SIMD, 4 accs", 1);
    run<1, 2>("2 waves/SIMD, 1 acc each", 2);
    run<2, 2>("2 waves/SIMD, 2 accs each", 2);
    run<2, 3>("3 waves/SIMD, 2 accs each", 3);
    run<4, 2>("2 waves/SIMD, 4 accs each", 2);
    return 0;
}
