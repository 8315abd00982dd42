// Micro-benchmark: do VALU instructions run in the shadow of a v_mfma_f32_32x32x16_f16 (8 passes = 32 cycles of the matrix pipe)?
// A wave runs  { MFMA (one of NACC independent accumulators) ; NV independent VALU instructions }  in a straight loop; the grid puts
// W waves on every SIMD.  If the vector ALU works beside the matrix pipe, cycles per MFMA = max(32, 4 NV + issue); if it does not, 32 + 4 NV.
// This is the question behind the count-tile kernel's roof: its list code is ~180 VALU instructions per 18 MFMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define V1 "v_xor_b32 %[x0], %[k], %[x0]\n"
#define V2 V1 "v_xor_b32 %[x1], %[k], %[x1]\n"
#define V4 V2 "v_xor_b32 %[x2], %[k], %[x2]\n v_xor_b32 %[x3], %[k], %[x3]\n"
#define V8 V4 V4
#define V12 V8 V4
#define V16 V8 V8

template <int NV>
__device__ __forceinline__ void valu(uint32_t& x0, uint32_t& x1, uint32_t& x2, uint32_t& x3, uint32_t k)
{
    if (NV == 1) asm volatile(V1 : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3) : [k] "v"(k));
    if (NV == 2) asm volatile(V2 : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3) : [k] "v"(k));
    if (NV == 4) asm volatile(V4 : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3) : [k] "v"(k));
    if (NV == 8) asm volatile(V8 : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3) : [k] "v"(k));
    if (NV == 12) asm volatile(V12 : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3) : [k] "v"(k));
    if (NV == 16) asm volatile(V16 : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3) : [k] "v"(k));
}

template <int NV, int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, int iters, float seed)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = seed + i + r;
    f32x4 a, b;
    for (int r = 0; r < 4; ++r) { a[r] = seed + threadIdx.x + r; b[r] = seed * 0.5f + threadIdx.x - r; }
    uint32_t x0 = threadIdx.x, x1 = threadIdx.x * 3u, x2 = threadIdx.x * 5u, x3 = threadIdx.x * 7u;
    const uint32_t kk = 0x9E3779B9u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                valu<NV>(x0, x1, x2, x3, kk);
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = (float)(x0 ^ x1 ^ x2 ^ x3);
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int NV, int NACC>
void run(int wg_per_cu)
{
    float* out; hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
    unsigned long long* ticks; hipMalloc(&ticks, 8);
    const int iters = 4000 / NACC, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, NACC>), dim3(grid), dim3(256), 0, 0, out, ticks, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NACC>), dim3(grid), dim3(256), 0, 0, out, ticks, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)wg_per_cu * iters * 8.0 * NACC;          // a workgroup = one wave per SIMD
    // cycles of the shader clock per MFMA of a SIMD, from the wall time at 2.4 GHz
    printf("NV %2d  accumulators %d  waves/SIMD %d :  %6.1f cycles per MFMA and SIMD at 2.4 GHz  (%.3f ms; 32 = the pipe, + 4 NV = %d if nothing overlaps)\n",
           NV, NACC, wg_per_cu, ms * 1e-3 * 2.4e9 / mfma_per_simd, ms, 32 + 4 * NV);
    hipFree(out); hipFree(ticks);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<1, 2>(w); run<2, 2>(w); run<4, 2>(w); run<8, 2>(w); run<12, 2>(w); run<16, 2>(w);
    }
    run<8, 1>(1); run<8, 1>(2); run<8, 4>(1); run<8, 4>(2);
    return 0;
}
