// Micro-benchmark: how many cycles a SIMD needs per wave64 VALU instruction, by instruction -- the evidence behind
// bench.py's VALU_LANE_OPS_PEAK_T (the roof of the Hamming popcount kernel).
//   (1) one wave per SIMD, a straight run of N independent instructions between two s_memtime reads -> cycles per instruction
//       as the shader clock counts them (no occupancy, no memory);
//   (2) the shader clock itself: s_memtime ticks per second of wall_clock64 (100 MHz) over the same run;
//   (3) the whole chip: 2048 workgroups x 256 threads of the same runs -> T lane-op/s.
// hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// KIND 0: v_xor_b32   1: v_bcnt_u32_b32 (accumulating)   2: v_fma_f32   3: v_pk_fma_f32   4: v_xor + v_bcnt alternating
template <int KIND>
__global__ __launch_bounds__(256) void k_issue(uint32_t* out, unsigned long long* ticks, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = threadIdx.x * 3u, a2 = threadIdx.x * 5u, a3 = threadIdx.x * 7u;
    uint32_t a4 = threadIdx.x + 11u, a5 = threadIdx.x + 13u, a6 = threadIdx.x + 17u, a7 = threadIdx.x + 19u;
    float f0 = threadIdx.x, f1 = 1.f, f2 = 2.f, f3 = 3.f, f4 = 4.f, f5 = 5.f, f6 = 6.f, f7 = 7.f;
    const uint32_t k = 0x9E3779B9u;
    const float fa = 0.999f, fb = 1e-6f;
    typedef float f2_t __attribute__((ext_vector_type(2)));
    f2_t p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7}, p4 = {f1, f0}, p5 = {f3, f2}, p6 = {f5, f4}, p7 = {f7, f6};
    const f2_t pa = {fa, fa}, pb = {fb, fb};
    const unsigned long long t0 = __builtin_readcyclecounter();          // s_memtime
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            REP8(asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n"
                              "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        } else if (KIND == 1) {
            REP8(asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                              "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        } else if (KIND == 2) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa), "v"(fb));)
        } else if (KIND == 3) {
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));)
        } else {
            REP8(asm volatile("v_xor_b32 %0, %8, %1\n v_bcnt_u32_b32 %4, %0, %4\n v_xor_b32 %1, %8, %2\n v_bcnt_u32_b32 %5, %1, %5\n"
                              "v_xor_b32 %2, %8, %3\n v_bcnt_u32_b32 %6, %2, %6\n v_xor_b32 %3, %8, %0\n v_bcnt_u32_b32 %7, %3, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = w1 - w0; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7)
                                          ^ __float_as_uint(p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y);
}

template <int KIND>
static void run(const char* name, double lanes_per_instr, uint32_t* out, unsigned long long* ticks)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int per_iter = 64;                                 // instructions per loop trip
    // (1) + (2): one workgroup of 256 threads = one wave per SIMD of one CU
    {
        const int iters = 200000;
        hipLaunchKernelGGL(k_issue<KIND>, dim3(1), dim3(256), 0, 0, out, ticks, iters);
        (void)hipDeviceSynchronize();
        unsigned long long h[2]; (void)hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
        const double n = (double)iters * per_iter;
        const double wall_s = (double)h[1] / 100e6;          // wall_clock64: 100 MHz
        printf("%-28s one wave per SIMD: %.3f s_memtime ticks per instruction; s_memtime runs at %.1f MHz; %.3f ns per instruction\n", name,
               (double)h[0] / n, (double)h[0] / wall_s / 1e6, wall_s / n * 1e9);
    }
    // (3) the whole chip
    for (int rep = 0; rep < 2; ++rep) {
        const int iters = 20000, grid = 2048;
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(256), 0, 0, out, ticks, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)grid * 4 * iters * per_iter;              // wave64 instructions
        printf("%-28s whole chip: %7.2f T lane-op/s (%.1f ms) = %.3f G wave-instructions/s = %.2f per CU-SIMD and ns\n", name,
               instr * 64.0 * lanes_per_instr / (ms * 1e-3) / 1e12, ms, instr / (ms * 1e-3) / 1e9, instr / (ms * 1e-3) / 1e9 / 1024.0);
    }
}

int main()
{
    uint32_t* out; (void)hipMalloc(&out, 2048 * 256 * 4);
    unsigned long long* ticks; (void)hipMalloc(&ticks, 16);
    run<0>("v_xor_b32", 1.0, out, ticks);
    run<1>("v_bcnt_u32_b32", 1.0, out, ticks);
    run<4>("v_xor_b32 + v_bcnt_u32_b32", 1.0, out, ticks);
    run<2>("v_fma_f32", 1.0, out, ticks);
    run<3>("v_pk_fma_f32", 2.0, out, ticks);
    return 0;
}
