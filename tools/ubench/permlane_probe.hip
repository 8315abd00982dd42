// what v_permlane32_swap_b32 does to two registers on gfx950 (the layout kernels_match.hip relies on)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned a = 100u + threadIdx.x, b = 200u + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 512); unsigned h[128];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("a = 100 + lane, b = 200 + lane; after swap(a, b):\n r[0]: lane 0 -> %u, lane 31 -> %u, lane 32 -> %u, lane 63 -> %u\n r[1]: lane 0 -> %u, lane 31 -> %u, lane 32 -> %u, lane 63 -> %u\n",
           h[0], h[31], h[32], h[63], h[64], h[95], h[96], h[127]);
    return 0;
}
