// tools/ubench/rcp_f64_accuracy.hip -- how good is v_rcp_f64 on gfx950, bare and after one Newton step?  The scout pass of the
// AC-RANSAC kernel (kernels_filter.hip: scout_residual) replaces IEEE divisions by it and carries error intervals; this is the
// measurement behind their margins.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o /tmp/rcp_acc tools/ubench/rcp_f64_accuracy.hip && /tmp/rcp_acc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void rcp_kernel(const double* __restrict__ a, double* __restrict__ r0, double* __restrict__ r1, size_t n)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i];
    const double r = __builtin_amdgcn_rcp(x);
    r0[i] = r;
    r1[i] = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}

int main()
{
    const size_t n = 1u << 24;
    std::vector<double> a(n);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        // random sign, exponent over the whole normal range for a quarter of the values and [-60, 60] for the rest, random mantissa
        uint64_t mant = st & ((1ull << 52) - 1);
        int e = (i & 3) == 0 ? (int)((st >> 52) % 2000) - 1000 : (int)((st >> 52) % 121) - 60;
        uint64_t bits = ((uint64_t)(e + 1023) << 52) | mant | ((st >> 63) << 63);
        std::memcpy(&a[i], &bits, 8);
    }
    double *da, *d0, *d1;
    hipMalloc(&da, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(rcp_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, 0, da, d0, d1, n);
    std::vector<double> r0(n), r1(n);
    hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    long double w0 = 0, w1 = 0;
    for (size_t i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)a[i];
        if (!std::isfinite((double)t) || std::fabs((double)t) < 1e-300) continue;
        const long double e0 = fabsl(((long double)r0[i] - t) / t), e1 = fabsl(((long double)r1[i] - t) / t);
        if (e0 > w0) w0 = e0;
        if (e1 > w1) w1 = e1;
    }
    printf("v_rcp_f64 over %zu values: worst relative error %.3Le = 2^%.1Lf; after one Newton step %.3Le = 2^%.1Lf\n", n, w0, log2l(w0), w1, log2l(w1));
    return 0;
}
