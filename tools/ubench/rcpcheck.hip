// Accuracy of v_rcp_f64 and of its Newton refinements against the correctly rounded quotient (developer tool).
//   hipcc --offload-arch=gfx950 -O3 -o tools/scratch/rcpcheck tools/ubench/rcpcheck.hip && tools/scratch/rcpcheck
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ __forceinline__ double ulps(double got, double want)
{
    const double u = fabs(want) * 0x1p-52;      // >= one ulp of want
    return fabs(got - want) / u;
}

__global__ void chk(unsigned long long seed, int per_thread, double *worst, unsigned long long *inexact)
{
    unsigned long long s = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    double w0 = 0, w1 = 0, w2 = 0, w1b = 0; unsigned long long n1 = 0, n2 = 0;
    for (int i = 0; i < per_thread; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        // mantissa uniform, exponent in [-300, 300]
        const int e = (int)((s >> 52) % 601) - 300;
        const double m = 1.0 + (double)(s & 0xFFFFFFFFFFFFFull) * 0x1p-52;
        const double x = ldexp(m, e);
        const double want = 1.0 / x;             // IEEE division (OCML expansion: correctly rounded)
        const double r0 = __builtin_amdgcn_rcp(x);
        const double e0 = __builtin_fma(-x, r0, 1.0);
        const double r1 = __builtin_fma(r0, e0, r0);
        const double e1 = __builtin_fma(-x, r1, 1.0);
        const double r2 = __builtin_fma(r1, e1, r1);
        // one second-order step instead of two first-order ones: r0 (1 + e0 + e0^2)
        const double r1b = __builtin_fma(r0, __builtin_fma(e0, e0, e0), r0);
        w0 = fmax(w0, ulps(r0, want)); w1 = fmax(w1, ulps(r1, want)); w2 = fmax(w2, ulps(r2, want)); w1b = fmax(w1b, ulps(r1b, want));
        n1 += (r1 != want); n2 += (r2 != want);
    }
    atomicMax((unsigned long long *)&worst[0], (unsigned long long)__double_as_longlong(w0));
    atomicMax((unsigned long long *)&worst[1], (unsigned long long)__double_as_longlong(w1));
    atomicMax((unsigned long long *)&worst[2], (unsigned long long)__double_as_longlong(w2));
    atomicMax((unsigned long long *)&worst[3], (unsigned long long)__double_as_longlong(w1b));
    atomicAdd(&inexact[0], n1); atomicAdd(&inexact[1], n2);
}

int main()
{
    double *worst; unsigned long long *inexact;
    hipMalloc(&worst, 4 * sizeof(double)); hipMalloc(&inexact, 2 * sizeof(unsigned long long));
    hipMemset(worst, 0, 4 * sizeof(double)); hipMemset(inexact, 0, 2 * sizeof(unsigned long long));
    const int blocks = 1024, threads = 256, per = 4096;
    hipLaunchKernelGGL(chk, dim3(blocks), dim3(threads), 0, 0, 12345ull, per, worst, inexact);
    double w[4]; unsigned long long n[2];
    hipMemcpy(w, worst, sizeof w, hipMemcpyDeviceToHost); hipMemcpy(n, inexact, sizeof n, hipMemcpyDeviceToHost);
    const double total = (double)blocks * threads * per;
    printf("arguments %.3g\n", total);
    printf("v_rcp_f64 raw:            worst %.4g ulp (2^%.1f relative)\n", w[0], log2(w[0]) - 52);
    printf("one Newton step:          worst %.4g ulp, differs from the rounded quotient in %.4g %% of the arguments\n", w[1], 100.0 * n[0] / total);
    printf("two Newton steps:         worst %.4g ulp, differs in %.4g %%\n", w[2], 100.0 * n[1] / total);
    printf("one second-order step:    worst %.4g ulp\n", w[3]);
    return 0;
}
