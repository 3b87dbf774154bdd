// wt_diag.hpp -- gfx950 device code of the reactor diagnostics (SURVEY.md section 8(f) NEXT-4):
// reductions over the zones of every reactor, fp64, one thread per reactor.
//
//   IntegratedCSTR.validate_conservation      core/reactor.py:570-611
//   TransportModel.calculate_mixing_quality   core/transport.py:338-384  (of pH and chlorine, reactor.py:638-639)
//   SpatialModel.identify_thermocline         core/spatial.py:352-379
//   SpatialModel.calculate_spatial_gradients  core/spatial.py:440-477    (of pH, chlorine, temperature)
//
// Sums follow numpy's pairwise order for a contiguous float64 vector of <= 128 elements (fewer than 8:
// left to right; otherwise eight interleaved accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
// then the tail one by one), so every field except the ones built on exp() is bit-identical to the
// reference.  The kernel reads 4 n doubles and writes 34 per reactor: a byte mover; the (N, n) state rows
// are read straight from the arrays the step kernel maintains (each thread walks its reactor's row, rows
// of neighbouring lanes are n * 8 bytes apart and stay in L2 between the passes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// numpy rounds every product and sum separately: no fused multiply-add in this file
#pragma clang fp contract(off)

namespace wtd {

constexpr int N_DIAG = 34;
enum { D_TOTAL_CL = 0, D_TOTAL_H, D_TOTAL_OH, D_CHARGE, D_THERMAL, D_PH_CV, D_PH_S, D_CL_CV, D_CL_S, D_THERMOCLINE, D_GRAD0 };
// D_GRAD0 + 8 * p + {mean, std, max, min, range, max_gradient, mean_gradient, gradient_location}, p = pH, chlorine, temperature

struct DiagArgs {
    int64_t N; int n;
    const double *par;       // [WT_NP][N]: rows 0 volume [L], 1 height [m], 10 stratification flag
    const double *pH, *Cl, *T, *H;   // [N][n]
    double *out;             // [N_DIAG][N]
};

// numpy's add.reduce over f(0..n-1)
template <class F>
__device__ __forceinline__ double np_sum(int n, F f)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += f(i);
        return res;
    }
    double r0 = f(0), r1 = f(1), r2 = f(2), r3 = f(3), r4 = f(4), r5 = f(5), r6 = f(6), r7 = f(7);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += f(i); r1 += f(i + 1); r2 += f(i + 2); r3 += f(i + 3); r4 += f(i + 4); r5 += f(i + 5); r6 += f(i + 6); r7 += f(i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += f(i);
    return res;
}

__device__ __forceinline__ void mean_std(const double *x, int n, double &mean, double &std_)
{   // np.mean, np.std (_methods._var: mean, subtract, square, mean, sqrt)
    mean = np_sum(n, [&](int i) { return x[i]; }) / (double)n;
    const double m = mean;
    std_ = sqrt(np_sum(n, [&](int i) { const double d = x[i] - m; return d * d; }) / (double)n);
}

__device__ __forceinline__ void mixing_quality(double mean, double std_, double &cv, double &seg)
{   // transport.py:366-384
    cv = (mean > 0.0) ? std_ / mean : 0.0;
    const double var = std_ * std_, vseg = mean * mean;
    seg = (vseg > 0.0) ? fmin(fmax(var / vseg, 0.0), 1.0) : 0.0;
}

__device__ __forceinline__ void gradient_stats(const double *x, int n, double zone_height, double mean, double std_,
                                               double *out, int64_t N)
{   // spatial.py:462-475
    double mx = x[0], mn = x[0];
    for (int i = 1; i < n; ++i) { mx = fmax(mx, x[i]); mn = fmin(mn, x[i]); }
    double gmax = -1.0; int gloc = 0;
    for (int i = 0; i + 1 < n; ++i) {                                  // np.argmax: first maximum
        const double g = fabs((x[i + 1] - x[i]) / zone_height);
        if (g > gmax) { gmax = g; gloc = i; }
    }
    const double gmean = np_sum(n - 1, [&](int i) { return fabs((x[i + 1] - x[i]) / zone_height); }) / (double)(n - 1);
    out[0 * N] = mean; out[1 * N] = std_; out[2 * N] = mx; out[3 * N] = mn; out[4 * N] = mx - mn;
    out[5 * N] = gmax; out[6 * N] = gmean; out[7 * N] = (double)gloc;
}

__global__ __launch_bounds__(64) void diagnostics_kernel(const DiagArgs a)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.N) return;
    const int64_t N = a.N; const int n = a.n;
    const double volume = a.par[0 * N + r], height = a.par[1 * N + r];
    const bool strat = a.par[10 * N + r] != 0.0;
    const double *pH = a.pH + r * n, *Cl = a.Cl + r * n, *T = a.T + r * n, *H = a.H + r * n;
    double *o = a.out + r;
    // ---- validate_conservation (reactor.py:579-600)
    const double zone_volume = volume / (double)n;
    o[D_TOTAL_CL * N] = np_sum(n, [&](int i) { return Cl[i]; }) * zone_volume;
    const double total_H = np_sum(n, [&](int i) { return H[i]; }) * zone_volume / 1000.0;
    const double T_K = T[0] + 273.15;                                  // Kw at the temperature of zone 0 (thermodynamics.py:219-226)
    const double Kw = 1.0e-14 * exp((55900.0 / 8.314) * (1.0 / 298.15 - 1.0 / T_K));
    const double total_OH = np_sum(n, [&](int i) { return Kw / H[i]; }) * zone_volume / 1000.0;
    o[D_TOTAL_H * N] = total_H; o[D_TOTAL_OH * N] = total_OH; o[D_CHARGE * N] = total_H - total_OH;
    double mT, sT; mean_std(T, n, mT, sT);
    const double mean_dT = np_sum(n, [&](int i) { return T[i] - 20.0; }) / (double)n;
    o[D_THERMAL * N] = 998.2 * 4184.0 * (volume / 1000.0) * mean_dT / 1000.0;
    // ---- mixing quality of pH and chlorine
    double mP, sP, mC, sC, cv, seg;
    mean_std(pH, n, mP, sP); mean_std(Cl, n, mC, sC);
    mixing_quality(mP, sP, cv, seg); o[D_PH_CV * N] = cv; o[D_PH_S * N] = seg;
    mixing_quality(mC, sC, cv, seg); o[D_CL_CV * N] = cv; o[D_CL_S * N] = seg;
    // ---- thermocline (spatial.py:359-379): depth below the top of the steepest temperature step, None -> NaN
    const double zh = height / (double)n;
    double gmax = -1.0; int gidx = 0;
    for (int i = 0; i + 1 < n; ++i) {
        const double g = fabs(T[i + 1] - T[i]) / zh;
        if (g > gmax) { gmax = g; gidx = i; }
    }
    o[D_THERMOCLINE * N] = (strat && gmax > 0.5) ? height - ((double)gidx + 0.5) * zh : __builtin_nan("");
    // ---- spatial gradient statistics
    gradient_stats(pH, n, zh, mP, sP, o + (int64_t)(D_GRAD0 + 0) * N, N);
    gradient_stats(Cl, n, zh, mC, sC, o + (int64_t)(D_GRAD0 + 8) * N, N);
    gradient_stats(T, n, zh, mT, sT, o + (int64_t)(D_GRAD0 + 16) * N, N);
}

} // namespace wtd

#pragma clang fp contract(fast)
