// wt_device.hpp -- gfx950 device code of the multi-zone CSTR physics step.
//
// Mapping (CDNA4, 64-wide wavefronts): one LANE per reactor zone, the n zones
// of a reactor in n consecutive lanes, floor(64/n) reactors per wavefront, one
// wavefront per workgroup.  Everything a reactor needs for a whole outer step
// (state, Radau stage vectors, Jacobian bands, tridiagonal factors) lives in
// that segment's registers; the 1-D inter-zone stencil, the tridiagonal solves
// (parallel cyclic reduction) and the RMS norms are wavefront shuffles.  HBM is
// touched once per launch: state in, state + derived out.
//
// What is computed is the reference's IntegratedCSTR.step():
//   RHS            reactor.py:272-448 (+ thermodynamics.py:160-193,
//                  chemistry.py:400-437,483-523, spatial.py:142-320)
//   time stepping  scipy 1.15.3 Radau IIA(5): radau.py:48-176,399-539,
//                  common.py:63-134 (initial step), :257-382 (num_jac)
//   post-step      reactor.py:493-541
// The decision sequence (initial step, Newton iteration counts, accept/reject,
// Jacobian refresh, LU reuse) is scipy's; the linear algebra exploits the
// structure of this RHS instead of a dense LU: with unknowns ordered
// [T | pH | Cl] the Jacobian is block lower-triangular with tridiagonal
// diagonal blocks, so (mu/h I - J) x = b is three tridiagonal solves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wt {

constexpr int SPH = 0, SCL = 1, STT = 2;  // species index inside a lane
constexpr double RTOL = 1e-6, ATOL = 1e-8; // reactor.py:481-483
constexpr int NEWTON_MAXITER = 6;          // radau.py:43
constexpr double MIN_FACTOR = 0.2, MAX_FACTOR = 10.0;
constexpr double LN10 = 2.302585092994046; // np.log(10)
constexpr double DEPS = 2.220446049250313e-16;

// status bits (include/wtphys.h)
constexpr uint32_t ST_T_RANGE = 1, ST_SOLVER_FAILED = 2, ST_CLAMP_PH = 4, ST_CLAMP_CL = 8,
                   ST_CLAMP_T = 16, ST_T_RANGE_POST = 32, ST_NONFINITE = 64;

// Radau IIA constants, filled on the host the way radau.py:11-40 computes them.
struct RadauConsts {
    double C[3], E[3];
    double MU_REAL, MU_CR, MU_CI;
    double T[3][3], TI[3][3], P[3][3];
    double NJ_REJECT, NJ_SMALL, NJ_BIG, NJ_MINF, NJ_F0; // common.py:248-253
    double newton_tol;                                   // radau.py:315
};

struct StepArgs {
    int64_t N;
    int n;            // zones per reactor
    int R;            // reactors per wavefront = 64 / n
    const double *par; // [WT_NP][N]
    const double *bc;  // [WT_NB][N]
    double *pH, *Cl, *T; // [N][n]
    double *time, *flow; // [N]
    double *dH, *dRho, *dK; // derived [N][n]
    uint32_t *status;    // [N]
    int32_t *stats;      // [N][5] or nullptr
    double dt;
    int n_steps;
    RadauConsts rc;
};

// ---------------------------------------------------------------- lane geometry
struct Lane {
    int n, z;
    bool has_lo, has_hi;
    int base;                 // lane id of zone 0 of this segment
    unsigned long long segmask;
};

__device__ __forceinline__ double shfl_lo(double x, int d = 1) { return __shfl_up(x, d, 64); }
__device__ __forceinline__ double shfl_hi(double x, int d = 1) { return __shfl_down(x, d, 64); }

__device__ __forceinline__ bool seg_any(const Lane &L, bool p) { return (__ballot(p) & L.segmask) != 0ull; }
__device__ __forceinline__ bool seg_all(const Lane &L, bool p) { return (__ballot(!p) & L.segmask) == 0ull; }

// Sum over the n lanes of a segment; every lane of the segment receives the
// bitwise-identical value (inclusive scan, then broadcast from the last zone).
__device__ __forceinline__ double seg_sum(const Lane &L, double x)
{
    for (int s = 1; s < L.n; s <<= 1) {
        double v = __shfl_up(x, s, 64);
        if (L.z >= s) x += v;
    }
    return __shfl(x, L.base + L.n - 1, 64);
}

// ---------------------------------------------------------------- reactor constants
struct RK {
    // chemistry.py:116-132 constants (frozen at configuration temperature)
    double Kw, Ka1, Ka1Ka2, KaH, cbeta;
    // transport / spatial
    double Kex, dz, u2, ri_crit, supp;
    int strat_mode; // 0: stratification off, 1: Richardson test, 2: u<=1e-6 (Ri=+inf)
    // boundary-derived (reactor.py:336,349-368,385-395,426-443)
    double Qv, H_in, Cl_in, T_in, acid_dH, cl_dose, UA, T_amb, inv_rcv;
    bool has_acid, has_cl, has_heat;
};

__device__ __forceinline__ void load_reactor(const StepArgs &a, int64_t r, int n, RK &k)
{
    const int64_t N = a.N;
    auto P = [&](int row) { return a.par[(int64_t)row * N + r]; };
    auto B = [&](int row) { return a.bc[(int64_t)row * N + r]; };
    const double V = P(0), height = P(1), diam = P(2);
    k.Kw = P(3); k.Ka1 = P(4); k.Ka1Ka2 = P(4) * P(5); k.KaH = P(6);
    k.cbeta = 2.303 * P(7);                    // chemistry.py:431-433
    k.Kex = P(8);
    const double u = P(9);
    k.u2 = u * u;
    k.dz = height / n;                          // spatial.py:119
    k.ri_crit = P(11); k.supp = P(12);
    k.strat_mode = (P(10) != 0.0) ? ((u > 1e-6) ? 1 : 2) : 0; // reactor.py:310, spatial.py:270-275
    const double Q_in = B(0);
    k.Qv = (Q_in / 60.0) / V;                   // reactor.py:336
    k.H_in = exp10(-B(1));                      // reactor.py:363
    k.Cl_in = B(2); k.T_in = B(3);
    const double zone_volume_L = V / n;
    k.has_acid = B(4) > 0;
    k.acid_dH = ((B(4) / 60.0) * B(5)) / zone_volume_L; // reactor.py:350-354
    k.has_cl = B(6) > 0;
    k.cl_dose = ((B(6) / 60.0) * B(7)) / zone_volume_L; // reactor.py:388-392
    k.has_heat = B(9) > 0;
    const double PI = 3.141592653589793;
    const double A_lat = PI * diam * height;
    const double rr = diam / 2;
    const double A_tot = A_lat + 2 * PI * (rr * rr);  // reactor.py:429-431
    k.UA = B(9) * A_tot;
    k.T_amb = B(8);
    k.inv_rcv = 1.0 / (998.2 * 4184 * (V / 1000)); // reactor.py:433-435
}

// ---------------------------------------------------------------- zone-local properties
struct PropPH { double H, iw, phi; bool bpos; }; // iw = 1/(beta*ln10)
struct PropT { double kT, rho; bool bad; };

// H = 10^-pH, buffering capacity beta (chemistry.py:400-437), HOCl/OCl- decay
// factor (chemistry.py:483-523).
__device__ __forceinline__ PropPH prop_pH(const RK &k, double pH)
{
    PropPH p;
    const double H = exp10(-pH);
    const double beta_w = 2.303 * (H + k.Kw / H);
    const double H2 = H * H;
    const double D = H2 + k.Ka1 * H + k.Ka1Ka2;
    const double iD = 1.0 / D;
    const double a0 = H2 * iD, a1 = (k.Ka1 * H) * iD, a2 = k.Ka1Ka2 * iD;
    const double beta = beta_w + k.cbeta * (a0 * a1 + 4 * a1 * a2 + a0 * a2);
    p.bpos = beta > 0;                           // reactor.py:358,367,375 guards
    p.iw = 1.0 / (beta * LN10);
    const double iHK = 1.0 / (H + k.KaH);
    p.phi = H * iHK * 1.0 + k.KaH * iHK * 0.02;
    p.H = H;
    return p;
}

// Arrhenius decay rate (thermodynamics.py:160-193) with its [0,100] C check
// (:146-157) and water density (spatial.py:177-189).
__device__ __forceinline__ PropT prop_T(double T)
{
    PropT p;
    p.bad = (T < 0.0) || (T > 100.0);
    const double TK = T + 273.15;
    const double ex = -(45000.0 / 8.314) * (1.0 / TK - 1.0 / 293.15);
    p.kT = 0.0001 * exp(ex);
    if (T <= 8.0) {
        const double d = T - 4.0;
        p.rho = 999.97 + (-0.008 * (d * d));
    } else {
        p.rho = 998.2 + ((-2.1e-4 * 998.2) * (T - 20.0));
    }
    return p;
}

// One row-triple (dpH, dCl, dT) of derivatives() for this lane's zone, given the
// lane's own (possibly perturbed / stage) values; neighbour values come from the
// adjacent lanes' arguments to the same call.  reactor.py:304-443.
__device__ __forceinline__ void rhs_rows(const Lane &L, const RK &k, const PropPH &pp, const PropT &pt,
                                         double Cl, double T, double f[3])
{
    // mixing suppression of the interface above this zone (spatial.py:239-320)
    const double rho_hi = shfl_hi(pt.rho);
    double s = 1.0;
    if (k.strat_mode == 1) {
        const double drho = rho_hi - pt.rho;
        const double ravg = 0.5 * (pt.rho + rho_hi);
        const double Ri = (9.81 * drho * k.dz) / (ravg * k.u2);
        if (Ri > k.ri_crit) s = k.supp;
    } else if (k.strat_mode == 2) {
        s = k.supp;
    }
    const double k_hi = L.has_hi ? k.Kex * s : 0.0;      // K[i,i+1]  reactor.py:321-325
    const double k_lo_raw = shfl_lo(k_hi);
    const double k_lo = L.has_lo ? k_lo_raw : 0.0;        // K[i,i-1]
    double kd = -(k_lo + k_hi);                           // reactor.py:329-332
    if (!L.has_hi) kd -= k.Qv;                            // reactor.py:337

    // shuffles are executed by every lane (never under a lane-dependent
    // condition: a masked-off source lane would read back as 0), then masked
    const double H_lo_r = shfl_lo(pp.H), H_hi_r = shfl_hi(pp.H);
    const double C_lo_r = shfl_lo(Cl), C_hi_r = shfl_hi(Cl);
    const double T_lo_r = shfl_lo(T), T_hi_r = shfl_hi(T);
    const double H_lo = L.has_lo ? H_lo_r : 0.0, H_hi = L.has_hi ? H_hi_r : 0.0;
    const double C_lo = L.has_lo ? C_lo_r : 0.0, C_hi = L.has_hi ? C_hi_r : 0.0;
    const double T_lo = L.has_lo ? T_lo_r : 0.0, T_hi = L.has_hi ? T_hi_r : 0.0;
    // K @ x with OpenBLAS' accumulation order: neighbours first, diagonal last
    const double mixH = (k_lo * H_lo + k_hi * H_hi) + kd * pp.H;
    const double mixC = (k_lo * C_lo + k_hi * C_hi) + kd * Cl;
    const double mixT = (k_lo * T_lo + k_hi * T_hi) + kd * T;

    double dpH = 0.0, dCl = 0.0, dT = 0.0;
    if (!L.has_lo) { // zone 0: dosing and inlet terms (reactor.py:349-368,388-395,420)
        if (k.has_acid && pp.bpos) dpH += (-k.acid_dH) * pp.iw;
        const double dH_in = k.Qv * (k.H_in - pp.H);
        if (pp.bpos) dpH += (-dH_in) * pp.iw;
        if (k.has_cl) dCl += k.cl_dose;
        dCl += k.Qv * (k.Cl_in - Cl);
        dT += k.Qv * (k.T_in - T);
    }
    if (pp.bpos) dpH += (-mixH) * pp.iw;                  // reactor.py:371-376
    dCl += mixC;                                          // reactor.py:398
    dCl -= (pt.kT * pp.phi) * Cl;                         // reactor.py:401-411
    dT += mixT;                                           // reactor.py:423
    if (k.has_heat) dT -= (k.UA * (T - k.T_amb)) * k.inv_rcv; // reactor.py:426-443
    f[SPH] = dpH; f[SCL] = dCl; f[STT] = dT;
}

// Full RHS at per-lane state y; returns true if the reference would have raised.
__device__ __forceinline__ bool rhs_full(const Lane &L, const RK &k, const double y[3], double f[3])
{
    const PropPH pp = prop_pH(k, y[SPH]);
    const PropT pt = prop_T(y[STT]);
    rhs_rows(L, k, pp, pt, y[SCL], y[STT], f);
    return pt.bad;
}

// ---------------------------------------------------------------- Jacobian bands
// Non-zero structure of d(dpH,dCl,dT)_i / d(pH,Cl,T)_j, j in {i-1,i,i+1}
// (index rel+1).  dT rows see only T; dpH rows see pH and (through the
// stratification switch only) T; dCl rows see Cl, own-zone pH and T.
struct Jac {
    double pp[3], cc[3], tt[3], pt[3], ct[3], cp;
};

// PCR-factored tridiagonal systems (real and complex shift)
template <int LV> struct TriR { double al[LV], ga[LV], invd; };
template <int LV> struct TriC { double alr[LV], ali[LV], gar[LV], gai[LV], invdr, invdi; };

template <int LV>
__device__ __forceinline__ void pcr_factor_real(const Lane &L, double a, double d, double c, TriR<LV> &F)
{
#pragma unroll
    for (int l = 0; l < LV; ++l) {
        const int s = 1 << l;
        const bool vlo = L.z - s >= 0, vhi = L.z + s < L.n;
        const double d_lo = shfl_lo(d, s), d_hi = shfl_hi(d, s);
        const double a_lo = shfl_lo(a, s), c_lo = shfl_lo(c, s);
        const double a_hi = shfl_hi(a, s), c_hi = shfl_hi(c, s);
        const double al = vlo ? a / d_lo : 0.0;
        const double ga = vhi ? c / d_hi : 0.0;
        d = d - al * (vlo ? c_lo : 0.0) - ga * (vhi ? a_hi : 0.0);
        a = vlo ? -al * a_lo : 0.0;
        c = vhi ? -ga * c_hi : 0.0;
        F.al[l] = al; F.ga[l] = ga;
    }
    F.invd = 1.0 / d;
}

template <int LV>
__device__ __forceinline__ double pcr_solve_real(const Lane &L, const TriR<LV> &F, double b)
{
#pragma unroll
    for (int l = 0; l < LV; ++l) {
        const int s = 1 << l;
        const bool vlo = L.z - s >= 0, vhi = L.z + s < L.n;
        const double b_lo = shfl_lo(b, s), b_hi = shfl_hi(b, s);
        b = b - F.al[l] * (vlo ? b_lo : 0.0) - F.ga[l] * (vhi ? b_hi : 0.0);
    }
    return b * F.invd;
}

struct cplx { double r, i; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
__device__ __forceinline__ cplx cinv(cplx a) { const double q = 1.0 / (a.r * a.r + a.i * a.i); return {a.r * q, -a.i * q}; }
__device__ __forceinline__ cplx cshfl_lo(cplx a, int s) { return {shfl_lo(a.r, s), shfl_lo(a.i, s)}; }
__device__ __forceinline__ cplx cshfl_hi(cplx a, int s) { return {shfl_hi(a.r, s), shfl_hi(a.i, s)}; }

template <int LV>
__device__ __forceinline__ void pcr_factor_cplx(const Lane &L, double a0, cplx d, double c0, TriC<LV> &F)
{
    cplx a = {a0, 0.0}, c = {c0, 0.0};
    const cplx zero = {0.0, 0.0};
#pragma unroll
    for (int l = 0; l < LV; ++l) {
        const int s = 1 << l;
        const bool vlo = L.z - s >= 0, vhi = L.z + s < L.n;
        const cplx d_lo = cshfl_lo(d, s), d_hi = cshfl_hi(d, s);
        const cplx a_lo = cshfl_lo(a, s), c_lo = cshfl_lo(c, s);
        const cplx a_hi = cshfl_hi(a, s), c_hi = cshfl_hi(c, s);
        const cplx al = vlo ? cmul(a, cinv(d_lo)) : zero;
        const cplx ga = vhi ? cmul(c, cinv(d_hi)) : zero;
        const cplx t1 = cmul(al, vlo ? c_lo : zero), t2 = cmul(ga, vhi ? a_hi : zero);
        d = {d.r - t1.r - t2.r, d.i - t1.i - t2.i};
        const cplx na = cmul(al, a_lo), nc = cmul(ga, c_hi);
        a = vlo ? cplx{-na.r, -na.i} : zero;
        c = vhi ? cplx{-nc.r, -nc.i} : zero;
        F.alr[l] = al.r; F.ali[l] = al.i; F.gar[l] = ga.r; F.gai[l] = ga.i;
    }
    const cplx inv = cinv(d);
    F.invdr = inv.r; F.invdi = inv.i;
}

template <int LV>
__device__ __forceinline__ cplx pcr_solve_cplx(const Lane &L, const TriC<LV> &F, cplx b)
{
    const cplx zero = {0.0, 0.0};
#pragma unroll
    for (int l = 0; l < LV; ++l) {
        const int s = 1 << l;
        const bool vlo = L.z - s >= 0, vhi = L.z + s < L.n;
        const cplx b_lo_r = cshfl_lo(b, s), b_hi_r = cshfl_hi(b, s);
        const cplx b_lo = vlo ? b_lo_r : zero;
        const cplx b_hi = vhi ? b_hi_r : zero;
        const cplx t1 = cmul({F.alr[l], F.ali[l]}, b_lo), t2 = cmul({F.gar[l], F.gai[l]}, b_hi);
        b = {b.r - t1.r - t2.r, b.i - t1.i - t2.i};
    }
    return cmul(b, {F.invdr, F.invdi});
}

// The six factored systems of one (h, J) pair: scipy's LU_real / LU_complex.
template <int LV> struct Factors {
    TriR<LV> rT, rP, rC;
    TriC<LV> cT, cP, cC;
};

template <int LV>
__device__ __forceinline__ void factorize(const Lane &L, const RadauConsts &rc, const Jac &J, double h, Factors<LV> &F)
{
    // radau.py:454-456: MU_REAL / h * I - J ; MU_COMPLEX / h * I - J
    const double mr = rc.MU_REAL / h, mcr = rc.MU_CR / h, mci = rc.MU_CI / h;
    pcr_factor_real<LV>(L, -J.tt[0], mr - J.tt[1], -J.tt[2], F.rT);
    pcr_factor_real<LV>(L, -J.pp[0], mr - J.pp[1], -J.pp[2], F.rP);
    pcr_factor_real<LV>(L, -J.cc[0], mr - J.cc[1], -J.cc[2], F.rC);
    pcr_factor_cplx<LV>(L, -J.tt[0], {mcr - J.tt[1], mci}, -J.tt[2], F.cT);
    pcr_factor_cplx<LV>(L, -J.pp[0], {mcr - J.pp[1], mci}, -J.pp[2], F.cP);
    pcr_factor_cplx<LV>(L, -J.cc[0], {mcr - J.cc[1], mci}, -J.cc[2], F.cC);
}

// x = (mu_real/h I - J)^-1 b, in place, b indexed by species
template <int LV>
__device__ __forceinline__ void solve_real(const Lane &L, const Jac &J, const Factors<LV> &F, double b[3])
{
    const double xT = pcr_solve_real<LV>(L, F.rT, b[STT]);
    const double xT_lo_r = shfl_lo(xT), xT_hi_r = shfl_hi(xT);
    const double xT_lo = L.has_lo ? xT_lo_r : 0.0, xT_hi = L.has_hi ? xT_hi_r : 0.0;
    const double bp = b[SPH] + (J.pt[0] * xT_lo + J.pt[1] * xT + J.pt[2] * xT_hi);
    const double xP = pcr_solve_real<LV>(L, F.rP, bp);
    const double bc = b[SCL] + (J.ct[0] * xT_lo + J.ct[1] * xT + J.ct[2] * xT_hi) + J.cp * xP;
    const double xC = pcr_solve_real<LV>(L, F.rC, bc);
    b[SPH] = xP; b[SCL] = xC; b[STT] = xT;
}

template <int LV>
__device__ __forceinline__ void solve_cplx(const Lane &L, const Jac &J, const Factors<LV> &F, double br[3], double bi[3])
{
    const cplx zero = {0.0, 0.0};
    const cplx xT = pcr_solve_cplx<LV>(L, F.cT, {br[STT], bi[STT]});
    const cplx xT_lo_r = cshfl_lo(xT, 1), xT_hi_r = cshfl_hi(xT, 1);
    const cplx xT_lo = L.has_lo ? xT_lo_r : zero, xT_hi = L.has_hi ? xT_hi_r : zero;
    const cplx bp = {br[SPH] + (J.pt[0] * xT_lo.r + J.pt[1] * xT.r + J.pt[2] * xT_hi.r),
                     bi[SPH] + (J.pt[0] * xT_lo.i + J.pt[1] * xT.i + J.pt[2] * xT_hi.i)};
    const cplx xP = pcr_solve_cplx<LV>(L, F.cP, bp);
    const cplx bc = {br[SCL] + (J.ct[0] * xT_lo.r + J.ct[1] * xT.r + J.ct[2] * xT_hi.r) + J.cp * xP.r,
                     bi[SCL] + (J.ct[0] * xT_lo.i + J.ct[1] * xT.i + J.ct[2] * xT_hi.i) + J.cp * xP.i};
    const cplx xC = pcr_solve_cplx<LV>(L, F.cC, bc);
    br[SPH] = xP.r; bi[SPH] = xP.i; br[SCL] = xC.r; bi[SCL] = xC.i; br[STT] = xT.r; bi[STT] = xT.i;
}

// ---------------------------------------------------------------- num_jac (common.py:257-382)
// Forward differences restated for the banded structure: perturbing zone j only
// changes rows of zones j-1..j+1, so zones of equal (j mod 3) are perturbed
// together (three colours per species) and each lane attributes the change of its
// rows to the single perturbed zone in its stencil.  The perturbed zone-local
// properties are evaluated once per species.  Per column this reproduces
// common.py's f(y + h e_j) - f(y) for the rows that can change; the step-size
// bookkeeping (factor growth/shrink, the one retry with 10x factor) is scipy's.
struct FdCols { double D[3][3]; double S[3][3]; }; // [row species][rel+1]: diff and max(|f|,|f_new|)

__device__ __forceinline__ void fd_species_pass(const Lane &L, const RK &k, int sp, const double y[3],
                                                const double f[3], const PropPH &bpp, const PropT &bpt,
                                                double hcol, bool colmask, FdCols &out, bool &bad)
{
    const double ypert = y[sp] + hcol;
    PropPH ppp = bpp; PropT ppt = bpt;
    if (sp == SPH) ppp = prop_pH(k, ypert);
    if (sp == STT) { ppt = prop_T(ypert); bad = bad || (colmask && ppt.bad); }
    const int zm = L.z % 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const bool mine = (zm == c) && colmask;
        PropPH p1 = bpp; PropT p2 = bpt;
        if (sp == SPH && mine) p1 = ppp;
        if (sp == STT && mine) p2 = ppt;
        const double cl = (sp == SCL && mine) ? ypert : y[SCL];
        const double tt = (sp == STT && mine) ? ypert : y[STT];
        double fn[3];
        rhs_rows(L, k, p1, p2, cl, tt, fn);
        const int rel1 = (c - zm + 4) % 3; // (column zone - this zone) + 1
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (rel1 == r) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    out.D[q][r] = fn[q] - f[q];
                    out.S[q][r] = fmax(fabs(f[q]), fabs(fn[q]));
                }
            }
        }
    }
}

// For the column owned by this lane (species sp): max |diff| over its rows with
// numpy argmax tie-breaking (first row in [pH.., Cl.., T..] order) and the
// matching scale (common.py:335-339).
__device__ __forceinline__ void fd_col_reduce(const Lane &L, int sp, const FdCols &c, double &maxd, double &scale)
{
    maxd = -1.0; scale = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        // rows of species q that can depend on a column of species sp
        const bool dep_nb = (q == sp) || (sp == STT);           // neighbour-zone rows
        const bool dep_own = dep_nb || (q == SCL && sp == SPH); // own-zone row
        if (!dep_own) continue;
        double d_lo = 0, s_lo = 0, d_hi = 0, s_hi = 0;
        if (dep_nb) {
            d_lo = shfl_lo(c.D[q][2]); s_lo = shfl_lo(c.S[q][2]); // lane z-1 saw this column at rel=+1
            d_hi = shfl_hi(c.D[q][0]); s_hi = shfl_hi(c.S[q][0]); // lane z+1 saw it at rel=-1
        }
        if (dep_nb && L.has_lo && fabs(d_lo) > maxd) { maxd = fabs(d_lo); scale = s_lo; }
        if (fabs(c.D[q][1]) > maxd) { maxd = fabs(c.D[q][1]); scale = c.S[q][1]; }
        if (dep_nb && L.has_hi && fabs(d_hi) > maxd) { maxd = fabs(d_hi); scale = s_hi; }
    }
}

__device__ __forceinline__ double fd_step(double y, double fac, double ysc)
{
    // h = (y + factor*y_scale) - y, evaluated without fusing (common.py:323)
    return __dadd_rn(__dadd_rn(y, __dmul_rn(fac, ysc)), -y);
}

__device__ __forceinline__ void num_jac(const Lane &L, const RK &k, const RadauConsts &rc, const double y[3],
                                        const double f[3], double fac[3], bool &have_fac, Jac &J, bool &bad)
{
    if (!have_fac) { fac[0] = fac[1] = fac[2] = rc.NJ_F0; have_fac = true; }
    const PropPH bpp = prop_pH(k, y[SPH]);
    const PropT bpt = prop_T(y[STT]);
    double hcol[3], maxd[3], scl[3];
    FdCols cols[3];
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
        const double fs = (f[sp] >= 0) ? 1.0 : -1.0;
        const double ysc = fs * fmax(ATOL, fabs(y[sp]));
        double h = fd_step(y[sp], fac[sp], ysc);
        while (h == 0) { fac[sp] *= 10; h = fd_step(y[sp], fac[sp], ysc); } // common.py:327-330
        hcol[sp] = h;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 3; ++r) { cols[sp].D[q][r] = 0.0; cols[sp].S[q][r] = 0.0; }
        fd_species_pass(L, k, sp, y, f, bpp, bpt, h, true, cols[sp], bad);
        fd_col_reduce(L, sp, cols[sp], maxd[sp], scl[sp]);
        const bool small = maxd[sp] < rc.NJ_REJECT * scl[sp];       // common.py:341
        if (__ballot(small) != 0ull) {                               // rare: one retry with 10x factor
            const double nf = 10 * fac[sp];
            const double hn = fd_step(y[sp], nf, ysc);
            FdCols c2;
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int r = 0; r < 3; ++r) { c2.D[q][r] = 0.0; c2.S[q][r] = 0.0; }
            fd_species_pass(L, k, sp, y, f, bpp, bpt, hn, small, c2, bad);
            double md2, sc2;
            fd_col_reduce(L, sp, c2, md2, sc2);
            const bool upd = small && (maxd[sp] * sc2 < md2 * scl[sp]); // common.py:354
            if (upd) { fac[sp] = nf; hcol[sp] = hn; maxd[sp] = md2; scl[sp] = sc2; }
            const int iu = upd ? 1 : 0;
            const int iu_lo = __shfl_up(iu, 1, 64), iu_hi = __shfl_down(iu, 1, 64);
            const bool upd_lo = L.has_lo && (iu_lo != 0);
            const bool upd_hi = L.has_hi && (iu_hi != 0);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (upd_lo) cols[sp].D[q][0] = c2.D[q][0];
                if (upd) cols[sp].D[q][1] = c2.D[q][1];
                if (upd_hi) cols[sp].D[q][2] = c2.D[q][2];
            }
        }
    }
    // diff /= h  (column-wise; the column's h lives in the column's lane)
    double ih[3][3];
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
        const double h_lo = shfl_lo(hcol[sp]), h_hi = shfl_hi(hcol[sp]);
        ih[sp][0] = L.has_lo ? h_lo : 1.0; ih[sp][1] = hcol[sp]; ih[sp][2] = L.has_hi ? h_hi : 1.0;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const bool valid = (r == 1) || (r == 0 ? L.has_lo : L.has_hi);
        J.pp[r] = valid ? cols[SPH].D[SPH][r] / ih[SPH][r] : 0.0;
        J.cc[r] = valid ? cols[SCL].D[SCL][r] / ih[SCL][r] : 0.0;
        J.tt[r] = valid ? cols[STT].D[STT][r] / ih[STT][r] : 0.0;
        J.pt[r] = valid ? cols[STT].D[SPH][r] / ih[STT][r] : 0.0;
        J.ct[r] = valid ? cols[STT].D[SCL][r] / ih[STT][r] : 0.0;
    }
    J.cp = cols[SPH].D[SCL][1] / ih[SPH][1];
    // factor adaptation common.py:363-365
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
        const bool sm = maxd[sp] < rc.NJ_SMALL * scl[sp];
        const bool bg = maxd[sp] > rc.NJ_BIG * scl[sp];
        if (sm) fac[sp] *= 10;
        if (bg) fac[sp] *= 0.1;
        fac[sp] = fmax(fac[sp], rc.NJ_MINF);
    }
}

// ---------------------------------------------------------------- helpers
__device__ __forceinline__ double rms3(const Lane &L, const double x[3], const double sc[3])
{
    // common.py:63-65 norm(x / scale) over the 3n components of one reactor
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) { const double v = x[q] / sc[q]; s += v * v; }
    return sqrt(seg_sum(L, s)) / sqrt((double)(3 * L.n));
}

__device__ __forceinline__ double ulp_above(double t)
{
    // |nextafter(t, +inf) - t| for t >= 0 (radau.py:408)
    return __longlong_as_double(__double_as_longlong(t) + 1) - t;
}

// radau.py:139-176
__device__ __forceinline__ double predict_factor(double h_abs, bool have_old, double h_abs_old,
                                                 double error_norm, double error_norm_old)
{
    double mult = 1.0;
    if (have_old && error_norm != 0) mult = h_abs / h_abs_old * pow(error_norm_old / error_norm, 0.25);
    return fmin(1.0, mult) * pow(error_norm, -0.25);
}

// ---------------------------------------------------------------- one outer step
struct SolverCounters { int nfev, njev, nlu, nsteps, nrej; };

// Advances y (per-lane pH, Cl, T) from t to t+dt exactly as
// solve_ivp(method="Radau", rtol=1e-6, atol=1e-8, max_step=min(dt,10)) would
// (reactor.py:476-484).  Returns status bits (ST_T_RANGE => y untouched).
template <int LV>
__device__ __forceinline__ uint32_t radau_outer_step(const Lane &L, const RK &k, const RadauConsts &rc,
                                                     double y[3], double t0, double dt, SolverCounters &cnt)
{
    const double t_bound = t0 + dt;
    const double max_step = fmin(dt, 10.0);
    const double newton_tol = rc.newton_tol;
    cnt = {0, 0, 0, 0, 0};
    if (t0 == t_bound) return 0;

    bool bad = false; // a T outside [0,100] reached the Arrhenius routine
    double yc[3] = {y[0], y[1], y[2]};
    double f[3];
    bad |= rhs_full(L, k, yc, f); cnt.nfev++;                         // radau.py:303

    // ---- select_initial_step (common.py:68-134), order 3
    double h_abs;
    {
        double sc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) sc[q] = ATOL + fabs(yc[q]) * RTOL;
        const double d0 = rms3(L, yc, sc), d1 = rms3(L, f, sc);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        const double interval = fabs(t_bound - t0);
        h0 = fmin(h0, interval);
        double y1[3], f1[3], df[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) y1[q] = yc[q] + h0 * f[q];
        bad |= rhs_full(L, k, y1, f1); cnt.nfev++;
#pragma unroll
        for (int q = 0; q < 3; ++q) df[q] = f1[q] - f[q];
        const double d2 = rms3(L, df, sc) / h0;
        double h1;
        if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
        else h1 = pow(0.01 / fmax(d1, d2), 0.25);
        h_abs = fmin(fmin(100 * h0, h1), fmin(interval, max_step));
    }

    // ---- solver object state (radau.py:295-346)
    Jac J;
    double fac[3]; bool have_fac = false;
    num_jac(L, k, rc, yc, f, fac, have_fac, J, bad); cnt.njev++;      // radau.py:359-365
    bool current_jac = true, have_lu = false, have_sol = false, have_old = false;
    double h_abs_old = 0.0, error_norm_old = 0.0;
    double lu_h = 1.0;
    Factors<LV> F;
    double Q[3][3];        // dense output coefficients [species][power]
    double y_old[3] = {0, 0, 0};
    double sol_t_old = 0.0, sol_h = 1.0;
    double t = t0;
    uint32_t status = 0;

    bad = seg_any(L, bad);
    // ---- base.py:182-197 / ivp.py:653 loop: one _step_impl per iteration
    while (!bad && (t - t_bound) < 0) {
        const double min_step = 10 * fabs(ulp_above(t));            // radau.py:408
        double h_abs_l, h_abs_old_l = 0.0, err_old_l = 0.0; bool have_old_l;
        if (h_abs > max_step) { h_abs_l = max_step; have_old_l = false; }
        else if (h_abs < min_step) { h_abs_l = min_step; have_old_l = false; }
        else { h_abs_l = h_abs; have_old_l = have_old; h_abs_old_l = h_abs_old; err_old_l = error_norm_old; }

        bool rejected = false, accepted = false, failed = false;
        double Z[3][3];    // [stage][species]
        double h = 0, t_new = 0, error_norm = 0, safety = 0, rate = 0;
        int n_iter = 0; bool have_rate = false;
        double y_new[3];

        while (!accepted) {
            if (h_abs_l < min_step) { failed = true; break; }        // radau.py:427-428
            h = h_abs_l;
            t_new = t + h;
            if (t_new - t_bound > 0) t_new = t_bound;
            h = t_new - t;
            h_abs_l = fabs(h);

            double Z0[3][3];
            if (!have_sol) {
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int q = 0; q < 3; ++q) Z0[s][q] = 0.0;
            } else {                                                  // radau.py:445-448,557-572
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const double x = ((t + h * rc.C[s]) - sol_t_old) / sol_h;
                    const double p1 = x * x, p2 = p1 * x;
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        Z0[s][q] = ((Q[q][0] * x + Q[q][1] * p1 + Q[q][2] * p2) + y_old[q]) - yc[q];
                }
            }
            double scale[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) scale[q] = ATOL + fabs(yc[q]) * RTOL;

            bool converged = false;
            while (!converged) {
                if (!have_lu) { factorize<LV>(L, rc, J, h, F); lu_h = h; have_lu = true; cnt.nlu += 2; }
                // ---- solve_collocation_system radau.py:48-136
                const double M_real = rc.MU_REAL / h, Mcr = rc.MU_CR / h, Mci = rc.MU_CI / h;
                double W[3][3];
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        W[s][q] = rc.TI[s][0] * Z0[0][q] + rc.TI[s][1] * Z0[1][q] + rc.TI[s][2] * Z0[2][q];
                        Z[s][q] = Z0[s][q];
                    }
                double dW_norm_old = 0.0; bool have_norm_old = false;
                have_rate = false; rate = 0.0;
                int kk;
                for (kk = 0; kk < NEWTON_MAXITER; ++kk) {
                    double Fs[3][3];
                    bool finite = true;
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        double ys[3];
#pragma unroll
                        for (int q = 0; q < 3; ++q) ys[q] = yc[q] + Z[s][q];
                        bad |= rhs_full(L, k, ys, Fs[s]); cnt.nfev++;
#pragma unroll
                        for (int q = 0; q < 3; ++q) finite = finite && isfinite(Fs[s][q]);
                    }
                    if (!seg_all(L, finite)) break;
                    double fr[3], fcr[3], fci[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        fr[q] = (Fs[0][q] * rc.TI[0][0] + Fs[1][q] * rc.TI[0][1] + Fs[2][q] * rc.TI[0][2]) - M_real * W[0][q];
                        const double re = Fs[0][q] * rc.TI[1][0] + Fs[1][q] * rc.TI[1][1] + Fs[2][q] * rc.TI[1][2];
                        const double im = Fs[0][q] * rc.TI[2][0] + Fs[1][q] * rc.TI[2][1] + Fs[2][q] * rc.TI[2][2];
                        fcr[q] = re - (Mcr * W[1][q] - Mci * W[2][q]);
                        fci[q] = im - (Mcr * W[2][q] + Mci * W[1][q]);
                    }
                    solve_real<LV>(L, J, F, fr);
                    solve_cplx<LV>(L, J, F, fcr, fci);
                    double ssum = 0.0;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const double a = fr[q] / scale[q], b = fcr[q] / scale[q], c = fci[q] / scale[q];
                        ssum += a * a + b * b + c * c;
                    }
                    const double dW_norm = sqrt(seg_sum(L, ssum)) / sqrt((double)(9 * L.n));
                    if (have_norm_old) { rate = dW_norm / dW_norm_old; have_rate = true; }
                    if (have_rate && (rate >= 1 || pow(rate, (double)(NEWTON_MAXITER - kk)) / (1 - rate) * dW_norm > newton_tol))
                        break;
#pragma unroll
                    for (int q = 0; q < 3; ++q) { W[0][q] += fr[q]; W[1][q] += fcr[q]; W[2][q] += fci[q]; }
#pragma unroll
                    for (int s = 0; s < 3; ++s)
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            Z[s][q] = rc.T[s][0] * W[0][q] + rc.T[s][1] * W[1][q] + rc.T[s][2] * W[2][q];
                    if (dW_norm == 0 || (have_rate && rate / (1 - rate) * dW_norm < newton_tol)) { converged = true; break; }
                    dW_norm_old = dW_norm; have_norm_old = true;
                }
                n_iter = (kk == NEWTON_MAXITER) ? NEWTON_MAXITER : kk + 1;
                if (seg_any(L, bad)) { bad = true; break; }
                if (!converged) {                                     // radau.py:462-470
                    if (current_jac) break;
                    num_jac(L, k, rc, yc, f, fac, have_fac, J, bad); cnt.njev++;
                    current_jac = true; have_lu = false;
                    if (seg_any(L, bad)) { bad = true; break; }
                }
            }
            if (bad) break;
            if (!converged) {                                         // radau.py:472-476
                h_abs_l *= 0.5; have_lu = false; cnt.nrej++;
                continue;
            }
            double ZE[3], err[3], esc[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                y_new[q] = yc[q] + Z[2][q];
                ZE[q] = (Z[0][q] * rc.E[0] + Z[1][q] * rc.E[1] + Z[2][q] * rc.E[2]) / h;
                err[q] = f[q] + ZE[q];
            }
            solve_real<LV>(L, J, F, err);
#pragma unroll
            for (int q = 0; q < 3; ++q) esc[q] = ATOL + fmax(fabs(yc[q]), fabs(y_new[q])) * RTOL;
            error_norm = rms3(L, err, esc);
            safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + n_iter);
            if (rejected && error_norm > 1) {                         // radau.py:485-487
                double yt[3], ft[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) yt[q] = yc[q] + err[q];
                bad |= rhs_full(L, k, yt, ft); cnt.nfev++;
                if (seg_any(L, bad)) { bad = true; break; }
#pragma unroll
                for (int q = 0; q < 3; ++q) err[q] = ft[q] + ZE[q];
                solve_real<LV>(L, J, F, err);
                error_norm = rms3(L, err, esc);
            }
            if (error_norm > 1) {                                     // radau.py:489-496
                const double fct = predict_factor(h_abs_l, have_old_l, h_abs_old_l, error_norm, err_old_l);
                h_abs_l *= fmax(MIN_FACTOR, safety * fct);
                have_lu = false; rejected = true; cnt.nrej++;
            } else {
                accepted = true;
            }
        }
        if (bad) break;
        if (failed) { status |= ST_SOLVER_FAILED; break; }

        // ---- accepted: radau.py:500-539
        const bool recompute_jac = (n_iter > 2) && have_rate && (rate > 1e-3);
        double fct = predict_factor(h_abs_l, have_old_l, h_abs_old_l, error_norm, err_old_l);
        fct = fmin(MAX_FACTOR, safety * fct);
        if (!recompute_jac && fct < 1.2) fct = 1.0; else have_lu = false;
        double f_new[3];
        bad |= rhs_full(L, k, y_new, f_new); cnt.nfev++;
        if (seg_any(L, bad)) { bad = true; break; }
        if (recompute_jac) {
            num_jac(L, k, rc, y_new, f_new, fac, have_fac, J, bad); cnt.njev++;
            if (seg_any(L, bad)) { bad = true; break; }
            current_jac = true;
        } else {
            current_jac = false;
        }
        h_abs_old = h_abs;            // sic radau.py:520: the solver-level value
        error_norm_old = error_norm;
        have_old = true;
        h_abs = h_abs_l * fct;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            y_old[q] = yc[q];
#pragma unroll
            for (int p = 0; p < 3; ++p) // Q = Z^T P  radau.py:541-543
                Q[q][p] = Z[0][q] * rc.P[0][p] + Z[1][q] * rc.P[1][p] + Z[2][q] * rc.P[2][p];
            yc[q] = y_new[q]; f[q] = f_new[q];
        }
        sol_t_old = t; sol_h = t_new - t; have_sol = true;
        t = t_new;
        cnt.nsteps++;
    }
    (void)lu_h;
    if (bad) return ST_T_RANGE;  // the reference raised: self.state untouched
    y[0] = yc[0]; y[1] = yc[1]; y[2] = yc[2];
    return status;
}

// ---------------------------------------------------------------- kernels
__device__ __forceinline__ bool lane_setup(const StepArgs &a, Lane &L, int64_t &r)
{
    const int lane = threadIdx.x & 63;
    const int n = a.n;
    const int seg = lane / n;
    L.n = n; L.z = lane - seg * n;
    L.base = seg * n;
    L.has_lo = L.z > 0; L.has_hi = L.z < n - 1;
    L.segmask = ((n >= 64) ? ~0ull : ((1ull << n) - 1ull)) << L.base;
    r = (int64_t)blockIdx.x * a.R + seg;
    return (seg < a.R) && (r < a.N);
}

template <int LV>
__global__ __launch_bounds__(64) void step_kernel(const StepArgs a)
{
    Lane L; int64_t r;
    if (!lane_setup(a, L, r)) return;
    const int64_t idx = r * a.n + L.z;
    uint32_t st = a.status[r];
    // a reactor whose last step raised stays frozen until the host rewrites its state
    if (st & (ST_T_RANGE | ST_T_RANGE_POST)) return;
    RK k; load_reactor(a, r, a.n, k);
    double y[3] = {a.pH[idx], a.Cl[idx], a.T[idx]};
    double t = a.time[r];
    SolverCounters cnt = {0, 0, 0, 0, 0};
    bool wrote_k = false; double dH = 0, dR = 0, dK = 0;
    bool advanced = false;
    for (int step = 0; step < a.n_steps; ++step) {
        const uint32_t s1 = radau_outer_step<LV>(L, k, a.rc, y, t, a.dt, cnt);
        st |= s1;
        if (s1 & ST_T_RANGE) break;
        advanced = true;
        t = t + a.dt;                                           // reactor.py:496
        // _update_derived_state reactor.py:511-524 (before the clamp)
        dH = exp10(-y[SPH]);
        const PropT pt = prop_T(y[STT]);
        dR = pt.rho;
        if (seg_any(L, pt.bad)) { st |= ST_T_RANGE_POST; break; }
        dK = pt.kT; wrote_k = true;
        // _enforce_physical_bounds reactor.py:526-541
        if (seg_any(L, y[SPH] < 0 || y[SPH] > 14)) { st |= ST_CLAMP_PH; y[SPH] = fmin(fmax(y[SPH], 0.0), 14.0); }
        if (seg_any(L, y[SCL] < 0)) { st |= ST_CLAMP_CL; y[SCL] = fmax(y[SCL], 0.0); }
        if (seg_any(L, y[STT] < 0 || y[STT] > 100)) { st |= ST_CLAMP_T; y[STT] = fmin(fmax(y[STT], 0.0), 100.0); }
        if (seg_any(L, !(isfinite(y[0]) && isfinite(y[1]) && isfinite(y[2])))) st |= ST_NONFINITE;
    }
    if (advanced) {
        a.pH[idx] = y[SPH]; a.Cl[idx] = y[SCL]; a.T[idx] = y[STT];
        a.dH[idx] = dH; a.dRho[idx] = dR;
        if (wrote_k) a.dK[idx] = dK;
    }
    if (L.z == 0) {
        if (advanced) {
            a.time[r] = t;
            // reactor.py:497-501
            a.flow[r] = a.bc[0 * a.N + r] + a.bc[4 * a.N + r] + a.bc[6 * a.N + r];
        }
        a.status[r] = st;
        if (a.stats) {
            int32_t *o = a.stats + r * 5;
            o[0] = cnt.nfev; o[1] = cnt.njev; o[2] = cnt.nlu; o[3] = cnt.nsteps; o[4] = cnt.nrej;
        }
    }
}

// derivatives() at caller-supplied states, for parity tests (reactor.py:272-448)
struct RhsArgs {
    int64_t N; int n; int R;
    const double *par, *bc;
    const double *pH, *Cl, *T;
    double *dpH, *dCl, *dT;
    uint32_t *flags;
};

__global__ __launch_bounds__(64) void rhs_kernel(const RhsArgs a)
{
    StepArgs sa{};
    sa.N = a.N; sa.n = a.n; sa.R = a.R; sa.par = a.par; sa.bc = a.bc;
    Lane L; int64_t r;
    if (!lane_setup(sa, L, r)) return;
    const int64_t idx = r * a.n + L.z;
    RK k; load_reactor(sa, r, a.n, k);
    double y[3] = {a.pH[idx], a.Cl[idx], a.T[idx]}, f[3];
    const bool bad = rhs_full(L, k, y, f);
    a.dpH[idx] = f[SPH]; a.dCl[idx] = f[SCL]; a.dT[idx] = f[STT];
    const bool anybad = seg_any(L, bad);
    if (L.z == 0) a.flags[r] = anybad ? ST_T_RANGE : 0u;
}

// AqueousChemistry.calculate_pH (chemistry.py:271-330), one system per thread.
struct PhArgs {
    int64_t n;
    const double *Kw, *Ka1, *Ka2, *CT, *alk, *guess;
    double tol; int max_iter;
    double *pH; int32_t *iters; int32_t *rc;
};

__global__ __launch_bounds__(256) void ph_solve_kernel(const PhArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const double Kw = a.Kw[i], Ka1 = a.Ka1[i], Ka2 = a.Ka2[i], CT = a.CT[i];
    const double alk_eq = a.alk[i] / 50000.0;                 // chemistry.py:223
    double pH = a.guess[i];
    int rc = 2, it = 0;
    for (it = 0; it < a.max_iter; ++it) {
        // charge_balance_error chemistry.py:193-228
        const double H = exp10(-pH);
        const double OH = Kw / H;
        const double H2 = H * H;
        const double D = H2 + Ka1 * H + Ka1 * Ka2;
        const double a1 = (Ka1 * H) / D, a2 = (Ka1 * Ka2) / D;
        const double fval = H - OH + a1 * CT + 2 * (a2 * CT) - alk_eq;
        // charge_balance_derivative chemistry.py:230-269
        const double dH = -LN10 * H;
        const double dOH = -(Kw / H2) * dH;
        const double dD = 2 * H + Ka1;
        const double D2 = D * D;
        const double da1 = Ka1 * (D - H * dD) / D2;
        const double da2 = -Ka1 * Ka2 * dD / D2;
        const double df = dH - dOH + CT * da1 * dH + 2 * (CT * da2 * dH);
        if (fabs(df) < 1e-15) { rc = 1; break; }              // chemistry.py:309-312
        const double delta = -fval / df;
        const double pH_new = fmin(fmax(pH + delta, 0.0), 14.0);
        if (fabs(delta) < a.tol) { pH = pH_new; rc = 0; ++it; break; }
        pH = pH_new;
    }
    a.pH[i] = pH; a.iters[i] = it; a.rc[i] = rc;
}

} // namespace wt
