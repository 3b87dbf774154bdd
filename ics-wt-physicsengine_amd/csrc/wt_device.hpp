// wt_device.hpp -- gfx950 device code of the multi-zone CSTR physics step.
//
// Mapping (CDNA4, 64-wide wavefronts): one LANE per reactor zone, the n zones
// of a reactor in n consecutive lanes, floor(64/n) reactors per wavefront, one
// wavefront per workgroup.  Everything a reactor needs for a whole outer step
// (state, Radau iterates, Jacobian bands, tridiagonal factors) lives in that
// segment's registers; the 1-D inter-zone stencil, the tridiagonal solves
// (parallel cyclic reduction) and the RMS norms are DPP / wavefront shuffles (zone counts that straddle a
// 16-lane DPP row exchange strides >= 2 through a row of LDS instead: both()).
// HBM is touched once per work item (a wavefront's reactors advanced by a few outer steps): state in,
// state + derived out.  Work items come from a device-side queue, so one launch advances the whole
// ensemble by any number of outer steps and no wavefront waits for a launch boundary; the sensor
// suite and the PLC register image / command path of an outer step run on the same wavefront right
// behind it (wt_sensors.hpp, wt_plc.hpp).
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
//
// Control structure: the solver is a per-reactor state machine (Phase) driven by
// one wave-wide loop.  Each trip evaluates the RHS once (three Radau stages for
// reactors in their Newton phase, one point for the others), so reactors of a
// wavefront that are at different places of scipy's algorithm (more Newton
// iterations, a rejected step, an extra internal step) still share the
// expensive transcendental work instead of waiting for each other phase by
// phase, and every heavy block exists exactly once in the code object.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "wt_sensors.hpp"

namespace wt {

#ifdef WT_STAMPS
constexpr int WT_DIAG_SLOTS = 16;
#else
constexpr int WT_DIAG_SLOTS = 8;
#endif

constexpr int SPH = 0, SCL = 1, STT = 2;  // species index inside a lane
// Branch weights matter beyond the branch: the register allocator keeps in VGPRs what the frequent blocks use
// and parks the rest in AGPRs, so the rare paths of the solver loop are marked as such.
#define WT_RARE(x) __builtin_expect(!!(x), 0)
#define WT_USUAL(x) __builtin_expect(!!(x), 1)
constexpr double RTOL = 1e-6, ATOL = 1e-8; // reactor.py:481-483
constexpr int NEWTON_MAXITER = 6;          // radau.py:43
constexpr double MIN_FACTOR = 0.2, MAX_FACTOR = 10.0;

// status bits (include/wtphys.h)
constexpr uint32_t ST_T_RANGE = 1, ST_SOLVER_FAILED = 2, ST_CLAMP_PH = 4, ST_CLAMP_CL = 8,
                   ST_CLAMP_T = 16, ST_T_RANGE_POST = 32, ST_NONFINITE = 64, ST_STEP_LIMIT = 128;

// Radau IIA / num_jac constants with the values scipy's module-level expressions
// produce (radau.py:11-40, common.py:248-253), as exact hex literals.
namespace rc {
constexpr double C0 = 0x1.3d8b64657caeap-3;       // (4 - sqrt6)/10
constexpr double C1 = 0x1.4a36c0803a6dfp-1;       // (4 + sqrt6)/10
constexpr double E0 = -0x1.418fd8baffe05p+3, E1 = 0x1.61d41b2d54580p+0, E2 = -0x1.5555555555555p-2;
constexpr double MU_REAL = 0x1.d1a48d83e731dp+1;  // 3.637834252744496
constexpr double MU_CR = 0x1.572db93e0c672p+1;    // 2.6810828736277523
constexpr double MU_CI = -0x1.86747f2c3fcb6p+1;   // -3.050430199247411
constexpr double T00 = 0.09443876248897524, T01 = -0.14125529502095421, T02 = 0.03002919410514742;
constexpr double T10 = 0.25021312296533332, T11 = 0.20412935229379994, T12 = -0.38294211275726192;
// T[2] = [1, 1, 0]
constexpr double TI00 = 4.17871859155190428, TI01 = 0.32768282076106237, TI02 = 0.52337644549944951;
constexpr double TI10 = -4.17871859155190428, TI11 = -0.32768282076106237, TI12 = 0.47662355450055044;
constexpr double TI20 = 0.50287263494578682, TI21 = -2.57192694985560522, TI22 = 0.59603920482822492;
constexpr double P00 = 0x1.418fd8baffe05p+3, P01 = -0x1.9a12ce7b30915p+4, P02 = 0x1.f295c43b61425p+3;
constexpr double P10 = -0x1.61d41b2d54580p+0, P11 = 0x1.497af24bb677ep+3, P12 = -0x1.1d406ee60becfp+3;
constexpr double P20 = 0x1.5555555555555p-2, P21 = -0x1.5555555555555p+1, P22 = 0x1.aaaaaaaaaaaabp+1;
constexpr double NJ_REJECT = 0x1.6a09e667f3bcdp-46; // EPS**0.875
constexpr double NJ_SMALL = 0x1.0p-39;             // EPS**0.75
constexpr double NJ_BIG = 0x1.0p-13;               // EPS**0.25
constexpr double NJ_MINF = 0x1.f4p-43;             // 1e3*EPS
constexpr double NJ_F0 = 0x1.0p-26;                // EPS**0.5
constexpr double NEWTON_TOL = 0x1.0624dd2f1a9fcp-10; // max(10 EPS/rtol, min(0.03, sqrt(rtol))) = 1e-3
constexpr double LN10 = 0x1.26bb1bbb55516p+1;      // np.log(10)
constexpr double K_ARR = -0x1.5248ea03d1718p+12;   // -(45000/8.314)  thermodynamics.py:188
constexpr double INV_TREF = 0x1.bf1da5ca77e69p-9;  // 1/293.15
} // namespace rc


// ---------------------------------------------------------------- fp64 constants of the RHS as scalar loads
// A VALU instruction on gfx950 cannot carry a 64-bit literal: every fp64 constant that is not an inline constant
// reaches it through an SGPR pair, i.e. two s_mov_b32 -- and with one wavefront per SIMD a scalar move costs the
// same issue slot as an fp64 FMA (measured: tools/ubench/issue.hip).  The exponential's polynomial alone is ten
// such pairs per inlined copy of the RHS.  So the constants of a section sit in the kernel-argument block
// (filled by the host, wtphys.hip: make_args) and are fetched at the top of the section with s_load_dwordx16:
// eight constants per issue slot instead of half a constant.
// exp / exp10 follow OCML's algorithm with OCML's coefficients (checked bit for bit against the library's on the
// device over 4M arguments each: tools/ubench/expcheck.hip), so nothing changes numerically.
typedef double d8 __attribute__((ext_vector_type(8)));
struct alignas(64) KTab {
    // section P -- pH properties: 24 doubles
    double pc[10];                                                   // exp polynomial, degree-11 term first
    double log2_10, lg2_hi, lg2_lo, ln10_hi, ln10_lo, t_hi, t_lo;    // exp10 argument reduction and range
    double c2303, ln10, c002, pad_p[4];
    // section T -- temperature properties: 32 doubles
    double tc[10];
    double log2e, ln2_hi, ln2_lo, e_hi, e_lo;                        // exp argument reduction and range
    double k_arr, inv_tref, c27315, c1em4;                           // Arrhenius (thermodynamics.py:160-193)
    double rho_max, rho_an, rho20, rho_sl, c20, c8, c100;            // density branches (spatial.py:177-189), T range
    double dense_bias;                                               // developer knob WT_DENSE_COUPLING: 1.0 = every Jacobian counts as coupling rows to neighbours' T
    double pad_t[5];
};
static_assert(sizeof(KTab) == 56 * 8, "KTab layout");

__host__ __device__ constexpr KTab default_ktab()
{
    KTab k{};
    constexpr double c[10] = {0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22, 0x1.71dee623fde64p-19, 0x1.a01997c89e6b0p-16,
                              0x1.a01a014761f6ep-13, 0x1.6c16c1852b7b0p-10, 0x1.1111111122322p-7, 0x1.55555555502a1p-5,
                              0x1.5555555555511p-3, 0x1.000000000000bp-1};
    for (int i = 0; i < 10; ++i) { k.pc[i] = c[i]; k.tc[i] = c[i]; }
    k.log2_10 = 0x1.a934f0979a371p+1; k.lg2_hi = 0x1.34413509f79ffp-2; k.lg2_lo = -0x1.9dc1da994fd21p-59;
    k.ln10_hi = 0x1.26bb1bbb55516p+1; k.ln10_lo = -0x1.f48ad494ea3e9p-53;
    k.t_hi = 0x1.34413509f79ffp+8; k.t_lo = -0x1.439b746e36b52p+8;   // 10^x overflows above / is 0 below
    k.c2303 = 2.303; k.ln10 = rc::LN10; k.c002 = 0.02;
    k.log2e = 0x1.71547652b82fep+0; k.ln2_hi = 0x1.62e42fefa39efp-1; k.ln2_lo = 0x1.abc9e3b39803fp-56;
    k.e_hi = 0x1.62e42fefa39efp+9; k.e_lo = -0x1.74910d52d3051p+9;
    k.k_arr = rc::K_ARR; k.inv_tref = rc::INV_TREF; k.c27315 = 273.15; k.c1em4 = 0.0001;
    k.rho_max = 999.97; k.rho_an = -0.008; k.rho20 = 998.2; k.rho_sl = -2.1e-4 * 998.2; k.c20 = 20.0; k.c8 = 8.0; k.c100 = 100.0;
    return k;
}

// the constants a section works with, as plain doubles (SGPR pairs after the loads below)
struct KP { double c[10], log2_10, lg2_hi, lg2_lo, ln10_hi, ln10_lo, t_hi, t_lo, c2303, ln10, c002; };
struct KT { double c[10], log2e, ln2_hi, ln2_lo, e_hi, e_lo, k_arr, inv_tref, c27315, c1em4, rho_max, rho_an, rho20, rho_sl, c20, c8, c100, dense_bias; };

__host__ __device__ constexpr KP kp_of(const KTab &t)
{
    KP k{};
    for (int i = 0; i < 10; ++i) k.c[i] = t.pc[i];
    k.log2_10 = t.log2_10; k.lg2_hi = t.lg2_hi; k.lg2_lo = t.lg2_lo; k.ln10_hi = t.ln10_hi; k.ln10_lo = t.ln10_lo;
    k.t_hi = t.t_hi; k.t_lo = t.t_lo; k.c2303 = t.c2303; k.ln10 = t.ln10; k.c002 = t.c002;
    return k;
}
__host__ __device__ constexpr KT kt_of(const KTab &t)
{
    KT k{};
    for (int i = 0; i < 10; ++i) k.c[i] = t.tc[i];
    k.log2e = t.log2e; k.ln2_hi = t.ln2_hi; k.ln2_lo = t.ln2_lo; k.e_hi = t.e_hi; k.e_lo = t.e_lo;
    k.k_arr = t.k_arr; k.inv_tref = t.inv_tref; k.c27315 = t.c27315; k.c1em4 = t.c1em4;
    k.rho_max = t.rho_max; k.rho_an = t.rho_an; k.rho20 = t.rho20; k.rho_sl = t.rho_sl; k.c20 = t.c20; k.c8 = t.c8; k.c100 = t.c100; k.dense_bias = t.dense_bias;
    return k;
}

typedef const __attribute__((address_space(4))) d8 *KVec;
// three / four s_load_dwordx16 from the kernel-argument segment
__device__ __forceinline__ KP load_kp(const __attribute__((address_space(4))) KTab *t)
{
    KVec q = (KVec)t;
    const d8 a = q[0], b = q[1], c = q[2];
    KP k;
#pragma unroll
    for (int i = 0; i < 8; ++i) k.c[i] = a[i];
    k.c[8] = b[0]; k.c[9] = b[1];
    k.log2_10 = b[2]; k.lg2_hi = b[3]; k.lg2_lo = b[4]; k.ln10_hi = b[5]; k.ln10_lo = b[6]; k.t_hi = b[7];
    k.t_lo = c[0]; k.c2303 = c[1]; k.ln10 = c[2]; k.c002 = c[3];
    return k;
}
__device__ __forceinline__ KT load_kt(const __attribute__((address_space(4))) KTab *t)
{
    KVec q = (KVec)t + 3;
    const d8 a = q[0], b = q[1], c = q[2], d = q[3];
    KT k;
#pragma unroll
    for (int i = 0; i < 8; ++i) k.c[i] = a[i];
    k.c[8] = b[0]; k.c[9] = b[1];
    k.log2e = b[2]; k.ln2_hi = b[3]; k.ln2_lo = b[4]; k.e_hi = b[5]; k.e_lo = b[6]; k.k_arr = b[7];
    k.inv_tref = c[0]; k.c27315 = c[1]; k.c1em4 = c[2]; k.rho_max = c[3]; k.rho_an = c[4]; k.rho20 = c[5]; k.rho_sl = c[6]; k.c20 = c[7];
    k.c8 = d[0]; k.c100 = d[1]; k.dense_bias = d[2];
    return k;
}

// e^t for the reduced argument t, times 2^dn (the tail both exponentials share)
__device__ __forceinline__ double exp_tail(const double c[10], double t, double dn)
{
    double p = c[0];
#pragma unroll
    for (int i = 1; i < 10; ++i) p = __builtin_fma(t, p, c[i]);
    p = __builtin_fma(t, p, 1.0);
    p = __builtin_fma(t, p, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)dn);
}
__device__ __forceinline__ double exp_k(const KT &k, double x)
{
    const double dn = __builtin_rint(x * k.log2e);
    const double t = __builtin_fma(-dn, k.ln2_lo, __builtin_fma(-dn, k.ln2_hi, x));
    double z = exp_tail(k.c, t, dn);
    z = (x > k.e_hi) ? __builtin_inf() : z;
    return (x < k.e_lo) ? 0.0 : z;
}
__device__ __forceinline__ double exp10_k(const KP &k, double x)
{
    const double dn = __builtin_rint(x * k.log2_10);
    const double u = __builtin_fma(-dn, k.lg2_lo, __builtin_fma(-dn, k.lg2_hi, x));
    const double t = __builtin_fma(u, k.ln10_hi, u * k.ln10_lo);
    double z = exp_tail(k.c, t, dn);
    z = (x > k.t_hi) ? __builtin_inf() : z;
    return (x < k.t_lo) ? 0.0 : z;
}

// The Radau constants of the solver sections, fetched the same way (radau.py:11-40 values, see rc::)
struct alignas(64) RTab {
    double T00, T01, T02, T10, T11, T12, rtol, atol;                 // section Z: Z = T W, norm scales
    double TI[9], mu_r, mu_cr, mu_ci, newton_tol, pad_n[3];           // section N: one Newton iteration
    double E0, E1, E2, pad_e[5];                                      // section E: error estimate
    double P[9], pad_a[7];                                            // section A: dense output of an accepted step
    double C0, C1, pad_g[6];                                          // section G: initial guess of an attempt
};
static_assert(sizeof(RTab) == 56 * 8, "RTab layout");
__host__ __device__ constexpr RTab default_rtab()
{
    RTab r{};
    r.T00 = rc::T00; r.T01 = rc::T01; r.T02 = rc::T02; r.T10 = rc::T10; r.T11 = rc::T11; r.T12 = rc::T12; r.rtol = RTOL; r.atol = ATOL;
    constexpr double ti[9] = {rc::TI00, rc::TI01, rc::TI02, rc::TI10, rc::TI11, rc::TI12, rc::TI20, rc::TI21, rc::TI22};
    constexpr double pm[9] = {rc::P00, rc::P01, rc::P02, rc::P10, rc::P11, rc::P12, rc::P20, rc::P21, rc::P22};
    for (int i = 0; i < 9; ++i) { r.TI[i] = ti[i]; r.P[i] = pm[i]; }
    r.mu_r = rc::MU_REAL; r.mu_cr = rc::MU_CR; r.mu_ci = rc::MU_CI; r.newton_tol = rc::NEWTON_TOL;
    r.E0 = rc::E0; r.E1 = rc::E1; r.E2 = rc::E2;
    r.C0 = rc::C0; r.C1 = rc::C1;
    return r;
}
struct KZ { double T00, T01, T02, T10, T11, T12, rtol, atol; };
struct KN { double TI[9], mu_r, mu_cr, mu_ci, newton_tol; };
struct KE { double E0, E1, E2; };
struct KA { double P[9]; };
struct KG { double C0, C1; };
typedef const __attribute__((address_space(4))) RTab *RTabPtr;
__device__ __forceinline__ KZ load_kz(RTabPtr t) { const d8 a = ((KVec)t)[0]; return {a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]}; }
__device__ __forceinline__ KN load_kn(RTabPtr t)
{
    const d8 a = ((KVec)t)[1], b = ((KVec)t)[2];
    return {{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], b[0]}, b[1], b[2], b[3], b[4]};
}
__device__ __forceinline__ KE load_ke(RTabPtr t) { const d8 a = ((KVec)t)[3]; return {a[0], a[1], a[2]}; }
__device__ __forceinline__ KA load_ka(RTabPtr t)
{
    const d8 a = ((KVec)t)[4], b = ((KVec)t)[5];
    return {{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], b[0]}};
}
__device__ __forceinline__ KG load_kg(RTabPtr t) { const d8 a = ((KVec)t)[6]; return {a[0], a[1]}; }
// The error estimate and the accept block keep literal constants: with loaded ones the compiler stops sharing their
// common subexpressions and fuses the remaining ones differently (last-bit changes; measured, tools/bits_check.py).
constexpr RTab RT0 = default_rtab();
__device__ __forceinline__ KZ lit_kz() { return {RT0.T00, RT0.T01, RT0.T02, RT0.T10, RT0.T11, RT0.T12, RT0.rtol, RT0.atol}; }
__device__ __forceinline__ KE lit_ke() { return {RT0.E0, RT0.E1, RT0.E2}; }
__device__ __forceinline__ KA lit_ka() { return {{RT0.P[0], RT0.P[1], RT0.P[2], RT0.P[3], RT0.P[4], RT0.P[5], RT0.P[6], RT0.P[7], RT0.P[8]}}; }


struct StepArgs {
    int64_t N;        // reactors in the ensemble (row stride of par / bc)
    int64_t r0, r1;   // stream schedule: this launch advances reactors [r0, r1); queue schedule: [0, N)
    int n;            // zones per reactor
    int R;            // reactors per wavefront = 64 / n
    const double *par; // [WT_NP][N]
    double *bc;        // [WT_NB][N]  (rows 0, 4, 6 are rewritten by the command path when plant I/O is on)
    double *pH, *Cl, *T; // [N][n]
    double *time, *flow; // [N]
    double *dH, *dRho, *dK; // derived [N][n]
    uint32_t *status;    // [N]
    int32_t *stats;      // [N][5] or nullptr
    int64_t *wave_diag;  // [n_groups][WT_DIAG_SLOTS] or nullptr, accumulated over the work items of a launch:
                         //   trips, Newton trips, shader clocks, wall clock (100 MHz), factorize / num_jac / deferred-f block executions, items
    double *bad_T;       // [N] the temperature the reference's ValueError names (thermodynamics.py:151)
    // Placement: slot q of the wavefront-groups (group q / R, segment q % R) holds reactor perm[q].  Reactors never
    // interact, so which of them share a wavefront is free -- and a wavefront costs what its slowest reactor costs,
    // so reactors of similar solver cost are put together (wt_place.hpp re-bins them from `cost` between calls).
    const int32_t *perm;  // [N]
    int32_t *cost;        // [N] RHS evaluations since the last re-binning (the solver's nfev, summed over outer steps)
    double dt;
    int n_steps;      // outer steps this launch advances every reactor by
    int first_step;   // index of this launch's first step within the wt_ensemble_step call (PLC scan phase)
    int call_steps;   // outer steps of the whole wt_ensemble_step call
    int step_limit;   // give up an outer step after this many step attempts (0 = never, as the reference)
    // Work queue (nullptr: stream schedule -- workgroup b advances the wavefront-group r0 / R + b by n_steps).
    // q_ctrl: Q_AVAIL, Q_HEAD, Q_TAIL, Q_ERROR; q_slots[q_cap]: (ticket + 1) << 32 | group; q_next[group]: next step.
    int32_t *q_ctrl; unsigned long long *q_slots; int32_t *q_next;
    int q_cap, item_steps, n_groups;
    int64_t *trace; int trace_cap;   // optional item trace (tools/): worker, group, step0 | cnt << 32, start, end (100 MHz ticks)
    wts::SuiteArgs sens; // fused sensor suite + plant I/O (sens.on == 0: none)
    KTab kt;             // fp64 constants of the RHS sections (scalar loads)
    RTab rt;             // ... of the solver sections
};
enum { Q_AVAIL = 0, Q_HEAD = 1, Q_TAIL = 2, Q_ERROR = 3, Q_TRACE = 4, Q_DONE = 5, Q_WORDS = 16 };

// ---------------------------------------------------------------- lane geometry and cross-lane moves
typedef __attribute__((address_space(3))) double LdsDouble;
typedef __attribute__((address_space(3))) char LdsByte;
struct Divisor { double d, inv; };     // a fixed divisor and RN(1 / d): div_by()
struct Lane {
    int n, z;
    bool has_lo, has_hi;
    // the same neighbour tests as all-ones / zero words, one pair per cyclic-reduction stride 2^l: masking
    // with a VGPR operand (v_and) keeps the tests out of the scalar register file, where each would be a
    // 64-bit lane mask for the whole solver loop
    uint32_t m_lo[7], m_hi[7];
    uint32_t m_pt;            // whether the one partner of the top cyclic-reduction level (from_partner) exists
    int a_me;                 // byte offset of this lane's own cell in the exchange row (8 * lane)
    LdsDouble *xrow;          // cell 0 of the exchange row (ROW = false kernels; nullptr otherwise): XROW_PAD cells either side
    int base;                 // lane id of zone 0 of this segment
    unsigned long long segmask;
    Divisor d3n, d9n;         // 3n, 9n: component counts of the RMS norms (common.py:63-65, radau.py:105)
};

template <int CTRL> __device__ __forceinline__ double dpp_mov(double x)
{
    // bound_ctrl: lanes whose source lies outside the row / wavefront read 0; no "old" value to keep
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// ROW = true: n divides 16, so a segment never straddles a 16-lane DPP row and
// every power-of-two stride is a row shift.  ROW = false: any n <= 64; stride 1
// is a whole-wave DPP shift; larger strides go through the exchange row (both(), below) in the solver, and
// through ds_bpermute in these two single-direction forms (self-test only).
// Values read from outside the segment are unspecified; callers mask them.
__device__ __forceinline__ double bpermute(int byte_addr, double x)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(x));
    return __hiloint2double(hi, lo);
}
constexpr int ilog2(int s) { return s <= 1 ? 0 : 1 + ilog2(s >> 1); }

template <bool ROW, int S> __device__ __forceinline__ double from_lo(const Lane &L, double x)
{   // value held by lane (this - S)
    if constexpr (ROW && S < 16) return dpp_mov<0x110 + S>(x);      // row_shr:S
    else if constexpr (S == 1) return dpp_mov<0x138>(x);  // wave_shr:1
    else return bpermute(((L.a_me >> 1) - 4 * S) & 255, x);
}
// A value read from outside the segment only ever meets a zero coefficient, so it is
// enough to make it FINITE: clearing the high dword (sign, exponent, top mantissa bits)
// turns any NaN/Inf another reactor may hold into a denormal.  One v_and per read (with the
// Lane's all-ones / zero mask words), which the compiler folds into the DPP move of that dword.
__device__ __forceinline__ double keep_m(uint32_t mask, double x)
{
    return __hiloint2double(__double2hiint(x) & (int)mask, __double2loint(x));
}

// Strides that neither a row shift nor the whole-wave shift by one can do (ROW = false, S > 1) go through a row of 64
// doubles in LDS: every lane stores its value, then loads its neighbours'.  One ds_write_b64 + two ds_read_b64 per
// double and level where ds_bpermute_b32 needs four moves, at 10-14 cycles each instead of 24 (tools/ubench/lds.hip:
// four wavefronts of a CU sharing its LDS pipe).  The LDS pipe executes a wavefront's operations in order, so the next
// value may be stored as soon as the loads of the last one are issued: no wait between exchanges, only before the use.
// Lane +/- stride is an immediate offset from the lane's own cell (no address registers per level); what a lane reads
// from beyond the row's ends (its neighbours in the allocation: LdsMap) or from another reactor's cells is masked by its
// caller like every out-of-segment value.
constexpr int XROW_PAD = 32, XROW_CELLS = 64 + 2 * XROW_PAD;      // a stand-alone row (self-test kernel) carries padding
// (To the compiler a lane's store to its own cell and its loads of other cells are unrelated accesses of one thread,
// free to be reordered; the wave barriers -- no instruction, an ordering point for memory operations -- say otherwise.)
__device__ __forceinline__ void x_put(const Lane &L, double x)
{
    __builtin_amdgcn_wave_barrier();            // the loads of the previous exchange come first
    *(LdsDouble *)((LdsByte *)L.xrow + L.a_me) = x;
    __builtin_amdgcn_wave_barrier();            // ... and this exchange's loads after the store
}
__device__ __forceinline__ double x_get(const Lane &L, int byte_off) { return *(LdsDouble *)((LdsByte *)L.xrow + byte_off); }
template <int D> __device__ __forceinline__ double x_rel(const Lane &L)
{
    static_assert(D >= -XROW_PAD && D <= XROW_PAD, "stride beyond the padding");
    return *(LdsDouble *)((LdsByte *)L.xrow + L.a_me + 8 * D);
}

// At the top level of the cyclic reduction (stride S = 2^(LV-1) >= n/2) a zone has at most ONE partner: zone z - S if
// z >= S, else zone z + S if that exists.  ROW kernels (n = 2 S): partner = z xor S, a quad permutation (n = 2, 4), a
// row rotation (n = 16) or two bank-masked row shifts into one register (n = 8).  Otherwise through the exchange row.
template <int CTRL, int BANKS> __device__ __forceinline__ double dpp_merge(double old, double x)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xf, BANKS, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xf, BANKS, false);
    return __hiloint2double(hi, lo);
}
template <bool ROW, int S> __device__ __forceinline__ double from_partner(const Lane &L, double x)
{
    if constexpr (ROW && S == 1) return dpp_mov<0xB1>(x);            // quad_perm [1,0,3,2]
    else if constexpr (ROW && S == 2) return dpp_mov<0x4E>(x);       // quad_perm [2,3,0,1]
    else if constexpr (ROW && S == 4) return dpp_merge<0x104, 0x5>(dpp_merge<0x114, 0xA>(x, x), x);   // row_shr:4 -> zones 4..7, row_shl:4 -> zones 0..3
    else if constexpr (ROW && S == 8) return dpp_mov<0x128>(x);      // row_ror:8
    else { x_put(L, x); return keep_m(L.m_pt, x_get(L, L.a_me + ((L.z >= S) ? -8 * S : 8 * S))); }   // (S: the top stride)
}

template <bool ROW, int S> __device__ __forceinline__ double from_hi(const Lane &L, double x)
{   // value held by lane (this + S)
    if constexpr (ROW && S < 16) return dpp_mov<0x100 + S>(x);      // row_shl:S
    else if constexpr (S == 1) return dpp_mov<0x130>(x);  // wave_shl:1
    else return bpermute(((L.a_me >> 1) + 4 * S) & 255, x);
}

// the values held by lanes (this - S) and (this + S)
template <bool ROW, int S> __device__ __forceinline__ void both(const Lane &L, double x, double &lo, double &hi)
{
    if constexpr (ROW || S == 1) { lo = from_lo<ROW, S>(L, x); hi = from_hi<ROW, S>(L, x); }
    else { x_put(L, x); lo = x_rel<-S>(L); hi = x_rel<S>(L); }
}

__device__ __forceinline__ bool seg_any(const Lane &L, bool p) { return (__ballot(p) & L.segmask) != 0ull; }
__device__ __forceinline__ bool seg_all(const Lane &L, bool p) { return (__ballot(!p) & L.segmask) == 0ull; }

// Sum over the n lanes of a segment; every lane of the segment receives the
// bitwise-identical value (each butterfly step adds the same two operands in
// both partner lanes; the generic path scans and broadcasts).
// (LV: the kernel's number of cyclic-reduction levels, n <= 2^LV, where the caller knows it: rounds that cannot
// contribute are not compiled)
template <bool ROW, int LV = 6> __device__ __forceinline__ double seg_sum(const Lane &L, double x)
{
    if constexpr (ROW) {
        x += dpp_mov<0xB1>(x);                     // quad_perm [1,0,3,2]
        if (L.n >= 4) x += dpp_mov<0x4E>(x);       // quad_perm [2,3,0,1]
        if (L.n >= 8) x += dpp_mov<0x141>(x);      // row_half_mirror
        if (L.n >= 16) x += dpp_mov<0x140>(x);     // row_mirror
        return x;
    } else {
        // The inclusive scan by strides 1, 2, 4, ... (each lane adds the value 2^k lanes below while that lane is in
        // the segment), two strides per exchange: lane z forms what lane z - 2s would have added in the skipped
        // round itself, from the same operands in the same order -- the same bits as one stride per round, in half
        // the LDS round trips.  Then the last lane's total goes to everyone.
        auto two = [&](auto S_, double y) {
            constexpr int S = decltype(S_)::value;
            x_put(L, y);
            const double v1 = x_rel<-S>(L), v2 = x_rel<-2 * S>(L), v3 = x_rel<-3 * S>(L);
            const double t = (L.z >= 3 * S) ? v2 + v3 : v2;
            y = (L.z >= S) ? y + v1 : y;
            return (L.z >= 2 * S) ? y + t : y;
        };
        x = two(std::integral_constant<int, 1>{}, x);
        if (LV >= 3 && L.n > 4) x = two(std::integral_constant<int, 4>{}, x);
        if (LV >= 5 && L.n > 16) {
            x_put(L, x);
            const double v1 = x_rel<-16>(L);
            double t = 0.0;
            if (LV >= 6 && L.n > 32) {      // (the row's neighbours in the allocation do not reach 48 cells down: clamp)
                const double v2 = x_get(L, max(L.a_me - 8 * 32, 0)), v3 = x_get(L, max(L.a_me - 8 * 48, 0));
                t = (L.z >= 48) ? v2 + v3 : v2;
            }
            x = (L.z >= 16) ? x + v1 : x;
            x = (L.z >= 32) ? x + t : x;
        }
        x_put(L, x);
        return x_get(L, (L.base + L.n - 1) << 3);
    }
}

// 1/x to ~1 ulp: hardware seed + two Newton steps (no denormal / inf handling:
// every divisor on this path is a finite, normal number or the result is discarded)
__device__ __forceinline__ double rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

// x^(1/4) and x^(-1/4) for the step-size controller (radau.py:171-174, common.py:130)
// through two square roots instead of the general pow(); the result only steers h.
// sqrt: the library's iteration (reciprocal square root seed, two Goldschmidt steps, one correction: correctly rounded)
// without its rescaling of arguments below 2^-767, which the norms and step-size ratios formed here never are
// (0, +inf and NaN behave as in the library).  Half the instructions.
__device__ __forceinline__ double sqrt_k(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x); g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x); g = __builtin_fma(d, h, g);
    return __builtin_amdgcn_class(x, 0x260) ? x : g;       // +-0, +inf
}
__device__ __forceinline__ double root4(double x) { return sqrt_k(sqrt_k(x)); }

// x / d for a fixed divisor, correctly rounded like the division it replaces (Markstein: q = x * RN(1/d) is within
// an ulp, its residual is exact in one fma, one more fma rounds correctly): three instructions instead of fourteen.
__device__ __forceinline__ double div_by(double x, const Divisor &c)
{
    const double q = x * c.inv;
    const double r = __builtin_fma(-c.d, q, x);
    const double q2 = __builtin_fma(r, c.inv, q);
    return __builtin_amdgcn_class(q, 0x204) ? q : q2;      // +-inf stays inf (its residual is NaN)
}

// rate ** k for k = 1..6 (radau.py:113) by repeated multiplication
__device__ __forceinline__ double powi6(double x, int k)
{
    const double x2 = x * x, x3 = x2 * x;
    double r = x;
    r = (k == 2) ? x2 : r; r = (k == 3) ? x3 : r; r = (k == 4) ? x2 * x2 : r;
    r = (k == 5) ? x2 * x3 : r; r = (k == 6) ? x3 * x3 : r;
    return r;
}

// ---------------------------------------------------------------- reactor constants
struct RK {
    // chemistry.py:116-132 constants (frozen at configuration temperature)
    double Kw, Ka1, Ka1Ka2, KaH, cbeta;
    // transport / spatial: Richardson number (g drho dz) / (rho_avg u^2) against Ri_crit (spatial.py:262-277,293)
    double Kex, dz, u2, ricrit, rihulp, supp, unsupp;   // rihulp: half an ulp of Ri_crit
    // boundary-derived (reactor.py:336,349-368,385-395,426-443)
    double Qv, H_in, Cl_in, T_in, acid_dH, cl_dose, UAr, T_amb;
    double flowsum;   // inlet + acid + chlorine flow: ReactorState.flow_rate (reactor.py:497-501)
    bool has_acid, has_cl, has_heat;
    // the same, pre-masked for this lane's zone so the RHS needs no per-term selects:
    // inlet / dosing terms act on zone 0 only, the outlet sink on zone n-1 only
    double Kex_hi;    // Kex if this zone has an upper neighbour else 0
    double Qv_in;     // Qv in zone 0 else 0
    double Qv_out;    // Qv in zone n-1 else 0
    double acid0;     // acid dosing dH/dt in zone 0 (0 elsewhere / when off)
    double dose0;     // chlorine dosing in zone 0 (0 elsewhere / when off)
    double UAr_on;    // heat-loss coefficient (0 when off)
};

__device__ __forceinline__ double ulp_above_pos(double t) { return __longlong_as_double(__double_as_longlong(t) + 1) - t; }

// cmd != nullptr: boundary rows 0 / 4 / 6 (inlet, acid, chlorine flow) as the command path has just set them
__device__ __forceinline__ void load_reactor(const double *par, const double *bc, int64_t N, int64_t r, int n, RK &k,
                                             const double *cmd = nullptr, int cmd_stride = 0)
{
    auto P = [&](int row) { return par[(int64_t)row * N + r]; };
    auto B = [&](int row) {
        if (cmd && row == 0) return cmd[0 * cmd_stride];
        if (cmd && row == 4) return cmd[1 * cmd_stride];
        if (cmd && row == 6) return cmd[2 * cmd_stride];
        return bc[(int64_t)row * N + r];
    };
    const double V = P(0), height = P(1), diam = P(2);
    k.Kw = P(3); k.Ka1 = P(4); k.Ka1Ka2 = P(4) * P(5); k.KaH = P(6);
    k.cbeta = 2.303 * P(7);                    // chemistry.py:431-433
    k.Kex = P(8);
    const double u = P(9);
    k.dz = height / n;                          // spatial.py:119
    k.u2 = u * u;                               // velocity_scale ** 2
    // the three cases of the stratification switch folded into the two outcomes of one branch-free test
    // (rhs_rows): stratification off (reactor.py:310-315) -> factor 1 either way; velocity scale <= 1e-6 -> Ri = +inf,
    // stable either way (spatial.py:270-275); else Ri against Ri_crit picks between the two
    const int strat_mode = (P(10) != 0.0) ? ((u > 1e-6) ? 1 : 2) : 0;
    k.ricrit = P(11);
    k.rihulp = 0.5 * ulp_above_pos(k.ricrit);
    k.supp = (strat_mode != 0) ? P(12) : 1.0;        // interface factor where the Richardson test says "stable"
    k.unsupp = (strat_mode == 2) ? P(12) : 1.0;      // ... and where it does not
    const double Q_in = B(0);
    k.Qv = (Q_in / 60.0) / V;                   // reactor.py:336
    k.H_in = exp10_k(kp_of(default_ktab()), -B(1));   // reactor.py:363
    k.Cl_in = B(2); k.T_in = B(3);
    const double zone_volume_L = V / n;
    k.has_acid = B(4) > 0;
    k.acid_dH = ((B(4) / 60.0) * B(5)) / zone_volume_L; // reactor.py:350-354
    k.has_cl = B(6) > 0;
    k.cl_dose = ((B(6) / 60.0) * B(7)) / zone_volume_L; // reactor.py:388-392
    k.has_heat = B(9) > 0;
    const double PI = 3.141592653589793;
    const double rr = diam / 2;
    const double A_tot = PI * diam * height + 2 * PI * (rr * rr);  // reactor.py:429-431
    k.UAr = (B(9) * A_tot) / (998.2 * 4184 * (V / 1000));          // reactor.py:433-443
    k.T_amb = B(8);
    k.flowsum = B(0) + B(4) + B(6);
}

__device__ __forceinline__ void mask_reactor_for_lane(const Lane &L, RK &k)
{
    k.Kex_hi = L.has_hi ? k.Kex : 0.0;
    k.Qv_in = L.has_lo ? 0.0 : k.Qv;
    k.Qv_out = L.has_hi ? 0.0 : k.Qv;
    k.acid0 = (!L.has_lo && k.has_acid) ? k.acid_dH : 0.0;
    k.dose0 = (!L.has_lo && k.has_cl) ? k.cl_dose : 0.0;
    k.UAr_on = k.has_heat ? k.UAr : 0.0;
}

// The constants are needed by the RHS evaluations only.  Between them (factorisation, Newton solve, error
// estimate) they would occupy 38 VGPRs of a register file that is already oversubscribed, so they are parked
// in LDS and fetched at the top of every RHS block: 14 per-reactor words (one copy per reactor, broadcast to
// its lanes) and 5 per-lane ones (the inlet / outlet / neighbour masks applied once, at parking time).  Where LDS is
// short (n > 8: every slot is wanted for tridiagonal factors) the per-lane words are not parked but re-masked from
// four more per-reactor words at every fetch: ten v_and instead of three LDS reads.
constexpr int RK_UNI = 20, RK_LANE = 5, RK_MAXR = 32;   // up to 32 reactors per wavefront (n = 2)
// reactors per wavefront a kernel instantiation can meet: LV levels serve n in (2^(LV-1), 2^LV]
constexpr int rk_maxr(int LV) { return LV <= 1 ? 32 : 64 / ((1 << (LV - 1)) + 1); }
constexpr bool rk_lane_in_lds(int LV) { return LV < 4; }
constexpr int rk_lane_doubles(int LV) { return rk_lane_in_lds(LV) ? RK_LANE * 64 : 0; }
// uni[c * stride], lane[c * 64]: already offset for this lane; lane == nullptr: no per-lane words, use the masks
struct RKStore { double *uni; double *lane; int stride; uint32_t m_has_lo, m_has_hi; bool lane_lds; };   // lane_lds: a compile-time constant of the kernel
__device__ __forceinline__ double mask64(uint32_t m, double x) { return __hiloint2double(__double2hiint(x) & (int)m, __double2loint(x) & (int)m); }

__device__ __forceinline__ void park_reactor(const RKStore &st, const RK &k)
{
    const double u[RK_UNI] = {k.Kw, k.Ka1, k.Ka1Ka2, k.KaH, k.cbeta, k.dz, k.u2, k.supp, k.H_in, k.Cl_in, k.T_in, k.T_amb, k.UAr_on,
                              k.unsupp, k.ricrit, k.flowsum,
                              k.Kex, k.Qv, k.has_acid ? k.acid_dH : 0.0, k.has_cl ? k.cl_dose : 0.0};   // (rihulp is re-derived from ricrit)
    const double l[RK_LANE] = {k.Kex_hi, k.Qv_in, k.Qv_out, k.acid0, k.dose0};
#pragma unroll
    for (int c = 0; c < RK_UNI; ++c) st.uni[c * st.stride] = u[c];   // every lane of the reactor stores the same value
    if (st.lane_lds) {
#pragma unroll
        for (int c = 0; c < RK_LANE; ++c) st.lane[c * 64] = l[c];
    }
}

__device__ __forceinline__ RK fetch_reactor(const RKStore &st)
{
    RK k;
    k.Kw = st.uni[0 * st.stride]; k.Ka1 = st.uni[1 * st.stride]; k.Ka1Ka2 = st.uni[2 * st.stride]; k.KaH = st.uni[3 * st.stride];
    k.cbeta = st.uni[4 * st.stride]; k.dz = st.uni[5 * st.stride]; k.u2 = st.uni[6 * st.stride]; k.supp = st.uni[7 * st.stride];
    k.H_in = st.uni[8 * st.stride]; k.Cl_in = st.uni[9 * st.stride]; k.T_in = st.uni[10 * st.stride]; k.T_amb = st.uni[11 * st.stride];
    k.UAr_on = st.uni[12 * st.stride]; k.unsupp = st.uni[13 * st.stride]; k.ricrit = st.uni[14 * st.stride]; k.rihulp = 0.5 * ulp_above_pos(k.ricrit);
    if (st.lane_lds) {
        k.Kex_hi = st.lane[0 * 64]; k.Qv_in = st.lane[1 * 64]; k.Qv_out = st.lane[2 * 64]; k.acid0 = st.lane[3 * 64]; k.dose0 = st.lane[4 * 64];
    } else {   // the same values as mask_reactor_for_lane() made: zero where the zone has no such term
        const double Kex = st.uni[16 * st.stride], Qv = st.uni[17 * st.stride];
        k.Kex_hi = mask64(st.m_has_hi, Kex);
        k.Qv_in = mask64(~st.m_has_lo, Qv); k.Qv_out = mask64(~st.m_has_hi, Qv);
        k.acid0 = mask64(~st.m_has_lo, st.uni[18 * st.stride]); k.dose0 = mask64(~st.m_has_lo, st.uni[19 * st.stride]);
    }
    return k;
}

// ---------------------------------------------------------------- zone-local properties
struct PropPH { double H, iw, phi; bool bpos; }; // iw = 1/(beta*ln10)
struct PropT { double kT, rho; bool bad; };

// The RHS is inlined at several places of the solver (stage points, single points, the deferred f(y_new),
// the finite-difference passes) and the same state must give the same bits at each of them -- f(y0) of an outer
// step may come from any of them depending on the schedule.  So nothing here is left to the compiler's choice of
// which products to fuse: contraction is off and every fused multiply-add is spelled out.

// H = 10^-pH, buffering capacity beta (chemistry.py:400-437), HOCl/OCl- decay
// factor (chemistry.py:483-523).

// ---- the same arithmetic for NS points at once, step by step: consecutive instructions belong to different points and
// are independent (a dependent fp64 instruction issues after 8 cycles, an independent one after 5: tools/ubench/issue.hip)
#define WT_EACH _Pragma("unroll") for (int s = 0; s < NS; ++s)
// The compiler, short of registers, pulls each point's chain together again; an empty asm that "uses and redefines"
// the step's results makes every step complete for all points before the next one starts (NS = 1: nothing to do).
template <int NS> __device__ __forceinline__ void row_fence(double (&v)[NS])
{
    // (one point: nothing to interleave, but the fence keeps the compiler from merging this section with the next)
    if constexpr (NS == 1) asm volatile("" : "+v"(v[0]));
    if constexpr (NS == 2) asm volatile("" : "+v"(v[0]), "+v"(v[1]));
    if constexpr (NS == 3) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
    if constexpr (NS == 4) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}
template <int NS> __device__ __forceinline__ void rcp_n(const double (&x)[NS], double (&r)[NS])
{
    double e[NS];
    WT_EACH r[s] = __builtin_amdgcn_rcp(x[s]); row_fence(r);
    WT_EACH e[s] = __builtin_fma(-x[s], r[s], 1.0); row_fence(e);
    WT_EACH r[s] = __builtin_fma(r[s], e[s], r[s]); row_fence(r);
    WT_EACH e[s] = __builtin_fma(-x[s], r[s], 1.0); row_fence(e);
    WT_EACH r[s] = __builtin_fma(r[s], e[s], r[s]); row_fence(r);
}
template <int NS> __device__ __forceinline__ void exp_tail_n(const double c[10], const double (&t)[NS], const double (&dn)[NS], double (&z)[NS])
{
    double p[NS];
    WT_EACH p[s] = c[0];
#pragma unroll
    for (int i = 1; i < 10; ++i) { WT_EACH p[s] = __builtin_fma(t[s], p[s], c[i]); row_fence(p); }
    WT_EACH p[s] = __builtin_fma(t[s], p[s], 1.0); row_fence(p);
    WT_EACH p[s] = __builtin_fma(t[s], p[s], 1.0); row_fence(p);
    WT_EACH z[s] = __builtin_amdgcn_ldexp(p[s], (int)dn[s]); row_fence(z);
}
template <int NS> __device__ __forceinline__ void prop_pH_n(const KP &c, const RK &k, const double (&pH)[NS], PropPH (&p)[NS])
{
#pragma clang fp contract(off)
    double x[NS], dn[NS], u[NS], t[NS], H[NS];
    WT_EACH x[s] = -pH[s];
    WT_EACH dn[s] = __builtin_rint(x[s] * c.log2_10); row_fence(dn);
    WT_EACH u[s] = __builtin_fma(-dn[s], c.lg2_hi, x[s]); row_fence(u);
    WT_EACH u[s] = __builtin_fma(-dn[s], c.lg2_lo, u[s]); row_fence(u);
    WT_EACH t[s] = u[s] * c.ln10_lo; row_fence(t);
    WT_EACH t[s] = __builtin_fma(u[s], c.ln10_hi, t[s]); row_fence(t);
    exp_tail_n<NS>(c.c, t, dn, H);
    WT_EACH H[s] = (x[s] > c.t_hi) ? __builtin_inf() : H[s];
    WT_EACH H[s] = (x[s] < c.t_lo) ? 0.0 : H[s];
    double iH[NS], H2[NS], D[NS], iD[NS], HK[NS], iHK[NS];
    WT_EACH H2[s] = H[s] * H[s];
    WT_EACH D[s] = __builtin_fma(k.Ka1, H[s], H2[s]) + k.Ka1Ka2;
    WT_EACH HK[s] = H[s] + k.KaH;
    rcp_n<NS>(H, iH); rcp_n<NS>(D, iD); rcp_n<NS>(HK, iHK);
    double bw[NS], a0[NS], a1[NS], a2[NS], mix[NS], beta[NS], bl[NS], ib[NS];
    WT_EACH bw[s] = c.c2303 * __builtin_fma(k.Kw, iH[s], H[s]);
    WT_EACH a0[s] = H2[s] * iD[s];
    WT_EACH a1[s] = (k.Ka1 * H[s]) * iD[s];
    WT_EACH a2[s] = k.Ka1Ka2 * iD[s];
    WT_EACH mix[s] = __builtin_fma(a0[s], a2[s], __builtin_fma(4 * a1[s], a2[s], a0[s] * a1[s]));
    WT_EACH beta[s] = __builtin_fma(k.cbeta, mix[s], bw[s]);
    WT_EACH bl[s] = beta[s] * c.ln10;
    rcp_n<NS>(bl, ib);
    WT_EACH {
        p[s].bpos = beta[s] > 0;                    // reactor.py:358,367,375 guards: no pH change unless beta > 0
        p[s].iw = p[s].bpos ? ib[s] : 0.0;
        p[s].phi = __builtin_fma(k.KaH * iHK[s], c.c002, H[s] * iHK[s]);
        p[s].H = H[s];
    }
}
template <int NS> __device__ __forceinline__ void prop_T_n(const KT &c, const double (&T)[NS], PropT (&p)[NS])
{
#pragma clang fp contract(off)
    double tk[NS], itk[NS], ex[NS], dn[NS], t[NS], z[NS];
    WT_EACH tk[s] = T[s] + c.c27315;
    rcp_n<NS>(tk, itk);
    WT_EACH ex[s] = c.k_arr * (itk[s] - c.inv_tref);
    WT_EACH dn[s] = __builtin_rint(ex[s] * c.log2e); row_fence(dn);
    WT_EACH t[s] = __builtin_fma(-dn[s], c.ln2_hi, ex[s]); row_fence(t);
    WT_EACH t[s] = __builtin_fma(-dn[s], c.ln2_lo, t[s]); row_fence(t);
    exp_tail_n<NS>(c.c, t, dn, z);
    WT_EACH z[s] = (ex[s] > c.e_hi) ? __builtin_inf() : z[s];
    WT_EACH z[s] = (ex[s] < c.e_lo) ? 0.0 : z[s];
    WT_EACH {
        p[s].bad = (T[s] < 0.0) || (T[s] > c.c100);
        p[s].kT = c.c1em4 * z[s];
        const double d = T[s] - 4.0;
        const double cold = c.rho_max + (c.rho_an * (d * d));
        const double warm = c.rho20 + (c.rho_sl * (T[s] - c.c20));
        p[s].rho = (T[s] <= c.c8) ? cold : warm;
    }
}

// H = 10^-pH, buffering capacity beta (chemistry.py:400-437), HOCl/OCl- decay factor (chemistry.py:483-523).
__device__ __forceinline__ PropPH prop_pH(const KP &c, const RK &k, double pH)
{
    const double x[1] = {pH}; PropPH p[1];
    prop_pH_n<1>(c, k, x, p);
    return p[0];
}

// Arrhenius decay rate (thermodynamics.py:160-193) with its [0,100] C check
// (:146-157) and water density (spatial.py:177-189).  The density feeds the stratification switch, so it
// is formed with the reference's roundings: products and sums separately, never fused.
__device__ __forceinline__ PropT prop_T(const KT &c, double T)
{
    const double x[1] = {T}; PropT p[1];
    prop_T_n<1>(c, x, p);
    return p[0];
}

// One row-triple (dpH, dCl, dT) of derivatives() for this lane's zone, given the
// lane's own (possibly perturbed / stage) values; neighbour values come from the
// adjacent lanes' arguments to the same call.  reactor.py:304-443.
// Cross-lane moves are executed by every lane of the segment, then masked.
// The pieces of a row, shared by rhs_rows and the finite-difference passes (one rounding behaviour per expression).
// Interface factor above this zone: K[i,i+1] = Kex * suppression(rho_i, rho_{i+1})  (spatial.py:239-320, reactor.py:321-325).
// The reference compares the correctly rounded quotient Ri = num / den, num = (g drho) dz, den = rho_avg u^2 > 0, with
// Ri_crit.  fl(num / den) > c  <=>  num / den > c + ulp(c)/2  <=>  num - c den > (ulp(c)/2) den, and the left side is
// exact in one fma whenever the two sides are close enough for rounding to matter: the same decision as the
// reference's on the same bits, without a division.  Stratification off / velocity scale <= 1e-6 (Ri = +inf) are
// folded into the two outcomes (load_reactor).
__device__ __forceinline__ double k_above(const RK &k, double rho, double rho_hi)
{
#pragma clang fp contract(off)
    const double drho = rho_hi - rho;
    const double ravg = 0.5 * (rho + rho_hi);
    const double num = (9.81 * drho) * k.dz, den = ravg * k.u2;
    const double s = (__builtin_fma(-k.ricrit, den, num) > k.rihulp * den) ? k.supp : k.unsupp;
    return k.Kex_hi * s;                                   // 0 above the top zone
}
// K @ x the way OpenBLAS' dgemv accumulates it inside the reference: neighbours first, diagonal last, every product
// rounded before it is added (pinned by tests/golden/g2_rhs_*.npz: temperature rows bit-identical).
__device__ __forceinline__ double k_diag(const RK &k, double k_lo, double k_hi)
{
#pragma clang fp contract(off)
    return -(k_lo + k_hi) - k.Qv_out;                      // reactor.py:329-337
}
__device__ __forceinline__ double mix3(double k_lo, double k_hi, double kd, double x_lo, double x_hi, double x)
{
#pragma clang fp contract(off)
    return (k_lo * x_lo + k_hi * x_hi) + kd * x;
}
// zone-0 dosing and inlet (reactor.py:349-368,388-395,420) through pre-masked coefficients; iw is 0 when the
// reference's `beta > 0` guard fails
__device__ __forceinline__ double row_pH(const RK &k, double mixH, double H, double iw)
{
#pragma clang fp contract(off)
    return -(__builtin_fma(k.Qv_in, k.H_in - H, k.acid0) + mixH) * iw;                        // reactor.py:349-376
}
__device__ __forceinline__ double row_Cl(const RK &k, double mixC, double Cl, double kphi)
{
#pragma clang fp contract(off)
    return __builtin_fma(-kphi, Cl, __builtin_fma(k.Qv_in, k.Cl_in - Cl, k.dose0) + mixC);    // reactor.py:385-411
}
__device__ __forceinline__ double row_T(const RK &k, double mixT, double T)
{
#pragma clang fp contract(off)
    return __builtin_fma(-k.UAr_on, T - k.T_amb, k.Qv_in * (k.T_in - T) + mixT);               // reactor.py:420-443
}

// One row-triple (dpH, dCl, dT) of derivatives() for this lane's zone, given the
// lane's own (possibly stage) values; neighbour values come from the
// adjacent lanes' arguments to the same call.  reactor.py:304-443.
// Cross-lane moves are executed by every lane of the segment, then masked.
template <bool ROW>
__device__ __forceinline__ void rhs_rows(const Lane &L, const RK &k, double H, double iw, bool bpos, double kphi,
                                         double rho, double Cl, double T, double f[3])
{
    const double k_hi = k_above(k, rho, from_hi<ROW, 1>(L, rho));       // K[i,i+1]
    const double k_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, k_hi));    // K[i,i-1] (bound_ctrl gives 0 below zone 0 of lane 0)
    const double kd = k_diag(k, k_lo, k_hi);
    const double H_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, H)), H_hi = keep_m(L.m_hi[0], from_hi<ROW, 1>(L, H));
    const double C_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, Cl)), C_hi = keep_m(L.m_hi[0], from_hi<ROW, 1>(L, Cl));
    const double T_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, T)), T_hi = keep_m(L.m_hi[0], from_hi<ROW, 1>(L, T));
    // k_lo / k_hi are exactly 0 where there is no neighbour, and what was read there is finite (keep_m)
    (void)bpos;
    f[SPH] = row_pH(k, mix3(k_lo, k_hi, kd, H_lo, H_hi, H), H, iw);
    f[SCL] = row_Cl(k, mix3(k_lo, k_hi, kd, C_lo, C_hi, Cl), Cl, kphi);
    f[STT] = row_T(k, mix3(k_lo, k_hi, kd, T_lo, T_hi, T), T);
}

template <bool ROW>
__device__ __forceinline__ bool rhs_full(const Lane &L, const KP &cp, const KT &ct, const RK &k, const double y[3], double f[3])
{
    const PropPH pp = prop_pH(cp, k, y[SPH]);
    const PropT pt = prop_T(ct, y[STT]);
    rhs_rows<ROW>(L, k, pp.H, pp.iw, pp.bpos, pt.kT * pp.phi, pt.rho, y[SCL], y[STT], f);
    return pt.bad;
}

// ---------------------------------------------------------------- Jacobian bands
// Non-zero structure of d(dpH,dCl,dT)_i / d(pH,Cl,T)_j, j in {i-1,i,i+1}
// (index rel+1).  dT rows see only T; dpH rows see pH and (through the
// stratification switch only) T; dCl rows see Cl, own-zone pH and T.
struct Jac {
    double pp[3], cc[3], tt[3], pt[3], ct[3], cp;
};
// Loop-carried state that is written rarely and read rarely, held in ACCUMULATION registers by name: the register
// file of a lone wavefront has 512 registers of which the VALU addresses 256, and what the allocator keeps beyond
// them it shuffles through `v_accvgpr` copies at the head of the solver loop -- on EVERY trip, whether the value is
// touched or not.  A value that lives in an AGPR by constraint costs its copies where it is written and where it is
// read, nothing in between.  (Writes are VALU instructions: under a lane mask they update the active lanes only.)
// (A = false: an ordinary variable -- the n > 16 kernels, at 480-512 registers, answer 74 pinned AGPRs with scratch;
// the n = 17...32 kernel takes the 42 of everything but the Jacobian)
template <bool A> struct Held;
template <> struct Held<false> {
    double v;
    __device__ __forceinline__ void init() { v = 0.0; }
    __device__ __forceinline__ void set(double x) { v = x; }
    __device__ __forceinline__ double get() const { return v; }
};
template <> struct Held<true> {
    int lo, hi;
    __device__ __forceinline__ void init() { asm volatile("" : "=a"(lo), "=a"(hi)); }     // (defined, value irrelevant)
    __device__ __forceinline__ void set(double x)
    {
        asm volatile("v_accvgpr_write_b32 %0, %2\n\tv_accvgpr_write_b32 %1, %3" : "+a"(lo), "+a"(hi) : "v"(__double2loint(x)), "v"(__double2hiint(x)));
    }
    __device__ __forceinline__ double get() const
    {
        int l, h;
        asm("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=v"(l), "=v"(h) : "a"(lo), "a"(hi));
        return __hiloint2double(h, l);
    }
};
template <bool A> struct JacA {
    Held<A> e[16];     // tt[3], pp[3], cc[3] | pt[3], ct[3], cp
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i].init();
    }
    __device__ __forceinline__ void put(const Jac &J)
    {
#pragma unroll
        for (int r = 0; r < 3; ++r) { e[r].set(J.tt[r]); e[3 + r].set(J.pp[r]); e[6 + r].set(J.cc[r]); e[9 + r].set(J.pt[r]); e[12 + r].set(J.ct[r]); }
        e[15].set(J.cp);
    }
    __device__ __forceinline__ void bands(Jac &J) const
    {
#pragma unroll
        for (int r = 0; r < 3; ++r) { J.tt[r] = e[r].get(); J.pp[r] = e[3 + r].get(); J.cc[r] = e[6 + r].get(); }
    }
    __device__ __forceinline__ void coupling(Jac &J) const
    {
#pragma unroll
        for (int r = 0; r < 3; ++r) { J.pt[r] = e[9 + r].get(); J.ct[r] = e[12 + r].get(); }
        J.cp = e[15].get();
    }
};

// The inter-zone exchange depends on temperature through the stratification switch only, a step function: unless a
// finite-difference perturbation happens to flip a switch, the pH rows do not see T at all and the Cl rows see their
// own zone's T only (the Arrhenius rate) -- exact zeros, column by column.  A wavefront whose lanes all find them zero
// skips the neighbours' T increments in every solve (12 cross-lane moves and 15 fused multiply-adds of a Newton
// iteration): adding the exact zeros would not change a bit.
__device__ __forceinline__ bool jac_t_dense(const Jac &J)
{
    return (J.pt[0] != 0.0) || (J.pt[1] != 0.0) || (J.pt[2] != 0.0) || (J.ct[0] != 0.0) || (J.ct[2] != 0.0);
}

// PCR-factored tridiagonal systems (real and complex shift) live in LDS, not in registers: pair-major
// [pair of slots][64 lanes] 16-byte cells (FStore below), conflict-free ds_read_b128 / ds_write_b128.
// A factor is written once per (h, J) and read once per solve.
// For n > 16 (LV >= 5) the store outgrows what a wavefront may have of the CU's 160 KiB at four wavefronts per CU:
// the first NREG slots (real-shift factors) stay in registers instead -- exactly as many as do not fit.  All of them
// in LDS means three wavefronts per CU and 0.55-0.6x the throughput (measured at n = 20 and n = 40); all real-shift
// factors in registers costs scratch spills (n <= 32) or more of them (n > 32: 292 B against 176 B, -1 %).
constexpr int fstore_total_slots(int LV) { return 3 * (2 * LV) + 3 * (4 * LV); }
constexpr int fstore_lds_slots(int LV)
{
    const int budget = 40960;                                  // bytes per wavefront at four per CU
    const int fixed = ((RK_UNI * rk_maxr(LV) + rk_lane_doubles(LV) + rk_maxr(LV) + 64 + 1) & ~1) * 8;   // LdsMap: reactor constants, history base, reactor indices, exchange row (F_OFF)
    const int fit = ((budget - fixed) / 512) & ~1;                 // slots are stored as pairs
    return fit < fstore_total_slots(LV) ? fit : fstore_total_slots(LV);
}
typedef double __attribute__((ext_vector_type(2))) double2v;
typedef __attribute__((address_space(3))) double2v LdsDouble2;
template <int LV> struct FStore {
    static constexpr int NREG = fstore_total_slots(LV) - fstore_lds_slots(LV);   // slots [0, NREG) in registers
    static_assert(NREG % 2 == 0, "slots come in pairs");
    double reg[NREG > 0 ? NREG : 1];
    // Slots live in LDS as PAIRS (2j, 2j + 1) -- (alpha, gamma) of a level, (re, im) of a complex factor, (top factor,
    // 1/d) -- one 16-byte cell per pair and lane, pair-major: cell[(pair) * 64 + lane].  A pair is written and read
    // together with one ds_write_b128 / ds_read_b128: the read costs half of a two-address ds_read2st64_b64
    // (16 vs 32 cycles per wavefront with the CU's four wavefronts on its LDS pipe: tools/ubench/lds.hip).
    LdsDouble2 *cell;   // this lane's column of pairs: cell[pair * 64]
    // `slot` is a compile-time constant at every call site after inlining / unrolling
    __device__ __forceinline__ double ld(int slot) const
    {
        if (slot < NREG) return reg[slot < NREG ? slot : 0];
        const LdsDouble *p = (const LdsDouble *)(cell + ((slot - NREG) >> 1) * 64);
        return p[(slot - NREG) & 1];
    }
    __device__ __forceinline__ void ld2(int slot, double &a, double &b) const      // slot even
    {
        if (slot < NREG) { a = reg[slot < NREG ? slot : 0]; b = reg[slot + 1 < NREG ? slot + 1 : 0]; return; }
        const double2v v = cell[((slot - NREG) >> 1) * 64];
        a = v.x; b = v.y;
    }
    __device__ __forceinline__ void st2(int slot, double a, double b)             // slot even
    {
        if (slot < NREG) { reg[slot < NREG ? slot : 0] = a; reg[slot + 1 < NREG ? slot + 1 : 0] = b; return; }
        double2v v; v.x = a; v.y = b;
        cell[((slot - NREG) >> 1) * 64] = v;
    }
};
// slot map: real system k (0..2): [k RS + 2l] = alpha_l, [+2l+1] = gamma_l for the levels l < LV-1 below the top one,
//           [+2(LV-1)] = the top level's one factor (alpha for zones >= 2^(LV-1), gamma below: never both), [+2LV-1] = 1/d
//           complex system k: CB + k CS + 4l + {0,1,2,3} = al.r, al.i, ga.r, ga.i; [+4(LV-1), +1] = top factor, [+2, +3] = 1/d
template <int LV> struct FSlots {
    static constexpr int RS = 2 * LV, CS = 4 * LV, CB = 3 * RS, TOTAL = 3 * RS + 3 * CS;
    static constexpr int LDS_SLOTS = TOTAL - FStore<LV>::NREG;
};

struct cplx { double r, i; };
// (which product is fused is spelled out: the same solve is inlined in alternative paths -- the systems one by one or
// two in lock step -- and a reactor must get the same bits whichever its wavefront takes)
__device__ __forceinline__ cplx cmul(cplx a, cplx b)
{
#pragma clang fp contract(off)
    return {__builtin_fma(a.r, b.r, -(a.i * b.i)), __builtin_fma(a.r, b.i, a.i * b.r)};
}
__device__ __forceinline__ cplx cinv(cplx a)
{
#pragma clang fp contract(off)
    const double q = rcp(__builtin_fma(a.r, a.r, a.i * a.i));
    return {a.r * q, -a.i * q};
}

template <bool ROW, int S> __device__ __forceinline__ cplx cfrom_lo(const Lane &L, cplx a) { return {from_lo<ROW, S>(L, a.r), from_lo<ROW, S>(L, a.i)}; }
template <bool ROW, int S> __device__ __forceinline__ cplx cfrom_hi(const Lane &L, cplx a) { return {from_hi<ROW, S>(L, a.r), from_hi<ROW, S>(L, a.i)}; }
template <bool ROW, int S> __device__ __forceinline__ void cboth(const Lane &L, cplx a, cplx &lo, cplx &hi)
{
    both<ROW, S>(L, a.r, lo.r, hi.r); both<ROW, S>(L, a.i, lo.i, hi.i);
}

// One cyclic-reduction level of all six systems (three real, three complex shift) at once: the
// six eliminations are independent, so issuing them together hides the reciprocal / DPP latency
// of each behind the others.
template <bool ROW, int LV, int l>
__device__ __forceinline__ void pcr_factor_level_all(const Lane &L, double ar[3], double dr[3], double cr[3],
                                                     cplx ac[3], cplx dc[3], cplx cc[3], FStore<LV> &F)
{
    using S = FSlots<LV>;
    constexpr int s = 1 << l;
    if constexpr (l == 0) {
        // Level 0: the off-diagonals of the complex-shift systems are still the real ones (-J's bands, imaginary part
        // exactly 0), so their neighbours' values are the real systems' (moved once, not three times) and every
        // product with a zero imaginary part drops out -- the same bits with 16 cross-lane moves and 14 fp64
        // instructions less per system.
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double a0 = ar[k], c0 = cr[k];
            const double id = rcp(dr[k]);
            const double id_lo = from_lo<ROW, s>(L, id), id_hi = from_hi<ROW, s>(L, id);
            const double a_lo = keep_m(L.m_lo[l], from_lo<ROW, s>(L, a0)), c_lo = keep_m(L.m_lo[l], from_lo<ROW, s>(L, c0));
            const double a_hi = keep_m(L.m_hi[l], from_hi<ROW, s>(L, a0)), c_hi = keep_m(L.m_hi[l], from_hi<ROW, s>(L, c0));
            const double al = a0 * keep_m(L.m_lo[l], id_lo);
            const double ga = c0 * keep_m(L.m_hi[l], id_hi);
            dr[k] = dr[k] - al * c_lo - ga * a_hi;
            ar[k] = -al * a_lo;
            cr[k] = -ga * c_hi;
            F.st2(k * S::RS + 2 * l, al, ga);
            // complex shift: alpha = a / d_lo, gamma = c / d_hi with real a, c
            const cplx cid = cinv(dc[k]);
            const cplx i_lo = cfrom_lo<ROW, s>(L, cid), i_hi = cfrom_hi<ROW, s>(L, cid);
            const cplx il = {keep_m(L.m_lo[l], i_lo.r), keep_m(L.m_lo[l], i_lo.i)};
            const cplx ih = {keep_m(L.m_hi[l], i_hi.r), keep_m(L.m_hi[l], i_hi.i)};
            const cplx cal = {a0 * il.r, a0 * il.i};
            const cplx cga = {c0 * ih.r, c0 * ih.i};
            double dre = dc[k].r, dim = dc[k].i;
            dre = __builtin_fma(-cal.r, c_lo, dre); dim = __builtin_fma(-cal.i, c_lo, dim);
            dre = __builtin_fma(-cga.r, a_hi, dre); dim = __builtin_fma(-cga.i, a_hi, dim);
            dc[k] = {dre, dim};
            ac[k] = {-(cal.r * a_lo), -(cal.i * a_lo)};
            cc[k] = {-(cga.r * c_hi), -(cga.i * c_hi)};
            const int c0s = S::CB + k * S::CS + 4 * l;
            F.st2(c0s, cal.r, cal.i); F.st2(c0s + 2, cga.r, cga.i);
        }
        if constexpr (l + 2 < LV) pcr_factor_level_all<ROW, LV, l + 1>(L, ar, dr, cr, ac, dc, cc, F);
        return;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // real shift.  Every lane inverts its own diagonal once and the neighbours fetch the reciprocal (the same
        // bits as inverting the fetched diagonal on both sides, half the reciprocals)
        const double id = rcp(dr[k]);
        double id_lo, id_hi, a_lo, a_hi, c_lo, c_hi;
        both<ROW, s>(L, id, id_lo, id_hi); both<ROW, s>(L, ar[k], a_lo, a_hi); both<ROW, s>(L, cr[k], c_lo, c_hi);
        // a == 0 where there is no lower neighbour (c likewise), so alpha/gamma vanish there by
        // themselves once the foreign operands are finite (keep_m folds into the cross-lane move)
        const double al = ar[k] * keep_m(L.m_lo[l], id_lo);
        const double ga = cr[k] * keep_m(L.m_hi[l], id_hi);
        dr[k] = dr[k] - al * keep_m(L.m_lo[l], c_lo) - ga * keep_m(L.m_hi[l], a_hi);
        ar[k] = -al * keep_m(L.m_lo[l], a_lo);
        cr[k] = -ga * keep_m(L.m_hi[l], c_hi);
        F.st2(k * S::RS + 2 * l, al, ga);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // complex shift, likewise
        const cplx cid = cinv(dc[k]);
        cplx i_lo, i_hi, a_lo, a_hi, c_lo, c_hi;
        cboth<ROW, s>(L, cid, i_lo, i_hi); cboth<ROW, s>(L, ac[k], a_lo, a_hi); cboth<ROW, s>(L, cc[k], c_lo, c_hi);
        const cplx il = {keep_m(L.m_lo[l], i_lo.r), keep_m(L.m_lo[l], i_lo.i)};
        const cplx ih = {keep_m(L.m_hi[l], i_hi.r), keep_m(L.m_hi[l], i_hi.i)};
        const cplx al = cmul(ac[k], il);
        const cplx ga = cmul(cc[k], ih);
        {   // d -= al * c_lo + ga * a_hi, eight fused multiply-adds
            const cplx cl = {keep_m(L.m_lo[l], c_lo.r), keep_m(L.m_lo[l], c_lo.i)};
            const cplx ah = {keep_m(L.m_hi[l], a_hi.r), keep_m(L.m_hi[l], a_hi.i)};
            double dre = dc[k].r, dim = dc[k].i;
            dre = __builtin_fma(-al.r, cl.r, dre); dim = __builtin_fma(-al.r, cl.i, dim);
            dre = __builtin_fma(al.i, cl.i, dre);  dim = __builtin_fma(-al.i, cl.r, dim);
            dre = __builtin_fma(-ga.r, ah.r, dre); dim = __builtin_fma(-ga.r, ah.i, dim);
            dre = __builtin_fma(ga.i, ah.i, dre);  dim = __builtin_fma(-ga.i, ah.r, dim);
            dc[k] = {dre, dim};
        }
        const cplx na = cmul(al, {keep_m(L.m_lo[l], a_lo.r), keep_m(L.m_lo[l], a_lo.i)});
        const cplx nc = cmul(ga, {keep_m(L.m_hi[l], c_hi.r), keep_m(L.m_hi[l], c_hi.i)});
        ac[k] = {-na.r, -na.i};
        cc[k] = {-nc.r, -nc.i};
        const int c0 = S::CB + k * S::CS + 4 * l;
        F.st2(c0, al.r, al.i); F.st2(c0 + 2, ga.r, ga.i);
    }
    if constexpr (l + 2 < LV) pcr_factor_level_all<ROW, LV, l + 1>(L, ar, dr, cr, ac, dc, cc, F);
}

// The top level (stride 2^(LV-1) >= n/2): a zone couples to its one partner only -- `a` is zero below the stride, `c`
// at and above it, so a + c is whichever is there (exactly), and the partner's a + c the coefficient that couples
// back.  One factor and a third of the cross-lane moves of a regular level; the same bits.
template <bool ROW, int LV>
__device__ __forceinline__ void pcr_factor_top_all(const Lane &L, double ar[3], double dr[3], double cr[3],
                                                   cplx ac[3], cplx dc[3], cplx cc[3], FStore<LV> &F, double ftop[3])
{
    using S = FSlots<LV>;
    constexpr int s = 1 << (LV - 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double id = rcp(dr[k]), w = ar[k] + cr[k];
        const double p_id = from_partner<ROW, s>(L, id), p_w = from_partner<ROW, s>(L, w);
        const double f = w * p_id;
        dr[k] = dr[k] - f * p_w;
        ftop[k] = f;                                  // stored with 1/d, its pair (factorize)
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const cplx cid = cinv(dc[k]), w = {ac[k].r + cc[k].r, ac[k].i + cc[k].i};
        const cplx p_id = {from_partner<ROW, s>(L, cid.r), from_partner<ROW, s>(L, cid.i)};
        const cplx p_w = {from_partner<ROW, s>(L, w.r), from_partner<ROW, s>(L, w.i)};
        const cplx f = cmul(w, p_id);
        double dre = dc[k].r, dim = dc[k].i;
        dre = __builtin_fma(-f.r, p_w.r, dre); dim = __builtin_fma(-f.r, p_w.i, dim);
        dre = __builtin_fma(f.i, p_w.i, dre);  dim = __builtin_fma(-f.i, p_w.r, dim);
        dc[k] = {dre, dim};
        const int c0 = S::CB + k * S::CS + 4 * (LV - 1);
        F.st2(c0, f.r, f.i);
    }
}

// The six factored systems of one (h, J) pair: scipy's LU_real / LU_complex.
template <bool ROW, int LV>
__device__ __forceinline__ void factorize(const Lane &L, const Jac &J, double h, FStore<LV> &F)
{
    using S = FSlots<LV>;
    // radau.py:454-456: MU_REAL / h * I - J ; MU_COMPLEX / h * I - J   (systems: 0 = T, 1 = pH, 2 = Cl)
    const double ih = rcp(h);
    const double mr = rc::MU_REAL * ih, mcr = rc::MU_CR * ih, mci = rc::MU_CI * ih;
    double ar[3] = {-J.tt[0], -J.pp[0], -J.cc[0]};
    double dr[3] = {mr - J.tt[1], mr - J.pp[1], mr - J.cc[1]};
    double cr[3] = {-J.tt[2], -J.pp[2], -J.cc[2]};
    cplx ac[3] = {{ar[0], 0.0}, {ar[1], 0.0}, {ar[2], 0.0}};
    cplx dc[3] = {{mcr - J.tt[1], mci}, {mcr - J.pp[1], mci}, {mcr - J.cc[1], mci}};
    cplx cc[3] = {{cr[0], 0.0}, {cr[1], 0.0}, {cr[2], 0.0}};
    if constexpr (LV > 1) pcr_factor_level_all<ROW, LV, 0>(L, ar, dr, cr, ac, dc, cc, F);
    double ftop[3];
    pcr_factor_top_all<ROW, LV>(L, ar, dr, cr, ac, dc, cc, F, ftop);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        F.st2(k * S::RS + 2 * (LV - 1), ftop[k], rcp(dr[k]));
        const cplx inv = cinv(dc[k]);
        F.st2(S::CB + k * S::CS + 4 * LV - 2, inv.r, inv.i);
    }
}

// x = (mu_real/h I - J)^-1 b, in place, b indexed by species.  The factors of all three systems are
// fetched from LDS in one batch up front (one exposed LDS round trip instead of one per level).
template <int LV> struct RealFactors { double a[LV > 1 ? LV - 1 : 1], g[LV > 1 ? LV - 1 : 1], f, inv; };

template <int LV>
__device__ __forceinline__ void load_real(const FStore<LV> &F, int k, RealFactors<LV> &s)
{
    using S = FSlots<LV>;
#pragma unroll
    for (int l = 0; l + 1 < LV; ++l) F.ld2(k * S::RS + 2 * l, s.a[l], s.g[l]);
    F.ld2(k * S::RS + 2 * (LV - 1), s.f, s.inv);
}

template <bool ROW, int LV, int l>
__device__ __forceinline__ void pcr_real_level(const Lane &L, const RealFactors<LV> &s, double &b)
{
    if constexpr (l + 1 < LV) {
        constexpr int st = 1 << l;
        double b_lo, b_hi;
        both<ROW, st>(L, b, b_lo, b_hi);
        b = b - s.a[l] * keep_m(L.m_lo[l], b_lo) - s.g[l] * keep_m(L.m_hi[l], b_hi);
        pcr_real_level<ROW, LV, l + 1>(L, s, b);
    } else {
        b = b - s.f * from_partner<ROW, (1 << (LV - 1))>(L, b);     // top level: one partner
    }
}

// A product that must be rounded on its own (the general path adds J_cT x_T as a finished sum; the t_local path's lone
// product must not be fused into the addition that follows, or the two paths differ in the last bit and a reactor's
// result would depend on its wavefront's other reactors).
__device__ __forceinline__ double rounded(double x) { asm volatile("" : "+v"(x)); return x; }

// Row-straddling zone counts, no coupling to a neighbour's temperature (t_local): the T and the pH system are
// independent, so they go through the levels in lock step -- both systems' neighbours in one batch of exchanges (one LDS
// round trip per level instead of two) and the level's factors fetched along with them (never more than one level of
// factors in registers).  The arithmetic per system is that of the one-system levels.
template <int LV, int l>
__device__ __forceinline__ void pcr_real_pair(const Lane &L, const FStore<LV> &F, double &b0, double &b1)
{
    using S = FSlots<LV>;
    if constexpr (l + 1 < LV) {
        constexpr int st = 1 << l;
        double lo0, hi0, lo1, hi1;
        both<false, st>(L, b0, lo0, hi0); both<false, st>(L, b1, lo1, hi1);
        double a0, g0, a1, g1;
        F.ld2(2 * l, a0, g0); F.ld2(S::RS + 2 * l, a1, g1);
        b0 = b0 - a0 * keep_m(L.m_lo[l], lo0) - g0 * keep_m(L.m_hi[l], hi0);
        b1 = b1 - a1 * keep_m(L.m_lo[l], lo1) - g1 * keep_m(L.m_hi[l], hi1);
        pcr_real_pair<LV, l + 1>(L, F, b0, b1);
    } else {
        constexpr int st = 1 << (LV - 1);
        const double p0 = from_partner<false, st>(L, b0), p1 = from_partner<false, st>(L, b1);
        const double f0 = F.ld(2 * (LV - 1)), f1 = F.ld(S::RS + 2 * (LV - 1));
        b0 = b0 - f0 * p0;
        b1 = b1 - f1 * p1;
    }
}

template <bool ROW, int LV>
__device__ __forceinline__ void solve_real(const Lane &L, const Jac &J, const FStore<LV> &F, double b[3], bool t_local)
{
    if constexpr (!ROW && LV >= 2) {
        if (t_local) {
            using S = FSlots<LV>;
            double xT = b[STT], xP = b[SPH];
            pcr_real_pair<LV, 0>(L, F, xT, xP);
            xT *= F.ld(2 * LV - 1); xP *= F.ld(S::RS + 2 * LV - 1);
            RealFactors<LV> fC;
            load_real<LV>(F, 2, fC);
            const double tC = rounded(J.ct[1] * xT);
            double xC = b[SCL] + tC + J.cp * xP;
            pcr_real_level<ROW, LV, 0>(L, fC, xC);
            xC *= fC.inv;
            b[SPH] = xP; b[SCL] = xC; b[STT] = xT;
            return;
        }
    }
    // (many levels: a system's factors are fetched when its turn comes, or the three sets together crowd the register file)
    RealFactors<LV> fT, fP, fC;
    load_real<LV>(F, 0, fT);
    if constexpr (LV < 4) { load_real<LV>(F, 1, fP); load_real<LV>(F, 2, fC); }
    double xT = b[STT];
    pcr_real_level<ROW, LV, 0>(L, fT, xT);
    xT *= fT.inv;
    if constexpr (LV >= 4) load_real<LV>(F, 1, fP);
    double xP, tC;
    if (t_local) {      // (wave-uniform) no row of this wavefront couples to a neighbour's temperature: see jac_t_local
        xP = b[SPH];
        tC = rounded(J.ct[1] * xT);
    } else {
        const double xT_lo_r = from_lo<ROW, 1>(L, xT), xT_hi_r = from_hi<ROW, 1>(L, xT);
        const double xT_lo = keep_m(L.m_lo[0], xT_lo_r), xT_hi = keep_m(L.m_hi[0], xT_hi_r); // J.pt/ct[0,2] are 0 there
        xP = b[SPH] + (J.pt[0] * xT_lo + J.pt[1] * xT + J.pt[2] * xT_hi);
        tC = J.ct[0] * xT_lo + J.ct[1] * xT + J.ct[2] * xT_hi;
    }
    pcr_real_level<ROW, LV, 0>(L, fP, xP);
    xP *= fP.inv;
    if constexpr (LV >= 4) load_real<LV>(F, 2, fC);
    double xC = b[SCL] + tC + J.cp * xP;
    pcr_real_level<ROW, LV, 0>(L, fC, xC);
    xC *= fC.inv;
    b[SPH] = xP; b[SCL] = xC; b[STT] = xT;
}

// Real and complex solve of one Newton iteration, level by level in lock step: the two
// cyclic-reduction chains are independent, so interleaving them doubles the instruction-level
// parallelism of what is otherwise one long dependent chain, and each system's factors are
// fetched from LDS in one batch (one wait per system instead of one per level).
template <int LV> struct SysFactors {
    static constexpr int NL = LV > 1 ? LV - 1 : 1;
    double ra[NL], rg[NL], rf, rinv; cplx ca[NL], cg[NL], cf, cinv;
};

template <int LV>
__device__ __forceinline__ void load_sys(const FStore<LV> &F, int k, SysFactors<LV> &s)
{
    using S = FSlots<LV>;
    const int r0 = k * S::RS, c0 = S::CB + k * S::CS;
#pragma unroll
    for (int l = 0; l + 1 < LV; ++l) {
        F.ld2(r0 + 2 * l, s.ra[l], s.rg[l]);
        F.ld2(c0 + 4 * l, s.ca[l].r, s.ca[l].i);
        F.ld2(c0 + 4 * l + 2, s.cg[l].r, s.cg[l].i);
    }
    F.ld2(r0 + 2 * (LV - 1), s.rf, s.rinv);
    F.ld2(c0 + 4 * (LV - 1), s.cf.r, s.cf.i);
    F.ld2(c0 + 4 * LV - 2, s.cinv.r, s.cinv.i);
}

// one level of a real + complex pair of systems, given the (masked) neighbours' values and the level's factors
__device__ __forceinline__ void rc_apply(double ra, double rg, cplx ca, cplx cg, double b_lo, double b_hi, cplx c_lo, cplx c_hi,
                                         double &b, cplx &c)
{
    b = b - ra * b_lo - rg * b_hi;
    // c -= ca * c_lo + cg * c_hi as eight fused multiply-adds (no separate products and subtractions)
    double cr = c.r, ci = c.i;
    cr = __builtin_fma(-ca.r, c_lo.r, cr); ci = __builtin_fma(-ca.r, c_lo.i, ci);
    cr = __builtin_fma(ca.i, c_lo.i, cr);  ci = __builtin_fma(-ca.i, c_lo.r, ci);
    cr = __builtin_fma(-cg.r, c_hi.r, cr); ci = __builtin_fma(-cg.r, c_hi.i, ci);
    cr = __builtin_fma(cg.i, c_hi.i, cr);  ci = __builtin_fma(-cg.i, c_hi.r, ci);
    c = {cr, ci};
}
// the top level: one partner, one factor
__device__ __forceinline__ void rc_apply_top(double rf, cplx cf, double b_p, cplx c_p, double &b, cplx &c)
{
    b = b - rf * b_p;
    double cr = c.r, ci = c.i;
    cr = __builtin_fma(-cf.r, c_p.r, cr); ci = __builtin_fma(-cf.r, c_p.i, ci);
    cr = __builtin_fma(cf.i, c_p.i, cr);  ci = __builtin_fma(-cf.i, c_p.r, ci);
    c = {cr, ci};
}
template <bool ROW, int LV, int l>
__device__ __forceinline__ void rc_neighbours(const Lane &L, double b, cplx c, double &b_lo, double &b_hi, cplx &c_lo, cplx &c_hi)
{
    constexpr int st = 1 << l;
    both<ROW, st>(L, b, b_lo, b_hi); cboth<ROW, st>(L, c, c_lo, c_hi);
    b_lo = keep_m(L.m_lo[l], b_lo); b_hi = keep_m(L.m_hi[l], b_hi);
    c_lo = {keep_m(L.m_lo[l], c_lo.r), keep_m(L.m_lo[l], c_lo.i)};
    c_hi = {keep_m(L.m_hi[l], c_hi.r), keep_m(L.m_hi[l], c_hi.i)};
}

template <bool ROW, int LV, int l>
__device__ __forceinline__ void pcr_rc_level(const Lane &L, const SysFactors<LV> &s, double &b, cplx &c)
{
    if constexpr (l + 1 < LV) {
        double b_lo, b_hi; cplx c_lo, c_hi;
        rc_neighbours<ROW, LV, l>(L, b, c, b_lo, b_hi, c_lo, c_hi);
        rc_apply(s.ra[l], s.rg[l], s.ca[l], s.cg[l], b_lo, b_hi, c_lo, c_hi, b, c);
        pcr_rc_level<ROW, LV, l + 1>(L, s, b, c);
    } else {
        constexpr int st = 1 << (LV - 1);
        const double b_p = from_partner<ROW, st>(L, b);
        const cplx c_p = {from_partner<ROW, st>(L, c.r), from_partner<ROW, st>(L, c.i)};
        rc_apply_top(s.rf, s.cf, b_p, c_p, b, c);
    }
}

// T and pH systems in lock step (see pcr_real_pair)
template <int LV, int l>
__device__ __forceinline__ void pcr_rc_pair(const Lane &L, const FStore<LV> &F, double &b0, cplx &c0, double &b1, cplx &c1)
{
    using S = FSlots<LV>;
    constexpr int r0 = 0, r1 = S::RS, q0 = S::CB, q1 = S::CB + S::CS;
    if constexpr (l + 1 < LV) {
        double bl0, bh0, bl1, bh1; cplx cl0, ch0, cl1, ch1;
        rc_neighbours<false, LV, l>(L, b0, c0, bl0, bh0, cl0, ch0);
        rc_neighbours<false, LV, l>(L, b1, c1, bl1, bh1, cl1, ch1);
        double ra0, rg0, ra1, rg1; cplx ca0, cg0, ca1, cg1;
        F.ld2(r0 + 2 * l, ra0, rg0); F.ld2(r1 + 2 * l, ra1, rg1);
        F.ld2(q0 + 4 * l, ca0.r, ca0.i); F.ld2(q0 + 4 * l + 2, cg0.r, cg0.i);
        F.ld2(q1 + 4 * l, ca1.r, ca1.i); F.ld2(q1 + 4 * l + 2, cg1.r, cg1.i);
        rc_apply(ra0, rg0, ca0, cg0, bl0, bh0, cl0, ch0, b0, c0);
        rc_apply(ra1, rg1, ca1, cg1, bl1, bh1, cl1, ch1, b1, c1);
        pcr_rc_pair<LV, l + 1>(L, F, b0, c0, b1, c1);
    } else {
        constexpr int st = 1 << (LV - 1);
        const double p0 = from_partner<false, st>(L, b0), p1 = from_partner<false, st>(L, b1);
        const cplx cp0 = {from_partner<false, st>(L, c0.r), from_partner<false, st>(L, c0.i)};
        const cplx cp1 = {from_partner<false, st>(L, c1.r), from_partner<false, st>(L, c1.i)};
        const double rf0 = F.ld(r0 + 2 * (LV - 1)), rf1 = F.ld(r1 + 2 * (LV - 1));
        cplx cf0, cf1;
        F.ld2(q0 + 4 * (LV - 1), cf0.r, cf0.i); F.ld2(q1 + 4 * (LV - 1), cf1.r, cf1.i);
        rc_apply_top(rf0, cf0, p0, cp0, b0, c0);
        rc_apply_top(rf1, cf1, p1, cp1, b1, c1);
    }
}

template <bool ROW, int LV>
__device__ __forceinline__ void solve_rc(const Lane &L, const Jac &J, const FStore<LV> &F,
                                         double br[3], double cr[3], double ci[3], bool t_local)
{
    // Few levels: all three systems' factors are fetched ahead of their use (one exposed LDS round trip instead
    // of three).  Many levels (n > 8): 6 LV + 3 doubles per system -- fetched system by system, or the three sets
    // together overflow the register file into scratch.
    SysFactors<LV> sT, sP, sC;
    if constexpr (!ROW && LV >= 2) {
        if (t_local) {
            using S = FSlots<LV>;
            double xT = br[STT], xP = br[SPH]; cplx zT = {cr[STT], ci[STT]}, zP = {cr[SPH], ci[SPH]};
            pcr_rc_pair<LV, 0>(L, F, xT, zT, xP, zP);
            xT *= F.ld(2 * LV - 1); zT = cmul(zT, {F.ld(S::CB + 4 * LV - 2), F.ld(S::CB + 4 * LV - 1)});
            xP *= F.ld(S::RS + 2 * LV - 1); zP = cmul(zP, {F.ld(S::CB + S::CS + 4 * LV - 2), F.ld(S::CB + S::CS + 4 * LV - 1)});
            load_sys<LV>(F, 2, sC);
            const double tC = rounded(J.ct[1] * xT); const cplx uC = {rounded(J.ct[1] * zT.r), rounded(J.ct[1] * zT.i)};
            double xC = br[SCL] + tC + J.cp * xP;
            cplx zC = {cr[SCL] + uC.r + J.cp * zP.r, ci[SCL] + uC.i + J.cp * zP.i};
            pcr_rc_level<ROW, LV, 0>(L, sC, xC, zC);
            xC *= sC.rinv; zC = cmul(zC, sC.cinv);
            br[SPH] = xP; br[SCL] = xC; br[STT] = xT;
            cr[SPH] = zP.r; ci[SPH] = zP.i; cr[SCL] = zC.r; ci[SCL] = zC.i; cr[STT] = zT.r; ci[STT] = zT.i;
            return;
        }
    }
    load_sys<LV>(F, 0, sT);
    if constexpr (LV < 4) load_sys<LV>(F, 1, sP);
    // temperature block
    double xT = br[STT]; cplx zT = {cr[STT], ci[STT]};
    pcr_rc_level<ROW, LV, 0>(L, sT, xT, zT);
    xT *= sT.rinv; zT = cmul(zT, sT.cinv);
    if constexpr (LV < 4) load_sys<LV>(F, 2, sC); else load_sys<LV>(F, 1, sP);
    double xP, tC; cplx zP, uC;
    if (t_local) {      // (wave-uniform) no row of this wavefront couples to a neighbour's temperature: see jac_t_local
        xP = br[SPH]; zP = {cr[SPH], ci[SPH]};
        tC = rounded(J.ct[1] * xT); uC = {rounded(J.ct[1] * zT.r), rounded(J.ct[1] * zT.i)};
    } else {
        const double xT_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, xT)), xT_hi = keep_m(L.m_hi[0], from_hi<ROW, 1>(L, xT));
        const cplx zT_lo = {keep_m(L.m_lo[0], from_lo<ROW, 1>(L, zT.r)), keep_m(L.m_lo[0], from_lo<ROW, 1>(L, zT.i))};
        const cplx zT_hi = {keep_m(L.m_hi[0], from_hi<ROW, 1>(L, zT.r)), keep_m(L.m_hi[0], from_hi<ROW, 1>(L, zT.i))};
        // pH block: rhs += J_pT x_T
        xP = br[SPH] + (J.pt[0] * xT_lo + J.pt[1] * xT + J.pt[2] * xT_hi);
        zP = {cr[SPH] + (J.pt[0] * zT_lo.r + J.pt[1] * zT.r + J.pt[2] * zT_hi.r),
              ci[SPH] + (J.pt[0] * zT_lo.i + J.pt[1] * zT.i + J.pt[2] * zT_hi.i)};
        tC = J.ct[0] * xT_lo + J.ct[1] * xT + J.ct[2] * xT_hi;
        uC = {J.ct[0] * zT_lo.r + J.ct[1] * zT.r + J.ct[2] * zT_hi.r, J.ct[0] * zT_lo.i + J.ct[1] * zT.i + J.ct[2] * zT_hi.i};
    }
    pcr_rc_level<ROW, LV, 0>(L, sP, xP, zP);
    xP *= sP.rinv; zP = cmul(zP, sP.cinv);
    if constexpr (LV >= 4) load_sys<LV>(F, 2, sC);
    // chlorine block: rhs += J_cT x_T + J_cp x_p
    double xC = br[SCL] + tC + J.cp * xP;
    cplx zC = {cr[SCL] + uC.r + J.cp * zP.r, ci[SCL] + uC.i + J.cp * zP.i};
    pcr_rc_level<ROW, LV, 0>(L, sC, xC, zC);
    xC *= sC.rinv; zC = cmul(zC, sC.cinv);
    br[SPH] = xP; br[SCL] = xC; br[STT] = xT;
    cr[SPH] = zP.r; ci[SPH] = zP.i; cr[SCL] = zC.r; ci[SCL] = zC.i; cr[STT] = zT.r; ci[STT] = zT.i;
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

struct ZoneProps { double H, iw, phi, kT, rho; bool bpos; };

// What the rows of this lane see of their neighbourhood at the unperturbed state (once per Jacobian)
struct FdBase { double k_lo, k_hi, kd, H_lo, H_hi, C_lo, C_hi, T_lo, T_hi, rho_hi; };

template <bool ROW>
__device__ __forceinline__ FdBase fd_base(const Lane &L, const RK &k, const double y[3], const ZoneProps &b)
{
    FdBase n;
    n.rho_hi = from_hi<ROW, 1>(L, b.rho);
    n.k_hi = k_above(k, b.rho, n.rho_hi);
    n.k_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, n.k_hi));
    n.kd = k_diag(k, n.k_lo, n.k_hi);
    n.H_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, b.H)); n.H_hi = keep_m(L.m_hi[0], from_hi<ROW, 1>(L, b.H));
    n.C_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, y[SCL])); n.C_hi = keep_m(L.m_hi[0], from_hi<ROW, 1>(L, y[SCL]));
    n.T_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, y[STT])); n.T_hi = keep_m(L.m_hi[0], from_hi<ROW, 1>(L, y[STT]));
    return n;
}

// The columns of one species: every zone's own value perturbed by its own step.  A row of zone i changes with the
// columns of zones i-1, i, i+1 only, and which of its inputs a column reaches is known: so each row is evaluated
// three times -- with its own zone's perturbed values, with what its lower neighbour exports when perturbed, with what
// its upper neighbour exports -- everything else at the base values.  This is f(y + h e_j) - f(y) of common.py:331-333
// restricted to the rows that can change; rows that cannot depend on a column are structural zeros there too.
// out.D[q][r] is the change of this lane's row q when the zone at offset r-1 was perturbed.
// colmask (retry pass, common.py:343-361): only the flagged columns are perturbed.
template <bool ROW, int SP, class KC>
__device__ __forceinline__ void fd_species_pass(const Lane &L, const KC &kc, const RK &k, const double y[3], const double f[3],
                                                const ZoneProps &b, const FdBase &n, double hcol, bool colmask, bool all_cols,
                                                FdCols &out, bool &bad, double &badval)
{
    const double ypert = y[SP] + hcol;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 3; ++r) { out.D[q][r] = 0.0; out.S[q][r] = 0.0; }
    auto put = [&](int q, int r, double fn) { out.D[q][r] = fn - f[q]; out.S[q][r] = fmax(fabs(f[q]), fabs(fn)); };
    const double kphi = b.kT * b.phi;
    if constexpr (SP == SPH) {
        PropPH p = prop_pH(kc, k, ypert);
        if (!all_cols && !colmask) { p.H = b.H; p.iw = b.iw; p.phi = b.phi; }        // this column is not part of the retry
        const double Hx_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, p.H)), Hx_hi = keep_m(L.m_hi[0], from_hi<ROW, 1>(L, p.H));
        put(SPH, 1, row_pH(k, mix3(n.k_lo, n.k_hi, n.kd, n.H_lo, n.H_hi, p.H), p.H, p.iw));
        put(SCL, 1, row_Cl(k, mix3(n.k_lo, n.k_hi, n.kd, n.C_lo, n.C_hi, y[SCL]), y[SCL], b.kT * p.phi));
        put(SPH, 0, row_pH(k, mix3(n.k_lo, n.k_hi, n.kd, Hx_lo, n.H_hi, b.H), b.H, b.iw));
        put(SPH, 2, row_pH(k, mix3(n.k_lo, n.k_hi, n.kd, n.H_lo, Hx_hi, b.H), b.H, b.iw));
    }
    if constexpr (SP == SCL) {
        const double cx = (all_cols || colmask) ? ypert : y[SCL];
        const double Cx_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, cx)), Cx_hi = keep_m(L.m_hi[0], from_hi<ROW, 1>(L, cx));
        put(SCL, 1, row_Cl(k, mix3(n.k_lo, n.k_hi, n.kd, n.C_lo, n.C_hi, cx), cx, kphi));
        put(SCL, 0, row_Cl(k, mix3(n.k_lo, n.k_hi, n.kd, Cx_lo, n.C_hi, y[SCL]), y[SCL], kphi));
        put(SCL, 2, row_Cl(k, mix3(n.k_lo, n.k_hi, n.kd, n.C_lo, Cx_hi, y[SCL]), y[SCL], kphi));
    }
    if constexpr (SP == STT) {
        PropT p = prop_T(kc, ypert);
        if (colmask && p.bad && !bad) { bad = true; badval = ypert; }   // the reference raises on this perturbed column
        double tx = ypert;
        if (!all_cols && !colmask) { p.kT = b.kT; p.rho = b.rho; tx = y[STT]; }
        const double Tx_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, tx)), Tx_hi = keep_m(L.m_hi[0], from_hi<ROW, 1>(L, tx));
        // the two interfaces of a zone move with its density: K[i,i+1] with this zone / with the zone above perturbed
        const double khi_own = k_above(k, p.rho, n.rho_hi);
        const double khi_up = k_above(k, b.rho, from_hi<ROW, 1>(L, p.rho));
        const double klo_own = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, khi_up));    // K[i,i-1] with this zone perturbed
        const double klo_dn = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, khi_own));    // ... with the zone below perturbed
        {   // own zone
            const double kd = k_diag(k, klo_own, khi_own);
            put(SPH, 1, row_pH(k, mix3(klo_own, khi_own, kd, n.H_lo, n.H_hi, b.H), b.H, b.iw));
            put(SCL, 1, row_Cl(k, mix3(klo_own, khi_own, kd, n.C_lo, n.C_hi, y[SCL]), y[SCL], p.kT * b.phi));
            put(STT, 1, row_T(k, mix3(klo_own, khi_own, kd, n.T_lo, n.T_hi, tx), tx));
        }
        {   // zone below
            const double kd = k_diag(k, klo_dn, n.k_hi);
            put(SPH, 0, row_pH(k, mix3(klo_dn, n.k_hi, kd, n.H_lo, n.H_hi, b.H), b.H, b.iw));
            put(SCL, 0, row_Cl(k, mix3(klo_dn, n.k_hi, kd, n.C_lo, n.C_hi, y[SCL]), y[SCL], kphi));
            put(STT, 0, row_T(k, mix3(klo_dn, n.k_hi, kd, Tx_lo, n.T_hi, y[STT]), y[STT]));
        }
        {   // zone above
            const double kd = k_diag(k, n.k_lo, khi_up);
            put(SPH, 2, row_pH(k, mix3(n.k_lo, khi_up, kd, n.H_lo, n.H_hi, b.H), b.H, b.iw));
            put(SCL, 2, row_Cl(k, mix3(n.k_lo, khi_up, kd, n.C_lo, n.C_hi, y[SCL]), y[SCL], kphi));
            put(STT, 2, row_T(k, mix3(n.k_lo, khi_up, kd, n.T_lo, Tx_hi, y[STT]), y[STT]));
        }
    }
}

// For the column owned by this lane (species SP): max |diff| over its rows with
// numpy argmax tie-breaking (first row in [pH.., Cl.., T..] order) and the
// matching scale (common.py:335-339).
template <bool ROW, int SP>
__device__ __forceinline__ void fd_col_reduce(const Lane &L, const FdCols &c, double &maxd, double &scale)
{
    maxd = -1.0; scale = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        // rows of species q that can depend on a column of species SP
        const bool dep_nb = (q == SP) || (SP == STT);           // neighbour-zone rows
        const bool dep_own = dep_nb || (q == SCL && SP == SPH); // own-zone row
        if (!dep_own) continue;
        double d_lo = 0, s_lo = 0, d_hi = 0, s_hi = 0;
        if (dep_nb) {
            d_lo = from_lo<ROW, 1>(L, c.D[q][2]); s_lo = from_lo<ROW, 1>(L, c.S[q][2]); // lane z-1 saw this column at rel=+1
            d_hi = from_hi<ROW, 1>(L, c.D[q][0]); s_hi = from_hi<ROW, 1>(L, c.S[q][0]); // lane z+1 saw it at rel=-1
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

// One species' columns: perturb, reduce, optional retry, factor update.  Leaves
// the finished difference quotients of this species' columns in `cols.D`
// (already divided by the column's h).
template <bool ROW, int SP, class KC>
__device__ __forceinline__ void num_jac_species(const Lane &L, const KC &kc, const RK &k, const double y[3], const double f[3],
                                                const ZoneProps &b, const FdBase &nb, double &fac, FdCols &cols, bool &bad, double &badval)
{
    const double fs = (f[SP] >= 0) ? 1.0 : -1.0;
    const double ysc = fs * fmax(ATOL, fabs(y[SP]));
    double h = fd_step(y[SP], fac, ysc);
    while (WT_RARE(h == 0)) { fac *= 10; h = fd_step(y[SP], fac, ysc); }    // common.py:327-330
    fd_species_pass<ROW, SP>(L, kc, k, y, f, b, nb, h, true, true, cols, bad, badval);
    double maxd, scl;
    fd_col_reduce<ROW, SP>(L, cols, maxd, scl);
    const bool small = maxd < rc::NJ_REJECT * scl;                  // common.py:341
    if (WT_RARE(__ballot(small) != 0ull)) {                         // rare: one retry with 10x factor
        const double nf = 10 * fac;
        const double hn = fd_step(y[SP], nf, ysc);
        FdCols c2;
        fd_species_pass<ROW, SP>(L, kc, k, y, f, b, nb, hn, small, false, c2, bad, badval);
        double md2, sc2;
        fd_col_reduce<ROW, SP>(L, c2, md2, sc2);
        const bool upd = small && (maxd * sc2 < md2 * scl);         // common.py:354
        if (upd) { fac = nf; h = hn; maxd = md2; scl = sc2; }
        const int iu = upd ? 1 : 0;
        const int iu_lo = __shfl_up(iu, 1, 64), iu_hi = __shfl_down(iu, 1, 64);
        const bool upd_lo = L.has_lo && (iu_lo != 0), upd_hi = L.has_hi && (iu_hi != 0);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (upd_lo) cols.D[q][0] = c2.D[q][0];
            if (upd) cols.D[q][1] = c2.D[q][1];
            if (upd_hi) cols.D[q][2] = c2.D[q][2];
        }
    }
    // diff /= h (column-wise; the column's h lives in the column's lane)
    const double h_lo = from_lo<ROW, 1>(L, h), h_hi = from_hi<ROW, 1>(L, h);
    const double ih0 = L.has_lo ? rcp(h_lo) : 0.0, ih1 = rcp(h), ih2 = L.has_hi ? rcp(h_hi) : 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) { cols.D[q][0] *= ih0; cols.D[q][1] *= ih1; cols.D[q][2] *= ih2; }
    // factor adaptation common.py:363-365
    const bool sm = maxd < rc::NJ_SMALL * scl, bg = maxd > rc::NJ_BIG * scl;
    if (sm) fac *= 10;
    if (bg) fac *= 0.1;
    fac = fmax(fac, rc::NJ_MINF);
}

// The three species' passes are sections like those of rhs_points: each fetches the constants it works with.
// KTP: pointer to the constant table (kernel-argument segment in the step kernel).
template <bool ROW, class KTP>
__device__ __forceinline__ void num_jac(const Lane &L, const RKStore &ks, KTP ktab, const double y[3], const double f[3],
                                        double fac[3], bool &have_fac, Jac &J, bool &bad, double &badval, bool &t_dense)
{
    if (!have_fac) { fac[0] = fac[1] = fac[2] = rc::NJ_F0; have_fac = true; }
    FdCols cols;
    ZoneProps b;
    FdBase nb;
    {
        const KT ct = load_kt(ktab());
        const PropT bpt = prop_T(ct, y[STT]);
        b.kT = bpt.kT; b.rho = bpt.rho;
    }
    {
        const KP cp = load_kp(ktab());
        const RK k = fetch_reactor(ks);
        const PropPH bpp = prop_pH(cp, k, y[SPH]);
        b.H = bpp.H; b.iw = bpp.iw; b.phi = bpp.phi; b.bpos = bpp.bpos;
        nb = fd_base<ROW>(L, k, y, b);
        num_jac_species<ROW, SPH>(L, cp, k, y, f, b, nb, fac[SPH], cols, bad, badval);
#pragma unroll
        for (int r = 0; r < 3; ++r) J.pp[r] = cols.D[SPH][r];
        J.cp = cols.D[SCL][1];
    }
    {
        const RK k = fetch_reactor(ks);
        num_jac_species<ROW, SCL>(L, 0, k, y, f, b, nb, fac[SCL], cols, bad, badval);
#pragma unroll
        for (int r = 0; r < 3; ++r) J.cc[r] = cols.D[SCL][r];
    }
    {
        const KT ct = load_kt(ktab());
        const RK k = fetch_reactor(ks);
        num_jac_species<ROW, STT>(L, ct, k, y, f, b, nb, fac[STT], cols, bad, badval);
#pragma unroll
        for (int r = 0; r < 3; ++r) { J.tt[r] = cols.D[STT][r]; J.pt[r] = cols.D[SPH][r]; J.ct[r] = cols.D[SCL][r]; }
        t_dense = jac_t_dense(J) || (ct.dense_bias != 0.0);
    }
}

// ---------------------------------------------------------------- helpers
template <bool ROW, int LV = 6>
__device__ __forceinline__ double rms3(const Lane &L, const double x[3], const double sc[3])
{
    // common.py:63-65 norm(x / scale) over the 3n components of one reactor
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) { const double v = x[q] * rcp(sc[q]); s += v * v; }
    return sqrt_k(div_by(seg_sum<ROW, LV>(L, s), L.d3n));
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
    const double ie = rcp(error_norm);           // +inf for error_norm == 0, as numpy's 0 ** -0.25
    if (have_old && error_norm != 0) mult = h_abs * rcp(h_abs_old) * root4(error_norm_old * ie);
    return fmin(1.0, mult) * root4(ie);
}

// ---------------------------------------------------------------- the solver state machine
// The per-reactor yes/no state of the solver lives in the bits of ONE VGPR (`fl` in step_kernel).  As separate
// `bool`s every one of them is a 64-bit lane mask in an SGPR pair for the whole loop -- two dozen of them exhaust
// the scalar register file and the compiler spills SGPRs through v_writelane / v_readlane.
struct Flag {
    uint32_t &w; const uint32_t m;
    __device__ __forceinline__ operator bool() const { return (w & m) != 0u; }
    __device__ __forceinline__ Flag &operator=(bool v) { w = v ? (w | m) : (w & ~m); return *this; }
    __device__ __forceinline__ Flag &operator=(const Flag &o) { return *this = (bool)o; }
    __device__ __forceinline__ Flag &operator|=(bool v) { w = v ? (w | m) : w; return *this; }
};

enum Phase : int {
    PH_OUTER_BEGIN = 0, // start an outer step: next trip evaluates f0 = f(y0)            radau.py:303
    PH_INIT_STEP,       // f0 known: first half of select_initial_step (no evaluation needed)  common.py:111-119
    PH_F1,              // next evaluates f(y0 + h0 f0)                                     common.py:120-122
    PH_STEP_BEGIN,      // _step_impl prologue (no evaluation needed)                       radau.py:399-424
    PH_ATTEMPT,         // (re)start an attempt with the current h_abs                      radau.py:426-448
    PH_NEWTON,          // one simplified-Newton iteration per trip (three evaluations)     radau.py:84-134
    PH_ERR_REFINE,      // second error estimate after a rejection (one evaluation)         radau.py:485-487
    PH_FNEW,            // step accepted: evaluate f(y_new), bookkeeping                     radau.py:500-539
    PH_DONE             // solve finished, failed or raised: wait for the wavefront's other reactors
};

struct SolverCounters { int nfev, njev, nlu, nsteps, nrej; };

// lane -> (segment, zone) geometry; the same for every work item of a wavefront
__device__ __forceinline__ void lane_geometry(int n, Lane &L)
{
    const int lane = threadIdx.x & 63;
    const int seg = lane / n;
    L.n = n; L.z = lane - seg * n;
    L.base = seg * n;
    L.a_me = lane << 3; L.xrow = nullptr;
    L.has_lo = L.z > 0; L.has_hi = L.z < n - 1;
#pragma unroll
    for (int l = 0; l < 7; ++l) {
        L.m_lo[l] = (L.z - (1 << l) >= 0) ? ~0u : 0u;
        L.m_hi[l] = (L.z + (1 << l) < n) ? ~0u : 0u;
        // opaque to the optimiser, or `x & mask` is canonicalised back into a select on the compare
        asm("" : "+v"(L.m_lo[l]));
        asm("" : "+v"(L.m_hi[l]));
    }
    L.segmask = ((n >= 64) ? ~0ull : ((1ull << n) - 1ull)) << L.base;
    {
        int top = 1;
        while (2 * top < n) top *= 2;                     // 2^(LV-1): the top stride
        const bool up = L.z >= top, has = up || (L.z + top < n);
        L.m_pt = has ? ~0u : 0u;
        asm("" : "+v"(L.m_pt));
    }
    L.d3n = {(double)(3 * n), 1.0 / (double)(3 * n)};
    L.d9n = {(double)(9 * n), 1.0 / (double)(9 * n)};
}

// rhs_kernel / selftest: one wavefront-group per workgroup
__device__ __forceinline__ bool lane_setup(int64_t r0, int64_t r1, int n, int R, Lane &L, int64_t &r)
{
    lane_geometry(n, L);
    const int seg = (threadIdx.x & 63) / n;
    r = r0 + (int64_t)blockIdx.x * R + seg;
    return (seg < R) && (r < r1);
}

// LDS of one wavefront, ONE array: [reactor constants | history base | factor store, reused between outer steps as StepIO]
template <int LV> struct LdsMap {
    static constexpr int RK_DOUBLES = RK_UNI * rk_maxr(LV) + rk_lane_doubles(LV);
    static constexpr int HIST_DOUBLES = rk_maxr(LV);                      // 2 x rk_maxr ints: history base, reactor index of each segment
    static constexpr int F_DOUBLES = FSlots<LV>::LDS_SLOTS * 64;
    static constexpr int IO_DOUBLES = (int)((sizeof(wts::StepIO) + 7) / 8);
    static constexpr int TAIL_DOUBLES = F_DOUBLES > IO_DOUBLES ? F_DOUBLES : IO_DOUBLES;
    // the exchange row of the ROW = false kernels (both(), from_partner()): 64 cells between the reactor constants and
    // the factor store.  A lane whose neighbour lies outside the wavefront reads up to 2^(LV-2) cells beyond either
    // end -- constants or factors of this same allocation, masked by the caller like every out-of-segment value.
    static constexpr int X_OFF = RK_DOUBLES + HIST_DOUBLES, X_DOUBLES = 64;
    static constexpr int F_OFF = (X_OFF + X_DOUBLES + 1) & ~1;       // 16-byte aligned: the factor store's cells are pairs
    static constexpr int TOTAL = F_OFF + TAIL_DOUBLES;
    static_assert(X_OFF >= (LV >= 2 ? (1 << (LV - 2)) : 0), "reads below the exchange row must stay inside the allocation");
};

// The argument block has ~80 pointers; held in SGPRs across the solver loop they would crowd out the loop's own
// scalars (the compiler hoists kernel-argument loads to the top of the kernel and then spills them).  Each section
// of a work item therefore re-reads what it needs from the kernel-argument segment through a pointer the
// optimiser cannot see through, which ends the live ranges at the section's end.
typedef const __attribute__((address_space(4))) StepArgs *ArgPtr;
__device__ __forceinline__ ArgPtr fresh(ArgPtr p)
{
    asm volatile("" : "+s"(p));
    return p;
}

// ---- device-side work queue: FIFO of wavefront-groups that are ready for their next item (wave-uniform calls) ----
// Tickets 0 .. n_groups-1 are the groups themselves, last group first (every group starts ready; the slots are dealt in
// order of solver cost, so the expensive groups are the ones that must not start late); ticket n_groups + p is the
// p-th push.
// Q_AVAIL counts published, unclaimed entries, so a claimed ticket is always (about to be) written: the only wait
// is for a pusher that sits between its tail increment and its slot store.
__device__ __forceinline__ int queue_resolve(ArgPtr a, int ticket)
{
    const int n_groups = a->n_groups;
    if (ticket < n_groups) return n_groups - 1 - ticket;
    const unsigned long long want = (unsigned long long)(unsigned)(ticket + 1);
    unsigned long long *slot = a->q_slots + (ticket - n_groups) % a->q_cap;
    for (int spin = 0; spin < (1 << 22); ++spin) {
        const unsigned long long w = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((w >> 32) == want) return (int)(w & 0xffffffffull);
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(a->q_ctrl + Q_ERROR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // never seen; the host reports it
    return -1;
}

__device__ __forceinline__ void queue_push(ArgPtr a, int group)
{
    int32_t *ctrl = a->q_ctrl;
    const int p = __hip_atomic_fetch_add(ctrl + Q_TAIL, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w = ((unsigned long long)(unsigned)(a->n_groups + p + 1) << 32) | (unsigned)group;
    __hip_atomic_store(a->q_slots + p % a->q_cap, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(ctrl + Q_AVAIL, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// Next group for this worker, or -1 to retire.  own >= 0: the group just advanced still has steps to go; it goes to
// the back of the queue if another group is waiting (rotation: with more groups than resident wavefronts nobody
// idles) -- unless `hold`: the group is behind the ensemble's average progress (an expensive group: the same worker
// time buys it fewer steps) and keeps its worker until it has caught up, so that all groups finish together instead of
// the expensive ones trailing at the end of the launch.  Otherwise this worker simply carries on with it.  Only an
// exchange hands data to another CU, so only then the wavefront releases what it wrote (and the taker acquires).
// (Tried and dropped: letting groups whose items run long keep their worker, and dealing last launch's slow groups
// first -- a group's cost comes in bursts when a reactor crosses a stratification switch, not as a persistent rate,
// so neither shortens the tail of a short launch; see DESIGN.md.)
__device__ __forceinline__ int queue_next(ArgPtr pa, int own, bool hold, bool &exchanged)
{
    ArgPtr a = fresh(pa);
    const bool lane0 = (threadIdx.x & 63) == 0;
    int ticket = -1;
    if (lane0 && !(own >= 0 && hold)) {
        // Claim one published entry.  A failed claim takes Q_AVAIL below its true value until it is restored, which
        // can make a concurrent claimer fail although an entry has just been published; so whoever fails looks again
        // after restoring: the last of the failed claimers to restore sees the true count.  (One atomic per claim in
        // the common case; a compare-and-swap loop here costs O(workers^2) atomics when a launch starts.)
        int32_t *avail = a->q_ctrl + Q_AVAIL;
        do {
            const int old = __hip_atomic_fetch_add(avail, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old > 0) { ticket = __hip_atomic_fetch_add(a->q_ctrl + Q_HEAD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __hip_atomic_fetch_add(avail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (__hip_atomic_load(avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0);
    }
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    exchanged = ticket >= 0;
    if (ticket < 0) return own;                 // nothing waiting (or holding on): carry on with the own group, or retire
    if (own >= 0) {
        // publish the group's state before anybody can take its next item
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    int next = -1;
    if (lane0) {
        if (own >= 0) queue_push(a, own);
        next = queue_resolve(a, ticket);
    }
    return __builtin_amdgcn_readfirstlane(next);
}

// derivatives() at the NS points of one trip, section by section: pH properties of all points, temperature properties
// of all points, then the rows.  Each section fetches its own fp64 constants (scalar loads from the argument block)
// and its own share of the reactor constants (LDS), so neither is live outside it, and inside a section the NS
// evaluations are independent chains for the scheduler to interleave.
template <bool ROW, int NS>
__device__ __forceinline__ void rhs_points(const Lane &L, const RKStore &ks, ArgPtr pa, const double (*y)[3], double (*F)[3], bool *bad)
{
    PropPH pp[NS]; PropT pt[NS];
    {
        ArgPtr a = fresh(pa);
        const KP c = load_kp(&a->kt);
        const RK k = fetch_reactor(ks);
        double x[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) x[s] = y[s][SPH];
        prop_pH_n<NS>(c, k, x, pp);
    }
    __builtin_amdgcn_sched_barrier(0);   // the next section's constants are fetched when this one is through (SGPR budget)
    {
        ArgPtr a = fresh(pa);
        const KT c = load_kt(&a->kt);
        double x[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) x[s] = y[s][STT];
        prop_T_n<NS>(c, x, pt);
#pragma unroll
        for (int s = 0; s < NS; ++s) bad[s] = pt[s].bad;
    }
    {
        const RK k = fetch_reactor(ks);
#pragma unroll
        for (int s = 0; s < NS; ++s)
            rhs_rows<ROW>(L, k, pp[s].H, pp[s].iw, pp[s].bpos, pt[s].kT * pp[s].phi, pt[s].rho, y[s][SCL], y[s][STT], F[s]);
    }
}

// One work item: the reactors of wavefront-group `group` advanced by `cnt` outer steps, starting with step `step0`
// of this launch.  Reactors of a wavefront start every outer step together (they wait for the slowest of them), so
// their Jacobians, factorisations and Newton trips coincide, and the end of an outer step is a wave-uniform point
// where the sensor suite and the PLC scan run.
template <int LV, bool ROW>
__device__ __forceinline__ void run_item(ArgPtr pa, const Lane &L, double *lds, int group, int step0, int cnt)
{
    using M = LdsMap<LV>;
    ArgPtr a = fresh(pa);                             // ---- section: load the group
    const int n_zones = L.n, R = a->R;
    const int lane = threadIdx.x & 63, seg = lane / n_zones;
    const int64_t q_first = (int64_t)group * R;       // slots of this group; slot q holds reactor perm[q]
    const int64_t q_end = a->q_ctrl ? a->N : a->r1;
    const bool present = (seg < R) && (q_first + seg < q_end);
    const int64_t r = present ? (int64_t)a->perm[q_first + seg] : 0;
    const int64_t idx = r * n_zones + L.z;
    const double dt = a->dt;
    const int step_limit = a->step_limit, sens_on = a->sens.on, plc_on = a->sens.plc_on;
    const bool want_diag = a->wave_diag != nullptr;
    const RKStore ks = {lds + seg, lds + RK_UNI * rk_maxr(LV) + lane, rk_maxr(LV), L.m_lo[0], L.m_hi[0], rk_lane_in_lds(LV)};
    int *hist0 = reinterpret_cast<int *>(lds + M::RK_DOUBLES);
    int *rix = hist0 + rk_maxr(LV);                   // reactor index of each segment, for the sensor / PLC lanes
    double *lds_factors = lds + M::F_OFF;
    wts::StepIO &io = *reinterpret_cast<wts::StepIO *>(lds_factors);

    // ---- per-reactor state carried from one outer step to the next (segment-uniform scalars replicated in every lane)
    double y0[3] = {7.0, 1.0, 20.0};                  // state at the start of the outer step
    double f[3] = {0, 0, 0};                          // f(y0) when f_valid
    double t_out = 0;                                 // ReactorState.time
    double dH = 0, dR = 0, dK = 0, badval = 0;
    double flow_used = 0;                             // ReactorState.flow_rate: the flows of the last step taken
    uint32_t st = 0;
    bool frozen = !present, f_valid = false, wrote_k = false, raised = false;
    int steps_done = 0, reads_done = 0, cost_acc = 0;
    SolverCounters last_cnt = {0, 0, 0, 0, 0};
    int diag_trips = 0, diag_newton = 0;              // per item: 32 bits are plenty
#ifdef WT_STAMPS  // block-execution counters cost a ballot and a branch each per trip: diagnostic builds only
    int diag_fact = 0, diag_jac = 0, diag_f3 = 0;
#define WT_COUNT(c) ++(c)
#else
    constexpr int diag_fact = 0, diag_jac = 0, diag_f3 = 0;
#define WT_COUNT(c) do { } while (0)
#endif
#ifdef WT_STAMPS  // diagnostic build only: shader-clock shares of the loop's sections (never in the product .so)
    long long sec[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tprev = __builtin_amdgcn_s_memtime();
#define WT_STAMP(i) do { const long long tn_ = __builtin_amdgcn_s_memtime(); sec[i] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define WT_STAMP(i) do { } while (0)
#endif
    const long long clk0 = want_diag ? __builtin_amdgcn_s_memtime() : 0, wall0 = want_diag ? __builtin_amdgcn_s_memrealtime() : 0;
    if (present) {
        st = a->status[r];
        // a reactor whose last step raised stays frozen until the host rewrites its state
        if (st & (ST_T_RANGE | ST_T_RANGE_POST)) frozen = true;
        y0[SPH] = a->pH[idx]; y0[SCL] = a->Cl[idx]; y0[STT] = a->T[idx];
        t_out = a->time[r];
        RK k0; load_reactor(a->par, a->bc, a->N, r, n_zones, k0); mask_reactor_for_lane(L, k0);
        park_reactor(ks, k0);
        if (sens_on && L.z == 0) hist0[seg] = a->sens.hist_value ? a->sens.hist_pos[r] : 0;
        if (L.z == 0) rix[seg] = (int)r;
    }

    for (int k = 0; k < cnt; ++k) {
        bool stepped = false;
        if (!frozen) {
          // scipy refuses a non-finite initial state: ValueError out of step(), self.state untouched (base.py:19-20)
          if (WT_RARE(seg_any(L, !(isfinite(y0[0]) && isfinite(y0[1]) && isfinite(y0[2]))))) { st |= ST_NONFINITE; frozen = true; }
          else {
            // ================= one IntegratedCSTR.step(): a fresh scipy solver object (reactor.py:476)
            double yc[3], W[3][3];                    // solver's current y; Newton iterate in transformed variables
            double aux[3] = {0, 0, 0};                // y0 + h0 f0 (initial step) / error vector (refinement)
            constexpr bool HELD = LV <= 5;           // (see Held; the n = 17...32 kernel holds all but the Jacobian: +0.7 %)
            typedef Held<HELD> AReg64;
            // dense output of the last accepted step: written when a step is accepted, read when the next attempt is set up
            AReg64 Qa[3][3], y_old_a[3], sol_t_old_a, sol_h_a;
#pragma unroll
            for (int q = 0; q < 3; ++q) { y_old_a[q].init(); Qa[q][0].init(); Qa[q][1].init(); Qa[q][2].init(); }
            sol_t_old_a.init(); sol_h_a.init();
            Jac J;                                    // num_jac's output; between its uses the Jacobian lives in `ja`
            JacA<(LV <= 4)> ja; ja.init();
            FStore<LV> F;
            F.cell = (LdsDouble2 *)lds_factors + lane;
            uint32_t fl = 1u << 4;                    // current_jac = true
            Flag have_fac{fl, 1u << 0}, have_old{fl, 1u << 1}, have_old_l{fl, 1u << 2}, have_sol{fl, 1u << 3}, current_jac{fl, 1u << 4},
                 have_lu{fl, 1u << 5}, rejected{fl, 1u << 6}, keep_h{fl, 1u << 7}, have_norm_old{fl, 1u << 8}, have_rate{fl, 1u << 9},
                 bad{fl, 1u << 10}, failed{fl, 1u << 11}, fv{fl, 1u << 12}, need_jac{fl, 1u << 13},
                 limit_hit{fl, 1u << 16}, pend_f{fl, 1u << 17}, jac_after_fnew{fl, 1u << 18},
                 j_dense{fl, 1u << 19};               // this lane's Jacobian couples a row to a neighbour's temperature
            fv = f_valid;
            AReg64 fac_a[3]; fac_a[0].init(); fac_a[1].init(); fac_a[2].init();   // num_jac's factors: touched once per Jacobian
            double t = t_out, t_bound = t_out + dt, max_step = fmin(dt, 10.0);
            double h = 0, t_new = 0, h_abs = 0, h_abs_l = 0, min_step = 0;
            // the step-size controller's memory: written when a step is accepted / begun, read when the next one is judged
            AReg64 h_abs_old_a, err_old_a, h_abs_old_l_a, err_old_l_a;
            h_abs_old_a.init(); err_old_a.init(); h_abs_old_l_a.init(); err_old_l_a.init();
            int kk = 0, n_iter = 0; double dW_norm_old = 0, rate = 0;
            double error_norm = 0, safety = 0;
            double d0 = 0, d1 = 0, h0 = 0;            // select_initial_step
            SolverCounters cnt_s = {0, 0, 0, 0, 0};
            int attempts = 0;   // guard against unbounded solves (sliding along a discontinuity): see limit_hit
            int badstage = 0;   // which evaluation of the trip raised: 0 deferred f(y_new), 1..3 stage / single point
            // pend_f: f(yc) of the last accepted step has not been evaluated yet
            // jac_after_fnew: that step also asked for a fresh Jacobian (radau.py:500,512)
            int phase = PH_OUTER_BEGIN;
#pragma unroll
            for (int q = 0; q < 3; ++q) { yc[q] = y0[q]; W[0][q] = W[1][q] = W[2][q] = 0.0; }

            // select_initial_step (common.py:68-134), order 3, up to the probe point y0 + h0 f0
            auto initial_step_first_half = [&]() {
                double sc[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) sc[q] = ATOL + fabs(yc[q]) * RTOL;
                d0 = rms3<ROW, LV>(L, yc, sc); d1 = rms3<ROW, LV>(L, f, sc);
                h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 * rcp(d1);
                h0 = fmin(h0, fabs(t_bound - t));
#pragma unroll
                for (int q = 0; q < 3; ++q) aux[q] = yc[q] + h0 * f[q];
            };
            // error_norm > 1: radau.py:489-496
            auto reject_step = [&]() {
                const double fct = predict_factor(h_abs_l, have_old_l, h_abs_old_l_a.get(), error_norm, err_old_l_a.get());
                h_abs_l *= fmax(MIN_FACTOR, safety * fct);
                have_lu = false; rejected = true; cnt_s.nrej++;
                phase = PH_ATTEMPT;
            };
            // step accepted: radau.py:500-539.  scipy evaluates f(y_new) right here; the value is first
            // needed by the next error estimate, so unless a Jacobian refresh or the end of the outer
            // step needs it at once, it is evaluated together with the next Newton trip (pend_f).
            auto accept_step = [&]() {
                const bool recompute_jac = (n_iter > 2) && have_rate && (rate > 1e-3);
                double fct = predict_factor(h_abs_l, have_old_l, h_abs_old_l_a.get(), error_norm, err_old_l_a.get());
                fct = fmin(MAX_FACTOR, safety * fct);
                if (!recompute_jac && fct < 1.2) fct = 1.0; else have_lu = false;
                h_abs_old_a.set(h_abs);       // sic radau.py:520: the solver-level value
                err_old_a.set(error_norm);
                have_old = true;
                h_abs = h_abs_l * fct;
                const KZ kz = lit_kz(); const KA ka = lit_ka();
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const double z0 = kz.T00 * W[0][q] + kz.T01 * W[1][q] + kz.T02 * W[2][q];
                    const double z1 = kz.T10 * W[0][q] + kz.T11 * W[1][q] + kz.T12 * W[2][q];
                    const double z2 = W[0][q] + W[1][q];
                    y_old_a[q].set(yc[q]);
                    Qa[q][0].set(z0 * ka.P[0] + z1 * ka.P[3] + z2 * ka.P[6]);  // Q = Z^T P  radau.py:541-543
                    Qa[q][1].set(z0 * ka.P[1] + z1 * ka.P[4] + z2 * ka.P[7]);
                    Qa[q][2].set(z0 * ka.P[2] + z1 * ka.P[5] + z2 * ka.P[8]);
                    yc[q] = yc[q] + z2;
                }
                sol_t_old_a.set(t); sol_h_a.set(t_new - t); have_sol = true;
                t = t_new;
                cnt_s.nsteps++; cnt_s.nfev++;     // f(y_new) counted where scipy calls it
                pend_f = true; fv = false;
                current_jac = recompute_jac;
                const bool more = (t - t_bound) < 0;
                if (recompute_jac || !more) { jac_after_fnew = recompute_jac; phase = PH_FNEW; }
                else phase = PH_STEP_BEGIN;
            };

            if (fv) {
                // f(y0) is already in f (last evaluation of the previous outer step, same y, same
                // boundary): count it as scipy does and go straight to the initial-step probe
                cnt_s.nfev++;
                phase = PH_INIT_STEP;
            }

            WT_STAMP(0);   // item / outer-step set-up
            while (true) {
                // reactors that finished their outer step wait here until every reactor of the wavefront has
                if (__ballot(phase != PH_DONE) == 0ull) break;
                // ================= trips that need no RHS evaluation (run first so the lane can join this trip's evaluation)
                if (phase == PH_INIT_STEP) {      // the one copy of this arithmetic, whether f0 was evaluated or carried over
                    initial_step_first_half();
                    phase = PH_F1;
                }
                if (phase == PH_STEP_BEGIN) {
                    min_step = 10 * fabs(ulp_above(t));                      // radau.py:408
                    if (h_abs > max_step) { h_abs_l = max_step; have_old_l = false; }
                    else if (h_abs < min_step) { h_abs_l = min_step; have_old_l = false; }
                    else { h_abs_l = h_abs; have_old_l = have_old; h_abs_old_l_a.set(h_abs_old_a.get()); err_old_l_a.set(err_old_a.get()); }
                    rejected = false; keep_h = false;
                    phase = PH_ATTEMPT;
                }
                if (phase == PH_ATTEMPT) {
                    if (!keep_h) {
                        if (WT_RARE(step_limit > 0 && attempts >= step_limit)) { failed = true; limit_hit = true; phase = PH_DONE; }
                        else if (WT_RARE(h_abs_l < min_step)) { failed = true; phase = PH_DONE; }  // radau.py:427-428
                        else {
                            attempts++;
                            h = h_abs_l;
                            t_new = t + h;
                            if (t_new - t_bound > 0) t_new = t_bound;
                            h = t_new - t;
                            h_abs_l = fabs(h);
                        }
                    }
                    if (phase == PH_ATTEMPT) {
                        keep_h = false;
                        // initial guess Z0 (radau.py:445-448,557-572) and W = TI Z0 (radau.py:88)
                        double Z0[3][3];
                        if (!have_sol) {
#pragma unroll
                            for (int s = 0; s < 3; ++s)
#pragma unroll
                                for (int q = 0; q < 3; ++q) Z0[s][q] = 0.0;
                        } else {
                            const double sol_t_old = sol_t_old_a.get(), isol = rcp(sol_h_a.get());
                            double Q[3][3], y_old[3];
#pragma unroll
                            for (int q = 0; q < 3; ++q) { y_old[q] = y_old_a[q].get(); Q[q][0] = Qa[q][0].get(); Q[q][1] = Qa[q][1].get(); Q[q][2] = Qa[q][2].get(); }
                            const KG kg = load_kg(&fresh(pa)->rt);
                            const double cs[3] = {kg.C0, kg.C1, 1.0};
#pragma unroll
                            for (int s = 0; s < 3; ++s) {
                                const double x = ((t + h * cs[s]) - sol_t_old) * isol;
                                const double p1 = x * x, p2 = p1 * x;
#pragma unroll
                                for (int q = 0; q < 3; ++q)
                                    // (which products fuse is spelled out: the polynomial's rounding steers Newton's start)
                                    Z0[s][q] = (__builtin_fma(Q[q][2], p2, __builtin_fma(Q[q][0], x, Q[q][1] * p1)) + y_old[q]) - yc[q];
                            }
                        }
                        const KN kn0 = load_kn(&fresh(pa)->rt);
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            W[0][q] = kn0.TI[0] * Z0[0][q] + kn0.TI[1] * Z0[1][q] + kn0.TI[2] * Z0[2][q];
                            W[1][q] = kn0.TI[3] * Z0[0][q] + kn0.TI[4] * Z0[1][q] + kn0.TI[5] * Z0[2][q];
                            W[2][q] = kn0.TI[6] * Z0[0][q] + kn0.TI[7] * Z0[1][q] + kn0.TI[8] * Z0[2][q];
                        }
                        kk = 0; have_norm_old = false; have_rate = false; rate = 0.0;
                        phase = PH_NEWTON;
                    }
                }
                WT_STAMP(1);   // step / attempt prologues
#ifdef WT_STAMPS
                if (__ballot(phase == PH_NEWTON && !have_lu) != 0ull) WT_COUNT(diag_fact);
#endif
                if (phase == PH_NEWTON && !have_lu) {
                    { Jac Jb; ja.bands(Jb); factorize<ROW, LV>(L, Jb, h, F); } have_lu = true; cnt_s.nlu += 2;      // radau.py:454-456
                }

                WT_STAMP(2);   // factorisation
                // ================= this trip's evaluation points
                const bool newton = (phase == PH_NEWTON);
                diag_trips++; if (__ballot(newton) != 0ull) diag_newton++;
                const bool eval0 = (phase == PH_OUTER_BEGIN || phase == PH_F1 || phase == PH_ERR_REFINE || phase == PH_FNEW || newton);
                // f(y) of a just-accepted step rides along with the next attempt's first Newton trip
                const bool eval3 = newton && pend_f;
                double ye[3][3], Fe[3][3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    double p0 = yc[q];                                        // PH_OUTER_BEGIN, PH_FNEW
                    if (phase == PH_F1) p0 = aux[q];
                    if (phase == PH_ERR_REFINE) p0 = yc[q] + aux[q];
                    ye[0][q] = p0;
                }
                // Z = T W (radau.py:124): Z[2] = W0 + W1 -- the stage points
                auto stage_points = [&]() {
                    const KZ kzp = load_kz(&fresh(pa)->rt);
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const double z0 = kzp.T00 * W[0][q] + kzp.T01 * W[1][q] + kzp.T02 * W[2][q];
                        const double z1 = kzp.T10 * W[0][q] + kzp.T11 * W[1][q] + kzp.T12 * W[2][q];
                        const double z2 = W[0][q] + W[1][q];
                        if (newton) ye[0][q] = yc[q] + z0;
                        ye[1][q] = yc[q] + z1; ye[2][q] = yc[q] + z2;
                    }
                };
                // formed only on trips that evaluate them (+1 % at n = 8) -- except in the n = 17...32 kernel, where the
                // extra block measures 0.6 % slower (and, before its solver state was restructured, cost a private segment)
                if constexpr (LV >= 5) stage_points();
                bool b0 = false, b1 = false, b2 = false, b3 = false;
                if (__ballot(newton) != 0ull) {
                    if constexpr (LV < 5) stage_points();
                    // some reactor of the wavefront is in its Newton phase: all three stage points (three independent
                    // chains for the scheduler to interleave); the other lanes' slot-1/2 results are simply not used
                    bool bb[3];
                    rhs_points<ROW, 3>(L, ks, pa, ye, Fe, bb);
                    b0 = bb[0] && eval0; b1 = bb[1] && newton; b2 = bb[2] && newton;
                    if (eval0 && phase != PH_FNEW) cnt_s.nfev++;
                    if (newton) cnt_s.nfev += 2;
                } else if (__ballot(eval0) != 0ull) {
                    bool bb[1];
                    rhs_points<ROW, 1>(L, ks, pa, ye, Fe, bb);
                    b0 = bb[0] && eval0;
                    if (eval0 && phase != PH_FNEW) cnt_s.nfev++;
                }
                if (__ballot(eval3) != 0ull) {
                    WT_COUNT(diag_f3);
                    double fy[1][3]; bool bb[1];
                    rhs_points<ROW, 1>(L, ks, pa, &yc, fy, bb);
                    b3 = bb[0] && eval3;
                    if (eval3) {
                        pend_f = false;       // (counted in nfev when the step was accepted)
#pragma unroll
                        for (int q = 0; q < 3; ++q) f[q] = fy[0][q];
                    }
                }
                if (WT_RARE(__ballot(b0 || b1 || b2 || b3) != 0ull)) {   // rare: a zone temperature outside [0, 100] C
                    // the reference raises in the first evaluation, at the first zone, that sees it: scipy calls
                    // f(y_new) of the accepted step before the stages of the next Newton iteration
                    const bool mine = b0 || b1 || b2 || b3;
                    if (mine && !bad) {
                        badstage = b3 ? 0 : (b0 ? 1 : (b1 ? 2 : 3));
                        badval = b3 ? yc[STT] : (b0 ? ye[0][STT] : (b1 ? ye[1][STT] : ye[2][STT]));
                    }
                    bad |= mine;
                    if (seg_any(L, bad)) { raised = true; phase = PH_DONE; }
                }

                WT_STAMP(3);   // RHS evaluations
                // ================= per-phase epilogues
                if (WT_RARE(phase == PH_OUTER_BEGIN)) {   // (rare: f(y0) is usually carried over from the previous outer step)
#pragma unroll
                    for (int q = 0; q < 3; ++q) f[q] = Fe[0][q];
                    phase = PH_INIT_STEP;
                } else if (phase == PH_F1) {
                    double sc[3], df[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) { sc[q] = ATOL + fabs(yc[q]) * RTOL; df[q] = Fe[0][q] - f[q]; }
                    const double d2 = rms3<ROW, LV>(L, df, sc) * rcp(h0);
                    double h1;
                    if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
                    else h1 = root4(0.01 * rcp(fmax(d1, d2)));
                    h_abs = fmin(fmin(100 * h0, h1), fmin(fabs(t_bound - t), max_step));
                    need_jac = true;                                          // radau.py:359-365
                    phase = PH_STEP_BEGIN;
                } else if (phase == PH_NEWTON) {
                    // ---- one iteration of solve_collocation_system radau.py:84-134
                    Jac Jc; ja.coupling(Jc);
                    bool finite = true;
#pragma unroll
                    for (int s = 0; s < 3; ++s)
#pragma unroll
                        for (int q = 0; q < 3; ++q) finite = finite && isfinite(Fe[s][q]);
                    bool conv = false, diverged = false;
                    if (WT_RARE(!seg_all(L, finite))) {
                        diverged = true;
                    } else {
                        const double ih = rcp(h);
                        const RTabPtr rt = &fresh(pa)->rt;
                        const KN kn = load_kn(rt); const KZ kz = load_kz(rt);
                        const double M_real = kn.mu_r * ih, Mcr = kn.mu_cr * ih, Mci = kn.mu_ci * ih;
                        double fr[3], fcr[3], fci[3], scale[3];
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            scale[q] = kz.atol + fabs(yc[q]) * kz.rtol;
                            fr[q] = (Fe[0][q] * kn.TI[0] + Fe[1][q] * kn.TI[1] + Fe[2][q] * kn.TI[2]) - M_real * W[0][q];
                            const double re = Fe[0][q] * kn.TI[3] + Fe[1][q] * kn.TI[4] + Fe[2][q] * kn.TI[5];
                            const double im = Fe[0][q] * kn.TI[6] + Fe[1][q] * kn.TI[7] + Fe[2][q] * kn.TI[8];
                            fcr[q] = re - (Mcr * W[1][q] - Mci * W[2][q]);
                            fci[q] = im - (Mcr * W[2][q] + Mci * W[1][q]);
                        }
                        solve_rc<ROW, LV>(L, Jc, F, fr, fcr, fci, __ballot(j_dense) == 0ull);
                        double ssum = 0.0;
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            const double is = rcp(scale[q]);
                            const double u = fr[q] * is, v = fcr[q] * is, w = fci[q] * is;
                            ssum += u * u + v * v + w * w;
                        }
                        const double dW_norm = sqrt_k(div_by(seg_sum<ROW, LV>(L, ssum), L.d9n));
                        if (have_norm_old) { rate = dW_norm * rcp(dW_norm_old); have_rate = true; }
                        const double i1r = rcp(1 - rate);
                        if (have_rate && (rate >= 1 || powi6(rate, NEWTON_MAXITER - kk) * i1r * dW_norm > kn.newton_tol)) {
                            diverged = true;
                        } else {
#pragma unroll
                            for (int q = 0; q < 3; ++q) { W[0][q] += fr[q]; W[1][q] += fcr[q]; W[2][q] += fci[q]; }
                            if (dW_norm == 0 || (have_rate && rate * i1r * dW_norm < kn.newton_tol)) conv = true;
                            dW_norm_old = dW_norm; have_norm_old = true;
                        }
                    }
                    n_iter = kk + 1;
                    kk++;
                    if (!conv && !diverged && kk == NEWTON_MAXITER) diverged = true;   // loop ran out: radau.py:136
                    if (WT_RARE(diverged)) {                                          // radau.py:462-476
                        if (current_jac) { h_abs_l *= 0.5; have_lu = false; cnt_s.nrej++; phase = PH_ATTEMPT; }
                        else { need_jac = true; current_jac = true; have_lu = false; keep_h = true; phase = PH_ATTEMPT; }
                    } else if (conv) {
                        // ---- error estimate radau.py:477-487
                        double err[3], esc[3];
                        const double ih_e = rcp(h);
                        const KZ kze = lit_kz(); const KE ke = lit_ke();
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            const double z0 = kze.T00 * W[0][q] + kze.T01 * W[1][q] + kze.T02 * W[2][q];
                            const double z1 = kze.T10 * W[0][q] + kze.T11 * W[1][q] + kze.T12 * W[2][q];
                            const double z2 = W[0][q] + W[1][q];
                            const double ZE = (z0 * ke.E0 + z1 * ke.E1 + z2 * ke.E2) * ih_e;
                            err[q] = f[q] + ZE;
                            esc[q] = kze.atol + fmax(fabs(yc[q]), fabs(yc[q] + z2)) * kze.rtol;
                        }
                        solve_real<ROW, LV>(L, Jc, F, err, __ballot(j_dense) == 0ull);
                        error_norm = rms3<ROW, LV>(L, err, esc);
                        safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + n_iter);
                        if (WT_RARE(rejected && error_norm > 1)) {
#pragma unroll
                            for (int q = 0; q < 3; ++q) aux[q] = err[q];
                            phase = PH_ERR_REFINE;
                        } else if (WT_RARE(error_norm > 1)) {                         // radau.py:489-496
                            reject_step();
                        } else {
                            accept_step();
                        }
                    }
                } else if (WT_RARE(phase == PH_ERR_REFINE)) {
                    Jac Jc; ja.coupling(Jc);
                    double err[3], esc[3];
                    const double ih_e = rcp(h);
                    const KZ kze = lit_kz(); const KE ke = lit_ke();
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const double z0 = kze.T00 * W[0][q] + kze.T01 * W[1][q] + kze.T02 * W[2][q];
                        const double z1 = kze.T10 * W[0][q] + kze.T11 * W[1][q] + kze.T12 * W[2][q];
                        const double z2 = W[0][q] + W[1][q];
                        const double ZE = (z0 * ke.E0 + z1 * ke.E1 + z2 * ke.E2) * ih_e;
                        err[q] = Fe[0][q] + ZE;
                        esc[q] = kze.atol + fmax(fabs(yc[q]), fabs(yc[q] + z2)) * kze.rtol;
                    }
                    solve_real<ROW, LV>(L, Jc, F, err, __ballot(j_dense) == 0ull);
                    error_norm = rms3<ROW, LV>(L, err, esc);
                    if (error_norm > 1) reject_step(); else accept_step();
                } else if (phase == PH_FNEW) {
                    // f(y_new) of an accepted step that needs it before anything else can happen:
                    // Jacobian refresh (radau.py:512-514) or the end of the outer step
#pragma unroll
                    for (int q = 0; q < 3; ++q) f[q] = Fe[0][q];
                    pend_f = false;
                    fv = true;
                    if (jac_after_fnew) { need_jac = true; jac_after_fnew = false; }
                    phase = ((t - t_bound) < 0) ? PH_STEP_BEGIN : PH_DONE;
                }

                WT_STAMP(4);   // epilogues (Newton solve, error estimate, accept / reject)
                // ================= finite-difference Jacobian at (yc, f) when a phase asked for it
#ifdef WT_STAMPS
                if (__ballot(need_jac) != 0ull) WT_COUNT(diag_jac);
#endif
                if (need_jac) {
                    bool jbad = false, hf = have_fac, jd = false; double jval = 0;
                    asm volatile("" ::: "memory");               // a fresh fetch: do not keep the constants live across the epilogue
                    double fac[3] = {fac_a[0].get(), fac_a[1].get(), fac_a[2].get()};
                    num_jac<ROW>(L, ks, [&]() { return &fresh(pa)->kt; }, yc, f, fac, hf, J, jbad, jval, jd); cnt_s.njev++;
                    fac_a[0].set(fac[0]); fac_a[1].set(fac[1]); fac_a[2].set(fac[2]);
                    have_fac = hf;
                    j_dense = jd;
                    ja.put(J);
                    need_jac = false;
                    if (WT_RARE(seg_any(L, jbad))) {
                        if (jbad && !bad) { badstage = 4; badval = jval; }
                        bad |= jbad; raised = true; phase = PH_DONE;
                    }
                }
                WT_STAMP(5);   // num_jac
            }
            last_cnt = cnt_s;
            cost_acc += cnt_s.nfev;

            // ================= after the solve: reactor.py:486-507
            if (WT_RARE(raised)) {
                // the reference raised (thermodynamics.py:146-157): self.state untouched; name the temperature its
                // message names -- first evaluation of the trip, lowest zone
                st |= ST_T_RANGE; frozen = true;
                double v = badval; int best = 1 << 30;
#pragma unroll 1
                for (int sidx = 0; sidx < 5; ++sidx) {
                    const unsigned long long m = __ballot(bad && badstage == sidx) & L.segmask;
                    if (m != 0ull && best == (1 << 30)) { best = sidx; v = __shfl(badval, (int)__builtin_ctzll(m), 64); }
                }
                badval = v;
                raised = false;
            } else {
                if (failed) st |= ST_SOLVER_FAILED;    // reactor.py:486-487; state <- last accepted y
                if (limit_hit) st |= ST_STEP_LIMIT;
#pragma unroll
                for (int q = 0; q < 3; ++q) y0[q] = yc[q];
                stepped = true; steps_done++;
                t_out = t_out + dt;                    // reactor.py:496
                flow_used = ks.uni[15 * ks.stride];    // reactor.py:497-501
                // _update_derived_state reactor.py:511-524 (before the clamp)
                double dHv; PropT pt;
                {
                    ArgPtr a2 = fresh(pa);
                    const KP cp = load_kp(&a2->kt); const KT ct = load_kt(&a2->kt);
                    dHv = exp10_k(cp, -y0[SPH]); pt = prop_T(ct, y0[STT]);
                }
                dH = dHv;
                dR = pt.rho;
                bool clamped = false;
                if (WT_RARE(seg_any(L, pt.bad))) {
                    st |= ST_T_RANGE_POST; frozen = true;
                    const unsigned long long m = __ballot(pt.bad) & L.segmask;
                    badval = __shfl(y0[STT], (int)__builtin_ctzll(m), 64);
                } else {
                    dK = pt.kT; wrote_k = true;
                    // _enforce_physical_bounds reactor.py:526-541
                    if (WT_RARE(seg_any(L, y0[SPH] < 0 || y0[SPH] > 14))) { st |= ST_CLAMP_PH; y0[SPH] = fmin(fmax(y0[SPH], 0.0), 14.0); clamped = true; }
                    if (WT_RARE(seg_any(L, y0[SCL] < 0))) { st |= ST_CLAMP_CL; y0[SCL] = fmax(y0[SCL], 0.0); clamped = true; }
                    if (WT_RARE(seg_any(L, y0[STT] < 0 || y0[STT] > 100))) { st |= ST_CLAMP_T; y0[STT] = fmin(fmax(y0[STT], 0.0), 100.0); clamped = true; }
                    // f(y) of the last accepted point is f0 of the next outer step when nothing touched y
                    f_valid = fv && !clamped && !failed;
                }
            }
          }
        }

        WT_STAMP(6);       // post-step (derived, clamps)
        // ================= what follows reactor.step() in the reference's loop body (__main__.py:403-423)
        if (sens_on) {
            ArgPtr b = fresh(pa);            // ---- section: sensors and plant I/O
            __syncthreads();                 // the factor store is dead now; the same LDS carries the hand-off
            if (seg < R) {
                const bool live = stepped && !(st & ST_T_RANGE_POST);     // the reference's loop stops where step() raises
                if (L.z == 0) {
                    io.stepped[seg] = live ? 1 : 0;
                    if (live) reads_done++;
                    io.t_after[seg] = t_out;
                    io.tap[0][seg] = (float)y0[SPH]; io.tap[2][seg] = (float)y0[SCL]; io.tap[4][seg] = (float)y0[STT];
                    io.tap[6][seg] = (float)flow_used;
                }
                if (!L.has_hi) { io.tap[1][seg] = (float)y0[SPH]; io.tap[3][seg] = (float)y0[SCL]; io.tap[5][seg] = (float)y0[STT]; }
            }
            __syncthreads();
            wts::suite_step(b->sens, io, rix, R, hist0, k);              // read_all_sensors
            if (plc_on) {
                const int gs = b->first_step + step0 + k;
                const bool scan = ((gs + 1) % b->sens.scan_every == 0) || (gs + 1 == b->call_steps);
                __syncthreads();
                if (lane < R && io.stepped[lane]) {                       // one lane per reactor
                    const int64_t rr = rix[lane];
                    const double lt = b->sens.pack.loop_time[rr];
                    if (scan) {
                        wtp::pack_inputs(b->sens.pack, rr, &io.val[0][lane], &io.fault[0][lane], wts::RMAX, lt);   // update_modbus_inputs
                        double c[3];
                        wtp::apply_commands(b->sens.cmd, rr, c);         // read_modbus_commands + apply_boundary_conditions
                        io.cmd[0][lane] = c[0]; io.cmd[1][lane] = c[1]; io.cmd[2][lane] = c[2];
                    }
                    b->sens.pack.loop_time[rr] = lt + dt;                 // sim_time += dt (__main__.py:446)
                }
                if (scan) {
                    __syncthreads();
                    if (present && io.stepped[seg]) {                    // the next step integrates under the new setpoints
                        RK k0; load_reactor(b->par, b->bc, b->N, r, n_zones, k0, &io.cmd[0][seg], wts::RMAX); mask_reactor_for_lane(L, k0);
                        park_reactor(ks, k0);
                        f_valid = false;
                    }
                }
            }
            __syncthreads();                 // hand-off read; the next step's factors may overwrite it
        }
        WT_STAMP(7);       // sensor suite, plant I/O
    }

    // ================= the item's results
    ArgPtr c = fresh(pa);
    if (present) {
        if (steps_done > 0) {
            c->pH[idx] = y0[SPH]; c->Cl[idx] = y0[SCL]; c->T[idx] = y0[STT];
            c->dH[idx] = dH; c->dRho[idx] = dR;
            if (wrote_k) c->dK[idx] = dK;
        }
        if (L.z == 0) {
            if (steps_done > 0) {
                c->time[r] = t_out;
                c->flow[r] = flow_used;
                if (c->stats) {
                    int32_t *o = c->stats + r * 5;
                    o[0] = last_cnt.nfev; o[1] = last_cnt.njev; o[2] = last_cnt.nlu; o[3] = last_cnt.nsteps; o[4] = last_cnt.nrej;
                }
                if (sens_on && c->sens.hist_value) c->sens.hist_pos[r] = hist0[seg] + reads_done;
            }
            c->status[r] = st;
            if (st & (ST_T_RANGE | ST_T_RANGE_POST)) c->bad_T[r] = badval;
            if (c->cost && cost_acc > 0) c->cost[r] += cost_acc;
        }
    }
    if (want_diag && lane == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(c->wave_diag + (int64_t)group * WT_DIAG_SLOTS);
        atomicAdd(o + 0, (unsigned long long)diag_trips); atomicAdd(o + 1, (unsigned long long)diag_newton);
        atomicAdd(o + 2, (unsigned long long)(__builtin_amdgcn_s_memtime() - clk0));
        atomicAdd(o + 3, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - wall0));
        atomicAdd(o + 4, (unsigned long long)diag_fact); atomicAdd(o + 5, (unsigned long long)diag_jac);
        atomicAdd(o + 6, (unsigned long long)diag_f3); atomicAdd(o + 7, 1ull);
#ifdef WT_STAMPS
        for (int i = 0; i < 8; ++i) atomicAdd(o + 8 + i, (unsigned long long)sec[i]);
#endif
    }
}

// Start of a queue-schedule launch: every group is ready for step 0, nothing has been pushed yet.
struct QueueResetArgs { int32_t *q_ctrl; unsigned long long *q_slots; int32_t *q_next; int n_groups, q_cap; };
__global__ __launch_bounds__(256) void queue_reset_kernel(const QueueResetArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.q_cap) a.q_slots[i] = 0ull;
    if (i < a.n_groups) a.q_next[i] = 0;
    if (i < Q_WORDS) a.q_ctrl[i] = (i == Q_AVAIL) ? a.n_groups : 0;
}

// The physics kernel.  Queue schedule: a grid of worker wavefronts that take (wavefront-group, next few outer
// steps) items from a device-side FIFO until the whole ensemble has advanced n_steps -- one launch, no launch
// tails: a slow wavefront delays nobody, and with more groups than resident wavefronts every SIMD stays busy.
// Stream schedule (q_ctrl == nullptr): workgroup b advances group r0 / R + b by n_steps and exits.
template <int LV, bool ROW>
__global__ __launch_bounds__(64) void step_kernel(const StepArgs a_unused)
{
    (void)a_unused;   // never read directly: every section fetches what it needs through `pa` (see fresh())
    __shared__ __attribute__((aligned(16))) double lds[LdsMap<LV>::TOTAL];
    // the by-value argument block sits at offset 0 of the kernel-argument segment
    const ArgPtr pa = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    Lane L;
    // ROW instantiations serve exactly one zone count (n = 2^LV): a compile-time constant for everything below
    lane_geometry(ROW ? (1 << LV) : fresh(pa)->n, L);
    if constexpr (!ROW) L.xrow = (LdsDouble *)(lds + LdsMap<LV>::X_OFF);   // cross-lane moves by strides >= 2
    bool exchanged = true;
    int group;
    {
        ArgPtr a = fresh(pa);
        group = a->q_ctrl ? queue_next(pa, -1, false, exchanged) : (int)(a->r0 / a->R) + (int)blockIdx.x;
    }
    while (group >= 0) {          // (one call site: the item body exists once in the code object)
        int step0 = 0, cnt;
        long long t0 = 0;
        bool queue;
        {
            ArgPtr a = fresh(pa);
            queue = a->q_ctrl != nullptr;
            cnt = a->n_steps;
            if (queue) {
                // taken over from another worker: its release (queue_next) -> this acquire -> plain loads of the group's state
                if (exchanged) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if ((threadIdx.x & 63) == 0) step0 = __hip_atomic_load(a->q_next + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                step0 = __builtin_amdgcn_readfirstlane(step0);
                const int left = a->n_steps - step0, item = a->item_steps;
                cnt = left < item ? left : item;
                if (a->trace) t0 = __builtin_amdgcn_s_memrealtime();
            }
        }
        run_item<LV, ROW>(pa, L, lds, group, step0, cnt);
        if (!queue) break;
        ArgPtr a = fresh(pa);
        const bool more = step0 + cnt < a->n_steps;
        int hold = 0;
        if ((threadIdx.x & 63) == 0) {
            __hip_atomic_store(a->q_next + group, step0 + cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // items this group has behind it against the ensemble's average
            const long long done = __hip_atomic_fetch_add(a->q_ctrl + Q_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            const int item = a->item_steps;
            const long long mine = (step0 + cnt + item - 1) / item;
            hold = (mine * a->n_groups < done) ? 1 : 0;     // (one item more or less of slack either way: measured worse)
            if (a->trace) {
                const int slot = __hip_atomic_fetch_add(a->q_ctrl + Q_TRACE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (slot < a->trace_cap) {
                    int64_t *o = a->trace + (int64_t)slot * 5;
                    o[0] = blockIdx.x; o[1] = group; o[2] = (int64_t)step0 | ((int64_t)cnt << 32); o[3] = t0; o[4] = __builtin_amdgcn_s_memrealtime();
                }
            }
        }
        hold = __builtin_amdgcn_readfirstlane(hold);
        group = queue_next(pa, more ? group : -1, hold != 0, exchanged);
    }
}

// After a queue-schedule launch: every group must have advanced by the launch's step count and the hand-off must
// not have timed out; anything else is recorded in a word that survives the next launch's queue reset and that every
// download path reports (a short-changed group must not pass as WT_OK).
struct QueueCheckArgs { const int32_t *q_ctrl, *q_next; int n_groups, n_steps; int32_t *sticky; };
__global__ __launch_bounds__(256) void queue_check_kernel(const QueueCheckArgs a)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    // the record is two words of host-coherent memory, each only ever set to 1: plain stores, no copy to read them
    if (g < a.n_groups && a.q_next[g] != a.n_steps) a.sticky[1] = 1;
    if (g == 0 && a.q_ctrl[Q_ERROR] != 0) a.sticky[0] = 1;
}

// One contiguous image of a small ensemble's state for a single device-to-host copy:
// [pH | Cl | T | H | rho | k] (N n doubles each), [time | flow] (N doubles each), status (N words), sticky word.
struct SnapshotArgs {
    int64_t cnt, N;
    const double *pH, *Cl, *T, *dH, *dRho, *dK, *time, *flow; const uint32_t *status; const int32_t *sticky;
    double *out;
};
__global__ __launch_bounds__(256) void snapshot_pack_kernel(const SnapshotArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.cnt) {
        a.out[i] = a.pH[i]; a.out[a.cnt + i] = a.Cl[i]; a.out[2 * a.cnt + i] = a.T[i];
        a.out[3 * a.cnt + i] = a.dH[i]; a.out[4 * a.cnt + i] = a.dRho[i]; a.out[5 * a.cnt + i] = a.dK[i];
    }
    if (i < a.N) {
        double *tail = a.out + 6 * a.cnt;
        tail[i] = a.time[i]; tail[a.N + i] = a.flow[i];
        uint32_t *w = reinterpret_cast<uint32_t *>(tail + 2 * a.N);
        w[i] = a.status[i];
        if (i == 0) w[a.N] = (uint32_t)(a.sticky[0] | a.sticky[1]);
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

template <bool ROW>
__global__ __launch_bounds__(64) void rhs_kernel(const RhsArgs a)
{
    Lane L; int64_t r;
    if (!lane_setup(0, a.N, a.n, a.R, L, r)) return;
    const int64_t idx = r * a.n + L.z;
    RK k; load_reactor(a.par, a.bc, a.N, r, a.n, k); mask_reactor_for_lane(L, k);
    double y[3] = {a.pH[idx], a.Cl[idx], a.T[idx]}, f[3];
    const bool bad = rhs_full<ROW>(L, kp_of(default_ktab()), kt_of(default_ktab()), k, y, f);
    a.dpH[idx] = f[SPH]; a.dCl[idx] = f[SCL]; a.dT[idx] = f[STT];
    const bool anybad = seg_any(L, bad);
    if (L.z == 0) a.flags[r] = anybad ? ST_T_RANGE : 0u;
}

// ReactorState.update_derived placeholders after the state was overwritten (reactor.py:137-147)
struct PlaceholderArgs { int64_t count; const double *pH; double *dH, *dRho, *dK; };
__global__ __launch_bounds__(256) void derived_placeholder_kernel(const PlaceholderArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.count) return;
    a.dH[i] = exp10_k(kp_of(default_ktab()), -a.pH[i]); a.dRho[i] = 998.2; a.dK[i] = 0.0001;
}

// Self-test of the cross-lane primitives against ds_bpermute-based __shfl:
// out[lane] = number of mismatching moves for that lane (parity tests assert 0).
struct ShuffleTestArgs { int n; int *out; };
template <bool ROW>
__global__ __launch_bounds__(64) void shuffle_selftest_kernel(const ShuffleTestArgs a)
{
    Lane L; int64_t r;
    __shared__ double xrow[XROW_CELLS];
    const bool present = lane_setup(0, 64 / a.n, a.n, 64 / a.n, L, r);
    if constexpr (!ROW) L.xrow = (LdsDouble *)xrow + XROW_PAD;
    if (!present) { a.out[threadIdx.x] = 0; return; }
    const int lane = threadIdx.x & 63;
    const double x = 1000.0 * (lane + 1) + 0.5;
    int bad = 0;
    auto chk = [&](double got, int src, bool valid) { if (valid && got != 1000.0 * (src + 1) + 0.5) bad++; };
    chk(from_lo<ROW, 1>(L, x), lane - 1, L.z >= 1); chk(from_hi<ROW, 1>(L, x), lane + 1, L.z + 1 < L.n);
    chk(from_lo<ROW, 2>(L, x), lane - 2, L.z >= 2); chk(from_hi<ROW, 2>(L, x), lane + 2, L.z + 2 < L.n);
    chk(from_lo<ROW, 4>(L, x), lane - 4, L.z >= 4); chk(from_hi<ROW, 4>(L, x), lane + 4, L.z + 4 < L.n);
    chk(from_lo<ROW, 8>(L, x), lane - 8, L.z >= 8); chk(from_hi<ROW, 8>(L, x), lane + 8, L.z + 8 < L.n);
    {   // the exchange-row moves of the cyclic reduction, every stride, and the top level's one partner
        double lo, hi;
        both<ROW, 1>(L, x, lo, hi); chk(lo, lane - 1, L.z >= 1); chk(hi, lane + 1, L.z + 1 < L.n);
        both<ROW, 2>(L, x, lo, hi); chk(lo, lane - 2, L.z >= 2); chk(hi, lane + 2, L.z + 2 < L.n);
        both<ROW, 4>(L, x, lo, hi); chk(lo, lane - 4, L.z >= 4); chk(hi, lane + 4, L.z + 4 < L.n);
        both<ROW, 8>(L, x, lo, hi); chk(lo, lane - 8, L.z >= 8); chk(hi, lane + 8, L.z + 8 < L.n);
        if constexpr (!ROW) {
            both<ROW, 16>(L, x, lo, hi); chk(lo, lane - 16, L.z >= 16); chk(hi, lane + 16, L.z + 16 < L.n);
            both<ROW, 32>(L, x, lo, hi); chk(lo, lane - 32, L.z >= 32); chk(hi, lane + 32, L.z + 32 < L.n);
            int top = 1;
            while (2 * top < L.n) top *= 2;
            const bool up = L.z >= top;
            const double got = top == 1 ? from_partner<false, 1>(L, x) : top == 2 ? from_partner<false, 2>(L, x)
                             : top == 4 ? from_partner<false, 4>(L, x) : top == 8 ? from_partner<false, 8>(L, x)
                             : top == 16 ? from_partner<false, 16>(L, x) : from_partner<false, 32>(L, x);
            chk(got, up ? lane - top : lane + top, up || (L.z + top < L.n));
        }
    }
    double ref = 0.0;
    for (int j = 0; j < L.n; ++j) ref += 1000.0 * (L.base + j + 1) + 0.5; // exact in fp64 (small integers + halves)
    if (seg_sum<ROW>(L, x) != ref) bad++;
    a.out[threadIdx.x] = bad;
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
    int rcode = 2, it = 0;
    for (it = 0; it < a.max_iter; ++it) {
        // charge_balance_error chemistry.py:193-228
        const double H = exp10_k(kp_of(default_ktab()), -pH);
        const double OH = Kw / H;
        const double H2 = H * H;
        const double D = H2 + Ka1 * H + Ka1 * Ka2;
        const double a1 = (Ka1 * H) / D, a2 = (Ka1 * Ka2) / D;
        const double fval = H - OH + a1 * CT + 2 * (a2 * CT) - alk_eq;
        // charge_balance_derivative chemistry.py:230-269
        const double dH = -rc::LN10 * H;
        const double dOH = -(Kw / H2) * dH;
        const double dD = 2 * H + Ka1;
        const double D2 = D * D;
        const double da1 = Ka1 * (D - H * dD) / D2;
        const double da2 = -Ka1 * Ka2 * dD / D2;
        const double df = dH - dOH + CT * da1 * dH + 2 * (CT * da2 * dH);
        if (fabs(df) < 1e-15) { rcode = 1; break; }           // chemistry.py:309-312
        const double delta = -fval / df;
        const double pH_new = fmin(fmax(pH + delta, 0.0), 14.0);
        if (fabs(delta) < a.tol) { pH = pH_new; rcode = 0; ++it; break; }
        pH = pH_new;
    }
    a.pH[i] = pH; a.iters[i] = it; a.rc[i] = rcode;
}

} // namespace wt
