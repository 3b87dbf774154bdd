// wt_triad.hpp -- the physics step with one wavefront PER SPECIES: a wavefront-group of reactors is advanced by a
// workgroup of three wavefronts (pH, chlorine, temperature), lane = reactor zone in each of them.
//
// Why: the one-wavefront mapping (wt_device.hpp) holds all three species of a zone in one lane -- 460 registers, one
// wavefront per SIMD, and at the metric's size (1 250 groups for 1 024 SIMDs) nothing to overlap a stall with.  The
// ODE system is block-structured by species (the Jacobian is block lower-triangular [T | pH | Cl] with tridiagonal
// diagonal blocks), so the per-lane state splits three ways: each wavefront keeps ONE species' Radau iterates,
// dense output, Jacobian band and cyclic-reduction factors (all in registers, no factor store in LDS), evaluates
// only its own rows of derivatives(), factorises and solves only its own tridiagonal systems.  Three times the
// wavefronts at a third of the registers: three wavefronts per SIMD, which is what hides the DPP / LDS / branch
// latencies a lone wavefront exposes.
//
// What crosses wavefronts goes through LDS at workgroup barriers:
//   B1  zone properties a row of another species needs: mixing coefficient K[i,i+1] and Arrhenius rate (from T),
//       HOCl decay factor (from pH); the temperature range check
//   B2  Newton increment of T  -> right-hand sides of pH and Cl (block forward substitution)
//   B3  Newton increment of pH -> right-hand side of Cl
//   B4  per-reactor partial sums of the norms (each wavefront reduces its own species over the zones)
//   B5-B7 the same for the error estimate of a converged iteration
// Every scalar decision of scipy's algorithm (step size, Newton convergence, accept / reject, Jacobian refresh) is
// taken by all three wavefronts redundantly from the same partial sums with the same instructions -- the solver
// state machine of wt_device.hpp, unchanged -- so they stay in lock step without a master.
//
// Same arithmetic per row as wt_device.hpp (prop_pH / prop_T / the stencil forms are shared), same decision
// sequence; only the order in which the three species' contributions enter a norm differs (per-species zone sums
// first).
#pragma once
#include "wt_device.hpp"

namespace wt {
namespace tri {

// exchange slots (64 doubles each) behind the reactor constants
enum : int {
    X_K = 0,     // [4] K[i,i+1] of evaluation slots 0..2 (stages / colours) and 3 (the point yc)
    X_KT = 4,    // [4] Arrhenius rate
    X_PHI = 8,   // [4] HOCl/OCl- decay factor
    X_DT = 12,   // [3] Newton increment of T: real, complex re, im
    X_DP = 15,   // [3] ... of pH
    X_NH = 18,   // [2 columns: pH, T][first, retry] finite-difference step of the column
    X_NM = 22,   // [2 columns][3 row species][max |diff|, scale]
    X_SLOTS = 34
};
constexpr int PART_KINDS = 4;
constexpr int TRIP_CAP = 1 << 20;      // no outer step comes near; an exit every wavefront reaches whatever happens

template <int LV> struct Lds3 {
    static constexpr int RK_DOUBLES = RK_UNI * rk_maxr(LV) + RK_LANE * 64;
    static constexpr int HIST_DOUBLES = RK_MAXR;
    static constexpr int X0 = RK_DOUBLES + HIST_DOUBLES;
    static constexpr int IO_DOUBLES = (int)((sizeof(wts::StepIO) + 7) / 8);
    static_assert(IO_DOUBLES <= X_SLOTS * 64, "StepIO aliases the exchange slots");
    static constexpr int BAD0 = X0 + X_SLOTS * 64;                      // 2 x 64 ints (stage block, single point)
    static constexpr int PART0 = BAD0 + 64;                             // [PART_KINDS][3 species][RK_MAXR]
    static constexpr int CHK0 = PART0 + PART_KINDS * 3 * RK_MAXR;       // 2 x 3 lock-step words, queue hand-off
    static constexpr int TOTAL = CHK0 + 8;
};

struct KC { double lo, hi, d; };   // K[i,i-1], K[i,i+1], diagonal of the mixing operator (reactor.py:321-337)

// mixing suppression of the interface above this zone: wt_device.hpp rhs_rows, same bits
template <bool ROW>
__device__ __forceinline__ double k_upper(const Lane &L, double Kex_hi, double dz, double u2, double ricrit, double rihulp,
                                          double supp, double unsupp, double rho)
{
#pragma clang fp contract(off)
    const double rho_hi = from_hi<ROW, 1>(L, rho);
    const double drho = rho_hi - rho;
    const double ravg = 0.5 * (rho + rho_hi);
    const double num = (9.81 * drho) * dz, den = ravg * u2;
    const double s = (__builtin_fma(-ricrit, den, num) > rihulp * den) ? supp : unsupp;
    return Kex_hi * s;
}
__device__ __forceinline__ KC kc_make(const Lane &L, double k_hi, double k_lo_raw, double Qv_out)
{
#pragma clang fp contract(off)
    KC k; k.hi = k_hi; k.lo = keep_m(L.m_lo[0], k_lo_raw); k.d = -(k.lo + k.hi) - Qv_out;
    return k;
}
template <bool ROW> __device__ __forceinline__ double mix_row(const Lane &L, const KC &k, double x)
{
#pragma clang fp contract(off)
    const double x_lo = keep_m(L.m_lo[0], from_lo<ROW, 1>(L, x)), x_hi = keep_m(L.m_hi[0], from_hi<ROW, 1>(L, x));
    return (k.lo * x_lo + k.hi * x_hi) + k.d * x;
}
// the three rows of derivatives() (reactor.py:349-376, :385-411, :420-443), forms of wt_device.hpp rhs_rows
template <bool ROW> __device__ __forceinline__ double row_pH(const Lane &L, const KC &k, double Qv_in, double H_in, double acid0, double H, double iw)
{
#pragma clang fp contract(off)
    return -(__builtin_fma(Qv_in, H_in - H, acid0) + mix_row<ROW>(L, k, H)) * iw;
}
template <bool ROW> __device__ __forceinline__ double row_Cl(const Lane &L, const KC &k, double Qv_in, double Cl_in, double dose0, double kphi, double Cl)
{
#pragma clang fp contract(off)
    return __builtin_fma(-kphi, Cl, __builtin_fma(Qv_in, Cl_in - Cl, dose0) + mix_row<ROW>(L, k, Cl));
}
template <bool ROW> __device__ __forceinline__ double row_T(const Lane &L, const KC &k, double Qv_in, double T_in, double UAr_on, double T_amb, double T)
{
#pragma clang fp contract(off)
    return __builtin_fma(-UAr_on, T - T_amb, Qv_in * (T_in - T) + mix_row<ROW>(L, k, T));
}

// per-species constants, fetched from the parked copy at the top of a block that needs them
struct CPH { RK k; double Qv_in, Qv_out, H_in, acid0; };
struct CCL { double Qv_in, Qv_out, Cl_in, dose0; };
struct CTT { double Kex_hi, dz, u2, ricrit, rihulp, supp, unsupp, Qv_in, Qv_out, T_in, UAr_on, T_amb; };
__device__ __forceinline__ CPH fetch_pH(const RKStore &st)
{
    CPH c;
    c.k.Kw = st.uni[0 * st.stride]; c.k.Ka1 = st.uni[1 * st.stride]; c.k.Ka1Ka2 = st.uni[2 * st.stride]; c.k.KaH = st.uni[3 * st.stride];
    c.k.cbeta = st.uni[4 * st.stride]; c.H_in = st.uni[8 * st.stride];
    c.Qv_in = st.lane[1 * 64]; c.Qv_out = st.lane[2 * 64]; c.acid0 = st.lane[3 * 64];
    return c;
}
__device__ __forceinline__ CCL fetch_Cl(const RKStore &st)
{
    CCL c;
    c.Cl_in = st.uni[9 * st.stride]; c.Qv_in = st.lane[1 * 64]; c.Qv_out = st.lane[2 * 64]; c.dose0 = st.lane[4 * 64];
    return c;
}
__device__ __forceinline__ CTT fetch_T(const RKStore &st)
{
    CTT c;
    c.dz = st.uni[5 * st.stride]; c.u2 = st.uni[6 * st.stride]; c.supp = st.uni[7 * st.stride]; c.T_in = st.uni[10 * st.stride];
    c.T_amb = st.uni[11 * st.stride]; c.UAr_on = st.uni[12 * st.stride]; c.unsupp = st.uni[13 * st.stride];
    c.ricrit = st.uni[14 * st.stride]; c.rihulp = st.uni[16 * st.stride];
    c.Kex_hi = st.lane[0 * 64]; c.Qv_in = st.lane[1 * 64]; c.Qv_out = st.lane[2 * 64];
    return c;
}

// ---------------------------------------------------------------- one species' tridiagonal systems
// (mu_real / h I - J_ss) and (mu_complex / h I - J_ss), factored by parallel cyclic reduction; the factors stay in
// registers (6 LV + 3 doubles).
template <bool ROW, int LV, int l>
__device__ __forceinline__ void factor_level(const Lane &L, double &ar, double &dr, double &cr, cplx &ac, cplx &dc, cplx &cc, SysFactors<LV> &F)
{
    constexpr int s = 1 << l;
    const bool vlo = L.z - s >= 0, vhi = L.z + s < L.n;
    {
        const double id = rcp(dr);
        const double id_lo = from_lo<ROW, s>(L, id), id_hi = from_hi<ROW, s>(L, id);
        const double a_lo = from_lo<ROW, s>(L, ar), c_lo = from_lo<ROW, s>(L, cr);
        const double a_hi = from_hi<ROW, s>(L, ar), c_hi = from_hi<ROW, s>(L, cr);
        const double al = ar * (vlo ? id_lo : 1.0);
        const double ga = cr * (vhi ? id_hi : 1.0);
        dr = dr - al * keep_m(L.m_lo[l], c_lo) - ga * keep_m(L.m_hi[l], a_hi);
        ar = -al * keep_m(L.m_lo[l], a_lo);
        cr = -ga * keep_m(L.m_hi[l], c_hi);
        F.ra[l] = al; F.rg[l] = ga;
    }
    {
        const cplx cid = cinv(dc);
        const cplx i_lo = cfrom_lo<ROW, s>(L, cid), i_hi = cfrom_hi<ROW, s>(L, cid);
        const cplx a_lo = cfrom_lo<ROW, s>(L, ac), c_lo = cfrom_lo<ROW, s>(L, cc);
        const cplx a_hi = cfrom_hi<ROW, s>(L, ac), c_hi = cfrom_hi<ROW, s>(L, cc);
        const cplx il = {vlo ? i_lo.r : 1.0, keep_m(L.m_lo[l], i_lo.i)};
        const cplx ih = {vhi ? i_hi.r : 1.0, keep_m(L.m_hi[l], i_hi.i)};
        const cplx al = cmul(ac, il);
        const cplx ga = cmul(cc, ih);
        const cplx cl = {keep_m(L.m_lo[l], c_lo.r), keep_m(L.m_lo[l], c_lo.i)};
        const cplx ah = {keep_m(L.m_hi[l], a_hi.r), keep_m(L.m_hi[l], a_hi.i)};
        double dre = dc.r, dim = dc.i;
        dre = __builtin_fma(-al.r, cl.r, dre); dim = __builtin_fma(-al.r, cl.i, dim);
        dre = __builtin_fma(al.i, cl.i, dre);  dim = __builtin_fma(-al.i, cl.r, dim);
        dre = __builtin_fma(-ga.r, ah.r, dre); dim = __builtin_fma(-ga.r, ah.i, dim);
        dre = __builtin_fma(ga.i, ah.i, dre);  dim = __builtin_fma(-ga.i, ah.r, dim);
        dc = {dre, dim};
        const cplx na = cmul(al, {keep_m(L.m_lo[l], a_lo.r), keep_m(L.m_lo[l], a_lo.i)});
        const cplx nc = cmul(ga, {keep_m(L.m_hi[l], c_hi.r), keep_m(L.m_hi[l], c_hi.i)});
        ac = {-na.r, -na.i};
        cc = {-nc.r, -nc.i};
        F.ca[l] = al; F.cg[l] = ga;
    }
    if constexpr (l + 1 < LV) factor_level<ROW, LV, l + 1>(L, ar, dr, cr, ac, dc, cc, F);
}

template <bool ROW, int LV>
__device__ __forceinline__ void factorize1(const Lane &L, const double jd[3], double h, SysFactors<LV> &F)
{
    const double ih = rcp(h);
    const double mr = rc::MU_REAL * ih, mcr = rc::MU_CR * ih, mci = rc::MU_CI * ih;
    double ar = -jd[0], dr = mr - jd[1], cr = -jd[2];
    cplx ac = {ar, 0.0}, dc = {mcr - jd[1], mci}, cc = {cr, 0.0};
    factor_level<ROW, LV, 0>(L, ar, dr, cr, ac, dc, cc, F);
    F.rinv = rcp(dr);
    F.cinv = cinv(dc);
}

template <bool ROW, int LV, int l>
__device__ __forceinline__ void real_level(const Lane &L, const SysFactors<LV> &s, double &b)
{
    constexpr int st = 1 << l;
    const double b_lo = from_lo<ROW, st>(L, b), b_hi = from_hi<ROW, st>(L, b);
    b = b - s.ra[l] * keep_m(L.m_lo[l], b_lo) - s.rg[l] * keep_m(L.m_hi[l], b_hi);
    if constexpr (l + 1 < LV) real_level<ROW, LV, l + 1>(L, s, b);
}

// a value the lane below / above published in an exchange slot (clamped at the wavefront's ends; masked by the caller)
__device__ __forceinline__ double x_lo(const double *slot, int lane) { return slot[lane > 0 ? lane - 1 : 0]; }
__device__ __forceinline__ double x_hi(const double *slot, int lane) { return slot[lane < 63 ? lane + 1 : 63]; }

// ---------------------------------------------------------------- the work item
template <int LV, bool ROW>
__device__ __forceinline__ void run_item3(ArgPtr pa, const Lane &L, double *lds, const int sp, int group, int step0, int cnt)
{
    using M = Lds3<LV>;
    ArgPtr a = fresh(pa);
    const int n_zones = a->n, R = a->R;
    const int lane = threadIdx.x & 63, seg = lane / n_zones;
    const int64_t q_first = (int64_t)group * R;
    const int64_t q_end = a->q_ctrl ? a->N : a->r1;
    const bool present = (seg < R) && (q_first + seg < q_end);
    const int64_t r = present ? (int64_t)a->perm[q_first + seg] : 0;
    const int64_t idx = r * n_zones + L.z;
    const double dt = a->dt;
    const int step_limit = a->step_limit, sens_on = a->sens.on, plc_on = a->sens.plc_on;
    const bool want_diag = a->wave_diag != nullptr;
    const RKStore ks = {lds + seg, lds + RK_UNI * rk_maxr(LV) + lane, rk_maxr(LV)};
    int *hist0 = reinterpret_cast<int *>(lds + M::RK_DOUBLES);
    int *rix = hist0 + RK_MAXR;
    double *X = lds + M::X0;
    int *xbad = reinterpret_cast<int *>(lds + M::BAD0);
    double *part = lds + M::PART0;                     // part[(kind * 3 + species) * RK_MAXR + seg]
    unsigned long long *chk = reinterpret_cast<unsigned long long *>(lds + M::CHK0);
    wts::StepIO &io = *reinterpret_cast<wts::StepIO *>(X);
    auto PART = [&](int kind, int s) -> double & { return part[(kind * 3 + s) * RK_MAXR + seg]; };
    auto psum = [&](int kind) { return (PART(kind, 0) + PART(kind, 1)) + PART(kind, 2); };
    double *const state = (sp == SPH) ? a->pH : (sp == SCL ? a->Cl : a->T);

    // ---- this species' state of the reactor, carried from one outer step to the next
    double y0 = (sp == SPH) ? 7.0 : (sp == SCL ? 1.0 : 20.0);
    double f = 0;                                     // f(y0) when f_valid
    double t_out = 0;
    double der0 = 0, der1 = 0, badval = 0;            // derived: pH wave [H+]; T wave density, rate
    double flow_used = 0;
    uint32_t st = 0;
    bool frozen = !present, f_valid = false, wrote_k = false, raised = false;
    int steps_done = 0;
    SolverCounters last_cnt = {0, 0, 0, 0, 0};
    long long diag_trips = 0, diag_newton = 0, diag_fact = 0, diag_jac = 0, diag_f3 = 0;
    const long long clk0 = want_diag ? __builtin_amdgcn_s_memtime() : 0, wall0 = want_diag ? __builtin_amdgcn_s_memrealtime() : 0;
    if (present) {
        st = a->status[r];
        if (st & (ST_T_RANGE | ST_T_RANGE_POST)) frozen = true;
        y0 = state[idx];
        t_out = a->time[r];
    }
    if (sp == SPH) {
        if (present) {
            RK k0; load_reactor(a->par, a->bc, a->N, r, n_zones, k0); mask_reactor_for_lane(L, k0);
            park_reactor(ks, k0);
            if (sens_on && L.z == 0) hist0[seg] = a->sens.hist_value ? a->sens.hist_pos[r] : 0;
            if (L.z == 0) rix[seg] = (int)r;
        }
    }
    __syncthreads();

    for (int k = 0; k < cnt; ++k) {
        bool stepped = false;
        // ================= one IntegratedCSTR.step(): a fresh scipy solver object (reactor.py:476)
        double yc = y0, W[3] = {0, 0, 0};
        double aux = 0;
        double Q[3] = {0, 0, 0}, y_old = 0;
        double jd[3] = {0, 0, 0}, jx[3] = {0, 0, 0}, jcp = 0;   // own band; coupling to T columns (pH, Cl rows); dCl/dpH own zone
        SysFactors<LV> F;
        uint32_t fl = 1u << 4;                        // current_jac = true
        Flag have_fac{fl, 1u << 0}, have_old{fl, 1u << 1}, have_old_l{fl, 1u << 2}, have_sol{fl, 1u << 3}, current_jac{fl, 1u << 4},
             have_lu{fl, 1u << 5}, rejected{fl, 1u << 6}, keep_h{fl, 1u << 7}, have_norm_old{fl, 1u << 8}, have_rate{fl, 1u << 9},
             bad{fl, 1u << 10}, failed{fl, 1u << 11}, fv{fl, 1u << 12}, need_jac{fl, 1u << 13},
             limit_hit{fl, 1u << 16}, pend_f{fl, 1u << 17}, jac_after_fnew{fl, 1u << 18};
        fv = f_valid;
        double fac = 0;
        double t = t_out, t_bound = t_out + dt, max_step = fmin(dt, 10.0);
        double h = 0, t_new = 0, h_abs = 0, h_abs_l = 0, min_step = 0;
        double h_abs_old = 0, err_old = 0, h_abs_old_l = 0, err_old_l = 0;
        double sol_t_old = 0, sol_h = 1;
        int kk = 0, n_iter = 0; double dW_norm_old = 0, rate = 0;
        double error_norm = 0, safety = 0;
        double d1 = 0, h0 = 0;
        SolverCounters cnt_s = {0, 0, 0, 0, 0};
        int attempts = 0, badstage = 0, trips = 0;
        bool desync = false;
        const double in3 = rcp((double)(3 * L.n)), in9 = rcp((double)(9 * L.n));
        int phase = frozen ? PH_DONE : PH_OUTER_BEGIN;
        F.rinv = 0; F.cinv = {0, 0};
#pragma unroll
        for (int l = 0; l < LV; ++l) { F.ra[l] = F.rg[l] = 0; F.ca[l] = {0, 0}; F.cg[l] = {0, 0}; }

        // ---- B0: a non-finite state anywhere in the reactor; the norms of select_initial_step when f(y0) is carried over
        {
            const double sc = ATOL + fabs(yc) * RTOL, isc = rcp(sc);
            const double v0 = yc * isc, v1 = f * isc;
            const double s0 = seg_sum<ROW>(L, v0 * v0), s1 = seg_sum<ROW>(L, v1 * v1);
            const bool nf = seg_any(L, !isfinite(y0));
            if (L.z == 0) { PART(0, sp) = s0; PART(1, sp) = s1; PART(3, sp) = nf ? 1.0 : 0.0; }
        }
        __syncthreads();
        if (!frozen) {
            // scipy refuses a non-finite initial state: ValueError out of step(), self.state untouched (base.py:19-20)
            if (psum(3) != 0.0) { st |= ST_NONFINITE; frozen = true; phase = PH_DONE; }
            else if (fv) { cnt_s.nfev++; phase = PH_INIT_STEP; }   // f(y0) known: counted as scipy counts it
        }
        const bool solving = !frozen;

        auto reject_step = [&]() {
#pragma clang fp contract(off)
            const double fct = predict_factor(h_abs_l, have_old_l, h_abs_old_l, error_norm, err_old_l);
            h_abs_l *= fmax(MIN_FACTOR, safety * fct);
            have_lu = false; rejected = true; cnt_s.nrej++;
            phase = PH_ATTEMPT;
        };
        auto accept_step = [&]() {
#pragma clang fp contract(off)
            const bool recompute_jac = (n_iter > 2) && have_rate && (rate > 1e-3);
            double fct = predict_factor(h_abs_l, have_old_l, h_abs_old_l, error_norm, err_old_l);
            fct = fmin(MAX_FACTOR, safety * fct);
            if (!recompute_jac && fct < 1.2) fct = 1.0; else have_lu = false;
            h_abs_old = h_abs;            // sic radau.py:520: the solver-level value
            err_old = error_norm;
            have_old = true;
            h_abs = h_abs_l * fct;
            {
                const double z0 = rc::T00 * W[0] + rc::T01 * W[1] + rc::T02 * W[2];
                const double z1 = rc::T10 * W[0] + rc::T11 * W[1] + rc::T12 * W[2];
                const double z2 = W[0] + W[1];
                y_old = yc;
                Q[0] = z0 * rc::P00 + z1 * rc::P10 + z2 * rc::P20;  // Q = Z^T P  radau.py:541-543
                Q[1] = z0 * rc::P01 + z1 * rc::P11 + z2 * rc::P21;
                Q[2] = z0 * rc::P02 + z1 * rc::P12 + z2 * rc::P22;
                yc = yc + z2;
            }
            sol_t_old = t; sol_h = t_new - t; have_sol = true;
            t = t_new;
            cnt_s.nsteps++; cnt_s.nfev++;     // f(y_new) counted where scipy calls it
            pend_f = true; fv = false;
            current_jac = recompute_jac;
            const bool more = (t - t_bound) < 0;
            if (recompute_jac || !more) { jac_after_fnew = recompute_jac; phase = PH_FNEW; }
            else phase = PH_STEP_BEGIN;
        };

        while (true) {
            if (__ballot(phase != PH_DONE) == 0ull) break;
            if (++trips > TRIP_CAP) { if (phase != PH_DONE) { failed = true; limit_hit = true; phase = PH_DONE; } break; }
            // ================= trips that need no RHS evaluation
            if (phase == PH_INIT_STEP) {
#pragma clang fp contract(off)
                // select_initial_step (common.py:68-134), order 3, up to the probe point y0 + h0 f0
                const double d0 = sqrt(psum(0) * in3);
                d1 = sqrt(psum(1) * in3);
                h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 * rcp(d1);
                h0 = fmin(h0, fabs(t_bound - t));
                aux = __builtin_fma(h0, f, yc);
                phase = PH_F1;
            }
            if (phase == PH_STEP_BEGIN) {
                min_step = 10 * fabs(ulp_above(t));                      // radau.py:408
                if (h_abs > max_step) { h_abs_l = max_step; have_old_l = false; }
                else if (h_abs < min_step) { h_abs_l = min_step; have_old_l = false; }
                else { h_abs_l = h_abs; have_old_l = have_old; h_abs_old_l = h_abs_old; err_old_l = err_old; }
                rejected = false; keep_h = false;
                phase = PH_ATTEMPT;
            }
            if (phase == PH_ATTEMPT) {
                if (!keep_h) {
                    if (step_limit > 0 && attempts >= step_limit) { failed = true; limit_hit = true; phase = PH_DONE; }
                    else if (h_abs_l < min_step) { failed = true; phase = PH_DONE; }  // radau.py:427-428
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
                    double Z0[3] = {0, 0, 0};
                    if (have_sol) {
                        const double isol = rcp(sol_h);
                        const double cs[3] = {rc::C0, rc::C1, 1.0};
#pragma unroll
                        for (int s = 0; s < 3; ++s) {
                            const double x = ((t + h * cs[s]) - sol_t_old) * isol;
                            const double p1 = x * x, p2 = p1 * x;
                            Z0[s] = ((Q[0] * x + Q[1] * p1 + Q[2] * p2) + y_old) - yc;
                        }
                    }
                    W[0] = rc::TI00 * Z0[0] + rc::TI01 * Z0[1] + rc::TI02 * Z0[2];
                    W[1] = rc::TI10 * Z0[0] + rc::TI11 * Z0[1] + rc::TI12 * Z0[2];
                    W[2] = rc::TI20 * Z0[0] + rc::TI21 * Z0[1] + rc::TI22 * Z0[2];
                    kk = 0; have_norm_old = false; have_rate = false; rate = 0.0;
                    phase = PH_NEWTON;
                }
            }
            if (__ballot(phase == PH_NEWTON && !have_lu) != 0ull) diag_fact++;
            if (phase == PH_NEWTON && !have_lu) {
                factorize1<ROW, LV>(L, jd, h, F); have_lu = true; cnt_s.nlu += 2;      // radau.py:454-456
            }

            // ================= this trip's evaluation points
            const bool newton = (phase == PH_NEWTON);
            const bool refine = (phase == PH_ERR_REFINE);
            diag_trips++;
            const bool any_newton = __ballot(newton) != 0ull;
            if (any_newton) diag_newton++;
            const bool eval0 = (phase == PH_OUTER_BEGIN || phase == PH_F1 || refine || phase == PH_FNEW || newton);
            const bool eval3 = newton && pend_f;
            const bool any_eval3 = __ballot(eval3) != 0ull;
            const bool single = any_eval3 || (!any_newton && __ballot(eval0) != 0ull);
            double ye[3], Fe[3] = {0, 0, 0}, fy = 0;
            {
                const double z0 = rc::T00 * W[0] + rc::T01 * W[1] + rc::T02 * W[2];
                const double z1 = rc::T10 * W[0] + rc::T11 * W[1] + rc::T12 * W[2];
                const double z2 = W[0] + W[1];
                double p0 = yc;                                           // PH_OUTER_BEGIN, PH_FNEW
                if (phase == PH_F1) p0 = aux;
                if (refine) p0 = yc + aux;
                if (newton) p0 = yc + z0;
                ye[0] = p0; ye[1] = yc + z1; ye[2] = yc + z2;
            }
            int badbits = 0;
            if (any_newton) {
                // ---- three stage points (the other lanes' slot-1/2 results are simply not used)
                if (sp == STT) {
                    const CTT c = fetch_T(ks);
                    double kh[3];
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const PropT pt = prop_T(ye[s]);
                        kh[s] = k_upper<ROW>(L, c.Kex_hi, c.dz, c.u2, c.ricrit, c.rihulp, c.supp, c.unsupp, pt.rho);
                        X[(X_K + s) * 64 + lane] = kh[s]; X[(X_KT + s) * 64 + lane] = pt.kT;
                        if (pt.bad && (s == 0 ? eval0 : newton)) badbits |= 1 << s;
                    }
                    xbad[lane] = badbits;
                    __syncthreads();                                      // B1
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const KC kc = kc_make(L, kh[s], from_lo<ROW, 1>(L, kh[s]), c.Qv_out);
                        Fe[s] = row_T<ROW>(L, kc, c.Qv_in, c.T_in, c.UAr_on, c.T_amb, ye[s]);
                    }
                } else if (sp == SPH) {
                    const CPH c = fetch_pH(ks);
                    double H[3], iw[3];
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const PropPH pp = prop_pH(c.k, ye[s]);
                        H[s] = pp.H; iw[s] = pp.iw;
                        X[(X_PHI + s) * 64 + lane] = pp.phi;
                    }
                    __syncthreads();                                      // B1
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const double *xk = X + (X_K + s) * 64;
                        const KC kc = kc_make(L, xk[lane], x_lo(xk, lane), c.Qv_out);
                        Fe[s] = row_pH<ROW>(L, kc, c.Qv_in, c.H_in, c.acid0, H[s], iw[s]);
                    }
                    badbits = xbad[lane];
                } else {
                    const CCL c = fetch_Cl(ks);
                    __syncthreads();                                      // B1
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const double *xk = X + (X_K + s) * 64;
                        const KC kc = kc_make(L, xk[lane], x_lo(xk, lane), c.Qv_out);
                        const double kphi = X[(X_KT + s) * 64 + lane] * X[(X_PHI + s) * 64 + lane];
                        Fe[s] = row_Cl<ROW>(L, kc, c.Qv_in, c.Cl_in, c.dose0, kphi, ye[s]);
                    }
                    badbits = xbad[lane];
                }
                if (eval0 && phase != PH_FNEW) cnt_s.nfev++;
                if (newton) cnt_s.nfev += 2;
            }
            if (single) {
                // ---- one point: the deferred f(y_new) riding along with a Newton trip (slot 3), or this trip's only point
                if (any_eval3) diag_f3++;
                const double p = any_eval3 ? yc : ye[0];
                const bool want = any_eval3 ? eval3 : eval0;
                double fo = 0;
                int bb = 0;
                if (sp == STT) {
                    const CTT c = fetch_T(ks);
                    const PropT pt = prop_T(p);
                    const double kh = k_upper<ROW>(L, c.Kex_hi, c.dz, c.u2, c.ricrit, c.rihulp, c.supp, c.unsupp, pt.rho);
                    X[(X_K + 3) * 64 + lane] = kh; X[(X_KT + 3) * 64 + lane] = pt.kT;
                    bb = (pt.bad && want) ? 1 : 0;
                    xbad[64 + lane] = bb;
                    __syncthreads();                                      // B1'
                    const KC kc = kc_make(L, kh, from_lo<ROW, 1>(L, kh), c.Qv_out);
                    fo = row_T<ROW>(L, kc, c.Qv_in, c.T_in, c.UAr_on, c.T_amb, p);
                } else if (sp == SPH) {
                    const CPH c = fetch_pH(ks);
                    const PropPH pp = prop_pH(c.k, p);
                    X[(X_PHI + 3) * 64 + lane] = pp.phi;
                    __syncthreads();
                    const double *xk = X + (X_K + 3) * 64;
                    const KC kc = kc_make(L, xk[lane], x_lo(xk, lane), c.Qv_out);
                    fo = row_pH<ROW>(L, kc, c.Qv_in, c.H_in, c.acid0, pp.H, pp.iw);
                    bb = xbad[64 + lane];
                } else {
                    const CCL c = fetch_Cl(ks);
                    __syncthreads();
                    const double *xk = X + (X_K + 3) * 64;
                    const KC kc = kc_make(L, xk[lane], x_lo(xk, lane), c.Qv_out);
                    const double kphi = X[(X_KT + 3) * 64 + lane] * X[(X_PHI + 3) * 64 + lane];
                    fo = row_Cl<ROW>(L, kc, c.Qv_in, c.Cl_in, c.dose0, kphi, p);
                    bb = xbad[64 + lane];
                }
                if (any_eval3) {
                    if (eval3) { pend_f = false; f = fo; }               // (counted in nfev when the step was accepted)
                    badbits |= bb << 3;
                } else {
                    Fe[0] = fo;
                    badbits = bb;
                    if (eval0 && phase != PH_FNEW) cnt_s.nfev++;
                }
                fy = fo;
            }
            if (__ballot(badbits != 0) != 0ull) {   // rare: a zone temperature outside [0, 100] C
                // the reference raises in the first evaluation, at the first zone, that sees it: scipy calls
                // f(y_new) of the accepted step before the stages of the next Newton iteration
                const bool mine = badbits != 0;
                if (mine && !bad && sp == STT) {
                    const bool b3 = badbits & 8, b0 = badbits & 1, b1 = badbits & 2;
                    badstage = b3 ? 0 : (b0 ? 1 : (b1 ? 2 : 3));
                    badval = b3 ? yc : (b0 ? ye[0] : (b1 ? ye[1] : ye[2]));
                }
                bad |= mine;
                if (seg_any(L, bad)) { raised = true; phase = PH_DONE; }
            }
            (void)fy;

            // ================= per-phase epilogues, species part: what the norms need
            const double scale = ATOL + fabs(yc) * RTOL;
            double dWr = 0, dWcr = 0, dWci = 0;
            if (phase == PH_OUTER_BEGIN) {
                f = Fe[0];
                const double isc = rcp(scale);
                const double v0 = yc * isc, v1 = f * isc;
                const double s0 = seg_sum<ROW>(L, v0 * v0), s1 = seg_sum<ROW>(L, v1 * v1);
                if (L.z == 0) { PART(0, sp) = s0; PART(1, sp) = s1; }
            } else if (phase == PH_F1) {
                const double v = (Fe[0] - f) * rcp(scale);
                const double s0 = seg_sum<ROW>(L, v * v);
                if (L.z == 0) PART(2, sp) = s0;
            }
            if (any_newton) {
                // ---- one iteration of solve_collocation_system radau.py:84-134: (mu/h I - J) dW = TI F - mu/h W,
                // block forward substitution T -> pH -> Cl
                const double ih = rcp(h);
                const double M_real = rc::MU_REAL * ih, Mcr = rc::MU_CR * ih, Mci = rc::MU_CI * ih;
                const bool finite = isfinite(Fe[0]) && isfinite(Fe[1]) && isfinite(Fe[2]);
                double xr = (Fe[0] * rc::TI00 + Fe[1] * rc::TI01 + Fe[2] * rc::TI02) - M_real * W[0];
                const double re = Fe[0] * rc::TI10 + Fe[1] * rc::TI11 + Fe[2] * rc::TI12;
                const double im = Fe[0] * rc::TI20 + Fe[1] * rc::TI21 + Fe[2] * rc::TI22;
                cplx xc = {re - (Mcr * W[1] - Mci * W[2]), im - (Mcr * W[2] + Mci * W[1])};
                double *xdt = X + X_DT * 64, *xdp = X + X_DP * 64;
                if (sp == STT) {
                    pcr_rc_level<ROW, LV, 0>(L, F, xr, xc);
                    xr *= F.rinv; xc = cmul(xc, F.cinv);
                    xdt[lane] = xr; xdt[64 + lane] = xc.r; xdt[128 + lane] = xc.i;
                }
                __syncthreads();                                          // B2
                if (sp != STT) {
                    // rhs += J_sT x_T (J.pt / J.ct are 0 where there is no neighbour; what is read there is made finite)
                    const double tl = keep_m(L.m_lo[0], x_lo(xdt, lane)), tm = xdt[lane], th = keep_m(L.m_hi[0], x_hi(xdt, lane));
                    const double tlr = keep_m(L.m_lo[0], x_lo(xdt + 64, lane)), tmr = xdt[64 + lane], thr = keep_m(L.m_hi[0], x_hi(xdt + 64, lane));
                    const double tli = keep_m(L.m_lo[0], x_lo(xdt + 128, lane)), tmi = xdt[128 + lane], thi = keep_m(L.m_hi[0], x_hi(xdt + 128, lane));
                    xr = xr + (jx[0] * tl + jx[1] * tm + jx[2] * th);
                    xc = {xc.r + (jx[0] * tlr + jx[1] * tmr + jx[2] * thr), xc.i + (jx[0] * tli + jx[1] * tmi + jx[2] * thi)};
                }
                if (sp == SPH) {
                    pcr_rc_level<ROW, LV, 0>(L, F, xr, xc);
                    xr *= F.rinv; xc = cmul(xc, F.cinv);
                    xdp[lane] = xr; xdp[64 + lane] = xc.r; xdp[128 + lane] = xc.i;
                }
                __syncthreads();                                          // B3
                if (sp == SCL) {
                    xr = xr + jcp * xdp[lane];
                    xc = {xc.r + jcp * xdp[64 + lane], xc.i + jcp * xdp[128 + lane]};
                    pcr_rc_level<ROW, LV, 0>(L, F, xr, xc);
                    xr *= F.rinv; xc = cmul(xc, F.cinv);
                }
                dWr = xr; dWcr = xc.r; dWci = xc.i;
                const double is = rcp(scale);
                const double u = xr * is, v = xc.r * is, w = xc.i * is;
                const double s0 = seg_sum<ROW>(L, u * u + v * v + w * w);
                const bool nonfin = seg_any(L, !finite);
                if (newton && L.z == 0) { PART(2, sp) = s0; PART(3, sp) = nonfin ? 1.0 : 0.0; }
            }
            {
                // lock-step word: the three wavefronts must be at the same place of the same trip
                const unsigned long long nb = __ballot(newton), db = __ballot(phase == PH_DONE);
                if (lane == 0) chk[(trips & 1) * 3 + sp] = ((unsigned long long)(unsigned)trips << 32 | (unsigned)(k & 0xffff)) ^ nb ^ (db << 1);
            }
            __syncthreads();                                              // B4
            {
                const unsigned long long c0 = chk[(trips & 1) * 3], c1 = chk[(trips & 1) * 3 + 1], c2 = chk[(trips & 1) * 3 + 2];
                if (c0 != c1 || c0 != c2) {           // cannot happen; if it ever does, every wavefront sees it here and leaves together
                    if (phase != PH_DONE) { failed = true; limit_hit = true; phase = PH_DONE; }
                    desync = true;
                    break;
                }
            }
            bool conv = false;
            {
#pragma clang fp contract(off)
                // ================= the decisions, identical in the three wavefronts
                if (phase == PH_OUTER_BEGIN) {
                    phase = PH_INIT_STEP;
                } else if (phase == PH_F1) {
                    const double d2 = sqrt(psum(2) * in3) * rcp(h0);
                    double h1;
                    if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
                    else h1 = root4(0.01 * rcp(fmax(d1, d2)));
                    h_abs = fmin(fmin(100 * h0, h1), fmin(fabs(t_bound - t), max_step));
                    need_jac = true;                                          // radau.py:359-365
                    phase = PH_STEP_BEGIN;
                } else if (phase == PH_NEWTON) {
                    bool diverged = false;
                    if (psum(3) != 0.0) {
                        diverged = true;
                    } else {
                        const double dW_norm = sqrt(psum(2) * in9);
                        if (have_norm_old) { rate = dW_norm * rcp(dW_norm_old); have_rate = true; }
                        const double i1r = rcp(1 - rate);
                        if (have_rate && (rate >= 1 || powi6(rate, NEWTON_MAXITER - kk) * i1r * dW_norm > rc::NEWTON_TOL)) {
                            diverged = true;
                        } else {
                            W[0] += dWr; W[1] += dWcr; W[2] += dWci;
                            if (dW_norm == 0 || (have_rate && rate * i1r * dW_norm < rc::NEWTON_TOL)) conv = true;
                            dW_norm_old = dW_norm; have_norm_old = true;
                        }
                    }
                    n_iter = kk + 1;
                    kk++;
                    if (!conv && !diverged && kk == NEWTON_MAXITER) diverged = true;   // loop ran out: radau.py:136
                    if (diverged) {                                                   // radau.py:462-476
                        if (current_jac) { h_abs_l *= 0.5; have_lu = false; cnt_s.nrej++; phase = PH_ATTEMPT; }
                        else { need_jac = true; current_jac = true; have_lu = false; keep_h = true; phase = PH_ATTEMPT; }
                    }
                } else if (phase == PH_FNEW) {
                    // f(y_new) of an accepted step that needs it before anything else can happen:
                    // Jacobian refresh (radau.py:512-514) or the end of the outer step
                    f = Fe[0];
                    pend_f = false;
                    fv = true;
                    if (jac_after_fnew) { need_jac = true; jac_after_fnew = false; }
                    phase = ((t - t_bound) < 0) ? PH_STEP_BEGIN : PH_DONE;
                }
            }
            // ================= error estimate of a converged iteration / the second one after a rejection (radau.py:477-496)
            const bool do_err = conv || (refine && phase == PH_ERR_REFINE);
            if (__ballot(do_err) != 0ull) {
                double err, esc;
                {
                    const double ih_e = rcp(h);
                    const double z0 = rc::T00 * W[0] + rc::T01 * W[1] + rc::T02 * W[2];
                    const double z1 = rc::T10 * W[0] + rc::T11 * W[1] + rc::T12 * W[2];
                    const double z2 = W[0] + W[1];
                    const double ZE = (z0 * rc::E0 + z1 * rc::E1 + z2 * rc::E2) * ih_e;
                    err = (refine ? Fe[0] : f) + ZE;
                    esc = ATOL + fmax(fabs(yc), fabs(yc + z2)) * RTOL;
                }
                double *xdt = X + X_DT * 64, *xdp = X + X_DP * 64;
                if (sp == STT) {
                    real_level<ROW, LV, 0>(L, F, err);
                    err *= F.rinv;
                    xdt[lane] = err;
                }
                __syncthreads();                                          // B5
                if (sp != STT) {
                    const double tl = keep_m(L.m_lo[0], x_lo(xdt, lane)), tm = xdt[lane], th = keep_m(L.m_hi[0], x_hi(xdt, lane));
                    err = err + (jx[0] * tl + jx[1] * tm + jx[2] * th);
                }
                if (sp == SPH) {
                    real_level<ROW, LV, 0>(L, F, err);
                    err *= F.rinv;
                    xdp[lane] = err;
                }
                __syncthreads();                                          // B6
                if (sp == SCL) {
                    err = err + jcp * xdp[lane];
                    real_level<ROW, LV, 0>(L, F, err);
                    err *= F.rinv;
                }
                {
                    const double v = err * rcp(esc);
                    const double s0 = seg_sum<ROW>(L, v * v);
                    if (do_err && L.z == 0) PART(2, sp) = s0;
                }
                __syncthreads();                                          // B7
                if (do_err) {
#pragma clang fp contract(off)
                    error_norm = sqrt(psum(2) * in3);
                    if (!refine) {
                        safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + n_iter);
                        if (rejected && error_norm > 1) { aux = err; phase = PH_ERR_REFINE; }
                        else if (error_norm > 1) reject_step();
                        else accept_step();
                    } else {
                        if (error_norm > 1) reject_step(); else accept_step();
                    }
                }
            }

            // ================= finite-difference Jacobian at (yc, f) when a phase asked for it (common.py:257-382)
            if (__ballot(need_jac) != 0ull) {
                diag_jac++;
                const bool need = need_jac;
                if (need) { cnt_s.njev++; if (!have_fac) { fac = rc::NJ_F0; have_fac = true; } }
                const int zm = L.z % 3;
                const double fs = (f >= 0) ? 1.0 : -1.0;
                const double ysc = fs * fmax(ATOL, fabs(yc));
                double hcol = fd_step(yc, fac, ysc);
                if (sp != SCL) while (need && hcol == 0) { fac *= 10; hcol = fd_step(yc, fac, ysc); }    // common.py:327-330
                // D*[r]: change of this lane's row when the zone at offset r-1 was perturbed; S*: max(|f|, |f_new|)
                double Dp[3] = {0, 0, 0}, Sp[3] = {0, 0, 0}, Dt[3] = {0, 0, 0}, St[3] = {0, 0, 0};
                double maxd_p = -1, scl_p = 0, maxd_t = -1, scl_t = 0;
                double h_p = 0, h_t = 0;               // the step of the pH / T column of this lane's zone (every wavefront's copy)
                bool jbad = false; double jval = 0;
                auto by_offset = [&](const double c[3], double o[3]) {
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) o[rr] = sel3(c[0], c[1], c[2], (zm + rr + 2) % 3);   // offset rr-1 has colour (zm+rr+2) mod 3
                };
                // max |diff| over the rows of this species for the column of this lane's zone, numpy argmax order
                auto col_cand = [&](const double D[3], const double S[3], bool nb, double &md, double &sc) {
                    md = -1.0; sc = 0.0;
                    double d_lo = 0, s_lo = 0, d_hi = 0, s_hi = 0;
                    if (nb) {
                        d_lo = from_lo<ROW, 1>(L, D[2]); s_lo = from_lo<ROW, 1>(L, S[2]);
                        d_hi = from_hi<ROW, 1>(L, D[0]); s_hi = from_hi<ROW, 1>(L, S[0]);
                    }
                    if (nb && L.has_lo && fabs(d_lo) > md) { md = fabs(d_lo); sc = s_lo; }
                    if (fabs(D[1]) > md) { md = fabs(D[1]); sc = S[1]; }
                    if (nb && L.has_hi && fabs(d_hi) > md) { md = fabs(d_hi); sc = s_hi; }
                };
                // ---- one round of the pH and T columns: publish, evaluate rows, candidates.  retry: only columns in `redo`
                double Hb = 0, iwb = 0, kTb = 0, rhob = 0;
                auto fd_round = [&](const int retry, const bool redo_p, const bool redo_t, const double hp_own, const double ht_own) {
                    double *xnh_p = X + (X_NH + 0 * 2 + retry) * 64, *xnh_t = X + (X_NH + 1 * 2 + retry) * 64;
                    if (sp == STT) {
                        const CTT c = fetch_T(ks);
                        if (!retry) {
                            const PropT pb = prop_T(yc); kTb = pb.kT; rhob = pb.rho;
                            const double khb = k_upper<ROW>(L, c.Kex_hi, c.dz, c.u2, c.ricrit, c.rihulp, c.supp, c.unsupp, rhob);
                            X[(X_K + 3) * 64 + lane] = khb; X[(X_KT + 3) * 64 + lane] = kTb;
                        }
                        const double ypert = yc + ht_own;
                        const PropT pq = prop_T(ypert);
                        if (need && redo_t && pq.bad && !jbad) { jbad = true; jval = ypert; }   // the reference raises on this perturbed column
                        double kh[3];
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {
                            const bool mine = (zm == cc) && redo_t;
                            kh[cc] = k_upper<ROW>(L, c.Kex_hi, c.dz, c.u2, c.ricrit, c.rihulp, c.supp, c.unsupp, mine ? pq.rho : rhob);
                            X[(X_K + cc) * 64 + lane] = kh[cc];
                        }
                        X[(X_KT + 0) * 64 + lane] = pq.kT;
                        xnh_t[lane] = ht_own;
                        xbad[lane] = jbad ? 1 : 0;
                        __syncthreads();                                  // N1
                        double Dc[3], Sc[3];
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {
                            const bool mine = (zm == cc) && redo_t;
                            const KC kc = kc_make(L, kh[cc], from_lo<ROW, 1>(L, kh[cc]), c.Qv_out);
                            const double fn = row_T<ROW>(L, kc, c.Qv_in, c.T_in, c.UAr_on, c.T_amb, mine ? ypert : yc);
                            Dc[cc] = fn - f; Sc[cc] = fmax(fabs(f), fabs(fn));
                        }
                        by_offset(Dc, Dt); by_offset(Sc, St);
                    } else if (sp == SPH) {
                        const CPH c = fetch_pH(ks);
                        if (!retry) {
                            const PropPH pb = prop_pH(c.k, yc); Hb = pb.H; iwb = pb.iw;
                            X[(X_PHI + 3) * 64 + lane] = pb.phi;
                        }
                        const PropPH pq = prop_pH(c.k, yc + hp_own);
                        X[(X_PHI + 0) * 64 + lane] = pq.phi;
                        xnh_p[lane] = hp_own;
                        __syncthreads();                                  // N1
                        const double *xb = X + (X_K + 3) * 64;
                        const KC kb = kc_make(L, xb[lane], x_lo(xb, lane), c.Qv_out);
                        double Dc[3], Sc[3];
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {                  // pH column: own properties perturbed
                            const bool mine = (zm == cc) && redo_p;
                            const double fn = row_pH<ROW>(L, kb, c.Qv_in, c.H_in, c.acid0, mine ? pq.H : Hb, mine ? pq.iw : iwb);
                            Dc[cc] = fn - f; Sc[cc] = fmax(fabs(f), fabs(fn));
                        }
                        by_offset(Dc, Dp); by_offset(Sc, Sp);
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {                  // T column: the mixing coefficients move
                            const double *xk = X + (X_K + cc) * 64;
                            const KC kc = kc_make(L, xk[lane], x_lo(xk, lane), c.Qv_out);
                            const double fn = row_pH<ROW>(L, kc, c.Qv_in, c.H_in, c.acid0, Hb, iwb);
                            Dc[cc] = fn - f; Sc[cc] = fmax(fabs(f), fabs(fn));
                        }
                        by_offset(Dc, Dt); by_offset(Sc, St);
                        jbad = xbad[lane] != 0;
                    } else {
                        const CCL c = fetch_Cl(ks);
                        __syncthreads();                                  // N1
                        const double *xb = X + (X_K + 3) * 64;
                        const KC kb = kc_make(L, xb[lane], x_lo(xb, lane), c.Qv_out);
                        const double kTbase = X[(X_KT + 3) * 64 + lane], phib = X[(X_PHI + 3) * 64 + lane];
                        {   // pH column: own-zone row only, through the decay factor
                            const double fn = row_Cl<ROW>(L, kb, c.Qv_in, c.Cl_in, c.dose0, kTbase * X[(X_PHI + 0) * 64 + lane], yc);
                            Dp[0] = Dp[2] = 0; Sp[0] = Sp[2] = 0;
                            Dp[1] = fn - f; Sp[1] = fmax(fabs(f), fabs(fn));
                        }
                        double Dc[3], Sc[3];
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {                  // T column: rate of the own zone, mixing coefficients
                            const bool mine = (zm == cc) && redo_t;
                            const double *xk = X + (X_K + cc) * 64;
                            const KC kc = kc_make(L, xk[lane], x_lo(xk, lane), c.Qv_out);
                            const double kphi = (mine ? X[(X_KT + 0) * 64 + lane] : kTbase) * phib;
                            const double fn = row_Cl<ROW>(L, kc, c.Qv_in, c.Cl_in, c.dose0, kphi, yc);
                            Dc[cc] = fn - f; Sc[cc] = fmax(fabs(f), fabs(fn));
                        }
                        by_offset(Dc, Dt); by_offset(Sc, St);
                        jbad = xbad[lane] != 0;
                    }
                    // candidates of this species' rows for the two columns of this lane's zone
                    double md, sc;
                    if (sp != STT) {
                        col_cand(Dp, Sp, sp == SPH, md, sc);
                        X[(X_NM + (0 * 3 + sp) * 2) * 64 + lane] = md; X[(X_NM + (0 * 3 + sp) * 2 + 1) * 64 + lane] = sc;
                    }
                    col_cand(Dt, St, true, md, sc);
                    X[(X_NM + (1 * 3 + sp) * 2) * 64 + lane] = md; X[(X_NM + (1 * 3 + sp) * 2 + 1) * 64 + lane] = sc;
                    h_p = xnh_p[lane]; h_t = xnh_t[lane];
                    __syncthreads();                                      // N2
                };
                auto combine = [&](int col, int nq, double &md, double &sc) {
                    md = -1.0; sc = 0.0;
                    for (int q = 0; q < nq; ++q) {
                        const double m = X[(X_NM + (col * 3 + q) * 2) * 64 + lane], s = X[(X_NM + (col * 3 + q) * 2 + 1) * 64 + lane];
                        if (m > md) { md = m; sc = s; }
                    }
                };
                fd_round(0, true, true, hcol, hcol);
                combine(0, 2, maxd_p, scl_p);
                combine(1, 3, maxd_t, scl_t);
                double Dc_own[3] = {0, 0, 0};          // chlorine column: rows of the chlorine wavefront only
                double maxd_c = -1, scl_c = 0;
                if (sp == SCL) {
                    const CCL c = fetch_Cl(ks);
                    const double *xb = X + (X_K + 3) * 64;
                    const KC kb = kc_make(L, xb[lane], x_lo(xb, lane), c.Qv_out);
                    const double kphib = X[(X_KT + 3) * 64 + lane] * X[(X_PHI + 3) * 64 + lane];
                    while (need && hcol == 0) { fac *= 10; hcol = fd_step(yc, fac, ysc); }
                    double Sc_own[3];
                    auto cl_pass = [&](double hh, bool colmask, double D[3], double S[3]) {
                        double Dc[3], Sc[3];
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {
                            const bool mine = (zm == cc) && colmask;
                            const double fn = row_Cl<ROW>(L, kb, c.Qv_in, c.Cl_in, c.dose0, kphib, mine ? yc + hh : yc);
                            Dc[cc] = fn - f; Sc[cc] = fmax(fabs(f), fabs(fn));
                        }
                        by_offset(Dc, D); by_offset(Sc, S);
                    };
                    cl_pass(hcol, true, Dc_own, Sc_own);
                    col_cand(Dc_own, Sc_own, true, maxd_c, scl_c);
                    const bool small = need && (maxd_c < rc::NJ_REJECT * scl_c);       // common.py:341
                    if (__ballot(small) != 0ull) {                                      // rare: one retry with 10x factor
                        const double nf = 10 * fac;
                        const double hn = fd_step(yc, nf, ysc);
                        double D2[3], S2[3], md2, sc2;
                        cl_pass(hn, small, D2, S2);
                        col_cand(D2, S2, true, md2, sc2);
                        const bool upd = small && (maxd_c * sc2 < md2 * scl_c);         // common.py:354
                        if (upd) { fac = nf; hcol = hn; maxd_c = md2; scl_c = sc2; }
                        const int iu = upd ? 1 : 0;
                        const int iu_lo = __shfl_up(iu, 1, 64), iu_hi = __shfl_down(iu, 1, 64);
                        if (L.has_lo && iu_lo != 0) Dc_own[0] = D2[0];
                        if (upd) Dc_own[1] = D2[1];
                        if (L.has_hi && iu_hi != 0) Dc_own[2] = D2[2];
                    }
                }
                // ---- the one retry with a 10x factor for pH / T columns whose difference drowned in rounding
                const bool small_p = need && (maxd_p < rc::NJ_REJECT * scl_p), small_t = need && (maxd_t < rc::NJ_REJECT * scl_t);
                if (__ballot(small_p || small_t) != 0ull) {
                    const double nf = 10 * fac;
                    const double hn = fd_step(yc, nf, ysc);
                    double Dp1[3], Dt1[3];
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) { Dp1[rr] = Dp[rr]; Dt1[rr] = Dt[rr]; }
                    const double hp1 = h_p, ht1 = h_t;
                    fd_round(1, small_p, small_t, hn, hn);
                    double md2, sc2;
                    combine(0, 2, md2, sc2);
                    const bool upd_p = small_p && (maxd_p * sc2 < md2 * scl_p);         // common.py:354
                    if (upd_p) { maxd_p = md2; scl_p = sc2; }
                    combine(1, 3, md2, sc2);
                    const bool upd_t = small_t && (maxd_t * sc2 < md2 * scl_t);
                    if (upd_t) { maxd_t = md2; scl_t = sc2; }
                    if (sp == SPH && upd_p) fac = nf;
                    if (sp == STT && upd_t) fac = nf;
                    const int iu = (upd_p ? 1 : 0) | (upd_t ? 2 : 0);
                    const int iu_lo = L.has_lo ? __shfl_up(iu, 1, 64) : 0, iu_hi = L.has_hi ? __shfl_down(iu, 1, 64) : 0;
                    // keep the first round's differences (and step) for columns that were not redone or not improved
                    if (!(iu_lo & 1)) Dp[0] = Dp1[0];
                    if (!(iu & 1)) Dp[1] = Dp1[1];
                    if (!(iu_hi & 1)) Dp[2] = Dp1[2];
                    if (!(iu_lo & 2)) Dt[0] = Dt1[0];
                    if (!(iu & 2)) Dt[1] = Dt1[1];
                    if (!(iu_hi & 2)) Dt[2] = Dt1[2];
                    if (!upd_p) h_p = hp1;
                    if (!upd_t) h_t = ht1;
                }
                if (need) {
                    // diff /= h (column-wise; the column's step as its owner published it), factor adaptation common.py:363-365
                    auto quot = [&](double hcolumn, const double D[3], double o[3]) {
                        const double h_lo = from_lo<ROW, 1>(L, hcolumn), h_hi = from_hi<ROW, 1>(L, hcolumn);
                        const double ih0 = L.has_lo ? rcp(h_lo) : 0.0, ih1 = rcp(hcolumn), ih2 = L.has_hi ? rcp(h_hi) : 0.0;
                        o[0] = D[0] * ih0; o[1] = D[1] * ih1; o[2] = D[2] * ih2;
                    };
                    double own_md = maxd_c, own_sc = scl_c;
                    if (sp == SPH) { quot(h_p, Dp, jd); quot(h_t, Dt, jx); own_md = maxd_p; own_sc = scl_p; }
                    else if (sp == SCL) { double o[3]; quot(h_p, Dp, o); jcp = o[1]; quot(hcol, Dc_own, jd); quot(h_t, Dt, jx); }
                    else { quot(h_t, Dt, jd); own_md = maxd_t; own_sc = scl_t; }
                    const bool sm = own_md < rc::NJ_SMALL * own_sc, bg = own_md > rc::NJ_BIG * own_sc;
                    if (sm) fac *= 10;
                    if (bg) fac *= 0.1;
                    fac = fmax(fac, rc::NJ_MINF);
                    need_jac = false;
                    if (seg_any(L, jbad)) {
                        if (jbad && !bad && sp == STT) { badstage = 4; badval = jval; }
                        bad |= jbad; raised = true; phase = PH_DONE;
                    }
                }
            }
        }
        last_cnt = cnt_s;
        if (desync && lane == 0 && a->q_ctrl) __hip_atomic_store(a->q_ctrl + Q_ERROR, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        // ================= after the solve: reactor.py:486-507
        bool clamped = false, post_bad = false;
        if (solving) {
            if (raised) {
                // the reference raised (thermodynamics.py:146-157): self.state untouched; name the temperature its
                // message names -- first evaluation of the trip, lowest zone
                st |= ST_T_RANGE; frozen = true;
                if (sp == STT) {
                    double v = badval; int best = 1 << 30;
#pragma unroll 1
                    for (int sidx = 0; sidx < 5; ++sidx) {
                        const unsigned long long m = __ballot(bad && badstage == sidx) & L.segmask;
                        if (m != 0ull && best == (1 << 30)) { best = sidx; v = __shfl(badval, (int)__builtin_ctzll(m), 64); }
                    }
                    badval = v;
                }
                raised = false;
            } else {
                if (failed) st |= ST_SOLVER_FAILED;    // reactor.py:486-487; state <- last accepted y
                if (limit_hit) st |= ST_STEP_LIMIT;
                y0 = yc;
                stepped = true;
                // _update_derived_state reactor.py:511-524 (before the clamp)
                if (sp == SPH) der0 = exp10(-y0);
                if (sp == STT) {
                    const PropT pt = prop_T(y0);
                    der0 = pt.rho;
                    post_bad = seg_any(L, pt.bad);
                    if (post_bad) {
                        const unsigned long long m = __ballot(pt.bad) & L.segmask;
                        badval = __shfl(y0, (int)__builtin_ctzll(m), 64);
                    } else { der1 = pt.kT; wrote_k = true; }
                }
            }
        }
        // ---- P1: the post-step temperature check and the clamps concern the whole reactor
        {
            double flag = 0.0;
            if (stepped) {
                if (sp == SPH && seg_any(L, y0 < 0 || y0 > 14)) flag = 1.0;
                if (sp == SCL && seg_any(L, y0 < 0)) flag = 1.0;
                if (sp == STT) flag = post_bad ? 2.0 : (seg_any(L, y0 < 0 || y0 > 100) ? 1.0 : 0.0);
            }
            if (L.z == 0) PART(3, sp) = flag;
        }
        __syncthreads();
        if (stepped) {
            steps_done++;
            t_out = t_out + dt;                    // reactor.py:496
            flow_used = ks.uni[15 * ks.stride];    // reactor.py:497-501
            const double fp = PART(3, SPH), fc = PART(3, SCL), ft = PART(3, STT);
            if (ft == 2.0) {
                st |= ST_T_RANGE_POST; frozen = true;
            } else {
                // _enforce_physical_bounds reactor.py:526-541
                if (fp != 0.0) { st |= ST_CLAMP_PH; clamped = true; if (sp == SPH) y0 = fmin(fmax(y0, 0.0), 14.0); }
                if (fc != 0.0) { st |= ST_CLAMP_CL; clamped = true; if (sp == SCL) y0 = fmax(y0, 0.0); }
                if (ft != 0.0) { st |= ST_CLAMP_T; clamped = true; if (sp == STT) y0 = fmin(fmax(y0, 0.0), 100.0); }
                // f(y) of the last accepted point is f0 of the next outer step when nothing touched y
                f_valid = fv && !clamped && !failed;
            }
        }

        // ================= what follows reactor.step() in the reference's loop body (__main__.py:403-423)
        if (sens_on) {
            ArgPtr b = fresh(pa);
            __syncthreads();                 // the exchange slots are dead now; the same LDS carries the hand-off
            if (seg < R) {
                const bool live = stepped && !(st & ST_T_RANGE_POST);     // the reference's loop stops where step() raises
                if (L.z == 0) {
                    if (sp == SCL) { io.stepped[seg] = live ? 1 : 0; io.t_after[seg] = t_out; io.tap[6][seg] = (float)flow_used; }
                    io.tap[2 * sp][seg] = (float)y0;       // tap order: pH in/out, Cl in/out, T in/out  (SPH, SCL, STT = 0, 1, 2)
                }
                if (!L.has_hi) io.tap[2 * sp + 1][seg] = (float)y0;
            }
            __syncthreads();
            if (sp == SCL) wts::suite_step(b->sens, io, rix, R, hist0, k);              // read_all_sensors
            if (plc_on) {
                const int gs = b->first_step + step0 + k;
                const bool scan = ((gs + 1) % b->sens.scan_every == 0) || (gs + 1 == b->call_steps);
                __syncthreads();
                if (sp == SCL && lane < R && io.stepped[lane]) {          // one lane per reactor
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
                        if (sp == SPH) {
                            RK k0; load_reactor(b->par, b->bc, b->N, r, n_zones, k0, &io.cmd[0][seg], wts::RMAX); mask_reactor_for_lane(L, k0);
                            park_reactor(ks, k0);
                        }
                        f_valid = false;
                    }
                }
            }
            __syncthreads();                 // hand-off read; the next step's exchange may overwrite it
        }
    }

    // ================= the item's results
    ArgPtr c = fresh(pa);
    if (present) {
        if (steps_done > 0) {
            double *const so = (sp == SPH) ? c->pH : (sp == SCL ? c->Cl : c->T);
            so[idx] = y0;
            if (sp == SPH) c->dH[idx] = der0;
            if (sp == STT) { c->dRho[idx] = der0; if (wrote_k) c->dK[idx] = der1; }
        }
        if (L.z == 0) {
            if (sp == SCL) {
                if (steps_done > 0) {
                    c->time[r] = t_out;
                    c->flow[r] = flow_used;
                    if (c->stats) {
                        int32_t *o = c->stats + r * 5;
                        o[0] = last_cnt.nfev; o[1] = last_cnt.njev; o[2] = last_cnt.nlu; o[3] = last_cnt.nsteps; o[4] = last_cnt.nrej;
                    }
                    if (sens_on && c->sens.hist_value) c->sens.hist_pos[r] = hist0[seg] + steps_done;
                }
                c->status[r] = st;
            }
            if (sp == STT && (st & (ST_T_RANGE | ST_T_RANGE_POST))) c->bad_T[r] = badval;
        }
    }
    if (want_diag && lane == 0 && sp == SPH) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(c->wave_diag + (int64_t)group * WT_DIAG_SLOTS);
        atomicAdd(o + 0, (unsigned long long)diag_trips); atomicAdd(o + 1, (unsigned long long)diag_newton);
        atomicAdd(o + 2, (unsigned long long)(__builtin_amdgcn_s_memtime() - clk0));
        atomicAdd(o + 3, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - wall0));
        atomicAdd(o + 4, (unsigned long long)diag_fact); atomicAdd(o + 5, (unsigned long long)diag_jac);
        atomicAdd(o + 6, (unsigned long long)diag_f3); atomicAdd(o + 7, 1ull);
    }
}

// The physics kernel, three wavefronts per workgroup.  Scheduling as wt::step_kernel: queue schedule = persistent
// worker workgroups taking (group, next few steps) items from the device-side FIFO; stream schedule = workgroup b
// advances group r0 / R + b.  Wavefront 0 talks to the queue, the hand-off to the other two goes through LDS.
template <int LV, bool ROW>
// (register budget: unconstrained 354; amdgpu_waves_per_eu(2, 2) = 256 spills 170 dwords, (3, 3) = 168 spills 330)
__global__ __launch_bounds__(192) void triad_kernel(const StepArgs a)
{
    __shared__ double lds[Lds3<LV>::TOTAL];
    Lane L;
    lane_geometry(a.n, L);
    const int sp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const ArgPtr pa = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    const bool queue = a.q_ctrl != nullptr;
    int *hand = reinterpret_cast<int *>(lds + Lds3<LV>::CHK0 + 6);      // group, exchanged
    bool exchanged = true;
    int group;
    if (queue) {
        if (sp == 0) { group = queue_next(a, -1, false, exchanged); if ((threadIdx.x & 63) == 0) { hand[0] = group; hand[1] = exchanged ? 1 : 0; } }
        __syncthreads();
        group = hand[0]; exchanged = hand[1] != 0;
    } else {
        group = (int)(a.r0 / a.R) + (int)blockIdx.x;
    }
    while (group >= 0) {
        int step0 = 0, cnt = a.n_steps;
        long long t0 = 0;
        if (queue) {
            if (exchanged) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if ((threadIdx.x & 63) == 0) step0 = __hip_atomic_load(a.q_next + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            step0 = __builtin_amdgcn_readfirstlane(step0);
            const int left = a.n_steps - step0;
            cnt = left < a.item_steps ? left : a.item_steps;
            if (a.trace) t0 = __builtin_amdgcn_s_memrealtime();
        }
        run_item3<LV, ROW>(pa, L, lds, sp, group, step0, cnt);
        if (!queue) break;
        const bool more = step0 + cnt < a.n_steps;
        // every wavefront's stores of the item are out before wavefront 0 may hand the group on
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (sp == 0) {
            if ((threadIdx.x & 63) == 0) {
                __hip_atomic_store(a.q_next + group, step0 + cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.trace) {
                    const int slot = __hip_atomic_fetch_add(a.q_ctrl + Q_TRACE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (slot < a.trace_cap) {
                        int64_t *o = a.trace + (int64_t)slot * 5;
                        o[0] = blockIdx.x; o[1] = group; o[2] = (int64_t)step0 | ((int64_t)cnt << 32); o[3] = t0; o[4] = __builtin_amdgcn_s_memrealtime();
                    }
                }
            }
            group = queue_next(a, more ? group : -1, false, exchanged);
            if ((threadIdx.x & 63) == 0) { hand[0] = group; hand[1] = exchanged ? 1 : 0; }
        }
        __syncthreads();
        group = hand[0]; exchanged = hand[1] != 0;
        __syncthreads();                                                 // (hand[] is rewritten only after everybody has read it)
    }
}

} // namespace tri
} // namespace wt
