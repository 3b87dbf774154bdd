/*
 * wt_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see wt_oracle.h).
 *
 * Restates, operation by operation, what the reference computes for one
 * IntegratedCSTR.step():
 *   - the ODE right-hand side          reactor.py:272-448 (+ thermodynamics.py,
 *                                       chemistry.py, spatial.py call-ees)
 *   - scipy 1.15.3 Radau IIA(5)         scipy/integrate/_ivp/radau.py
 *   - select_initial_step / num_jac     scipy/integrate/_ivp/common.py
 *   - post-step derived + clamps        reactor.py:493-541
 *   - Newton-Raphson pH solver          chemistry.py:193-330
 * Nothing here is copied; the reference is Python, this is a C restatement.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: numpy scalar math
 * does not fuse multiply-adds).
 */
#include "wt_oracle.h"
#include <math.h>
#include <float.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXZ 64
#define MAXM (3 * MAXZ)

static int g_linsolve = 0;
void wto_set_linsolve(int mode) { g_linsolve = mode; }

/* Guard that the reference does not have: give up an outer step after this many step attempts
 * (accepted + rejected + halved).  0 = unlimited = the reference's behaviour.  Mirrors
 * wt_ensemble_set_step_limit of the HIP library so that statuses can be compared. */
static long g_step_limit = 0;
void wto_set_step_limit(long max_attempts) { g_step_limit = max_attempts; }

/* ------------------------------------------------------------------ */
/* RHS: IntegratedCSTR.derivatives  reactor.py:272-448                 */
/* ------------------------------------------------------------------ */

/* SpatialModel.calculate_water_density spatial.py:142-197 (salinity 0) */
static double water_density(double T)
{
    if (T <= 8.0) {
        double d = T - 4.0;
        double delta = -0.008 * (d * d);
        return 999.97 + delta;
    } else {
        double delta = (-2.1e-4 * 998.2) * (T - 20.0);
        return 998.2 + delta;
    }
}

/* TemperatureDependentKinetics.arrhenius_rate thermodynamics.py:160-193 */
static double chlorine_decay_rate(double T)
{
    double T_K = T + 273.15;
    double exponent = -(45000.0 / 8.314) * (1.0 / T_K - 1.0 / 293.15);
    return 0.0001 * exp(exponent);
}

/* AqueousChemistry.buffering_capacity chemistry.py:400-437 */
static double buffering_capacity(const double *par, double pH)
{
    double Kw = par[WTO_P_KW], Ka1 = par[WTO_P_KA1], Ka2 = par[WTO_P_KA2];
    double CT = par[WTO_P_CT_MOL];
    double H = pow(10.0, -pH);
    double beta_water = 2.303 * (H + Kw / H);
    double H2 = H * H;
    double D = H2 + Ka1 * H + Ka1 * Ka2;
    double a0 = H2 / D;
    double a1 = (Ka1 * H) / D;
    double a2 = (Ka1 * Ka2) / D;
    double beta_carb = 2.303 * CT * (a0 * a1 + 4 * a1 * a2 + a0 * a2);
    return beta_water + beta_carb;
}

/* AqueousChemistry.pH_dependent_chlorine_decay_factor chemistry.py:483-523 */
static double decay_factor(const double *par, double pH)
{
    double Ka = par[WTO_P_KA_HOCL];
    double H = pow(10.0, -pH);
    double aH = H / (H + Ka);
    double aO = Ka / (H + Ka);
    return aH * 1.0 + aO * 0.02;
}

int wto_rhs(int n, const double *par, const double *bc, const double *y, double *dydt)
{
    const double *pH = y, *Cl = y + n, *T = y + 2 * n;
    double *dpH = dydt, *dCl = dydt + n, *dT = dydt + 2 * n;
    double rho[MAXZ], sup[MAXZ], kl[MAXZ], kd[MAXZ], ku[MAXZ], H[MAXZ];
    int status = 0;
    const double V = par[WTO_P_VOLUME];
    const double dz = par[WTO_P_HEIGHT] / n; /* SpatialModel.zone_height spatial.py:119 */
    const double u = par[WTO_P_USUP];
    const double Kex = par[WTO_P_KEX];
    const double LN10 = log(10.0);

    for (int i = 0; i < n; i++) { dpH[i] = 0.0; dCl[i] = 0.0; dT[i] = 0.0; }

    /* spatial.update_density_profile reactor.py:304 */
    for (int i = 0; i < n; i++) rho[i] = water_density(T[i]);

    /* mixing suppression reactor.py:307-315, spatial.py:239-320 */
    for (int i = 0; i < n - 1; i++) {
        sup[i] = 1.0;
        if (par[WTO_P_STRAT] != 0.0) {
            double delta_rho = rho[i + 1] - rho[i];
            double rho_avg = 0.5 * (rho[i] + rho[i + 1]);
            double Ri;
            if (u > 1e-6)
                Ri = (9.81 * delta_rho * dz) / (rho_avg * (u * u));
            else
                Ri = INFINITY;
            if (Ri > par[WTO_P_RI_CRIT]) sup[i] = par[WTO_P_SUPP];
        }
    }

    /* K matrix rebuild reactor.py:318-337 (tridiagonal rows kl,kd,ku) */
    const double Qv = (bc[WTO_B_Q_IN] / 60) / V;
    for (int i = 0; i < n; i++) {
        kl[i] = (i > 0) ? Kex * sup[i - 1] : 0.0;
        ku[i] = (i < n - 1) ? Kex * sup[i] : 0.0;
        double off = 0.0;
        if (i > 0) off += kl[i];
        if (i < n - 1) off += ku[i];
        kd[i] = -off;
    }
    kd[n - 1] -= Qv;

/* K_matrix @ x (reactor.py:371,398,423).  numpy hands the dense product to
 * OpenBLAS dgemv, whose even/odd accumulators add the two neighbour terms
 * first and the diagonal term last (measured in the build container against
 * the reference for n = 4, 8, 20; tests/golden/g2_rhs_*.npz pins it). */
#define MATVEC(x, i) \
    ((i) == 0 ? (kd[0] * (x)[0] + ku[0] * (x)[1]) \
     : (i) == n - 1 ? (kl[i] * (x)[(i) - 1] + kd[i] * (x)[i]) \
     : ((kl[i] * (x)[(i) - 1] + ku[i] * (x)[(i) + 1]) + kd[i] * (x)[i]))

    /* --- pH dynamics reactor.py:346-376 --- */
    for (int i = 0; i < n; i++) H[i] = pow(10.0, -pH[i]);
    const double zone_volume_L = V / n;
    if (bc[WTO_B_Q_ACID] > 0) {
        double H_added = (bc[WTO_B_Q_ACID] / 60) * bc[WTO_B_C_ACID];
        double dH = H_added / zone_volume_L;
        double beta = buffering_capacity(par, pH[0]);
        if (beta > 0) dpH[0] += -dH / (beta * LN10);
    }
    {
        double H_inlet = pow(10.0, -bc[WTO_B_PH_IN]);
        double dH_inlet = Qv * (H_inlet - H[0]);
        double beta0 = buffering_capacity(par, pH[0]);
        if (beta0 > 0) dpH[0] += -dH_inlet / (beta0 * LN10);
    }
    for (int i = 0; i < n; i++) {
        double dHmix = MATVEC(H, i);
        double beta = buffering_capacity(par, pH[i]);
        if (beta > 0) dpH[i] += -dHmix / (beta * LN10);
    }

    /* --- chlorine dynamics reactor.py:385-411 --- */
    if (bc[WTO_B_Q_CL] > 0) {
        double Cl_added = (bc[WTO_B_Q_CL] / 60) * bc[WTO_B_C_CL];
        dCl[0] += Cl_added / zone_volume_L;
    }
    dCl[0] += Qv * (bc[WTO_B_CL_IN] - Cl[0]);
    for (int i = 0; i < n; i++) dCl[i] += MATVEC(Cl, i);
    for (int i = 0; i < n; i++) {
        if (T[i] < 0.0 || T[i] > 100.0) status |= WTO_ST_T_RANGE; /* thermodynamics.py:146-157 */
        double k_base = chlorine_decay_rate(T[i]);
        double f = decay_factor(par, pH[i]);
        double k_eff = k_base * f;
        dCl[i] -= k_eff * Cl[i];
    }

    /* --- temperature dynamics reactor.py:420-443 --- */
    dT[0] += Qv * (bc[WTO_B_T_IN] - T[0]);
    for (int i = 0; i < n; i++) dT[i] += MATVEC(T, i);
    if (bc[WTO_B_U] > 0) {
        double D = par[WTO_P_DIAMETER], Hh = par[WTO_P_HEIGHT];
        double A_lat = M_PI * D * Hh;
        double r = D / 2;
        double A_ends = 2 * M_PI * (r * r);
        double A_tot = A_lat + A_ends;
        double V_m3 = V / 1000;
        double rcv = 998.2 * 4184 * V_m3;
        for (int i = 0; i < n; i++) {
            double Q_loss = bc[WTO_B_U] * A_tot * (T[i] - bc[WTO_B_T_AMB]);
            dT[i] -= Q_loss / rcv;
        }
    }
#undef MATVEC
    return status;
}

/* ------------------------------------------------------------------ */
/* scipy Radau restatement                                             */
/* ------------------------------------------------------------------ */

#define EPS DBL_EPSILON
static const double RTOL = 1e-6, ATOL = 1e-8; /* reactor.py:481-483 */
#define NEWTON_MAXITER 6
#define MIN_FACTOR 0.2
#define MAX_FACTOR 10.0

typedef struct {
    int n, m;
    const double *par, *bc;
    int nfev, njev, nlu, nrhs;
    int abort_status; /* non-zero => the reference would have raised */
    long attempts;    /* step attempts of this outer step */
    int limit_hit;
} rctx;

/* constants radau.py:11-40, computed the way the module computes them */
static double S6, C_[3], E_[3], MU_REAL, MU_CR, MU_CI;
static double T_[3][3], TI_[3][3], P_[3][3];
static int consts_ready = 0;
static void init_consts(void)
{
    if (consts_ready) return;
    S6 = pow(6.0, 0.5);
    C_[0] = (4 - S6) / 10; C_[1] = (4 + S6) / 10; C_[2] = 1.0;
    E_[0] = (-13 - 7 * S6) / 3; E_[1] = (-13 + 7 * S6) / 3; E_[2] = -1.0 / 3;
    MU_REAL = 3 + pow(3.0, 2.0 / 3) - pow(3.0, 1.0 / 3);
    MU_CR = 3 + 0.5 * (pow(3.0, 1.0 / 3) - pow(3.0, 2.0 / 3));
    MU_CI = -0.5 * (pow(3.0, 5.0 / 6) + pow(3.0, 7.0 / 6));
    double Tm[3][3] = {{0.09443876248897524, -0.14125529502095421, 0.03002919410514742},
                       {0.25021312296533332, 0.20412935229379994, -0.38294211275726192},
                       {1, 1, 0}};
    double TIm[3][3] = {{4.17871859155190428, 0.32768282076106237, 0.52337644549944951},
                        {-4.17871859155190428, -0.32768282076106237, 0.47662355450055044},
                        {0.50287263494578682, -2.57192694985560522, 0.59603920482822492}};
    double Pm[3][3] = {{13.0 / 3 + 7 * S6 / 3, -23.0 / 3 - 22 * S6 / 3, 10.0 / 3 + 5 * S6},
                       {13.0 / 3 - 7 * S6 / 3, -23.0 / 3 + 22 * S6 / 3, 10.0 / 3 - 5 * S6},
                       {1.0 / 3, -8.0 / 3, 10.0 / 3}};
    memcpy(T_, Tm, sizeof Tm); memcpy(TI_, TIm, sizeof TIm); memcpy(P_, Pm, sizeof Pm);
    consts_ready = 1;
}

/* common.py:63-65 */
static double rms_norm(const double *x, int cnt)
{
    double s = 0.0;
    for (int i = 0; i < cnt; i++) s += x[i] * x[i];
    return sqrt(s) / sqrt((double)cnt);
}

static void fun_raw(rctx *c, const double *y, double *f)
{
    int st = wto_rhs(c->n, c->par, c->bc, y, f);
    c->nrhs++;
    if (st) c->abort_status |= st;
}
static void fun(rctx *c, const double *y, double *f) /* base.py:133-135 counts nfev */
{
    c->nfev++;
    fun_raw(c, y, f);
}

/* common.py:68-134 */
static double select_initial_step(rctx *c, const double *y0, const double *f0,
                                  double interval, double max_step)
{
    int m = c->m;
    double scale[MAXM], tmp[MAXM], y1[MAXM], f1[MAXM];
    if (interval == 0.0) return 0.0;
    for (int i = 0; i < m; i++) scale[i] = ATOL + fabs(y0[i]) * RTOL;
    for (int i = 0; i < m; i++) tmp[i] = y0[i] / scale[i];
    double d0 = rms_norm(tmp, m);
    for (int i = 0; i < m; i++) tmp[i] = f0[i] / scale[i];
    double d1 = rms_norm(tmp, m);
    double h0;
    if (d0 < 1e-5 || d1 < 1e-5) h0 = 1e-6; else h0 = 0.01 * d0 / d1;
    h0 = fmin(h0, interval);
    for (int i = 0; i < m; i++) y1[i] = y0[i] + h0 * 1.0 * f0[i];
    fun(c, y1, f1);
    if (c->abort_status) return 0.0;
    for (int i = 0; i < m; i++) tmp[i] = (f1[i] - f0[i]) / scale[i];
    double d2 = rms_norm(tmp, m) / h0;
    double h1;
    if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
    else h1 = pow(0.01 / fmax(d1, d2), 1.0 / (3 + 1));
    return fmin(fmin(100 * h0, h1), fmin(interval, max_step));
}

/* common.py:257-382 (dense branch). J is row-major m x m: J[i*m+j]=df_i/dy_j */
static void num_jac(rctx *c, const double *y, const double *f, double *J,
                    double *factor, int *have_factor)
{
    const double REJECT = pow(EPS, 0.875), SMALL = pow(EPS, 0.75), BIG = pow(EPS, 0.25);
    const double MINF = 1e3 * EPS;
    int m = c->m;
    double y_scale[MAXM], h[MAXM], max_diff[MAXM], scale[MAXM];
    double ycol[MAXM], fnew[MAXM];
    static __thread double diff[MAXM * MAXM];
    c->njev++;
    if (!*have_factor) {
        double f0 = pow(EPS, 0.5);
        for (int i = 0; i < m; i++) factor[i] = f0;
        *have_factor = 1;
    }
    for (int i = 0; i < m; i++) {
        double fs = (f[i] >= 0) ? 1.0 : -1.0;
        y_scale[i] = fs * fmax(ATOL, fabs(y[i]));
        h[i] = (y[i] + factor[i] * y_scale[i]) - y[i];
        while (h[i] == 0) {
            factor[i] *= 10;
            h[i] = (y[i] + factor[i] * y_scale[i]) - y[i];
        }
    }
    int any_small = 0;
    for (int j = 0; j < m; j++) {
        memcpy(ycol, y, sizeof(double) * m);
        ycol[j] = y[j] + h[j];
        fun_raw(c, ycol, fnew);
        if (c->abort_status) return;
        int mi = 0; double md = -1.0;
        for (int i = 0; i < m; i++) {
            double d = fnew[i] - f[i];
            diff[i * m + j] = d;
            if (fabs(d) > md) { md = fabs(d); mi = i; }
        }
        max_diff[j] = md;
        scale[j] = fmax(fabs(f[mi]), fabs(fnew[mi]));
        if (md < REJECT * scale[j]) any_small = 1;
    }
    if (any_small) {
        for (int j = 0; j < m; j++) {
            if (!(max_diff[j] < REJECT * scale[j])) continue;
            double new_factor = 10 * factor[j];
            double h_new = (y[j] + new_factor * y_scale[j]) - y[j];
            memcpy(ycol, y, sizeof(double) * m);
            ycol[j] = y[j] + h_new;
            fun_raw(c, ycol, fnew);
            if (c->abort_status) return;
            int mi = 0; double md = -1.0;
            double dnew[MAXM];
            for (int i = 0; i < m; i++) {
                dnew[i] = fnew[i] - f[i];
                if (fabs(dnew[i]) > md) { md = fabs(dnew[i]); mi = i; }
            }
            double scale_new = fmax(fabs(f[mi]), fabs(fnew[mi]));
            if (max_diff[j] * scale_new < md * scale[j]) {
                factor[j] = new_factor;
                h[j] = h_new;
                for (int i = 0; i < m; i++) diff[i * m + j] = dnew[i];
                scale[j] = scale_new;
                max_diff[j] = md;
            }
        }
    }
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) J[i * m + j] = diff[i * m + j] / h[j];
    for (int j = 0; j < m; j++) {
        int small = max_diff[j] < SMALL * scale[j];
        int big = max_diff[j] > BIG * scale[j];
        if (small) factor[j] *= 10;
        if (big) factor[j] *= 0.1;
        factor[j] = fmax(factor[j], MINF);
    }
}

/* ---- dense LU with partial pivoting (what lu_factor/lu_solve do) ---- */
typedef struct {
    int m;
    double a[MAXM * MAXM];   /* real LU */
    int piv[MAXM];
    double cr[MAXM * MAXM], ci[MAXM * MAXM]; /* complex LU */
    int pivc[MAXM];
    /* structured variant: tridiagonal blocks (see struct_factor) */
    double h_used;
} lu_t;

static void lu_real_factor(lu_t *L, const double *J, double mu_over_h)
{
    int m = L->m; double *a = L->a;
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++)
            a[i * m + j] = mu_over_h * (i == j ? 1.0 : 0.0) - J[i * m + j];
    for (int k = 0; k < m; k++) {
        int p = k; double best = fabs(a[k * m + k]);
        for (int i = k + 1; i < m; i++)
            if (fabs(a[i * m + k]) > best) { best = fabs(a[i * m + k]); p = i; }
        L->piv[k] = p;
        if (p != k)
            for (int j = 0; j < m; j++) { double t = a[k * m + j]; a[k * m + j] = a[p * m + j]; a[p * m + j] = t; }
        double d = a[k * m + k];
        if (d != 0.0) {
            double r = 1.0 / d;
            for (int i = k + 1; i < m; i++) a[i * m + k] *= r;
        }
        for (int i = k + 1; i < m; i++) {
            double l = a[i * m + k];
            if (l != 0.0)
                for (int j = k + 1; j < m; j++) a[i * m + j] -= l * a[k * m + j];
        }
    }
}
static void lu_real_solve(const lu_t *L, double *b)
{
    int m = L->m; const double *a = L->a;
    for (int k = 0; k < m; k++) { int p = L->piv[k]; if (p != k) { double t = b[k]; b[k] = b[p]; b[p] = t; } }
    for (int i = 1; i < m; i++) { double s = b[i]; for (int j = 0; j < i; j++) s -= a[i * m + j] * b[j]; b[i] = s; }
    for (int i = m - 1; i >= 0; i--) { double s = b[i]; for (int j = i + 1; j < m; j++) s -= a[i * m + j] * b[j]; b[i] = s / a[i * m + i]; }
}
static inline void cdiv(double ar, double ai, double br, double bi, double *cr, double *ci)
{   /* Smith's algorithm */
    if (fabs(br) >= fabs(bi)) { double r = bi / br, d = br + bi * r; *cr = (ar + ai * r) / d; *ci = (ai - ar * r) / d; }
    else { double r = br / bi, d = br * r + bi; *cr = (ar * r + ai) / d; *ci = (ai * r - ar) / d; }
}
static void lu_cplx_factor(lu_t *L, const double *J, double mr, double mi)
{
    int m = L->m; double *ar = L->cr, *ai = L->ci;
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) {
            ar[i * m + j] = mr * (i == j ? 1.0 : 0.0) - J[i * m + j];
            ai[i * m + j] = mi * (i == j ? 1.0 : 0.0);
        }
    for (int k = 0; k < m; k++) {
        int p = k; double best = fabs(ar[k * m + k]) + fabs(ai[k * m + k]);
        for (int i = k + 1; i < m; i++) {
            double v = fabs(ar[i * m + k]) + fabs(ai[i * m + k]);
            if (v > best) { best = v; p = i; }
        }
        L->pivc[k] = p;
        if (p != k)
            for (int j = 0; j < m; j++) {
                double t = ar[k * m + j]; ar[k * m + j] = ar[p * m + j]; ar[p * m + j] = t;
                t = ai[k * m + j]; ai[k * m + j] = ai[p * m + j]; ai[p * m + j] = t;
            }
        double dr = ar[k * m + k], di = ai[k * m + k];
        if (dr != 0.0 || di != 0.0) {
            double rr, ri; cdiv(1.0, 0.0, dr, di, &rr, &ri);
            for (int i = k + 1; i < m; i++) {
                double xr = ar[i * m + k], xi = ai[i * m + k];
                ar[i * m + k] = xr * rr - xi * ri; ai[i * m + k] = xr * ri + xi * rr;
            }
        }
        for (int i = k + 1; i < m; i++) {
            double lr = ar[i * m + k], li = ai[i * m + k];
            if (lr != 0.0 || li != 0.0)
                for (int j = k + 1; j < m; j++) {
                    double ur = ar[k * m + j], ui = ai[k * m + j];
                    ar[i * m + j] -= lr * ur - li * ui;
                    ai[i * m + j] -= lr * ui + li * ur;
                }
        }
    }
}
static void lu_cplx_solve(const lu_t *L, double *br, double *bi)
{
    int m = L->m; const double *ar = L->cr, *ai = L->ci;
    for (int k = 0; k < m; k++) { int p = L->pivc[k]; if (p != k) { double t = br[k]; br[k] = br[p]; br[p] = t; t = bi[k]; bi[k] = bi[p]; bi[p] = t; } }
    for (int i = 1; i < m; i++) {
        double sr = br[i], si = bi[i];
        for (int j = 0; j < i; j++) { double lr = ar[i * m + j], li = ai[i * m + j]; sr -= lr * br[j] - li * bi[j]; si -= lr * bi[j] + li * br[j]; }
        br[i] = sr; bi[i] = si;
    }
    for (int i = m - 1; i >= 0; i--) {
        double sr = br[i], si = bi[i];
        for (int j = i + 1; j < m; j++) { double ur = ar[i * m + j], ui = ai[i * m + j]; sr -= ur * br[j] - ui * bi[j]; si -= ur * bi[j] + ui * br[j]; }
        cdiv(sr, si, ar[i * m + i], ai[i * m + i], &br[i], &bi[i]);
    }
}

/* ---- structured variant (mode 1): the linear algebra the HIP kernel uses.
 * With unknown order [T | pH | Cl] the Jacobian is block lower triangular with
 * tridiagonal diagonal blocks:
 *   dT  depends on T only; dpH on pH and T; dCl on Cl, pH (diagonal) and T.
 * Solve (mu/h I - J) x = b by three tridiagonal solves.  Used to check that
 * the GPU's solve is the same mathematics as scipy's dense LU. */
static void tri_solve_real(int n, const double *a, const double *d, const double *cc, double *b)
{
    double cp[MAXZ], dp[MAXZ];
    cp[0] = cc[0] / d[0]; dp[0] = b[0] / d[0];
    for (int i = 1; i < n; i++) {
        double den = d[i] - a[i] * cp[i - 1];
        cp[i] = cc[i] / den;
        dp[i] = (b[i] - a[i] * dp[i - 1]) / den;
    }
    b[n - 1] = dp[n - 1];
    for (int i = n - 2; i >= 0; i--) b[i] = dp[i] - cp[i] * b[i + 1];
}
static void tri_solve_cplx(int n, const double *a, const double *dr, double di, const double *cc,
                           double *br, double *bi)
{   /* off-diagonals real, diagonal dr[i] + i*di */
    double cpr[MAXZ], cpi[MAXZ], dpr[MAXZ], dpi[MAXZ];
    cdiv(cc[0], 0.0, dr[0], di, &cpr[0], &cpi[0]);
    cdiv(br[0], bi[0], dr[0], di, &dpr[0], &dpi[0]);
    for (int i = 1; i < n; i++) {
        double denr = dr[i] - a[i] * cpr[i - 1], deni = di - a[i] * cpi[i - 1];
        cdiv(cc[i], 0.0, denr, deni, &cpr[i], &cpi[i]);
        cdiv(br[i] - a[i] * dpr[i - 1], bi[i] - a[i] * dpi[i - 1], denr, deni, &dpr[i], &dpi[i]);
    }
    br[n - 1] = dpr[n - 1]; bi[n - 1] = dpi[n - 1];
    for (int i = n - 2; i >= 0; i--) {
        double xr = br[i + 1], xi = bi[i + 1];
        br[i] = dpr[i] - (cpr[i] * xr - cpi[i] * xi);
        bi[i] = dpi[i] - (cpr[i] * xi + cpi[i] * xr);
    }
}
/* x = (mr + i*mi) I - J applied inverse on b (complex if mi != 0), J dense row-major,
 * state order [pH(0..n) | Cl(n..2n) | T(2n..3n)] */
static void struct_solve(int n, const double *J, double mr, double mi, double *br, double *bi)
{
    int m = 3 * n;
    double a[MAXZ], d[MAXZ], cc[MAXZ];
    int cplx = (bi != NULL);
#define JJ(r, c) J[(r) * m + (c)]
    /* 1. T block */
    for (int i = 0; i < n; i++) {
        a[i] = (i > 0) ? -JJ(2 * n + i, 2 * n + i - 1) : 0.0;
        cc[i] = (i < n - 1) ? -JJ(2 * n + i, 2 * n + i + 1) : 0.0;
        d[i] = mr - JJ(2 * n + i, 2 * n + i);
    }
    if (cplx) tri_solve_cplx(n, a, d, mi, cc, br + 2 * n, bi + 2 * n); else tri_solve_real(n, a, d, cc, br + 2 * n);
    /* 2. pH block: rhs += J_pT x_T */
    for (int i = 0; i < n; i++) {
        for (int j = (i > 0 ? i - 1 : 0); j <= (i < n - 1 ? i + 1 : n - 1); j++) {
            br[i] += JJ(i, 2 * n + j) * br[2 * n + j];
            if (cplx) bi[i] += JJ(i, 2 * n + j) * bi[2 * n + j];
        }
        a[i] = (i > 0) ? -JJ(i, i - 1) : 0.0;
        cc[i] = (i < n - 1) ? -JJ(i, i + 1) : 0.0;
        d[i] = mr - JJ(i, i);
    }
    if (cplx) tri_solve_cplx(n, a, d, mi, cc, br, bi); else tri_solve_real(n, a, d, cc, br);
    /* 3. Cl block: rhs += J_cT x_T + J_cp x_p */
    for (int i = 0; i < n; i++) {
        for (int j = (i > 0 ? i - 1 : 0); j <= (i < n - 1 ? i + 1 : n - 1); j++) {
            br[n + i] += JJ(n + i, 2 * n + j) * br[2 * n + j];
            if (cplx) bi[n + i] += JJ(n + i, 2 * n + j) * bi[2 * n + j];
        }
        br[n + i] += JJ(n + i, i) * br[i];
        if (cplx) bi[n + i] += JJ(n + i, i) * bi[i];
        a[i] = (i > 0) ? -JJ(n + i, n + i - 1) : 0.0;
        cc[i] = (i < n - 1) ? -JJ(n + i, n + i + 1) : 0.0;
        d[i] = mr - JJ(n + i, n + i);
    }
    if (cplx) tri_solve_cplx(n, a, d, mi, cc, br + n, bi + n); else tri_solve_real(n, a, d, cc, br + n);
#undef JJ
}

/* persistent solver state of one scipy Radau object (radau.py:295-346) */
typedef struct {
    double t, y[MAXM], f[MAXM];
    double h_abs, h_abs_old, error_norm_old;
    int have_old; /* h_abs_old/error_norm_old not None */
    double J[MAXM * MAXM];
    double jac_factor[MAXM]; int have_factor;
    int current_jac;
    int have_lu; double lu_h;
    int have_sol; double sol_t_old, sol_h; double sol_y_old[MAXM]; double Q[MAXM][3];
    double Z[3][MAXM];
} radau_t;

static void do_factor(rctx *c, lu_t *L, const radau_t *R, double h)
{
    L->m = c->m; L->h_used = h;
    if (g_linsolve == 0) {
        lu_real_factor(L, R->J, MU_REAL / h);
        lu_cplx_factor(L, R->J, MU_CR / h, MU_CI / h);
    }
    c->nlu += 2;
}
static void solve_real(rctx *c, const lu_t *L, const radau_t *R, double *b)
{
    if (g_linsolve == 0) lu_real_solve(L, b);
    else struct_solve(c->n, R->J, MU_REAL / L->h_used, 0.0, b, NULL);
}
static void solve_cplx(rctx *c, const lu_t *L, const radau_t *R, double *br, double *bi)
{
    if (g_linsolve == 0) lu_cplx_solve(L, br, bi);
    else struct_solve(c->n, R->J, MU_CR / L->h_used, MU_CI / L->h_used, br, bi);
}

/* radau.py:48-136 */
static int solve_collocation_system(rctx *c, const radau_t *R, const double *y, double h,
                                    double Z0[3][MAXM], const double *scale, double tol,
                                    const lu_t *L, double Z[3][MAXM], int *n_iter, double *rate_out,
                                    int *have_rate)
{
    int m = c->m;
    double M_real = MU_REAL / h, Mcr = MU_CR / h, Mci = MU_CI / h;
    double W[3][MAXM], F[3][MAXM], dW[3][MAXM], ytmp[MAXM], tmp[3 * MAXM];
    for (int s = 0; s < 3; s++)
        for (int i = 0; i < m; i++)
            W[s][i] = TI_[s][0] * Z0[0][i] + TI_[s][1] * Z0[1][i] + TI_[s][2] * Z0[2][i];
    for (int s = 0; s < 3; s++) memcpy(Z[s], Z0[s], sizeof(double) * m);
    double dW_norm_old = 0, rate = 0; int have_norm_old = 0; *have_rate = 0;
    int converged = 0, k;
    for (k = 0; k < NEWTON_MAXITER; k++) {
        for (int s = 0; s < 3; s++) {
            for (int i = 0; i < m; i++) ytmp[i] = y[i] + Z[s][i];
            fun(c, ytmp, F[s]);
            if (c->abort_status) { *n_iter = k + 1; return 0; }
        }
        int finite = 1;
        for (int s = 0; s < 3 && finite; s++)
            for (int i = 0; i < m; i++) if (!isfinite(F[s][i])) { finite = 0; break; }
        if (!finite) break;
        double f_real[MAXM], f_cr[MAXM], f_ci[MAXM];
        for (int i = 0; i < m; i++) {
            f_real[i] = (F[0][i] * TI_[0][0] + F[1][i] * TI_[0][1] + F[2][i] * TI_[0][2]) - M_real * W[0][i];
            double re = F[0][i] * TI_[1][0] + F[1][i] * TI_[1][1] + F[2][i] * TI_[1][2];
            double im = F[0][i] * TI_[2][0] + F[1][i] * TI_[2][1] + F[2][i] * TI_[2][2];
            /* M_complex * (W1 + i W2) */
            double pr = Mcr * W[1][i] - Mci * W[2][i];
            double pi = Mcr * W[2][i] + Mci * W[1][i];
            f_cr[i] = re - pr; f_ci[i] = im - pi;
        }
        solve_real(c, L, R, f_real);
        solve_cplx(c, L, R, f_cr, f_ci);
        for (int i = 0; i < m; i++) { dW[0][i] = f_real[i]; dW[1][i] = f_cr[i]; dW[2][i] = f_ci[i]; }
        for (int s = 0; s < 3; s++) for (int i = 0; i < m; i++) tmp[s * m + i] = dW[s][i] / scale[i];
        double dW_norm = rms_norm(tmp, 3 * m);
        if (have_norm_old) { rate = dW_norm / dW_norm_old; *have_rate = 1; }
        if (*have_rate && (rate >= 1 || pow(rate, NEWTON_MAXITER - k) / (1 - rate) * dW_norm > tol)) break;
        for (int s = 0; s < 3; s++) for (int i = 0; i < m; i++) W[s][i] += dW[s][i];
        for (int s = 0; s < 3; s++)
            for (int i = 0; i < m; i++)
                Z[s][i] = T_[s][0] * W[0][i] + T_[s][1] * W[1][i] + T_[s][2] * W[2][i];
        if (dW_norm == 0 || (*have_rate && rate / (1 - rate) * dW_norm < tol)) { converged = 1; break; }
        dW_norm_old = dW_norm; have_norm_old = 1;
    }
    /* python: `return converged, k + 1, Z, rate`; after a full loop k==MAXITER-1 */
    if (k == NEWTON_MAXITER) k = NEWTON_MAXITER - 1;
    *n_iter = k + 1; *rate_out = rate;
    return converged;
}

/* radau.py:139-176 */
static double predict_factor(double h_abs, int have_old, double h_abs_old, double error_norm, double error_norm_old)
{
    double multiplier;
    if (!have_old || error_norm == 0) multiplier = 1;
    else multiplier = h_abs / h_abs_old * pow(error_norm_old / error_norm, 0.25);
    return fmin(1.0, multiplier) * pow(error_norm, -0.25);
}

/* radau.py:399-539; returns 1 success, 0 TOO_SMALL_STEP, -1 aborted (exception) */
static int step_impl(rctx *c, radau_t *R, lu_t *L, double t_bound, double max_step, double newton_tol, wto_stats *st)
{
    int m = c->m;
    double t = R->t; double *y = R->y, *f = R->f;
    double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
    double h_abs, h_abs_old = 0, error_norm_old = 0; int have_old;
    if (R->h_abs > max_step) { h_abs = max_step; have_old = 0; }
    else if (R->h_abs < min_step) { h_abs = min_step; have_old = 0; }
    else { h_abs = R->h_abs; have_old = R->have_old; h_abs_old = R->h_abs_old; error_norm_old = R->error_norm_old; }

    int rejected = 0, step_accepted = 0;
    double Z0[3][MAXM], Z[3][MAXM], scale[MAXM], y_new[MAXM], error[MAXM], ZE[MAXM], tmp[MAXM];
    double h = 0, t_new = 0, error_norm = 0, safety = 0, rate = 0; int n_iter = 0, have_rate = 0;
    while (!step_accepted) {
        if (g_step_limit > 0 && c->attempts >= g_step_limit) { c->limit_hit = 1; return 0; }
        c->attempts++;
        if (h_abs < min_step) return 0;
        h = h_abs * 1.0;
        t_new = t + h;
        if (1.0 * (t_new - t_bound) > 0) t_new = t_bound;
        h = t_new - t;
        h_abs = fabs(h);
        if (!R->have_sol) {
            for (int s = 0; s < 3; s++) for (int i = 0; i < m; i++) Z0[s][i] = 0.0;
        } else {
            /* RadauDenseOutput._call_impl radau.py:557-572 at t + h*C */
            for (int s = 0; s < 3; s++) {
                double x = ((t + h * C_[s]) - R->sol_t_old) / R->sol_h;
                double p0 = x, p1 = x * x, p2 = p1 * x;
                for (int i = 0; i < m; i++) {
                    double v = R->Q[i][0] * p0 + R->Q[i][1] * p1 + R->Q[i][2] * p2;
                    v += R->sol_y_old[i];
                    Z0[s][i] = v - y[i];
                }
            }
        }
        for (int i = 0; i < m; i++) scale[i] = ATOL + fabs(y[i]) * RTOL;
        int converged = 0;
        while (!converged) {
            if (!R->have_lu) { do_factor(c, L, R, h); R->have_lu = 1; }
            converged = solve_collocation_system(c, R, y, h, Z0, scale, newton_tol, L, Z, &n_iter, &rate, &have_rate);
            if (c->abort_status) return -1;
            if (!converged) {
                if (R->current_jac) break;
                num_jac(c, y, f, R->J, R->jac_factor, &R->have_factor);
                if (c->abort_status) return -1;
                R->current_jac = 1;
                R->have_lu = 0;
            }
        }
        if (!converged) {
            h_abs *= 0.5;
            R->have_lu = 0;
            if (st) st->nrej++;
            continue;
        }
        for (int i = 0; i < m; i++) y_new[i] = y[i] + Z[2][i];
        for (int i = 0; i < m; i++) ZE[i] = (Z[0][i] * E_[0] + Z[1][i] * E_[1] + Z[2][i] * E_[2]) / h;
        for (int i = 0; i < m; i++) error[i] = f[i] + ZE[i];
        solve_real(c, L, R, error);
        for (int i = 0; i < m; i++) scale[i] = ATOL + fmax(fabs(y[i]), fabs(y_new[i])) * RTOL;
        for (int i = 0; i < m; i++) tmp[i] = error[i] / scale[i];
        error_norm = rms_norm(tmp, m);
        safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + n_iter);
        if (rejected && error_norm > 1) {
            double yt[MAXM], ft[MAXM];
            for (int i = 0; i < m; i++) yt[i] = y[i] + error[i];
            fun(c, yt, ft);
            if (c->abort_status) return -1;
            for (int i = 0; i < m; i++) error[i] = ft[i] + ZE[i];
            solve_real(c, L, R, error);
            for (int i = 0; i < m; i++) tmp[i] = error[i] / scale[i];
            error_norm = rms_norm(tmp, m);
        }
        if (error_norm > 1) {
            double factor = predict_factor(h_abs, have_old, h_abs_old, error_norm, error_norm_old);
            h_abs *= fmax(MIN_FACTOR, safety * factor);
            R->have_lu = 0;
            rejected = 1;
            if (st) st->nrej++;
        } else
            step_accepted = 1;
    }
    int recompute_jac = (n_iter > 2) && have_rate && (rate > 1e-3);
    /* python: `n_iter > 2 and rate > 1e-3`; n_iter>2 implies rate is set */
    double factor = predict_factor(h_abs, have_old, h_abs_old, error_norm, error_norm_old);
    factor = fmin(MAX_FACTOR, safety * factor);
    if (!recompute_jac && factor < 1.2) factor = 1;
    else R->have_lu = 0;
    double f_new[MAXM];
    fun(c, y_new, f_new);
    if (c->abort_status) return -1;
    if (recompute_jac) {
        num_jac(c, y_new, f_new, R->J, R->jac_factor, &R->have_factor);
        if (c->abort_status) return -1;
        R->current_jac = 1;
    } else
        R->current_jac = 0;
    R->h_abs_old = R->h_abs;       /* sic: the solver-level value (radau.py:520) */
    R->error_norm_old = error_norm;
    R->have_old = 1;
    R->h_abs = h_abs * factor;
    /* dense output radau.py:541-543: Q = Z.T . P */
    memcpy(R->sol_y_old, y, sizeof(double) * m);
    for (int i = 0; i < m; i++)
        for (int k = 0; k < 3; k++)
            R->Q[i][k] = Z[0][i] * P_[0][k] + Z[1][i] * P_[1][k] + Z[2][i] * P_[2][k];
    R->sol_t_old = t; R->sol_h = t_new - t; R->have_sol = 1;
    R->t = t_new;
    memcpy(R->y, y_new, sizeof(double) * m);
    memcpy(R->f, f_new, sizeof(double) * m);
    return 1;
}

int wto_step(int n, const double *par, const double *bc, double dt,
             double *y, double *t, double *derived, wto_stats *st)
{
    init_consts();
    int m = 3 * n;
    int status = 0;
    rctx c; memset(&c, 0, sizeof c);
    c.n = n; c.m = m; c.par = par; c.bc = bc;
    if (st) memset(st, 0, sizeof *st);
    static __thread radau_t R; static __thread lu_t L;
    R.have_old = 0; R.have_factor = 0; R.have_lu = 0; R.have_sol = 0; R.current_jac = 1;

    /* scipy refuses a non-finite initial state (base.py:19-20 check_arguments: ValueError "All components
     * of the initial state `y0` must be finite."); it escapes step(), self.state and the clock stay as they are */
    for (int i = 0; i < m; i++) if (!isfinite(y[i])) return WTO_ST_NONFINITE;

    double t0 = *t, t_bound = *t + dt;           /* reactor.py:472 */
    double max_step = fmin(dt, 10.0);            /* reactor.py:480 */
    double newton_tol = fmax(10 * EPS / RTOL, fmin(0.03, pow(RTOL, 0.5)));
    int solver_failed = 0, aborted = 0;

    R.t = t0; memcpy(R.y, y, sizeof(double) * m);
    if (t0 != t_bound) {
        fun(&c, R.y, R.f);                                         /* radau.py:303 */
        if (!c.abort_status)
            R.h_abs = select_initial_step(&c, R.y, R.f, fabs(t_bound - t0), max_step); /* :307 */
        if (!c.abort_status)
            num_jac(&c, R.y, R.f, R.J, R.jac_factor, &R.have_factor);  /* :359-365 */
        if (c.abort_status) aborted = 1;
        while (!aborted && (R.t - t_bound) < 0) {                   /* base.py:182-197 */
            int ok = step_impl(&c, &R, &L, t_bound, max_step, newton_tol, st);
            if (ok < 0) { aborted = 1; break; }
            if (ok == 0) { solver_failed = 1; break; }
            if (st) { if (st->nsteps < 64) st->t_internal[st->nsteps] = R.t; st->nsteps++; }
        }
    }
    if (st) { st->nfev = c.nfev; st->njev = c.njev; st->nlu = c.nlu; st->nrhs_total = c.nrhs; }
    if (aborted) return status | (c.abort_status & WTO_ST_T_RANGE ? WTO_ST_T_RANGE : WTO_ST_NONFINITE);
    if (solver_failed) status |= WTO_ST_SOLVER_FAILED;
    if (c.limit_hit) status |= WTO_ST_STEP_LIMIT;

    /* reactor.py:493-501: state <- last accepted y; time += dt */
    memcpy(y, R.y, sizeof(double) * m);
    *t = *t + dt;
    /* _update_derived_state reactor.py:511-524 */
    int post_range = 0;
    for (int i = 0; i < n; i++) if (y[2 * n + i] < 0.0 || y[2 * n + i] > 100.0) post_range = 1;
    if (derived) {
        for (int i = 0; i < n; i++) derived[i] = pow(10.0, -y[i]);
        for (int i = 0; i < n; i++) derived[n + i] = water_density(y[2 * n + i]);
        if (!post_range)
            for (int i = 0; i < n; i++) derived[2 * n + i] = chlorine_decay_rate(y[2 * n + i]);
    }
    if (post_range) return status | WTO_ST_T_RANGE_POST; /* raises before the clamp */
    /* _enforce_physical_bounds reactor.py:526-541 */
    int cp = 0, cc = 0, ct = 0;
    for (int i = 0; i < n; i++) {
        if (y[i] < 0 || y[i] > 14) cp = 1;
        if (y[n + i] < 0) cc = 1;
        if (y[2 * n + i] < 0 || y[2 * n + i] > 100) ct = 1;
    }
    if (cp) { status |= WTO_ST_CLAMP_PH; for (int i = 0; i < n; i++) y[i] = fmin(fmax(y[i], 0.0), 14.0); }
    if (cc) { status |= WTO_ST_CLAMP_CL; for (int i = 0; i < n; i++) y[n + i] = fmax(y[n + i], 0.0); }
    if (ct) { status |= WTO_ST_CLAMP_T; for (int i = 0; i < n; i++) y[2 * n + i] = fmin(fmax(y[2 * n + i], 0.0), 100.0); }
    for (int i = 0; i < m; i++) if (!isfinite(y[i])) status |= WTO_ST_NONFINITE;
    return status;
}

void wto_ensemble_step(int N, int n, const double *par, const double *bc, double dt,
                       int nsteps, double *y, double *t, double *derived,
                       int *status, int nthreads)
{
    init_consts();
    int m = 3 * n;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int r = 0; r < N; r++) {
        for (int s = 0; s < nsteps; s++) {
            int st = wto_step(n, par + (size_t)r * WTO_NP, bc + (size_t)r * WTO_NB, dt,
                              y + (size_t)r * m, t + r, derived ? derived + (size_t)r * m : NULL, NULL);
            if (status) status[r] |= st;
            if (st & (WTO_ST_T_RANGE | WTO_ST_T_RANGE_POST | WTO_ST_NONFINITE)) break; /* reference raises; loop stops */
        }
    }
}

/* ------------------------------------------------------------------ */
/* Newton-Raphson pH solver chemistry.py:193-330                       */
/* ------------------------------------------------------------------ */
int wto_calculate_pH(double Kw, double Ka1, double Ka2, double CT_mol, double alk_mgL,
                     double guess, double tol, int max_iter, double *pH_out, int *iters)
{
    double pH = guess;
    const double LN10 = log(10.0);
    for (int it = 0; it < max_iter; it++) {
        /* charge_balance_error chemistry.py:193-228 */
        double H = pow(10.0, -pH);
        double OH = Kw / H;
        double H2 = H * H;
        double D = H2 + Ka1 * H + Ka1 * Ka2;
        double a1 = (Ka1 * H) / D, a2 = (Ka1 * Ka2) / D;
        double HCO3 = a1 * CT_mol, CO3 = a2 * CT_mol;
        double alk_eq = alk_mgL / 50000.0;
        double f = H - OH + HCO3 + 2 * CO3 - alk_eq;
        /* charge_balance_derivative chemistry.py:230-269 */
        double dH = -LN10 * H;
        double dOH = -(Kw / H2) * dH;
        double dD = 2 * H + Ka1;
        double D2 = D * D;
        double da1 = Ka1 * (D - H * dD) / D2;
        double da2 = -Ka1 * Ka2 * dD / D2;
        double dHCO3 = CT_mol * da1 * dH;
        double dCO3 = CT_mol * da2 * dH;
        double df = dH - dOH + dHCO3 + 2 * dCO3;
        if (fabs(df) < 1e-15) { *pH_out = pH; if (iters) *iters = it; return 1; }
        double delta = -f / df;
        double pH_new = pH + delta;
        pH_new = fmin(fmax(pH_new, 0.0), 14.0);
        if (fabs(delta) < tol) { *pH_out = pH_new; if (iters) *iters = it + 1; return 0; }
        pH = pH_new;
    }
    *pH_out = pH; if (iters) *iters = max_iter;
    return 2;
}
