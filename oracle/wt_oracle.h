/*
 * wt_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of the reference's multi-zone CSTR physics step
 * (wt_simulator.core, /root/reference/src/wt_simulator/core/reactor.py:272-509)
 * including the third-party integrator it delegates to
 * (scipy 1.15.3 scipy/integrate/_ivp/radau.py + common.py + base.py + ivp.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The shipped path is
 * ics-wt-physicsengine_amd/csrc (HIP) and never links or calls this.
 *
 * Parity pin: tests/golden/ (vectors produced by importing the Python
 * reference in the build container with oracle/gen_golden.py).
 */
#ifndef WT_ORACLE_H
#define WT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* per-reactor constant vector (doubles) */
enum {
    WTO_P_VOLUME = 0,   /* [L]   ReactorConfiguration.volume   reactor.py:61 */
    WTO_P_HEIGHT = 1,   /* [m]                                  reactor.py:62 */
    WTO_P_DIAMETER = 2, /* [m]                                  reactor.py:63 */
    WTO_P_KW = 3,       /* Kw(T_cfg)        chemistry.py:119 */
    WTO_P_KA1 = 4,      /* Ka1(T_cfg)       chemistry.py:123 */
    WTO_P_KA2 = 5,      /* Ka2(T_cfg)       chemistry.py:126 */
    WTO_P_KA_HOCL = 6,  /* Ka_HOCl(T_cfg)   chemistry.py:132 */
    WTO_P_CT_MOL = 7,   /* total_carbonate/1000 [mol/L] chemistry.py:428 */
    WTO_P_KEX = 8,      /* K_exchange_per_s  transport.py:282-290 */
    WTO_P_USUP = 9,     /* superficial_velocity transport.py:221-222 */
    WTO_P_STRAT = 10,   /* enable_thermal_stratification (0/1) reactor.py:310 */
    WTO_P_RI_CRIT = 11, /* 0.25 spatial.py:71 */
    WTO_P_SUPP = 12,    /* 0.5  spatial.py:72 */
    WTO_NP = 16
};

/* boundary vector (doubles), BoundaryConditions field order reactor.py:169-186 */
enum {
    WTO_B_Q_IN = 0, WTO_B_PH_IN = 1, WTO_B_CL_IN = 2, WTO_B_T_IN = 3,
    WTO_B_Q_ACID = 4, WTO_B_C_ACID = 5, WTO_B_Q_CL = 6, WTO_B_C_CL = 7,
    WTO_B_T_AMB = 8, WTO_B_U = 9,
    WTO_NB = 10
};

/* status bits (same meaning as include/wtphys.h WT_ST_*) */
enum {
    WTO_ST_T_RANGE = 1,       /* ValueError from celsius_to_kelvin during the solve (thermodynamics.py:146-157) */
    WTO_ST_SOLVER_FAILED = 2, /* TOO_SMALL_STEP (radau.py:427-428) -> reactor.py:486-487 warning */
    WTO_ST_CLAMP_PH = 4,      /* reactor.py:529-531 */
    WTO_ST_CLAMP_CL = 8,      /* reactor.py:534-536 */
    WTO_ST_CLAMP_T = 16,      /* reactor.py:539-541 */
    WTO_ST_T_RANGE_POST = 32, /* ValueError from _update_derived_state reactor.py:522-524 */
    WTO_ST_NONFINITE = 64,    /* non-finite state at the start of a step: scipy raises ValueError (base.py:19-20), state untouched */
    WTO_ST_STEP_LIMIT = 128   /* guard (not in the reference): attempt limit of wto_set_step_limit hit */
};

typedef struct {
    int nfev;      /* counted like scipy: calls through solver.fun          */
    int njev;      /* num_jac calls                                          */
    int nlu;       /* LU factorisations (real and complex counted singly)    */
    int nsteps;    /* accepted internal steps                                */
    int nrej;      /* rejected (error) + halved (newton) attempts            */
    int nrhs_total;/* every derivatives() call incl. num_jac columns         */
    double t_internal[64]; /* accepted internal times (first 64), absolute   */
} wto_stats;

/* linear-solve flavour: 0 = dense partial-pivot LU (what scipy does),
 * 1 = block-triangular tridiagonal solve (what the HIP kernel does). */
void wto_set_linsolve(int mode);
/* 0 = unlimited (reference behaviour); otherwise stop an outer step after that many step attempts
 * with WTO_ST_SOLVER_FAILED | WTO_ST_STEP_LIMIT, state = last accepted y, as a solver failure. */
void wto_set_step_limit(long max_attempts);

/* derivatives(t, y, boundary): reactor.py:272-448.  y = [pH.., Cl.., T..].
 * returns 0, or WTO_ST_T_RANGE if any T outside [0,100] (reference raises). */
int wto_rhs(int n, const double *par, const double *bc, const double *y, double *dydt);

/* One IntegratedCSTR.step(dt, boundary): reactor.py:450-509.
 * y (3n) and *t are updated in place exactly when the reference would have
 * updated self.state; derived (3n: H, density, decay) written when computed.
 * Returns status bits. */
int wto_step(int n, const double *par, const double *bc, double dt,
             double *y, double *t, double *derived, wto_stats *st);

/* Ensemble convenience for the CPU baseline: N reactors, AoS inputs
 * par[N][WTO_NP], bc[N][WTO_NB], y[N][3n], t[N], derived[N][3n] (may be NULL),
 * status[N] OR-accumulated; nsteps outer steps; nthreads OpenMP threads. */
void wto_ensemble_step(int N, int n, const double *par, const double *bc, double dt,
                       int nsteps, double *y, double *t, double *derived,
                       int *status, int nthreads);

/* AqueousChemistry.calculate_pH: chemistry.py:271-330.
 * returns 0 ok, 1 derivative too small (RuntimeError), 2 no convergence. */
int wto_calculate_pH(double Kw, double Ka1, double Ka2, double CT_mol, double alk_mgL,
                     double guess, double tol, int max_iter, double *pH_out, int *iters);

#ifdef __cplusplus
}
#endif
#endif
