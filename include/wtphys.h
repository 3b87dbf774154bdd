/*
 * wtphys.h -- C ABI of libwtphys.so: the MI355X (gfx950) implementation of the
 * multi-zone CSTR physics step of wt_simulator.core, batched over an ensemble
 * of independent reactors.
 *
 * The reference has no FFI: its boundary for this path is the Python class
 * API  IntegratedCSTR(config).step(dt, boundary) -> ReactorState
 * (/root/reference/src/wt_simulator/core/reactor.py:203, :450-509).  Each entry
 * point below names the reference interface it stands in for.  All functions
 * return 0 on success or a WT_E_* code; wt_last_error() gives the text.  No
 * exception or torch type crosses this boundary; pointers are plain host (or,
 * where stated, device) pointers.
 *
 * Data layout (all fp64):
 *   state   pH, Cl, T : [N][n]   reactor-major, zone fastest  (= numpy (N, n))
 *   par     [WT_NP][N]           per-reactor constants, SoA   (see WT_P_*)
 *   bc      [WT_NB][N]           BoundaryConditions columns, SoA (see WT_B_*)
 * One handle = one device, one HIP stream, one caller thread (not re-entrant).
 */
#ifndef WTPHYS_H
#define WTPHYS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WT_ABI_VERSION 1
#define WT_MAX_ZONES 64 /* one reactor's zones live in one 64-lane wavefront */
#define WT_MAX_STREAMS 8
#define WT_DEFAULT_CHUNK 50   /* outer steps per launch unless wt_ensemble_set_schedule says otherwise */

/* rows of the per-reactor constant block (values as the reference's __init__
 * computes them: reactor.py:229-270, chemistry.py:116-132, transport.py:202-290) */
enum {
    WT_P_VOLUME = 0, WT_P_HEIGHT = 1, WT_P_DIAMETER = 2,
    WT_P_KW = 3, WT_P_KA1 = 4, WT_P_KA2 = 5, WT_P_KA_HOCL = 6, WT_P_CT_MOL = 7,
    WT_P_KEX = 8, WT_P_USUP = 9, WT_P_STRAT = 10, WT_P_RI_CRIT = 11, WT_P_SUPP = 12,
    WT_NP = 16
};

/* rows of the boundary block = BoundaryConditions fields, reactor.py:169-186 */
enum {
    WT_B_Q_IN = 0, WT_B_PH_IN = 1, WT_B_CL_IN = 2, WT_B_T_IN = 3,
    WT_B_Q_ACID = 4, WT_B_C_ACID = 5, WT_B_Q_CL = 6, WT_B_C_CL = 7,
    WT_B_T_AMB = 8, WT_B_U = 9,
    WT_NB = 10
};

/* per-reactor status bits (OR-accumulated until cleared) */
enum {
    WT_ST_T_RANGE = 1,        /* a zone temperature left [0,100] C inside the solve: the reference
                                 raises ValueError (thermodynamics.py:146-157); state is NOT advanced
                                 and the reactor stays frozen until the host rewrites its state */
    WT_ST_SOLVER_FAILED = 2,  /* Radau "step size too small" -> reactor.py:486-487 warning; state is
                                 still overwritten with the last accepted y, as the reference does */
    WT_ST_CLAMP_PH = 4,       /* reactor.py:529-531 */
    WT_ST_CLAMP_CL = 8,       /* reactor.py:534-536 */
    WT_ST_CLAMP_T = 16,       /* reactor.py:539-541 */
    WT_ST_T_RANGE_POST = 32,  /* ValueError out of _update_derived_state (reactor.py:522-524):
                                 state/time/H/density were updated, decay rate and clamps were not */
    WT_ST_NONFINITE = 64,     /* the state is not finite at the start of a step: scipy's solve_ivp raises ValueError
                                 ("All components of the initial state `y0` must be finite."), self.state untouched;
                                 the reactor does not advance until the host rewrites its state */
    WT_ST_STEP_LIMIT = 128    /* not a reference behaviour: the attempt limit of wt_ensemble_set_step_limit was hit;
                                 always together with WT_ST_SOLVER_FAILED, state = last accepted y */
};

enum {
    WT_OK = 0, WT_E_ARG = 1, WT_E_HIP = 2, WT_E_NOGPU = 3, WT_E_STATE = 4
};

typedef struct wt_ensemble wt_ensemble;

/* per-reactor solver counters of the LAST outer step (scipy's nfev/njev/nlu,
 * accepted and rejected internal steps); used by the parity tests to check the
 * decision sequence against the oracle. */
typedef struct { int32_t nfev, njev, nlu, nsteps, nrej; } wt_solver_stats;

int wt_abi_version(void);
const char *wt_last_error(void);
int wt_device_count(int *count);

/* IntegratedCSTR.__init__ (reactor.py:203-227) for N reactors with n zones each
 * on HIP device `device`.  `par` is host memory, [WT_NP][N]. */
int wt_ensemble_create(int64_t n_reactors, int n_zones, int device, const double *par,
                       wt_ensemble **out);
int wt_ensemble_destroy(wt_ensemble *h);

/* overwrite reactor.state.{pH,chlorine,temperature,time} (reactor.py:217-222,
 * and the documented "callers may edit state between steps" path :467-469).
 * Host arrays [N][n]; `time` [N] may be NULL (keeps current).  Clears status. */
int wt_ensemble_set_state(wt_ensemble *h, const double *pH, const double *Cl, const double *T,
                          const double *time);
/* BoundaryConditions for every reactor (reactor.py:150-186), host [WT_NB][N]. */
int wt_ensemble_set_boundary(wt_ensemble *h, const double *bc);

/* IntegratedCSTR.step(dt, boundary) n_steps times (reactor.py:450-509); asynchronous, ordered after and
 * before other work on the handle's stream.  The boundary is held constant unless plant I/O is on (then the
 * command path rewrites it at every PLC scan).  fused == 0 makes every outer step a PLC scan, as the reference's
 * loop does; otherwise a scan happens every chunk_steps outer steps and at the end of the call.  Results do not
 * depend on the schedule (tests assert bitwise equality). */
int wt_ensemble_step(wt_ensemble *h, double dt, int n_steps, int fused);
/* Schedule.  n_streams == 0 (default), WT_SCHED_QUEUE: one kernel launch per call; worker wavefronts take
 * (wavefront-group, next few outer steps) work items from a device-side FIFO, so no wavefront ever waits for a
 * launch boundary and a scan per outer step costs no launch.  n_streams >= 1, WT_SCHED_STREAMS (the round-1
 * schedule, kept for comparison): n_streams contiguous reactor ranges on internal HIP streams (fork/join around
 * the handle's stream), launches of at most chunk_steps outer steps.  chunk_steps is the PLC scan interval under
 * both (0 = one scan per call).  Default: queue, WT_DEFAULT_CHUNK. */
int wt_ensemble_set_schedule(wt_ensemble *h, int n_streams, int chunk_steps);
/* the schedule in force: mode (WT_SCHED_*), reactor ranges / streams (0 under WT_SCHED_QUEUE), scan interval,
 * worker wavefronts of the queue schedule (0 under WT_SCHED_STREAMS) */
enum { WT_SCHED_STREAMS = 0, WT_SCHED_QUEUE = 1 };
int wt_ensemble_get_schedule(wt_ensemble *h, int *mode, int *n_streams, int *chunk_steps, int *workers);
/* outer steps a wavefront-group's state stays in registers before it goes back to memory, for a call of n_steps:
 * the work-item length of the queue schedule (n_steps / 6, at most 32), the launch length of the stream schedule */
int wt_ensemble_item_steps(wt_ensemble *h, int n_steps);
/* Launch-completeness record, sticky for the life of the handle: bit 0 = a work-queue hand-off gave up waiting, bit 1 =
 * a launch ended with a wavefront-group short of its step count (checked on the device after every queue launch).
 * Never observed; while it is non-zero wt_ensemble_synchronize and every state download (get_state / get_snapshot /
 * get_status) return WT_E_HIP instead of passing an incomplete state off as WT_OK. */
int wt_ensemble_queue_error(wt_ensemble *h, int *error);
/* Kept for ABI compatibility, no effect: the reactors sharing a wavefront always start an outer step
 * together (they wait for the slowest of them), which is what makes the end of an outer step a
 * wavefront-uniform point for the sensor suite and the PLC scan.  Results never depended on it. */
int wt_ensemble_set_sync(wt_ensemble *h, int sync_outer);
/* Placement of reactors into wavefronts.  Reactors never interact, so which of them share a wavefront changes no
 * result bit -- but a wavefront costs what its slowest reactor costs.  WT_PLACE_ADAPTIVE (default): once the cost
 * history (the solver's RHS evaluations per reactor) covers WT_PLACE_MIN_STEPS outer steps, the next wt_ensemble_step
 * call re-deals the wavefront slots in order of cost (device-side stable counting sort on the handle's stream, no
 * synchronisation) and restarts the history.  WT_PLACE_IDENTITY: reactor r sits in slot r, as in round 1.
 * wt_ensemble_get_placement: current mode and slot -> reactor table [N] (either pointer may be NULL; the table
 * synchronises the stream).  No counterpart in the reference (one reactor per process). */
#define WT_PLACE_IDENTITY 0
#define WT_PLACE_ADAPTIVE 1
#define WT_PLACE_MIN_STEPS 32
int wt_ensemble_set_placement(wt_ensemble *h, int mode);
int wt_ensemble_get_placement(wt_ensemble *h, int *mode, int32_t *perm);
/* how often the slots have been re-dealt so far, and how many outer steps the running cost history covers
 * (a short run -- fewer than WT_PLACE_MIN_STEPS steps before its timed call -- never re-deals) */
int wt_ensemble_placement_info(wt_ensemble *h, int64_t *redeals, int64_t *history_steps);

/* Guard the reference lacks.  Where the solution slides along a discontinuity of the RHS (the
 * 8 degC density branch, spatial.py:177-189, under strong heat loss) scipy's Radau takes millions
 * of internal steps for one outer step; the reference would grind through them for hours.  A
 * reactor that needs more than max_attempts step attempts (accepted + rejected) in one outer step
 * is stopped like a solver failure (WT_ST_SOLVER_FAILED | WT_ST_STEP_LIMIT).  Default 2000
 * (a normal step needs 2-4, a hard one a few dozen); 0 = unlimited, as the reference -- an explicit opt-in: the
 * step kernel cannot be cancelled, and a reactor on that discontinuity then holds its wavefront (and every download)
 * for as long as the reference would take. */
int wt_ensemble_set_step_limit(wt_ensemble *h, int max_attempts);
int wt_ensemble_synchronize(wt_ensemble *h);

/* ReactorState read-back (reactor.py:113-147).  Any pointer may be NULL.
 * pH/Cl/T [N][n]; time, flow [N]. Synchronises the stream.  Ensembles whose whole image fits 256 KiB (the single-reactor
 * drop-in above all) come back as one packed copy through pinned memory: one kernel, one copy, one synchronisation. */
int wt_ensemble_get_state(wt_ensemble *h, double *pH, double *Cl, double *T, double *time,
                          double *flow);
/* everything a ReactorState holds plus the status words, one synchronisation (any pointer may be NULL) */
int wt_ensemble_get_snapshot(wt_ensemble *h, double *pH, double *Cl, double *T, double *time, double *flow,
                             double *H, double *rho, double *kdecay, uint32_t *flags);
/* H_concentration, density, chlorine_decay_rate (reactor.py:511-524), [N][n] each. */
int wt_ensemble_get_derived(wt_ensemble *h, double *H, double *rho, double *kdecay);
int wt_ensemble_get_status(wt_ensemble *h, uint32_t *flags /* [N] */);
/* the temperature the reference's ValueError names (thermodynamics.py:151: the first out-of-range zone of the
 * evaluation that raised) for reactors flagged WT_ST_T_RANGE / WT_ST_T_RANGE_POST; [N], undefined elsewhere */
int wt_ensemble_get_bad_temperature(wt_ensemble *h, double *value /* [N] */);
int wt_ensemble_clear_status(wt_ensemble *h);
int wt_ensemble_get_stats(wt_ensemble *h, wt_solver_stats *stats /* [N] */);

/* IntegratedCSTR.derivatives(t, y, boundary) (reactor.py:272-448) for every
 * reactor at caller-supplied states (host [N][n]); uses the handle's constants
 * and boundary.  flags[N] gets WT_ST_T_RANGE where the reference would raise. */
int wt_ensemble_rhs(wt_ensemble *h, const double *pH, const double *Cl, const double *T,
                    double *dpH, double *dCl, double *dT, uint32_t *flags);

/* Device-side access for zero-copy consumers (RCCL gather, fused sensors):
 * copies the current state into caller DEVICE memory laid out [3][N][n]
 * (pH, Cl, T), asynchronously on the handle's stream. */
int wt_ensemble_export_state_device(wt_ensemble *h, void *dst_device);
/* Use a caller-owned hipStream_t (e.g. torch's current stream) for all work. */
int wt_ensemble_set_stream(wt_ensemble *h, void *hip_stream);

/* Per-launch HIP-event timing (start/stop events around every step-kernel launch on
 * the stream it is launched on).  launch_stats synchronises, returns the number of
 * launches since timing was switched on / last read, the sum and the maximum of their
 * durations, and resets the counters. */
int wt_ensemble_launch_timing(wt_ensemble *h, int enable);
int wt_ensemble_launch_stats(wt_ensemble *h, int64_t *n_launches, double *sum_ms, double *max_ms);

/* HIP-event bracketing on the handle's stream, for benchmarks. */
int wt_ensemble_timer_start(wt_ensemble *h);
int wt_ensemble_timer_stop(wt_ensemble *h, float *elapsed_ms /* synchronises */);

/* ---- fused sensor suite (SURVEY.md section 8(f) NEXT-1, BASELINE config 5; fp32 signal path) ----
 * create_realistic_sensor_suite + initialize_sensors (sensors/__init__.py:41-120, __main__.py:84-118) for
 * every reactor: pH inlet/outlet, chlorine amperometric inlet / DPD outlet, magnetic flow, RTD inlet/outlet,
 * calibrated at the current ensemble time.  From then on every outer step of wt_ensemble_step is followed
 * by read_all_sensors (__main__.py:121-163) on the device.  Randomness: Philox4x32-10, key = seed,
 * counter = (reactor_base + reactor, sensor, draw index).  cfg_* are host arrays [N] of
 * ReactorConfiguration.flow_rate / initial_chlorine / temperature.  history_capacity > 0 keeps the first
 * that many reads of every sensor on the device (tests, replay). */
#define WT_N_SENSORS 7  /* order: pH_inlet, pH_outlet, chlorine_inlet, chlorine_outlet, flow_main, temp_inlet, temp_outlet */
int wt_ensemble_sensors_enable(wt_ensemble *h, uint64_t seed, int64_t reactor_base, const double *cfg_flow,
                               const double *cfg_chlorine, const double *cfg_temperature, int history_capacity);
/* last SensorReading.value / status / fault of every sensor: [WT_N_SENSORS][N]; status and fault use the
 * declaration order of SensorStatus / SensorFault (sensors/base_sensor.py:49-75). */
int wt_ensemble_sensors_get(wt_ensemble *h, float *values, uint8_t *status, uint8_t *fault);
/* recorded reads [history_capacity][WT_N_SENSORS][N] and the number of reads taken per reactor [N] */
int wt_ensemble_sensors_history(wt_ensemble *h, float *values, uint8_t *status, uint8_t *fault, int32_t *n_filled);

/* ---- plant I/O around the step (SURVEY.md section 8(f) NEXT-2 driver loop, NEXT-3 register image) ----
 * One virtual Modbus slave per reactor, as the reference's loop body keeps it (__main__.py:398-427):
 * after every launch of wt_ensemble_step (= one PLC scan; wt_ensemble_set_schedule's chunk_steps, or
 * fused = 0 for a scan per outer step as in the reference) the device runs, for the reactors of the launch,
 *   update_modbus_inputs   (__main__.py:166-224; encoder modbus/protocols.py:35-58; map register_map.py:119-244,364-401)
 *   read_modbus_commands + apply_boundary_conditions (__main__.py:227-271; decoder protocols.py:155-177)
 * so a command written to the holding image acts from the next scan on.  Needs the sensor suite.
 * Input image: [N][WT_IR_WORDS] uint16, words 0..15 = input registers 0..15 (pH_inlet 0-1, pH_middle 2-3
 * (never written), pH_outlet 4-5, chlorine_inlet 6-7, chlorine_outlet 8-9, flow_rate 10-11, temperature_inlet
 * 12-13, temperature_outlet 14-15), 16-17 = simulation_time (registers 100-101: the loop's sim_time, which
 * lags ReactorState.time by one dt, __main__.py:413,446), 18 = system_status (register 102),
 * 19 = discrete inputs 0..2 (sensor_fault_pH_inlet, _pH_outlet, _chlorine) in bits 0..2.  float32 values
 * occupy (high word, low word).  Holding image: [N][WT_HR_WORDS] uint16 = holding registers 0..5
 * (acid_flow_rate, chlorine_flow_rate, inlet_flow_rate), all 0 at start like the reference's data block. */
#define WT_IR_WORDS 20
#define WT_HR_WORDS 6
int wt_ensemble_plc_enable(wt_ensemble *h);
/* what Modbus masters wrote since the last scan, for reactors [first_reactor, first_reactor + count) */
int wt_ensemble_plc_write_holding(wt_ensemble *h, const uint16_t *words, int64_t first_reactor, int64_t count);
/* input image [N][WT_IR_WORDS]; update_ok[N] = 0 where the last update raised in the reference
 * (a value outside +-1e9, modbus/slave.py:146-147: registers written before it are new, the rest stale) */
int wt_ensemble_plc_read_inputs(wt_ensemble *h, uint16_t *words, uint8_t *update_ok);
/* device pointers of both images for co-resident servers / controllers (synchronise first) */
int wt_ensemble_plc_device(wt_ensemble *h, void **input_image, void **holding_image);
/* current boundary block [WT_NB][N] (after the command path acted on it) */
int wt_ensemble_get_boundary(wt_ensemble *h, double *bc);

/* ---- reactor diagnostics (SURVEY.md section 8(f) NEXT-4): reductions over the zones of every reactor ----
 * out: host [WT_N_DIAG][N] doubles, rows
 *   0 total_chlorine_mg, 1 total_H_mol, 2 total_OH_mol, 3 charge_balance_mol, 4 thermal_energy_kJ
 *                                   IntegratedCSTR.validate_conservation (reactor.py:570-611)
 *   5 pH CV, 6 pH segregation index, 7 chlorine CV, 8 chlorine segregation index
 *                                   TransportModel.calculate_mixing_quality (transport.py:338-384, reactor.py:638-639)
 *   9 thermocline depth from the top [m], NaN where the reference returns None
 *                                   SpatialModel.identify_thermocline (spatial.py:352-379)
 *   10 + 8 p + {0 mean_value, 1 std_value, 2 max_value, 3 min_value, 4 range, 5 max_gradient, 6 mean_gradient,
 *   7 gradient_location}, p = 0 pH, 1 chlorine, 2 temperature
 *                                   SpatialModel.calculate_spatial_gradients (spatial.py:440-477)
 * evaluated on the current state (the derived H+ of the last step; call after wt_ensemble_step). */
#define WT_N_DIAG 34
int wt_ensemble_diagnostics(wt_ensemble *h, double *out);

/* Self-test of the kernel's cross-lane primitives (DPP row / wave shifts, segment
 * sums) against ds_bpermute for a given zone count; *mismatches must come back 0. */
int wt_selftest_shuffles(int device, int n_zones, int *mismatches);

/* Per-wavefront diagnostics of the LAST launch: {loop trips, trips with a Newton
 * evaluation, shader clocks, 100 MHz wall ticks, factorisation / Jacobian /
 * deferred-f block executions, spare} per wavefront.  The first call allocates the
 * buffer and switches recording on (out may be NULL); later calls copy
 * [n_waves][wt_wave_diag_slots()] int64 into `out` (8 slots; 16 in -DWT_STAMPS diagnostic builds,
 * which add the shader-clock shares of the kernel loop's sections). */
int wt_wave_diag_slots(void);
/* Developer trace of the LAST launch's work items under the queue schedule: {worker, group, first step | steps << 32,
 * start, end} in 100 MHz ticks, 5 int64 per item.  The first call allocates `capacity` items and switches tracing on. */
int wt_ensemble_item_trace(wt_ensemble *h, int64_t *out, int capacity, int *n_items);
int wt_ensemble_wave_diag(wt_ensemble *h, int64_t *out, int64_t capacity, int64_t *n_waves);

int64_t wt_ensemble_size(const wt_ensemble *h);
int wt_ensemble_zones(const wt_ensemble *h);

/* AqueousChemistry.calculate_pH (chemistry.py:271-330) batched: Newton-Raphson
 * on the charge balance, one system per element.  Host arrays of length n.
 * rc[i]: 0 converged, 1 "derivative too small" (RuntimeError), 2 no convergence. */
int wt_ph_solve(int device, int64_t n, const double *Kw, const double *Ka1, const double *Ka2,
                const double *CT_mol, const double *alk_mgL, const double *guess,
                double tol, int max_iter, double *pH_out, int32_t *iters, int32_t *rc);

#ifdef __cplusplus
}
#endif
#endif /* WTPHYS_H */
