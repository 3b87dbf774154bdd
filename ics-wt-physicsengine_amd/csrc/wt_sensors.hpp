// wt_sensors.hpp -- gfx950 device code of the fused sensor suite (SURVEY.md section 8(f) NEXT-1,
// BASELINE config 5): the seven sensors create_realistic_sensor_suite() builds for one reactor
// (sensors/__init__.py:41-120), read once per outer step the way read_all_sensors does
// (__main__.py:121-163), for every reactor of the ensemble in one kernel.
//
//   BaseSensor.read            sensors/base_sensor.py:509-699   (power check, warm-up gate, calibration
//                              validity, sample-line delay, drift, noise, 0.5 lag, installation effects,
//                              rate-of-change, fault draw, clamp, status)
//   SampleLine.transport_sample sensors/base_sensor.py:177-216  (100-entry delay buffer, closest sample;
//                              the suite SHARES one buffer between a pH and a temperature sensor)
//   pHSensor.read              sensors/ph_sensor.py:216-336
//   ChlorineSensor.read        sensors/chlorine_sensor.py:351-484 (amperometric inlet, DPD outlet)
//   FlowSensor.read            sensors/flow_sensor.py:125-219     (magnetic)
//   TemperatureSensor.read     sensors/temperature_sensor.py:110-171 (RTD Pt100)
//
// Mapping: the suite runs INSIDE the physics kernel (wt_device.hpp) at the end of every outer step, on the
// lanes of the wavefront that has just integrated the reactors: the seven taps the sensors look at (pH, Cl, T
// of zones 0 and n-1, flow) go from the physics lanes' registers through a few hundred bytes of LDS to the
// sensor lanes -- no tap buffer in HBM, no second kernel.  The suite splits into five groups that do not
// interact (two of them hold the two sensors that share a delay line, read one after the other as
// read_all_sensors does); lane l of a pass runs group l / R of reactor l % R of the wavefront's R reactors.
// Sensor state is structure-of-arrays in HBM (L2-resident), read and written back by the same lane every step.
// Signal path in fp32 (config 5); time and the slow ageing accumulators in fp64.
//
// Randomness: the reference seeds numpy from `secrets`; here every rng.normal / rng.random /
// rng.choice of the reference is one draw of Philox4x32-10 with key = suite seed and
// counter = (global reactor index, sensor, running draw index) -- the stream the oracle and the
// golden vectors use, so the whole stochastic pipeline is comparable value by value.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wt_plc.hpp"

namespace wts {

constexpr int NSENS = 7;
constexpr int NTAP = 7;          // pH0, pHN, Cl0, ClN, T0, TN, flow
constexpr int RING = 100;        // max(100, int(30 s) + 10)  base_sensor.py:174
// per-sensor float rows
enum { F_CURRENT = 0, F_SUPPLY, F_CAL_OFFSET, F_LAST_VALUE, NF };
// per-sensor double rows (times are seconds since the suite was enabled)
enum { D_CAL_TIME = 0, D_POWER_ON, D_LAST_T, D_PREV_T, D_SLOW0, D_SLOW1, D_SLOW2, ND };
// per-sensor int rows
enum { I_STATUS = 0, I_FAULT, I_HIST_N, I_DRAWS, NI };

enum { ST_NORMAL = 0, ST_CALIBRATING, ST_WARMING_UP, ST_FAILED, ST_SATURATED, ST_DRIFT_WARNING, ST_CAL_EXPIRED,
       ST_OPEN_CIRCUIT, ST_SHORT_CIRCUIT, ST_OUT_OF_RANGE, ST_POWER_FAULT, ST_RATE_FAULT };
enum { FL_NONE = 0, FL_OPEN_CIRCUIT, FL_SHORT_CIRCUIT, FL_OUT_OF_RANGE, FL_RATE_FAULT, FL_POWER_LOW, FL_POWER_HIGH };
enum { K_PH = 0, K_CL_AMP, K_CL_DPD, K_FLOW_MAG, K_T_RTD };

constexpr int RMAX = 32;         // reactors per wavefront at n = 2

// Everything the suite touches in HBM, SoA with the reactor index fastest.
struct SuiteArgs {
    int on;                // 0: no sensor suite attached to this ensemble
    int plc_on;            // publish the readings into the Modbus images and run the command path (wt_plc.hpp)
    int scan_every;        // PLC scan interval in outer steps (a scan also closes every wt_ensemble_step call)
    int64_t N;             // reactors in the ensemble (row stride)
    int64_t reactor_base;  // global index of reactor 0 (sharded ensembles keep distinct streams)
    uint32_t seed_lo, seed_hi;
    const double *t_enable;   // [N] ReactorState.time when the suite was enabled (calibration time)
    float *fs;             // [NSENS][NF][N]
    double *ds;            // [NSENS][ND][N]
    int32_t *is;           // [NSENS][NI][N]
    float *full_scale;     // [N] flow sensor range
    float *ring_t, *ring_v; // [2][RING][N]
    int32_t *ring_push, *ring_cursor; // [2][N] appends so far; push index of the last sample returned
    float *out_value;      // [NSENS][N] last reading
    uint8_t *out_status, *out_fault; // [NSENS][N]
    float *hist_value;     // optional [hist_cap][NSENS][N]
    uint8_t *hist_status, *hist_fault;
    int hist_cap;
    int32_t *hist_pos;     // [N] reads taken so far (next history slot)
    wtp::PackArgs pack;
    wtp::CommandArgs cmd;
};

// One wavefront's per-step hand-off between the lanes that integrate (one per zone), the lanes that run the
// sensor groups and the lanes that publish the register image (one per reactor).  Lives in LDS, on top of
// the tridiagonal-factor store, which is dead between two outer steps.
struct StepIO {
    float tap[NTAP][RMAX];      // pH0, pHN, Cl0, ClN, T0, TN, flow as the sensors see them (fp32)
    double t_after[RMAX];       // ReactorState.time after the step
    int stepped[RMAX];          // the reactor completed this outer step (frozen / absent reactors are not read)
    float val[NSENS][RMAX];     // this step's readings, for the register image
    int fault[NSENS][RMAX];
    double cmd[3][RMAX];        // boundary rows in force after the command path: inlet, acid, chlorine flow
    // sample-line hand-off from the line's first sensor (pH) to its second (RTD): appends / winner after the first,
    // and the entry the first has just appended (push index, -1: none)
    int lpush[2][RMAX], lcur[2][RMAX], lfresh[2][RMAX];
    float lt[2][RMAX], lv[2][RMAX];
};

// ---------------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct Rng {
    uint32_t reactor, sensor, draws, k0, k1;
    __device__ __forceinline__ void next(uint32_t x[4]) { philox4x32_10(reactor, sensor, draws, 0u, k0, k1, x); draws++; }
    __device__ __forceinline__ float uniform() { uint32_t x[4]; next(x); return (float)(x[0] >> 8) * (1.0f / 16777216.0f); }
    __device__ __forceinline__ float normal(float scale)
    {   // Box-Muller on two 24-bit uniforms, u1 in (0,1]
        uint32_t x[4]; next(x);
        const float u1 = (float)((x[0] >> 8) + 1u) * (1.0f / 16777216.0f);
        const float u2 = (float)(x[1] >> 8) * (1.0f / 16777216.0f);
        // hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_cos_f32 (argument in revolutions, so u2 as it
        // is); ~1e-6 absolute on z -- two orders below what the fp32 signal path is held to (tests: 2e-5)
        const float l2 = __builtin_amdgcn_logf(u1);
        return scale * (__builtin_amdgcn_sqrtf(-1.38629436111989061883f * l2) * __builtin_amdgcn_cosf(u2));
    }
};

// ---------------------------------------------------------------- suite constants (sensors/__init__.py:41-120)
struct SensorSpec { int kind, tap_self, tap_T, tap_pH, line; float lo, hi, precision, drift_rate, warmup, cal_valid_h, max_rate; };

__device__ __forceinline__ SensorSpec spec_of(int i, float fs)
{
    // taps: 0 pH0, 1 pHN, 2 Cl0, 3 ClN, 4 T0, 5 TN, 6 flow
    switch (i) {
    case 0: return {K_PH, 0, 4, 0, 0, 0.f, 14.f, 0.01f, 0.01f / 24.f, 1800.f, 24.f, 0.5f};
    case 1: return {K_PH, 1, 5, 1, 1, 0.f, 14.f, 0.01f, 0.01f / 24.f, 1800.f, 24.f, 0.5f};
    case 2: return {K_CL_AMP, 2, 4, 0, -1, 0.f, 10.f, 0.01f, 0.02f / 24.f, 300.f, 24.f, 1.0f};
    case 3: return {K_CL_DPD, 3, 5, 1, -1, 0.f, 10.f, 0.02f, 0.02f / 24.f, 60.f, 24.f, 1.0f};
    case 4: return {K_FLOW_MAG, 6, 4, 0, -1, 0.f, fs, 0.005f * fs, 0.f, 10.f, 8760.f, fs};
    case 5: return {K_T_RTD, 4, 4, 0, 0, -10.f, 110.f, 0.1f, 0.f, 30.f, 8760.f, 10.f};
    default: return {K_T_RTD, 5, 5, 1, 1, -10.f, 110.f, 0.1f, 0.f, 30.f, 8760.f, 10.f};
    }
}

// InstallationQuality of the suite (sensors/__init__.py:53-59): none of the installation-effect
// branches of base_sensor.py:464-507 fires (flow 0.5 >= 0.1, no bubbles, grounding 0.9 >= 0.8,
// vibration 0.1 <= 0.2), so they draw nothing; ambient temperature 30 degC enters the RTD stem error.
constexpr float AMBIENT_T = 30.0f;

// Sensors only interact through the two sample lines ({pH_inlet, temp_inlet} share the inlet line, {pH_outlet,
// temp_outlet} the outlet line): one lane runs one sensor of one reactor, and the two sensors of a line take their
// turns at it one after the other (suite_step).
struct SState {
    float current, supply, cal_offset, last_value;
    double cal_time, power_on, last_t, prev_t, slow0, slow1, slow2;
    int status, fault, hist_n;
    Rng rng;
};

template <class A> __device__ __forceinline__ SState load_state(const A &a, int i, int64_t r)
{
    const int64_t N = a.N;
    const float *F = a.fs + ((int64_t)i * NF) * N + r; const double *D = a.ds + ((int64_t)i * ND) * N + r;
    const int32_t *I = a.is + ((int64_t)i * NI) * N + r;
    SState s;
    s.current = F[F_CURRENT * N]; s.supply = F[F_SUPPLY * N]; s.cal_offset = F[F_CAL_OFFSET * N]; s.last_value = F[F_LAST_VALUE * N];
    s.cal_time = D[D_CAL_TIME * N]; s.power_on = D[D_POWER_ON * N]; s.last_t = D[D_LAST_T * N]; s.prev_t = D[D_PREV_T * N];
    s.slow0 = D[D_SLOW0 * N]; s.slow1 = D[D_SLOW1 * N]; s.slow2 = D[D_SLOW2 * N];
    s.status = I[I_STATUS * N]; s.fault = I[I_FAULT * N]; s.hist_n = I[I_HIST_N * N];
    s.rng = {(uint32_t)(a.reactor_base + r), (uint32_t)i, (uint32_t)I[I_DRAWS * N], a.seed_lo, a.seed_hi};
    return s;
}

template <class A> __device__ __forceinline__ void store_state(const A &a, int i, int64_t r, const SState &s)
{
    const int64_t N = a.N;
    float *F = a.fs + ((int64_t)i * NF) * N + r; double *D = a.ds + ((int64_t)i * ND) * N + r; int32_t *I = a.is + ((int64_t)i * NI) * N + r;
    F[F_CURRENT * N] = s.current; F[F_SUPPLY * N] = s.supply; F[F_LAST_VALUE * N] = s.last_value;
    D[D_LAST_T * N] = s.last_t; D[D_PREV_T * N] = s.prev_t; D[D_SLOW0 * N] = s.slow0; D[D_SLOW1 * N] = s.slow1; D[D_SLOW2 * N] = s.slow2;
    I[I_STATUS * N] = s.status; I[I_FAULT * N] = s.fault; I[I_HIST_N * N] = s.hist_n; I[I_DRAWS * N] = (int32_t)s.rng.draws;
}

// SampleLine.transport_sample base_sensor.py:177-216: append (t, value) to a deque(maxlen = 100), then
// return the value whose timestamp is closest to t - 30 s (min() keeps the first of equal distances).
// Timestamps never decrease and neither does the target, so the winner never moves backwards: the
// search resumes from the previous winner (`cursor`, an absolute push index) and walks forward while a
// later, different timestamp is strictly closer -- the same element a scan of all 100 entries finds
// (the fp32 differences are exact, so equal distances only occur for equal or mirror-image timestamps).
struct Line {
    float *rt, *rv; int64_t N; int pushes, cursor;
    // an entry another lane of this wavefront appended during this read (taken from the hand-off, not from memory)
    int fresh_j; float fresh_t, fresh_v;
    __device__ __forceinline__ float mem_t(int j) const { return (j == fresh_j) ? fresh_t : rt[(int64_t)(j % RING) * N]; }
    __device__ __forceinline__ float mem_v(int j) const { return (j == fresh_j) ? fresh_v : rv[(int64_t)(j % RING) * N]; }
    __device__ __forceinline__ float transport(float t, float tv)
    {
        const int slot = pushes % RING;
        rt[(int64_t)slot * N] = t; rv[(int64_t)slot * N] = tv;
        ++pushes;
        const int oldest = max(0, pushes - RING);
        int c = max(cursor, oldest);
        const float target = t - 30.0f;
        // The winner moves on by about one entry per read, so the entries c .. c+3 are fetched in one go (eight
        // independent loads, one memory round trip) instead of one dependent load per comparison; the walk below
        // falls back to memory only beyond them.  The entry just pushed comes from registers.
        constexpr int WIN = 4;
        float wt[WIN], wv[WIN];
#pragma unroll
        for (int i = 0; i < WIN; ++i) {
            const int j = c + i;
            const bool mem = j < pushes - 1;
            wt[i] = mem ? mem_t(j) : t;
            wv[i] = mem ? mem_v(j) : tv;
        }
        const int c0 = c;
        auto ts = [&](int j) -> float {
            const int i = j - c0;
            if (i < WIN) return i == 0 ? wt[0] : (i == 1 ? wt[1] : (i == 2 ? wt[2] : wt[3]));
            return (j == pushes - 1) ? t : mem_t(j);
        };
        float tc = wt[0];
        float dc = fabsf(tc - target);
        for (int j = c + 1; j < pushes; ++j) {
            const float tj = ts(j);
            if (tj == tc) continue;                     // same timestamp: the earlier entry wins
            const float dj = fabsf(tj - target);
            if (!(dj < dc)) break;
            c = j; tc = tj; dc = dj;
        }
        cursor = c;
        const int i = c - c0;
        if (i < WIN) return i == 0 ? wv[0] : (i == 1 ? wv[1] : (i == 2 ? wv[2] : wv[3]));
        return (c == pushes - 1) ? tv : mem_v(c);
    }
};

// One BaseSensor.read + type-specific read() of sensor `sensor` at time t, in two halves around the sample line
// (the two sensors of a line run in different lanes and take their turns at the line in between).  `tap` points at
// this reactor's column of StepIO::tap (stride RMAX); lanes of one call run different sensors, so every branch on
// the sensor kind is a lane mask.
struct ReadCtx { bool cal_expired; float temp, tv; double cal_hours; };

// up to the sample line; false: the read ended here (power fault, warming up)
__device__ __forceinline__ bool read_begin(const SensorSpec &sp, SState &s, const float *tap, double t, ReadCtx &c,
                                           float &value, int &rstatus, int &rfault)
{
    constexpr int N = RMAX;
    value = __builtin_nanf("");
    // ---------------- BaseSensor.read
    if (!(20.0f < s.supply && s.supply < 28.0f)) {                               // :549-569 (and stays so)
        rstatus = ST_POWER_FAULT; rfault = (s.supply < 20.0f) ? FL_POWER_LOW : FL_POWER_HIGH;
        s.prev_t = s.last_t; s.last_t = t; s.last_value = value; s.hist_n = min(s.hist_n + 1, 2);
        return false;
    }
    s.supply = 24.0f + s.rng.normal(1.0f);                                       // :572
    if (!(t - s.power_on >= (double)sp.warmup)) {                                // :575-588
        rstatus = ST_WARMING_UP; rfault = FL_NONE;
        s.prev_t = s.last_t; s.last_t = t; s.last_value = value; s.hist_n = min(s.hist_n + 1, 2);
        return false;
    }
    const double cal_hours = (t - s.cal_time) / 3600.0;
    const bool cal_expired = (cal_hours > (double)sp.cal_valid_h);                   // :590-593
    if (cal_expired) s.status = ST_CAL_EXPIRED;
    const float temp = tap[sp.tap_T * N];
    float tv;
    if (sp.kind == K_PH) tv = tap[sp.tap_self * N] + 0.003f * (temp - 25.0f);   // ph_sensor.py:169-180
    else if (sp.kind == K_CL_AMP || sp.kind == K_CL_DPD) {                       // chlorine_sensor.py:189-227
        const float ratio = exp10f(7.5f - tap[sp.tap_pH * N]);
        tv = tap[sp.tap_self * N] * (0.5f + 0.5f * (ratio / (1.0f + ratio)));
    } else tv = tap[sp.tap_self * N];                                            // flow_sensor.py:98-102, temperature_sensor.py:105-108
    c.cal_expired = cal_expired; c.temp = temp; c.tv = tv; c.cal_hours = cal_hours;
    return true;
}

// from the sample line's output (c.tv) on                                                           :598-699
__device__ __forceinline__ void read_finish(const SensorSpec &sp, SState &s, const ReadCtx &c, double t, float fs,
                                            float &value, int &rstatus, int &rfault)
{
    const bool cal_expired = c.cal_expired;
    const float temp = c.temp, tv = c.tv;
    const float drift = sp.drift_rate * (float)c.cal_hours + s.cal_offset;       // :612-616
    const float noise = s.rng.normal(sp.precision);                              // :619
    float cur = 0.5f * (tv + noise + drift) + 0.5f * s.current;                  // :622-626 (hysteresis :630 is a no-op)
    float rate = 0.0f;                                                           // :638-648
    if (s.hist_n > 0) {
        const float dtl = (float)(t - s.last_t);
        if (dtl > 0.0f && isfinite(s.last_value)) rate = (cur - s.last_value) / dtl;
    }
    int f = -1;                                                                  // _check_for_faults :377-407
    if (!(20.0f < s.supply && s.supply < 28.0f)) f = (s.supply < 20.0f) ? FL_POWER_LOW : FL_POWER_HIGH;
    else {
        const float span = sp.hi - sp.lo;
        if (cur < sp.lo - 0.1f * span || cur > sp.hi + 0.1f * span) f = FL_OUT_OF_RANGE;
        else if (fabsf(rate) > sp.max_rate) f = FL_RATE_FAULT;
        else if (s.rng.uniform() < 0.0001f) f = (s.rng.uniform() * 2.0f < 1.0f) ? FL_OPEN_CIRCUIT : FL_SHORT_CIRCUIT;
    }
    if (f >= 0) {                                                                // :651-663
        s.fault = f;
        if (f == FL_OPEN_CIRCUIT || f == FL_SHORT_CIRCUIT) { s.status = ST_FAILED; cur = __builtin_nanf(""); }
        else if (f == FL_OUT_OF_RANGE) s.status = ST_OUT_OF_RANGE;
        else if (f == FL_POWER_LOW || f == FL_POWER_HIGH) s.status = ST_POWER_FAULT;
        else s.status = ST_RATE_FAULT;
    } else {                                                                     // :664-682
        s.fault = FL_NONE;
        if (!isnan(cur)) {
            const float b = fminf(fmaxf(cur, sp.lo), sp.hi);
            if (b != cur) s.status = ST_SATURATED; else if (!cal_expired) s.status = ST_NORMAL;
            cur = b;
        }
        if (fabsf(drift) > 0.1f * (sp.hi - sp.lo) && s.status != ST_CAL_EXPIRED) s.status = ST_DRIFT_WARNING;
    }
    s.current = cur;
    const bool have_dt = s.hist_n >= 1;                                          // len(reading_history) >= 2 after the append
    const double dtp = t - s.last_t;
    s.prev_t = s.last_t; s.last_t = t; s.last_value = cur; s.hist_n = min(s.hist_n + 1, 2);
    rstatus = s.status; rfault = s.fault; value = cur;
    if (!isfinite(cur)) return;
    // ---------------- type-specific read()
    // Every kind draws one to three normal deviates; the lanes of a wavefront run all kinds at once, so the draws are
    // taken together -- first the scales of this kind's draws in the order the reference takes them, then up to
    // three Philox calls for the whole wavefront instead of one call per draw and kind.
    float sc1, sc2 = 0.0f, sc3 = 0.0f; int ndraw = 1;
    double days = 0.0;
    const double dday = dtp / 86400.0;           // (one division for every kind; the quotient is the same wherever it is formed)
    if (sp.kind == K_PH) {                                                       // ph_sensor.py:182-214,236-336
        // slow0 = membrane_fouling, slow1 = days_since_cleaning, slow2 = reference_contamination
        if (have_dt) {
            const double bio = (s.slow0 > 0.05) ? 0.1 * exp(0.05 * ((double)temp - 25.0)) : 0.001;
            s.slow0 = fmin(1.0, s.slow0 + (bio + 100.0 * 0.00001) * dday);
            s.slow1 += dday;
        }
        days = (t - s.cal_time) / 86400.0;
        sc1 = 0.002f * (1.0f + 0.1f * fabsf(cur - 7.0f));                        // electrode noise
        sc2 = 0.005f * (1.0f + (float)s.slow2);                                  // junction potential
        sc3 = (float)s.slow0 * 0.05f;                                            // fouling noise
        ndraw = 3;
    } else if (sp.kind == K_CL_AMP) {                                            // chlorine_sensor.py:310-331,405-449
        // slow0 = membrane fouling, slow1 = membrane age [d]
        if (have_dt) { s.slow0 = fmin(1.0, s.slow0 + 0.01 * dday); s.slow1 += dday; }
        sc1 = 0.005f * (1.0f + (float)s.slow1 / 365.0f);                         // polarisation
        sc2 = 0.003f;                                                            // diffusion
        ndraw = 2;
    } else if (sp.kind == K_CL_DPD) {                                            // chlorine_sensor.py:274-308,451-484
        // slow0 = reagent potency, slow1 = light exposure [h], slow2 = reagent age [d]
        if (have_dt) {
            s.slow1 += dtp / 3600.0;
            const double photo = 1.0 + 0.1 * (s.slow1 / 100.0);
            s.slow0 = fmax(0.0, s.slow0 - 1.0 * photo * 0.01 * dday);
            s.slow2 += dday;
        }
        sc1 = 0.005f;
    } else if (sp.kind == K_FLOW_MAG) {                                          // flow_sensor.py:138-178,201-219
        if (have_dt) s.slow0 += 0.001 * dday;                                    // electrode fouling
        sc1 = 0.001f * fs;
    } else {                                                                     // temperature_sensor.py:149-171,118-128
        sc1 = 0.001f;
    }
    const float z1 = s.rng.normal(sc1);
    float z2 = 0.0f, z3 = 0.0f;
    if (ndraw >= 2) z2 = s.rng.normal(sc2);
    if (ndraw >= 3) z3 = s.rng.normal(sc3);
    float fin;
    if (sp.kind == K_PH) {
        const float slope_pct = fmaxf(90.0f, 100.0f - 0.001f * (float)days);     // ph_sensor.py:262-266
        float slope_err = 0.0f;
        if (!(4.0f < cur && cur < 7.0f)) slope_err = fminf(fabsf(cur - 4.0f), fabsf(cur - 7.0f)) * (100.0f - slope_pct) / 100.0f;
        const float foul_off = (float)s.slow0 * 0.2f;
        s.slow2 = fmin(0.5, s.slow2 + 0.0001 * (days / 30.0));
        fin = cur + z1 + z2 + slope_err + foul_off + z3 + (float)s.slow2 * 0.1f;
    } else if (sp.kind == K_CL_AMP) {
        fin = cur * (1.0f - 0.8f * (float)s.slow0) + z1 + z2;
    } else if (sp.kind == K_CL_DPD) {
        fin = cur * (float)s.slow0 * 0.95f + z1;
    } else if (sp.kind == K_FLOW_MAG) {
        fin = cur * fmaxf(0.9f, 1.0f - 0.005f * (float)s.slow0) + z1;
        if (fin < 0.01f * fs) fin = 0.0f;
    } else {
        const float R_meas = 100.0f * (1.0f + 0.00385f * cur) + 2.0f * 0.5f;
        const float power_mW = (1.0e-3f * 1.0e-3f) * R_meas * 1000.0f;
        const float T_meas = (R_meas / 100.0f - 1.0f) / 0.00385f;
        fin = T_meas + 0.001f * power_mW + z1;
        fin += 0.01f * (cur - AMBIENT_T);
    }
    fin = fminf(fmaxf(fin, sp.lo), sp.hi);
    s.current = fin; s.last_value = fin; value = fin;
}

template <class A> __device__ __forceinline__ void emit(const A &a, int i, int64_t r, int pos, float value, int rstatus, int rfault)
{
    const int64_t N = a.N;
    a.out_value[(int64_t)i * N + r] = value; a.out_status[(int64_t)i * N + r] = (uint8_t)rstatus; a.out_fault[(int64_t)i * N + r] = (uint8_t)rfault;
    if (a.hist_value && pos < a.hist_cap) {
        const int64_t o = ((int64_t)pos * NSENS + i) * N + r;
        a.hist_value[o] = value; a.hist_status[o] = (uint8_t)rstatus; a.hist_fault[o] = (uint8_t)rfault;
    }
}

// read_all_sensors (__main__.py:121-163) for the R reactors of this wavefront after one outer step; called by all
// 64 lanes.  One lane per sensor: lane l of a pass reads sensor l / R of reactor l % R (56 lanes at n = 8).  The only
// coupling between sensors is the sample line a pH electrode shares with the RTD next to it: the reference reads
// the pH sensor first (dict order), so the lanes of the pH sensors take their turn at the line, hand the line's
// state on through LDS, then the lanes of the RTDs take theirs.
// rix[s]: ensemble index of the reactor in segment s; hist0[s]: reads reactor s had taken before this work
// item; k: outer steps of the item completed before this one.  Leaves the readings in io.val / io.fault.
template <class A> __device__ __forceinline__ void suite_step(const A &a, StepIO &io, const int *rix, int R, const int *hist0, int k)
{
    const int lane = threadIdx.x & 63;
    for (int base = 0; base < NSENS * R; base += 64) {
        const int idx = base + lane;
        const int i = idx / R, sl = idx - i * R;            // sensor, reactor slot
        const bool active = (i < NSENS) && io.stepped[sl];
        const int64_t r = rix[sl & (RMAX - 1)], N = a.N;
        const int line = (i == 0 || i == 5) ? 0 : ((i == 1 || i == 6) ? 1 : -1);
        const bool first = active && i < 2, second = active && i >= 5;
        float fs = 0; double t = 0; int pos = 0;
        float value = 0; int rstatus = 0, rfault = 0;
        SState st; ReadCtx c = {false, 0.0f, 0.0f, 0.0}; bool go = false;
        SensorSpec sp = spec_of(0, 0.0f);
        Line ln = {nullptr, nullptr, N, 0, 0, -1, 0.0f, 0.0f};
        if (active) {
            fs = a.full_scale[r];
            t = io.t_after[sl] - a.t_enable[r];
            pos = hist0[sl] + k;
            sp = spec_of(i, fs);
            st = load_state(a, i, r);
            go = read_begin(sp, st, &io.tap[0][sl], t, c, value, rstatus, rfault);
            if (line >= 0) { ln.rt = a.ring_t + ((int64_t)line * RING) * N + r; ln.rv = a.ring_v + ((int64_t)line * RING) * N + r; }
        }
        if (first) {                                         // SampleLine.transport_sample of the pH sensors  :598-609
            ln.pushes = a.ring_push[(int64_t)line * N + r]; ln.cursor = a.ring_cursor[(int64_t)line * N + r];
            const float pushed = c.tv;
            if (go) c.tv = ln.transport((float)t, c.tv);
            io.lpush[line][sl] = ln.pushes; io.lcur[line][sl] = ln.cursor;
            io.lfresh[line][sl] = go ? ln.pushes - 1 : -1; io.lt[line][sl] = (float)t; io.lv[line][sl] = pushed;
        }
        // (one wavefront: its LDS operations complete in order; the fence only stops the compiler from moving them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (second) {                                        // ... then of the RTDs on the same lines
            ln.pushes = io.lpush[line][sl]; ln.cursor = io.lcur[line][sl];
            ln.fresh_j = io.lfresh[line][sl]; ln.fresh_t = io.lt[line][sl]; ln.fresh_v = io.lv[line][sl];
            if (go) c.tv = ln.transport((float)t, c.tv);
            a.ring_push[(int64_t)line * N + r] = ln.pushes; a.ring_cursor[(int64_t)line * N + r] = ln.cursor;
        }
        if (active) {
            if (go) read_finish(sp, st, c, t, fs, value, rstatus, rfault);
            store_state(a, i, r, st);
            emit(a, i, r, pos, value, rstatus, rfault);
            io.val[i][sl] = value; io.fault[i][sl] = rfault;
        }
    }
}

// initialize_sensors (__main__.py:84-118): construct the suite and calibrate every sensor at the
// current time: offset = reference - current_value, warm-up restarts (base_sensor.py:701-755)
struct SensorInitArgs {
    int64_t N;
    const double *cfg_flow, *cfg_cl, *cfg_temp; // [N] ReactorConfiguration.flow_rate / initial_chlorine / temperature
    float *fs; double *ds; int32_t *is; float *full_scale;
    int32_t *ring_push, *ring_cursor;
    float *out_value; uint8_t *out_status, *out_fault;
    int32_t *hist_pos;
};

__global__ __launch_bounds__(256) void sensor_init_kernel(const SensorInitArgs a)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.N) return;
    const int64_t N = a.N;
    const float fsr = (float)(a.cfg_flow[r] * 2.0);                   // full_scale = flow_rate * 2  sensors/__init__.py:100
    a.full_scale[r] = fsr;
    const float init_current[NSENS] = {7.0f, 7.0f, 0.0f, 0.0f, 0.0f, 20.0f, 20.0f};
    const float ref[NSENS] = {7.0f, 7.0f, (float)a.cfg_cl[r], (float)a.cfg_cl[r], (float)a.cfg_flow[r], (float)a.cfg_temp[r], (float)a.cfg_temp[r]};
    for (int i = 0; i < NSENS; ++i) {
        float *F = a.fs + ((int64_t)i * NF) * N + r; double *D = a.ds + ((int64_t)i * ND) * N + r; int32_t *I = a.is + ((int64_t)i * NI) * N + r;
        F[F_CURRENT * N] = init_current[i]; F[F_SUPPLY * N] = 24.0f; F[F_CAL_OFFSET * N] = ref[i] - init_current[i];
        F[F_LAST_VALUE * N] = __builtin_nanf("");
        for (int q = 0; q < ND; ++q) D[q * N] = 0.0;
        if (i == 3) D[D_SLOW0 * N] = 1.0;                             // DPD reagent_potency = 1 (chlorine_sensor.py:171)
        I[I_STATUS * N] = ST_NORMAL; I[I_FAULT * N] = FL_NONE; I[I_HIST_N * N] = 0; I[I_DRAWS * N] = 0;
        a.out_value[(int64_t)i * N + r] = __builtin_nanf(""); a.out_status[(int64_t)i * N + r] = ST_NORMAL; a.out_fault[(int64_t)i * N + r] = FL_NONE;
    }
    a.ring_push[r] = 0; a.ring_push[N + r] = 0; a.ring_cursor[r] = 0; a.ring_cursor[N + r] = 0;
    if (a.hist_pos) a.hist_pos[r] = 0;
}

} // namespace wts
