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
// Mapping: one thread per reactor (the seven reads of a step are sequential by construction: two of
// them communicate through the shared delay lines), structure-of-arrays state so that lane r of a
// wavefront touches word r of every state row -- the kernel is a byte mover, bound by HBM/L2
// bandwidth.  Signal path in fp32 (config 5); time and the slow ageing accumulators in fp64.
// Input: per-step "taps" (pH, Cl, T of zones 0 and n-1, flow) written by the physics kernel for the
// steps of one launch, so the suite sees every outer step although the physics keeps its state in
// registers across them.
//
// Randomness: the reference seeds numpy from `secrets`; here every rng.normal / rng.random /
// rng.choice of the reference is one draw of Philox4x32-10 with key = suite seed and
// counter = (global reactor index, sensor, running draw index) -- the stream the oracle and the
// golden vectors use, so the whole stochastic pipeline is comparable value by value.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

struct SensorArgs {
    int64_t N;             // reactors in the ensemble (row stride)
    int64_t r0, r1;        // this launch handles reactors [r0, r1)
    int64_t reactor_base;  // global index of reactor 0 (sharded ensembles keep distinct streams)
    uint32_t seed_lo, seed_hi;
    int n_steps;           // taps of this launch
    double dt;
    const float *taps;     // [n_steps][NTAP][N]
    const int32_t *tap_count; // [N] outer steps the physics completed in this launch
    const double *time_end;   // [N] ReactorState.time after the launch
    const double *t_enable;   // [N] ReactorState.time when the suite was enabled (calibration time)
    float *fs;             // [NSENS][NF][N]
    double *ds;            // [NSENS][ND][N]
    int32_t *is;           // [NSENS][NI][N]
    float *full_scale;     // [N] flow sensor range
    float *ring_t, *ring_v; // [2][RING][N]
    int32_t *ring_head, *ring_cnt; // [2][N]
    float *out_value;      // [NSENS][N] last reading
    uint8_t *out_status, *out_fault; // [NSENS][N]
    float *hist_value;     // optional [hist_cap][NSENS][N]
    uint8_t *hist_status, *hist_fault;
    int hist_cap;
    int32_t *hist_pos;     // [N] next history slot
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
        return scale * (sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2));
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

__global__ __launch_bounds__(64) void sensor_suite_kernel(const SensorArgs a)
{
    const int64_t r = a.r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.r1) return;
    const int64_t N = a.N;
    const int steps = a.tap_count[r];
    if (steps <= 0) return;
    const float fs = a.full_scale[r];
    const double t_first = a.time_end[r] - a.t_enable[r] - (double)(steps - 1) * a.dt;   // time of the first read

    // two delay lines (inlet: sensors 0 and 5, outlet: 1 and 6)
    int head[2] = {a.ring_head[0 * N + r], a.ring_head[1 * N + r]};
    int cnt[2] = {a.ring_cnt[0 * N + r], a.ring_cnt[1 * N + r]};

    for (int i = 0; i < NSENS; ++i) {
        const SensorSpec sp = spec_of(i, fs);
        float *F = a.fs + ((int64_t)i * NF) * N + r;
        double *D = a.ds + ((int64_t)i * ND) * N + r;
        int32_t *I = a.is + ((int64_t)i * NI) * N + r;
        float current = F[F_CURRENT * N], supply = F[F_SUPPLY * N];
        const float cal_offset = F[F_CAL_OFFSET * N];
        float last_value = F[F_LAST_VALUE * N];
        const double cal_time = D[D_CAL_TIME * N], power_on = D[D_POWER_ON * N];
        double last_t = D[D_LAST_T * N], prev_t = D[D_PREV_T * N];
        double slow0 = D[D_SLOW0 * N], slow1 = D[D_SLOW1 * N], slow2 = D[D_SLOW2 * N];
        int status = I[I_STATUS * N], fault = I[I_FAULT * N], hist_n = I[I_HIST_N * N];
        Rng rng = {(uint32_t)(a.reactor_base + r), (uint32_t)i, (uint32_t)I[I_DRAWS * N], a.seed_lo, a.seed_hi};
        float slope_pct = 100.0f;   // recomputed from the calibration age at every read (ph_sensor.py:262-266)

        // NOTE: the reference reads all seven sensors per step; sensors only interact through the delay
        // lines, and there the order of pushes matters.  Looping sensor-major over the steps of a launch
        // keeps each sensor's state in registers, so the lines are processed in a second, step-major
        // pass below for the four sensors that use them.  Sensors without a line are independent.
        if (sp.line >= 0) {
            F[F_CURRENT * N] = current;  // (untouched here; handled in the step-major pass)
            continue;
        }
        for (int k = 0; k < steps; ++k) {
            const double t = t_first + (double)k * a.dt;
            const float *tap = a.taps + ((int64_t)k * NTAP) * N + r;
            float value = __builtin_nanf(""); int rstatus, rfault;
            // ---------------- BaseSensor.read
            if (!(20.0f < supply && supply < 28.0f)) {                           // :549-569 (and stays so)
                rstatus = ST_POWER_FAULT; rfault = (supply < 20.0f) ? FL_POWER_LOW : FL_POWER_HIGH;
                prev_t = last_t; last_t = t; last_value = value; hist_n = min(hist_n + 1, 2);
            } else {
                supply = 24.0f + rng.normal(1.0f);                               // :572
                if (!(t - power_on >= (double)sp.warmup)) {                      // :575-588
                    rstatus = ST_WARMING_UP; rfault = FL_NONE;
                    prev_t = last_t; last_t = t; last_value = value; hist_n = min(hist_n + 1, 2);
                } else {
                    const bool cal_expired = ((t - cal_time) / 3600.0 > (double)sp.cal_valid_h);  // :590-593
                    if (cal_expired) status = ST_CAL_EXPIRED;
                    // true value
                    float tv;
                    if (sp.kind == K_CL_AMP || sp.kind == K_CL_DPD) {            // chlorine_sensor.py:189-227
                        const float ratio = exp10f(7.5f - tap[sp.tap_pH * N]);
                        tv = tap[sp.tap_self * N] * (0.5f + 0.5f * (ratio / (1.0f + ratio)));
                    } else {
                        tv = tap[sp.tap_self * N];                               // flow_sensor.py:98-102
                    }
                    const float drift = sp.drift_rate * (float)((t - cal_time) / 3600.0) + cal_offset;  // :612-616
                    const float noise = rng.normal(sp.precision);                // :619
                    float cur = 0.5f * (tv + noise + drift) + 0.5f * current;    // :622-626 (hysteresis :630 is a no-op)
                    float rate = 0.0f;                                           // :638-648
                    if (hist_n > 0) {
                        const float dtl = (float)(t - last_t);
                        if (dtl > 0.0f && isfinite(last_value)) rate = (cur - last_value) / dtl;
                    }
                    int f = -1;                                                  // _check_for_faults :377-407
                    if (!(20.0f < supply && supply < 28.0f)) f = (supply < 20.0f) ? FL_POWER_LOW : FL_POWER_HIGH;
                    else {
                        const float span = sp.hi - sp.lo;
                        if (cur < sp.lo - 0.1f * span || cur > sp.hi + 0.1f * span) f = FL_OUT_OF_RANGE;
                        else if (fabsf(rate) > sp.max_rate) f = FL_RATE_FAULT;
                        else if (rng.uniform() < 0.0001f) f = (rng.uniform() * 2.0f < 1.0f) ? FL_OPEN_CIRCUIT : FL_SHORT_CIRCUIT;
                    }
                    if (f >= 0) {                                                // :651-663
                        fault = f;
                        if (f == FL_OPEN_CIRCUIT || f == FL_SHORT_CIRCUIT) { status = ST_FAILED; cur = __builtin_nanf(""); }
                        else if (f == FL_OUT_OF_RANGE) status = ST_OUT_OF_RANGE;
                        else if (f == FL_POWER_LOW || f == FL_POWER_HIGH) status = ST_POWER_FAULT;
                        else status = ST_RATE_FAULT;
                    } else {                                                     // :664-682
                        fault = FL_NONE;
                        if (!isnan(cur)) {
                            const float b = fminf(fmaxf(cur, sp.lo), sp.hi);
                            if (b != cur) status = ST_SATURATED; else if (!cal_expired) status = ST_NORMAL;
                            cur = b;
                        }
                        if (fabsf(drift) > 0.1f * (sp.hi - sp.lo) && status != ST_CAL_EXPIRED) status = ST_DRIFT_WARNING;
                    }
                    current = cur;
                    const bool have_dt = hist_n >= 1;                            // len(reading_history) >= 2 after the append
                    const double dtp = t - last_t;
                    prev_t = last_t; last_t = t; last_value = cur; hist_n = min(hist_n + 1, 2);
                    rstatus = status; rfault = fault; value = cur;
                    // ---------------- type-specific read()
                    if (isfinite(cur)) {
                        float fin;
                        if (sp.kind == K_CL_AMP) {                               // chlorine_sensor.py:310-331,405-449
                            if (have_dt) { slow0 = fmin(1.0, slow0 + 0.01 * (dtp / 86400.0)); slow1 += dtp / 86400.0; }
                            const float pol = rng.normal(0.005f * (1.0f + (float)slow1 / 365.0f));
                            const float dif = rng.normal(0.003f);
                            fin = cur * (1.0f - 0.8f * (float)slow0) + pol + dif;
                        } else if (sp.kind == K_CL_DPD) {                        // chlorine_sensor.py:274-308,451-484
                            if (have_dt) {
                                slow1 += dtp / 3600.0;
                                const double photo = 1.0 + 0.1 * (slow1 / 100.0);
                                slow0 = fmax(0.0, slow0 - 1.0 * photo * 0.01 * (dtp / 86400.0));
                                slow2 += dtp / 86400.0;
                            }
                            fin = cur * (float)slow0 * 0.95f + rng.normal(0.005f);
                        } else {                                                 // flow_sensor.py:138-178,201-219
                            if (have_dt) slow0 += 0.001 * (dtp / 86400.0);
                            fin = cur * fmaxf(0.9f, 1.0f - 0.005f * (float)slow0) + rng.normal(0.001f * fs);
                            if (fin < 0.01f * fs) fin = 0.0f;
                        }
                        fin = fminf(fmaxf(fin, sp.lo), sp.hi);
                        current = fin; last_value = fin; value = fin;
                    }
                }
            }
            if (k == steps - 1) { a.out_value[(int64_t)i * N + r] = value; a.out_status[(int64_t)i * N + r] = (uint8_t)rstatus; a.out_fault[(int64_t)i * N + r] = (uint8_t)rfault; }
            if (a.hist_value) {
                const int pos = a.hist_pos[r] + k;
                if (pos < a.hist_cap) {
                    const int64_t o = ((int64_t)pos * NSENS + i) * N + r;
                    a.hist_value[o] = value; a.hist_status[o] = (uint8_t)rstatus; a.hist_fault[o] = (uint8_t)rfault;
                }
            }
        }
        F[F_CURRENT * N] = current; F[F_SUPPLY * N] = supply; F[F_LAST_VALUE * N] = last_value;
        D[D_LAST_T * N] = last_t; D[D_PREV_T * N] = prev_t; D[D_SLOW0 * N] = slow0; D[D_SLOW1 * N] = slow1; D[D_SLOW2 * N] = slow2;
        I[I_STATUS * N] = status; I[I_FAULT * N] = fault; I[I_HIST_N * N] = hist_n; I[I_DRAWS * N] = (int32_t)rng.draws;
        (void)slope_pct;
    }

    // ---------------- step-major pass for the four sensors that sit on (shared) delay lines:
    // per step: pH_inlet (0), pH_outlet (1), ..., temp_inlet (5), temp_outlet (6) in suite order
    const int lined[4] = {0, 1, 5, 6};
    float current[4], supply[4], cal_offset[4], last_value[4];
    double cal_time[4], power_on[4], last_t[4], prev_t[4], slow0[4], slow1[4], slow2[4];
    int status[4], fault[4], hist_n[4];
    Rng rng[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lined[j];
        const float *F = a.fs + ((int64_t)i * NF) * N + r; const double *D = a.ds + ((int64_t)i * ND) * N + r;
        const int32_t *I = a.is + ((int64_t)i * NI) * N + r;
        current[j] = F[F_CURRENT * N]; supply[j] = F[F_SUPPLY * N]; cal_offset[j] = F[F_CAL_OFFSET * N]; last_value[j] = F[F_LAST_VALUE * N];
        cal_time[j] = D[D_CAL_TIME * N]; power_on[j] = D[D_POWER_ON * N]; last_t[j] = D[D_LAST_T * N]; prev_t[j] = D[D_PREV_T * N];
        slow0[j] = D[D_SLOW0 * N]; slow1[j] = D[D_SLOW1 * N]; slow2[j] = D[D_SLOW2 * N];
        status[j] = I[I_STATUS * N]; fault[j] = I[I_FAULT * N]; hist_n[j] = I[I_HIST_N * N];
        rng[j] = {(uint32_t)(a.reactor_base + r), (uint32_t)i, (uint32_t)I[I_DRAWS * N], a.seed_lo, a.seed_hi};
    }
    for (int k = 0; k < steps; ++k) {
        const double t = t_first + (double)k * a.dt;
        const float *tap = a.taps + ((int64_t)k * NTAP) * N + r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lined[j];
            const SensorSpec sp = spec_of(i, fs);
            float value = __builtin_nanf(""); int rstatus, rfault;
            if (!(20.0f < supply[j] && supply[j] < 28.0f)) {
                rstatus = ST_POWER_FAULT; rfault = (supply[j] < 20.0f) ? FL_POWER_LOW : FL_POWER_HIGH;
                prev_t[j] = last_t[j]; last_t[j] = t; last_value[j] = value; hist_n[j] = min(hist_n[j] + 1, 2);
            } else {
                supply[j] = 24.0f + rng[j].normal(1.0f);
                if (!(t - power_on[j] >= (double)sp.warmup)) {
                    rstatus = ST_WARMING_UP; rfault = FL_NONE;
                    prev_t[j] = last_t[j]; last_t[j] = t; last_value[j] = value; hist_n[j] = min(hist_n[j] + 1, 2);
                } else {
                    const bool cal_expired = ((t - cal_time[j]) / 3600.0 > (double)sp.cal_valid_h);
                    if (cal_expired) status[j] = ST_CAL_EXPIRED;
                    const float temp = tap[sp.tap_T * N];
                    float tv = (sp.kind == K_PH) ? tap[sp.tap_self * N] + 0.003f * (temp - 25.0f)    // ph_sensor.py:169-180
                                                 : tap[sp.tap_self * N];                             // temperature_sensor.py:105-108
                    {   // SampleLine.transport_sample base_sensor.py:177-216: push, then closest to t - 30 s
                        const int ln = sp.line;
                        float *rt = a.ring_t + ((int64_t)ln * RING) * N + r, *rv = a.ring_v + ((int64_t)ln * RING) * N + r;
                        rt[(int64_t)head[ln] * N] = (float)t; rv[(int64_t)head[ln] * N] = tv;
                        head[ln] = (head[ln] + 1) % RING; cnt[ln] = min(cnt[ln] + 1, RING);
                        const float target = (float)t - 30.0f;
                        int idx = (head[ln] - cnt[ln] + RING) % RING;          // oldest entry
                        float best = fabsf(rt[(int64_t)idx * N] - target); int besti = idx;
                        for (int q = 1; q < cnt[ln]; ++q) {                      // deque order, strict '<': first minimum wins
                            idx = (idx + 1 == RING) ? 0 : idx + 1;
                            const float dq = fabsf(rt[(int64_t)idx * N] - target);
                            if (dq < best) { best = dq; besti = idx; }
                        }
                        tv = rv[(int64_t)besti * N];
                    }
                    const float drift = sp.drift_rate * (float)((t - cal_time[j]) / 3600.0) + cal_offset[j];
                    const float noise = rng[j].normal(sp.precision);
                    float cur = 0.5f * (tv + noise + drift) + 0.5f * current[j];
                    float rate = 0.0f;
                    if (hist_n[j] > 0) {
                        const float dtl = (float)(t - last_t[j]);
                        if (dtl > 0.0f && isfinite(last_value[j])) rate = (cur - last_value[j]) / dtl;
                    }
                    int f = -1;
                    if (!(20.0f < supply[j] && supply[j] < 28.0f)) f = (supply[j] < 20.0f) ? FL_POWER_LOW : FL_POWER_HIGH;
                    else {
                        const float span = sp.hi - sp.lo;
                        if (cur < sp.lo - 0.1f * span || cur > sp.hi + 0.1f * span) f = FL_OUT_OF_RANGE;
                        else if (fabsf(rate) > sp.max_rate) f = FL_RATE_FAULT;
                        else if (rng[j].uniform() < 0.0001f) f = (rng[j].uniform() * 2.0f < 1.0f) ? FL_OPEN_CIRCUIT : FL_SHORT_CIRCUIT;
                    }
                    if (f >= 0) {
                        fault[j] = f;
                        if (f == FL_OPEN_CIRCUIT || f == FL_SHORT_CIRCUIT) { status[j] = ST_FAILED; cur = __builtin_nanf(""); }
                        else if (f == FL_OUT_OF_RANGE) status[j] = ST_OUT_OF_RANGE;
                        else if (f == FL_POWER_LOW || f == FL_POWER_HIGH) status[j] = ST_POWER_FAULT;
                        else status[j] = ST_RATE_FAULT;
                    } else {
                        fault[j] = FL_NONE;
                        if (!isnan(cur)) {
                            const float b = fminf(fmaxf(cur, sp.lo), sp.hi);
                            if (b != cur) status[j] = ST_SATURATED; else if (!cal_expired) status[j] = ST_NORMAL;
                            cur = b;
                        }
                        if (fabsf(drift) > 0.1f * (sp.hi - sp.lo) && status[j] != ST_CAL_EXPIRED) status[j] = ST_DRIFT_WARNING;
                    }
                    current[j] = cur;
                    const bool have_dt = hist_n[j] >= 1;
                    const double dtp = t - last_t[j];
                    prev_t[j] = last_t[j]; last_t[j] = t; last_value[j] = cur; hist_n[j] = min(hist_n[j] + 1, 2);
                    rstatus = status[j]; rfault = fault[j]; value = cur;
                    if (isfinite(cur)) {
                        float fin;
                        if (sp.kind == K_PH) {                                   // ph_sensor.py:182-214,236-336
                            // slow0 = membrane_fouling, slow1 = days_since_cleaning, slow2 = reference_contamination
                            if (have_dt) {
                                const double bio = (slow0[j] > 0.05) ? 0.1 * exp(0.05 * ((double)temp - 25.0)) : 0.001;
                                slow0[j] = fmin(1.0, slow0[j] + (bio + 100.0 * 0.00001) * (dtp / 86400.0));
                                slow1[j] += dtp / 86400.0;
                            }
                            const float elec = rng[j].normal(0.002f * (1.0f + 0.1f * fabsf(cur - 7.0f)));
                            const float junc = rng[j].normal(0.005f * (1.0f + (float)slow2[j]));
                            const double days = (t - cal_time[j]) / 86400.0;
                            const float slope_pct = fmaxf(90.0f, 100.0f - 0.001f * (float)days);
                            float slope_err = 0.0f;
                            if (!(4.0f < cur && cur < 7.0f)) slope_err = fminf(fabsf(cur - 4.0f), fabsf(cur - 7.0f)) * (100.0f - slope_pct) / 100.0f;
                            const float foul_off = (float)slow0[j] * 0.2f;
                            const float foul_noise = rng[j].normal((float)slow0[j] * 0.05f);
                            slow2[j] = fmin(0.5, slow2[j] + 0.0001 * (days / 30.0));
                            fin = cur + elec + junc + slope_err + foul_off + foul_noise + (float)slow2[j] * 0.1f;
                        } else {                                                 // temperature_sensor.py:149-171,118-128
                            const float R_meas = 100.0f * (1.0f + 0.00385f * cur) + 2.0f * 0.5f;
                            const float power_mW = (1.0e-3f * 1.0e-3f) * R_meas * 1000.0f;
                            const float T_meas = (R_meas / 100.0f - 1.0f) / 0.00385f;
                            fin = T_meas + 0.001f * power_mW + rng[j].normal(0.001f);
                            fin += 0.01f * (cur - AMBIENT_T);
                        }
                        fin = fminf(fmaxf(fin, sp.lo), sp.hi);
                        current[j] = fin; last_value[j] = fin; value = fin;
                    }
                }
            }
            if (k == steps - 1) { a.out_value[(int64_t)i * N + r] = value; a.out_status[(int64_t)i * N + r] = (uint8_t)rstatus; a.out_fault[(int64_t)i * N + r] = (uint8_t)rfault; }
            if (a.hist_value) {
                const int pos = a.hist_pos[r] + k;
                if (pos < a.hist_cap) {
                    const int64_t o = ((int64_t)pos * NSENS + i) * N + r;
                    a.hist_value[o] = value; a.hist_status[o] = (uint8_t)rstatus; a.hist_fault[o] = (uint8_t)rfault;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lined[j];
        float *F = a.fs + ((int64_t)i * NF) * N + r; double *D = a.ds + ((int64_t)i * ND) * N + r; int32_t *I = a.is + ((int64_t)i * NI) * N + r;
        F[F_CURRENT * N] = current[j]; F[F_SUPPLY * N] = supply[j]; F[F_LAST_VALUE * N] = last_value[j];
        D[D_LAST_T * N] = last_t[j]; D[D_PREV_T * N] = prev_t[j]; D[D_SLOW0 * N] = slow0[j]; D[D_SLOW1 * N] = slow1[j]; D[D_SLOW2 * N] = slow2[j];
        I[I_STATUS * N] = status[j]; I[I_FAULT * N] = fault[j]; I[I_HIST_N * N] = hist_n[j]; I[I_DRAWS * N] = (int32_t)rng[j].draws;
    }
    a.ring_head[0 * N + r] = head[0]; a.ring_head[1 * N + r] = head[1];
    a.ring_cnt[0 * N + r] = cnt[0]; a.ring_cnt[1 * N + r] = cnt[1];
    if (a.hist_value) a.hist_pos[r] += steps;
}

// initialize_sensors (__main__.py:84-118): construct the suite and calibrate every sensor at the
// current time: offset = reference - current_value, warm-up restarts (base_sensor.py:701-755)
struct SensorInitArgs {
    int64_t N;
    const double *cfg_flow, *cfg_cl, *cfg_temp; // [N] ReactorConfiguration.flow_rate / initial_chlorine / temperature
    float *fs; double *ds; int32_t *is; float *full_scale;
    int32_t *ring_head, *ring_cnt;
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
    a.ring_head[r] = 0; a.ring_head[N + r] = 0; a.ring_cnt[r] = 0; a.ring_cnt[N + r] = 0;
    if (a.hist_pos) a.hist_pos[r] = 0;
}

} // namespace wts
