"""CPU ORACLE of the reference's sensor suite (TEST INFRASTRUCTURE ONLY).

Pure-Python restatement, for one reactor, of what ``create_realistic_sensor_suite`` +
``initialize_sensors`` + ``read_all_sensors`` do once per outer step
(/root/reference/src/wt_simulator/sensors/__init__.py:41-120, __main__.py:84-163):
seven sensors (pH in/out, chlorine amperometric in / DPD out, magnetic flow, RTD in/out) built on
``BaseSensor.read`` (sensors/base_sensor.py:509-699) with the type-specific ``read`` overrides
(ph_sensor.py:216-336, chlorine_sensor.py:351-484, flow_sensor.py:125-219,
temperature_sensor.py:110-171), including the two sample lines that the suite *shares* between a
pH and a temperature sensor (sensors/__init__.py:62-67,74,108).

The reference draws from ``numpy.random.default_rng(secrets.randbits(128))`` and is therefore not
reproducible.  Parity is pinned by replacing that generator -- in the reference, when the golden
vectors are made (oracle/gen_golden_sensors.py), here and in the HIP kernel -- by the same
counter-based stream: Philox4x32-10 keyed by the suite seed, counter = (reactor, sensor, draw
index).  Every ``rng.normal / rng.random / rng.choice`` call of the reference consumes one draw, in
the reference's call order, so the whole pipeline (noise, faults, warm-up, lag, hysteresis, drift,
delay lines, fouling, ...) can be compared value by value.

Only tests/ and the golden generator import this module.
"""
from __future__ import annotations

import math
from collections import deque
from typing import Dict, List, Tuple

import numpy as np

# ---------------------------------------------------------------- the shared random stream
_M0, _M1 = 0xD2511F53, 0xCD9E8D57
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = 0xFFFFFFFF


def philox4x32_10(counter: Tuple[int, int, int, int], key: Tuple[int, int]) -> Tuple[int, int, int, int]:
    """Philox4x32 with 10 rounds (Salmon et al., SC'11), plain integer arithmetic."""
    c0, c1, c2, c3 = counter
    k0, k1 = key
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = (p0 >> 32) & _MASK, p0 & _MASK
        hi1, lo1 = (p1 >> 32) & _MASK, p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _MASK, lo1, (hi0 ^ c3 ^ k1) & _MASK, lo0
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return c0, c1, c2, c3


class SuiteRng:
    """Drop-in for the three ``numpy.random.Generator`` methods the sensors call."""

    def __init__(self, seed: int, reactor: int, sensor: int):
        self.key = (seed & _MASK, (seed >> 32) & _MASK)
        self.reactor, self.sensor = reactor, sensor
        self.draws = 0

    def _next(self) -> Tuple[int, int, int, int]:
        out = philox4x32_10((self.reactor & _MASK, self.sensor, self.draws & _MASK, 0), self.key)
        self.draws += 1
        return out

    def random(self) -> float:
        x = self._next()
        return (x[0] >> 8) * (1.0 / 16777216.0)

    def normal(self, loc: float = 0.0, scale: float = 1.0) -> float:
        x = self._next()
        u1 = ((x[0] >> 8) + 1) * (1.0 / 16777216.0)      # (0, 1]
        u2 = (x[1] >> 8) * (1.0 / 16777216.0)            # [0, 1)
        z = math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)
        return loc + scale * z

    def choice(self, seq):
        x = self._next()
        return seq[int((x[0] >> 8) * (1.0 / 16777216.0) * len(seq))]


# ---------------------------------------------------------------- codes
# SensorStatus / SensorFault in declaration order (base_sensor.py:49-75)
ST_NORMAL, ST_CALIBRATING, ST_WARMING_UP, ST_FAILED, ST_SATURATED, ST_DRIFT_WARNING, ST_CAL_EXPIRED, \
    ST_OPEN_CIRCUIT, ST_SHORT_CIRCUIT, ST_OUT_OF_RANGE, ST_POWER_FAULT, ST_RATE_FAULT = range(12)
FL_NONE, FL_OPEN_CIRCUIT, FL_SHORT_CIRCUIT, FL_OUT_OF_RANGE, FL_RATE_FAULT, FL_POWER_LOW, FL_POWER_HIGH = range(7)

SENSOR_NAMES = ("pH_inlet", "pH_outlet", "chlorine_inlet", "chlorine_outlet", "flow_main", "temp_inlet", "temp_outlet")
S_PH_IN, S_PH_OUT, S_CL_IN, S_CL_OUT, S_FLOW, S_T_IN, S_T_OUT = range(7)
K_PH, K_CL_AMP, K_CL_DPD, K_FLOW_MAG, K_T_RTD = range(5)


class SampleLine:
    """sensors/base_sensor.py:149-216 (250 mL at 500 mL/min -> 30 s, 100-entry deque)."""

    def __init__(self, volume_mL=250.0, flow_rate_mL_min=500.0, ambient_temp=25.0):
        volume_L = volume_mL / 1000.0
        flow_L_s = flow_rate_mL_min / 1000.0 / 60.0
        self.transport_delay_s = volume_L / flow_L_s if flow_L_s > 0 else 0.0
        self.ambient_temp = ambient_temp
        self.buf = deque(maxlen=max(100, int(self.transport_delay_s) + 10))

    def transport_sample(self, value, temp, timestamp):
        self.buf.append((timestamp, value, temp))
        target = timestamp - self.transport_delay_s
        closest = self.buf[0]
        best = abs(closest[0] - target)
        for s in self.buf:
            d = abs(s[0] - target)
            if d < best:
                best, closest = d, s
        t_s, v_s, temp_s = closest
        frac = math.exp(-0.1 * (timestamp - t_s))
        return v_s, self.ambient_temp + (temp_s - self.ambient_temp) * frac


def _sensor(kind, zone, lo, hi, precision, drift_rate, warmup, hyst, cal_valid_h, max_rate, line, current):
    return dict(kind=kind, zone=zone, lo=lo, hi=hi, precision=precision, drift_rate=drift_rate, warmup=warmup,
                hyst=hyst, cal_valid_h=cal_valid_h, max_rate=max_rate, line=line, current=current,
                status=ST_NORMAL, fault=FL_NONE, supply=24.0, cal_offset=0.0, cal_time=0.0, power_on=0.0,
                have_cal=False, last_dir=0, hist_n=0, last_t=0.0, last_value=float("nan"), prev_t=0.0,
                # type-specific slow state
                fouling=0.0, days_clean=0.0, slope_pct=100.0, ref_contam=0.0,       # pH
                membrane_age=0.0,                                                     # amperometric
                potency=1.0, light_h=0.0, reagent_age=0.0,                            # DPD
                electrode_fouling=0.0)                                                # magnetic flow


class SensorSuite:
    """The seven sensors of one reactor, calibrated at ``t0`` like ``initialize_sensors``."""

    # InstallationQuality of the suite (sensors/__init__.py:53-59)
    flow_velocity, air_bubble_frequency, grounding_quality, pipe_vibration_g, ambient_temperature = 0.5, 0.0, 0.9, 0.1, 30.0

    def __init__(self, cfg_flow_rate: float, cfg_initial_chlorine: float, cfg_temperature: float, t0: float,
                 seed: int, reactor: int):
        inlet, outlet = SampleLine(), SampleLine()
        self.lines = (inlet, outlet)
        fs = cfg_flow_rate * 2.0
        S = [None] * 7
        S[S_PH_IN] = _sensor(K_PH, 0, 0.0, 14.0, 0.01, 0.01 / 24.0, 1800.0, 0.02, 24.0, 0.5, 0, 7.0)
        S[S_PH_OUT] = _sensor(K_PH, -1, 0.0, 14.0, 0.01, 0.01 / 24.0, 1800.0, 0.02, 24.0, 0.5, 1, 7.0)
        S[S_CL_IN] = _sensor(K_CL_AMP, 0, 0.0, 10.0, 0.01, 0.02 / 24.0, 300.0, 0.01, 24.0, 1.0, None, 0.0)
        S[S_CL_OUT] = _sensor(K_CL_DPD, -1, 0.0, 10.0, 0.02, 0.02 / 24.0, 60.0, 0.01, 24.0, 1.0, None, 0.0)
        S[S_FLOW] = _sensor(K_FLOW_MAG, 0, 0.0, fs, 0.005 * fs, 0.0, 10.0, 0.005 * fs, 8760.0, fs, None, 0.0)
        S[S_T_IN] = _sensor(K_T_RTD, 0, -10.0, 110.0, 0.1, 0.0, 30.0, 0.05, 8760.0, 10.0, 0, 20.0)
        S[S_T_OUT] = _sensor(K_T_RTD, -1, -10.0, 110.0, 0.1, 0.0, 30.0, 0.05, 8760.0, 10.0, 1, 20.0)
        self.full_scale = fs
        self.sensors = S
        self.rng = [SuiteRng(seed, reactor, i) for i in range(7)]
        # initialize_sensors (__main__.py:96-105) -> BaseSensor.calibrate (base_sensor.py:701-755)
        refs = (7.0, 7.0, cfg_initial_chlorine, cfg_initial_chlorine, cfg_flow_rate, cfg_temperature, cfg_temperature)
        for s, ref in zip(S, refs):
            s["cal_offset"] = ref - s["current"]
            s["cal_time"] = t0
            s["power_on"] = t0
            s["have_cal"] = True
            s["status"], s["fault"] = ST_NORMAL, FL_NONE

    # ------------------------------------------------------------ BaseSensor.read
    def _true_value(self, s, st):
        pH, Cl, T, flow = st
        z = s["zone"]
        if s["kind"] == K_PH:                       # ph_sensor.py:151-180
            return pH[z] + 0.003 * (T[z] - 25.0)
        if s["kind"] in (K_CL_AMP, K_CL_DPD):       # chlorine_sensor.py:189-227
            ratio = 10 ** (7.5 - pH[z])
            return Cl[z] * (0.5 + 0.5 * (ratio / (1 + ratio)))
        if s["kind"] == K_FLOW_MAG:                 # flow_sensor.py:98-102
            return flow
        return T[z]                                 # temperature_sensor.py:103-108

    def _base_read(self, i, st, t):
        """base_sensor.py:509-699.  Returns (value, status, fault, finite_path)."""
        s, rng = self.sensors[i], self.rng[i]
        if not (20.0 < s["supply"] < 28.0):                                   # :549-569
            status = ST_POWER_FAULT
            fault = FL_POWER_LOW if s["supply"] < 20.0 else FL_POWER_HIGH
            self._append(s, t, float("nan"))
            return float("nan"), status, fault
        s["supply"] = 24.0 + rng.normal(0.0, 1.0)                             # :572
        if not (t - s["power_on"] >= s["warmup"]):                            # :575-588
            self._append(s, t, float("nan"))
            return float("nan"), ST_WARMING_UP, FL_NONE
        cal_expired = not (s["have_cal"] and not ((t - s["cal_time"]) / 3600.0 > s["cal_valid_h"]))  # :590-593
        if cal_expired:
            s["status"] = ST_CAL_EXPIRED
        true_value = self._true_value(s, st)
        if s["line"] is not None:                                             # :598-609
            temp = st[2][s["zone"]]
            true_value, _ = self.lines[s["line"]].transport_sample(true_value, temp, t)
        drift = s["drift_rate"] * ((t - s["cal_time"]) / 3600.0) + s["cal_offset"]   # :612-616
        noise = rng.normal(0.0, s["precision"])                               # :619
        cur = 0.5 * (true_value + noise + drift) + 0.5 * s["current"]         # :622-626
        # hysteresis :438-462 is evaluated AFTER self.current_value was overwritten with the lagged value
        # (:626-630), so its direction is sign(x - x) = 0 and it never changes anything; kept as that no-op
        # installation effects :464-507
        if self.flow_velocity < 0.1:
            cur += rng.normal(0.0, s["precision"] * 2.0)
        if self.air_bubble_frequency > 0:
            if rng.random() < self.air_bubble_frequency / 60.0:
                cur = float("nan")
        if not math.isnan(cur):
            if self.grounding_quality < 0.8:
                cur += rng.normal(0.0, s["precision"] * (2.0 - self.grounding_quality))
            if self.pipe_vibration_g > 0.2:
                cur += rng.normal(0.0, self.pipe_vibration_g * s["precision"])
        s["current"] = cur
        # rate of change :638-648
        rate = 0.0
        if s["hist_n"] > 0:
            dt = t - s["last_t"]
            if dt > 0 and math.isfinite(s["last_value"]):
                rate = (cur - s["last_value"]) / dt
        # fault check :377-407
        fault = None
        if not (20.0 < s["supply"] < 28.0):
            fault = FL_POWER_LOW if s["supply"] < 20.0 else FL_POWER_HIGH
        else:
            span = s["hi"] - s["lo"]
            if cur < s["lo"] - 0.1 * span or cur > s["hi"] + 0.1 * span:
                fault = FL_OUT_OF_RANGE
            elif s["max_rate"] is not None and abs(rate) > s["max_rate"]:
                fault = FL_RATE_FAULT
            elif rng.random() < 0.0001:
                fault = rng.choice([FL_OPEN_CIRCUIT, FL_SHORT_CIRCUIT])
        if fault is not None:                                                 # :651-663
            s["fault"] = fault
            if fault in (FL_OPEN_CIRCUIT, FL_SHORT_CIRCUIT):
                s["status"] = ST_FAILED
                s["current"] = cur = float("nan")
            elif fault == FL_OUT_OF_RANGE:
                s["status"] = ST_OUT_OF_RANGE
            elif fault in (FL_POWER_LOW, FL_POWER_HIGH):
                s["status"] = ST_POWER_FAULT
            else:
                s["status"] = ST_RATE_FAULT
        else:                                                                 # :664-682
            s["fault"] = FL_NONE
            if not math.isnan(cur):
                bounded = min(max(cur, s["lo"]), s["hi"])
                if bounded != cur:
                    s["status"] = ST_SATURATED
                elif not cal_expired:
                    s["status"] = ST_NORMAL
                s["current"] = cur = bounded
            if abs(drift) > 0.1 * (s["hi"] - s["lo"]):
                if s["status"] != ST_CAL_EXPIRED:
                    s["status"] = ST_DRIFT_WARNING
        self._append(s, t, cur)
        return cur, s["status"], s["fault"]

    @staticmethod
    def _append(s, t, value):
        """reading_history.append: keep what later reads look at (last and one-before-last timestamps)."""
        s["prev_t"] = s["last_t"]
        s["last_t"], s["last_value"] = t, value
        s["hist_n"] += 1

    # ------------------------------------------------------------ type-specific read()
    def _read(self, i, st, t):
        s, rng = self.sensors[i], self.rng[i]
        v, status, fault = self._base_read(i, st, t)
        if not math.isfinite(v):
            return v, status, fault
        dt = (t - s["prev_t"]) if s["hist_n"] >= 2 else None
        k = s["kind"]
        if k == K_PH:                                                         # ph_sensor.py:216-336
            temp = st[2][s["zone"]]
            if dt is not None:                                                # _update_fouling :182-214
                bio = 0.1 * math.exp(0.05 * (temp - 25)) if s["fouling"] > 0.05 else 0.001
                scaling = 100.0 * (0.0001 if self.flow_velocity < 0.1 else 0.00001)
                s["fouling"] = min(1.0, s["fouling"] + (bio + scaling) * (dt / 86400.0))
                s["days_clean"] += dt / 86400.0
            elec = rng.normal(0.0, 0.002 * (1.0 + 0.1 * abs(v - 7.0)))
            junc = rng.normal(0.0, 0.005 * (1.0 + s["ref_contam"]))
            days = 0.0
            if s["have_cal"]:
                days = (t - s["cal_time"]) / 86400.0
                s["slope_pct"] = max(90.0, 100.0 - 0.001 * days)
            if 4.0 < v < 7.0:
                slope_err = 0.0
            else:
                slope_err = min(abs(v - 4.0), abs(v - 7.0)) * (100.0 - s["slope_pct"]) / 100.0
            foul_off = s["fouling"] * 0.2
            foul_noise = rng.normal(0.0, s["fouling"] * 0.05)
            s["ref_contam"] = min(0.5, s["ref_contam"] + 0.0001 * (days / 30.0))
            ref_off = s["ref_contam"] * 0.1
            final = v + elec + junc + slope_err + foul_off + foul_noise + ref_off
        elif k == K_CL_AMP:                                                   # chlorine_sensor.py:351-449
            if dt is not None:
                s["fouling"] = min(1.0, s["fouling"] + (0.05 if self.flow_velocity < 0.1 else 0.01) * (dt / 86400.0))
                s["membrane_age"] += dt / 86400.0
            pol = rng.normal(0.0, 0.005 * (1.0 + s["membrane_age"] / 365.0))
            dif = rng.normal(0.0, 0.003)
            final = (v + 0.0) * (1.0 - 0.8 * s["fouling"]) + pol + dif
        elif k == K_CL_DPD:                                                   # chlorine_sensor.py:274-308,451-484
            if dt is not None:
                thermal = math.exp((50000 / 8.314) * (1 / 293.15 - 1 / (20.0 + 273.15)))
                s["light_h"] += dt / 3600.0
                photo = 1.0 + 0.1 * (s["light_h"] / 100.0)
                s["potency"] = max(0.0, s["potency"] - thermal * photo * 0.01 * (dt / 86400.0))
                s["reagent_age"] += dt / 86400.0
            final = v * s["potency"] * 0.95 + rng.normal(0.0, 0.005)
        elif k == K_FLOW_MAG:                                                 # flow_sensor.py:125-219
            if dt is not None:
                s["electrode_fouling"] += 0.001 * (dt / 86400.0)
            fs = self.full_scale
            final = v * max(0.9, 1.0 - 0.005 * s["electrode_fouling"]) * 1.0 + rng.normal(0.0, 0.001 * fs)
            if self.air_bubble_frequency > 0 and rng.random() < self.air_bubble_frequency / 60.0:
                final = 0.0
            if final < 0.01 * fs:
                final = 0.0
        else:                                                                 # temperature_sensor.py:110-171
            R_meas = 100.0 * (1.0 + 0.00385 * v) + 2.0 * 0.5
            power_mW = ((1.0 / 1000.0) ** 2) * R_meas * 1000.0
            T_meas = (R_meas / 100.0 - 1.0) / 0.00385
            final = T_meas + 0.001 * power_mW + rng.normal(0.0, 0.001)
            final += 0.01 * (v - self.ambient_temperature)
        final = min(max(final, s["lo"]), s["hi"])
        s["current"] = final
        s["last_value"] = final                     # reading_history[-1] = final_reading
        return final, status, fault

    def read_all(self, pH, Cl, T, flow, t):
        """read_all_sensors (__main__.py:121-163), dict order = SENSOR_NAMES.  Returns three lists of 7."""
        st = (pH, Cl, T, flow)
        out = [self._read(i, st, t) for i in range(7)]
        return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]
