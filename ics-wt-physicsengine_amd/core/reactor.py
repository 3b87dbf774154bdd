"""Host-side mirror of the reference's reactor API over the HIP ensemble library.

``IntegratedCSTR(config).step(dt, boundary) -> ReactorState`` keeps the
reference's signature, field names, defaults and error behaviour
(/root/reference/src/wt_simulator/core/reactor.py:52-186, 203-227, 450-541) so
the unchanged sensor and Modbus layers can sit on top of it.
``ReactorEnsemble`` is the batched front door the reference does not have: N
independent reactors advanced by one kernel launch.  All arithmetic of the step
runs in ``csrc/`` on the GPU; this module only packs SoA blocks, moves buffers
and turns status bits back into the reference's exceptions and log lines.
"""
from __future__ import annotations

import ctypes as C
import logging
from dataclasses import dataclass, field, fields
from typing import Dict, Iterable, List, Optional, Sequence, Union

import numpy as np

from . import _native, params

logger = logging.getLogger(__name__)

ST_T_RANGE, ST_SOLVER_FAILED, ST_CLAMP_PH, ST_CLAMP_CL, ST_CLAMP_T, ST_T_RANGE_POST, ST_NONFINITE = (
    1, 2, 4, 8, 16, 32, 64)
ST_STEP_LIMIT = 128

_T_RANGE_TEXT = (
    "Temperature {value}°C outside liquid water range [0.0, 100.0]°C. This indicates either:\n"
    "  1. Invalid input data\n"
    "  2. Numerical instability in ODE integration (reduce tolerances)\n"
    "  3. System requires pressurized/supercooled water model")
_SOLVER_FAILED_TEXT = "ODE solver failed: Required step size is less than spacing between numbers."
_NONFINITE_TEXT = "All components of the initial state `y0` must be finite."   # scipy base.py:19-20, out of reactor.py:476


# --------------------------------------------------------------------------- dataclasses
@dataclass
class ReactorConfiguration:
    """Geometry, flow, chemistry and operating point of one CSTR (reactor.py:52-110)."""

    volume: float = 1000.0            # [L]
    height: float = 2.0               # [m]
    diameter: float = 0.798           # [m]
    n_zones: int = 5

    flow_rate: float = 5.0            # [L/min]
    turbulent_intensity: float = 0.15
    recirculation_ratio: float = 5.0
    impeller_speed: float = 60.0      # [rpm]
    impeller_diameter: float = 0.3    # [m]
    power_number: float = 5.0

    initial_pH: float = 7.0
    alkalinity: float = 100.0         # [mg/L as CaCO3]
    total_carbonate: float = 2.0      # [mmol/L]

    initial_chlorine: float = 2.0     # [mg/L]

    temperature: float = 20.0         # [degC]
    enable_thermal_stratification: bool = True

    inlet_pH: float = 7.5
    inlet_chlorine: float = 0.0
    inlet_temperature: float = 20.0

    def validate(self) -> None:
        """Same checks, same exception types as reactor.py:91-110."""
        calculated_volume = np.pi * (self.diameter / 2) ** 2 * self.height * 1000
        volume_error = abs(calculated_volume - self.volume) / self.volume
        if volume_error > 0.01:
            raise ValueError(
                f"Volume mismatch: specified {self.volume}L, "
                f"calculated {calculated_volume:.1f}L from geometry. "
                f"Error: {volume_error*100:.1f}%")
        assert 0 < self.volume < 1e6, "Volume out of range"
        assert 0 <= self.flow_rate < 1e5, "Flow rate out of range (use 0 for batch mode)"
        assert 0 <= self.initial_pH <= 14, "pH out of range"
        assert 0 <= self.initial_chlorine <= 10, "Chlorine out of range"
        assert 0 <= self.temperature <= 40, "Temperature out of typical range"


@dataclass
class ReactorState:
    """State of one reactor; every per-zone quantity is an array of length n_zones
    (reactor.py:113-147)."""

    time: float = 0.0
    pH: np.ndarray = field(default_factory=lambda: np.full(5, 7.0))
    chlorine: np.ndarray = field(default_factory=lambda: np.full(5, 2.0))
    temperature: np.ndarray = field(default_factory=lambda: np.full(5, 20.0))
    flow_rate: float = 5.0

    H_concentration: np.ndarray = field(init=False)
    density: np.ndarray = field(init=False)
    chlorine_decay_rate: np.ndarray = field(init=False)

    def __post_init__(self):
        self.update_derived()

    def update_derived(self):
        self.H_concentration = 10 ** (-self.pH)
        if not hasattr(self, "density"):
            self.density = np.full_like(self.pH, 998.2)
        if not hasattr(self, "chlorine_decay_rate"):
            self.chlorine_decay_rate = np.full_like(self.pH, 0.0001)

    def state_dict(self) -> Dict[str, object]:
        """Plain-dict view (the 'state-dict API' BASELINE.json mentions)."""
        return {f.name: getattr(self, f.name) for f in fields(self)}


@dataclass
class BoundaryConditions:
    """Physical streams entering the tank during a step (reactor.py:150-186)."""

    inlet_flow_rate: float = 5.0          # [L/min]
    inlet_pH: float = 7.5
    inlet_chlorine: float = 0.0           # [mg/L]
    inlet_temperature: float = 20.0       # [degC]
    acid_flow_rate: float = 0.0           # [L/min]
    acid_concentration: float = 0.1       # [mol/L]
    chlorine_flow_rate: float = 0.0       # [L/min]
    chlorine_concentration: float = 50.0  # [mg/L]
    ambient_temperature: float = 20.0     # [degC]
    heat_loss_coefficient: float = 0.0    # [W/K]


@dataclass
class EnsembleState:
    """State of N reactors; per-zone arrays are (N, n_zones), per-reactor arrays (N,)."""

    time: np.ndarray
    pH: np.ndarray
    chlorine: np.ndarray
    temperature: np.ndarray
    flow_rate: np.ndarray
    H_concentration: np.ndarray
    density: np.ndarray
    chlorine_decay_rate: np.ndarray
    status: np.ndarray

    def reactor(self, r: int) -> ReactorState:
        s = ReactorState(time=float(self.time[r]), pH=self.pH[r].copy(), chlorine=self.chlorine[r].copy(),
                         temperature=self.temperature[r].copy(), flow_rate=float(self.flow_rate[r]))
        s.H_concentration = self.H_concentration[r].copy()
        s.density = self.density[r].copy()
        s.chlorine_decay_rate = self.chlorine_decay_rate[r].copy()
        return s


_CFG_FIELDS = [f.name for f in fields(ReactorConfiguration)]


def _columns_from_configs(configs: Sequence[ReactorConfiguration]) -> Dict[str, np.ndarray]:
    return {name: np.array([getattr(c, name) for c in configs]) for name in _CFG_FIELDS if name != "n_zones"}


def boundary_block(boundaries, n_reactors: int) -> np.ndarray:
    """(NB, N) float64 SoA block from a BoundaryConditions, a sequence of them, a
    dict of columns (scalars broadcast) or a ready block."""
    NB = params.NB
    if isinstance(boundaries, np.ndarray):
        blk = np.ascontiguousarray(boundaries, dtype=np.float64)
        if blk.shape != (NB, n_reactors):
            raise ValueError(f"boundary block must have shape {(NB, n_reactors)}, got {blk.shape}")
        return blk
    blk = np.empty((NB, n_reactors), dtype=np.float64)
    if isinstance(boundaries, BoundaryConditions):
        for i, name in enumerate(params.BOUNDARY_FIELDS):
            blk[i] = getattr(boundaries, name)
    elif isinstance(boundaries, dict):
        defaults = BoundaryConditions()
        for i, name in enumerate(params.BOUNDARY_FIELDS):
            blk[i] = np.asarray(boundaries.get(name, getattr(defaults, name)), dtype=np.float64)
    else:
        seq = list(boundaries)
        if len(seq) != n_reactors:
            raise ValueError(f"expected {n_reactors} boundary conditions, got {len(seq)}")
        for i, name in enumerate(params.BOUNDARY_FIELDS):
            blk[i] = [getattr(b, name) for b in seq]
    return blk


# --------------------------------------------------------------------------- ensemble
class ReactorEnsemble:
    """N independent multi-zone CSTRs resident on one MI355X.

    The whole of ``IntegratedCSTR.step`` (RHS + adaptive Radau + derived + clamps)
    runs inside one kernel; the host never sees intermediate values.
    """

    def __init__(self, configs: Union[Sequence[ReactorConfiguration], Dict[str, np.ndarray]],
                 n_zones: Optional[int] = None, device: int = 0, validate: bool = True):
        if isinstance(configs, dict):
            if n_zones is None:
                raise ValueError("n_zones is required with column input")
            cols = {k: np.atleast_1d(np.asarray(v)) for k, v in configs.items()}
            N = max(v.shape[0] for v in cols.values())
            defaults = ReactorConfiguration()
            for name in _CFG_FIELDS:
                if name == "n_zones":
                    continue
                v = cols.get(name, np.asarray([getattr(defaults, name)]))
                cols[name] = np.broadcast_to(v, (N,)).copy()
            if validate:
                _validate_columns(cols)
        else:
            configs = list(configs)
            if not configs:
                raise ValueError("need at least one reactor")
            nz = {c.n_zones for c in configs}
            if len(nz) != 1:
                raise ValueError("all reactors of one ensemble must have the same n_zones")
            n_zones = nz.pop()
            if validate:
                for c in configs:
                    c.validate()
            cols = _columns_from_configs(configs)
            N = len(configs)
        if n_zones < 2:
            raise ValueError(f"Need at least 2 zones, got {n_zones}")       # transport.py:88-89
        self.n_reactors = int(N)
        self.n_zones = int(n_zones)
        self.device = int(device)
        self.columns = cols
        self.constants = params.derive_constants(cols, self.n_zones)
        self._h = C.c_void_p()
        L = _native.lib()
        _native.check(L.wt_ensemble_create(self.n_reactors, self.n_zones, self.device,
                                           _native.dptr(np.ascontiguousarray(self.constants)), C.byref(self._h)))
        shape = (self.n_reactors, self.n_zones)
        self.set_state(np.broadcast_to(np.asarray(cols["initial_pH"], dtype=np.float64)[:, None], shape),
                       np.broadcast_to(np.asarray(cols["initial_chlorine"], dtype=np.float64)[:, None], shape),
                       np.broadcast_to(np.asarray(cols["temperature"], dtype=np.float64)[:, None], shape),
                       np.zeros(self.n_reactors))
        self._boundary: Optional[np.ndarray] = None
        self._device_boundary_moved = False     # the command path (plant I/O) rewrites the device's boundary block
        self._plant_io = False

    # -- lifetime
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            _native.lib().wt_ensemble_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data movement
    def set_state(self, pH, chlorine, temperature, time=None) -> None:
        shape = (self.n_reactors, self.n_zones)
        a = [np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float64), shape)) for x in (pH, chlorine, temperature)]
        t = None if time is None else np.ascontiguousarray(np.broadcast_to(np.asarray(time, dtype=np.float64), (self.n_reactors,)))
        _native.check(_native.lib().wt_ensemble_set_state(self._h, _native.dptr(a[0]), _native.dptr(a[1]),
                                                          _native.dptr(a[2]), _native.dptr(t)))

    def set_boundary(self, boundaries) -> None:
        blk = boundary_block(boundaries, self.n_reactors)
        if self._boundary is not None and not self._device_boundary_moved and np.array_equal(blk, self._boundary):
            return                      # what the device already holds
        _native.check(_native.lib().wt_ensemble_set_boundary(self._h, _native.dptr(blk)))
        self._boundary = blk.copy()
        self._device_boundary_moved = False

    def step(self, dt: float, boundaries=None, n_steps: int = 1, fused: bool = True,
             download: bool = True) -> Optional[EnsembleState]:
        """Advance every reactor ``n_steps`` outer steps of length ``dt`` [s]."""
        if boundaries is not None:
            self.set_boundary(boundaries)
        if self._boundary is None:
            raise ValueError("boundary conditions have not been set")
        try:
            _native.check(_native.lib().wt_ensemble_step(self._h, float(dt), int(n_steps), 1 if fused else 0))
        except _native.WtError as e:
            if e.code == _native.WT_E_ARG:
                raise ValueError(e.message) from None
            raise
        if getattr(self, "_plant_io", False):
            self._device_boundary_moved = True    # every PLC scan rewrites the device's boundary rows 0 / 4 / 6
        return self.state if download else None

    def set_schedule(self, n_streams: int = 0, chunk_steps: int = 50) -> None:
        """Advance the ensemble as ``n_streams`` contiguous reactor ranges on internal HIP
        streams (0 = the library default), at most ``chunk_steps`` outer steps per launch
        (0 = one launch per call)."""
        _native.check(_native.lib().wt_ensemble_set_schedule(self._h, int(n_streams), int(chunk_steps)))

    def schedule(self) -> Dict[str, object]:
        """The schedule in force: {"mode", "streams", "chunk", "workers", "kernel", "placement"}."""
        m, s, c, w = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        _native.check(_native.lib().wt_ensemble_get_schedule(self._h, C.byref(m), C.byref(s), C.byref(c), C.byref(w)))
        mode = {0: "streams", 1: "queue"}.get(m.value, str(m.value))
        pm = C.c_int(0)
        _native.check(_native.lib().wt_ensemble_get_placement(self._h, C.byref(pm), None))
        rd, hs = C.c_int64(0), C.c_int64(0)
        _native.check(_native.lib().wt_ensemble_placement_info(self._h, C.byref(rd), C.byref(hs)))
        return {"mode": mode, "streams": s.value, "chunk": c.value, "workers": w.value,
                "kernel": "wt::step_kernel", "placement": "adaptive" if pm.value else "identity",
                "redeals": int(rd.value), "cost_history_steps": int(hs.value)}

    def item_steps(self, n_steps: int) -> int:
        """Outer steps a reactor's state stays in registers before it returns to memory in a call of ``n_steps``."""
        return int(_native.lib().wt_ensemble_item_steps(self._h, int(n_steps)))

    def set_step_limit(self, max_attempts: int) -> None:
        """Stop a reactor that needs more than ``max_attempts`` Radau step attempts in one outer step
        (status SOLVER_FAILED | STEP_LIMIT).  0 = unlimited, which is what the reference does."""
        _native.check(_native.lib().wt_ensemble_set_step_limit(self._h, int(max_attempts)))

    def set_placement(self, adaptive: bool) -> None:
        """Which reactors share a wavefront.  ``True`` (default): once 32 outer steps of solver counters are in, the
        next :meth:`step` call re-deals the wavefront slots in order of solver cost, so a wavefront no longer waits for
        one expensive reactor among cheap ones.  ``False``: reactor r in slot r.  Results do not depend on it."""
        _native.check(_native.lib().wt_ensemble_set_placement(self._h, 1 if adaptive else 0))

    def placement(self):
        """(adaptive?, slot -> reactor index table)."""
        mode = C.c_int(0)
        perm = np.empty(self.n_reactors, dtype=np.int32)
        _native.check(_native.lib().wt_ensemble_get_placement(self._h, C.byref(mode), perm.ctypes.data_as(C.POINTER(C.c_int32))))
        return bool(mode.value), perm

    def set_sync(self, sync_outer: bool) -> None:
        _native.check(_native.lib().wt_ensemble_set_sync(self._h, 1 if sync_outer else 0))

    def launch_timing(self, enable: bool = True) -> None:
        _native.check(_native.lib().wt_ensemble_launch_timing(self._h, 1 if enable else 0))

    def launch_stats(self):
        """(number of step-kernel launches, sum of their durations [ms], longest [ms]) since the last call."""
        n, s, m = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
        _native.check(_native.lib().wt_ensemble_launch_stats(self._h, C.byref(n), C.byref(s), C.byref(m)))
        return int(n.value), float(s.value), float(m.value)

    def synchronize(self) -> None:
        _native.check(_native.lib().wt_ensemble_synchronize(self._h))

    @property
    def state(self) -> EnsembleState:
        N, n = self.n_reactors, self.n_zones
        pH, Cl, T = np.empty((N, n)), np.empty((N, n)), np.empty((N, n))
        H, rho, kd = np.empty((N, n)), np.empty((N, n)), np.empty((N, n))
        t, flow = np.empty(N), np.empty(N)
        st = np.zeros(N, dtype=np.uint32)
        _native.check(_native.lib().wt_ensemble_get_snapshot(
            self._h, _native.dptr(pH), _native.dptr(Cl), _native.dptr(T), _native.dptr(t), _native.dptr(flow),
            _native.dptr(H), _native.dptr(rho), _native.dptr(kd), st.ctypes.data_as(C.POINTER(C.c_uint32))))
        return EnsembleState(time=t, pH=pH, chlorine=Cl, temperature=T, flow_rate=flow,
                             H_concentration=H, density=rho, chlorine_decay_rate=kd, status=st)

    def status(self) -> np.ndarray:
        st = np.zeros(self.n_reactors, dtype=np.uint32)
        _native.check(_native.lib().wt_ensemble_get_status(self._h, st.ctypes.data_as(C.POINTER(C.c_uint32))))
        return st

    def bad_temperature(self) -> np.ndarray:
        """(N,) the temperature the reference's ValueError names (thermodynamics.py:151) for reactors whose
        status carries T_RANGE / T_RANGE_POST; undefined for the others."""
        v = np.zeros(self.n_reactors)
        _native.check(_native.lib().wt_ensemble_get_bad_temperature(self._h, _native.dptr(v)))
        return v

    def clear_status(self) -> None:
        _native.check(_native.lib().wt_ensemble_clear_status(self._h))

    def solver_stats(self) -> np.ndarray:
        """(N, 5) int32: nfev, njev, nlu, accepted, rejected of the last outer step."""
        arr = np.zeros((self.n_reactors, 5), dtype=np.int32)
        _native.check(_native.lib().wt_ensemble_get_stats(self._h, arr.ctypes.data_as(C.POINTER(_native.SolverStats))))
        return arr

    def derivatives(self, pH, chlorine, temperature):
        """Batched ``IntegratedCSTR.derivatives`` at the given states; returns (dpH, dCl, dT, flags)."""
        shape = (self.n_reactors, self.n_zones)
        a = [np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float64), shape)) for x in (pH, chlorine, temperature)]
        out = [np.empty(shape) for _ in range(3)]
        fl = np.zeros(self.n_reactors, dtype=np.uint32)
        _native.check(_native.lib().wt_ensemble_rhs(self._h, *[_native.dptr(x) for x in a], *[_native.dptr(x) for x in out],
                                                   fl.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out[0], out[1], out[2], fl

    # -- fused sensor suite (NEXT-1)
    SENSOR_NAMES = ("pH_inlet", "pH_outlet", "chlorine_inlet", "chlorine_outlet", "flow_main", "temp_inlet", "temp_outlet")

    def enable_sensors(self, seed: int = 0x5EED, reactor_base: int = 0, history: int = 0) -> None:
        """Attach the reference's seven-sensor suite (``create_realistic_sensor_suite`` +
        ``initialize_sensors``) to every reactor, calibrated now; every later outer step is followed by
        ``read_all_sensors`` on the device.  ``history`` > 0 keeps that many reads per sensor."""
        c = self.columns
        d = ReactorConfiguration()
        col = lambda k: np.ascontiguousarray(np.broadcast_to(np.asarray(c.get(k, getattr(d, k)), dtype=np.float64), (self.n_reactors,)))
        _native.check(_native.lib().wt_ensemble_sensors_enable(
            self._h, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), int(reactor_base), _native.dptr(col("flow_rate")),
            _native.dptr(col("initial_chlorine")), _native.dptr(col("temperature")), int(history)))
        self._sensor_history = int(history)

    def sensor_readings(self):
        """(values float32 (7, N), status uint8 (7, N), fault uint8 (7, N)) of the last read."""
        N = self.n_reactors
        v = np.empty((7, N), dtype=np.float32); s = np.empty((7, N), dtype=np.uint8); f = np.empty((7, N), dtype=np.uint8)
        u8 = C.POINTER(C.c_uint8)
        _native.check(_native.lib().wt_ensemble_sensors_get(self._h, v.ctypes.data_as(C.POINTER(C.c_float)),
                                                            s.ctypes.data_as(u8), f.ctypes.data_as(u8)))
        return v, s, f

    def sensor_history(self):
        """(values (H, 7, N), status, fault, reads per reactor (N,)) recorded since enable_sensors(history=H)."""
        N, H = self.n_reactors, self._sensor_history
        v = np.empty((H, 7, N), dtype=np.float32); s = np.empty((H, 7, N), dtype=np.uint8); f = np.empty((H, 7, N), dtype=np.uint8)
        n = np.zeros(N, dtype=np.int32)
        u8 = C.POINTER(C.c_uint8)
        _native.check(_native.lib().wt_ensemble_sensors_history(self._h, v.ctypes.data_as(C.POINTER(C.c_float)), s.ctypes.data_as(u8),
                                                                f.ctypes.data_as(u8), n.ctypes.data_as(C.POINTER(C.c_int32))))
        return v, s, f, n

    # -- plant I/O: one virtual Modbus slave per reactor (NEXT-2 / NEXT-3)
    INPUT_REGISTERS = {"pH_inlet": 0, "pH_middle": 2, "pH_outlet": 4, "chlorine_inlet": 6, "chlorine_outlet": 8, "flow_rate": 10,
                       "temperature_inlet": 12, "temperature_outlet": 14, "simulation_time": 100, "system_status": 102}
    HOLDING_REGISTERS = {"acid_flow_rate": 0, "chlorine_flow_rate": 2, "inlet_flow_rate": 4}
    DISCRETE_INPUTS = {"sensor_fault_pH_inlet": 0, "sensor_fault_pH_outlet": 1, "sensor_fault_chlorine": 2}
    IR_WORDS, HR_WORDS = 20, 6

    def enable_plant_io(self) -> None:
        """Keep the reference loop's Modbus data blocks for every reactor on the device: after each launch
        of :meth:`step` (one PLC scan; ``fused=False`` scans every outer step like ``__main__.main``) the
        input image is refreshed from the sensor readings (``update_modbus_inputs``) and the holding image
        is validated into the boundary conditions (``read_modbus_commands`` + ``apply_boundary_conditions``)."""
        _native.check(_native.lib().wt_ensemble_plc_enable(self._h))
        self._plant_io = True
        self._device_boundary_moved = True

    @staticmethod
    def encode_float32(values) -> np.ndarray:
        """``ModbusEncoder.float32_to_registers`` for an array: (..., 2) uint16 = (high word, low word)."""
        bits = np.ascontiguousarray(values, dtype=np.float64).astype(np.float32).view(np.uint32)
        return np.stack([(bits >> 16).astype(np.uint16), (bits & 0xFFFF).astype(np.uint16)], axis=-1)

    @staticmethod
    def decode_float32(words) -> np.ndarray:
        """``ModbusDecoder.registers_to_float32`` for an array (..., 2) of (high, low) words."""
        w = np.asarray(words).astype(np.uint32)
        return ((w[..., 0] << 16) | w[..., 1]).astype(np.uint32).view(np.float32)

    def write_holding(self, words, first_reactor: int = 0) -> None:
        """Overwrite the holding image (count, 6) uint16 of reactors first_reactor.. (what masters wrote)."""
        w = np.ascontiguousarray(words, dtype=np.uint16)
        if w.ndim != 2 or w.shape[1] != self.HR_WORDS:
            raise ValueError(f"holding image must be (count, {self.HR_WORDS}) uint16")
        _native.check(_native.lib().wt_ensemble_plc_write_holding(self._h, w.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                                 int(first_reactor), int(w.shape[0])))

    def write_commands(self, acid_flow_rate, chlorine_flow_rate, inlet_flow_rate, first_reactor: int = 0) -> None:
        """``slave.write_holding_register`` of the three actuator setpoints for a block of reactors
        (arrays: one entry per reactor starting at ``first_reactor``; all scalars: every reactor from there on)."""
        raw = [np.asarray(c, dtype=np.float64) for c in (acid_flow_rate, chlorine_flow_rate, inlet_flow_rate)]
        cols = [np.atleast_1d(c) for c in raw]
        # scalars only: the same setpoints for every reactor from first_reactor on
        cnt = (self.n_reactors - int(first_reactor)) if all(c.ndim == 0 for c in raw) else max(c.shape[0] for c in cols)
        w = np.concatenate([self.encode_float32(np.broadcast_to(c, (cnt,))) for c in cols], axis=1)
        self.write_holding(w, first_reactor)

    def input_image(self):
        """(image (N, 20) uint16, update_ok (N,) bool) -- layout in include/wtphys.h."""
        img = np.empty((self.n_reactors, self.IR_WORDS), dtype=np.uint16); ok = np.empty(self.n_reactors, dtype=np.uint8)
        _native.check(_native.lib().wt_ensemble_plc_read_inputs(self._h, img.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                               ok.ctypes.data_as(C.POINTER(C.c_uint8))))
        return img, ok.astype(bool)

    def input_blocks(self, reactor: int, image: Optional[np.ndarray] = None):
        """The reference's ``ir_block`` (200 words) and ``di_block`` (100 bits) of one reactor as lists."""
        img = self.input_image()[0] if image is None else image
        row = img[reactor]
        ir = [0] * 200; di = [0] * 100
        ir[0:16] = [int(x) for x in row[0:16]]; ir[100:103] = [int(x) for x in row[16:19]]
        for b in range(3):
            di[b] = (int(row[19]) >> b) & 1
        return ir, di

    def boundary(self) -> np.ndarray:
        """Current boundary block (10, N) in BoundaryConditions field order (after the command path)."""
        out = np.empty((len(params.BOUNDARY_FIELDS), self.n_reactors), dtype=np.float64)
        _native.check(_native.lib().wt_ensemble_get_boundary(self._h, _native.dptr(out)))
        return out

    # -- diagnostics (NEXT-4)
    DIAGNOSTIC_FIELDS = ("total_chlorine_mg", "total_H_mol", "total_OH_mol", "charge_balance_mol", "thermal_energy_kJ",
                         "pH_CV", "pH_segregation", "chlorine_CV", "chlorine_segregation", "thermocline_depth_m") + tuple(
        f"{p}_{k}" for p in ("pH", "chlorine", "temperature")
        for k in ("mean_value", "std_value", "max_value", "min_value", "range", "max_gradient", "mean_gradient", "gradient_location"))

    def diagnostics(self, as_dict: bool = True):
        """``validate_conservation`` + ``calculate_mixing_quality`` (pH, chlorine) + ``identify_thermocline``
        (NaN = None) + ``calculate_spatial_gradients`` (pH, chlorine, temperature) of every reactor,
        reduced on the device: dict name -> (N,) array, or the raw (34, N) block."""
        out = np.empty((len(self.DIAGNOSTIC_FIELDS), self.n_reactors), dtype=np.float64)
        _native.check(_native.lib().wt_ensemble_diagnostics(self._h, _native.dptr(out)))
        return dict(zip(self.DIAGNOSTIC_FIELDS, out)) if as_dict else out

    def wave_diag(self) -> Optional[np.ndarray]:
        """Per-wavefront diagnostics of the last launch, (n_waves, 8) int64:
        loop trips, Newton trips, shader clocks, 100 MHz wall ticks, factorize / num_jac /
        deferred-f block executions, spare.  The first
        call only switches recording on and returns None."""
        nw = C.c_int64(0)
        L = _native.lib()
        if not getattr(self, "_diag_on", False):
            _native.check(L.wt_ensemble_wave_diag(self._h, None, 0, C.byref(nw)))
            self._diag_on = True
            return None
        _native.check(L.wt_ensemble_wave_diag(self._h, None, 0, C.byref(nw)))
        out = np.zeros((nw.value, L.wt_wave_diag_slots()), dtype=np.int64)
        _native.check(L.wt_ensemble_wave_diag(self._h, out.ctypes.data_as(C.POINTER(C.c_int64)), nw.value, C.byref(nw)))
        return out

    def export_state_device(self, device_ptr: int) -> None:
        """Async device-to-device copy of [pH, Cl, T] (3, N, n) into caller device memory."""
        _native.check(_native.lib().wt_ensemble_export_state_device(self._h, C.c_void_p(device_ptr)))

    def set_stream(self, hip_stream: int) -> None:
        _native.check(_native.lib().wt_ensemble_set_stream(self._h, C.c_void_p(hip_stream)))

    def timer_start(self) -> None:
        _native.check(_native.lib().wt_ensemble_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float(0.0)
        _native.check(_native.lib().wt_ensemble_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)


def _validate_columns(cols: Dict[str, np.ndarray]) -> None:
    """Vectorised ReactorConfiguration.validate (reactor.py:91-110)."""
    calc = np.pi * (cols["diameter"] / 2) ** 2 * cols["height"] * 1000
    err = np.abs(calc - cols["volume"]) / cols["volume"]
    if np.any(err > 0.01):
        r = int(np.argmax(err > 0.01))
        raise ValueError(f"Volume mismatch: specified {cols['volume'][r]}L, calculated {calc[r]:.1f}L "
                         f"from geometry. Error: {err[r]*100:.1f}%")
    assert np.all((0 < cols["volume"]) & (cols["volume"] < 1e6)), "Volume out of range"
    assert np.all((0 <= cols["flow_rate"]) & (cols["flow_rate"] < 1e5)), "Flow rate out of range (use 0 for batch mode)"
    assert np.all((0 <= cols["initial_pH"]) & (cols["initial_pH"] <= 14)), "pH out of range"
    assert np.all((0 <= cols["initial_chlorine"]) & (cols["initial_chlorine"] <= 10)), "Chlorine out of range"
    assert np.all((0 <= cols["temperature"]) & (cols["temperature"] <= 40)), "Temperature out of typical range"


# --------------------------------------------------------------------------- single-reactor drop-in
class IntegratedCSTR:
    """Drop-in for the reference class of the same name (reactor.py:189-611),
    backed by a one-reactor ensemble on the GPU."""

    #: Radau step attempts per outer step before the solve is given up with a warning.  scipy's solve_ivp has no such
    #: limit, but the step kernel cannot be cancelled: where the solution slides along the 8 degC density jump the
    #: reference needs millions of internal steps (hours), and an unbounded device solve looks like a hang from the
    #: host.  Ten million attempts is far beyond anything a finite reference run was seen to need; pass
    #: ``step_limit=0`` for the reference's unbounded behaviour.
    DEFAULT_STEP_LIMIT = 10_000_000

    def __init__(self, config: ReactorConfiguration, device: int = 0, step_limit: Optional[int] = None):
        config.validate()
        self.config = config
        self._ens = ReactorEnsemble([config], device=device, validate=False)
        self._ens.set_step_limit(self.DEFAULT_STEP_LIMIT if step_limit is None else int(step_limit))
        self._last = None                # (pH, chlorine, temperature, time) as the device holds them, when known
        n = config.n_zones
        self._initialize_physics_modules()
        self.state = ReactorState(
            pH=np.full(n, config.initial_pH),
            chlorine=np.full(n, config.initial_chlorine),
            temperature=np.full(n, config.temperature),
            flow_rate=config.flow_rate)
        # same log line as reactor.py:224-227 (and, as there, a TypeError when
        # flow_rate == 0 because residence_time is None)
        logger.info(f"Reactor initialized: {config.n_zones} zones, "
                    f"V={config.volume}L, τ={self.transport.residence_time:.1f}min")

    def _initialize_physics_modules(self) -> None:
        """The sub-objects the reference builds (reactor.py:229-270): host-side holders of the reactor's init-time
        constants and scalar diagnostics; the step itself runs on the device from ``self._ens.constants``."""
        from .chemistry import AqueousChemistry, BufferSystem
        from .physics import (FlowParameters, GeometryParameters, SpatialModel, StratificationParameters,
                              TemperatureDependentKinetics, TransportModel)
        c = self.config
        self.thermo = TemperatureDependentKinetics()
        self.buffer = BufferSystem(alkalinity=c.alkalinity, total_carbonate=c.total_carbonate, temperature=c.temperature)
        self.chemistry = AqueousChemistry(self.buffer, device=self._ens.device)
        geometry = GeometryParameters(volume=c.volume, height=c.height, diameter=c.diameter, n_zones=c.n_zones)
        flow = FlowParameters(flow_rate=c.flow_rate, turbulent_intensity=c.turbulent_intensity,
                              recirculation_ratio=c.recirculation_ratio, impeller_speed=c.impeller_speed,
                              impeller_diameter=c.impeller_diameter, power_number=c.power_number)
        self.transport = TransportModel(geometry, flow, c.temperature)
        self.spatial = SpatialModel(n_zones=c.n_zones, height=c.height, stratification_params=StratificationParameters(
            enable_thermal_stratification=c.enable_thermal_stratification))

    def derivatives(self, t: float, y: np.ndarray, boundary: BoundaryConditions) -> np.ndarray:
        n = self.config.n_zones
        y = np.asarray(y, dtype=np.float64)
        self._ens.set_boundary(boundary)
        dpH, dCl, dT, fl = self._ens.derivatives(y[None, 0:n], y[None, n:2 * n], y[None, 2 * n:3 * n])
        if fl[0] & ST_T_RANGE:
            T = y[2 * n:3 * n]
            bad = T[(T < 0) | (T > 100)]
            raise ValueError(_T_RANGE_TEXT.format(value=bad[0] if bad.size else "nan"))
        return np.concatenate([dpH[0], dCl[0], dT[0]])

    def step(self, dt: float, boundary: BoundaryConditions) -> ReactorState:
        """Advance by ``dt`` seconds under ``boundary`` (reactor.py:450-509)."""
        ens = self._ens
        s = self.state
        # honour host-side edits of self.state between steps (reactor.py:467-469): upload unless the state is
        # exactly what the device left there
        last = self._last
        if not (last is not None and s.time == last[3] and np.array_equal(s.pH, last[0], equal_nan=True)
                and np.array_equal(s.chlorine, last[1], equal_nan=True) and np.array_equal(s.temperature, last[2], equal_nan=True)):
            ens.set_state(np.asarray(s.pH, dtype=np.float64)[None, :], np.asarray(s.chlorine, dtype=np.float64)[None, :],
                          np.asarray(s.temperature, dtype=np.float64)[None, :], np.array([s.time], dtype=np.float64))
        self._last = None
        es = ens.step(dt, boundary, n_steps=1)
        flags = int(es.status[0])
        if flags & ST_NONFINITE and float(es.time[0]) == float(s.time):
            raise ValueError(_NONFINITE_TEXT)         # solve_ivp refused y0: self.state untouched
        if flags & ST_T_RANGE:
            raise ValueError(_T_RANGE_TEXT.format(value=np.float64(ens.bad_temperature()[0])))
        if flags & ST_STEP_LIMIT:
            logger.warning("ODE solver stopped: internal step-attempt limit reached (see set_step_limit)")
        elif flags & ST_SOLVER_FAILED:
            logger.warning(_SOLVER_FAILED_TEXT)
        pre = None
        s.pH, s.chlorine, s.temperature = es.pH[0].copy(), es.chlorine[0].copy(), es.temperature[0].copy()
        s.time = float(es.time[0])
        s.flow_rate = float(es.flow_rate[0])
        s.H_concentration = es.H_concentration[0].copy()
        s.density = es.density[0].copy()
        if flags & ST_T_RANGE_POST:
            raise ValueError(_T_RANGE_TEXT.format(value=np.float64(ens.bad_temperature()[0])))
        s.chlorine_decay_rate = es.chlorine_decay_rate[0].copy()
        if flags & ST_CLAMP_PH:
            logger.error("pH out of bounds: clipped to [0, 14]")
        if flags & ST_CLAMP_CL:
            logger.warning("Negative chlorine detected: clipped to 0")
        if flags & ST_CLAMP_T:
            logger.error("Temperature out of bounds: clipped to [0, 100]")
        if flags == 0:
            self._last = (s.pH.copy(), s.chlorine.copy(), s.temperature.copy(), s.time)
        return s

    def get_state_at_location(self, zone_idx: int, parameter: str) -> float:
        if zone_idx < 0 or zone_idx >= self.config.n_zones:
            raise ValueError(f"Zone index {zone_idx} out of range [0, {self.config.n_zones-1}]")
        table = {"pH": self.state.pH, "chlorine": self.state.chlorine,
                 "temperature": self.state.temperature, "density": self.state.density}
        if parameter not in table:
            raise ValueError(f"Unknown parameter: {parameter}")
        return table[parameter][zone_idx]

    def _device_diagnostics(self) -> Dict[str, np.ndarray]:
        s = self.state
        self._ens.set_state(np.asarray(s.pH, dtype=np.float64)[None, :], np.asarray(s.chlorine, dtype=np.float64)[None, :],
                            np.asarray(s.temperature, dtype=np.float64)[None, :], np.array([s.time], dtype=np.float64))
        return self._ens.diagnostics()

    def validate_conservation(self) -> Dict[str, float]:
        """Mass / charge / energy inventory of ``self.state`` (reactor.py:570-611), reduced on the device."""
        d = self._device_diagnostics()
        out = {k: float(d[k][0]) for k in ("total_chlorine_mg", "total_H_mol", "total_OH_mol", "charge_balance_mol", "thermal_energy_kJ")}
        out["zones"] = self.config.n_zones
        out["timestamp"] = self.state.time
        return out

    def mixing_quality(self) -> Dict[str, float]:
        """(CV, segregation index) of pH and chlorine as ``print_diagnostics`` reports them
        (transport.py:338-384 via reactor.py:638-639)."""
        d = self._device_diagnostics()
        return {k: float(d[k][0]) for k in ("pH_CV", "pH_segregation", "chlorine_CV", "chlorine_segregation")}


    def print_diagnostics(self) -> None:
        """The report of reactor.py:613-645 (numbers from the device diagnostics)."""
        d = self._device_diagnostics()
        s, n = self.state, self.config.n_zones
        print("\n" + "=" * 70 + "\nCSTR PHYSICS DIAGNOSTICS\n" + "=" * 70)
        print(f"\nTime: {s.time:.1f} s")
        print(f"Residence time: {self.transport.residence_time:.1f} min")
        print(f"Mixing time: {self.transport.mixing_time_seconds:.1f} s")
        print(f"\n{'Zone':<6} {'pH':<8} {'Cl(mg/L)':<10} {'T(°C)':<8} {'ρ(kg/m³)':<10}\n" + "-" * 50)
        for i in range(n):
            print(f"{i:<6} {s.pH[i]:<8.3f} {s.chlorine[i]:<10.3f} {s.temperature[i]:<8.2f} {s.density[i]:<10.2f}")
        print("\nConservation Laws:")
        print(f"  Total Chlorine: {d['total_chlorine_mg'][0]:.2f} mg")
        print(f"  Charge Balance: {d['charge_balance_mol'][0]:.2e} mol")
        print("\nMixing Quality:")
        print(f"  pH segregation index: {d['pH_segregation'][0]:.4f}")
        print(f"  Chlorine segregation index: {d['chlorine_segregation'][0]:.4f}")
        print("=" * 70 + "\n")


PhysicsEngine = IntegratedCSTR  # the name BASELINE.json uses for this API
