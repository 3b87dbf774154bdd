"""The reference's physics sub-objects as views over this build's constant columns.

``wt_simulator.core`` exports a handful of classes next to ``IntegratedCSTR`` (core/__init__.py:238-263): the
thermodynamic relations, the transport and spatial models, their parameter records and one self-check per module.
An ``IntegratedCSTR`` carries one instance of each (reactor.py:229-270).  In this build the arithmetic behind them
lives in ``params`` -- written once, column-wise, for N reactors, because that is what fills the constant block the
GPU kernel reads -- and the classes below only give that arithmetic the reference's names, signatures, defaults and
exception texts for N = 1.  Nothing here is on the hot path: ``step()`` / ``derivatives()`` run in ``csrc/``.
Values are bit-identical to what the reference's objects hold (tests/golden/g1_constants.json).
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass
from typing import Callable, Dict, Iterable, Optional, Tuple

import numpy as np

from . import params

R_GAS = params.R_GAS
T_REFERENCE_K = params.T_REFERENCE_K
T_REFERENCE_C = 20.0


def _plain(value):
    """numpy scalar / 0-d array -> Python float, arrays untouched (the reference's methods take and return floats)."""
    value = np.asarray(value)
    return float(value) if value.ndim == 0 else value


def _insist(rules: Iterable[Tuple[bool, str]]) -> None:
    """``validate()`` of the parameter records: the first broken rule raises ValueError with the reference's text."""
    for ok, message in rules:
        if not ok:
            raise ValueError(message)


# --------------------------------------------------------------------------- thermodynamics.py:59-383
@dataclass(frozen=True)
class ArrheniusParameters:
    k_ref: float
    E_a: float
    T_ref: float = T_REFERENCE_K

    def validate(self) -> None:
        _insist(((self.k_ref > 0, f"Rate constant must be positive: {self.k_ref}"),
                 (self.E_a >= 0, f"Activation energy cannot be negative: {self.E_a}"),
                 (self.T_ref > 0, f"Reference temperature must be positive: {self.T_ref}")))


class TemperatureDependentKinetics:
    """Rate and equilibrium constants as functions of temperature.  Every method accepts a float (returns a float, as
    in the reference) or an array of temperatures (returns an array): one vectorised routine in ``params`` serves the
    drop-in and the ensemble's constant block alike."""

    CHLORINE_DECAY = ArrheniusParameters(k_ref=0.0001, E_a=45000.0)
    DELTA_H_WATER, KW_25C = 55900.0, 1.0e-14
    PKA1_25C, PKA2_25C, DPKA_DT = 6.35, 10.33, -0.008
    D_MOLECULAR_REF = 1.0e-9
    T_MIN_C, T_MAX_C = params.T_MIN_C, params.T_MAX_C
    TOLERANCE_KINETICS, TOLERANCE_EQUILIBRIUM, TOLERANCE_PH = 1e-10, 1e-6, 1e-4

    def __init__(self):
        type(self).CHLORINE_DECAY.validate()

    @staticmethod
    def celsius_to_kelvin(temp_c):
        return _plain(params.kelvin(temp_c))

    def arrhenius_rate(self, temp_c, law: ArrheniusParameters):
        return _plain(params.arrhenius(temp_c, law.k_ref, law.E_a, law.T_ref))

    def chlorine_decay_rate(self, temp_c):
        """Per-zone k(T) of the step (csrc/wt_device.hpp: prop_T_n)."""
        law = self.CHLORINE_DECAY
        return _plain(params.arrhenius(temp_c, law.k_ref, law.E_a, law.T_ref))

    def water_ionization_constant(self, temp_c):
        params.kelvin(temp_c)
        return _plain(params.water_ionization_constant(temp_c))

    def neutral_pH(self, temp_c):
        return _plain(-np.log10(self.water_ionization_constant(temp_c)) / 2.0)

    def carbonate_pKa(self, temp_c, dissociation: int = 1):
        return _plain(params.carbonate_pKa(temp_c, dissociation))

    def diffusion_coefficient(self, temp_c):
        params.kelvin(temp_c)
        return _plain(params.diffusion_coefficient(temp_c))


# --------------------------------------------------------------------------- transport.py:57-384
@dataclass
class GeometryParameters:
    volume: float
    height: float
    diameter: float
    n_zones: int = 5

    def validate(self) -> None:
        from_geometry = self.cross_sectional_area * self.height * 1000
        _insist(((abs(from_geometry - self.volume) / self.volume <= 0.1,
                  f"Volume inconsistency: specified {self.volume}L, calculated {from_geometry:.1f}L from geometry"),
                 (self.n_zones >= 2, f"Need at least 2 zones, got {self.n_zones}")))

    zone_height = property(lambda self: self.height / self.n_zones)
    zone_volume = property(lambda self: self.volume / self.n_zones)
    cross_sectional_area = property(lambda self: np.pi * (self.diameter / 2) ** 2)


@dataclass
class FlowParameters:
    flow_rate: float
    turbulent_intensity: float = 0.15
    recirculation_ratio: float = 5.0
    impeller_speed: float = 60.0
    impeller_diameter: float = 0.3
    power_number: float = 5.0

    def validate(self) -> None:
        _insist(((self.flow_rate >= 0, f"Flow rate cannot be negative: {self.flow_rate}"),
                 (0 <= self.turbulent_intensity <= 1, f"Turbulent intensity must be in [0,1]: {self.turbulent_intensity}"),
                 (self.recirculation_ratio >= 0, f"Recirculation ratio cannot be negative: {self.recirculation_ratio}"),
                 (self.impeller_speed >= 0, f"Impeller speed cannot be negative: {self.impeller_speed}"),
                 (self.impeller_diameter > 0, f"Impeller diameter must be positive: {self.impeller_diameter}")))


class TransportModel:
    """One reactor's column of ``params.transport_columns``: the coefficients the reference computes at construction
    (transport.py:202-254) under the reference's attribute names, and the constant exchange matrix.  Tracer
    response, dispersion number and the report are not provided (the latter two fail in the reference itself:
    transport.py:463,499 read an undefined ``self.velocity``)."""

    WATER_VISCOSITY, C_MIXING = params.WATER_VISCOSITY, params.C_MIXING
    COLUMNS = ("superficial_velocity", "impeller_tip_speed", "Re", "D_turbulent", "D_molecular", "D_effective",
               "mixing_time_seconds", "mixing_time", "Pe", "K_exchange_per_s")

    def __init__(self, geometry: GeometryParameters, flow: FlowParameters, temperature: float = 20.0):
        for record in (geometry, flow):
            record.validate()
        self.geometry, self.flow, self.temperature = geometry, flow, temperature
        self.thermo = TemperatureDependentKinetics()
        self.is_batch_mode = flow.flow_rate == 0.0
        one = {name: np.array([float(getattr(src, name))]) for src, names in (
            (geometry, ("volume", "height", "diameter")),
            (flow, ("flow_rate", "impeller_speed", "impeller_diameter", "power_number"))) for name in names}
        self.thermo.celsius_to_kelvin(temperature)          # the liquid-range check the reference makes on the way
        one["temperature"] = np.array([float(temperature)])
        column = params.transport_columns(one, geometry.n_zones)
        for name in self.COLUMNS:
            setattr(self, name, float(column[name][0]))
        self.residence_time = params.residence_time_min(geometry.volume, flow.flow_rate)
        self.K_matrix = params.exchange_matrix(self.K_exchange_per_s, float(column["Q_per_V"][0]), geometry.n_zones)

    def calculate_mixing_quality(self, concentrations):
        """(CV, segregation index) of any profile; a reactor's own state is reduced on the device
        (``IntegratedCSTR.mixing_quality`` / ``ReactorEnsemble.diagnostics``)."""
        cv, seg = params.mixing_quality(concentrations)
        return float(cv), float(seg)


# --------------------------------------------------------------------------- spatial.py:57-509
@dataclass
class StratificationParameters:
    enable_thermal_stratification: bool = True
    enable_density_stratification: bool = True
    critical_richardson: float = 0.25
    mixing_suppression_factor: float = 0.5


class SpatialModel:
    """Density profile and stratification switch of one water column on the host, for diagnostics; inside ``step()``
    the same relations run per zone per evaluation in the kernel (k_above in csrc/wt_device.hpp).  Brunt-Vaisala
    frequency, jet penetration and dead-zone listing are not provided."""

    G_GRAVITY = params.G_GRAVITY
    WATER_DENSITY_20C, THERMAL_EXPANSION_COEFF, DENSITY_ANOMALY_COEFF = 998.2, 2.1e-4, 0.008

    def __init__(self, n_zones: int, height: float, stratification_params: Optional[StratificationParameters] = None):
        _insist(((n_zones >= 2, f"Need at least 2 zones, got {n_zones}"),))
        self.n_zones, self.height, self.zone_height = n_zones, height, height / n_zones
        self.strat_params = stratification_params or StratificationParameters()
        self.thermo = TemperatureDependentKinetics()
        self.zone_centers = (np.arange(n_zones) + 0.5) * self.zone_height
        self.temperatures, self.densities = np.zeros(n_zones), np.zeros(n_zones)
        self.mixing_suppression = np.ones(len(self.zone_centers) - 1)

    def _profile(self, values, what: str) -> np.ndarray:
        values = np.asarray(values, dtype=np.float64)
        _insist(((values.shape == (self.n_zones,), f"Expected {self.n_zones} {what}, got {len(values)}"),))
        return values

    def calculate_water_density(self, temperature, salinity_g_L=0.0) -> float:
        return float(params.water_density(temperature, salinity_g_L))

    def update_density_profile(self, temperatures, salinities=None) -> np.ndarray:
        self.temperatures = self._profile(temperatures, "temperatures").copy()
        self.densities = params.water_density(self.temperatures, 0.0 if salinities is None else salinities)
        return self.densities

    def calculate_richardson_number(self, zone_idx, velocity_scale) -> float:
        _insist(((0 <= zone_idx < self.n_zones - 1, f"Invalid zone index for interface: {zone_idx}"),))
        return float(params.interface_richardson(self.densities, self.zone_height, velocity_scale)[zone_idx])

    def is_stratification_stable(self, zone_idx, velocity_scale) -> bool:
        return self.calculate_richardson_number(zone_idx, velocity_scale) > self.strat_params.critical_richardson

    def calculate_mixing_suppression(self, velocity_scale) -> np.ndarray:
        sp = self.strat_params
        if not sp.enable_thermal_stratification:
            return np.ones(self.n_zones - 1)
        self.mixing_suppression = params.suppression_factors(self.densities, self.zone_height, velocity_scale,
                                                             sp.critical_richardson, sp.mixing_suppression_factor)
        return self.mixing_suppression

    def calculate_spatial_gradients(self, parameter, parameter_name="parameter") -> Dict[str, float]:
        stats = params.profile_statistics(self._profile(parameter, "values"), self.zone_height)
        return {k: (int(v) if k == "gradient_location" else v) for k, v in stats.items()}

    def interpolate_to_depth(self, parameter, depth_from_top) -> float:
        """Piecewise linear through the zone centres, the outermost pieces extended to the walls."""
        y = self._profile(parameter, "values")
        _insist(((0 <= depth_from_top <= self.height, f"Depth {depth_from_top}m outside tank [0, {self.height}]"),))
        x, level = self.zone_centers, self.height - depth_from_top
        piece = int(np.clip(np.searchsorted(x, level) - 1, 0, self.n_zones - 2))
        slope = (y[piece + 1] - y[piece]) / (x[piece + 1] - x[piece])
        return float(y[piece] + slope * (level - x[piece]))


# --------------------------------------------------------------------------- self-checks
# The reference ships one ``validate_*`` per module and ``run_all_validations`` (core/__init__.py:266-294).  Their
# names are part of the export list; what they check here are this build's own known answers, as tables of
# (description, measured, accept): the same values tests/test_host_api.py pins against the reference's fixtures.
Check = Tuple[str, Callable[[], object], Callable[[object], bool]]


def _raises(fn: Callable[[], object]) -> bool:
    with contextlib.suppress(ValueError):
        fn()
        return False
    return True


def _run_checks(title: str, checks: Iterable[Check]) -> None:
    for what, measure, accept in checks:
        got = measure()
        assert accept(got), f"{title}: {what} (got {got!r})"
    print(f"✓ All {title} validations passed")


def validate_thermodynamics() -> None:
    th = TemperatureDependentKinetics()
    ladder = th.chlorine_decay_rate(np.array([0.0, 10.0, 20.0, 30.0, 40.0]))
    _run_checks("thermodynamic", (
        ("k(20 degC) is the reference rate", lambda: ladder[2], lambda k: abs(k - 1e-4) < th.TOLERANCE_KINETICS),
        ("Kw(25 degC) = 1e-14", lambda: th.water_ionization_constant(25.0), lambda kw: abs(kw / 1e-14 - 1) < th.TOLERANCE_EQUILIBRIUM),
        ("neutral pH at 25 degC", lambda: th.neutral_pH(25.0), lambda p: abs(p - 7.0) < th.TOLERANCE_PH),
        ("pKa1(25 degC)", lambda: th.carbonate_pKa(25.0, 1), lambda p: abs(p - th.PKA1_25C) < th.TOLERANCE_PH),
        ("decay accelerates with temperature", lambda: np.diff(ladder), lambda d: bool((d > 0).all())),
        ("Q10 between 1.5 and 2.5", lambda: ladder[3] / ladder[2], lambda q: 1.5 < q < 2.5),
        ("ice and steam are refused", lambda: [_raises(lambda t=t: th.celsius_to_kelvin(t)) for t in (-10.0, 110.0)], all)))


def validate_chemistry() -> None:
    """Needs the GPU: the equilibrium pH comes from the batched Newton-Raphson kernel."""
    from .chemistry import AqueousChemistry, BufferSystem
    chem = AqueousChemistry(BufferSystem(alkalinity=100, total_carbonate=2.0, temperature=20))
    pH = chem.calculate_pH(initial_guess=7.0)
    _run_checks("chemistry", (
        ("equilibrium pH of the default buffer", lambda: pH, lambda p: 6.0 < p < 9.0),
        ("carbonate fractions sum to one", lambda: sum(chem.alpha_carbonate(pH)), lambda t: abs(t - 1.0) < 1e-10),
        ("acid lowers, base raises the pH", lambda: (chem.add_acid(1000, 0.001, pH), chem.add_base(1000, 0.001, pH)),
         lambda ab: ab[0] < pH < ab[1]),
        ("buffering peaks near pKa1", lambda: chem.buffering_capacity(6.35) / chem.buffering_capacity(8.0), lambda r: r > 1),
        ("chlorine species add up", lambda: chem.chlorine_speciation(2.0, 7.0), lambda s: abs(s["HOCl"] + s["OCl"] - 2.0) < 1e-10)))


def validate_transport() -> None:
    geometry = GeometryParameters(volume=1000, height=2.0, diameter=2 * np.sqrt(1.0 / (np.pi * 2.0)), n_zones=5)
    flow = FlowParameters(flow_rate=5.0)
    tm = TransportModel(geometry, flow, temperature=20.0)
    rows = tm.K_matrix.sum(axis=1)
    _run_checks("transport", (
        ("exchange conserves mass in every zone but the outlet", lambda: np.abs(rows[:-1]).max(), lambda e: e < 1e-12),
        ("the outlet zone loses Q/V", lambda: rows[-1] + (flow.flow_rate / 60.0) / geometry.volume, lambda e: abs(e) < 1e-12),
        ("exchange never amplifies", lambda: np.linalg.eigvals(tm.K_matrix).real.max(), lambda e: e <= 1e-10),
        ("a uniform profile is perfectly mixed", lambda: tm.calculate_mixing_quality(np.full(5, 2.0)), lambda q: max(q) < 1e-10),
        ("impeller flow is turbulent", lambda: tm.Re, lambda re: re > 1000),
        ("mixing time of the default tank", lambda: tm.mixing_time_seconds, lambda t: 30 < t < 300)))


def validate_spatial() -> None:
    sp = SpatialModel(n_zones=5, height=2.0)
    rho = lambda t: sp.calculate_water_density(t)
    warm_top, cold_top = np.array([25.0, 23, 21, 19, 17]), np.array([17.0, 19, 21, 23, 25])

    def richardson(profile):
        sp.update_density_profile(profile)
        return sp.calculate_richardson_number(0, 0.01)

    bump = np.array([7.0, 7.1, 7.2, 7.1, 7.0])
    _run_checks("spatial", (
        ("density maximum at 4 degC", lambda: (rho(3.0), rho(4.0), rho(5.0), rho(20.0)),
         lambda r: abs(r[1] - 999.97) < 0.5 and r[0] < r[1] and r[2] > r[3]),
        ("sign of Ri follows the density gradient", lambda: (richardson(warm_top), richardson(cold_top)), lambda r: r[0] > 0 > r[1]),
        ("profile mean", lambda: sp.calculate_spatial_gradients(bump, "pH")["mean_value"], lambda m: abs(m - 7.08) < 0.01),
        ("interpolation stays inside the profile", lambda: sp.interpolate_to_depth(bump, 1.0), lambda v: 7.0 <= v <= 7.2)))


def validate_integrated_reactor() -> None:
    """Needs the GPU: thirty steps of the drop-in, first undisturbed, then under acid dosing."""
    from .reactor import BoundaryConditions, IntegratedCSTR, ReactorConfiguration
    reactor = IntegratedCSTR(ReactorConfiguration(n_zones=5, initial_pH=7.5))
    closed = dict(inlet_flow_rate=0.0, chlorine_flow_rate=0.0)

    def run(steps: int, **streams) -> float:
        for _ in range(steps):
            reactor.step(dt=1.0, boundary=BoundaryConditions(**closed, **streams))
        return float(reactor.state.pH[0])

    before = run(10)
    _run_checks("integrated reactor", (
        ("pH and chlorine stay physical", lambda: (reactor.state.pH.mean(), reactor.state.chlorine.mean()),
         lambda m: 6.0 < m[0] < 9.0 and 0.0 < m[1] < 5.0),
        ("chlorine inventory is positive", lambda: reactor.validate_conservation()["total_chlorine_mg"], lambda m: m > 0),
        ("acid dosing lowers the inlet-zone pH", lambda: run(20, acid_flow_rate=0.5, acid_concentration=0.1), lambda p: p < before)))


def run_all_validations() -> None:
    suites = (("Thermodynamics", validate_thermodynamics), ("Chemistry", validate_chemistry), ("Transport", validate_transport),
              ("Spatial", validate_spatial), ("Integrated Reactor", validate_integrated_reactor))
    bar = "=" * 70
    print(f"Running Physics Engine Validation Suite\n{bar}")
    for number, (name, suite) in enumerate(suites, 1):
        print(f"\n{number}. {name}...")
        suite()
    print(f"\n{bar}\nALL VALIDATIONS PASSED ✓\nPhysics engine verified for correctness.\n{bar}")
