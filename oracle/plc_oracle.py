"""CPU oracle of the Modbus register image and the command path (TEST INFRASTRUCTURE ONLY).

Restates, for one reactor, what the reference's driver loop does around the physics step
(SURVEY.md section 8(f) NEXT-2 / NEXT-3):

    modbus/protocols.py:35-58, 155-177      float32 <-> (high word, low word), big-endian IEEE-754
    modbus/register_map.py:119-401          where each value lives (pinned by tests/golden/g8_modbus.json)
    modbus/slave.py:113-137                 block sizes: max(last address + 10, 200) words / max(.., 100) bits
    modbus/slave.py:139-180                 update_input_register (|value| <= 1e9 or ValueError), update_discrete_input
    __main__.py:166-224                     update_modbus_inputs: NaN/inf -> 0.0, system_status, fault bits
    __main__.py:57-63, 227-271              validate_flow_rate, read_modbus_commands, apply_boundary_conditions

protocols.py / register_map.py are pinned by golden vectors generated from the reference modules
(oracle/gen_golden_modbus.py).  __main__.py and slave.py import pymodbus (absent here), so the
semantics of those call sites are "parity unpinned": restated from the source text only.
"""
from __future__ import annotations

import math

import numpy as np

INPUT_REGISTERS = {"pH_inlet": 0, "pH_middle": 2, "pH_outlet": 4, "chlorine_inlet": 6, "chlorine_outlet": 8, "flow_rate": 10,
                   "temperature_inlet": 12, "temperature_outlet": 14, "simulation_time": 100, "system_status": 102}
HOLDING_REGISTERS = {"acid_flow_rate": 0, "chlorine_flow_rate": 2, "inlet_flow_rate": 4, "acid_concentration": 10,
                     "chlorine_concentration": 12, "simulation_timestep": 100}
COILS = {"acid_pump_enable": 0, "chlorine_pump_enable": 1, "simulation_running": 2}
DISCRETE_INPUTS = {"sensor_fault_pH_inlet": 0, "sensor_fault_pH_outlet": 1, "sensor_fault_chlorine": 2}
IR_SIZE, HR_SIZE, CO_SIZE, DI_SIZE = max(103 + 10, 200), max(102 + 10, 200), max(3 + 10, 100), max(3 + 10, 100)

# suite order (sensor_oracle.SENSOR_NAMES) -> input register that receives it (__main__.py:199-209)
SENSOR_TO_REGISTER = ("pH_inlet", "pH_outlet", "chlorine_inlet", "chlorine_outlet", "flow_rate", "temperature_inlet", "temperature_outlet")


def float32_to_registers(value: float):
    """(high, low) 16-bit words of the IEEE-754 single nearest to ``value`` (protocols.py:35-58)."""
    if not (math.isnan(value) or math.isinf(value)) and abs(value) >= 2.0 ** 128 * (1 - 2.0 ** -25):
        raise OverflowError("float too large to pack with f format")          # what struct.pack('>f') does
    with np.errstate(over="ignore"):
        bits = int(np.array([value], dtype=np.float64).astype(np.float32).view(np.uint32)[0])
    return bits >> 16, bits & 0xFFFF


def registers_to_float32(high: int, low: int) -> float:
    """protocols.py:155-177."""
    return float(np.array([(high << 16) | low], dtype=np.uint32).view(np.float32)[0])


def validate_flow_rate(value, max_value: float = 20.0) -> float:
    """__main__.py:57-63."""
    if not isinstance(value, (int, float)):
        return 0.0
    if value != value:
        return 0.0
    return max(0.0, min(float(value), max_value))


class PlantIO:
    """The four data blocks of one virtual PLC slave as plain lists (slave.py:113-137)."""

    def __init__(self):
        self.ir, self.hr = [0] * IR_SIZE, [0] * HR_SIZE
        self.co, self.di = [0] * CO_SIZE, [0] * DI_SIZE

    def _update_input_register(self, name: str, value: float):      # slave.py:139-164
        if not (-1e9 <= value <= 1e9):
            raise ValueError("Value out of range")
        a = INPUT_REGISTERS[name]
        if name == "system_status":
            self.ir[a] = int(value)
        else:
            self.ir[a], self.ir[a + 1] = float32_to_registers(value)

    def update_inputs(self, values, faults, sim_time: float) -> bool:
        """update_modbus_inputs (__main__.py:166-224).  values / faults: the 7 readings in suite order."""
        def safe(v):
            return 0.0 if (v != v or v == float("inf") or v == float("-inf")) else v
        try:
            for i, reg in enumerate(SENSOR_TO_REGISTER):
                self._update_input_register(reg, safe(values[i]))
            self._update_input_register("simulation_time", sim_time)
            self._update_input_register("system_status", 1 if any(f != 0 for f in faults) else 0)
            self.di[0] = 1 if faults[0] != 0 else 0
            self.di[1] = 1 if faults[1] != 0 else 0
            self.di[2] = 1 if (faults[2] != 0 or faults[3] != 0) else 0
            return True
        except Exception:
            return False

    def write_holding(self, name: str, value: float):                # slave.py:221-245 (what a master's write leaves)
        a = HOLDING_REGISTERS[name]
        self.hr[a], self.hr[a + 1] = float32_to_registers(value)

    def read_commands(self):
        """read_modbus_commands (__main__.py:227-252) -> (acid, chlorine, inlet)."""
        rd = lambda n: registers_to_float32(self.hr[HOLDING_REGISTERS[n]], self.hr[HOLDING_REGISTERS[n] + 1])
        return (validate_flow_rate(rd("acid_flow_rate"), 2.0), validate_flow_rate(rd("chlorine_flow_rate"), 1.0),
                validate_flow_rate(rd("inlet_flow_rate"), 20.0))


def apply_boundary_conditions(bc, commands):
    """__main__.py:255-271 on a boundary vector in BoundaryConditions field order
    (0 inlet_flow_rate, 4 acid_flow_rate, 6 chlorine_flow_rate)."""
    acid, chlorine, inlet = commands
    bc[4] = validate_flow_rate(acid, 2.0)
    bc[6] = validate_flow_rate(chlorine, 1.0)
    if inlet > 0.1:
        bc[0] = validate_flow_rate(inlet, 20.0)
