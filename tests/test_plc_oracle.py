"""NEXT-2/NEXT-3 oracle (oracle/plc_oracle.py) against vectors produced by the reference's own
modbus/protocols.py and modbus/register_map.py (tests/golden/g8_modbus.json)."""
import math

import numpy as np
import pytest

from conftest import golden_json


@pytest.fixture(scope="module")
def PO():
    import plc_oracle
    return plc_oracle


def test_encoder_matches_reference(PO):
    g = golden_json("g8_modbus.json")
    for e in g["encode"]:
        x = float("nan") if e["x"] == "nan" else float.fromhex(e["x"])
        assert PO.float32_to_registers(x) == (e["hi"], e["lo"]), e
    assert PO.float32_to_registers(7.25) == (16616, 0)      # SURVEY.md known answer (struct, not the 16480 of the docstring)
    with pytest.raises(OverflowError):
        PO.float32_to_registers(1e39)


def test_decoder_matches_reference(PO):
    g = golden_json("g8_modbus.json")
    for d in g["decode"]:
        x = PO.registers_to_float32(d["hi"], d["lo"])
        if d["x"] == "nan":
            assert math.isnan(x)
        else:
            assert x == float.fromhex(d["x"]) and math.copysign(1, x) == math.copysign(1, float.fromhex(d["x"]))


def test_register_addresses_match_reference(PO):
    m = golden_json("g8_modbus.json")["map"]
    assert {k: v[0] for k, v in m["input_registers"].items()} == PO.INPUT_REGISTERS
    assert {k: v[0] for k, v in m["holding_registers"].items()} == PO.HOLDING_REGISTERS
    assert {k: v[0] for k, v in m["coils"].items()} == PO.COILS
    assert {k: v[0] for k, v in m["discrete_inputs"].items()} == PO.DISCRETE_INPUTS
    assert all(v[1] == "float32" and v[2] == 2 for k, v in m["input_registers"].items() if k != "system_status")
    assert m["input_registers"]["system_status"][1:] == ["uint16", 1]
    last = lambda t: max(v[0] + v[2] for v in m[t].values())
    assert PO.IR_SIZE == max(last("input_registers") + 10, 200) and PO.HR_SIZE == max(last("holding_registers") + 10, 200)


def test_update_inputs_semantics(PO):
    io = PO.PlantIO()
    vals = [7.25, float("nan"), 1.5, float("inf"), 10.0, 19.5, float("-inf")]
    assert io.update_inputs(vals, [0, 0, 0, 3, 0, 0, 0], 41.0)
    assert io.ir[0:2] == [16616, 0] and io.ir[2:4] == [0, 0] and io.ir[4:6] == [0, 0]          # NaN -> 0.0, pH_middle never written
    assert PO.registers_to_float32(*io.ir[6:8]) == 1.5 and io.ir[8:10] == [0, 0]
    assert PO.registers_to_float32(*io.ir[100:102]) == 41.0 and io.ir[102] == 1 and io.di[:3] == [0, 0, 1]
    # a value outside +-1e9 aborts the update where it happens: earlier registers are new, later ones stale
    assert not io.update_inputs([1.0] * 7, [0] * 7, 2e9)
    assert PO.registers_to_float32(*io.ir[0:2]) == 1.0 and PO.registers_to_float32(*io.ir[100:102]) == 41.0 and io.ir[102] == 1


def test_command_path_semantics(PO):
    io = PO.PlantIO()
    bc = [5.0, 7.0, 0.0, 20.0, 0.0, 0.1, 0.0, 1000.0, 20.0, 5.0]
    PO.apply_boundary_conditions(bc, io.read_commands())               # untouched registers: zeros -> inlet flow kept
    assert bc[0] == 5.0 and bc[4] == 0.0 and bc[6] == 0.0
    io.write_holding("acid_flow_rate", 3.5); io.write_holding("chlorine_flow_rate", float("nan")); io.write_holding("inlet_flow_rate", 0.1)
    PO.apply_boundary_conditions(bc, io.read_commands())
    assert bc[4] == 2.0 and bc[6] == 0.0 and bc[0] == float(np.float32(0.1))     # float32(0.1) > 0.1 in double
    io.write_holding("inlet_flow_rate", 1e30); io.write_holding("acid_flow_rate", -1.0); io.write_holding("chlorine_flow_rate", float("inf"))
    PO.apply_boundary_conditions(bc, io.read_commands())
    assert bc[0] == 20.0 and bc[4] == 0.0 and bc[6] == 1.0


def test_host_codec_matches_reference(wt):
    """the host-side vectorised encoder / decoder / address tables of the product package"""
    g = golden_json("g8_modbus.json")
    E = wt.ReactorEnsemble
    xs = np.array([float("nan") if e["x"] == "nan" else float.fromhex(e["x"]) for e in g["encode"]])
    with np.errstate(over="ignore"):
        w = E.encode_float32(xs)
    assert np.array_equal(w[:, 0], [e["hi"] for e in g["encode"]]) and np.array_equal(w[:, 1], [e["lo"] for e in g["encode"]])
    d = E.decode_float32(np.array([[e["hi"], e["lo"]] for e in g["decode"]]))
    ref = np.array([float("nan") if e["x"] == "nan" else float.fromhex(e["x"]) for e in g["decode"]], dtype=np.float64)
    with np.errstate(invalid="ignore"):
        assert np.array_equal(d.astype(np.float64), ref, equal_nan=True)
    m = g["map"]
    assert {k: v[0] for k, v in m["input_registers"].items()} == E.INPUT_REGISTERS
    assert all(m["holding_registers"][k][0] == a for k, a in E.HOLDING_REGISTERS.items())
    assert {k: v[0] for k, v in m["discrete_inputs"].items()} == E.DISCRETE_INPUTS
