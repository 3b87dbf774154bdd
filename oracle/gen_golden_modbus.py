#!/usr/bin/env python3
"""Golden vectors for NEXT-2/NEXT-3 (the Modbus register image and the command path).

Runs in the build container only (needs /root/reference).  The reference's
``wt_simulator.modbus`` package cannot be imported as a package (its __init__
pulls in slave.py -> pymodbus, which is not installed), but the two modules
the register image depends on are stdlib/numpy-only, so they are loaded by
file path:

    modbus/protocols.py     ModbusEncoder / ModbusDecoder   (float32 <-> 2 x uint16, big-endian)
    modbus/register_map.py  ModbusRegisterMap               (addresses, types)

Output: tests/golden/g8_modbus.json
  * "encode": float64 inputs (as hex) -> (high, low) words from ModbusEncoder.float32_to_registers
  * "decode": (high, low) -> float from ModbusDecoder.registers_to_float32 (as hex, NaN as "nan")
  * "map": name -> (address, data_type, size_words) for the four register tables
__main__.py (the caller that fills the image: update_modbus_inputs :166-224, read_modbus_commands
:227-252, apply_boundary_conditions :255-271) imports pymodbus at module level and cannot be run
here; its semantics are restated in oracle/plc_oracle.py and are "parity unpinned" beyond these
vectors.
"""
import importlib.util
import json
import math
import os
import struct
import sys

import numpy as np

REF = "/root/reference/src/wt_simulator/modbus"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    spec = importlib.util.spec_from_file_location(f"_ref_{name}", os.path.join(REF, f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    prot = load("protocols")
    rmap = load("register_map")
    enc, dec = prot.ModbusEncoder(), prot.ModbusDecoder()
    rng = np.random.default_rng(20260208)
    vals = [0.0, -0.0, 7.25, 7.0, 14.0, 1.0, -1.0, 0.1, 2.0, 20.0, 5.0, 1e9, -1e9, 1e-45, 1e-40, 3.4028234e38,
            1.17549435e-38, 123456.789, 1.0 + 2.0 ** -24, 1.0 + 3 * 2.0 ** -25, 16777217.0, float("inf"), float("-inf")]
    vals += list(rng.normal(7.0, 2.0, 40)) + list(rng.uniform(0, 1e9, 20)) + list(10.0 ** rng.uniform(-30, 30, 40) * rng.choice([-1, 1], 40))
    encode = []
    for v in vals:
        hi, lo = enc.float32_to_registers(float(v))
        encode.append({"x": float(v).hex(), "hi": int(hi), "lo": int(lo)})
    hi, lo = enc.float32_to_registers(float("nan"))
    encode.append({"x": "nan", "hi": int(hi), "lo": int(lo)})
    decode = []
    words = [(16616, 0), (16480, 0), (0, 0), (0x7FC0, 0), (0x7F80, 0), (0xFF80, 0), (0x8000, 0), (0x3DCC, 0xCCCD), (0, 1), (0x4120, 0),
             (0x41A0, 0), (0x41A0, 1), (0x3DCC, 0xCCCC), (0x3DCC, 0xCCCE), (0x7F80, 1), (0xFFFF, 0xFFFF)]
    words += [(int(a), int(b)) for a, b in rng.integers(0, 65536, (64, 2))]
    for hi, lo in words:
        x = dec.registers_to_float32(hi, lo)
        decode.append({"hi": hi, "lo": lo, "x": "nan" if math.isnan(x) else float(x).hex()})
    m = rmap.ModbusRegisterMap()
    tables = {}
    for tname in ("input_registers", "holding_registers", "coils", "discrete_inputs"):
        tables[tname] = {r.name: [int(r.address), r.data_type, int(r.size_words)] for r in getattr(m, tname)}
    out = {"source": "reference modbus/protocols.py + modbus/register_map.py loaded by path (oracle/gen_golden_modbus.py)",
           "encode": encode, "decode": decode, "map": tables}
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "g8_modbus.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0)
    print(path, len(encode), len(decode), {k: len(v) for k, v in tables.items()})


if __name__ == "__main__":
    main()
