// wt_plc.hpp -- gfx950 device code of the plant I/O layer around the physics step (SURVEY.md
// section 8(f) NEXT-2 driver loop, NEXT-3 Modbus register image), one virtual PLC slave per reactor.
//
//   pack_inputs            update_modbus_inputs  __main__.py:166-224  (NaN/inf -> 0.0, system_status, fault bits)
//                          ModbusEncoder.float32_to_registers  modbus/protocols.py:35-58 (high word, low word)
//                          addresses  modbus/register_map.py:119-244, 364-401
//   apply_commands         read_modbus_commands  __main__.py:227-252, validate_flow_rate :57-63,
//                          apply_boundary_conditions :255-271, ModbusDecoder.registers_to_float32 protocols.py:155-177
//
// Both are byte movers over a few dozen bytes per reactor; they run inside the physics kernel
// (wt_device.hpp), right behind the fused sensor suite of an outer step that is a PLC scan: one lane
// per reactor of the wavefront takes the seven readings from LDS, publishes the image and decodes the
// holding registers; the new setpoints go back through LDS to the lanes that integrate the reactor.
// Image layout (array of structures: a Modbus server answers "registers a..b of unit r" from one
// contiguous 40-byte record):
//   input image  ir[r][20] uint16: words 0..15 = input registers 0..15, 16..17 = simulation_time
//                (registers 100..101), 18 = system_status (register 102), 19 = discrete inputs 0..2 in bits 0..2
//   holding image hr[r][6] uint16: holding registers 0..5 (acid, chlorine, inlet flow-rate setpoints)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wtp {

constexpr int IR_WORDS = 20, HR_WORDS = 6, NSENS = 7;
// suite order -> first input register: pH_inlet 0, pH_outlet 4, chlorine_inlet 6, chlorine_outlet 8, flow_rate 10,
// temperature_inlet 12, temperature_outlet 14 (register 2, pH_middle, is never written by the reference loop)
__device__ constexpr int SENSOR_REG[NSENS] = {0, 4, 6, 8, 10, 12, 14};

struct PackArgs {
    double *loop_time;       // [N] the loop's sim_time accumulator (starts at 0, += dt per step)  __main__.py:388,446
    uint16_t *ir;            // [N][IR_WORDS]
    uint8_t *update_ok;      // [N] 0 after an update that raised (value outside +-1e9: slave.py:146-147)
};

__device__ __forceinline__ uint32_t f32_bits_from_double(double x) { return __float_as_uint((float)x); }   // struct.pack('>f'): RN-even

// value / fault: the seven readings of this reactor, element i at [i * stride] (LDS, written by the sensor lanes);
// sim_time: the loop publishes it BEFORE incrementing it (__main__.py:413 vs :446), so the image of step k carries (k-1)*dt
template <class A> __device__ __forceinline__ void pack_inputs(const A &a, int64_t r, const float *value, const int *fault, int stride, double sim_time)
{
    uint16_t *ir = a.ir + r * IR_WORDS;
    bool any_fault = false; uint32_t fbits = 0;
#pragma unroll
    for (int i = 0; i < NSENS; ++i) {
        float v = value[i * stride];
        if (!(fabsf(v) <= 3.402823466e38f)) v = 0.0f;                           // safe_value: NaN, +-inf -> 0.0
        const uint32_t b = __float_as_uint(v);                                  // |v| <= range of the sensor << 1e9
        ir[SENSOR_REG[i]] = (uint16_t)(b >> 16); ir[SENSOR_REG[i] + 1] = (uint16_t)(b & 0xffffu);
        const bool f = fault[i * stride] != 0;
        any_fault |= f;
        if (f) fbits |= (i == 0) ? 1u : (i == 1) ? 2u : (i == 2 || i == 3) ? 4u : 0u;
    }
    if (!(sim_time >= -1e9 && sim_time <= 1e9)) { a.update_ok[r] = 0; return; } // ValueError: the rest of the image stays stale
    const uint32_t tb = f32_bits_from_double(sim_time);
    ir[16] = (uint16_t)(tb >> 16); ir[17] = (uint16_t)(tb & 0xffffu);
    ir[18] = any_fault ? 1 : 0;
    ir[19] = (uint16_t)fbits;
    a.update_ok[r] = 1;
}

struct CommandArgs {
    int64_t N;
    const uint16_t *hr;      // [N][HR_WORDS]
    double *bc;              // [NB][N] boundary block of the ensemble (rows: 0 inlet flow, 4 acid flow, 6 chlorine flow)
};

__device__ __forceinline__ double validate_flow_rate(float v, double max_value)
{   // __main__.py:57-63 on the float the decoder returns
    if (v != v) return 0.0;
    return fmax(0.0, fmin((double)v, max_value));
}

// read_modbus_commands validates, apply_boundary_conditions validates again (idempotent).  Writes the boundary
// block and returns the three rows in force afterwards (cmd[0] inlet, cmd[1] acid, cmd[2] chlorine flow).
template <class A> __device__ __forceinline__ void apply_commands(const A &a, int64_t r, double cmd[3])
{
    const uint16_t *hr = a.hr + r * HR_WORDS;
    const float acid = __uint_as_float(((uint32_t)hr[0] << 16) | hr[1]);
    const float chlorine = __uint_as_float(((uint32_t)hr[2] << 16) | hr[3]);
    const float inlet = __uint_as_float(((uint32_t)hr[4] << 16) | hr[5]);
    cmd[1] = validate_flow_rate(acid, 2.0);
    cmd[2] = validate_flow_rate(chlorine, 1.0);
    const double inlet_v = validate_flow_rate(inlet, 20.0);
    cmd[0] = (inlet_v > 0.1) ? inlet_v : a.bc[0 * a.N + r];                     // "only update inlet flow if command is significant"
    a.bc[4 * a.N + r] = cmd[1];
    a.bc[6 * a.N + r] = cmd[2];
    if (inlet_v > 0.1) a.bc[0 * a.N + r] = inlet_v;
}

} // namespace wtp
