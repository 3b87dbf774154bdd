// LDS throughput and round-trip latency of one wavefront per SIMD on gfx950 (developer tool).
//   hipcc --offload-arch=gfx950 -O3 -o tools/scratch/lds tools/ubench/lds.hip && tools/scratch/lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

// MODE: 0 ds_bpermute_b32 x16 back to back, one wait | 1 the same, a wait after each | 2 ds_read_b64 x16, one wait
//       3 ds_read_b128 x16, one wait | 4 ds_write_b64 + 2 ds_read_b64 (neighbour exchange through memory) x8, one wait
//       5 ds_write_b128 + 2 ds_read_b128 x8 | 6 v_mov_b32_dpp wave_shr:1 x16 (independent) | 7 ds_read2st64_b64 x16
//       8 bpermute x16 interleaved with 16 independent v_fma_f64
template <int MODE> __global__ __launch_bounds__(64) void k(long long *out, int iters)
{
    __shared__ double buf[64 * 2 * 40];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 2 * 40; i += 64) buf[i] = i;
    __syncthreads();
    const int a8 = lane * 8, a16 = lane * 16, lo8 = ((lane + 63) & 63) * 8, hi8 = ((lane + 1) & 63) * 8, lo16 = ((lane + 63) & 63) * 16, hi16 = ((lane + 1) & 63) * 16;
    const int bp = ((lane + 62) & 63) * 4;
    int x = lane, y = 0; double d = lane, e = 0, m = 1.0000001;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 q = {1.0, 2.0}, r = {0, 0}, r2 = {0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) { asm volatile(REP16("ds_bpermute_b32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)" : "=&v"(y) : "v"(bp), "v"(x) : "memory"); }
        if constexpr (MODE == 1) { asm volatile(REP16("ds_bpermute_b32 %0, %1, %2\n s_waitcnt lgkmcnt(0)\n") : "=&v"(y) : "v"(bp), "v"(x) : "memory"); }
        if constexpr (MODE == 2) { asm volatile(REP16("ds_read_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=&v"(e) : "v"(a8) : "memory"); }
        if constexpr (MODE == 3) { asm volatile(REP16("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a16) : "memory"); }
        if constexpr (MODE == 4) { asm volatile(REP4(REP4("ds_write_b64 %2, %4\n ds_read_b64 %0, %3\n ds_read_b64 %1, %5\n")) "s_waitcnt lgkmcnt(0)" : "=&v"(e), "=&v"(d) : "v"(a8), "v"(lo8), "v"(m), "v"(hi8) : "memory"); }
        if constexpr (MODE == 5) { asm volatile(REP4(REP4("ds_write_b128 %2, %4\n ds_read_b128 %0, %3\n ds_read_b128 %1, %5\n")) "s_waitcnt lgkmcnt(0)" : "=&v"(r), "=&v"(r2) : "v"(a16), "v"(lo16), "v"(q), "v"(hi16) : "memory"); }
        if constexpr (MODE == 6) { asm volatile(REP16("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "=&v"(y) : "v"(x)); }
        if constexpr (MODE == 7) { asm volatile(REP16("ds_read2st64_b64 %0, %1 offset0:1 offset1:2\n") "s_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a8) : "memory"); }
        if constexpr (MODE == 9) { asm volatile(REP16("v_fma_f64 %0, %1, %1, %1\n") : "=&v"(e) : "v"(m)); }
        if constexpr (MODE == 10) { asm volatile(REP16("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "=&v"(y) : "v"(x)); }
        if constexpr (MODE == 11) { asm volatile(REP16("ds_read2_b64 %0, %1 offset0:0 offset1:1\n") "s_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a16) : "memory"); }
        if constexpr (MODE == 12) { asm volatile(REP16("ds_write2st64_b64 %0, %1, %2 offset0:1 offset1:2\n") "s_waitcnt lgkmcnt(0)" :: "v"(a8), "v"(m), "v"(d) : "memory"); }
        if constexpr (MODE == 13) { asm volatile(REP16("ds_write_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a16), "v"(q) : "memory"); }
        if constexpr (MODE == 14) { asm volatile(REP16("ds_write_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a8), "v"(m) : "memory"); }
        if constexpr (MODE == 8) { asm volatile(REP16("ds_bpermute_b32 %0, %2, %3\n v_fma_f64 %1, %4, %4, %4\n") "s_waitcnt lgkmcnt(0)" : "=&v"(y), "=&v"(e) : "v"(bp), "v"(x), "v"(m) : "memory"); }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    if (y + e + d + r.x + r2.y == 12345.678) out[0] = 0;
}

template <int MODE> void run(const char *name, int blocks)
{
    long long *d; (void)hipMalloc(&d, sizeof(long long) * blocks);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters);
    std::vector<long long> h(blocks); (void)hipMemcpy(h.data(), d, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-64s %.2f ticks per group of 16\n", name, s / blocks / iters);
    (void)hipFree(d);
}

int main()
{
    const int blocks = 1024;
    run<9>("16 v_fma_f64 independent (calibration: 5.3 cycles each)", blocks);
    run<10>("16 v_mov_b32_dpp row_shr:1", blocks);
    run<0>("16 ds_bpermute_b32, one wait", blocks);
    run<1>("16 x (ds_bpermute_b32 + wait)", blocks);
    run<2>("16 ds_read_b64, one wait", blocks);
    run<3>("16 ds_read_b128, one wait", blocks);
    run<7>("16 ds_read2st64_b64, one wait", blocks);
    run<11>("16 ds_read2_b64 (adjacent doubles), one wait", blocks);
    run<12>("16 ds_write2st64_b64, one wait", blocks);
    run<13>("16 ds_write_b128, one wait", blocks);
    run<14>("16 ds_write_b64, one wait", blocks);
    run<4>("16 x (ds_write_b64 + 2 ds_read_b64 of the neighbours), one wait", blocks);
    run<5>("16 x (ds_write_b128 + 2 ds_read_b128 of the neighbours), one wait", blocks);
    run<6>("16 v_mov_b32_dpp wave_shr:1", blocks);
    run<8>("16 x (ds_bpermute_b32 + v_fma_f64), one wait", blocks);
    return 0;
}
