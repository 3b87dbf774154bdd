// Issue / latency microbenchmarks for one wavefront per SIMD on gfx950 (developer tool).
//   hipcc --offload-arch=gfx950 -O3 -o tools/scratch/issue tools/ubench/issue.hip && tools/scratch/issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int MODE> __global__ __launch_bounds__(64) void k(long long *out, double seed, int iters)
{
    double a = seed + threadIdx.x, b = seed * 0.5, c = seed * 0.25, d = seed * 0.125;
    double e = a + 1, f = b + 1, g = c + 1, h = d + 1;
    const double m = 1.0000001, ad = 1e-9;
    int lo = threadIdx.x, hi = threadIdx.x * 3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(ad));) }
        if constexpr (MODE == 1) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(ad));) }
        if constexpr (MODE == 2) { REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(m), "v"(ad));) }
        if constexpr (MODE == 3) { REP16(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(lo) : "v"(hi));) }
        if constexpr (MODE == 4) { REP16(asm volatile("s_mov_b32 s20, 0x12345\n s_mov_b32 s21, 0x12346\n s_mov_b32 s22, 0x12347\n s_mov_b32 s23, 0x12348" ::: "s20", "s21", "s22", "s23");) }
        if constexpr (MODE == 5) { REP16(asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_read_b32 %1, a1" : "+v"(lo), "+v"(hi) :: "a0", "a1");) }
        if constexpr (MODE == 6) { REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));) }
        if constexpr (MODE == 7) { REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if constexpr (MODE == 8) { REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n s_mov_b32 s20, 0x12345\n v_fma_f64 %1, %1, %2, %3\n s_mov_b32 s21, 0x12345" : "+v"(a), "+v"(b) : "v"(m), "v"(ad) : "s20", "s21");) }
        if constexpr (MODE == 9) { REP16(asm volatile("v_fma_f64 %0, %0, %3, %4\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f64 %0, %0, %3, %4\n v_mov_b32_dpp %2, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(lo), "+v"(hi) : "v"(m), "v"(ad));) }
        if constexpr (MODE == 10) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(lo), "+v"(hi) :: "vcc");) }
        if constexpr (MODE == 11) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_mov_b32 %2, %3\n v_fma_f64 %1, %1, %4, %5\n v_mov_b32 %3, %2" : "+v"(a), "+v"(b), "+v"(lo), "+v"(hi) : "v"(m), "v"(ad));) }
        if constexpr (MODE == 12) { REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n v_add_f64 %0, %0, %1" : "+v"(a) : "v"(m) : "vcc", "s20", "s21");) }
        if constexpr (MODE == 13) { REP16(asm volatile("v_mul_f64 %0, %0, %2\n v_add_f64 %1, %1, %3\n v_mul_f64 %0, %0, %2\n v_add_f64 %1, %1, %3" : "+v"(a), "+v"(b) : "v"(m), "v"(ad));) }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a + b + c + d + e + f + g + h + lo + hi == 12345.678) out[0] = 0;
}

template <int MODE> void run(const char *name, int per_iter, int blocks)
{
    long long *d; hipMalloc(&d, sizeof(long long) * blocks);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 1.0, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 1.0, iters);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), d, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    // s_memtime counts at 100 MHz on gfx94x/950 (constant clock); convert with the shader clock measured separately
    printf("%-52s blocks %5d: %.3f ticks(100MHz)/instr -> %.2f cycles @2.4GHz\n", name, blocks, s / blocks / iters / per_iter, s / blocks / iters / per_iter * 24.0);
    hipFree(d);
}

int main()
{
    for (int blocks : {1024, 2048}) {
        run<0>("fma_f64 dependent chain", 64, blocks);
        run<1>("fma_f64 4 independent chains", 64, blocks);
        run<2>("fma_f64 2 independent chains", 64, blocks);
        run<13>("mul_f64/add_f64 2 chains", 64, blocks);
        run<3>("v_mov_b32_dpp (same dst)", 64, blocks);
        run<4>("s_mov_b32", 64, blocks);
        run<5>("v_accvgpr write/read dependent", 64, blocks);
        run<6>("v_rcp_f64 dependent", 64, blocks);
        run<7>("v_rcp_f64 4 independent", 64, blocks);
        run<8>("fma_f64 + s_mov interleaved (per instr)", 64, blocks);
        run<9>("fma_f64 chain + dpp interleaved (per instr)", 64, blocks);
        run<10>("v_cndmask dependent", 64, blocks);
        run<11>("fma_f64 + v_mov_b32 interleaved (per instr)", 64, blocks);
        run<12>("cmp + saveexec + or exec + add (per group)", 16, blocks);
    }
    return 0;
}
