// wt_place.hpp -- which reactors share a wavefront.
//
// A wavefront's outer step takes as many solver trips as its slowest reactor needs, and a reactor's need is a
// property of its regime that persists (correlation of the per-reactor RHS-evaluation count between consecutive
// 30-step windows on the bench ensemble: 0.92; tools/balance_probe.py).  Mixed at random, most wavefronts hold at
// least one expensive reactor and everybody in them waits for it.  So the slots of the wavefront-groups are dealt
// in order of cost: a stable counting sort of the reactors by their mean RHS evaluations per outer step (1/8
// resolution, 256 bins) since the last re-binning, three small kernels on the handle's stream ahead of a launch.
// Reactors never interact, so every reactor's results are the same bits wherever it sits (tests assert it).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wtpl {

constexpr int BINS = 256, CHUNK = 1024;   // reactors per workgroup of the sort

struct PlaceArgs {
    int64_t N;
    int32_t *cost;      // [N] RHS evaluations accumulated over `steps` outer steps (zeroed by the scatter)
    int steps;
    int32_t *hist;      // [blocks][BINS]
    int32_t *perm;      // [N] out
};

__device__ __forceinline__ int bin_of(int cost, int steps)
{
    const int b = (int)(((int64_t)cost * 8) / steps);       // mean evaluations per outer step, in eighths
    return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
}

__global__ __launch_bounds__(256) void iota_kernel(int32_t *perm, int64_t N)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) perm[i] = (int32_t)i;
}

// per-chunk histogram: thread t counts bin t over the chunk's keys (kept in LDS)
__global__ __launch_bounds__(BINS) void place_count_kernel(const PlaceArgs a)
{
    __shared__ unsigned char key[CHUNK];
    const int64_t base = (int64_t)blockIdx.x * CHUNK;
    const int cnt = (int)((a.N - base < CHUNK) ? a.N - base : CHUNK);
    for (int i = threadIdx.x; i < cnt; i += BINS) key[i] = (unsigned char)bin_of(a.cost[base + i], a.steps);
    __syncthreads();
    int c = 0;
    for (int i = 0; i < cnt; ++i) c += (key[i] == threadIdx.x) ? 1 : 0;
    a.hist[(int64_t)blockIdx.x * BINS + threadIdx.x] = c;
}

// exclusive offsets: bin-major, chunk-minor (one workgroup; hist becomes the offset table)
__global__ __launch_bounds__(BINS) void place_scan_kernel(const PlaceArgs a, int blocks)
{
    __shared__ int total[BINS];
    const int t = threadIdx.x;
    int sum = 0;
    for (int b = 0; b < blocks; ++b) sum += a.hist[(int64_t)b * BINS + t];
    total[t] = sum;
    __syncthreads();
    int start = 0;
    for (int k = 0; k < t; ++k) start += total[k];
    for (int b = 0; b < blocks; ++b) {
        const int c = a.hist[(int64_t)b * BINS + t];
        a.hist[(int64_t)b * BINS + t] = start;
        start += c;
    }
}

// stable scatter: thread t walks the chunk in order and places the reactors of bin t; the cost history restarts
__global__ __launch_bounds__(BINS) void place_scatter_kernel(const PlaceArgs a)
{
    __shared__ unsigned char key[CHUNK];
    const int64_t base = (int64_t)blockIdx.x * CHUNK;
    const int cnt = (int)((a.N - base < CHUNK) ? a.N - base : CHUNK);
    for (int i = threadIdx.x; i < cnt; i += BINS) {
        key[i] = (unsigned char)bin_of(a.cost[base + i], a.steps);
        a.cost[base + i] = 0;
    }
    __syncthreads();
    int o = a.hist[(int64_t)blockIdx.x * BINS + threadIdx.x];
    for (int i = 0; i < cnt; ++i)
        if (key[i] == threadIdx.x) a.perm[o++] = (int32_t)(base + i);
}

} // namespace wtpl
