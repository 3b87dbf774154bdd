// wtphys.hip -- host side of libwtphys.so: the C ABI declared in include/wtphys.h.
// Owns device memory behind an opaque handle, uploads the SoA constant / boundary
// blocks, launches the gfx950 kernels of wt_device.hpp on the handle's stream.
#include "wt_device.hpp"
#include "wt_diag.hpp"
#include "wt_place.hpp"
#include "../../include/wtphys.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define WT_SNAPSHOT_PACK_MAX (256u * 1024u)   /* ensembles up to this image size download as one packed copy */

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(WT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

int levels_for(int n)
{
    int l = 0;
    while ((1 << l) < n) ++l;
    return l < 1 ? 1 : l;
}

} // namespace

struct wt_ensemble {
    int64_t N = 0;
    int n = 0, R = 0, device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    double *par = nullptr, *bc = nullptr;
    double *pH = nullptr, *Cl = nullptr, *T = nullptr, *time = nullptr, *flow = nullptr;
    double *dH = nullptr, *dRho = nullptr, *dK = nullptr;
    uint32_t *status = nullptr;
    int32_t *stats = nullptr;
    int64_t *wave_diag = nullptr; // optional per-wavefront diagnostics (wt_ensemble_enable_wave_diag)
    double *bad_T = nullptr;      // [N] temperature named by the reference's ValueError
    // placement of reactors into wavefront-groups (wt_place.hpp): slot -> reactor, cost history, sort scratch
    int32_t *perm = nullptr, *cost = nullptr, *place_hist = nullptr;
    int placement = WT_PLACE_ADAPTIVE;
    int64_t cost_steps = 0;       // outer steps the cost history covers
    // device-side work queue of the default schedule (wt_device.hpp): control words, FIFO slots, next step per group
    int32_t *q_ctrl = nullptr; unsigned long long *q_slots = nullptr; int32_t *q_next = nullptr;
    int q_cap = 0, q_workers = 0;
    int64_t n_groups = 0;
    int sched_mode = WT_SCHED_QUEUE;
    int64_t *trace = nullptr; int trace_cap = 0;   // developer item trace (wt_ensemble_item_trace)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_bc = false, have_state = false;
    // Schedule.  WT_SCHED_QUEUE (default): one launch per wt_ensemble_step call, worker wavefronts take
    // (wavefront-group, next few outer steps) items from a device-side FIFO.  WT_SCHED_STREAMS (round-1
    // schedule, kept for comparison): n_sub contiguous reactor ranges on their own HIP streams, launches of
    // at most chunk_steps outer steps.  chunk_steps is also the PLC scan interval.
    // developer knobs (tools/), read from the environment once at creation: WT_Q_ITEM (outer steps per work item),
    // WT_PLACE_MIN (history before a re-deal), WT_Q_TICKETS (forces the launch split), WT_FULL_WAVES,
    // WT_DENSE_COUPLING (every solve takes the general path: the test that the fast paths give the same bits)
    int knob_item = 0; int64_t knob_place_min = 0, knob_tickets = 0; int knob_dense = 0;
    // sticky record of a launch that did not advance every group (device word + pinned host mirror)
    int32_t *q_sticky = nullptr;   // device view of err_host (host-coherent pinned memory: the check kernel writes it in place)
    // one contiguous snapshot of a small ensemble: packed on the device, one copy into pinned memory
    void *snap_dev = nullptr, *snap_host = nullptr; size_t snap_bytes = 0;
    int32_t *err_host = nullptr;   // two words, each only ever set to 1: [0] a hand-off timed out, [1] a group was left behind
    int64_t redeals = 0;          // times the slots were re-dealt (bench.py reports it)
    int n_sub = 1, chunk_steps = WT_DEFAULT_CHUNK;
    int step_limit = 2000;    // attempts per outer step before a reactor is given up (reference: unlimited)
    hipStream_t sub_stream[WT_MAX_STREAMS] = {};
    hipEvent_t sub_done[WT_MAX_STREAMS] = {};
    hipEvent_t ev_fork = nullptr;
    // optional fused sensor suite (wt_sensors.hpp)
    bool sensors_on = false;
    uint64_t sens_seed = 0; int64_t sens_reactor_base = 0;
    float *s_fs = nullptr, *s_full_scale = nullptr, *s_ring_t = nullptr, *s_ring_v = nullptr, *s_out_value = nullptr, *s_hist_value = nullptr;
    double *s_ds = nullptr, *s_t_enable = nullptr;
    int32_t *s_is = nullptr, *s_ring_head = nullptr, *s_ring_cnt = nullptr, *s_hist_pos = nullptr;
    uint8_t *s_out_status = nullptr, *s_out_fault = nullptr, *s_hist_status = nullptr, *s_hist_fault = nullptr;
    int s_hist_cap = 0;
    // optional plant I/O: Modbus register images per reactor (wt_plc.hpp); one PLC scan every chunk_steps outer steps
    bool plc_on = false;
    double *diag_out = nullptr;
    uint16_t *p_ir = nullptr, *p_hr = nullptr; double *p_loop_time = nullptr; uint8_t *p_update_ok = nullptr;
    // optional per-launch HIP-event timing (bench.py roofline accounting)
    bool time_launches = false;
    std::vector<hipEvent_t> lt_pool;   // start/stop pairs
    size_t lt_used = 0;
};

namespace {

wt::StepArgs make_args(const wt_ensemble *h, double dt, int n_steps, int first_step, int call_steps, int scan_every)
{
    wt::StepArgs a;
    a.N = h->N; a.r0 = 0; a.r1 = h->N; a.n = h->n; a.R = h->R;
    a.par = h->par; a.bc = h->bc;
    a.pH = h->pH; a.Cl = h->Cl; a.T = h->T; a.time = h->time; a.flow = h->flow;
    a.dH = h->dH; a.dRho = h->dRho; a.dK = h->dK;
    a.status = h->status; a.stats = h->stats; a.wave_diag = h->wave_diag; a.bad_T = h->bad_T;
    a.perm = h->perm; a.cost = (h->placement == WT_PLACE_ADAPTIVE) ? h->cost : nullptr;
    a.dt = dt; a.n_steps = n_steps; a.first_step = first_step; a.call_steps = call_steps; a.step_limit = h->step_limit;
    a.q_ctrl = nullptr; a.q_slots = nullptr; a.q_next = nullptr; a.q_cap = 0; a.item_steps = n_steps; a.n_groups = (int)h->n_groups;
    a.trace = h->trace; a.trace_cap = h->trace_cap;
    a.kt = wt::default_ktab(); a.rt = wt::default_rtab();
    a.kt.dense_bias = h->knob_dense ? 1.0 : 0.0;
    wts::SuiteArgs &s = a.sens;
    memset(&s, 0, sizeof s);
    s.on = h->sensors_on ? 1 : 0; s.plc_on = h->plc_on ? 1 : 0; s.scan_every = scan_every > 0 ? scan_every : 1;
    s.N = h->N; s.reactor_base = h->sens_reactor_base;
    s.seed_lo = (uint32_t)(h->sens_seed & 0xffffffffu); s.seed_hi = (uint32_t)(h->sens_seed >> 32);
    s.t_enable = h->s_t_enable; s.fs = h->s_fs; s.ds = h->s_ds; s.is = h->s_is; s.full_scale = h->s_full_scale;
    s.ring_t = h->s_ring_t; s.ring_v = h->s_ring_v; s.ring_push = h->s_ring_head; s.ring_cursor = h->s_ring_cnt;
    s.out_value = h->s_out_value; s.out_status = h->s_out_status; s.out_fault = h->s_out_fault;
    s.hist_value = h->s_hist_value; s.hist_status = h->s_hist_status; s.hist_fault = h->s_hist_fault;
    s.hist_cap = h->s_hist_cap; s.hist_pos = h->s_hist_pos;
    s.pack.loop_time = h->p_loop_time; s.pack.ir = h->p_ir; s.pack.update_ok = h->p_update_ok;
    s.cmd.N = h->N; s.cmd.hr = h->p_hr; s.cmd.bc = h->bc;
    return a;
}

template <class T> void free_and_null(T *&p) { if (p) (void)hipFree(p); p = nullptr; }

void release_sensor_buffers(wt_ensemble *h)
{
    free_and_null(h->s_fs); free_and_null(h->s_full_scale); free_and_null(h->s_ring_t); free_and_null(h->s_ring_v);
    free_and_null(h->s_out_value); free_and_null(h->s_hist_value); free_and_null(h->s_ds); free_and_null(h->s_t_enable);
    free_and_null(h->s_is); free_and_null(h->s_ring_head); free_and_null(h->s_ring_cnt); free_and_null(h->s_hist_pos);
    free_and_null(h->s_out_status); free_and_null(h->s_out_fault);
    free_and_null(h->s_hist_status); free_and_null(h->s_hist_fault);
    h->s_hist_cap = 0; h->sensors_on = false;
}

void release_plc_buffers(wt_ensemble *h)
{
    free_and_null(h->p_ir); free_and_null(h->p_hr); free_and_null(h->p_loop_time); free_and_null(h->p_update_ok);
    h->plc_on = false;
}

bool row_mode(int n) { return n == 2 || n == 4 || n == 8 || n == 16; }

// default stream schedule: up to 4 ranges, but keep at least 64 wavefronts per range
int default_streams(int64_t n_reactors, int R)
{
    const int64_t waves = (n_reactors + R - 1) / R;
    const int ns = (int)(waves / 64);
    return ns < 1 ? 1 : (ns > 4 ? 4 : ns);
}

// the kernel instantiation for this zone count
template <class F> void with_step_kernel(const wt_ensemble *h, F &&f)
{
    const int n = h->n;
#if defined(WT_ONLY_LV3)   // scratch builds for kernel tuning: n = 8 only (the row-shift variant)
    (void)n;
    f(wt::step_kernel<3, true>, 64);
#elif defined(WT_ONLY_LV5)   // ... n in 17..32 only
    (void)n;
    f(wt::step_kernel<5, false>, 64);
#else
    const int lv = levels_for(n);
    if (row_mode(n)) { // n in {2,4,8,16}: every cross-lane move is a DPP row operation
        switch (lv) {
        case 1: f(wt::step_kernel<1, true>, 64); break;
        case 2: f(wt::step_kernel<2, true>, 64); break;
        case 3: f(wt::step_kernel<3, true>, 64); break;
        default: f(wt::step_kernel<4, true>, 64); break;
        }
    } else {
        switch (lv) {
        case 2: f(wt::step_kernel<2, false>, 64); break;
        case 3: f(wt::step_kernel<3, false>, 64); break;
        case 4: f(wt::step_kernel<4, false>, 64); break;
        case 5: f(wt::step_kernel<5, false>, 64); break;
        default: f(wt::step_kernel<6, false>, 64); break;
        }
    }
#endif
}

void launch_step_raw(const wt_ensemble *h, const wt::StepArgs &a, unsigned grid, hipStream_t stream)
{
    with_step_kernel(h, [&](auto kernel, int block) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, a); });
}

// one kernel launch, bracketed by HIP events on its own stream when launch timing is on
void launch_step(wt_ensemble *h, const wt::StepArgs &a, unsigned grid, hipStream_t stream)
{
    bool timed = h->time_launches;
    if (timed && h->lt_used + 2 > h->lt_pool.size()) {
        for (int i = 0; i < 2 && timed; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) timed = false; else h->lt_pool.push_back(e);
        }
        if (!timed) while (h->lt_pool.size() > h->lt_used) { (void)hipEventDestroy(h->lt_pool.back()); h->lt_pool.pop_back(); }
    }
    if (timed) (void)hipEventRecord(h->lt_pool[h->lt_used], stream);
    launch_step_raw(h, a, grid, stream);
    if (timed) { (void)hipEventRecord(h->lt_pool[h->lt_used + 1], stream); h->lt_used += 2; }
}

// Outer steps per work item of the queue schedule.  A group changes hands at item boundaries, which costs a few
// microseconds (state out and in, release / acquire), so not every step -- but often enough that the groups
// sharing the workers take turns at least half a dozen times; at most 32 steps.
int queue_item_steps(const wt_ensemble *h, int n_steps)
{
    int item = n_steps / 6;
    if (item > 32) item = 32;
    if (item < 1) item = 1;
    if (h->knob_item > 0) item = h->knob_item;
    return item;
}

// worker wavefronts of the queue schedule: as many as the device keeps resident (no more than there are groups)
int queue_workers(const wt_ensemble *h)
{
    int per_cu = 0, cus = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) cus = prop.multiProcessorCount;
    with_step_kernel(h, [&](auto kernel, int block) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess) per_cu = 0;
    });
    if (cus <= 0) cus = 256;
    if (per_cu <= 0) per_cu = 4;
    const int64_t cap = (int64_t)cus * per_cu;
    return (int)(h->n_groups < cap ? h->n_groups : cap);
}

const char *k_incomplete = "a step launch did not advance every wavefront-group (work-queue hand-off timed out or a group was left behind); the state on the device is incomplete";

// queue the copy of the sticky launch-error word (pinned destination) behind what is already on the stream
int fetch_sticky(wt_ensemble *h)
{
    (void)h;   // nothing to copy: the record lives in host-coherent memory, valid once the stream is synchronised
    return WT_OK;
}

} // namespace

extern "C" {

int wt_abi_version(void) { return WT_ABI_VERSION; }
const char *wt_last_error(void) { return g_err.c_str(); }

int wt_device_count(int *count)
{
    if (!count) return fail(WT_E_ARG, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; return fail(WT_E_NOGPU, hipGetErrorString(e)); }
    *count = c;
    return WT_OK;
}

int wt_ensemble_create(int64_t n_reactors, int n_zones, int device, const double *par, wt_ensemble **out)
{
    if (!out || !par) return fail(WT_E_ARG, "NULL argument");
    *out = nullptr;
    if (n_reactors <= 0) return fail(WT_E_ARG, "n_reactors must be positive");
    if (n_zones < 2 || n_zones > WT_MAX_ZONES)
        return fail(WT_E_ARG, "Need at least 2 zones and at most 64, got " + std::to_string(n_zones));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(WT_E_NOGPU, "no HIP device available: libwtphys has no CPU path");
    if (device < 0 || device >= ndev) return fail(WT_E_ARG, "bad device index");
    HIP_TRY(hipSetDevice(device));
    wt_ensemble *h = new wt_ensemble();
    h->N = n_reactors; h->n = n_zones; h->R = 64 / n_zones; h->device = device;
    {   // A small ensemble is spread over all SIMDs rather than packed into full wavefronts: a wavefront costs what
        // its slowest reactor costs, so fewer reactors per wavefront is faster as long as every wavefront still
        // finds a SIMD (about 4 per CU).  Results do not depend on it (reactors never interact).
        int cus = 256; hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        const int64_t slots = (int64_t)cus * 4;
        int64_t r = (n_reactors + slots - 1) / slots;
        if (const char *e = getenv("WT_FULL_WAVES")) if (atoi(e) != 0) r = h->R;      // tuning knob (tools/)
        if (r < 1) r = 1;
        if (r < h->R) h->R = (int)r;
    }
    const size_t N = (size_t)n_reactors, nz = (size_t)n_zones;
    auto cleanup = [&]() { wt_ensemble_destroy(h); };
#define ALLOC(ptr, bytes)                                                                   \
    do {                                                                                    \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes));                                \
        if (e_ != hipSuccess) { cleanup(); return fail(WT_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e_)); } \
    } while (0)
    ALLOC(h->par, sizeof(double) * WT_NP * N);
    ALLOC(h->bc, sizeof(double) * WT_NB * N);
    ALLOC(h->pH, sizeof(double) * N * nz);
    ALLOC(h->Cl, sizeof(double) * N * nz);
    ALLOC(h->T, sizeof(double) * N * nz);
    ALLOC(h->dH, sizeof(double) * N * nz);
    ALLOC(h->dRho, sizeof(double) * N * nz);
    ALLOC(h->dK, sizeof(double) * N * nz);
    ALLOC(h->time, sizeof(double) * N);
    ALLOC(h->flow, sizeof(double) * N);
    ALLOC(h->status, sizeof(uint32_t) * N);
    ALLOC(h->stats, sizeof(int32_t) * 5 * N);
    ALLOC(h->bad_T, sizeof(double) * N);
    ALLOC(h->perm, sizeof(int32_t) * N);
    ALLOC(h->cost, sizeof(int32_t) * N);
    ALLOC(h->place_hist, sizeof(int32_t) * wtpl::BINS * ((N + wtpl::CHUNK - 1) / wtpl::CHUNK));
    h->n_groups = (n_reactors + h->R - 1) / h->R;
    if (h->n_groups > 0x3fffffff) { cleanup(); return fail(WT_E_ARG, "too many reactors for one ensemble"); }
    h->q_cap = (int)(2 * h->n_groups + 64);
    ALLOC(h->q_ctrl, sizeof(int32_t) * wt::Q_WORDS);
    ALLOC(h->q_slots, sizeof(unsigned long long) * (size_t)h->q_cap);
    ALLOC(h->q_next, sizeof(int32_t) * (size_t)h->n_groups);
    {   // small ensembles (the drop-in's N = 1 above all) are downloaded as one packed image through pinned memory
        const size_t image = sizeof(double) * (6 * N * nz + 2 * N) + sizeof(uint32_t) * (N + 1);
        if (image <= WT_SNAPSHOT_PACK_MAX) {
            h->snap_bytes = (image + 7) & ~(size_t)7;
            ALLOC(h->snap_dev, h->snap_bytes);
            if (hipHostMalloc(&h->snap_host, h->snap_bytes, hipHostMallocDefault) != hipSuccess) { cleanup(); return fail(WT_E_HIP, "hipHostMalloc failed"); }
        }
        if (hipHostMalloc((void **)&h->err_host, 2 * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { cleanup(); return fail(WT_E_HIP, "hipHostMalloc failed"); }
        h->err_host[0] = h->err_host[1] = 0;
        if (hipHostGetDevicePointer((void **)&h->q_sticky, h->err_host, 0) != hipSuccess) { cleanup(); return fail(WT_E_HIP, "hipHostGetDevicePointer failed"); }
    }
    if (const char *e = getenv("WT_Q_ITEM")) h->knob_item = atoi(e);
    if (const char *e = getenv("WT_PLACE_MIN")) h->knob_place_min = atoll(e);
    if (const char *e = getenv("WT_Q_TICKETS")) h->knob_tickets = atoll(e);
    if (const char *e = getenv("WT_DENSE_COUPLING")) h->knob_dense = atoi(e) != 0;
#undef ALLOC
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { cleanup(); return fail(WT_E_HIP, "hipStreamCreate failed"); }
    h->own_stream = true;
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) { cleanup(); return fail(WT_E_HIP, "hipEventCreate failed"); }
    if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) { cleanup(); return fail(WT_E_HIP, "hipEventCreate failed"); }
    h->n_sub = default_streams(n_reactors, h->R);
    h->chunk_steps = WT_DEFAULT_CHUNK;
    h->sched_mode = WT_SCHED_QUEUE;
    h->q_workers = queue_workers(h);
    hipError_t e = hipMemcpyAsync(h->par, par, sizeof(double) * WT_NP * N, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->status, 0, sizeof(uint32_t) * N, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->stats, 0, sizeof(int32_t) * 5 * N, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->time, 0, sizeof(double) * N, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->flow, 0, sizeof(double) * N, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->bad_T, 0, sizeof(double) * N, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->cost, 0, sizeof(int32_t) * N, h->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(wtpl::iota_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, h->perm, (int64_t)N);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemsetAsync(h->q_ctrl, 0, sizeof(int32_t) * wt::Q_WORDS, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { cleanup(); return fail(WT_E_HIP, std::string("upload: ") + hipGetErrorString(e)); }
    *out = h;
    return WT_OK;
}

int wt_ensemble_destroy(wt_ensemble *h)
{
    if (!h) return WT_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->trace) (void)hipFree(h->trace);
    void *ptrs[] = {h->par, h->bc, h->pH, h->Cl, h->T, h->time, h->flow, h->dH, h->dRho, h->dK, h->status, h->stats, h->wave_diag,
                    h->bad_T, h->q_ctrl, h->q_slots, h->q_next, h->perm, h->cost, h->place_hist, h->snap_dev};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (int s = 0; s < WT_MAX_STREAMS; ++s) {
        if (h->sub_stream[s]) { (void)hipStreamSynchronize(h->sub_stream[s]); (void)hipStreamDestroy(h->sub_stream[s]); }
        if (h->sub_done[s]) (void)hipEventDestroy(h->sub_done[s]);
    }
    void *sp[] = {h->s_fs, h->s_full_scale, h->s_ring_t, h->s_ring_v, h->s_out_value, h->s_hist_value, h->s_ds, h->s_t_enable, h->s_is,
                  h->s_ring_head, h->s_ring_cnt, h->s_hist_pos, h->s_out_status, h->s_out_fault, h->s_hist_status,
                  h->s_hist_fault, h->p_ir, h->p_hr, h->p_loop_time, h->p_update_ok, h->diag_out};
    for (void *p : sp) if (p) (void)hipFree(p);
    if (h->snap_host) (void)hipHostFree(h->snap_host);
    if (h->err_host) (void)hipHostFree(h->err_host);
    for (hipEvent_t e : h->lt_pool) (void)hipEventDestroy(e);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return WT_OK;
}

int wt_ensemble_set_stream(wt_ensemble *h, void *hip_stream)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    return WT_OK;
}

int wt_ensemble_set_state(wt_ensemble *h, const double *pH, const double *Cl, const double *T, const double *time)
{
    if (!h || !pH || !Cl || !T) return fail(WT_E_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t cnt = (size_t)h->N * h->n;
    HIP_TRY(hipMemcpyAsync(h->pH, pH, sizeof(double) * cnt, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->Cl, Cl, sizeof(double) * cnt, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->T, T, sizeof(double) * cnt, hipMemcpyHostToDevice, h->stream));
    if (time) HIP_TRY(hipMemcpyAsync(h->time, time, sizeof(double) * h->N, hipMemcpyHostToDevice, h->stream));
    // ReactorState.update_derived placeholders (reactor.py:137-147), filled where the state now lives
    wt::PlaceholderArgs pa{(int64_t)cnt, h->pH, h->dH, h->dRho, h->dK};
    hipLaunchKernelGGL(wt::derived_placeholder_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, pa);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(h->status, 0, sizeof(uint32_t) * h->N, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));   // the caller's buffers are free on return
    h->have_state = true;
    return WT_OK;
}

int wt_ensemble_set_boundary(wt_ensemble *h, const double *bc)
{
    if (!h || !bc) return fail(WT_E_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->bc, bc, sizeof(double) * WT_NB * h->N, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->have_bc = true;
    return WT_OK;
}

int wt_ensemble_step(wt_ensemble *h, double dt, int n_steps, int fused)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    if (!h->have_state || !h->have_bc) return fail(WT_E_STATE, "set_state and set_boundary must precede step");
    if (!(dt > 0)) return fail(WT_E_ARG, "`max_step` must be positive."); // scipy validate_max_step (reactor.py:480)
    if (n_steps < 0) return fail(WT_E_ARG, "n_steps must be >= 0");
    if (n_steps == 0) return WT_OK;
    HIP_TRY(hipSetDevice(h->device));
    // chunk_steps: PLC scan interval; under the stream schedule also the launch length
    const int chunk = fused ? (h->chunk_steps > 0 ? h->chunk_steps : n_steps) : 1;
    if (h->wave_diag)
        HIP_TRY(hipMemsetAsync(h->wave_diag, 0, sizeof(int64_t) * wt::WT_DIAG_SLOTS * (size_t)h->n_groups, h->stream));
    // (the stream schedule feeds the cost history too; only the queue schedule re-deals)
    struct CountSteps { wt_ensemble *h; int n; ~CountSteps() { if (h->placement == WT_PLACE_ADAPTIVE && h->sched_mode != WT_SCHED_QUEUE) h->cost_steps += n; } } count_steps{h, n_steps};
    if (h->sched_mode == WT_SCHED_QUEUE) {
        // One launch of q_workers worker wavefronts (more than one only if the call is so long that the queue's
        // 32-bit tickets -- one per work item -- could run out: groups x items per launch stays below 2^30).
        const int W = h->q_workers > 0 ? h->q_workers : 1;
        const int item = queue_item_steps(h, n_steps);
        // Reactors of similar solver cost share a wavefront: once the cost history covers enough outer steps to tell
        // a reactor's regime from a burst, the slots are re-dealt in cost order (three small kernels, no sync).
        const int64_t min_steps = h->knob_place_min > 0 ? h->knob_place_min : WT_PLACE_MIN_STEPS;
        auto redeal = [&]() {
            if (h->placement != WT_PLACE_ADAPTIVE || h->cost_steps < min_steps) return;
            const int blocks = (int)((h->N + wtpl::CHUNK - 1) / wtpl::CHUNK);
            const wtpl::PlaceArgs pa{h->N, h->cost, (int)(h->cost_steps > 0x7fffffff ? 0x7fffffff : h->cost_steps), h->place_hist, h->perm};
            hipLaunchKernelGGL(wtpl::place_count_kernel, dim3(blocks), dim3(wtpl::BINS), 0, h->stream, pa);
            hipLaunchKernelGGL(wtpl::place_scan_kernel, dim3(1), dim3(wtpl::BINS), 0, h->stream, pa, blocks);
            hipLaunchKernelGGL(wtpl::place_scatter_kernel, dim3(blocks), dim3(wtpl::BINS), 0, h->stream, pa);
            h->cost_steps = 0; h->redeals++;
        };
        const int64_t tickets = h->knob_tickets > 0 ? h->knob_tickets : (int64_t)1 << 30;
        int64_t per_launch = tickets / h->n_groups * item;
        if (per_launch < item) per_launch = item;
        if (h->n_groups <= W && h->knob_tickets <= 0) {
            // every group has a worker of its own: nothing to hand over, so no queue -- one plain launch in which
            // workgroup g advances group g by the whole call (the drop-in's N = 1 lives here: one kernel per step())
            redeal();
            const wt::StepArgs a = make_args(h, dt, n_steps, 0, n_steps, chunk);
            launch_step(h, a, (unsigned)h->n_groups, h->stream);
            if (h->placement == WT_PLACE_ADAPTIVE) h->cost_steps += n_steps;
            HIP_TRY(hipGetLastError());
            return WT_OK;
        }
        for (int64_t done = 0; done < n_steps; done += per_launch) {
            const int cnt = (int)((n_steps - done < per_launch) ? n_steps - done : per_launch);
            redeal();
            wt::StepArgs a = make_args(h, dt, cnt, (int)done, n_steps, chunk);
            a.q_ctrl = h->q_ctrl; a.q_slots = h->q_slots; a.q_next = h->q_next; a.q_cap = h->q_cap; a.item_steps = item;
            wt::QueueResetArgs qr{h->q_ctrl, h->q_slots, h->q_next, (int)h->n_groups, h->q_cap};
            hipLaunchKernelGGL(wt::queue_reset_kernel, dim3((unsigned)((h->q_cap + 255) / 256)), dim3(256), 0, h->stream, qr);
            launch_step(h, a, (unsigned)W, h->stream);
            const wt::QueueCheckArgs qc{h->q_ctrl, h->q_next, (int)h->n_groups, cnt, h->q_sticky};
            hipLaunchKernelGGL(wt::queue_check_kernel, dim3((unsigned)((h->n_groups + 255) / 256)), dim3(256), 0, h->stream, qc);
            if (h->placement == WT_PLACE_ADAPTIVE) h->cost_steps += cnt;
        }
        HIP_TRY(hipGetLastError());
        return WT_OK;
    }
    const int S = h->n_sub;
    if (S <= 1) {
        for (int done = 0; done < n_steps; done += chunk) {
            const wt::StepArgs a = make_args(h, dt, (n_steps - done < chunk) ? n_steps - done : chunk, done, n_steps, chunk);
            launch_step(h, a, (unsigned)h->n_groups, h->stream);
        }
        HIP_TRY(hipGetLastError());
        return WT_OK;
    }
    for (int s = 0; s < S; ++s) {   // streams of the ranges are created on first use
        if (!h->sub_stream[s]) {
            HIP_TRY(hipStreamCreateWithFlags(&h->sub_stream[s], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&h->sub_done[s], hipEventDisableTiming));
        }
    }
    // fork: every range's stream waits for what is already queued on the handle's stream
    HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
    for (int s = 0; s < S; ++s) HIP_TRY(hipStreamWaitEvent(h->sub_stream[s], h->ev_fork, 0));
    const int64_t groups = h->n_groups;   // wavefront-sized groups of reactors
    for (int done = 0; done < n_steps; done += chunk) {
        wt::StepArgs a = make_args(h, dt, (n_steps - done < chunk) ? n_steps - done : chunk, done, n_steps, chunk);
        for (int s = 0; s < S; ++s) {
            const int64_t g0 = groups * s / S, g1 = groups * (s + 1) / S;
            a.r0 = g0 * h->R; a.r1 = (g1 * h->R < h->N) ? g1 * h->R : h->N;
            if (a.r1 > a.r0) launch_step(h, a, (unsigned)(g1 - g0), h->sub_stream[s]);
        }
    }
    HIP_TRY(hipGetLastError());
    // join: later work on the handle's stream (copies, timers) sees every range finished
    for (int s = 0; s < S; ++s) {
        HIP_TRY(hipEventRecord(h->sub_done[s], h->sub_stream[s]));
        HIP_TRY(hipStreamWaitEvent(h->stream, h->sub_done[s], 0));
    }
    return WT_OK;
}

int wt_ensemble_launch_timing(wt_ensemble *h, int enable)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    h->time_launches = enable != 0;
    h->lt_used = 0;
    // the first few event pairs exist before the timed region starts (event creation is not part of a launch)
    while (enable && h->lt_pool.size() < 16) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        h->lt_pool.push_back(e);
    }
    return WT_OK;
}

int wt_ensemble_launch_stats(wt_ensemble *h, int64_t *n_launches, double *sum_ms, double *max_ms)
{
    if (!h || !n_launches || !sum_ms || !max_ms) return fail(WT_E_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    double sum = 0.0, mx = 0.0;
    for (size_t i = 0; i + 1 < h->lt_used; i += 2) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(h->lt_pool[i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, h->lt_pool[i], h->lt_pool[i + 1]));
        sum += ms; if (ms > mx) mx = ms;
    }
    *n_launches = (int64_t)(h->lt_used / 2); *sum_ms = sum; *max_ms = mx;
    h->lt_used = 0;
    return WT_OK;
}

int wt_ensemble_sensors_enable(wt_ensemble *h, uint64_t seed, int64_t reactor_base, const double *cfg_flow,
                               const double *cfg_chlorine, const double *cfg_temperature, int history_capacity)
{
    if (!h || !cfg_flow || !cfg_chlorine || !cfg_temperature) return fail(WT_E_ARG, "NULL argument");
    if (history_capacity < 0) return fail(WT_E_ARG, "history_capacity must be >= 0");
    if (h->sensors_on) return fail(WT_E_STATE, "sensor suite already enabled");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t N = (size_t)h->N;
    double *cfg = nullptr;
    HIP_TRY(hipMalloc((void **)&cfg, sizeof(double) * 3 * N));
#define SALLOC(ptr, bytes) do { if (hipMalloc((void **)&(ptr), (bytes)) != hipSuccess) { (void)hipFree(cfg); release_sensor_buffers(h); return fail(WT_E_HIP, "hipMalloc (sensors) failed"); } } while (0)
    SALLOC(h->s_fs, sizeof(float) * wts::NSENS * wts::NF * N);
    SALLOC(h->s_ds, sizeof(double) * wts::NSENS * wts::ND * N);
    SALLOC(h->s_is, sizeof(int32_t) * wts::NSENS * wts::NI * N);
    SALLOC(h->s_full_scale, sizeof(float) * N);
    SALLOC(h->s_ring_t, sizeof(float) * 2 * wts::RING * N);
    SALLOC(h->s_ring_v, sizeof(float) * 2 * wts::RING * N);
    SALLOC(h->s_ring_head, sizeof(int32_t) * 2 * N);
    SALLOC(h->s_ring_cnt, sizeof(int32_t) * 2 * N);
    SALLOC(h->s_out_value, sizeof(float) * wts::NSENS * N);
    SALLOC(h->s_out_status, wts::NSENS * N);
    SALLOC(h->s_out_fault, wts::NSENS * N);
    SALLOC(h->s_t_enable, sizeof(double) * N);
    h->s_hist_cap = history_capacity;
    if (history_capacity > 0) {
        SALLOC(h->s_hist_value, sizeof(float) * (size_t)history_capacity * wts::NSENS * N);
        SALLOC(h->s_hist_status, (size_t)history_capacity * wts::NSENS * N);
        SALLOC(h->s_hist_fault, (size_t)history_capacity * wts::NSENS * N);
        SALLOC(h->s_hist_pos, sizeof(int32_t) * N);
        // slots that are never written (a reactor whose step raised takes no reading) must not hold garbage
        (void)hipMemsetAsync(h->s_hist_value, 0, sizeof(float) * (size_t)history_capacity * wts::NSENS * N, h->stream);
        (void)hipMemsetAsync(h->s_hist_status, 0, (size_t)history_capacity * wts::NSENS * N, h->stream);
        (void)hipMemsetAsync(h->s_hist_fault, 0, (size_t)history_capacity * wts::NSENS * N, h->stream);
    }
#undef SALLOC
    hipError_t e = hipMemcpy(cfg, cfg_flow, sizeof(double) * N, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(cfg + N, cfg_chlorine, sizeof(double) * N, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(cfg + 2 * N, cfg_temperature, sizeof(double) * N, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpyAsync(h->s_t_enable, h->time, sizeof(double) * N, hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->s_ring_t, 0, sizeof(float) * 2 * wts::RING * N, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->s_ring_v, 0, sizeof(float) * 2 * wts::RING * N, h->stream);
    if (e == hipSuccess) {
        wts::SensorInitArgs a;
        a.N = h->N; a.cfg_flow = cfg; a.cfg_cl = cfg + N; a.cfg_temp = cfg + 2 * N;
        a.fs = h->s_fs; a.ds = h->s_ds; a.is = h->s_is; a.full_scale = h->s_full_scale;
        a.ring_push = h->s_ring_head; a.ring_cursor = h->s_ring_cnt;
        a.out_value = h->s_out_value; a.out_status = h->s_out_status; a.out_fault = h->s_out_fault; a.hist_pos = h->s_hist_pos;
        hipLaunchKernelGGL(wts::sensor_init_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(cfg);
    if (e != hipSuccess) { release_sensor_buffers(h); return fail(WT_E_HIP, std::string("sensors_enable: ") + hipGetErrorString(e)); }
    h->sens_seed = seed; h->sens_reactor_base = reactor_base;
    h->sensors_on = true;
    return WT_OK;
}

int wt_ensemble_sensors_get(wt_ensemble *h, float *values, uint8_t *status, uint8_t *fault)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    if (!h->sensors_on) return fail(WT_E_STATE, "sensor suite not enabled");
    HIP_TRY(hipSetDevice(h->device));
    const size_t cnt = (size_t)wts::NSENS * h->N;
    if (values) HIP_TRY(hipMemcpyAsync(values, h->s_out_value, sizeof(float) * cnt, hipMemcpyDeviceToHost, h->stream));
    if (status) HIP_TRY(hipMemcpyAsync(status, h->s_out_status, cnt, hipMemcpyDeviceToHost, h->stream));
    if (fault) HIP_TRY(hipMemcpyAsync(fault, h->s_out_fault, cnt, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WT_OK;
}

int wt_ensemble_sensors_history(wt_ensemble *h, float *values, uint8_t *status, uint8_t *fault, int32_t *n_filled)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    if (!h->sensors_on || h->s_hist_cap <= 0) return fail(WT_E_STATE, "sensor history not enabled");
    HIP_TRY(hipSetDevice(h->device));
    const size_t cnt = (size_t)h->s_hist_cap * wts::NSENS * h->N;
    if (values) HIP_TRY(hipMemcpyAsync(values, h->s_hist_value, sizeof(float) * cnt, hipMemcpyDeviceToHost, h->stream));
    if (status) HIP_TRY(hipMemcpyAsync(status, h->s_hist_status, cnt, hipMemcpyDeviceToHost, h->stream));
    if (fault) HIP_TRY(hipMemcpyAsync(fault, h->s_hist_fault, cnt, hipMemcpyDeviceToHost, h->stream));
    if (n_filled) HIP_TRY(hipMemcpyAsync(n_filled, h->s_hist_pos, sizeof(int32_t) * h->N, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WT_OK;
}

int wt_ensemble_plc_enable(wt_ensemble *h)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    if (!h->sensors_on) return fail(WT_E_STATE, "the register image publishes sensor readings: enable the sensor suite first");
    if (h->plc_on) return fail(WT_E_STATE, "plant I/O already enabled");
    HIP_TRY(hipSetDevice(h->device));
    const size_t N = (size_t)h->N;
    hipError_t e = hipMalloc((void **)&h->p_ir, sizeof(uint16_t) * wtp::IR_WORDS * N);
    if (e == hipSuccess) e = hipMalloc((void **)&h->p_hr, sizeof(uint16_t) * wtp::HR_WORDS * N);
    if (e == hipSuccess) e = hipMalloc((void **)&h->p_loop_time, sizeof(double) * N);
    if (e == hipSuccess) e = hipMalloc((void **)&h->p_update_ok, N);
    // ModbusSequentialDataBlock(0, [0] * size): every register starts at 0 (slave.py:134-137); sim_time = 0.0
    if (e == hipSuccess) e = hipMemsetAsync(h->p_ir, 0, sizeof(uint16_t) * wtp::IR_WORDS * N, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->p_hr, 0, sizeof(uint16_t) * wtp::HR_WORDS * N, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->p_loop_time, 0, sizeof(double) * N, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->p_update_ok, 1, N, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { release_plc_buffers(h); return fail(WT_E_HIP, std::string("plc_enable: ") + hipGetErrorString(e)); }
    h->plc_on = true;
    return WT_OK;
}

int wt_ensemble_plc_write_holding(wt_ensemble *h, const uint16_t *words, int64_t first_reactor, int64_t count)
{
    if (!h || !words) return fail(WT_E_ARG, "NULL argument");
    if (!h->plc_on) return fail(WT_E_STATE, "plant I/O not enabled");
    if (first_reactor < 0 || count < 0 || first_reactor + count > h->N) return fail(WT_E_ARG, "reactor range outside the ensemble");
    if (count == 0) return WT_OK;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->p_hr + first_reactor * wtp::HR_WORDS, words, sizeof(uint16_t) * wtp::HR_WORDS * (size_t)count,
                           hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));   // the caller's buffer is free on return
    return WT_OK;
}

int wt_ensemble_plc_read_inputs(wt_ensemble *h, uint16_t *words, uint8_t *update_ok)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    if (!h->plc_on) return fail(WT_E_STATE, "plant I/O not enabled");
    HIP_TRY(hipSetDevice(h->device));
    if (words) HIP_TRY(hipMemcpyAsync(words, h->p_ir, sizeof(uint16_t) * wtp::IR_WORDS * (size_t)h->N, hipMemcpyDeviceToHost, h->stream));
    if (update_ok) HIP_TRY(hipMemcpyAsync(update_ok, h->p_update_ok, (size_t)h->N, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WT_OK;
}

int wt_ensemble_plc_device(wt_ensemble *h, void **input_image, void **holding_image)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    if (!h->plc_on) return fail(WT_E_STATE, "plant I/O not enabled");
    if (input_image) *input_image = h->p_ir;
    if (holding_image) *holding_image = h->p_hr;
    return WT_OK;
}

int wt_ensemble_get_boundary(wt_ensemble *h, double *bc)
{
    if (!h || !bc) return fail(WT_E_ARG, "NULL argument");
    if (!h->have_bc) return fail(WT_E_STATE, "set_boundary must precede get_boundary");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(bc, h->bc, sizeof(double) * WT_NB * (size_t)h->N, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WT_OK;
}

int wt_ensemble_diagnostics(wt_ensemble *h, double *out)
{
    if (!h || !out) return fail(WT_E_ARG, "NULL argument");
    if (!h->have_state) return fail(WT_E_STATE, "set_state must precede diagnostics");
    HIP_TRY(hipSetDevice(h->device));
    const size_t bytes = sizeof(double) * wtd::N_DIAG * (size_t)h->N;
    if (!h->diag_out) HIP_TRY(hipMalloc((void **)&h->diag_out, bytes));
    wtd::DiagArgs a;
    a.N = h->N; a.n = h->n; a.par = h->par; a.pH = h->pH; a.Cl = h->Cl; a.T = h->T; a.H = h->dH; a.out = h->diag_out;
    hipLaunchKernelGGL(wtd::diagnostics_kernel, dim3((unsigned)((h->N + 63) / 64)), dim3(64), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->diag_out, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WT_OK;
}

int wt_ensemble_set_placement(wt_ensemble *h, int mode)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    if (mode != WT_PLACE_IDENTITY && mode != WT_PLACE_ADAPTIVE) return fail(WT_E_ARG, "placement mode must be WT_PLACE_IDENTITY or WT_PLACE_ADAPTIVE");
    HIP_TRY(hipSetDevice(h->device));
    if (mode == WT_PLACE_IDENTITY)
        hipLaunchKernelGGL(wtpl::iota_kernel, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, h->perm, h->N);
    HIP_TRY(hipMemsetAsync(h->cost, 0, sizeof(int32_t) * (size_t)h->N, h->stream));
    HIP_TRY(hipGetLastError());
    h->placement = mode;
    h->cost_steps = 0;
    return WT_OK;
}

int wt_ensemble_get_placement(wt_ensemble *h, int *mode, int32_t *perm)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    if (mode) *mode = h->placement;
    if (perm) {
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMemcpyAsync(perm, h->perm, sizeof(int32_t) * (size_t)h->N, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return WT_OK;
}

int wt_ensemble_placement_info(wt_ensemble *h, int64_t *redeals, int64_t *history_steps)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    if (redeals) *redeals = h->redeals;
    if (history_steps) *history_steps = h->cost_steps;
    return WT_OK;
}

int wt_ensemble_set_step_limit(wt_ensemble *h, int max_attempts)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    if (max_attempts < 0) return fail(WT_E_ARG, "max_attempts must be >= 0 (0 = unlimited)");
    h->step_limit = max_attempts;
    return WT_OK;
}

int wt_ensemble_set_sync(wt_ensemble *h, int sync_outer)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    (void)sync_outer;   // the reactors of a wavefront always start an outer step together (see include/wtphys.h)
    return WT_OK;
}

int wt_ensemble_set_schedule(wt_ensemble *h, int n_streams, int chunk_steps)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    if (n_streams < 0 || n_streams > WT_MAX_STREAMS) return fail(WT_E_ARG, "n_streams out of range");
    if (chunk_steps < 0) return fail(WT_E_ARG, "chunk_steps must be >= 0 (0 = one scan / launch per call)");
    h->sched_mode = n_streams > 0 ? WT_SCHED_STREAMS : WT_SCHED_QUEUE;
    h->n_sub = n_streams > 0 ? n_streams : default_streams(h->N, h->R);
    h->chunk_steps = chunk_steps;
    return WT_OK;
}

int wt_ensemble_get_schedule(wt_ensemble *h, int *mode, int *n_streams, int *chunk_steps, int *workers)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    if (mode) *mode = h->sched_mode;
    if (n_streams) *n_streams = h->sched_mode == WT_SCHED_STREAMS ? h->n_sub : 0;
    if (chunk_steps) *chunk_steps = h->chunk_steps;
    if (workers) *workers = h->sched_mode == WT_SCHED_QUEUE ? h->q_workers : 0;
    return WT_OK;
}

int wt_ensemble_item_steps(wt_ensemble *h, int n_steps)
{
    if (!h || n_steps <= 0) return 0;
    if (h->sched_mode == WT_SCHED_QUEUE) return (h->n_groups <= h->q_workers && h->knob_tickets <= 0) ? n_steps : queue_item_steps(h, n_steps);
    const int chunk = h->chunk_steps > 0 ? h->chunk_steps : n_steps;
    return chunk < n_steps ? chunk : n_steps;
}

int wt_ensemble_queue_error(wt_ensemble *h, int *error)
{
    if (!h || !error) return fail(WT_E_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc;
    if ((rc = fetch_sticky(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    *error = (h->err_host[0] ? 1 : 0) | (h->err_host[1] ? 2 : 0);      // bit 0: hand-off timed out, bit 1: a group was left behind; sticky until the handle is destroyed
    return WT_OK;
}

int wt_ensemble_item_trace(wt_ensemble *h, int64_t *out, int capacity, int *n_items)
{
    if (!h || !n_items) return fail(WT_E_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!h->trace) {     // first call switches tracing on
        if (capacity <= 0) return fail(WT_E_ARG, "capacity must be positive");
        HIP_TRY(hipMalloc((void **)&h->trace, sizeof(int64_t) * 5 * (size_t)capacity));
        h->trace_cap = capacity; *n_items = 0;
        return WT_OK;
    }
    int32_t w[wt::Q_WORDS];
    HIP_TRY(hipMemcpy(w, h->q_ctrl, sizeof w, hipMemcpyDeviceToHost));
    const int n = w[wt::Q_TRACE] < h->trace_cap ? w[wt::Q_TRACE] : h->trace_cap;
    *n_items = n;
    if (out && n > 0) {
        if (capacity < n) return fail(WT_E_ARG, "trace buffer too small");
        HIP_TRY(hipMemcpy(out, h->trace, sizeof(int64_t) * 5 * (size_t)n, hipMemcpyDeviceToHost));
    }
    return WT_OK;
}

int wt_ensemble_get_bad_temperature(wt_ensemble *h, double *value)
{
    if (!h || !value) return fail(WT_E_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(value, h->bad_T, sizeof(double) * h->N, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WT_OK;
}

int wt_ensemble_synchronize(wt_ensemble *h)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    int rc;
    if ((rc = fetch_sticky(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->err_host[0] | h->err_host[1]) return fail(WT_E_HIP, k_incomplete);
    return WT_OK;
}

static int d2h(wt_ensemble *h, void *dst, const void *src, size_t bytes)
{
    if (!dst) return WT_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    return WT_OK;
}

// Small ensembles: pack everything a snapshot can ask for into one device image, one copy into pinned memory, one
// synchronisation, then scatter into the caller's arrays on the host.  (N = 1: 9 pageable copies of a few dozen
// bytes each were most of the drop-in's per-step latency.)
static int packed_snapshot(wt_ensemble *h, double *pH, double *Cl, double *T, double *time, double *flow,
                           double *H, double *rho, double *kdecay, uint32_t *flags)
{
    const int64_t cnt = h->N * h->n;
    wt::SnapshotArgs a{cnt, h->N, h->pH, h->Cl, h->T, h->dH, h->dRho, h->dK, h->time, h->flow, h->status, h->q_sticky, (double *)h->snap_dev};
    hipLaunchKernelGGL(wt::snapshot_pack_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h->snap_host, h->snap_dev, h->snap_bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const double *img = (const double *)h->snap_host;
    const size_t b = sizeof(double) * (size_t)cnt, bn = sizeof(double) * (size_t)h->N;
    double *zone[6] = {pH, Cl, T, H, rho, kdecay};
    for (int i = 0; i < 6; ++i) if (zone[i]) memcpy(zone[i], img + (size_t)i * cnt, b);
    const double *tail = img + 6 * (size_t)cnt;
    if (time) memcpy(time, tail, bn);
    if (flow) memcpy(flow, tail + h->N, bn);
    const uint32_t *w = (const uint32_t *)(tail + 2 * h->N);
    if (flags) memcpy(flags, w, sizeof(uint32_t) * (size_t)h->N);
    if (w[h->N] != 0) return fail(WT_E_HIP, k_incomplete);
    return WT_OK;
}

int wt_ensemble_get_state(wt_ensemble *h, double *pH, double *Cl, double *T, double *time, double *flow)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    if (h->snap_host) return packed_snapshot(h, pH, Cl, T, time, flow, nullptr, nullptr, nullptr, nullptr);
    const size_t b = sizeof(double) * (size_t)h->N * h->n;
    int rc;
    if ((rc = d2h(h, pH, h->pH, b))) return rc;
    if ((rc = d2h(h, Cl, h->Cl, b))) return rc;
    if ((rc = d2h(h, T, h->T, b))) return rc;
    if ((rc = d2h(h, time, h->time, sizeof(double) * h->N))) return rc;
    if ((rc = d2h(h, flow, h->flow, sizeof(double) * h->N))) return rc;
    if ((rc = fetch_sticky(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->err_host[0] | h->err_host[1]) return fail(WT_E_HIP, k_incomplete);
    return WT_OK;
}

int wt_ensemble_get_snapshot(wt_ensemble *h, double *pH, double *Cl, double *T, double *time, double *flow,
                             double *H, double *rho, double *kdecay, uint32_t *flags)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    if (h->snap_host) return packed_snapshot(h, pH, Cl, T, time, flow, H, rho, kdecay, flags);
    const size_t b = sizeof(double) * (size_t)h->N * h->n;
    int rc;
    if ((rc = d2h(h, pH, h->pH, b)) || (rc = d2h(h, Cl, h->Cl, b)) || (rc = d2h(h, T, h->T, b))) return rc;
    if ((rc = d2h(h, time, h->time, sizeof(double) * h->N)) || (rc = d2h(h, flow, h->flow, sizeof(double) * h->N))) return rc;
    if ((rc = d2h(h, H, h->dH, b)) || (rc = d2h(h, rho, h->dRho, b)) || (rc = d2h(h, kdecay, h->dK, b))) return rc;
    if ((rc = d2h(h, flags, h->status, sizeof(uint32_t) * h->N))) return rc;
    if ((rc = fetch_sticky(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->err_host[0] | h->err_host[1]) return fail(WT_E_HIP, k_incomplete);
    return WT_OK;
}

int wt_ensemble_get_derived(wt_ensemble *h, double *H, double *rho, double *kdecay)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    const size_t b = sizeof(double) * (size_t)h->N * h->n;
    int rc;
    if ((rc = d2h(h, H, h->dH, b))) return rc;
    if ((rc = d2h(h, rho, h->dRho, b))) return rc;
    if ((rc = d2h(h, kdecay, h->dK, b))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WT_OK;
}

int wt_ensemble_get_status(wt_ensemble *h, uint32_t *flags)
{
    if (!h || !flags) return fail(WT_E_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(flags, h->status, sizeof(uint32_t) * h->N, hipMemcpyDeviceToHost, h->stream));
    int rc;
    if ((rc = fetch_sticky(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->err_host[0] | h->err_host[1]) return fail(WT_E_HIP, k_incomplete);
    return WT_OK;
}

int wt_ensemble_clear_status(wt_ensemble *h)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemsetAsync(h->status, 0, sizeof(uint32_t) * h->N, h->stream));
    return WT_OK;
}

int wt_ensemble_get_stats(wt_ensemble *h, wt_solver_stats *stats)
{
    if (!h || !stats) return fail(WT_E_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(stats, h->stats, sizeof(int32_t) * 5 * h->N, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WT_OK;
}

int wt_ensemble_rhs(wt_ensemble *h, const double *pH, const double *Cl, const double *T,
                    double *dpH, double *dCl, double *dT, uint32_t *flags)
{
    if (!h || !pH || !Cl || !T || !dpH || !dCl || !dT || !flags) return fail(WT_E_ARG, "NULL argument");
    if (!h->have_bc) return fail(WT_E_STATE, "set_boundary must precede rhs");
    HIP_TRY(hipSetDevice(h->device));
    const size_t cnt = (size_t)h->N * h->n, b = sizeof(double) * cnt;
    double *buf = nullptr; uint32_t *fl = nullptr;
    HIP_TRY(hipMalloc((void **)&buf, 6 * b));
    if (hipMalloc((void **)&fl, sizeof(uint32_t) * h->N) != hipSuccess) { (void)hipFree(buf); return fail(WT_E_HIP, "hipMalloc failed"); }
    hipError_t e = hipMemcpyAsync(buf, pH, b, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(buf + cnt, Cl, b, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(buf + 2 * cnt, T, b, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        wt::RhsArgs a;
        a.N = h->N; a.n = h->n; a.R = h->R; a.par = h->par; a.bc = h->bc;
        a.pH = buf; a.Cl = buf + cnt; a.T = buf + 2 * cnt;
        a.dpH = buf + 3 * cnt; a.dCl = buf + 4 * cnt; a.dT = buf + 5 * cnt; a.flags = fl;
        const unsigned grid = (unsigned)((h->N + h->R - 1) / h->R);
        if (row_mode(h->n)) hipLaunchKernelGGL(wt::rhs_kernel<true>, dim3(grid), dim3(64), 0, h->stream, a);
        else hipLaunchKernelGGL(wt::rhs_kernel<false>, dim3(grid), dim3(64), 0, h->stream, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(dpH, buf + 3 * cnt, b, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dCl, buf + 4 * cnt, b, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dT, buf + 5 * cnt, b, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(flags, fl, sizeof(uint32_t) * h->N, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(buf); (void)hipFree(fl);
    if (e != hipSuccess) return fail(WT_E_HIP, std::string("rhs: ") + hipGetErrorString(e));
    return WT_OK;
}

int wt_ensemble_export_state_device(wt_ensemble *h, void *dst_device)
{
    if (!h || !dst_device) return fail(WT_E_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t cnt = (size_t)h->N * h->n, b = sizeof(double) * cnt;
    double *d = (double *)dst_device;
    HIP_TRY(hipMemcpyAsync(d, h->pH, b, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d + cnt, h->Cl, b, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d + 2 * cnt, h->T, b, hipMemcpyDeviceToDevice, h->stream));
    return WT_OK;
}

int wt_ensemble_timer_start(wt_ensemble *h)
{
    if (!h) return fail(WT_E_ARG, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    return WT_OK;
}

int wt_ensemble_timer_stop(wt_ensemble *h, float *elapsed_ms)
{
    if (!h || !elapsed_ms) return fail(WT_E_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    return WT_OK;
}

int wt_selftest_shuffles(int device, int n_zones, int *mismatches)
{
    if (!mismatches || n_zones < 2 || n_zones > WT_MAX_ZONES) return fail(WT_E_ARG, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(WT_E_NOGPU, "no HIP device available: libwtphys has no CPU path");
    HIP_TRY(hipSetDevice(device));
    int *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 64 * sizeof(int)));
    wt::ShuffleTestArgs a{n_zones, d};
    if (row_mode(n_zones)) hipLaunchKernelGGL(wt::shuffle_selftest_kernel<true>, dim3(1), dim3(64), 0, 0, a);
    else hipLaunchKernelGGL(wt::shuffle_selftest_kernel<false>, dim3(1), dim3(64), 0, 0, a);
    int host[64];
    hipError_t e = hipMemcpy(host, d, sizeof host, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(WT_E_HIP, hipGetErrorString(e));
    int total = 0;
    for (int i = 0; i < 64; ++i) total += host[i];
    *mismatches = total;
    return WT_OK;
}

int wt_wave_diag_slots(void) { return wt::WT_DIAG_SLOTS; }

int wt_ensemble_wave_diag(wt_ensemble *h, int64_t *out, int64_t capacity, int64_t *n_waves)
{
    if (!h || !n_waves) return fail(WT_E_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    const int64_t nw = h->n_groups;
    *n_waves = nw;
    if (!h->wave_diag) {   // first call switches the diagnostics on
        HIP_TRY(hipMalloc((void **)&h->wave_diag, sizeof(int64_t) * wt::WT_DIAG_SLOTS * (size_t)nw));
        HIP_TRY(hipMemsetAsync(h->wave_diag, 0, sizeof(int64_t) * wt::WT_DIAG_SLOTS * (size_t)nw, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        return WT_OK;
    }
    if (out) {
        if (capacity < nw) return fail(WT_E_ARG, "wave_diag buffer too small");
        HIP_TRY(hipMemcpyAsync(out, h->wave_diag, sizeof(int64_t) * wt::WT_DIAG_SLOTS * (size_t)nw, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return WT_OK;
}

int64_t wt_ensemble_size(const wt_ensemble *h) { return h ? h->N : 0; }
int wt_ensemble_zones(const wt_ensemble *h) { return h ? h->n : 0; }

int wt_ph_solve(int device, int64_t n, const double *Kw, const double *Ka1, const double *Ka2,
                const double *CT_mol, const double *alk_mgL, const double *guess,
                double tol, int max_iter, double *pH_out, int32_t *iters, int32_t *rc)
{
    if (n < 0 || !Kw || !Ka1 || !Ka2 || !CT_mol || !alk_mgL || !guess || !pH_out || !iters || !rc)
        return fail(WT_E_ARG, "NULL argument");
    if (n == 0) return WT_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(WT_E_NOGPU, "no HIP device available: libwtphys has no CPU path");
    if (device < 0 || device >= ndev) return fail(WT_E_ARG, "bad device index");
    HIP_TRY(hipSetDevice(device));
    const size_t b = sizeof(double) * (size_t)n;
    double *buf = nullptr; int32_t *ib = nullptr;
    HIP_TRY(hipMalloc((void **)&buf, 7 * b));
    if (hipMalloc((void **)&ib, 2 * sizeof(int32_t) * (size_t)n) != hipSuccess) { (void)hipFree(buf); return fail(WT_E_HIP, "hipMalloc failed"); }
    const double *src[6] = {Kw, Ka1, Ka2, CT_mol, alk_mgL, guess};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipMemcpy(buf + (size_t)i * n, src[i], b, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        wt::PhArgs a;
        a.n = n; a.Kw = buf; a.Ka1 = buf + n; a.Ka2 = buf + 2 * n; a.CT = buf + 3 * n; a.alk = buf + 4 * n;
        a.guess = buf + 5 * n; a.tol = tol; a.max_iter = max_iter; a.pH = buf + 6 * n; a.iters = ib; a.rc = ib + n;
        const unsigned grid = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(wt::ph_solve_kernel, dim3(grid), dim3(256), 0, 0, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(pH_out, buf + 6 * (size_t)n, b, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(iters, ib, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(rc, ib + n, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost);
    (void)hipFree(buf); (void)hipFree(ib);
    if (e != hipSuccess) return fail(WT_E_HIP, std::string("ph_solve: ") + hipGetErrorString(e));
    return WT_OK;
}

} // extern "C"
