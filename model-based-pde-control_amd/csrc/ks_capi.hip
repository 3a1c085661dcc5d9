// ks_capi.hip -- the C ABI of libkspde.so (declared in include/kspde.h).
//
// Host-side only: handle bookkeeping, buffer ownership, layout choice and launches.  device >= 0 names a HIP device
// (without one every call fails with KS_ERR_NO_DEVICE / KS_ERR_HIP: nothing falls back silently); device = -1 is an
// explicit request for the CPU twin (ks_cpu.cpp: the same arithmetic on host memory, BASELINE configs[0]).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/kspde.h"
#include "ks_cpu.h"
#include "ks_internal.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define KS_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(KS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                \
    } while (0)

}  // namespace

struct ks_handle {
    int device = 0;
    bool cpu = false;           // device == -1: every "device" buffer below is host memory, launches run in ks_cpu.cpp
    int cpu_threads = 1;
    int E = 0, N = 0;
    double L = 0, dt = 0, dx = 0;
    int mode = KS_MODE_FAST;
    int variant = KS_VARIANT_AUTO;
    int block_threads = 0;
    int n_act = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // device buffers
    double* d_u = nullptr;      // [E,N]
    float* d_phi = nullptr;     // [E,N] staging for host phi
    float* d_act = nullptr;     // [E,n_act]
    float* d_F = nullptr;       // [n_act,N]
    // step outputs: ONE device block [ obs E*N f32 | ssq E f64 | status E i32 ] and a pinned host mirror of it, so that the
    // host-boundary entries (ks_step / ks_step_actions / ks_step_rows) fetch everything with one asynchronous copy
    // instead of three (or 3 per row) pageable ones
    char* d_out = nullptr;
    char* h_out = nullptr;      // pinned
    size_t out_bytes = 0, off_ssq = 0, off_status = 0;
    float* h_act = nullptr;     // pinned staging of the actions, [E,n_act]
    float* d_obs = nullptr;     // [E,N]   (inside d_out)
    double* d_ssq = nullptr;    // [E]     (inside d_out)
    int* d_status = nullptr;    // [E]     (inside d_out)
    int* d_ids = nullptr;       // [E]
    int* h_ids = nullptr;       // pinned [E]: the env list of the step in flight (ks_step_begin ... ks_step_end)
    // split host-boundary step in flight
    bool pending = false, pend_obs = false, pend_rows_only = false;
    int pend_n = 0;             // 0: all envs; > 0: the first pend_n entries of h_ids
    double* d_rows = nullptr;   // [E,N] staging for ks_set_state_rows / ks_rhs
    unsigned* d_flag = nullptr; // selftest
    int num_cus = 256;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (dev < 0) {   // CPU twin: no HIP call at all
            ok = true;
            return;
        }
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// memory and ordering primitives of a handle: HIP on a device handle, the C library on the CPU twin
hipError_t dev_alloc(const ks_handle* h, void** p, size_t bytes) {
    if (!h->cpu) return hipMalloc(p, bytes);
    *p = std::calloc(1, bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
void dev_free(const ks_handle* h, void* p) {
    if (!p) return;
    if (h->cpu)
        std::free(p);
    else
        (void)hipFree(p);
}
hipError_t pinned_alloc(const ks_handle* h, void** p, size_t bytes) {
    if (!h->cpu) return hipHostMalloc(p, bytes, hipHostMallocDefault);
    *p = std::calloc(1, bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
void pinned_free(const ks_handle* h, void* p) {
    if (!p) return;
    if (h->cpu)
        std::free(p);
    else
        (void)hipHostFree(p);
}
hipError_t copy_async(const ks_handle* h, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (!h->cpu) return hipMemcpyAsync(dst, src, bytes, kind, h->stream);
    if (bytes) std::memcpy(dst, src, bytes);
    return hipSuccess;
}
hipError_t stream_sync(const ks_handle* h) { return h->cpu ? hipSuccess : hipStreamSynchronize(h->stream); }

void fill_args(const ks_handle* h, ks::StepArgs& a) {
    std::memset(&a, 0, sizeof(a));
    const double dx = h->dx, dt = h->dt;
    a.N = h->N;
    a.n_act = h->n_act;
    a.F = h->d_F;
    a.dt = dt;
    a.dx = dx;
    a.dx2 = dx * dx;          // python: self.dx**2
    a.dx4 = std::pow(dx, 4.0);  // python: self.dx**4
    a.r_dx = 1.0 / a.dx;        // IEEE divisions on the host: correctly rounded reciprocals
    a.r_dx2 = 1.0 / a.dx2;
    a.r_dx4 = 1.0 / a.dx4;
    // merged linear stencil  -(u_xxxx + u_xx):  c_k = -(D4_k/dx^4 + D2_k/dx^2)
    const double d2[5] = {-49.0 / 18, 3.0 / 2, -3.0 / 20, 1.0 / 90, 0.0};
    const double d4[5] = {91.0 / 8, -122.0 / 15, 169.0 / 60, -2.0 / 5, 7.0 / 240};
    for (int k = 0; k < 5; ++k) a.c_lin[k] = -(d4[k] / a.dx4 + d2[k] / a.dx2);
    a.mh_inv_dx = -0.5 / dx;
    a.hdt = dt / 2.0;
    a.dt6 = dt / 6.0;
    a.dt3 = dt / 3.0;
}

// Choose (variant, G, P, block, grid) for n_rows envs.
int choose_layout(const ks_handle* h, int n_rows, ks::Layout& lay) {
    const int N = h->N;
    int variant = h->variant;
    if (variant == KS_VARIANT_AUTO) {
        // Prefer 16-lane groups (cheapest halo: one DPP row rotation per halo value, most points
        // per lane) when that still yields >= 1 wave per SIMD (4 * CUs waves); otherwise spread
        // each env over a full wavefront to expose more waves.
        const long waves16 = ((long)n_rows * 16 + 63) / 64;
        const bool ok16 = ks::layout_supported(KS_VARIANT_ROW16_DPP, N);
        const bool ok64 = ks::layout_supported(KS_VARIANT_WAVE64_DPP, N);
        if (ok16 && (waves16 >= 4L * h->num_cus || !ok64))
            variant = KS_VARIANT_ROW16_DPP;
        else if (ok64)
            variant = KS_VARIANT_WAVE64_DPP;
        else if (ks::layout_supported(KS_VARIANT_HALF32_BPERM, N))
            variant = KS_VARIANT_HALF32_BPERM;
        else
            variant = KS_VARIANT_LDS;
    }
    if (!ks::layout_supported(variant, N))
        return fail(KS_ERR_UNSUPPORTED, "kernel variant %d has no instantiation for N=%d", variant, N);
    lay.variant = variant;
    lay.lds_bytes = 0;
    if (variant == KS_VARIANT_LDS) {
        int block = h->block_threads ? h->block_threads : (N <= 64 ? 64 : (N <= 128 ? 128 : 256));
        lay.G = 0;
        lay.P = 0;
        lay.block = block;
        lay.grid = n_rows;
        lay.lds_bytes = sizeof(double) * (5 * (size_t)N + block / 64);
        return KS_OK;
    }
    lay.G = (variant == KS_VARIANT_ROW16_DPP || variant == KS_VARIANT_ROW16_BPERM) ? 16
            : (variant == KS_VARIANT_HALF32_BPERM)                                  ? 32
                                                                                    : 64;
    lay.P = N / lay.G;
    const int epw = 64 / lay.G;
    const long waves = ((long)n_rows + epw - 1) / epw;
    int block = h->block_threads;
    if (!block) {
        // one wave per SIMD per workgroup when there are enough waves to give every CU a full
        // workgroup; single-wave workgroups otherwise so the dispatcher can spread them
        block = (waves >= 4L * h->num_cus) ? 256 : 64;
    }
    const int wpb = block / 64;
    lay.block = block;
    lay.grid = (int)((waves + wpb - 1) / wpb);
    return KS_OK;
}

int do_step(ks_handle* h, const float* d_phi, const float* d_actions, const int* d_env_ids, int n_rows,
            long n_substeps, float* d_obs, double* d_ssq, int* d_status) {
    if (n_substeps < 0) return fail(KS_ERR_INVALID, "n_substeps < 0");
    if (d_phi && d_actions) return fail(KS_ERR_INVALID, "give phi or actions, not both");
    if (d_actions && (!h->d_F || h->n_act <= 0))
        return fail(KS_ERR_INVALID, "ks_step_actions needs ks_set_forcing first");
    const int rows = d_env_ids ? n_rows : h->E;
    if (rows < 0 || rows > h->E) return fail(KS_ERR_INVALID, "n_rows out of range");
    ks::StepArgs a;
    fill_args(h, a);
    if (h->cpu) {
        kscpu::Params p{};
        p.N = h->N;
        p.dt = a.dt;
        p.dx = a.dx;
        p.dx2 = a.dx2;
        p.dx4 = a.dx4;
        for (int k = 0; k < 5; ++k) p.c_lin[k] = a.c_lin[k];
        p.mh_inv_dx = a.mh_inv_dx;
        p.hdt = a.hdt;
        p.dt6 = a.dt6;
        p.dt3 = a.dt3;
        kscpu::step(p, h->mode, h->d_u, d_phi, d_actions, h->d_F, h->n_act, d_env_ids, rows, n_substeps, d_obs, d_ssq,
                    d_status, h->cpu_threads);
        return KS_OK;
    }
    ks::Layout lay;
    int rc = choose_layout(h, rows, lay);
    if (rc != KS_OK) return rc;
    a.u = h->d_u;
    a.phi = d_phi;
    a.actions = d_actions;
    a.env_ids = d_env_ids;
    a.obs = d_obs;
    a.ssq_sum = d_ssq;
    a.status = d_status;
    a.n_rows = rows;
    a.n_substeps = n_substeps;
    KS_HIP(ks::launch_step(lay, h->mode, a, h->stream));
    return KS_OK;
}

}  // namespace

extern "C" {

const char* ks_last_error(void) { return g_err; }
const char* ks_version(void) { return "kspde 0.2 (gfx950 + CPU twin)"; }

int ks_create(int device, int num_envs, int N, double L, double dt, ks_handle** out) {
    if (!out) return fail(KS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (num_envs <= 0 || N < 9 || !(L > 0) || !(dt > 0))
        return fail(KS_ERR_INVALID, "need num_envs > 0, N >= 9, L > 0, dt > 0 (got %d, %d, %g, %g)", num_envs, N,
                    L, dt);
    if ((size_t)num_envs * (size_t)N > (size_t)1 << 31)
        return fail(KS_ERR_INVALID, "num_envs * N too large");
    const bool cpu = device == -1;
    if (!cpu) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            return fail(KS_ERR_NO_DEVICE, "no HIP device available (device = -1 selects the CPU twin explicitly; nothing "
                                          "falls back to it)");
        if (device < 0 || device >= count)
            return fail(KS_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, count);
    }
    DeviceGuard g(device);
    if (!g.ok) return fail(KS_ERR_HIP, "hipSetDevice(%d) failed", device);
    ks_handle* h = new (std::nothrow) ks_handle();
    if (!h) return fail(KS_ERR_INVALID, "out of host memory");
    h->device = device;
    h->cpu = cpu;
    h->cpu_threads = cpu ? kscpu::default_threads() : 1;
    h->E = num_envs;
    h->N = N;
    h->L = L;
    h->dt = dt;
    h->dx = L / N;  // kuramoto.py:55
    hipDeviceProp_t prop;
    if (!cpu && hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        h->num_cus = prop.multiProcessorCount;
    const size_t en = (size_t)num_envs * N;
#define KS_ALLOC(ptr, bytes)                                                        \
    do {                                                                            \
        hipError_t e_ = dev_alloc(h, (void**)&(ptr), (bytes));                      \
        if (e_ != hipSuccess) {                                                     \
            ks_destroy(h);                                                          \
            return fail(KS_ERR_HIP, "hipMalloc(%zu) failed: %s", (size_t)(bytes),   \
                        hipGetErrorString(e_));                                     \
        }                                                                           \
    } while (0)
    KS_ALLOC(h->d_u, en * sizeof(double));
    KS_ALLOC(h->d_phi, en * sizeof(float));
    h->off_ssq = (en * sizeof(float) + 255) / 256 * 256;
    h->off_status = h->off_ssq + (size_t)num_envs * sizeof(double);
    h->out_bytes = h->off_status + (size_t)num_envs * sizeof(int);
    KS_ALLOC(h->d_out, h->out_bytes);
    h->d_obs = reinterpret_cast<float*>(h->d_out);
    h->d_ssq = reinterpret_cast<double*>(h->d_out + h->off_ssq);
    h->d_status = reinterpret_cast<int*>(h->d_out + h->off_status);
    if (pinned_alloc(h, (void**)&h->h_out, h->out_bytes) != hipSuccess ||
        pinned_alloc(h, (void**)&h->h_ids, (size_t)num_envs * sizeof(int)) != hipSuccess) {
        ks_destroy(h);
        return fail(KS_ERR_HIP, "pinned host allocation (%zu bytes) failed", h->out_bytes);
    }
    KS_ALLOC(h->d_rows, en * sizeof(double));
    KS_ALLOC(h->d_ids, num_envs * sizeof(int));
    KS_ALLOC(h->d_flag, sizeof(unsigned));
#undef KS_ALLOC
    if (!cpu) {   // (the CPU twin's buffers come zeroed from calloc)
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
            ks_destroy(h);
            return fail(KS_ERR_HIP, "hipStreamCreate failed");
        }
        h->own_stream = true;
        if (hipMemsetAsync(h->d_u, 0, en * sizeof(double), h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) {
            ks_destroy(h);
            return fail(KS_ERR_HIP, "initial memset failed");
        }
    }
    *out = h;
    return KS_OK;
}

int ks_destroy(ks_handle* h) {
    if (!h) return KS_OK;
    DeviceGuard g(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* bufs[] = {h->d_u, h->d_phi, h->d_act, h->d_F, h->d_out, h->d_ids, h->d_rows, h->d_flag};
    for (void* b : bufs) dev_free(h, b);
    pinned_free(h, h->h_out);
    pinned_free(h, h->h_act);
    pinned_free(h, h->h_ids);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return KS_OK;
}

int ks_set_stream(ks_handle* h, void* hip_stream) {
    if (!h) return fail(KS_ERR_INVALID, "NULL handle");
    if (h->cpu) return fail(KS_ERR_UNSUPPORTED, "the CPU twin has no stream");
    DeviceGuard g(h->device);
    if (h->stream) KS_HIP(hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    return KS_OK;
}

int ks_set_mode(ks_handle* h, int mode) {
    if (!h) return fail(KS_ERR_INVALID, "NULL handle");
    if (mode != KS_MODE_FAST && mode != KS_MODE_EXACT) return fail(KS_ERR_INVALID, "unknown mode %d", mode);
    h->mode = mode;
    return KS_OK;
}

int ks_set_variant(ks_handle* h, int variant) {
    if (!h) return fail(KS_ERR_INVALID, "NULL handle");
    if (h->cpu && variant != KS_VARIANT_AUTO) return fail(KS_ERR_UNSUPPORTED, "the CPU twin has one layout");
    if (variant != KS_VARIANT_AUTO && !ks::layout_supported(variant, h->N))
        return fail(KS_ERR_UNSUPPORTED, "variant %d not available for N=%d", variant, h->N);
    h->variant = variant;
    return KS_OK;
}

int ks_set_block_size(ks_handle* h, int threads) {
    if (!h) return fail(KS_ERR_INVALID, "NULL handle");
    if (threads != 0 && threads != 64 && threads != 128 && threads != 256)
        return fail(KS_ERR_INVALID, "block size must be 0, 64, 128 or 256");
    h->block_threads = threads;
    return KS_OK;
}

int ks_get_layout(ks_handle* h, int* variant, int* lanes_per_env, int* points_per_lane, int* block_threads,
                  int* grid_blocks) {
    if (!h) return fail(KS_ERR_INVALID, "NULL handle");
    ks::Layout lay;
    if (h->cpu) {   // one env per host thread at a time: reported as variant 0, "block" = host threads
        lay.variant = KS_VARIANT_AUTO;
        lay.G = 0;
        lay.P = h->N;
        lay.block = h->cpu_threads;
        lay.grid = 1;
    } else {
        int rc = choose_layout(h, h->E, lay);
        if (rc != KS_OK) return rc;
    }
    if (variant) *variant = lay.variant;
    if (lanes_per_env) *lanes_per_env = lay.G;
    if (points_per_lane) *points_per_lane = lay.P;
    if (block_threads) *block_threads = lay.block;
    if (grid_blocks) *grid_blocks = lay.grid;
    return KS_OK;
}

int ks_set_forcing(ks_handle* h, const float* F_host, int n_act) {
    if (!h || !F_host) return fail(KS_ERR_INVALID, "NULL argument");
    if (n_act <= 0 || n_act > 64) return fail(KS_ERR_INVALID, "n_act out of range");
    if (h->pending) return fail(KS_ERR_INVALID, "a step is in flight (ks_step_begin without ks_step_end)");
    DeviceGuard g(h->device);
    KS_HIP(stream_sync(h));
    dev_free(h, h->d_F);
    dev_free(h, h->d_act);
    pinned_free(h, h->h_act);
    h->d_F = nullptr;
    h->d_act = nullptr;
    h->h_act = nullptr;
    h->n_act = 0;
    // all three or none: a handle with d_F set but no staging buffer would pass ks_step_actions' check and then copy
    // through a NULL pointer
    hipError_t e = dev_alloc(h, (void**)&h->d_F, sizeof(float) * (size_t)n_act * h->N);
    if (e == hipSuccess) e = dev_alloc(h, (void**)&h->d_act, sizeof(float) * (size_t)n_act * h->E);
    if (e == hipSuccess) e = pinned_alloc(h, (void**)&h->h_act, sizeof(float) * (size_t)n_act * h->E);
    if (e == hipSuccess) e = copy_async(h, h->d_F, F_host, sizeof(float) * (size_t)n_act * h->N, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = stream_sync(h);
    if (e != hipSuccess) {
        dev_free(h, h->d_F);
        dev_free(h, h->d_act);
        pinned_free(h, h->h_act);
        h->d_F = nullptr;
        h->d_act = nullptr;
        h->h_act = nullptr;
        return fail(KS_ERR_HIP, "ks_set_forcing: %s", hipGetErrorString(e));
    }
    h->n_act = n_act;
    return KS_OK;
}

int ks_set_state(ks_handle* h, const double* u_host) {
    if (!h || !u_host) return fail(KS_ERR_INVALID, "NULL argument");
    DeviceGuard g(h->device);
    KS_HIP(copy_async(h, h->d_u, u_host, sizeof(double) * (size_t)h->E * h->N, hipMemcpyHostToDevice));
    KS_HIP(stream_sync(h));
    return KS_OK;
}

int ks_get_state(ks_handle* h, double* u_host) {
    if (!h || !u_host) return fail(KS_ERR_INVALID, "NULL argument");
    DeviceGuard g(h->device);
    KS_HIP(copy_async(h, u_host, h->d_u, sizeof(double) * (size_t)h->E * h->N, hipMemcpyDeviceToHost));
    KS_HIP(stream_sync(h));
    return KS_OK;
}

int ks_set_state_rows(ks_handle* h, const int* env_ids_host, int n, const double* u_host) {
    if (!h || (n > 0 && (!env_ids_host || !u_host))) return fail(KS_ERR_INVALID, "NULL argument");
    if (n < 0 || n > h->E) return fail(KS_ERR_INVALID, "n out of range");
    for (int i = 0; i < n; ++i)
        if (env_ids_host[i] < 0 || env_ids_host[i] >= h->E)
            return fail(KS_ERR_INVALID, "env id %d out of range", env_ids_host[i]);
    DeviceGuard g(h->device);
    // one copy per run of consecutive env ids (a reset of every env, or of a shard's whole block, is ONE copy instead of
    // thousands of row-sized ones: 4096 rows cost 20 ms of host time that way)
    const size_t row = sizeof(double) * (size_t)h->N;
    for (int i = 0; i < n;) {
        int j = i + 1;
        while (j < n && env_ids_host[j] == env_ids_host[j - 1] + 1) ++j;
        KS_HIP(copy_async(h, h->d_u + (size_t)env_ids_host[i] * h->N, u_host + (size_t)i * h->N, row * (size_t)(j - i),
                          hipMemcpyHostToDevice));
        i = j;
    }
    KS_HIP(stream_sync(h));
    return KS_OK;
}

int ks_state_device_ptr(ks_handle* h, double** d_u) {
    if (!h || !d_u) return fail(KS_ERR_INVALID, "NULL argument");
    *d_u = h->d_u;
    return KS_OK;
}

constexpr size_t PINNED_OBS_LIMIT = 512 * 1024;
// a listed-rows step fetches single observation rows (one small copy each) up to this many rows; beyond it the whole
// block is one copy (4096 x 256: 4.2 MB, ~0.2 ms -- not something to pay for an autoreset of one env)
constexpr int ROWWISE_FETCH_MAX = 32;

// the step outputs of ALL envs into the pinned mirror: one copy of the whole block, or of its [ssq | status] tail only
static int fetch_outputs_async(ks_handle* h, bool with_obs) {
    const size_t from = with_obs ? 0 : h->off_ssq;
    KS_HIP(copy_async(h, h->h_out + from, h->d_out + from, h->out_bytes - from, hipMemcpyDeviceToHost));
    return KS_OK;
}

// the outputs of the n listed envs into the pinned mirror (at their env-id positions): the small tail whole, the
// observation rows one by one while they are few
static int fetch_rows_async(ks_handle* h, const int* ids, int n, bool with_obs) {
    if (with_obs && n > ROWWISE_FETCH_MAX) return fetch_outputs_async(h, true);
    if (int rc = fetch_outputs_async(h, false)) return rc;
    if (with_obs) {
        const size_t row = sizeof(float) * (size_t)h->N;
        for (int i = 0; i < n; ++i)
            KS_HIP(copy_async(h, h->h_out + (size_t)ids[i] * row, h->d_out + (size_t)ids[i] * row, row,
                              hipMemcpyDeviceToHost));
    }
    return KS_OK;
}

// mirror -> caller, in list order
static void hand_over_rows(const ks_handle* h, const int* ids, int n, float* obs_f32, double* ssq_sum, int* status) {
    const size_t row = sizeof(float) * (size_t)h->N;
    const float* hobs = reinterpret_cast<const float*>(h->h_out);
    const double* hssq = reinterpret_cast<const double*>(h->h_out + h->off_ssq);
    const int* hst = reinterpret_cast<const int*>(h->h_out + h->off_status);
    for (int i = 0; i < n; ++i) {
        const int e = ids[i];
        if (obs_f32) memcpy(obs_f32 + (size_t)i * h->N, hobs + (size_t)e * h->N, row);
        if (ssq_sum) ssq_sum[i] = hssq[e];
        if (status) status[i] = hst[e];
    }
}

static int check_ids(const ks_handle* h, const int* ids, int n) {
    for (int i = 0; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= h->E) return fail(KS_ERR_INVALID, "env id %d out of range", ids[i]);
    return KS_OK;
}

static int step_common(ks_handle* h, const float* d_phi, const float* d_act, long n_substeps, float* obs_f32,
                       double* ssq_sum, int* status) {
    int rc = do_step(h, d_phi, d_act, nullptr, 0, n_substeps, obs_f32 ? h->d_obs : nullptr,
                     ssq_sum ? h->d_ssq : nullptr, status ? h->d_status : nullptr);
    if (rc != KS_OK) return rc;
    // small observation blocks ride in the pinned mirror with the reward sums (one copy, one sync: 0.20 -> 0.16 ms per
    // step at 1024 x 64); large ones go straight into the caller's buffer -- the extra host memcpy of 4 MB costs more
    // than the runtime's own staged pageable copy (4096 x 256: 1.00 vs 1.12 ms)
    const size_t obs_bytes = sizeof(float) * (size_t)h->E * h->N;
    const bool direct_obs = obs_f32 && obs_bytes > PINNED_OBS_LIMIT;
    if (direct_obs) KS_HIP(copy_async(h, obs_f32, h->d_obs, obs_bytes, hipMemcpyDeviceToHost));
    if (int rc2 = fetch_outputs_async(h, obs_f32 != nullptr && !direct_obs)) return rc2;
    KS_HIP(stream_sync(h));
    if (obs_f32 && !direct_obs) memcpy(obs_f32, h->h_out, obs_bytes);
    if (ssq_sum) memcpy(ssq_sum, h->h_out + h->off_ssq, sizeof(double) * h->E);
    if (status) memcpy(status, h->h_out + h->off_status, sizeof(int) * h->E);
    return KS_OK;
}

#define KS_NOT_PENDING(h) \
    if ((h)->pending) return fail(KS_ERR_INVALID, "a step is in flight (ks_step_begin without ks_step_end)")

int ks_step(ks_handle* h, const float* phi_host, long n_substeps, float* obs_f32, double* ssq_sum, int* status) {
    if (!h) return fail(KS_ERR_INVALID, "NULL handle");
    KS_NOT_PENDING(h);
    DeviceGuard g(h->device);
    const float* d_phi = nullptr;
    if (phi_host) {
        KS_HIP(copy_async(h, h->d_phi, phi_host, sizeof(float) * (size_t)h->E * h->N, hipMemcpyHostToDevice));
        d_phi = h->d_phi;
    }
    return step_common(h, d_phi, nullptr, n_substeps, obs_f32, ssq_sum, status);
}

int ks_step_actions(ks_handle* h, const float* actions_host, long n_substeps, float* obs_f32, double* ssq_sum,
                    int* status) {
    if (!h || !actions_host) return fail(KS_ERR_INVALID, "NULL argument");
    if (!h->d_F || !h->h_act) return fail(KS_ERR_INVALID, "ks_step_actions needs ks_set_forcing first");
    KS_NOT_PENDING(h);
    DeviceGuard g(h->device);
    const size_t abytes = sizeof(float) * (size_t)h->E * h->n_act;
    memcpy(h->h_act, actions_host, abytes);     // the previous step's copy has completed: every entry synchronises
    KS_HIP(copy_async(h, h->d_act, h->h_act, abytes, hipMemcpyHostToDevice));
    return step_common(h, nullptr, h->d_act, n_substeps, obs_f32, ssq_sum, status);
}

int ks_step_rows(ks_handle* h, const int* env_ids_host, int n, long n_substeps, float* obs_f32, double* ssq_sum,
                 int* status) {
    if (!h || (n > 0 && !env_ids_host)) return fail(KS_ERR_INVALID, "NULL argument");
    if (n < 0 || n > h->E) return fail(KS_ERR_INVALID, "n out of range");
    if (n == 0) return KS_OK;
    if (int rc = check_ids(h, env_ids_host, n)) return rc;
    KS_NOT_PENDING(h);
    DeviceGuard g(h->device);
    KS_HIP(copy_async(h, h->d_ids, env_ids_host, sizeof(int) * n, hipMemcpyHostToDevice));
    int rc = do_step(h, nullptr, nullptr, h->d_ids, n, n_substeps, obs_f32 ? h->d_obs : nullptr,
                     ssq_sum ? h->d_ssq : nullptr, status ? h->d_status : nullptr);
    if (rc != KS_OK) return rc;
    // outputs are indexed by env id on the device; the listed rows come back in list order
    if (int rc2 = fetch_rows_async(h, env_ids_host, n, obs_f32 != nullptr)) return rc2;
    KS_HIP(stream_sync(h));
    hand_over_rows(h, env_ids_host, n, obs_f32, ssq_sum, status);
    return KS_OK;
}

// ---- split host-boundary step (one handle per device driven from one host thread: begin on every handle, then end) ----

int ks_step_begin(ks_handle* h, const float* actions_host, const int* env_ids_host, int n_rows, long n_substeps,
                  int want_obs) {
    if (!h) return fail(KS_ERR_INVALID, "NULL handle");
    KS_NOT_PENDING(h);
    if (actions_host && (!h->d_F || !h->h_act)) return fail(KS_ERR_INVALID, "actions need ks_set_forcing first");
    if (env_ids_host) {
        if (n_rows < 0 || n_rows > h->E) return fail(KS_ERR_INVALID, "n_rows out of range");
        if (int rc = check_ids(h, env_ids_host, n_rows)) return rc;
    }
    DeviceGuard g(h->device);
    const float* d_act = nullptr;
    if (actions_host) {
        const size_t abytes = sizeof(float) * (size_t)h->E * h->n_act;
        memcpy(h->h_act, actions_host, abytes);   // pinned staging: the copy below returns at once
        KS_HIP(copy_async(h, h->d_act, h->h_act, abytes, hipMemcpyHostToDevice));
        d_act = h->d_act;
    }
    const int* d_ids = nullptr;
    const int n = env_ids_host ? n_rows : 0;
    if (env_ids_host && n > 0) {
        memcpy(h->h_ids, env_ids_host, sizeof(int) * n);
        KS_HIP(copy_async(h, h->d_ids, h->h_ids, sizeof(int) * n, hipMemcpyHostToDevice));
        d_ids = h->d_ids;
    }
    h->pend_rows_only = env_ids_host != nullptr;
    h->pend_n = n;
    h->pend_obs = want_obs != 0;
    if (!(h->pend_rows_only && n == 0)) {
        int rc = do_step(h, nullptr, d_act, d_ids, n, n_substeps, want_obs ? h->d_obs : nullptr, h->d_ssq, h->d_status);
        if (rc != KS_OK) return rc;
        rc = h->pend_rows_only ? fetch_rows_async(h, h->h_ids, n, h->pend_obs) : fetch_outputs_async(h, h->pend_obs);
        if (rc != KS_OK) return rc;
    }
    h->pending = true;
    return KS_OK;
}

int ks_step_end(ks_handle* h, float* obs_f32, double* ssq_sum, int* status) {
    if (!h) return fail(KS_ERR_INVALID, "NULL handle");
    if (!h->pending) return fail(KS_ERR_INVALID, "ks_step_end without ks_step_begin");
    if (obs_f32 && !h->pend_obs) return fail(KS_ERR_INVALID, "ks_step_begin was told not to fetch observations");
    DeviceGuard g(h->device);
    h->pending = false;
    KS_HIP(stream_sync(h));
    if (h->pend_rows_only) {
        hand_over_rows(h, h->h_ids, h->pend_n, obs_f32, ssq_sum, status);
    } else {
        if (obs_f32) memcpy(obs_f32, h->h_out, sizeof(float) * (size_t)h->E * h->N);
        if (ssq_sum) memcpy(ssq_sum, h->h_out + h->off_ssq, sizeof(double) * h->E);
        if (status) memcpy(status, h->h_out + h->off_status, sizeof(int) * h->E);
    }
    return KS_OK;
}

int ks_step_device(ks_handle* h, const float* d_phi, const float* d_actions, const int* d_env_ids, int n_rows,
                   long n_substeps, float* d_obs_f32, double* d_ssq_sum, int* d_status) {
    if (!h) return fail(KS_ERR_INVALID, "NULL handle");
    KS_NOT_PENDING(h);
    DeviceGuard g(h->device);   // (on the CPU twin "device pointers" are host pointers and the call is synchronous)
    return do_step(h, d_phi, d_actions, d_env_ids, n_rows, n_substeps, d_obs_f32, d_ssq_sum, d_status);
}

int ks_sync(ks_handle* h) {
    if (!h) return fail(KS_ERR_INVALID, "NULL handle");
    DeviceGuard g(h->device);
    KS_HIP(stream_sync(h));
    return KS_OK;
}

int ks_rhs(ks_handle* h, const double* u_host, const float* phi_host, int n_rows, double* rhs, double* ux,
           double* uxx, double* uxxxx) {
    if (!h || !u_host || !phi_host || !rhs) return fail(KS_ERR_INVALID, "NULL argument");
    if (n_rows <= 0) return fail(KS_ERR_INVALID, "n_rows <= 0");
    if (h->cpu) {
        const double dx = h->dx;
        kscpu::rhs(h->N, dx, dx * dx, std::pow(dx, 4.0), u_host, phi_host, n_rows, rhs, ux, uxx, uxxxx);
        return KS_OK;
    }
    DeviceGuard g(h->device);
    const size_t n = (size_t)n_rows * h->N;
    double *d_u = nullptr, *d_out = nullptr;
    float* d_phi = nullptr;
    KS_HIP(hipMalloc((void**)&d_u, n * sizeof(double)));
    hipError_t e1 = hipMalloc((void**)&d_out, 4 * n * sizeof(double));
    hipError_t e2 = hipMalloc((void**)&d_phi, n * sizeof(float));
    int rc = KS_OK;
    if (e1 != hipSuccess || e2 != hipSuccess) {
        rc = fail(KS_ERR_HIP, "hipMalloc failed in ks_rhs");
    } else {
        hipError_t e = hipMemcpyAsync(d_u, u_host, n * sizeof(double), hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_phi, phi_host, n * sizeof(float), hipMemcpyHostToDevice, h->stream);
        const double dx = h->dx;
        if (e == hipSuccess)
            e = ks::launch_rhs(d_u, d_phi, n_rows, h->N, dx, dx * dx, std::pow(dx, 4.0), d_out, d_out + n,
                               d_out + 2 * n, d_out + 3 * n, h->stream);
        double* outs[4] = {rhs, ux, uxx, uxxxx};
        for (int k = 0; k < 4 && e == hipSuccess; ++k)
            if (outs[k]) e = hipMemcpyAsync(outs[k], d_out + k * n, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(KS_ERR_HIP, "ks_rhs: %s", hipGetErrorString(e));
    }
    if (d_u) (void)hipFree(d_u);
    if (d_out) (void)hipFree(d_out);
    if (d_phi) (void)hipFree(d_phi);
    return rc;
}

int ks_selftest(ks_handle* h, unsigned* failed_mask) {
    if (!h) return fail(KS_ERR_INVALID, "NULL handle");
    if (h->cpu) {   // no cross-lane primitive exists on the CPU twin
        if (failed_mask) *failed_mask = 0;
        return KS_OK;
    }
    DeviceGuard g(h->device);
    unsigned m = 0;
    KS_HIP(hipMemsetAsync(h->d_flag, 0, sizeof(unsigned), h->stream));
    KS_HIP(ks::launch_selftest(h->d_flag, h->stream));
    KS_HIP(hipMemcpyAsync(&m, h->d_flag, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    KS_HIP(hipStreamSynchronize(h->stream));
    if (failed_mask) *failed_mask = m;
    if (m) return fail(KS_ERR_SELFTEST, "cross-lane self test failed, variant mask 0x%x", m);
    return KS_OK;
}

}  // extern "C"
