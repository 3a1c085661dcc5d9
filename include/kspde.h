/*
 * kspde.h -- C ABI of libkspde.so: the MI355X (gfx950) batched Kuramoto-Sivashinsky stepper.
 *
 * This is the drop-in boundary for the reference's KS hot path.  The reference
 * (stwerner97/model-based-pde-control) is pure Python and has no FFI of its own; each entry
 * point below names the reference interface it replaces (paths relative to the reference
 * root).  The Python side (model-based-pde-control_amd/kspde, ctypes) binds exactly these
 * symbols; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C linkage, plain pointers and sizes; no C++/torch types cross this boundary
 *   - every function returns KS_OK (0) or a negative ks_status code; the message of the last
 *     failure on the calling thread is available from ks_last_error()
 *   - a handle owns its device buffers and (unless ks_set_stream was called) its HIP stream;
 *     a handle is not re-entrant; distinct handles (e.g. one per GPU) may be driven from
 *     distinct host threads
 *   - "host" pointers are ordinary host memory owned by the caller; the *_device variants
 *     take device pointers, enqueue on the handle's stream and do NOT synchronise
 *   - device >= 0 names a HIP device; nothing ever falls back to the CPU.  device = -1 is an EXPLICIT request for
 *     the CPU twin (csrc/ks_cpu.cpp: the same arithmetic on host memory, same entry points; BASELINE configs[0],
 *     SURVEY 8(b) b4 / 8(d) baseline (B)); on such a handle the *_device entries take host pointers and run
 *     synchronously, ks_set_stream / a non-auto ks_set_variant return KS_ERR_UNSUPPORTED
 *
 * State layout in HBM: u is fp64, row-major [num_envs, N] (one env = one contiguous row of
 * N doubles); phi / obs are fp32 [num_envs, N]; actions fp32 [num_envs, n_act];
 * forcing matrix F fp32 [n_act, N].
 */
#ifndef KSPDE_H
#define KSPDE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ks_handle ks_handle;

typedef enum ks_status {
    KS_OK = 0,
    KS_ERR_INVALID = -1,     /* bad argument (NULL handle, N unsupported, ...)       */
    KS_ERR_HIP = -2,         /* a HIP runtime call failed (message has the detail)   */
    KS_ERR_NO_DEVICE = -3,   /* no such HIP device                                   */
    KS_ERR_UNSUPPORTED = -4, /* feature not available for this N / layout            */
    KS_ERR_SELFTEST = -5     /* ks_selftest found a mismatching primitive            */
} ks_status;

/* Arithmetic mode of the fused stepper.
 *   KS_MODE_FAST  : merged 9-point linear stencil, FMA everywhere, reward reduced once per
 *                   call.  Differs from the reference by rounding only (<= 1e-12 per sub-step;
 *                   the contract is L_inf < 1e-9 per sub-step).
 *   KS_MODE_EXACT : the reference's operation order (scipy correlate1d summation order,
 *                   true divisions, no FMA contraction).  State is bit-identical to the
 *                   reference CPU stepper.  ~3-4x slower; used as the parity anchor. */
typedef enum ks_mode { KS_MODE_FAST = 0, KS_MODE_EXACT = 1 } ks_mode;

/* Kernel layout / halo-exchange variant.  KS_VARIANT_AUTO picks per (N, num_envs). */
typedef enum ks_variant {
    KS_VARIANT_AUTO = 0,
    KS_VARIANT_ROW16_DPP = 1,  /* env = 16 lanes x N/16 points, halo by DPP row_ror          */
    KS_VARIANT_ROW16_BPERM = 2,/* same layout, halo by ds_bpermute                            */
    KS_VARIANT_WAVE64_DPP = 3, /* env = 64 lanes x N/64 points, halo by DPP wave_ror/rol:1    */
    KS_VARIANT_WAVE64_BPERM = 4,/* same layout, halo by ds_bpermute                           */
    KS_VARIANT_HALF32_BPERM = 5,/* env = 32 lanes x N/32 points, halo by ds_bpermute          */
    KS_VARIANT_LDS = 6,        /* one workgroup per env, state staged in LDS; any 9 <= N <= 2048 */
    KS_VARIANT_WAVE64_HYBRID = 7, /* N = 64: env = one wavefront, halo +-1, +-2 by DPP wave_ror/rol, +-3, +-4 by ds_bpermute */
    KS_VARIANT_WAVE64_HYBRID1 = 8 /* N = 64: halo +-1..+-3 by the DPP chain, +-4 by ds_bpermute */
} ks_variant;

/* ---- lifetime -------------------------------------------------------------------------- */

/* Replaces: KuramotoSivashinskyEnv.__init__ (pdegym/kuramoto/kuramoto.py:29-57) for a batch of
 * num_envs independent environments: grid size N, domain length L (dx = L/N), RK4 step dt.
 * device: HIP device ordinal, or -1 for the CPU twin. */
int ks_create(int device, int num_envs, int N, double L, double dt, ks_handle** out);
int ks_destroy(ks_handle* h);

/* Run on a caller-provided hipStream_t (e.g. torch's current stream) instead of the handle's own. */
int ks_set_stream(ks_handle* h, void* hip_stream);
int ks_set_mode(ks_handle* h, int mode /* ks_mode */);
int ks_set_variant(ks_handle* h, int variant /* ks_variant */);
/* Threads per workgroup of the fused kernels (64, 128 or 256; 0 = default). */
int ks_set_block_size(ks_handle* h, int threads);
/* Reports the layout the next ks_step will use. */
int ks_get_layout(ks_handle* h, int* variant, int* lanes_per_env, int* points_per_lane,
                  int* block_threads, int* grid_blocks);

/* ---- forcing ----------------------------------------------------------------------------- */

/* Replaces: GaussianForcing.forcing, the fp32 [n_act, N] matrix built in
 * pdegym/common/transforms.py:256-260.  The host computes it once (same torch ops as the
 * reference) and uploads it; ks_step_actions then evaluates phi = actions @ F on the device as
 * the fp32 FMA chain a0*F0 (+) a1*F1 (+) ... that torch's CPU matmul uses (transforms.py:262-265). */
int ks_set_forcing(ks_handle* h, const float* F_host, int n_act);

/* ---- state ------------------------------------------------------------------------------- */

/* Replaces: assignment to KuramotoSivashinskyEnv.u (kuramoto.py:106) / reading it (:94). */
int ks_set_state(ks_handle* h, const double* u_host /* [num_envs, N] */);
int ks_get_state(ks_handle* h, double* u_host /* [num_envs, N] */);
/* Scatter n rows: row i of u_host goes to env env_ids[i] (masked reset of finished episodes). */
int ks_set_state_rows(ks_handle* h, const int* env_ids_host, int n, const double* u_host /* [n, N] */);
/* Device pointer of the [num_envs, N] fp64 state (for zero-copy consumers). */
int ks_state_device_ptr(ks_handle* h, double** d_u);

/* ---- the hot path ------------------------------------------------------------------------ */

/* Replaces: the loop body of KuramotoSivashinskyEnv.step (kuramoto.py:83-90) run n_substeps
 * times for every env: reward term, then one classical RK4 update of u' = rhs(u, phi).
 *   phi_host   fp32 [num_envs, N] forcing field, or NULL for phi = 0 (reset burn-in, :108-109)
 *   obs_f32    out, fp32 [num_envs, N] copy of the new state (gym observation dtype), or NULL
 *   ssq_sum    out, fp64 [num_envs]: sum over the n_substeps of sum_i u_i^2 taken BEFORE each
 *              update; the l2control reward of kuramoto.py:64-65,84,96 is
 *              -(1/N) * ssq_sum / cfg_steps.  May be NULL.
 *   status     out, int [num_envs]: 1 if the env's state is non-finite after the call (the
 *              reference raises FloatingPointError via np.seterr(over="raise"), kuramoto.py:12).
 * Synchronous: results are in the host buffers on return. */
int ks_step(ks_handle* h, const float* phi_host, long n_substeps, float* obs_f32, double* ssq_sum,
            int* status);

/* Same, with phi computed on the device from actions [num_envs, n_act] and the uploaded forcing
 * matrix.  Replaces kuramoto.py:79-80 + :83-90. */
int ks_step_actions(ks_handle* h, const float* actions_host, long n_substeps, float* obs_f32,
                    double* ssq_sum, int* status);

/* Step only the n envs listed in env_ids_host with phi = 0 (burn-in of freshly reset envs,
 * kuramoto.py:103-109: Tsteps * cfg_steps sub-steps).  Outputs are indexed by position i. */
int ks_step_rows(ks_handle* h, const int* env_ids_host, int n, long n_substeps, float* obs_f32,
                 double* ssq_sum, int* status);

/* Split form of the host-boundary step, for ONE host thread driving several handles (one per GPU; replaces the
 * reference's one-subprocess-per-env AsyncVectorEnv.step_async / step_wait, pdecontrol/mbrl/mbrl.py:81-86):
 * ks_step_begin stages its inputs through pinned memory, enqueues the launch and the copy of the outputs into the
 * handle's pinned mirror, and returns WITHOUT waiting; ks_step_end waits for that step and hands the results over.
 * Call begin on every handle first, then end on every handle: all devices run concurrently.
 *   actions_host  fp32 [num_envs, n_act] or NULL (phi = 0)
 *   env_ids_host  int [n_rows] subset (outputs then in list order, as ks_step_rows) or NULL = all envs
 *   want_obs      0: ks_step_end may only be given obs_f32 = NULL
 * Between begin and end the handle accepts no other call that touches the stream (KS_ERR_INVALID). */
int ks_step_begin(ks_handle* h, const float* actions_host, const int* env_ids_host, int n_rows, long n_substeps,
                  int want_obs);
int ks_step_end(ks_handle* h, float* obs_f32, double* ssq_sum, int* status);

/* Asynchronous, device-resident form: every pointer is a DEVICE pointer (or NULL), the launch is
 * enqueued on the handle's stream and the call returns immediately.  Exactly one of d_phi /
 * d_actions may be non-NULL (both NULL: phi = 0).  d_env_ids (int [n_rows]) selects a subset of
 * envs, NULL = all num_envs (then n_rows is ignored); outputs are indexed by env id. */
int ks_step_device(ks_handle* h, const float* d_phi, const float* d_actions, const int* d_env_ids,
                   int n_rows, long n_substeps, float* d_obs_f32, double* d_ssq_sum, int* d_status);
int ks_sync(ks_handle* h);

/* ---- test hooks -------------------------------------------------------------------------- */

/* Replaces: KuramotoSivashinskyEnv.rhs (kuramoto.py:118-129) on a batch, in the reference's
 * operation order.  All buffers host, [n_rows, N]; any of ux/uxx/uxxxx may be NULL.
 * Does not touch the handle's state. */
int ks_rhs(ks_handle* h, const double* u_host, const float* phi_host, int n_rows, double* rhs,
           double* ux, double* uxx, double* uxxxx);

/* Runs every cross-lane primitive the fused kernels rely on (DPP row_ror, wave_ror/rol,
 * ds_bpermute) on lane ids and checks the result against the definition.  failed_mask gets one
 * bit per ks_variant value whose primitive misbehaved. */
int ks_selftest(ks_handle* h, unsigned* failed_mask);

const char* ks_last_error(void);
const char* ks_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KSPDE_H */
