// burgers.hip -- gfx950 (MI355X / CDNA4) batched viscous-Burgers stepper; C ABI in include/burgers_hip.h.
//
// What is computed (the reference's only Burgers artefact, pdecontrol/surrogates/phyloss/phyloss.py:36-86):
//   residual(u) = nu * u_xx - u * u_x + phi, u_x by the 2nd-order central stencil [-1/2, 0, 1/2]/dx, u_xx by the
//   4th-order central stencil [-1/12, 4/3, -5/2, 4/3, -1/12]/dx^2, periodic; one sub-step is the explicit midpoint
//   rule u <- u + dt * residual(u + dt/2 * residual(u))   (:83-86).
//
// Design (same as the KS stepper, csrc/ks_kernels.hip): one launch = all envs x all sub-steps, the fp32 state lives in
// VGPRs; one env is one 64-lane wavefront with P = N/64 contiguous points per lane; the +-2 halo comes from the two
// neighbouring lanes by DPP wave rotations (wave_ror:1 / wave_rol:1), which ARE the periodic boundary: no LDS, no
// barrier, no index arithmetic.  HBM is touched once on entry and once on exit; the kernel is VALU-issue bound.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/burgers_hip.h"

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

struct Args {
    float* u;
    const float* actions;
    const float* F;
    float* obs;
    double* ssq_sum;
    int* status;
    int n_act, n_envs, N;
    long n_substeps;
    float half_inv_dx;        // 1 / (2 dx)
    float l0, l1, l2;         // nu * laplace coefficients / dx^2: centre, +-1, +-2
    float dt, hdt;
};

// DPP control words (GFX9): wave_ror:1 = 0x13C (lane i receives lane i-1), wave_rol:1 = 0x134 (lane i receives lane i+1)
__device__ __forceinline__ float from_lower(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x13C, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_upper(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x134, 0xf, 0xf, false));
}

// window w[0 .. P+3]: w[2 + j] = v[j]; w[0], w[1] the two points left of the lane's first, w[P+2], w[P+3] right of its last
template <int P>
__device__ __forceinline__ void window(const float (&v)[P], float (&w)[P + 4]) {
#pragma unroll
    for (int j = 0; j < P; ++j) w[2 + j] = v[j];
    if constexpr (P == 1) {
        w[1] = from_lower(v[0]);
        w[0] = from_lower(w[1]);
        w[3] = from_upper(v[0]);
        w[4] = from_upper(w[3]);
    } else {
        w[1] = from_lower(v[P - 1]);
        w[0] = from_lower(v[P - 2]);
        w[P + 2] = from_upper(v[0]);
        w[P + 3] = from_upper(v[1]);
    }
}

// Two points per instruction: gfx950 has packed fp32 VALU ops (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32, IEEE per
// half, so the results are bit-identical to the scalar form), and this kernel is VALU-issue bound.  The stencil is
// written on <2 x float> values whose halves are neighbouring points; the compiler pairs the registers.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int P>
__device__ __forceinline__ void residual(const float (&w)[P + 4], const float (&phi)[P], const Args& a, float (&r)[P]) {
    if constexpr (P >= 2) {
        const f32x2 hid = {a.half_inv_dx, a.half_inv_dx}, l0 = {a.l0, a.l0}, l1 = {a.l1, a.l1}, l2 = {a.l2, a.l2};
#pragma unroll
        for (int j = 0; j < P; j += 2) {
            const f32x2 wm2 = {w[j], w[j + 1]}, wm1 = {w[j + 1], w[j + 2]}, w0 = {w[j + 2], w[j + 3]},
                        wp1 = {w[j + 3], w[j + 4]}, wp2 = {w[j + 4], w[j + 5]}, ph = {phi[j], phi[j + 1]};
            const f32x2 grad = (wp1 - wm1) * hid;
            f32x2 lap = __builtin_elementwise_fma(l0, w0, ph);
            lap = __builtin_elementwise_fma(l1, wm1 + wp1, lap);
            lap = __builtin_elementwise_fma(l2, wm2 + wp2, lap);
            const f32x2 rr = __builtin_elementwise_fma(-w0, grad, lap);
            r[j] = rr[0];
            r[j + 1] = rr[1];
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const float grad = (w[j + 3] - w[j + 1]) * a.half_inv_dx;
        float lap = fmaf(a.l0, w[j + 2], phi[j]);
        lap = fmaf(a.l1, w[j + 1] + w[j + 3], lap);
        lap = fmaf(a.l2, w[j] + w[j + 4], lap);
        r[j] = fmaf(-w[j + 2], grad, lap);
    }
}

// Pair-native form of one residual evaluation for P >= 4 (P even): the state is held as P/2 <2 x float> pairs of
// neighbouring points.  E[k] are the aligned pairs of the window (E[0] / E[P/2 + 1] the halo pairs from the neighbouring
// lanes), O[k] = (E[k].hi, E[k+1].lo) the pairs at odd offsets -- ONE v_pk_mov_b32 each instead of two v_mov_b32.
template <int H>   // H = P / 2 pairs per lane
__device__ __forceinline__ void residual_pairs(const f32x2 (&U)[H], const f32x2 (&PH)[H], const Args& a, f32x2 (&R)[H]) {
    f32x2 E[H + 2], O[H + 1];
#pragma unroll
    for (int k = 0; k < H; ++k) E[k + 1] = U[k];
    E[0] = f32x2{from_lower(U[H - 1][0]), from_lower(U[H - 1][1])};
    E[H + 1] = f32x2{from_upper(U[0][0]), from_upper(U[0][1])};
#pragma unroll
    for (int k = 0; k <= H; ++k) O[k] = __builtin_shufflevector(E[k], E[k + 1], 1, 2);
    const f32x2 hid = {a.half_inv_dx, a.half_inv_dx}, l0 = {a.l0, a.l0}, l1 = {a.l1, a.l1}, l2 = {a.l2, a.l2};
#pragma unroll
    for (int k = 0; k < H; ++k) {
        // points 2k, 2k + 1: centre E[k+1], +-2 E[k] / E[k+2], -1 O[k], +1 O[k+1]
        const f32x2 grad = (O[k + 1] - O[k]) * hid;
        f32x2 lap = __builtin_elementwise_fma(l0, E[k + 1], PH[k]);
        lap = __builtin_elementwise_fma(l1, O[k] + O[k + 1], lap);
        lap = __builtin_elementwise_fma(l2, E[k] + E[k + 2], lap);
        R[k] = __builtin_elementwise_fma(-E[k + 1], grad, lap);
    }
}

template <int P>
__global__ void __launch_bounds__(256) bg_step_kernel(const Args a) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool active = slot < a.n_envs;
    const int env = active ? slot : a.n_envs - 1;   // tail waves redo the last env (their lanes must still rotate)
    const size_t off = (size_t)env * a.N + (size_t)lane * P;

    float u[P], phi[P];
#pragma unroll
    for (int j = 0; j < P; ++j) u[j] = a.u[off + j];
    if (a.actions) {
        const float* act = a.actions + (size_t)env * a.n_act;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int i = lane * P + j;
            float acc = act[0] * a.F[i];
            for (int k = 1; k < a.n_act; ++k) acc = fmaf(act[k], a.F[(size_t)k * a.N + i], acc);
            phi[j] = acc;
        }
    } else {
#pragma unroll
        for (int j = 0; j < P; ++j) phi[j] = 0.0f;
    }

    double racc = 0.0;
    if constexpr (P >= 4) {
        constexpr int H = P / 2;
        f32x2 U[H], PH[H];
#pragma unroll
        for (int k = 0; k < H; ++k) {
            U[k] = f32x2{u[2 * k], u[2 * k + 1]};
            PH[k] = f32x2{phi[2 * k], phi[2 * k + 1]};
        }
        const f32x2 hdt2 = {a.hdt, a.hdt}, dt2 = {a.dt, a.dt};
        for (long s = 0; s < a.n_substeps; ++s) {
            float q = 0.0f;
#pragma unroll
            for (int k = 0; k < H; ++k) {            // same order as the scalar form: u[0], u[1], ...
                q = fmaf(U[k][0], U[k][0], q);
                q = fmaf(U[k][1], U[k][1], q);
            }
            racc += (double)q;
            f32x2 R[H], UT[H];
            residual_pairs<H>(U, PH, a, R);
#pragma unroll
            for (int k = 0; k < H; ++k) UT[k] = __builtin_elementwise_fma(hdt2, R[k], U[k]);
            residual_pairs<H>(UT, PH, a, R);
#pragma unroll
            for (int k = 0; k < H; ++k) U[k] = __builtin_elementwise_fma(dt2, R[k], U[k]);
        }
#pragma unroll
        for (int k = 0; k < H; ++k) {
            u[2 * k] = U[k][0];
            u[2 * k + 1] = U[k][1];
        }
    } else
    for (long s = 0; s < a.n_substeps; ++s) {
        float w[P + 4], r[P], ut[P];
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < P; ++j) q = fmaf(u[j], u[j], q);
        racc += (double)q;                       // reward term of this sub-step, before the update
        window<P>(u, w);
        residual<P>(w, phi, a, r);
#pragma unroll
        for (int j = 0; j < P; ++j) ut[j] = fmaf(a.hdt, r[j], u[j]);
        window<P>(ut, w);
        residual<P>(w, phi, a, r);
#pragma unroll
        for (int j = 0; j < P; ++j) u[j] = fmaf(a.dt, r[j], u[j]);
    }

    int bad = 0;
#pragma unroll
    for (int j = 0; j < P; ++j) bad |= !__builtin_isfinite(u[j]);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        racc += __shfl_xor(racc, m, 64);
        bad |= __shfl_xor(bad, m, 64);
    }
    if (active) {
#pragma unroll
        for (int j = 0; j < P; ++j) a.u[off + j] = u[j];
        if (a.obs) {
#pragma unroll
            for (int j = 0; j < P; ++j) a.obs[off + j] = u[j];
        }
        if (lane == 0) {
            if (a.ssq_sum) a.ssq_sum[env] = racc;
            if (a.status) a.status[env] = bad;
        }
    }
}

__global__ void bg_residual_kernel(const float* __restrict__ u, const float* __restrict__ phi, int n_rows, int N, float half_inv_dx,
                                   float l0, float l1, float l2, float* __restrict__ out) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n_rows * N) return;
    const int i = (int)(gid % N);
    const float* row = u + (gid - i);
    auto at = [&](int k) { int idx = i + k; idx = idx < 0 ? idx + N : (idx >= N ? idx - N : idx); return row[idx]; };
    const float grad = (at(1) - at(-1)) * half_inv_dx;
    float lap = fmaf(l0, at(0), phi ? phi[gid] : 0.0f);
    lap = fmaf(l1, at(-1) + at(1), lap);
    lap = fmaf(l2, at(-2) + at(2), lap);
    out[gid] = fmaf(-at(0), grad, lap);
}

template <int P>
hipError_t launch(const Args& a, hipStream_t st) {
    const int waves_per_block = 4;
    const int grid = (a.n_envs + waves_per_block - 1) / waves_per_block;
    hipLaunchKernelGGL(bg_step_kernel<P>, dim3(grid), dim3(64 * waves_per_block), 0, st, a);
    return hipGetLastError();
}

}  // namespace

extern "C" {

const char* bg_last_error(void) { return g_err; }

int bg_step(void* stream, float* u, const float* actions, const float* F, int n_act, int n_envs, int N, float dx, float dt,
            float nu, long n_substeps, float* obs, double* ssq_sum, int* status) {
    if (!u || n_envs <= 0 || N <= 0 || n_substeps < 0) return fail(-1, "bg_step: bad argument");
    if (actions && (!F || n_act <= 0)) return fail(-1, "bg_step: actions need the forcing matrix F and n_act > 0");
    if (!(dx > 0.0f) || !(dt > 0.0f) || !(nu >= 0.0f)) return fail(-1, "bg_step: dx, dt must be positive, nu non-negative");
    if (N % 64) return fail(-4, "bg_step: N = %d is not a multiple of 64", N);
    Args a{};
    a.u = u; a.actions = actions; a.F = F; a.obs = obs; a.ssq_sum = ssq_sum; a.status = status;
    a.n_act = n_act; a.n_envs = n_envs; a.N = N; a.n_substeps = n_substeps;
    a.half_inv_dx = 0.5f / dx;
    const float s = nu / (dx * dx);
    a.l0 = s * (-5.0f / 2.0f); a.l1 = s * (4.0f / 3.0f); a.l2 = s * (-1.0f / 12.0f);
    a.dt = dt; a.hdt = 0.5f * dt;
    hipError_t e;
    switch (N / 64) {
        case 1: e = launch<1>(a, (hipStream_t)stream); break;
        case 2: e = launch<2>(a, (hipStream_t)stream); break;
        case 4: e = launch<4>(a, (hipStream_t)stream); break;
        case 8: e = launch<8>(a, (hipStream_t)stream); break;
        case 16: e = launch<16>(a, (hipStream_t)stream); break;
        default: return fail(-4, "bg_step: N = %d: supported sizes are 64, 128, 256, 512, 1024", N);
    }
    if (e != hipSuccess) return fail(-2, "bg_step launch failed: %s", hipGetErrorString(e));
    return 0;
}

int bg_residual(void* stream, const float* u, const float* phi, int n_rows, int N, float dx, float nu, float* out) {
    if (!u || !out || n_rows <= 0 || N < 5 || !(dx > 0.0f)) return fail(-1, "bg_residual: bad argument");
    const float s = nu / (dx * dx);
    const size_t total = (size_t)n_rows * N;
    hipLaunchKernelGGL(bg_residual_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, phi, n_rows, N,
                       0.5f / dx, s * (-5.0f / 2.0f), s * (4.0f / 3.0f), s * (-1.0f / 12.0f), out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "bg_residual launch failed: %s", hipGetErrorString(e));
    return 0;
}

}  // extern "C"
