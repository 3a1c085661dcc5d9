/*
 * burgers_hip.h -- C ABI of libburgers_hip.so: batched viscous-Burgers stepper for gfx950 (SURVEY.md 8(f) row f4,
 * BASELINE.json configs[4]: "Burgers-1D env, 512 grid pts ... fp32").
 *
 * The reference ships NO Burgers environment (pdegym/__init__.py:2 imports a module that does not exist).  The
 * discretisation is the one its BurgersPhyPDELoss defines (pdecontrol/surrogates/phyloss/phyloss.py:36-86):
 *
 *   residual(u) = nu * laplace(u) - u * grad(u) (+ phi)        grad    = [-1/2, 0, 1/2] / dx            (circular)
 *                                                              laplace = [-1/12, 4/3, -5/2, 4/3, -1/12] / dx^2
 *   one sub-step: u <- u + dt * residual(u + dt/2 * residual(u))            ("improved Euler", :83-86)
 *
 * and these entry points replace what a pdegym/burgers env's step()/reset() would loop over in Python, in the same
 * way libkspde.so replaces pdegym/kuramoto/kuramoto.py:78-129: ONE launch advances every env by all n_substeps with
 * the fp32 state in registers (one env = one 64-lane wavefront x N/64 points per lane, +-2 halo by DPP wave
 * rotations = the periodic boundary), the forcing phi = actions @ F evaluated in-kernel, the l2 reward term
 * accumulated per sub-step (before the update, like the KS env) and reduced once.
 *
 * All pointers are DEVICE pointers; launches are asynchronous on the given hipStream_t.  Return 0 on success,
 * negative on error (bg_last_error()).  Supported grid sizes: N = 64 * P with P in {1, 2, 4, 8, 16}.
 */
#ifndef BURGERS_HIP_H
#define BURGERS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* u [E,N] fp32 in/out; actions [E,n_act] fp32 with F [n_act,N] fp32 (phi = actions @ F, fp32 FMA chain), or
 * actions == NULL for phi = 0; obs [E,N] fp32 or NULL; ssq_sum [E] fp64 (sum over sub-steps of sum_i u_i^2) or NULL;
 * status [E] int (1 = non-finite state) or NULL. */
int bg_step(void* stream, float* u, const float* actions, const float* F, int n_act, int n_envs, int N, float dx, float dt,
            float nu, long n_substeps, float* obs, double* ssq_sum, int* status);

/* test hook: out [n_rows,N] = nu * laplace(u) - u * grad(u) + phi (phi may be NULL) */
int bg_residual(void* stream, const float* u, const float* phi, int n_rows, int N, float dx, float nu, float* out);

const char* bg_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
