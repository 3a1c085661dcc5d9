// ks_cpu.h -- the CPU twin of the fused stepper (device = -1 behind the C ABI of include/kspde.h).
//
// Plain C++17, no HIP: the same arithmetic as ks_kernels.hip (both modes, same operation order) on host memory, envs
// spread over host threads.  It serves BASELINE configs[0] (one env on a GPU-less host), SURVEY 8(d) baseline (B) and the
// sanitizer build (make -C csrc asan).  It shares nothing with oracle/ (test infrastructure); the two are compared by
// tests/test_cpu_twin.py through the golden vectors.
#pragma once
#include <cstddef>

namespace kscpu {

struct Params {
    int N;
    double dt, dx, dx2, dx4;     // exact mode: the reference's divisors (kuramoto.py:55,122,126,129)
    double c_lin[5];             // fast mode: merged linear stencil -(D4_k/dx^4 + D2_k/dx^2)
    double mh_inv_dx;            // -0.5 / dx
    double hdt, dt6, dt3;        // dt/2, dt/6, dt/3
};

// Advance the n_rows envs listed in env_ids (nullptr = rows 0..n_rows-1) by n_substeps RK4 sub-steps.
//   u        [E,N] fp64 in/out            phi      [E,N] fp32 or nullptr
//   actions  [E,n_act] fp32 or nullptr    F        [n_act,N] fp32 (needed iff actions)
//   obs / ssq_sum / status  outputs indexed by env id, any may be nullptr
// mode: 0 = fast (merged stencil, FMA), 1 = exact (reference operation order, true divisions, no contraction).
void step(const Params& p, int mode, double* u, const float* phi, const float* actions, const float* F, int n_act,
          const int* env_ids, int n_rows, long n_substeps, float* obs, double* ssq_sum, int* status, int n_threads);

// rhs test hook in the reference's operation order; u, phi and outputs [n_rows, N]; ux / uxx / uxxxx may be nullptr.
void rhs(int N, double dx, double dx2, double dx4, const double* u, const float* phi, int n_rows, double* out,
         double* ux, double* uxx, double* uxxxx);

// Host threads the twin uses by default: the affinity mask, capped by KSPDE_CPU_THREADS.
int default_threads();

}  // namespace kscpu
