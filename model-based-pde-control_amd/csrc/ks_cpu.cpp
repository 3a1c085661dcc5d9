// ks_cpu.cpp -- CPU twin of the fused KS stepper (see ks_cpu.h).  Compiled with -ffp-contract=off: the exact mode must keep
// the reference's unfused multiply / add order; the fast mode spells its FMAs out.
//
// What is computed (reference, paths relative to the reference root):
//   pdegym/kuramoto/kuramoto.py:118-129  rhs(u, phi) with the periodic stencil tables :24-27
//   pdegym/kuramoto/kuramoto.py:83-90    per sub-step: reward term (before the update), then classical RK4
//   pdegym/common/transforms.py:262-265  phi = action @ F in fp32
// One env is advanced through all its sub-steps by one thread on a window padded by the +-4 periodic halo, so the
// stencil loops carry no index arithmetic and vectorise; envs are spread over host threads (they never interact).
#include "ks_cpu.h"

#include <sched.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <thread>
#include <vector>

namespace kscpu {
namespace {

struct Scratch {
    std::vector<double> w, q, k, acc, us, phi;
    explicit Scratch(int N) : w(N + 8), q(N + 8), k(N), acc(N), us(N), phi(N) {}
};

// w[4 + i] = x[i], halo of 4 on both sides wrapped periodically; q = w * w
inline void window(const double* x, int N, double* w, double* q) {
    for (int i = 0; i < N; ++i) w[4 + i] = x[i];
    for (int m = 0; m < 4; ++m) {
        w[m] = x[N - 4 + m];
        w[N + 4 + m] = x[m];
    }
    for (int i = 0; i < N + 8; ++i) q[i] = w[i] * w[i];
}

// reference operation order (scipy correlate1d walks the flipped kernel from the far right tap; kuramoto.py:118-129)
inline void rhs_exact(const Params& p, const double* w, const double* q, const double* phi, int N, double* out,
                      double* ux, double* uxx, double* uxxxx) {
    const double dx = p.dx, dx2 = p.dx2, dx4 = p.dx4;
    for (int i = 0; i < N; ++i) {
        const int c = i + 4;
        double fwd = q[c + 4] * (-1.0 / 4);
        fwd += q[c] * (-25.0 / 12);
        fwd += q[c + 1] * 4.0;
        fwd += q[c + 2] * (-3.0);
        fwd += q[c + 3] * (4.0 / 3);
        double bwd = q[c - 4] * (1.0 / 4);
        bwd += q[c - 3] * (-4.0 / 3);
        bwd += q[c - 2] * 3.0;
        bwd += q[c - 1] * (-4.0);
        bwd += q[c] * (25.0 / 12);
        const double f = fwd / dx, b = bwd / dx, u = w[c];
        const double d1 = (u < 0.0 ? 1.0 : 0.0) * f + (u >= 0.0 ? 1.0 : 0.0) * b;   // u == 0 -> backward
        double d2 = u * (-49.0 / 18);
        d2 += (w[c - 3] + w[c + 3]) * (1.0 / 90);
        d2 += (w[c - 2] + w[c + 2]) * (-3.0 / 20);
        d2 += (w[c - 1] + w[c + 1]) * (3.0 / 2);
        d2 = d2 / dx2;
        double d4 = u * (91.0 / 8);
        d4 += (w[c - 4] + w[c + 4]) * (7.0 / 240);
        d4 += (w[c - 3] + w[c + 3]) * (-2.0 / 5);
        d4 += (w[c - 2] + w[c + 2]) * (169.0 / 60);
        d4 += (w[c - 1] + w[c + 1]) * (-122.0 / 15);
        d4 = d4 / dx4;
        out[i] = ((-d4 - d2) - 0.5 * d1) + phi[i];
        if (ux) ux[i] = d1;
        if (uxx) uxx[i] = d2;
        if (uxxxx) uxxxx[i] = d4;
    }
}

// merged linear stencil + both upwind sums sharing the centre term, FMA chains in the order of rhs_tile_fast
inline void rhs_fast(const Params& p, const double* w, const double* q, const double* phi, int N, double* out) {
    const double c0 = p.c_lin[0], c1 = p.c_lin[1], c2 = p.c_lin[2], c3 = p.c_lin[3], c4 = p.c_lin[4];
    const double m = p.mh_inv_dx;
    for (int i = 0; i < N; ++i) {
        const int c = i + 4;
        double lin = __builtin_fma(c0, w[c], phi[i]);
        lin = __builtin_fma(c1, w[c - 1] + w[c + 1], lin);
        lin = __builtin_fma(c2, w[c - 2] + w[c + 2], lin);
        lin = __builtin_fma(c3, w[c - 3] + w[c + 3], lin);
        lin = __builtin_fma(c4, w[c - 4] + w[c + 4], lin);
        double bw = (25.0 / 12) * q[c];
        double fw = __builtin_fma(4.0, q[c + 1], -bw);   // fw holds MINUS the forward sum
        bw = __builtin_fma(-4.0, q[c - 1], bw);
        fw = __builtin_fma(-3.0, q[c + 2], fw);
        bw = __builtin_fma(3.0, q[c - 2], bw);
        fw = __builtin_fma(4.0 / 3, q[c + 3], fw);
        bw = __builtin_fma(-4.0 / 3, q[c - 3], bw);
        fw = __builtin_fma(-0.25, q[c + 4], fw);
        bw = __builtin_fma(0.25, q[c - 4], bw);
        const double sel = (w[c] < 0.0) ? fw : bw;       // u == 0 selects the backward stencil
        out[i] = __builtin_fma(m, sel, lin);
    }
}

template <bool EXACT>
void advance_env(const Params& p, double* u, const double* phi, long n_substeps, Scratch& s, double* ssq_out) {
    const int N = p.N;
    double* w = s.w.data();
    double* q = s.q.data();
    double* k = s.k.data();
    double* acc = s.acc.data();
    double* us = s.us.data();
    const double dt = p.dt;
    double racc = 0.0;
    for (long step = 0; step < n_substeps; ++step) {
        // stage 1 + the reward term of this sub-step (taken BEFORE the update, kuramoto.py:84)
        window(u, N, w, q);
        double row = 0.0;
        for (int i = 0; i < N; ++i) row += q[4 + i];
        racc += row;
        if constexpr (EXACT) {
            rhs_exact(p, w, q, phi, N, k, nullptr, nullptr, nullptr);
            for (int i = 0; i < N; ++i) {
                acc[i] = k[i];
                us[i] = u[i] + dt * k[i] / 2.0;
            }
            window(us, N, w, q);
            rhs_exact(p, w, q, phi, N, k, nullptr, nullptr, nullptr);
            for (int i = 0; i < N; ++i) {
                acc[i] = acc[i] + 2.0 * k[i];
                us[i] = u[i] + dt * k[i] / 2.0;
            }
            window(us, N, w, q);
            rhs_exact(p, w, q, phi, N, k, nullptr, nullptr, nullptr);
            for (int i = 0; i < N; ++i) {
                acc[i] = acc[i] + 2.0 * k[i];
                us[i] = u[i] + dt * k[i];
            }
            window(us, N, w, q);
            rhs_exact(p, w, q, phi, N, k, nullptr, nullptr, nullptr);
            for (int i = 0; i < N; ++i) {
                acc[i] = acc[i] + k[i];
                u[i] = u[i] + dt * acc[i] / 6.0;
            }
        } else {
            rhs_fast(p, w, q, phi, N, k);
            for (int i = 0; i < N; ++i) {
                acc[i] = __builtin_fma(p.dt6, k[i], u[i]);
                us[i] = __builtin_fma(p.hdt, k[i], u[i]);
            }
            window(us, N, w, q);
            rhs_fast(p, w, q, phi, N, k);
            for (int i = 0; i < N; ++i) {
                acc[i] = __builtin_fma(p.dt3, k[i], acc[i]);
                us[i] = __builtin_fma(p.hdt, k[i], u[i]);
            }
            window(us, N, w, q);
            rhs_fast(p, w, q, phi, N, k);
            for (int i = 0; i < N; ++i) {
                acc[i] = __builtin_fma(p.dt3, k[i], acc[i]);
                us[i] = __builtin_fma(dt, k[i], u[i]);
            }
            window(us, N, w, q);
            rhs_fast(p, w, q, phi, N, k);
            for (int i = 0; i < N; ++i) u[i] = __builtin_fma(p.dt6, k[i], acc[i]);
        }
    }
    *ssq_out = racc;
}

void run_rows(const Params& p, int mode, double* u, const float* phi, const float* actions, const float* F, int n_act,
              const int* env_ids, int lo, int hi, long n_substeps, float* obs, double* ssq_sum, int* status) {
    const int N = p.N;
    Scratch s(N);
    for (int r = lo; r < hi; ++r) {
        const int env = env_ids ? env_ids[r] : r;
        double* ue = u + (size_t)env * N;
        if (phi) {
            for (int i = 0; i < N; ++i) s.phi[i] = (double)phi[(size_t)env * N + i];
        } else if (actions) {
            // fp32 FMA chain in action-index order == torch's CPU matmul (transforms.py:264)
            const float* act = actions + (size_t)env * n_act;
            for (int i = 0; i < N; ++i) {
                float a = act[0] * F[i];
                for (int k = 1; k < n_act; ++k) a = __builtin_fmaf(act[k], F[(size_t)k * N + i], a);
                s.phi[i] = (double)a;
            }
        } else {
            std::fill(s.phi.begin(), s.phi.end(), 0.0);
        }
        double ssq = 0.0;
        if (mode == 1)
            advance_env<true>(p, ue, s.phi.data(), n_substeps, s, &ssq);
        else
            advance_env<false>(p, ue, s.phi.data(), n_substeps, s, &ssq);
        int bad = 0;
        for (int i = 0; i < N; ++i) bad |= !std::isfinite(ue[i]);
        if (obs)
            for (int i = 0; i < N; ++i) obs[(size_t)env * N + i] = (float)ue[i];
        if (ssq_sum) ssq_sum[env] = ssq;
        if (status) status[env] = bad;
    }
}

}  // namespace

int default_threads() {
    int n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::max(1, CPU_COUNT(&set));
    if (const char* e = std::getenv("KSPDE_CPU_THREADS")) {
        const int v = std::atoi(e);
        if (v > 0) n = std::min(n, v);
    }
    return n;
}

void step(const Params& p, int mode, double* u, const float* phi, const float* actions, const float* F, int n_act,
          const int* env_ids, int n_rows, long n_substeps, float* obs, double* ssq_sum, int* status, int n_threads) {
    if (n_rows <= 0) return;
    // a thread is worth starting for >= ~1e6 point-sub-steps of its own
    const double work = (double)n_rows * (double)p.N * (double)std::max<long>(n_substeps, 1);
    int T = std::max(1, std::min({n_threads, n_rows, (int)(work / 1e6) + 1}));
    if (T == 1) {
        run_rows(p, mode, u, phi, actions, F, n_act, env_ids, 0, n_rows, n_substeps, obs, ssq_sum, status);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(T - 1);
    const int per = n_rows / T, extra = n_rows % T;
    int lo = 0;
    for (int t = 0; t < T; ++t) {
        const int hi = lo + per + (t < extra ? 1 : 0);
        if (t + 1 < T)
            pool.emplace_back(run_rows, std::cref(p), mode, u, phi, actions, F, n_act, env_ids, lo, hi, n_substeps, obs,
                              ssq_sum, status);
        else
            run_rows(p, mode, u, phi, actions, F, n_act, env_ids, lo, hi, n_substeps, obs, ssq_sum, status);
        lo = hi;
    }
    for (auto& th : pool) th.join();
}

void rhs(int N, double dx, double dx2, double dx4, const double* u, const float* phi, int n_rows, double* out, double* ux,
         double* uxx, double* uxxxx) {
    Params p{};
    p.N = N;
    p.dx = dx;
    p.dx2 = dx2;
    p.dx4 = dx4;
    Scratch s(N);
    for (int r = 0; r < n_rows; ++r) {
        const size_t off = (size_t)r * N;
        for (int i = 0; i < N; ++i) s.phi[i] = (double)phi[off + i];
        window(u + off, N, s.w.data(), s.q.data());
        rhs_exact(p, s.w.data(), s.q.data(), s.phi.data(), N, out + off, ux ? ux + off : nullptr,
                  uxx ? uxx + off : nullptr, uxxxx ? uxxxx + off : nullptr);
    }
}

}  // namespace kscpu
