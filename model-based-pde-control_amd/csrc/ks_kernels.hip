// ks_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the batched Kuramoto-Sivashinsky stepper.
//
// What is computed (reference, paths relative to the reference root):
//   pdegym/kuramoto/kuramoto.py:118-129  rhs(u, phi) = -u_xxxx - u_xx - 1/2 (u^2)_x + phi with
//                                        periodic finite-difference stencils (tables :24-27)
//   pdegym/kuramoto/kuramoto.py:83-90    per sub-step: reward term, then classical RK4
//   pdegym/common/transforms.py:262-265  phi = action @ F in fp32
//
// Design (MI355X-first):
//   * One launch advances every env by ALL n_substeps.  The fp64 state lives in VGPRs for the
//     whole launch; HBM is touched once on entry (u, phi/actions) and once on exit (u, fp32 obs,
//     reward sum, status).  The kernel is therefore bound by the fp64 VALU pipe, not by HBM.
//   * An env occupies G consecutive lanes of a 64-wide wavefront (G = 16, 32 or 64) with
//     P = N/G contiguous grid points per lane.  The +-4 stencil halo comes from the neighbouring
//     lanes by DPP (row_ror inside a 16-lane DPP row, wave_ror/wave_rol across the whole wave:
//     both are *rotations*, i.e. exactly the periodic boundary) or by ds_bpermute.  No LDS
//     staging, no barriers, no inter-workgroup traffic: envs never interact.
//   * The reward (sum_i u_i^2 per sub-step) is accumulated per lane over all sub-steps and
//     reduced across the G lanes once, at the end of the launch.
//   * KS_MODE_EXACT keeps the reference's operation order (and no FMA contraction: this file is
//     compiled with -ffp-contract=off) so the state is bit-identical to the CPU reference;
//     KS_MODE_FAST merges the two linear stencils, uses explicit FMAs and pre-scaled constants.
//   * A generic workgroup-per-env LDS kernel covers every other N (9 <= N <= 2048).
#include "ks_internal.h"

#include "../../include/kspde.h"

namespace ks {

enum { HALO_BPERM = 0, HALO_DPP_ROW = 1, HALO_DPP_WAVE = 2, HALO_HYBRID = 3, HALO_HYBRID1 = 4 };

// ------------------------------------------------------------------------------------------
// cross-lane primitives
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double bperm64(int byte_addr, double x) {
    int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(x));
    int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(x));
    return __hiloint2double(hi, lo);
}

// DPP control words (GFX9): row_ror:n = 0x120+n, wave_rol:1 = 0x134, wave_ror:1 = 0x13C.
// "rotate right" moves data towards HIGHER lane ids: lane i receives lane i-n (mod width).
template <int D>
__device__ __forceinline__ double row_from_lower(double x) {  // value of lane (i - D) mod 16
    return dpp_mov64<0x120 + D>(x);
}
template <int D>
__device__ __forceinline__ double row_from_upper(double x) {  // value of lane (i + D) mod 16
    return dpp_mov64<0x120 + (16 - D)>(x);
}
__device__ __forceinline__ double wave_from_lower1(double x) { return dpp_mov64<0x13C>(x); }
__device__ __forceinline__ double wave_from_upper1(double x) { return dpp_mov64<0x134>(x); }

// Neighbour access inside a group of G lanes.  lower(d, x): x as held by lane (gl - d) mod G;
// upper(d, x): lane (gl + d) mod G.  CHAIN: only distance-1 moves exist, callers chain them.
template <int G, int HALO>
struct Halo;

template <int G>
struct Halo<G, HALO_BPERM> {
    static constexpr bool CHAIN = false;
    int lo_addr[4], up_addr[4];
    __device__ __forceinline__ Halo() {
        const int lane = threadIdx.x & 63, base = lane & ~(G - 1), gl = lane & (G - 1);
#pragma unroll
        for (int d = 1; d <= 4; ++d) {
            lo_addr[d - 1] = (base | ((gl - d) & (G - 1))) << 2;
            up_addr[d - 1] = (base | ((gl + d) & (G - 1))) << 2;
        }
    }
    __device__ __forceinline__ double lower(int d, double x) const { return bperm64(lo_addr[d - 1], x); }
    __device__ __forceinline__ double upper(int d, double x) const { return bperm64(up_addr[d - 1], x); }
};

template <>
struct Halo<16, HALO_DPP_ROW> {
    static constexpr bool CHAIN = false;
    __device__ __forceinline__ double lower(int d, double x) const {
        switch (d) {
            case 1: return row_from_lower<1>(x);
            case 2: return row_from_lower<2>(x);
            case 3: return row_from_lower<3>(x);
            default: return row_from_lower<4>(x);
        }
    }
    __device__ __forceinline__ double upper(int d, double x) const {
        switch (d) {
            case 1: return row_from_upper<1>(x);
            case 2: return row_from_upper<2>(x);
            case 3: return row_from_upper<3>(x);
            default: return row_from_upper<4>(x);
        }
    }
};

template <>
struct Halo<64, HALO_DPP_WAVE> {
    static constexpr bool CHAIN = true;
    __device__ __forceinline__ double lower(int, double x) const { return wave_from_lower1(x); }
    __device__ __forceinline__ double upper(int, double x) const { return wave_from_upper1(x); }
};

// One point per lane (N = 64), hybrid: distances 1 and 2 by the DPP chain (4 + 4 VALU moves), distances 3 and 4 through
// the LDS crossbar (8 ds_bpermute_b32: LDS-pipe instructions that cost no VALU issue slot and are in flight while the
// near-neighbour terms are computed).  16 -> 8 VALU moves per RK stage against the pure DPP chain.
template <>
struct Halo<64, HALO_HYBRID> {
    static constexpr bool CHAIN = false;
    int lo3, lo4, up3, up4;
    __device__ __forceinline__ Halo() {
        const int lane = threadIdx.x & 63;
        lo3 = ((lane - 3) & 63) << 2;
        lo4 = ((lane - 4) & 63) << 2;
        up3 = ((lane + 3) & 63) << 2;
        up4 = ((lane + 4) & 63) << 2;
    }
    // generic accessors (self test, exact mode): x is always the lane's own value
    __device__ __forceinline__ double lower(int d, double x) const {
        switch (d) {
            case 1: return wave_from_lower1(x);
            case 2: return wave_from_lower1(wave_from_lower1(x));
            case 3: return bperm64(lo3, x);
            default: return bperm64(lo4, x);
        }
    }
    __device__ __forceinline__ double upper(int d, double x) const {
        switch (d) {
            case 1: return wave_from_upper1(x);
            case 2: return wave_from_upper1(wave_from_upper1(x));
            case 3: return bperm64(up3, x);
            default: return bperm64(up4, x);
        }
    }
};

// same, with only distance 4 through the LDS crossbar (4 ds_bpermute_b32, 12 DPP moves per stage)
template <>
struct Halo<64, HALO_HYBRID1> : Halo<64, HALO_HYBRID> {
    __device__ __forceinline__ double lower(int d, double x) const {
        if (d == 3) return wave_from_lower1(wave_from_lower1(wave_from_lower1(x)));
        return Halo<64, HALO_HYBRID>::lower(d, x);
    }
    __device__ __forceinline__ double upper(int d, double x) const {
        if (d == 3) return wave_from_upper1(wave_from_upper1(wave_from_upper1(x)));
        return Halo<64, HALO_HYBRID>::upper(d, x);
    }
};

// x / d for a divisor d that is constant over the launch, bit-identical to the IEEE division the reference performs:
//   q = RN(x * r),  e = x - d * q (exact in one FMA),  result = RN(q + e * r),   r = RN(1 / d) from the host
// (Markstein's correction step: with a correctly rounded reciprocal one step from the faithful q lands on the correctly
// rounded quotient).  3 VALU instructions instead of the ~11 of the generic fp64 division sequence (v_div_scale x 2, v_rcp,
// 5 FMA, v_div_fmas, v_div_fixup) -- the exact mode spends 17 divisions per point and sub-step, 47 % of its instructions.
// Checked bit for bit against x / d on 7e9 dividends incl. ones placed next to rounding boundaries
// (tools/micro/markstein_check.c) and, end to end, by the golden tests (200 000-sub-step reset, bit-identical state).
// Not covered: x = -0.0 (gives +0.0; unreachable -- every dividend here is a sum whose coefficients have both signs, so a
// vanishing sum is +0.0) and non-finite x (the env raises FloatingPointError on those anyway).
__device__ __forceinline__ double div_const(double x, double d, double r) {
    const double q = x * r;
    const double e = __builtin_fma(-d, q, x);
    return __builtin_fma(e, r, q);
}

// ------------------------------------------------------------------------------------------
// rhs at one grid point.  w[] is the lane's window of u (4 halo + P + 4 halo), q[] = w[]^2,
// c the index of the point inside the window.
// ------------------------------------------------------------------------------------------
template <bool EXACT>
__device__ __forceinline__ double rhs_point(const double* w, const double* q, int c, double phi,
                                            const StepArgs& a) {
    if constexpr (EXACT) {
        // scipy correlate1d summation order (ni_filters.c): see oracle/ks_oracle.c
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
        const double f = div_const(fwd, a.dx, a.r_dx), b = div_const(bwd, a.dx, a.r_dx);
        const double u = w[c];
        const double d1 = (u < 0.0 ? 1.0 : 0.0) * f + (u >= 0.0 ? 1.0 : 0.0) * b;
        double d2 = u * (-49.0 / 18);
        d2 += (w[c - 3] + w[c + 3]) * (1.0 / 90);
        d2 += (w[c - 2] + w[c + 2]) * (-3.0 / 20);
        d2 += (w[c - 1] + w[c + 1]) * (3.0 / 2);
        d2 = div_const(d2, a.dx2, a.r_dx2);
        double d4 = u * (91.0 / 8);
        d4 += (w[c - 4] + w[c + 4]) * (7.0 / 240);
        d4 += (w[c - 3] + w[c + 3]) * (-2.0 / 5);
        d4 += (w[c - 2] + w[c + 2]) * (169.0 / 60);
        d4 += (w[c - 1] + w[c + 1]) * (-122.0 / 15);
        d4 = div_const(d4, a.dx4, a.r_dx4);
        return ((-d4 - d2) - 0.5 * d1) + phi;
    } else {
        double lin = __builtin_fma(a.c_lin[0], w[c], phi);
        lin = __builtin_fma(a.c_lin[1], w[c - 1] + w[c + 1], lin);
        lin = __builtin_fma(a.c_lin[2], w[c - 2] + w[c + 2], lin);
        lin = __builtin_fma(a.c_lin[3], w[c - 3] + w[c + 3], lin);
        lin = __builtin_fma(a.c_lin[4], w[c - 4] + w[c + 4], lin);
        // backward upwind table b = (25/12, -4, 3, -4/3, 1/4); forward table is its negation
        const double t0 = (25.0 / 12) * q[c];
        double bw = __builtin_fma(-4.0, q[c - 1], t0);
        bw = __builtin_fma(3.0, q[c - 2], bw);
        bw = __builtin_fma(-4.0 / 3, q[c - 3], bw);
        bw = __builtin_fma(0.25, q[c - 4], bw);
        double fw = __builtin_fma(-4.0, q[c + 1], t0);
        fw = __builtin_fma(3.0, q[c + 2], fw);
        fw = __builtin_fma(-4.0 / 3, q[c + 3], fw);
        fw = __builtin_fma(0.25, q[c + 4], fw);
        const double sel = (w[c] < 0.0) ? -fw : bw;  // u == 0 selects the backward stencil
        return __builtin_fma(a.mh_inv_dx, sel, lin);
    }
}


// FAST-mode rhs for a tile of TJ consecutive points, written op-major ("vector across the tile") so that
// consecutive instructions are independent: one wave per SIMD cannot hide the fp64 dependent-issue
// latency by switching waves, the instruction stream itself has to.
template <int TJ>
__device__ __forceinline__ void rhs_tile_fast(const double* w, const double* q, int c0, const double* phi,
                                              const StepArgs& a, double* k) {
    double lin[TJ], s1[TJ], s2[TJ], s3[TJ], s4[TJ], bw[TJ], fw[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t) s1[t] = w[c0 + t - 1] + w[c0 + t + 1];
#pragma unroll
    for (int t = 0; t < TJ; ++t) s2[t] = w[c0 + t - 2] + w[c0 + t + 2];
#pragma unroll
    for (int t = 0; t < TJ; ++t) lin[t] = __builtin_fma(a.c_lin[0], w[c0 + t], phi[t]);
#pragma unroll
    for (int t = 0; t < TJ; ++t) bw[t] = (25.0 / 12) * q[c0 + t];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TJ; ++t) s3[t] = w[c0 + t - 3] + w[c0 + t + 3];
#pragma unroll
    for (int t = 0; t < TJ; ++t) lin[t] = __builtin_fma(a.c_lin[1], s1[t], lin[t]);
#pragma unroll
    for (int t = 0; t < TJ; ++t) fw[t] = __builtin_fma(4.0, q[c0 + t + 1], -bw[t]);  // fw holds MINUS the forward sum
#pragma unroll
    for (int t = 0; t < TJ; ++t) bw[t] = __builtin_fma(-4.0, q[c0 + t - 1], bw[t]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TJ; ++t) s4[t] = w[c0 + t - 4] + w[c0 + t + 4];
#pragma unroll
    for (int t = 0; t < TJ; ++t) lin[t] = __builtin_fma(a.c_lin[2], s2[t], lin[t]);
#pragma unroll
    for (int t = 0; t < TJ; ++t) fw[t] = __builtin_fma(-3.0, q[c0 + t + 2], fw[t]);
#pragma unroll
    for (int t = 0; t < TJ; ++t) bw[t] = __builtin_fma(3.0, q[c0 + t - 2], bw[t]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TJ; ++t) lin[t] = __builtin_fma(a.c_lin[3], s3[t], lin[t]);
#pragma unroll
    for (int t = 0; t < TJ; ++t) fw[t] = __builtin_fma(4.0 / 3, q[c0 + t + 3], fw[t]);
#pragma unroll
    for (int t = 0; t < TJ; ++t) bw[t] = __builtin_fma(-4.0 / 3, q[c0 + t - 3], bw[t]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TJ; ++t) lin[t] = __builtin_fma(a.c_lin[4], s4[t], lin[t]);
#pragma unroll
    for (int t = 0; t < TJ; ++t) fw[t] = __builtin_fma(-0.25, q[c0 + t + 4], fw[t]);
#pragma unroll
    for (int t = 0; t < TJ; ++t) bw[t] = __builtin_fma(0.25, q[c0 + t - 4], bw[t]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
        const double sel = (w[c0 + t] < 0.0) ? fw[t] : bw[t];  // u == 0 selects the backward stencil
        k[t] = __builtin_fma(a.mh_inv_dx, sel, lin[t]);
    }
}

// FAST-mode rhs of the hybrid one-point-per-lane layout: far neighbours requested first, every term that only needs
// distances <= 2 computed while they travel.  Same operations in the same per-accumulator order as rhs_tile_fast, so
// the result is bit-identical to the other fast-mode layouts.
template <int FAR>   // FAR = 2: distances 3 and 4 by ds_bpermute; FAR = 1: distance 4 only
__device__ __forceinline__ double rhs_hybrid_fast(const Halo<64, HALO_HYBRID>& h, double u, double phi, const StepArgs& a,
                                                  double& q0) {
    double l3, r3;
    if constexpr (FAR == 2) {
        l3 = bperm64(h.lo3, u);
        r3 = bperm64(h.up3, u);
    }
    const double l4 = bperm64(h.lo4, u), r4 = bperm64(h.up4, u);
    __builtin_amdgcn_sched_barrier(0);          // the LDS requests go out FIRST (the scheduler would sink them to their use)
    const double l1 = wave_from_lower1(u), r1 = wave_from_upper1(u);
    const double l2 = wave_from_lower1(l1), r2 = wave_from_upper1(r1);
    if constexpr (FAR == 1) {
        l3 = wave_from_lower1(l2);
        r3 = wave_from_upper1(r2);
    }
    q0 = u * u;
    const double ql1 = l1 * l1, qr1 = r1 * r1, ql2 = l2 * l2, qr2 = r2 * r2;
    double lin = __builtin_fma(a.c_lin[0], u, phi);
    lin = __builtin_fma(a.c_lin[1], l1 + r1, lin);
    lin = __builtin_fma(a.c_lin[2], l2 + r2, lin);
    double bw = (25.0 / 12) * q0;
    double fw = __builtin_fma(4.0, qr1, -bw);   // fw holds MINUS the forward sum
    bw = __builtin_fma(-4.0, ql1, bw);
    fw = __builtin_fma(-3.0, qr2, fw);
    bw = __builtin_fma(3.0, ql2, bw);
    if constexpr (FAR == 1) {
        lin = __builtin_fma(a.c_lin[3], l3 + r3, lin);
        fw = __builtin_fma(4.0 / 3, r3 * r3, fw);
        bw = __builtin_fma(-4.0 / 3, l3 * l3, bw);
    }
    __builtin_amdgcn_sched_barrier(0);          // keep the far-neighbour terms (and their lgkmcnt wait) behind the near ones
    if constexpr (FAR == 2) {
        lin = __builtin_fma(a.c_lin[3], l3 + r3, lin);
        fw = __builtin_fma(4.0 / 3, r3 * r3, fw);
        bw = __builtin_fma(-4.0 / 3, l3 * l3, bw);
    }
    lin = __builtin_fma(a.c_lin[4], l4 + r4, lin);
    fw = __builtin_fma(-0.25, r4 * r4, fw);
    bw = __builtin_fma(0.25, l4 * l4, bw);
    const double sel = (u < 0.0) ? fw : bw;     // u == 0 selects the backward stencil
    return __builtin_fma(a.mh_inv_dx, sel, lin);
}

template <int P>
__host__ __device__ constexpr int tile_of() { return P % 4 == 0 ? 4 : (P % 3 == 0 ? 3 : (P % 2 == 0 ? 2 : 1)); }


template <int P, bool EXACT>
__device__ __forceinline__ void eval_rhs(const double* w, const double* q, const double (&phi)[P],
                                         const StepArgs& a, double (&kk)[P]) {
    if constexpr (EXACT) {
#pragma unroll
        for (int j = 0; j < P; ++j) kk[j] = rhs_point<true>(w, q, 4 + j, phi[j], a);
    } else {
        constexpr int TJ = tile_of<P>();
#pragma unroll
        for (int jb = 0; jb < P; jb += TJ) rhs_tile_fast<TJ>(w, q, 4 + jb, &phi[jb], a, &kk[jb]);
    }
}

// ------------------------------------------------------------------------------------------
// fused register-resident stepper
// ------------------------------------------------------------------------------------------
template <int P>
__host__ __device__ constexpr int dmax() { return (4 + P - 1) / P; }

// Is (distance d, local index r) part of the lower/left halo?  m = 1..4 is the offset to the
// left of the lane's first point: it lives in lane gl-ceil(m/P) at local index ceil(m/P)*P - m.
template <int P>
__host__ __device__ constexpr bool left_needed(int d, int r, bool chain) {
    for (int m = 1; m <= 4; ++m) {
        const int dd = (m + P - 1) / P, rr = dd * P - m;
        if (rr == r && (chain ? dd >= d : dd == d)) return true;
    }
    return false;
}
// m = 1..4 to the right of the lane's last point: lane gl+(P-1+m)/P, local index (P-1+m)%P.
template <int P>
__host__ __device__ constexpr bool right_needed(int d, int r, bool chain) {
    for (int m = 1; m <= 4; ++m) {
        const int t = P - 1 + m, dd = t / P, rr = t % P;
        if (rr == r && (chain ? dd >= d : dd == d)) return true;
    }
    return false;
}

template <int P, int G, int HALO, bool EXACT>
__device__ __forceinline__ void build_window(const Halo<G, HALO>& halo, const double (&us)[P],
                                             double (&w)[P + 8]) {
    constexpr bool CHAIN = Halo<G, HALO>::CHAIN;
#pragma unroll
    for (int j = 0; j < P; ++j) w[4 + j] = us[j];
    double lc[P], rc[P];
#pragma unroll
    for (int r = 0; r < P; ++r) lc[r] = rc[r] = us[r];
#pragma unroll
    for (int d = 1; d <= dmax<P>(); ++d) {
#pragma unroll
        for (int r = 0; r < P; ++r) {
            if (left_needed<P>(d, r, CHAIN)) lc[r] = halo.lower(d, CHAIN ? lc[r] : us[r]);
            if (right_needed<P>(d, r, CHAIN)) rc[r] = halo.upper(d, CHAIN ? rc[r] : us[r]);
        }
#pragma unroll
        for (int m = 1; m <= 4; ++m) {
            if ((m + P - 1) / P == d) w[4 - m] = lc[d * P - m];
            if ((P - 1 + m) / P == d) w[P + 3 + m] = rc[(P - 1 + m) % P];
        }
    }
}

template <int P, int G, int HALO, bool EXACT>
__global__ void __launch_bounds__(256) ks_rk4_fused(const StepArgs a) {
    constexpr int EPW = 64 / G;  // envs per wavefront
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int gl = lane & (G - 1);
    const int slot = wave * EPW + lane / G;
    const bool active = slot < a.n_rows;
    // tail groups redo the last env (their lanes must still take part in the cross-lane moves)
    const int slot_c = active ? slot : a.n_rows - 1;
    const int env = a.env_ids ? a.env_ids[slot_c] : slot_c;
    const size_t off = (size_t)env * a.N + (size_t)gl * P;

    const Halo<G, HALO> halo;

    double u[P], phi[P];
#pragma unroll
    for (int j = 0; j < P; ++j) u[j] = a.u[off + j];
    if (a.phi) {
#pragma unroll
        for (int j = 0; j < P; ++j) phi[j] = (double)a.phi[off + j];
    } else if (a.actions) {
        // fp32 FMA chain in action-index order == torch CPU matmul (transforms.py:264)
        const float* act = a.actions + (size_t)env * a.n_act;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int i = gl * P + j;
            float acc = act[0] * a.F[i];
            for (int k = 1; k < a.n_act; ++k) acc = __builtin_fmaf(act[k], a.F[(size_t)k * a.N + i], acc);
            phi[j] = (double)acc;
        }
    } else {
#pragma unroll
        for (int j = 0; j < P; ++j) phi[j] = 0.0;
    }

    double racc = 0.0;
    for (long s = 0; s < a.n_substeps; ++s) {
        double acc[P], us[P], usn[P], w[P + 8], q[P + 8], kk[P];
        // ---- stage 1 (k1 at u) + reward term of this sub-step ----
        if constexpr ((HALO == HALO_HYBRID || HALO == HALO_HYBRID1) && !EXACT) {
            double q0;
            kk[0] = rhs_hybrid_fast<(HALO == HALO_HYBRID ? 2 : 1)>(halo, u[0], phi[0], a, q0);
            racc += q0;
        } else {
            build_window<P, G, HALO, EXACT>(halo, u, w);
#pragma unroll
            for (int i = 0; i < P + 8; ++i) q[i] = w[i] * w[i];
#pragma unroll
            for (int j = 0; j < P; ++j) racc += q[4 + j];
            eval_rhs<P, EXACT>(w, q, phi, a, kk);
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const double k = kk[j];
            if constexpr (EXACT) {
                acc[j] = k;
                usn[j] = u[j] + a.dt * k / 2.0;
            } else {
                acc[j] = __builtin_fma(a.dt6, k, u[j]);
                usn[j] = __builtin_fma(a.hdt, k, u[j]);
            }
        }
        // ---- stage 2 ----
#pragma unroll
        for (int j = 0; j < P; ++j) us[j] = usn[j];
        if constexpr ((HALO == HALO_HYBRID || HALO == HALO_HYBRID1) && !EXACT) {
            double q0;
            kk[0] = rhs_hybrid_fast<(HALO == HALO_HYBRID ? 2 : 1)>(halo, us[0], phi[0], a, q0);
        } else {
            build_window<P, G, HALO, EXACT>(halo, us, w);
#pragma unroll
            for (int i = 0; i < P + 8; ++i) q[i] = w[i] * w[i];
            eval_rhs<P, EXACT>(w, q, phi, a, kk);
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const double k = kk[j];
            if constexpr (EXACT) {
                acc[j] = acc[j] + 2.0 * k;
                usn[j] = u[j] + a.dt * k / 2.0;
            } else {
                acc[j] = __builtin_fma(a.dt3, k, acc[j]);
                usn[j] = __builtin_fma(a.hdt, k, u[j]);
            }
        }
        // ---- stage 3 ----
#pragma unroll
        for (int j = 0; j < P; ++j) us[j] = usn[j];
        if constexpr ((HALO == HALO_HYBRID || HALO == HALO_HYBRID1) && !EXACT) {
            double q0;
            kk[0] = rhs_hybrid_fast<(HALO == HALO_HYBRID ? 2 : 1)>(halo, us[0], phi[0], a, q0);
        } else {
            build_window<P, G, HALO, EXACT>(halo, us, w);
#pragma unroll
            for (int i = 0; i < P + 8; ++i) q[i] = w[i] * w[i];
            eval_rhs<P, EXACT>(w, q, phi, a, kk);
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const double k = kk[j];
            if constexpr (EXACT) {
                acc[j] = acc[j] + 2.0 * k;
                usn[j] = u[j] + a.dt * k;
            } else {
                acc[j] = __builtin_fma(a.dt3, k, acc[j]);
                usn[j] = __builtin_fma(a.dt, k, u[j]);
            }
        }
        // ---- stage 4 + update ----
#pragma unroll
        for (int j = 0; j < P; ++j) us[j] = usn[j];
        if constexpr ((HALO == HALO_HYBRID || HALO == HALO_HYBRID1) && !EXACT) {
            double q0;
            kk[0] = rhs_hybrid_fast<(HALO == HALO_HYBRID ? 2 : 1)>(halo, us[0], phi[0], a, q0);
        } else {
            build_window<P, G, HALO, EXACT>(halo, us, w);
#pragma unroll
            for (int i = 0; i < P + 8; ++i) q[i] = w[i] * w[i];
            eval_rhs<P, EXACT>(w, q, phi, a, kk);
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const double k = kk[j];
            if constexpr (EXACT) {
                acc[j] = acc[j] + k;
                u[j] = u[j] + div_const(a.dt * acc[j], 6.0, 1.0 / 6.0);
            } else {
                u[j] = __builtin_fma(a.dt6, k, acc[j]);
            }
        }
    }

    // ---- epilogue: state, fp32 observation, reward sum, non-finite flag ----
    int bad = 0;
#pragma unroll
    for (int j = 0; j < P; ++j) bad |= !__builtin_isfinite(u[j]);
#pragma unroll
    for (int m = 1; m < G; m <<= 1) {
        racc += __shfl_xor(racc, m, 64);
        bad |= __shfl_xor(bad, m, 64);
    }
    if (active) {
#pragma unroll
        for (int j = 0; j < P; ++j) a.u[off + j] = u[j];
        if (a.obs) {
#pragma unroll
            for (int j = 0; j < P; ++j) a.obs[off + j] = (float)u[j];
        }
        if (gl == 0) {
            if (a.ssq_sum) a.ssq_sum[env] = racc;
            if (a.status) a.status[env] = bad;
        }
    }
}

// ------------------------------------------------------------------------------------------
// generic stepper: one workgroup per env, state staged in LDS; any 9 <= N <= 2048
// LDS: U[N] ACC[N] PHI[N] S0[N] S1[N] (fp64) + reduction scratch
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int wrap_idx(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

template <bool EXACT>
__global__ void __launch_bounds__(256) ks_rk4_lds(const StepArgs a) {
    extern __shared__ __align__(16) double lds[];
    const int N = a.N, tid = threadIdx.x, T = blockDim.x;
    double* U = lds;
    double* ACC = U + N;
    double* PHI = ACC + N;
    double* S0 = PHI + N;
    double* S1 = S0 + N;
    double* red = S1 + N;  // [T/64] partials
    __shared__ int bad_any;

    const int slot = blockIdx.x;  // grid == n_rows
    const int env = a.env_ids ? a.env_ids[slot] : slot;
    const size_t off = (size_t)env * N;

    for (int i = tid; i < N; i += T) {
        U[i] = a.u[off + i];
        double p = 0.0;
        if (a.phi) {
            p = (double)a.phi[off + i];
        } else if (a.actions) {
            const float* act = a.actions + (size_t)env * a.n_act;
            float acc = act[0] * a.F[i];
            for (int k = 1; k < a.n_act; ++k) acc = __builtin_fmaf(act[k], a.F[(size_t)k * N + i], acc);
            p = (double)acc;
        }
        PHI[i] = p;
    }
    if (tid == 0) bad_any = 0;
    __syncthreads();

    double racc = 0.0;
    for (long s = 0; s < a.n_substeps; ++s) {
        const double* src = U;
        double* dst = S0;
#pragma unroll 1
        for (int stage = 0; stage < 4; ++stage) {
            for (int i = tid; i < N; i += T) {
                double w[9], q[9];
#pragma unroll
                for (int k = -4; k <= 4; ++k) {
                    w[k + 4] = src[wrap_idx(i + k, N)];
                    q[k + 4] = w[k + 4] * w[k + 4];
                }
                if (stage == 0) racc += q[4];
                const double k = rhs_point<EXACT>(w, q, 4, PHI[i], a);
                const double u0 = U[i];
                if constexpr (EXACT) {
                    if (stage == 0) { ACC[i] = k; dst[i] = u0 + a.dt * k / 2.0; }
                    else if (stage == 1) { ACC[i] = ACC[i] + 2.0 * k; dst[i] = u0 + a.dt * k / 2.0; }
                    else if (stage == 2) { ACC[i] = ACC[i] + 2.0 * k; dst[i] = u0 + a.dt * k; }
                    else { const double s4 = ACC[i] + k; dst[i] = u0 + div_const(a.dt * s4, 6.0, 1.0 / 6.0); }
                } else {
                    if (stage == 0) { ACC[i] = __builtin_fma(a.dt6, k, u0); dst[i] = __builtin_fma(a.hdt, k, u0); }
                    else if (stage == 1) { ACC[i] = __builtin_fma(a.dt3, k, ACC[i]); dst[i] = __builtin_fma(a.hdt, k, u0); }
                    else if (stage == 2) { ACC[i] = __builtin_fma(a.dt3, k, ACC[i]); dst[i] = __builtin_fma(a.dt, k, u0); }
                    else { dst[i] = __builtin_fma(a.dt6, k, ACC[i]); }
                }
            }
            __syncthreads();
            // stage outputs ping-pong S0 -> S1 -> S0 -> S1; the 4th result (in S1) is the new U
            src = dst;
            dst = (dst == S0) ? S1 : S0;
        }
        for (int i = tid; i < N; i += T) U[i] = S1[i];
        __syncthreads();
    }

    int bad = 0;
    for (int i = tid; i < N; i += T) {
        const double v = U[i];
        bad |= !__builtin_isfinite(v);
        a.u[off + i] = v;
        if (a.obs) a.obs[off + i] = (float)v;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) racc += __shfl_xor(racc, m, 64);
    if ((tid & 63) == 0) red[tid >> 6] = racc;
    if (bad) atomicOr(&bad_any, 1);
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int wv = 0; wv < (T >> 6); ++wv) tot += red[wv];
        if (a.ssq_sum) a.ssq_sum[env] = tot;
        if (a.status) a.status[env] = bad_any;
    }
}

// ------------------------------------------------------------------------------------------
// rhs test hook (reference operation order), one thread per grid point
// ------------------------------------------------------------------------------------------
__global__ void ks_rhs_kernel(const double* __restrict__ u, const float* __restrict__ phi, int n_rows,
                              int N, double dx, double dx2, double dx4, double* __restrict__ rhs,
                              double* __restrict__ ux, double* __restrict__ uxx,
                              double* __restrict__ uxxxx) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n_rows * N) return;
    const int row = (int)(gid / N), i = (int)(gid % N);
    const double* ur = u + (size_t)row * N;
    double w[9], q[9];
#pragma unroll
    for (int k = -4; k <= 4; ++k) {
        int idx = (i + k) % N;
        if (idx < 0) idx += N;
        w[k + 4] = ur[idx];
        q[k + 4] = w[k + 4] * w[k + 4];
    }
    const int c = 4;
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
    const double f = fwd / dx, b = bwd / dx, uc = w[c];
    const double d1 = (uc < 0.0 ? 1.0 : 0.0) * f + (uc >= 0.0 ? 1.0 : 0.0) * b;
    double d2 = uc * (-49.0 / 18);
    d2 += (w[c - 3] + w[c + 3]) * (1.0 / 90);
    d2 += (w[c - 2] + w[c + 2]) * (-3.0 / 20);
    d2 += (w[c - 1] + w[c + 1]) * (3.0 / 2);
    d2 = d2 / dx2;
    double d4 = uc * (91.0 / 8);
    d4 += (w[c - 4] + w[c + 4]) * (7.0 / 240);
    d4 += (w[c - 3] + w[c + 3]) * (-2.0 / 5);
    d4 += (w[c - 2] + w[c + 2]) * (169.0 / 60);
    d4 += (w[c - 1] + w[c + 1]) * (-122.0 / 15);
    d4 = d4 / dx4;
    rhs[gid] = ((-d4 - d2) - 0.5 * d1) + (double)phi[gid];
    if (ux) ux[gid] = d1;
    if (uxx) uxx[gid] = d2;
    if (uxxxx) uxxxx[gid] = d4;
}

// ------------------------------------------------------------------------------------------
// cross-lane self test: every primitive moves the lane id and must deliver the defined source
// ------------------------------------------------------------------------------------------
template <int G, int HALO>
__device__ unsigned check_halo() {
    const Halo<G, HALO> halo;
    const int lane = threadIdx.x & 63, base = lane & ~(G - 1), gl = lane & (G - 1);
    unsigned fail = 0;
    // payload: distinct high and low words per lane
    const double x = __hiloint2double(0x40000000 | (lane << 8), 0x1234 + lane * 7);
    double lo = x, up = x;
#pragma unroll
    for (int d = 1; d <= 4; ++d) {
        lo = halo.lower(d, Halo<G, HALO>::CHAIN ? lo : x);
        up = halo.upper(d, Halo<G, HALO>::CHAIN ? up : x);
        const int sl = base | ((gl - d) & (G - 1)), su = base | ((gl + d) & (G - 1));
        const double el = __hiloint2double(0x40000000 | (sl << 8), 0x1234 + sl * 7);
        const double eu = __hiloint2double(0x40000000 | (su << 8), 0x1234 + su * 7);
        if (__double_as_longlong(lo) != __double_as_longlong(el)) fail = 1;
        if (__double_as_longlong(up) != __double_as_longlong(eu)) fail = 1;
    }
    return fail;
}

__global__ void ks_selftest_kernel(unsigned* out) {
    unsigned mask = 0;
    if (check_halo<16, HALO_DPP_ROW>()) mask |= 1u << KS_VARIANT_ROW16_DPP;
    if (check_halo<16, HALO_BPERM>()) mask |= 1u << KS_VARIANT_ROW16_BPERM;
    if (check_halo<64, HALO_DPP_WAVE>()) mask |= 1u << KS_VARIANT_WAVE64_DPP;
    if (check_halo<64, HALO_BPERM>()) mask |= 1u << KS_VARIANT_WAVE64_BPERM;
    if (check_halo<32, HALO_BPERM>()) mask |= 1u << KS_VARIANT_HALF32_BPERM;
    if (check_halo<64, HALO_HYBRID>()) mask |= 1u << KS_VARIANT_WAVE64_HYBRID;
    if (check_halo<64, HALO_HYBRID1>()) mask |= 1u << KS_VARIANT_WAVE64_HYBRID1;
    if (mask) atomicOr(out, mask);
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------
template <int P, int G, int HALO>
static hipError_t launch_fused(const Layout& lay, int mode, const StepArgs& a, hipStream_t st) {
    if (mode == KS_MODE_EXACT)
        hipLaunchKernelGGL((ks_rk4_fused<P, G, HALO, true>), dim3(lay.grid), dim3(lay.block), 0, st, a);
    else
        hipLaunchKernelGGL((ks_rk4_fused<P, G, HALO, false>), dim3(lay.grid), dim3(lay.block), 0, st, a);
    return hipGetLastError();
}

#define KS_P_CASES(G_, HALO_)                                              \
    switch (lay.P) {                                                       \
        case 1: return launch_fused<1, G_, HALO_>(lay, mode, a, st);       \
        case 2: return launch_fused<2, G_, HALO_>(lay, mode, a, st);       \
        case 3: return launch_fused<3, G_, HALO_>(lay, mode, a, st);       \
        case 4: return launch_fused<4, G_, HALO_>(lay, mode, a, st);       \
        case 6: return launch_fused<6, G_, HALO_>(lay, mode, a, st);       \
        case 8: return launch_fused<8, G_, HALO_>(lay, mode, a, st);       \
        case 12: return launch_fused<12, G_, HALO_>(lay, mode, a, st);     \
        case 16: return launch_fused<16, G_, HALO_>(lay, mode, a, st);     \
        default: return hipErrorInvalidValue;                              \
    }

static bool p_supported(int P) {
    return P == 1 || P == 2 || P == 3 || P == 4 || P == 6 || P == 8 || P == 12 || P == 16;
}

bool layout_supported(int variant, int N) {
    switch (variant) {
        case KS_VARIANT_ROW16_DPP:
        case KS_VARIANT_ROW16_BPERM: return N % 16 == 0 && p_supported(N / 16);
        case KS_VARIANT_HALF32_BPERM: return N % 32 == 0 && p_supported(N / 32);
        case KS_VARIANT_WAVE64_DPP:
        case KS_VARIANT_WAVE64_BPERM: return N % 64 == 0 && p_supported(N / 64);
        case KS_VARIANT_WAVE64_HYBRID:
        case KS_VARIANT_WAVE64_HYBRID1: return N == 64;
        case KS_VARIANT_LDS: return N >= 9 && N <= 2048;
        default: return false;
    }
}

hipError_t launch_step(const Layout& lay, int mode, const StepArgs& a, hipStream_t st) {
    if (a.n_rows <= 0) return hipSuccess;
    switch (lay.variant) {
        case KS_VARIANT_ROW16_DPP: KS_P_CASES(16, HALO_DPP_ROW)
        case KS_VARIANT_ROW16_BPERM: KS_P_CASES(16, HALO_BPERM)
        case KS_VARIANT_HALF32_BPERM: KS_P_CASES(32, HALO_BPERM)
        case KS_VARIANT_WAVE64_DPP: KS_P_CASES(64, HALO_DPP_WAVE)
        case KS_VARIANT_WAVE64_BPERM: KS_P_CASES(64, HALO_BPERM)
        case KS_VARIANT_WAVE64_HYBRID: return lay.P == 1 ? launch_fused<1, 64, HALO_HYBRID>(lay, mode, a, st) : hipErrorInvalidValue;
        case KS_VARIANT_WAVE64_HYBRID1: return lay.P == 1 ? launch_fused<1, 64, HALO_HYBRID1>(lay, mode, a, st) : hipErrorInvalidValue;
        case KS_VARIANT_LDS:
            if (mode == KS_MODE_EXACT)
                hipLaunchKernelGGL((ks_rk4_lds<true>), dim3(lay.grid), dim3(lay.block), lay.lds_bytes, st, a);
            else
                hipLaunchKernelGGL((ks_rk4_lds<false>), dim3(lay.grid), dim3(lay.block), lay.lds_bytes, st, a);
            return hipGetLastError();
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_rhs(const double* u, const float* phi, int n_rows, int N, double dx, double dx2,
                      double dx4, double* rhs, double* ux, double* uxx, double* uxxxx, hipStream_t st) {
    const size_t total = (size_t)n_rows * N;
    if (total == 0) return hipSuccess;
    const int block = 256;
    const unsigned grid = (unsigned)((total + block - 1) / block);
    hipLaunchKernelGGL(ks_rhs_kernel, dim3(grid), dim3(block), 0, st, u, phi, n_rows, N, dx, dx2, dx4, rhs,
                       ux, uxx, uxxxx);
    return hipGetLastError();
}

hipError_t launch_selftest(unsigned* d_fail, hipStream_t st) {
    hipLaunchKernelGGL(ks_selftest_kernel, dim3(4), dim3(256), 0, st, d_fail);
    return hipGetLastError();
}

}  // namespace ks
