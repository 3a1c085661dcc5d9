// fno.hip -- one launch per FNO model evaluation (forward) and one per evaluation (backward) for gfx950; part of
// libspectral_hip.so (C ABI: include/spectral_hip.h, fno_*).
//
// The FNO-style surrogate of BASELINE configs[4] (pdecontrol/architectures/fno.py: lift -> 4 x [spectral convolution +
// pointwise convolution, GELU] -> project) has no counterpart in the reference (parity unpinned); what IS fixed
// is how the reference's training step drives a surrogate (pdecontrol/surrogates/training.py:64-130, surrogate.py:79-133):
// a Python loop over time steps, ~170 tiny kernels per step and sample batch.  Here one workgroup owns one (time step,
// sample) pair and walks the whole network with the activations [32 x N] in LDS:
//
//   forward   x0 = lift(u, a);  per layer: S = DFT_M(x) (truncated DFT as a [32 x N] @ [N x 2M] MFMA GEMM) -> complex mode
//             mixing with the layer's weights (streamed from L2) -> pre = iDFT(Z) + Wp x + b (both as MFMA GEMMs into the
//             same accumulator tile, position tile by position tile, IN PLACE) -> x = gelu(pre);  project (two pointwise
//             layers) and the integration  out = u + cscale * delta + cshift  in the epilogue.
//             Saved for the backward pass: the four pre-activations (HBM is the activation store) and the truncated
//             spectra of the layer inputs.
//   backward  the same walk in reverse: gelu', the pointwise weight gradients as K = N MFMA contractions of the gradient
//             buffer with the recomputed layer input, dx = iDFT(conj(W) . DFT(d_pre)) + Wp^T d_pre in place; the spectral
//             weight gradient is a contraction over (step, sample) pairs of two saved spectra, done once per rollout by
//             the caller; every other parameter gradient leaves as one row per pair (deterministic, no atomics across
//             workgroups), summed by fno_reduce_rows.
//
// Geometry: width 32, 16 modes, N a power of two in [64, 512] (LDS: two [32][N + 4] buffers + spectra = 156 KB at N = 512).
// fp32 throughout, v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <utility>

#include "../../include/spectral_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 32;        // width
constexpr int M = 16;        // modes
constexpr int K2 = 2 * M;    // real | imaginary columns of a truncated spectrum
constexpr int KP = K2 + 4;   // padded row of a spectrum / of a 32 x 32 weight matrix in LDS
constexpr int KS = 2;        // K-split of the K = N contractions (8 waves: 4 output tiles x 2 halves of K)
constexpr int TPB = 512;
constexpr int LAYERS = 4;

thread_local char g_err[256] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

struct Weights {            // device pointers, fp32
    const float* lift_w;    // [32][2]
    const float* lift_b;    // [32]
    const float* wr[LAYERS];  // [32 in][32 out][16]
    const float* wi[LAYERS];
    const float* pw[LAYERS];  // [32 out][32 in]
    const float* pb[LAYERS];  // [32]
    const float* p1_w;      // [32][32]
    const float* p1_b;      // [32]
    const float* p2_w;      // [32]
    const float* p2_b;      // [1]
};

// row p = t * nb + b of a strided [steps][batch][N] view
struct Rows {
    const float* ptr;
    long stride_t, stride_b;
    __device__ const float* row(int t, int b) const { return ptr + t * stride_t + b * stride_b; }
};

struct FwdArgs {
    Weights w;
    Rows u, act;            // state / action field of every pair
    float* delta;           // [pairs][N]  model output (scaled delta)
    float* out;             // [pairs][N]  u + cscale * delta + cshift, or nullptr
    float* pre;             // [pairs][4][32][N] saved pre-activations, or nullptr (inference)
    float* xspec;           // [4][32 k][spec_pairs][32 c] truncated spectra of the layer inputs, or nullptr
    const float* tabg;      // [32][N] twiddle matrix (global, per device and N)
    int n, nb, pairs;
    int spec_pairs, spec_pair0;   // the spectra buffer spans spec_pairs pairs; this launch's pair p is its pair spec_pair0 + p
    float cscale, cshift;
};

struct BwdArgs {
    Weights w;
    Rows u, act;
    const float* gdelta;    // [pairs][N] d loss / d delta
    const float* gout;      // [nb][N] d loss / d out of step gout_t (from the step that used it as its base), or nullptr
    int gout_t;
    const float* pre;       // [pairs][4][32][N]
    float* gspec;           // [4][32 k][spec_pairs][32 c] scaled spectra of d_pre
    float* rows;            // [pairs][ROW] parameter-gradient rows (everything but the spectral weights)
    float* dbase;           // [pairs][N] d loss / d u (including gout passed through), or nullptr
    const float* tabg;      // [32][N] twiddle matrix (global, per device and N)
    int n, nb, pairs;
    int spec_pairs, spec_pair0;
    float cscale;
};

// gradient-row layout
constexpr int R_LIFT_W = 0, R_LIFT_B = 64, R_LAYER = 96, R_LAYER_SZ = 1024 + 32;
constexpr int R_P1W = R_LAYER + LAYERS * R_LAYER_SZ, R_P1B = R_P1W + 1024, R_P2W = R_P1B + 32, R_P2B = R_P2W + 32;
constexpr int ROW = ((R_P2B + 1 + 63) / 64) * 64;

// GELU (exact, erf form) and its derivative from ONE exponential: with z = |x| / sqrt(2), erf(z) by Abramowitz & Stegun
// 7.1.26 (|error| <= 1.5e-7, at fp32 resolution) needs exp(-z^2) = exp(-x^2 / 2), which is also the Gaussian of gelu'.
// ~14 VALU instructions against ~40 of erff() -- the activation epilogues were a third of the forward kernel.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x);
    const float e = __expf(-0.5f * x * x);
    const float t = __frcp_rn(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erf_abs = fmaf(-poly * t, e, 1.0f);            // erf(|x| / sqrt 2)
    cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
    pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return fmaf(x, pdf, cdf);
}

// diagnostic build (-DFNO_STAMP, tools/fno_stamp_run.py): shader-clock stamps of workgroup 0 at the phase boundaries
#ifdef FNO_STAMP
__device__ unsigned long long g_stamps[256];
#define STAMP(k)                                                                      \
    do {                                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[(k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

struct Lds {
    float *xs, *tb, *S, *Z, *wps, *gv, *misc;
    int NP;
    __device__ Lds(float* base, int N) {
        NP = N + 4;
        xs = base;
        tb = xs + C * NP;
        S = tb + C * NP;           // [KS][32][KP]
        Z = S + KS * C * KP;       // [32][KP]
        wps = Z + C * KP;          // [32][KP]
        gv = wps + C * KP;         // [N]
        misc = gv + N;             // [256]
    }
    static size_t floats(int N) { return 2 * (size_t)C * (N + 4) + (KS + 2) * C * KP + (size_t)N + 256; }
};

// tab[k][n], k < 2M: k < M -> cos(2 pi k n / N), else -sin(2 pi (k - M) n / N).  One matrix serves the forward DFT
// (B[n][k]) and the inverse (B[k][n]); rows padded like the activations, so both gathers are bank-conflict free.
// The matrix is computed ONCE per (device, N) into global memory (twiddle_init_kernel, host cache below) and copied into
// LDS with 16-byte loads: rebuilding it from a cosine table cost 8 k of a forward evaluation's 165 k clocks, and the
// backward pass needs it four times per evaluation (its buffer doubles as the layer-input buffer).
__global__ void twiddle_init_kernel(float* tab, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K2 * N) return;
    const int k = i / N, n = i - k * N;
    const bool is_sin = k >= M;
    const int mode = is_sin ? k - M : k;
    const int idx = (int)(((long)mode * n) % N);          // exact argument reduction in integers
    const float ang = 2.0f * (float)idx / (float)N;       // in units of pi
    tab[i] = is_sin ? -sinpif(ang) : cospif(ang);
}
__device__ __forceinline__ void load_table(float* tab, const float* __restrict__ tabg, int N, int NP) {
    const int sh = __ffs(N) - 1;
    for (int i = threadIdx.x; i < (K2 * N) >> 2; i += blockDim.x) {
        const int e = i << 2, k = e >> sh, n = e & (N - 1);
        *reinterpret_cast<float4*>(tab + k * NP + n) = reinterpret_cast<const float4*>(tabg)[i];
    }
}

// S[ks][32][KP] (+)= A[32][N] . B^T with B given as rows: out[row][col] = sum_n arows[row][n] * brows[col][n].
// 4 output tiles x KS halves of the contraction = 8 units, one per wave.
__device__ __forceinline__ void contract_n(const float* arows, const float* brows, float* S, int N, int NP) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    for (int t = wave; t < 4 * KS; t += nwaves) {
        const int tile = t & 3, ks = t >> 2, rt = tile >> 1, ct = tile & 1;
        const int span = N / KS, kbeg = ks * span;
        const float* arow = arows + (16 * rt + r) * NP + kbeg;
        const float* brow = brows + (16 * ct + r) * NP + kbeg;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < span; k0 += 16) {
            const float a0 = arow[k0 + q], a1 = arow[k0 + 4 + q], a2 = arow[k0 + 8 + q], a3 = arow[k0 + 12 + q];
            const float b0 = brow[k0 + q], b1 = brow[k0 + 4 + q], b2 = brow[k0 + 8 + q], b3 = brow[k0 + 12 + q];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc3, 0, 0, 0);
        }
        float* dst = S + ks * C * KP;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[(16 * rt + 4 * q + j) * KP + 16 * ct + r] = (acc0[j] + acc1[j]) + (acc2[j] + acc3[j]);
    }
}

__device__ __forceinline__ float s_sum(const float* S, int idx) {
    float v = S[idx];
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) v += S[ks * C * KP + idx];
    return v;
}

// [32][32] row-major matrix from global memory into a padded LDS matrix
__device__ __forceinline__ void load_mat(float* dst, const float* src) {
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) dst[(i >> 5) * KP + (i & 31)] = src[i];
}

// x0[c][n] = lift_w[c][0] u[n] + lift_w[c][1] a[n] + lift_b[c]
__device__ __forceinline__ void lift_into(float* xs, const float* u, const float* act, const Weights& w, int N, int NP) {
    const int sh = __ffs(N) - 1;
    for (int i = threadIdx.x; i < C * N; i += blockDim.x) {
        const int c = i >> sh, n = i & (N - 1);
        xs[c * NP + n] = fmaf(w.lift_w[2 * c], u[n], fmaf(w.lift_w[2 * c + 1], act[n], w.lift_b[c]));
    }
}

// Complex mode mixing of one pair, 512 threads: thread (row, m4, q4) owns 4 consecutive modes of output row `row` and a
// quarter of the 32-term contraction (16-byte weight loads: a quarter of the load instructions of one (row, mode) per
// thread; this phase is latency bound), the quarters are folded by lane shuffles.
//   forward  (CONJ = false): out[o][m] = s_m sum_i in[i][m] W[i][o][m]          in = S (summed over its K-split halves)
//   backward (CONJ = true) : out[i][m] =     sum_o in[o][m] conj(W[i][o][m])    in = Z (one copy)
// Spectra rows are [re 0..15 | im 0..15].
struct MixRegs {
    float4 vr[8], vi[8];
};
// the thread's 16 weight loads (issued BEFORE the contraction that produces the spectrum: L2 latency under MFMA work)
template <bool CONJ>
__device__ __forceinline__ void mix_load(const float* __restrict__ wr, const float* __restrict__ wi, MixRegs& m) {
    const int t = threadIdx.x, q4 = t & 3, m4 = (t >> 2) & 3, row = t >> 4;
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const int c = 8 * q4 + c8;                       // the contracted channel
        const size_t widx = CONJ ? ((size_t)row * C + c) * M + 4 * m4 : ((size_t)c * C + row) * M + 4 * m4;
        m.vr[c8] = *reinterpret_cast<const float4*>(wr + widx);
        m.vi[c8] = *reinterpret_cast<const float4*>(wi + widx);
    }
}
template <bool CONJ>
__device__ __forceinline__ void mix_compute(const MixRegs& w, const float* in, float* out, float s0, float s1) {
    const int t = threadIdx.x, q4 = t & 3, m4 = (t >> 2) & 3, row = t >> 4;
    float yr[4] = {0.f, 0.f, 0.f, 0.f}, yi[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const int c = 8 * q4 + c8;
        float xr[4], xi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = 4 * m4 + j;
            xr[j] = CONJ ? in[c * KP + m] : s_sum(in, c * KP + m);
            xi[j] = CONJ ? in[c * KP + M + m] : s_sum(in, c * KP + M + m);
        }
        const float wrv[4] = {w.vr[c8].x, w.vr[c8].y, w.vr[c8].z, w.vr[c8].w};
        const float wiv[4] = {w.vi[c8].x, w.vi[c8].y, w.vi[c8].z, w.vi[c8].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (CONJ) {
                yr[j] = fmaf(xr[j], wrv[j], fmaf(xi[j], wiv[j], yr[j]));
                yi[j] = fmaf(xi[j], wrv[j], fmaf(-xr[j], wiv[j], yi[j]));
            } else {
                yr[j] = fmaf(xr[j], wrv[j], fmaf(-xi[j], wiv[j], yr[j]));
                yi[j] = fmaf(xr[j], wiv[j], fmaf(xi[j], wrv[j], yi[j]));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        yr[j] += __shfl_xor(yr[j], 1, 64);
        yi[j] += __shfl_xor(yi[j], 1, 64);
        yr[j] += __shfl_xor(yr[j], 2, 64);
        yi[j] += __shfl_xor(yi[j], 2, 64);
    }
    if (q4 == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = 4 * m4 + j;
            const float s = m == 0 ? s0 : s1;
            out[row * KP + m] = s * yr[j];
            out[row * KP + M + m] = s * yi[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
template <bool SAVE>
__global__ void __launch_bounds__(TPB) fno_forward_kernel(const FwdArgs a) {
    extern __shared__ __align__(16) float lds_raw[];
    const int N = a.n;
    const Lds L(lds_raw, N);
    const int NP = L.NP;
    const int p = blockIdx.x, t = p / a.nb, b = p - t * a.nb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const float* u = a.u.row(t, b);
    const float* act = a.act.row(t, b);

    STAMP(0);
    lift_into(L.xs, u, act, a.w, N, NP);
    STAMP(1);
    load_table(L.tb, a.tabg, N, NP);
    __syncthreads();
    STAMP(2);

    const float s0 = 1.0f / (float)N, s1 = 2.0f / (float)N;
    for (int l = 0; l < LAYERS; ++l) {
        // ---- A: truncated DFT of the layer input; the pointwise weights ride along -----------------------------
        STAMP(3 + 4 * l);
        MixRegs mw;
        mix_load<false>(a.w.wr[l], a.w.wi[l], mw);
        contract_n(L.xs, L.tb, L.S, N, NP);
        load_mat(L.wps, a.w.pw[l]);
        if (threadIdx.x < C) L.misc[threadIdx.x] = a.w.pb[l][threadIdx.x];
        __syncthreads();
        STAMP(4 + 4 * l);
        // ---- B: spectrum out (for the weight gradient), complex mode mixing -------------------------------------
        if (SAVE) {
            for (int i = threadIdx.x; i < C * K2; i += blockDim.x) {
                const int k = i >> 5, c = i & 31;
                a.xspec[(((size_t)l * K2 + k) * a.spec_pairs + a.spec_pair0 + p) * C + c] = s_sum(L.S, c * KP + k);
            }
        }
        mix_compute<false>(mw, L.S, L.Z, s0, s1);
        __syncthreads();
        STAMP(5 + 4 * l);
        // ---- C: pre = iDFT(Z) + Wp x + b, position tile by position tile, in place ------------------------------
        float* pre_out = SAVE ? a.pre + ((size_t)p * LAYERS + l) * C * N : nullptr;
        for (int ct = wave; ct < (N >> 4); ct += nwaves) {
            const int pos = 16 * ct + r;
            f32x4 acc[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                acc[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* zrow = L.Z + (16 * rt + r) * KP;
                const float* wrow = L.wps + (16 * rt + r) * KP;
#pragma unroll
                for (int k0 = 0; k0 < K2; k0 += 4) {
                    acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(zrow[k0 + q], L.tb[(k0 + q) * NP + pos], acc[rt][0], 0, 0, 0);
                    acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[k0 + q], L.xs[(k0 + q) * NP + pos], acc[rt][1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 16 * rt + 4 * q + j;
                    const float v = (acc[rt][0][j] + acc[rt][1][j]) + L.misc[row];
                    if (SAVE) pre_out[(size_t)row * N + pos] = v;
                    L.xs[row * NP + pos] = l + 1 < LAYERS ? gelu(v) : v;
                }
        }
        __syncthreads();
        STAMP(6 + 4 * l);
    }

    // ---- project: delta = p2 . gelu(p1 h + b1) + b2;  out = u + cscale * delta + cshift ---------------------------
    STAMP(19);
    load_mat(L.wps, a.w.p1_w);
    if (threadIdx.x < C) {
        L.misc[threadIdx.x] = a.w.p1_b[threadIdx.x];
        L.misc[C + threadIdx.x] = a.w.p2_w[threadIdx.x];
    }
    __syncthreads();
    const float b2 = a.w.p2_b[0];
    for (int ct = wave; ct < (N >> 4); ct += nwaves) {
        const int pos = 16 * ct + r;
        float part = 0.0f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* wrow = L.wps + (16 * rt + r) * KP;
#pragma unroll
            for (int k0 = 0; k0 < C; k0 += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[k0 + q], L.xs[(k0 + q) * NP + pos], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 16 * rt + 4 * q + j;
                part = fmaf(L.misc[C + row], gelu(acc[j] + L.misc[row]), part);
            }
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (q == 0) {
            const float d = part + b2;
            a.delta[(size_t)p * N + pos] = d;
            if (a.out) a.out[(size_t)p * N + pos] = fmaf(a.cscale, d, u[pos]) + a.cshift;
        }
    }
    STAMP(20);
}

// ------------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------------
// rowsum[row] = sum_n buf[row][n] * (v ? v[n] : 1), 16 threads per row (512 threads = 32 rows)
__device__ __forceinline__ float row_dot(const float* buf, const float* v, int N, int NP) {
    const int row = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    float s = 0.0f;
    if (row < C) {
        const float* src = buf + row * NP;
        if (v) {
            for (int n = l16; n < N; n += 16) s = fmaf(src[n], v[n], s);
        } else {
            for (int n = l16; n < N; n += 16) s += src[n];
        }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    return s;    // valid on every lane of the row's 16-lane group
}

__global__ void __launch_bounds__(TPB) fno_backward_kernel(const BwdArgs a) {
    extern __shared__ __align__(16) float lds_raw[];
    const int N = a.n;
    const Lds L(lds_raw, N);
    const int NP = L.NP;
    float* db = L.tb;           // the gradient buffer lives where the forward pass keeps its table
    const int p = blockIdx.x, t = p / a.nb, b = p - t * a.nb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int sh = __ffs(N) - 1;
    const float* u = a.u.row(t, b);
    const float* act = a.act.row(t, b);
    float* grow = a.rows + (size_t)p * ROW;
    const float* pre_p = a.pre + (size_t)p * LAYERS * C * N;
    const bool has_gout = a.gout && t == a.gout_t;

    // ---- g = d loss / d delta (+ cscale * d loss / d out);  h = pre_3;  W1, b1, W2 --------------------------------
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float g = a.gdelta[(size_t)p * N + n];
        if (has_gout) g = fmaf(a.cscale, a.gout[(size_t)b * N + n], g);
        L.gv[n] = g;
    }
    {
        const float4* src = reinterpret_cast<const float4*>(pre_p + (size_t)(LAYERS - 1) * C * N);
        for (int i = threadIdx.x; i < (C * N) >> 2; i += blockDim.x) {
            const int e = i << 2, c = e >> sh, n = e & (N - 1);
            *reinterpret_cast<float4*>(L.xs + c * NP + n) = src[i];
        }
    }
    load_mat(L.wps, a.w.p1_w);
    if (threadIdx.x < C) {
        L.misc[threadIdx.x] = a.w.p1_b[threadIdx.x];
        L.misc[C + threadIdx.x] = a.w.p2_w[threadIdx.x];
        L.misc[2 * C + threadIdx.x] = 0.0f;        // dW2 accumulators
    }
    __syncthreads();

    // ---- project backward, position-tile local: d_z1 -> db; dW2 partial sums --------------------------------------
    {
        float dw2[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int ct = wave; ct < (N >> 4); ct += nwaves) {
            const int pos = 16 * ct + r;
            const float g = L.gv[pos];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* wrow = L.wps + (16 * rt + r) * KP;
#pragma unroll
                for (int k0 = 0; k0 < C; k0 += 4)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[k0 + q], L.xs[(k0 + q) * NP + pos], acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 16 * rt + 4 * q + j;
                    const float z1 = acc[j] + L.misc[row];
                    dw2[rt][j] = fmaf(g, gelu(z1), dw2[rt][j]);
                    db[row * NP + pos] = L.misc[C + row] * g * gelu_grad(z1);
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = dw2[rt][j];
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 8, 64);
                if (r == 0) atomicAdd(&L.misc[2 * C + 16 * rt + 4 * q + j], v);   // LDS, 8 waves: order-insensitive to 1 ulp
            }
    }
    __syncthreads();
    // dW1 = d_z1 . h^T (K = N), db1 = rowsum(d_z1), dW2, db2
    contract_n(db, L.xs, L.S, N, NP);
    {
        const float s = row_dot(db, nullptr, N, NP);
        if ((threadIdx.x & 15) == 0 && (threadIdx.x >> 4) < C) grow[R_P1B + (threadIdx.x >> 4)] = s;
        if (threadIdx.x < C) grow[R_P2W + threadIdx.x] = L.misc[2 * C + threadIdx.x];
        if (wave == 0) {
            float g = 0.0f;
            for (int n = lane; n < N; n += 64) g += L.gv[n];
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) g += __shfl_xor(g, m, 64);
            if (lane == 0) grow[R_P2B] = g;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) grow[R_P1W + i] = s_sum(L.S, (i >> 5) * KP + (i & 31));
    // d_h = W1^T d_z1, in place (position-tile local)
    for (int ct = wave; ct < (N >> 4); ct += nwaves) {
        const int pos = 16 * ct + r;
        f32x4 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k0 = 0; k0 < C; k0 += 4)      // A[row i][k o] = W1[o][i]
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(L.wps[(k0 + q) * KP + 16 * rt + r], db[(k0 + q) * NP + pos], acc[rt], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) db[(16 * rt + 4 * q + j) * NP + pos] = acc[rt][j];
    }
    __syncthreads();

    const float s0 = 1.0f / (float)N, s1 = 2.0f / (float)N;
    for (int l = LAYERS - 1; l >= 0; --l) {
        // ---- d_pre = d_y * gelu'(pre_l) (the last layer has no activation); xs <- the layer's input x_l -----------
        if (l + 1 < LAYERS) {
            const float4* src = reinterpret_cast<const float4*>(pre_p + (size_t)l * C * N);
            for (int i = threadIdx.x; i < (C * N) >> 2; i += blockDim.x) {
                const int e = i << 2, c = e >> sh, n = e & (N - 1);
                const float4 v = src[i];
                float4* d = reinterpret_cast<float4*>(db + c * NP + n);
                float4 g = *d;
                g.x *= gelu_grad(v.x);
                g.y *= gelu_grad(v.y);
                g.z *= gelu_grad(v.z);
                g.w *= gelu_grad(v.w);
                *d = g;
            }
        }
        if (l > 0) {
            const float4* src = reinterpret_cast<const float4*>(pre_p + (size_t)(l - 1) * C * N);
            for (int i = threadIdx.x; i < (C * N) >> 2; i += blockDim.x) {
                const int e = i << 2, c = e >> sh, n = e & (N - 1);
                float4 v = src[i];
                v.x = gelu(v.x);
                v.y = gelu(v.y);
                v.z = gelu(v.z);
                v.w = gelu(v.w);
                *reinterpret_cast<float4*>(L.xs + c * NP + n) = v;
            }
        } else {
            lift_into(L.xs, u, act, a.w, N, NP);
        }
        load_mat(L.wps, a.w.pw[l]);
        __syncthreads();
        // ---- dWp = d_pre . x^T (K = N), dbp = rowsum(d_pre) ------------------------------------------------------
        contract_n(db, L.xs, L.S, N, NP);
        {
            const float s = row_dot(db, nullptr, N, NP);
            if ((threadIdx.x & 15) == 0 && (threadIdx.x >> 4) < C) grow[R_LAYER + l * R_LAYER_SZ + 1024 + (threadIdx.x >> 4)] = s;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < C * C; i += blockDim.x)
            grow[R_LAYER + l * R_LAYER_SZ + i] = s_sum(L.S, (i >> 5) * KP + (i & 31));
        load_table(L.xs, a.tabg, N, NP);           // x_l is dead: its buffer becomes the twiddle table
        __syncthreads();
        // ---- G = s (.) DFT(d_pre) -------------------------------------------------------------------------------
        MixRegs mw;
        mix_load<true>(a.w.wr[l], a.w.wi[l], mw);
        contract_n(db, L.xs, L.S, N, NP);
        __syncthreads();
        for (int i = threadIdx.x; i < C * K2; i += blockDim.x) {
            const int k = i >> 5, c = i & 31, m = k & (M - 1);
            const float g = s_sum(L.S, c * KP + k) * (m == 0 ? s0 : s1);
            L.Z[c * KP + k] = g;                   // scaled spectrum parked in Z for the mixing below
            a.gspec[(((size_t)l * K2 + k) * a.spec_pairs + a.spec_pair0 + p) * C + c] = g;
        }
        __syncthreads();
        // ---- GX[i][m] = sum_o G[o][m] conj(W[i][o][m])  -> S[0] ----------------------------------------------------
        mix_compute<true>(mw, L.Z, L.S, 1.0f, 1.0f);
        __syncthreads();
        // ---- dx = iDFT(GX) + Wp^T d_pre, in place (position-tile local) ------------------------------------------
        for (int ct = wave; ct < (N >> 4); ct += nwaves) {
            const int pos = 16 * ct + r;
            f32x4 acc[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                acc[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* zrow = L.S + (16 * rt + r) * KP;
#pragma unroll
                for (int k0 = 0; k0 < K2; k0 += 4) {
                    acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(zrow[k0 + q], L.xs[(k0 + q) * NP + pos], acc[rt][0], 0, 0, 0);
                    acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(L.wps[(k0 + q) * KP + 16 * rt + r], db[(k0 + q) * NP + pos], acc[rt][1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int j = 0; j < 4; ++j) db[(16 * rt + 4 * q + j) * NP + pos] = acc[rt][0][j] + acc[rt][1][j];
        }
        __syncthreads();
    }

    // ---- lift backward: db = d x0 ----------------------------------------------------------------------------------
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        L.gv[n] = u[n];
        L.Z[n] = act[n];          // Z / wps are contiguous and dead: N <= 2 * 32 * KP floats
    }
    if (threadIdx.x < C) L.misc[threadIdx.x] = a.w.lift_w[2 * threadIdx.x];
    __syncthreads();
    {
        const float su = row_dot(db, L.gv, N, NP), sa = row_dot(db, L.Z, N, NP), sb = row_dot(db, nullptr, N, NP);
        if ((threadIdx.x & 15) == 0 && (threadIdx.x >> 4) < C) {
            const int c = threadIdx.x >> 4;
            grow[R_LIFT_W + 2 * c] = su;
            grow[R_LIFT_W + 2 * c + 1] = sa;
            grow[R_LIFT_B + c] = sb;
        }
    }
    if (a.dbase) {
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
            float s = has_gout ? a.gout[(size_t)b * N + n] : 0.0f;
#pragma unroll 8
            for (int c = 0; c < C; ++c) s = fmaf(L.misc[c], db[c * NP + n], s);
            a.dbase[(size_t)p * N + n] = s;
        }
    }
}

// out[j] = sum_p rows[p][j]   (fixed order: deterministic).  Block = 64 columns x 8 row groups: the sum over pairs is a
// latency chain, so it is cut into 8 interleaved chains per column that meet in LDS.
__global__ void __launch_bounds__(512) fno_reduce_rows_kernel(const float* rows, int pairs, int width, float* out) {
    __shared__ float part[8][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + col;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (j < width) {
        int p = grp;
        for (; p + 24 < pairs; p += 32) {
            s0 += rows[(size_t)p * width + j];
            s1 += rows[(size_t)(p + 8) * width + j];
            s2 += rows[(size_t)(p + 16) * width + j];
            s3 += rows[(size_t)(p + 24) * width + j];
        }
        for (; p < pairs; p += 8) s0 += rows[(size_t)p * width + j];
    }
    part[grp][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && j < width) {
        float v = part[0][col];
#pragma unroll
        for (int g = 1; g < 8; ++g) v += part[g][col];
        out[j] = v;
    }
}

// Spectral weight gradient of every layer from the saved spectra (layout [4][32 k][pairs][32 c]):
//   dWr[i][o][m] = sum_p Gr[p][o][m] Xr[p][i][m] + Gi Xi,   dWi[i][o][m] = sum_p Gi Xr - Gr Xi
// one workgroup per (layer, mode): four [32 x P] @ [P x 32] contractions on MFMA, K = pairs.
struct WgradOut {
    float* dwr[LAYERS];
    float* dwi[LAYERS];
};

// 16 waves per workgroup: 4 output tiles x 4 interleaved quarters of the pair axis (the contraction is a latency chain of
// 4-byte gathers: K = 1 280 pairs at B = 64, T = 20), the quarters meet in LDS.
constexpr int WG_KS = 4;
__global__ void __launch_bounds__(256 * WG_KS) fno_spec_wgrad_kernel(const float* xspec, const float* gspec, int pairs, const WgradOut dst) {
    __shared__ float part[WG_KS][4][2][256];     // [k quarter][tile][re | im][lane * 4 + j]
    const int l = blockIdx.x / M, m = blockIdx.x % M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = wave & 3, ks = wave >> 2;
    const int r = lane & 15, q = lane >> 4;
    const int rt = tile >> 1, ct = tile & 1;                  // output tile: rows i (16 rt ..), cols o (16 ct ..)
    const size_t plane = (size_t)pairs * C;
    const float* xr = xspec + ((size_t)l * K2 + m) * plane;      // [pairs][32 c]
    const float* xi = xspec + ((size_t)l * K2 + M + m) * plane;
    const float* gr = gspec + ((size_t)l * K2 + m) * plane;
    const float* gi = gspec + ((size_t)l * K2 + M + m) * plane;
    f32x4 rr = {0.f, 0.f, 0.f, 0.f}, ii = {0.f, 0.f, 0.f, 0.f}, ir = {0.f, 0.f, 0.f, 0.f}, ri = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = 4 * ks; p0 < pairs; p0 += 4 * WG_KS) {      // zero-padded operands beyond the last pair
        const int pp = p0 + q;
        const bool ok = pp < pairs;
        const size_t row = (size_t)(ok ? pp : 0) * C;
        const float axr = ok ? xr[row + 16 * rt + r] : 0.f, axi = ok ? xi[row + 16 * rt + r] : 0.f;   // A[row i][k p] = X[p][i]
        const float bgr = ok ? gr[row + 16 * ct + r] : 0.f, bgi = ok ? gi[row + 16 * ct + r] : 0.f;   // B[k p][col o] = G[p][o]
        rr = __builtin_amdgcn_mfma_f32_16x16x4f32(axr, bgr, rr, 0, 0, 0);
        ii = __builtin_amdgcn_mfma_f32_16x16x4f32(axi, bgi, ii, 0, 0, 0);
        ir = __builtin_amdgcn_mfma_f32_16x16x4f32(axr, bgi, ir, 0, 0, 0);
        ri = __builtin_amdgcn_mfma_f32_16x16x4f32(axi, bgr, ri, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        part[ks][tile][0][lane * 4 + j] = rr[j] + ii[j];
        part[ks][tile][1][lane * 4 + j] = ir[j] - ri[j];
    }
    __syncthreads();
    if (ks == 0) {
        float* outr = dst.dwr[l];
        float* outi = dst.dwi[l];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float vr = part[0][tile][0][lane * 4 + j], vi = part[0][tile][1][lane * 4 + j];
#pragma unroll
            for (int k = 1; k < WG_KS; ++k) {
                vr += part[k][tile][0][lane * 4 + j];
                vi += part[k][tile][1][lane * 4 + j];
            }
            const int i = 16 * rt + 4 * q + j, o = 16 * ct + r;
            const size_t idx = ((size_t)i * C + o) * M + m;
            outr[idx] = vr;
            outi[idx] = vi;
        }
    }
}

int check(const char* who, int n, int nb, int pairs, int width, int modes, int layers) {
    if (width != C || modes != M || layers != LAYERS)
        return fail(-4, "%s: the fused FNO kernels are built for width %d, %d modes, %d layers (got %d, %d, %d)", who, C, M, LAYERS,
                    width, modes, layers);
    if (n < 64 || n > 512 || (n & (n - 1))) return fail(-4, "%s: N = %d must be a power of two in [64, 512]", who, n);
    if (nb <= 0 || pairs <= 0 || pairs % nb) return fail(-1, "%s: pairs (%d) must be a positive multiple of the batch (%d)", who, pairs, nb);
    return 0;
}

template <typename K, typename A>
int launch(K kernel, const char* who, void* stream, int pairs, int n, const A& args) {
    const size_t lds = sizeof(float) * Lds::floats(n);
    if (lds > 160 * 1024) return fail(-4, "%s: needs %zu B of LDS", who, lds);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3(pairs), dim3(TPB), lds, (hipStream_t)stream, args);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "%s launch failed: %s", who, hipGetErrorString(e));
    return 0;
}

// twiddle matrix of (current device, N): created on first use (one tiny kernel + one synchronisation), then shared by every
// launch.  First use inside a stream capture is refused -- a warm-up call (which any capture needs anyway) creates it.
std::mutex g_tab_mutex;
std::map<std::pair<int, int>, float*> g_tabs;

const float* twiddle_table(const char* who, int n, hipStream_t stream, int* rc) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        *rc = fail(-2, "%s: hipGetDevice failed", who);
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    auto it = g_tabs.find({dev, n});
    if (it != g_tabs.end()) return it->second;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
        *rc = fail(-4, "%s: first use for N = %d inside a stream capture (run one eager call first)", who, n);
        return nullptr;
    }
    float* tab = nullptr;
    if (hipMalloc((void**)&tab, sizeof(float) * K2 * n) != hipSuccess) {
        *rc = fail(-2, "%s: hipMalloc of the twiddle matrix failed", who);
        return nullptr;
    }
    hipLaunchKernelGGL(twiddle_init_kernel, dim3((K2 * n + 255) / 256), dim3(256), 0, stream, tab, n);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
        (void)hipFree(tab);
        *rc = fail(-2, "%s: twiddle matrix initialisation failed", who);
        return nullptr;
    }
    g_tabs[{dev, n}] = tab;
    return tab;
}

Weights weights_of(const fno_weights* w) {
    Weights o;
    o.lift_w = w->lift_w;
    o.lift_b = w->lift_b;
    for (int l = 0; l < LAYERS; ++l) {
        o.wr[l] = w->spec_wr[l];
        o.wi[l] = w->spec_wi[l];
        o.pw[l] = w->pw_w[l];
        o.pb[l] = w->pw_b[l];
    }
    o.p1_w = w->p1_w;
    o.p1_b = w->p1_b;
    o.p2_w = w->p2_w;
    o.p2_b = w->p2_b;
    return o;
}

bool complete(const fno_weights* w) {
    if (!w || !w->lift_w || !w->lift_b || !w->p1_w || !w->p1_b || !w->p2_w || !w->p2_b) return false;
    for (int l = 0; l < LAYERS; ++l)
        if (!w->spec_wr[l] || !w->spec_wi[l] || !w->pw_w[l] || !w->pw_b[l]) return false;
    return true;
}

}  // namespace

extern "C" {

int fno_row_width(void) { return ROW; }

int fno_forward(void* stream, const fno_weights* w, int width, int modes, int layers, int n, int nb, int pairs, const float* u,
                long u_stride_t, long u_stride_b, const float* act, long a_stride_t, long a_stride_b, float cscale, float cshift,
                float* delta, float* out, float* pre, float* xspec, int spec_pairs, int spec_pair0) {
    if (!complete(w) || !u || !act || !delta) return fail(-1, "fno_forward: bad argument");
    if (xspec && (spec_pair0 < 0 || spec_pair0 + pairs > spec_pairs)) return fail(-1, "fno_forward: spectra window out of range");
    if (int rc = check("fno_forward", n, nb, pairs, width, modes, layers)) return rc;
    if ((pre == nullptr) != (xspec == nullptr)) return fail(-1, "fno_forward: pre and xspec are saved together");
    FwdArgs a;
    a.w = weights_of(w);
    a.u = Rows{u, u_stride_t, u_stride_b};
    a.act = Rows{act, a_stride_t, a_stride_b};
    a.delta = delta;
    a.out = out;
    a.pre = pre;
    a.xspec = xspec;
    a.spec_pairs = spec_pairs;
    a.spec_pair0 = spec_pair0;
    int trc = 0;
    a.tabg = twiddle_table("fno_forward", n, (hipStream_t)stream, &trc);
    if (!a.tabg) return trc;
    a.n = n;
    a.nb = nb;
    a.pairs = pairs;
    a.cscale = cscale;
    a.cshift = cshift;
    return pre ? launch(fno_forward_kernel<true>, "fno_forward", stream, pairs, n, a)
               : launch(fno_forward_kernel<false>, "fno_forward", stream, pairs, n, a);
}

int fno_backward(void* stream, const fno_weights* w, int width, int modes, int layers, int n, int nb, int pairs, const float* u,
                 long u_stride_t, long u_stride_b, const float* act, long a_stride_t, long a_stride_b, float cscale,
                 const float* gdelta, const float* gout, int gout_t, const float* pre, float* gspec, int spec_pairs, int spec_pair0,
                 float* rows, float* dbase) {
    if (!complete(w) || !u || !act || !gdelta || !pre || !gspec || !rows) return fail(-1, "fno_backward: bad argument");
    if (spec_pair0 < 0 || spec_pair0 + pairs > spec_pairs) return fail(-1, "fno_backward: spectra window out of range");
    if (int rc = check("fno_backward", n, nb, pairs, width, modes, layers)) return rc;
    BwdArgs a;
    a.w = weights_of(w);
    a.u = Rows{u, u_stride_t, u_stride_b};
    a.act = Rows{act, a_stride_t, a_stride_b};
    a.gdelta = gdelta;
    a.gout = gout;
    a.gout_t = gout_t;
    a.pre = pre;
    a.gspec = gspec;
    a.spec_pairs = spec_pairs;
    a.spec_pair0 = spec_pair0;
    a.rows = rows;
    a.dbase = dbase;
    int trc = 0;
    a.tabg = twiddle_table("fno_backward", n, (hipStream_t)stream, &trc);
    if (!a.tabg) return trc;
    a.n = n;
    a.nb = nb;
    a.pairs = pairs;
    a.cscale = cscale;
    return launch(fno_backward_kernel, "fno_backward", stream, pairs, n, a);
}

int fno_reduce_rows(void* stream, const float* rows, int pairs, float* out) {
    if (!rows || !out || pairs <= 0) return fail(-1, "fno_reduce_rows: bad argument");
    hipLaunchKernelGGL(fno_reduce_rows_kernel, dim3((ROW + 63) / 64), dim3(512), 0, (hipStream_t)stream, rows, pairs, ROW, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "fno_reduce_rows launch failed: %s", hipGetErrorString(e));
    return 0;
}

int fno_spec_wgrad(void* stream, const float* xspec, const float* gspec, int pairs, float* const* dwr, float* const* dwi) {
    if (!xspec || !gspec || !dwr || !dwi || pairs <= 0) return fail(-1, "fno_spec_wgrad: bad argument");
    WgradOut dst;
    for (int l = 0; l < LAYERS; ++l) {
        if (!dwr[l] || !dwi[l]) return fail(-1, "fno_spec_wgrad: NULL output");
        dst.dwr[l] = dwr[l];
        dst.dwi[l] = dwi[l];
    }
    hipLaunchKernelGGL(fno_spec_wgrad_kernel, dim3(LAYERS * M), dim3(256 * WG_KS), 0, (hipStream_t)stream, xspec, gspec, pairs, dst);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "fno_spec_wgrad launch failed: %s", hipGetErrorString(e));
    return 0;
}

const char* fno_last_error(void) { return g_err; }

#ifdef FNO_STAMP
int fno_read_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (n < 256 ? n : 256)) == hipSuccess ? 0 : -2;
}
#endif

}  // extern "C"
