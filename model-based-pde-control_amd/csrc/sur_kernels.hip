// sur_kernels.hip -- fused gfx950 kernels for the surrogate's TBPTT step (C ABI: include/surrogate_hip.h).
//
// Reference arithmetic (paths relative to the reference root):
//   pdecontrol/surrogates/models/cnn.py:126-145   ResidualBlock.forward (conv3 -> SiLU -> LN, twice; 1x1 skip; LN)
//   pdecontrol/surrogates/models/cnn.py:35-41,64-70  ConvBlock / DeConvolutionBlock (conv -> act -> LN)
//   pdecontrol/surrogates/transition.py:218-226   CNNLSTMCell.forward
//   pdecontrol/surrogates/surrogate.py:97-107     one rollout step: cell -> decoder -> integrate
//   pdecontrol/architectures/autoreg.py:51-94     channel / kernel / stride / padding choices
//
// Execution model: one workgroup (256 threads) works on one sample -- or one (rollout step, sample) pair -- at a
// time with every activation of the module in LDS and the module's weights staged into LDS once per launch.  At
// the reference's sizes (channels <= 16, widths <= 64) a layer is a few hundred to a few thousand MACs: bound by
// launch latency when run as separate kernels, so a whole module is one launch:
//   enc_fwd / enc_bwd(_multi)      3 residual blocks, persistent workgroups over the samples, 2 per CU
//   cell_fwd / cell_bwd            the ConvLSTM recurrence of a TBPTT chunk: time loop inside the kernel, weights and
//                                  hidden state in LDS across steps, one workgroup per sample
//   dec_fwd / dec_bwd / cell_wgrad everything of a chunk that is NOT recurrent (decoder, dx, LSTM weight gradients),
//                                  for all (step, sample) pairs in parallel
//   integrate, dgrad_scan, delta_loss, flush_*   prefix sums over time, the loss section, gradient reduction (+ Adam)
// Conv-like layers are gather-GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32).  Backward kernels do not recompute:
// forward kernels write every intermediate to HBM and backward kernels read it back (HBM capacity and bandwidth are
// free here, dependent-phase latency is not).  Parameter gradients are summed over space, time and the workgroup's
// samples in LDS, added to the workgroup's own row of a partial buffer (no atomics, so results are deterministic)
// and reduced over rows by a flush kernel.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/surrogate_hip.h"

// Diagnostic build only (-DSUR_STAMP): shader-clock stamps per phase for workgroup 0 of each launch,
// accumulated in a __device__ buffer nothing else reads (guide section 7, In-kernel stamps).
#ifdef SUR_STAMP
__device__ long long sur_stamp_buf[128];
__device__ long long sur_stamp_last;
#define STAMP(id)                                                                 \
    do {                                                                          \
        __syncthreads();                                                          \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                \
            const long long now_ = (long long)__builtin_amdgcn_s_memtime();       \
            sur_stamp_buf[id] += now_ - sur_stamp_last;                           \
            sur_stamp_last = now_;                                                \
        }                                                                         \
    } while (0)
#else
#define STAMP(id) do { } while (0)
#endif

namespace {

constexpr int TPB = 256;
constexpr float LN_EPS = 1e-5f;

thread_local char g_err[256] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

template <int CTRL>
__device__ __forceinline__ float dpp_rot(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, false));
}

// v + (v of lane ^ 16) / (v of lane ^ 32): the swap instructions exchange the odd rows of 16 (the upper 32 lanes) of their first
// operand with the even rows (the lower 32 lanes) of the second; on two copies of v that leaves [r0 r0 r2 r2] and [r1 r1 r3 r3]
typedef unsigned swap_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float swap_add16(float v) {
    const swap_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add32(float v) {
    const swap_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// sums of a and b over aligned groups of `lpc` lanes (16, 32 or 64); every lane gets both totals
__device__ __forceinline__ void group_sum2(float& a, float& b, int lpc) {
    a += dpp_rot<0x128>(a);  // row_ror:8
    b += dpp_rot<0x128>(b);
    a += dpp_rot<0x124>(a);  // row_ror:4
    b += dpp_rot<0x124>(b);
    a += dpp_rot<0x122>(a);  // row_ror:2
    b += dpp_rot<0x122>(b);
    a += dpp_rot<0x121>(a);  // row_ror:1
    b += dpp_rot<0x121>(b);
    if (lpc >= 32) {   // lane ^ 16: v_permlane16_swap of a value with itself (gfx950) -- same operands as a ds_bpermute exchange,
        a = swap_add16(a);   // without the trip through the LDS crossbar
        b = swap_add16(b);
    }
    if (lpc >= 64) {   // lane ^ 32
        a = swap_add32(a);
        b = swap_add32(b);
    }
}

// sigmoid / tanh on the hardware transcendentals (v_exp_f32, v_rcp_f32: ~1 ulp each) instead of libm's expf / tanhf and an
// IEEE division: 5 / 8 VALU instructions instead of ~25 / ~35.  These kernels are VALU-issue bound,
// SiLU / gate non-linearities are a third of their VALU work; parity against the golden tensors is unchanged
// (loss <= 1e-5 relative, gradients <= 2e-4 of the tensor scale: tests/test_surrogate_gpu.py).
#ifndef SUR_LIBM_ACTIVATIONS
__device__ __forceinline__ float sigmoid_(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) {
    const float t = __expf(-2.0f * fabsf(x));          // in (0, 1]: no overflow for any x
    return copysignf((1.0f - t) * __frcp_rn(1.0f + t), x);
}
#else
__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return tanhf(x); }
#endif
__device__ __forceinline__ int wrapi(int j, int n) { return j < 0 ? j + n : (j >= n ? j - n : j); }


// ---------------------------------------------------------------------------------------------
// block-cooperative layer primitives on LDS-resident activations [C][H] (row-major, H contiguous).
// Every primitive ends with __syncthreads().
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// Every convolution-like layer (forward, data-gradient, weight-gradient; strided circular Conv1d and
// zero-padded ConvTranspose1d) is a small GEMM  C[M][N] (+)= sum_k A(m,k) * B(k,n)  whose operands are
// *gathered* from LDS-resident activations / weights through index functors.  The GEMM itself runs
// on the matrix cores with v_mfma_f32_16x16x4_f32 (exact fp32: bit-for-bit an fmaf chain in k order),
// one 16x16 output tile per wavefront at a time, so a layer costs ~2 LDS reads + 1 MFMA per 1024
// MACs instead of ~5 VALU/LDS instructions per 64.  Lane map (guide, section 3): A[l&15][k=l>>4],
// B[k=l>>4][l&15], result reg i of lane l = C[4*(l>>4)+i][l&15].
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f;   // a float the compiler KNOWS to be in LDS (ds_read / ds_write, 32-bit address)

// Flavour 1 -- reduction over (segment, tap, channel):
//   C[m][n] = sum_seg sum_{t<T} sum_{c<cin} A_seg[m*a_sm + c*a_sc + t] * B_seg[c*b_sc + col(seg, t, n)]
// col() returns the gathered column of B for tap t and output column n, or -1 when that tap does not
// contribute (zero padding / stride parity).  Per tap the channel loop is affine in both operands, so
// the loads of successive MFMAs are independent of each other (no div/mod/wrap inside the loop).
struct GemmSeg {
    const float* a;
    int a_sm, a_sc;
    const float* b;
    int b_sc, cin;
};

// A subset of the workgroup's wavefronts.  Independent small GEMMs are issued back to back on
// disjoint wave sets WITHOUT a barrier in between (sync = false) so that they run concurrently; the
// last one of a group (or an explicit __syncthreads()) closes the phase.
struct WaveSet {
    int lo, cnt;
};
__device__ __forceinline__ WaveSet all_waves() { return WaveSet{0, (int)(blockDim.x >> 6)}; }
__device__ __forceinline__ WaveSet lower_half() { const int nw = blockDim.x >> 6; return WaveSet{0, nw > 1 ? nw / 2 : 1}; }
__device__ __forceinline__ WaveSet upper_half() {
    const int nw = blockDim.x >> 6;
    return nw > 1 ? WaveSet{nw / 2, nw - nw / 2} : WaveSet{0, 1};
}

template <int T, int NSEG, class ColFn, class PutFn>
__device__ __forceinline__ void gemm_taps(WaveSet ws, bool sync, int M, int N, const GemmSeg (&seg)[NSEG], ColFn col,
                                          PutFn put) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (wid >= ws.lo && wid < ws.lo + ws.cnt) {
        const int nwg = blockDim.x >> 6;
        const int wave = wid - ws.lo, nw = ws.cnt < nwg ? ws.cnt : nwg;
        const int r = lane & 15, q = lane >> 4;
        const int tn_count = (N + 15) >> 4, tiles = ((M + 15) >> 4) * tn_count;
        for (int t = wave; t < tiles; t += nw) {
            const int m0 = (t / tn_count) << 4, n0 = (t % tn_count) << 4;
            const int m = m0 + r, n = n0 + r;
            const bool m_ok = m < M;
            const int m_safe = m_ok ? m : M - 1;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) {
                const GemmSeg& S = seg[sg];
#pragma unroll
                for (int tap = 0; tap < T; ++tap) {
                    const int cidx = (n < N) ? col(sg, tap, n) : -1;
                    const float* ap = S.a + m_safe * S.a_sm + tap;
                    const float* bp = S.b + (cidx >= 0 ? cidx : 0);
                    const bool b_ok = cidx >= 0;
                    if (__builtin_amdgcn_ballot_w64(b_ok) == 0) continue;   // this tap reaches none of the tile's columns
                    int c0 = 0;
                    for (; c0 + 8 <= S.cin; c0 += 8) {  // 2 MFMAs per trip on independent accumulators
                        const float a0 = ap[(c0 + q) * S.a_sc], b0 = bp[(c0 + q) * S.b_sc];
                        const float a1 = ap[(c0 + 4 + q) * S.a_sc], b1 = bp[(c0 + 4 + q) * S.b_sc];
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(m_ok ? a0 : 0.f, b_ok ? b0 : 0.f, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(m_ok ? a1 : 0.f, b_ok ? b1 : 0.f, acc1, 0, 0, 0);
                    }
                    for (; c0 < S.cin; c0 += 4) {
                        const int c = c0 + q;
                        const bool c_ok = c < S.cin;
                        const int cs_ = c_ok ? c : S.cin - 1;
                        const float a0 = ap[cs_ * S.a_sc], b0 = bp[cs_ * S.b_sc];
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((m_ok && c_ok) ? a0 : 0.f, (b_ok && c_ok) ? b0 : 0.f, acc0,
                                                                    0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) put(m0 + q * 4 + i, n0 + r, acc0[i] + acc1[i]);
        }
    }
    if (sync) __syncthreads();
}

// Flavour 2 -- reduction over positions p (P a multiple of 4):
//   C[m][n] = sum_p A[m*a_sm + p] * b_at(state(n), p);   state(n) decodes the output column once.
template <class PrepFn, class FetchFn, class PutFn>
__device__ __forceinline__ void gemm_pos(WaveSet ws, bool sync, int M, int N, int P, const float* a, int a_sm,
                                         PrepFn prep, FetchFn fetch, PutFn put, int t_lo = 0, int t_hi = 1 << 30) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (wid >= ws.lo && wid < ws.lo + ws.cnt) {
        const int nwg = blockDim.x >> 6;
        const int wave = wid - ws.lo, nw = ws.cnt < nwg ? ws.cnt : nwg;
        const int r = lane & 15, q = lane >> 4;
        const int tn_count = (N + 15) >> 4, tiles_all = ((M + 15) >> 4) * tn_count;
        const int tiles = tiles_all < t_hi ? tiles_all : t_hi;  // only tiles [t_lo, t_hi): lets a caller share one GEMM between wave sets
        for (int t = t_lo + wave; t < tiles; t += nw) {
            const int m0 = (t / tn_count) << 4, n0 = (t % tn_count) << 4;
            const int m = m0 + r, n = n0 + r;
            const bool m_ok = m < M, n_ok = n < N;
            const float* ap = a + (m_ok ? m : M - 1) * a_sm;
            const auto st = prep(n_ok ? n : N - 1);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            int p = 0;
            for (; p + 8 <= P; p += 8) {
                const float a0 = ap[p + q], b0 = fetch(st, p + q);
                const float a1 = ap[p + 4 + q], b1 = fetch(st, p + 4 + q);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(m_ok ? a0 : 0.f, n_ok ? b0 : 0.f, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(m_ok ? a1 : 0.f, n_ok ? b1 : 0.f, acc1, 0, 0, 0);
            }
            for (; p < P; p += 4) {
                const float a0 = ap[p + q], b0 = fetch(st, p + q);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(m_ok ? a0 : 0.f, n_ok ? b0 : 0.f, acc0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) put(m0 + q * 4 + i, n0 + r, acc0[i] + acc1[i]);
        }
    }
    if (sync) __syncthreads();
}

// circular Conv1d forward: out[o][p] (+)= bias[o] + sum_{ci,k} W[o][ci][k] * in[ci][(p*stride + k - pad) mod hin]
template <int K>
__device__ void conv_fwd(const float* in, int cin, int hin, const float* W, const float* bias, int cout, int stride,
                         int pad, float* out, bool accumulate, WaveSet ws = WaveSet{0, 1 << 20}, bool sync = true) {
    const int hout = hin / stride;
    if (cout == 1) {  // a single output row would waste 15/16 of an MFMA tile: split-lane VALU reduction
        const int total = hout;
        int split = 1;
        while (split < 8 && total * split * 2 <= (int)blockDim.x && cin % (split * 2) == 0) split *= 2;
        const int cper = cin / split;
        for (int base = 0; base < total * split; base += blockDim.x) {
            const int t = base + threadIdx.x;
            const bool live = t < total * split;
            const int p = live ? t / split : 0, part = t % split;
            float acc = 0.0f;
            for (int ci = part * cper; ci < (part + 1) * cper; ++ci) {
#pragma unroll
                for (int k = 0; k < K; ++k) acc = fmaf(W[ci * K + k], in[ci * hin + wrapi(p * stride + k - pad, hin)], acc);
            }
            for (int m = 1; m < split; m <<= 1) acc += __shfl_xor(acc, m, 64);
            if (live && part == 0) {
                if (bias) acc += bias[0];
                out[p] = accumulate ? out[p] + acc : acc;
            }
        }
        if (sync) __syncthreads();
        return;
    }
    const GemmSeg seg[1] = {{W, cin * K, K, in, hin, cin}};
    gemm_taps<K, 1>(ws, sync, cout, hout, seg, [&](int, int tap, int n) { return wrapi(n * stride + tap - pad, hin); },
                    [&](int m, int n, float v) {
                        if (m < cout) {
                            if (bias) v += bias[m];
                            out[m * hout + n] = accumulate ? out[m * hout + n] + v : v;
                        }
                    });
}

// din[ci][j] (+)= sum_{o,k : (p*stride + k - pad) mod hin == j} W[o][ci][k] * dout[o][p]
template <int K>
__device__ void conv_bwd_data(const float* dout, int cout, int hin, const float* W, int cin, int stride, int pad,
                              float* din, bool accumulate, WaveSet ws = WaveSet{0, 1 << 20}, bool sync = true) {
    const int hout = hin / stride;
    if (cout == 1 && stride == 1 && cin == 1) {   // one K-tap stencil: a position per thread on the VALU
        const int wid = threadIdx.x >> 6, nwg = blockDim.x >> 6;
        const int nw = ws.cnt < nwg ? ws.cnt : nwg;
        if (wid >= ws.lo && wid < ws.lo + nw)
            for (int j = threadIdx.x - 64 * ws.lo; j < hin; j += 64 * nw) {
                float acc = accumulate ? din[j] : 0.0f;
#pragma unroll
                for (int k = 0; k < K; ++k) acc = fmaf(W[k], dout[wrapi(j - k + pad, hin)], acc);
                din[j] = acc;
            }
        if (sync) __syncthreads();
        return;
    }
    if (cout == 1 && stride == 1 && cin <= 16 && K <= 8 && (hin & 15) == 0) {
        // din[ci][j] = sum_k W[ci][k] * dout[(j - k + pad) mod hin]: the TAPS are the reduction dimension -- rows = input
        // channels, columns = 16 positions, ceil(K / 4) MFMA steps per tile.  (The generic gather-GEMM reduces over
        // (output channel, tap) with the channels four at a time: with one output channel K steps per tile, three of
        // four k-slots empty -- 7 k cycles for the decoder's 7-tap layer at N = 256.)
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwg = blockDim.x >> 6;
        const int nw = ws.cnt < nwg ? ws.cnt : nwg;
        if (wid >= ws.lo && wid < ws.lo + nw) {
            const int r = lane & 15, q = lane >> 4;
            const float w0 = (r < cin && q < K) ? W[r * K + q] : 0.0f;
            const float w1 = (r < cin && 4 + q < K) ? W[r * K + 4 + q] : 0.0f;
            for (int t = wid - ws.lo; t < (hin >> 4); t += nw) {
                const int j = 16 * t + r;
                const float d0 = q < K ? dout[wrapi(j - q + pad, hin)] : 0.0f;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, d0, acc, 0, 0, 0);
                if (K > 4) {
                    const float d1 = 4 + q < K ? dout[wrapi(j - 4 - q + pad, hin)] : 0.0f;
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, d1, acc, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ci = 4 * q + i;
                    if (ci < cin) din[ci * hin + j] = accumulate ? din[ci * hin + j] + acc[i] : acc[i];
                }
            }
        }
        if (sync) __syncthreads();
        return;
    }
    const GemmSeg seg[1] = {{W, K, cin * K, dout, hout, cout}};  // A[m=ci][c=o][tap] = W[(o*cin + ci)*K + tap]
    // stride 2: an input position receives from the taps of one parity only.  As in deconv_fwd the GEMM's columns are the even
    // positions followed by the odd ones (hin a multiple of 32), so a tile of 16 is of one parity and gemm_taps skips the taps
    // that reach none of its columns (half of the MFMA steps of the strided block's data gradients).
    const bool sorted = stride == 2 && (hin & 31) == 0;
    const int half = hin >> 1;
    auto pos = [&](int n) { return sorted ? (n < half ? 2 * n : 2 * (n - half) + 1) : n; };
    gemm_taps<K, 1>(ws, sync, cin, hin, seg,
                    [&](int, int tap, int n) {
                        const int t = wrapi(pos(n) - tap + pad, hin);
                        return (t % stride) ? -1 : t / stride;
                    },
                    [&](int m, int n, float v) {
                        const int j = pos(n);
                        if (m < cin) din[m * hin + j] = accumulate ? din[m * hin + j] + v : v;
                    });
}

// gW[o][ci][k] += sum_p dout[o][p] * in[ci][(p*stride + k - pad) mod hin];  gb[o] += sum_p dout[o][p]
// `scratch` (LDS, scratch_floats floats, not aliasing any operand): lets the single-output-channel case run as an MFMA tile.
// GP: the accumulators' pointer type -- `lds_f*` when the kernel knows them to be in LDS (plain ds_read / ds_write), `float*` when
// they may be the partial row in global memory (a generic pointer: flat_load / flat_store, whose s_waitcnt also drains every
// outstanding global load and store of the wave).
template <int K, class GP>
__device__ void conv_bwd_weight(const float* dout, int cout, const float* in, int cin, int hin, int stride, int pad,
                                GP gW, GP gb, WaveSet ws = WaveSet{0, 1 << 20}, bool sync = true, float* scratch = nullptr,
                                int scratch_floats = 0) {
    const int hout = hin / stride, ncols = cin * K;
    if (cout == 1 && stride == 1 && K <= 16 && cin < 16 && (cin + 1) * K <= 64 && scratch_floats >= 64 && (hin & 3) == 0) {
        // gW[ci][k] = sum_j dout[(j - k + pad) mod hin] * in[ci][j]: ONE tile with the taps as rows, the input channels (and
        // a column of ones: the bias) as columns, reduced over the input positions four at a time -- the positions split
        // over the waves, their partial tiles added in wave order through `scratch`.  hin / 16 MFMA steps per wave where the
        // 16-lane dot products below took 13 k cycles for the decoder's 7-tap layer at N = 256.
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwg = blockDim.x >> 6;
        const int r = lane & 15, q = lane >> 4;
        int nw = ws.cnt < nwg ? ws.cnt : nwg;
        if (nw > scratch_floats >> 6) nw = scratch_floats >> 6;
        const int steps = hin >> 2, per = (steps + nw - 1) / nw;
        if (wid >= ws.lo && wid < ws.lo + nw) {
            const int wv = wid - ws.lo;
            const bool a_ok = r < K, b_in = r < cin;
            const float b_const = (r == cin && gb) ? 1.0f : 0.0f;
            const float* brow = in + (b_in ? r : 0) * hin;
            const int shift = pad - (a_ok ? r : 0);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const int s1 = (wv + 1) * per < steps ? (wv + 1) * per : steps;
            int st = wv * per;
            for (; st + 2 <= s1; st += 2) {
                const int ja = 4 * st + q, jb = ja + 4;
                const float a0 = dout[wrapi(ja + shift, hin)], b0 = b_in ? brow[ja] : b_const;
                const float a1 = dout[wrapi(jb + shift, hin)], b1 = b_in ? brow[jb] : b_const;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? a0 : 0.f, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? a1 : 0.f, b1, acc1, 0, 0, 0);
            }
            if (st < s1) {
                const int ja = 4 * st + q;
                const float a0 = dout[wrapi(ja + shift, hin)], b0 = b_in ? brow[ja] : b_const;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? a0 : 0.f, b0, acc0, 0, 0, 0);
            }
            if (r <= cin) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * q + i < K) scratch[wv * 64 + r * K + 4 * q + i] = acc0[i] + acc1[i];
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < ncols + (gb ? 1 : 0); t += blockDim.x) {
            float v = 0.0f;
            for (int wv = 0; wv < nw; ++wv) v += scratch[wv * 64 + t];
            if (t < ncols) gW[t] += v;
            else gb[0] += v;      // t = cin * K: row 0 of the ones column
        }
        if (sync) __syncthreads();
        return;
    }
    if (cout == 1) {
        // ncols + 1 dot products of length hout (the weight taps and the bias): one per 16-lane group, positions
        // strided over the group's lanes, partial sums folded with DPP row rotations.  (One thread per dot product
        // with a serial loop over hout cost 8 k cycles per call: the slowest phase of the decoder backward.)
        const int group = threadIdx.x >> 4, gl = threadIdx.x & 15, ngroups = blockDim.x >> 4;
        const int nout = ncols + (gb ? 1 : 0);
        for (int idx0 = 0; idx0 < nout; idx0 += ngroups) {
            const int idx = idx0 + group;
            float a0 = 0.0f, a1 = 0.0f;
            if (idx < ncols) {
                const int ci = idx / K, k = idx - ci * K;
                const float* row = in + ci * hin;
                for (int p = gl; p < hout; p += 32) {
                    a0 = fmaf(dout[p], row[wrapi(p * stride + k - pad, hin)], a0);
                    if (p + 16 < hout) a1 = fmaf(dout[p + 16], row[wrapi((p + 16) * stride + k - pad, hin)], a1);
                }
            } else if (idx == ncols && gb) {
                for (int p = gl; p < hout; p += 32) {
                    a0 += dout[p];
                    if (p + 16 < hout) a1 += dout[p + 16];
                }
            }
            group_sum2(a0, a1, 16);
            if (gl == 0) {
                if (idx < ncols) gW[idx] += a0 + a1;
                else if (idx == ncols && gb) gb[0] += a0 + a1;
            }
        }
        if (sync) __syncthreads();
        return;
    }
    if (gb) {
        // bias gradient: one 16-lane group per output channel
        const int group = threadIdx.x >> 4, gl = threadIdx.x & 15, ngroups = blockDim.x >> 4;
        for (int o0 = 0; o0 < cout; o0 += ngroups) {
            const int o = o0 + group;
            float a0 = 0.0f, a1 = 0.0f;
            if (o < cout)
                for (int p = gl; p < hout; p += 32) {
                    a0 += dout[o * hout + p];
                    if (p + 16 < hout) a1 += dout[o * hout + p + 16];
                }
            group_sum2(a0, a1, 16);
            if (gl == 0 && o < cout) gb[o] += a0 + a1;
        }
    }
    struct St { const float* row; int off; };
    gemm_pos(ws, sync, cout, ncols, hout, dout, hout,
             [&](int n) {
                 const int ci = n / K, k = n - ci * K;
                 return St{in + ci * hin, k - pad};
             },
             [&](const St& st, int p) { return st.row[wrapi(p * stride + st.off, hin)]; },
             [&](int m, int n, float v) {
                 if (m < cout && n < ncols) gW[m * ncols + n] += v;
             });
}

// ConvTranspose1d(k=3, stride=2, padding=1, output_padding=1), zero padded; W[ci][o][k]; hout = 2*hin
// out[o][j] = b[o] + sum_{ci,k : j = 2i - 1 + k} W[ci][o][k] * in[ci][i]
__device__ void deconv_fwd(const float* in, int cin, int hin, const float* W, const float* bias, int cout,
                           float* out, WaveSet ws = WaveSet{0, 1 << 20}, bool sync = true) {
    const int hout = 2 * hin;
    const GemmSeg seg[1] = {{W, 3, cout * 3, in, hin, cin}};  // A[m=o][c=ci][tap] = W[(ci*cout + o)*3 + tap]
    // An even output position takes the middle tap only, an odd one the outer two.  With the GEMM's columns in output
    // order every tile of 16 holds both parities and issues all three taps, half of them on zeros; with the even positions
    // in the first hin columns and the odd ones behind them (hin a multiple of 16) a tile is of one parity and gemm_taps
    // skips the taps that reach none of its columns: 1 or 2 tap passes per tile instead of 3.
    const bool sorted = (hin & 15) == 0;
    auto pos = [&](int n) { return sorted ? (n < hin ? 2 * n : 2 * (n - hin) + 1) : n; };
    gemm_taps<3, 1>(ws, sync, cout, hout, seg,
                    [&](int, int tap, int n) {
                        const int t = pos(n) + 1 - tap;
                        return (t < 0 || (t & 1) || (t >> 1) >= hin) ? -1 : (t >> 1);
                    },
                    [&](int m, int n, float v) {
                        if (m < cout) out[m * hout + pos(n)] = v + bias[m];
                    });
}

__device__ void deconv_bwd_data(const float* dout, int cout, int hin, const float* W, int cin, float* din,
                                WaveSet ws = WaveSet{0, 1 << 20}, bool sync = true) {
    const int hout = 2 * hin;
    const GemmSeg seg[1] = {{W, cout * 3, 3, dout, hout, cout}};  // A[m=ci][c=o][tap] = W[(ci*cout + o)*3 + tap]
    gemm_taps<3, 1>(ws, sync, cin, hin, seg,
                    [&](int, int tap, int i) {
                        const int j = 2 * i - 1 + tap;
                        return (j < 0 || j >= hout) ? -1 : j;
                    },
                    [&](int m, int i, float v) {
                        if (m < cin) din[m * hin + i] = v;
                    });
}

template <class GP>
__device__ void deconv_bwd_weight(const float* dout, int cout, const float* in, int cin, int hin, GP gW, GP gb,
                                  WaveSet ws = WaveSet{0, 1 << 20}, bool sync = true) {
    const int hout = 2 * hin, ncols = cout * 3;
    {
        // bias gradient: one 16-lane group per output channel, positions strided over the group's lanes.  (One thread per
        // channel with a serial loop over hout was 16 k cycles at hout = 256 -- a fifth of the decoder backward -- all of
        // it in the wave that then starts on the first tile of the GEMM below.)
        const int group = threadIdx.x >> 4, gl = threadIdx.x & 15, ngroups = blockDim.x >> 4;
        for (int o0 = 0; o0 < cout; o0 += ngroups) {
            const int o = o0 + group;
            float a0 = 0.0f, a1 = 0.0f;
            if (o < cout)
                for (int j = gl; j < hout; j += 32) {
                    a0 += dout[o * hout + j];
                    if (j + 16 < hout) a1 += dout[o * hout + j + 16];
                }
            group_sum2(a0, a1, 16);
            if (gl == 0 && o < cout) gb[o] += a0 + a1;
        }
    }
    struct St { const float* row; int off; };
    gemm_pos(ws, sync, cin, ncols, hin, in, hin,
             [&](int n) {
                 const int o = n / 3, k = n - o * 3;
                 return St{dout + o * hout, k - 1};
             },
             [&](const St& st, int i) {
                 const int j = 2 * i + st.off;
                 return (j < 0 || j >= hout) ? 0.0f : st.row[j];
             },
             [&](int m, int n, float v) {
                 if (m < cin && n < ncols) gW[m * ncols + n] += v;
             });
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the spatial axis fused with the preceding activation.  A channel is handled by a
// group of LPC = min(64, H) lanes (64/LPC channels per wavefront at a time), each lane keeping its
// H/LPC activated values in registers; mean and E[y^2] are reduced TOGETHER (two interleaved chains)
// with DPP row rotations inside 16-lane rows and one ds_bpermute per doubling beyond a row.
// ---------------------------------------------------------------------------------------------
constexpr int LN_MAX_EPL = 4;  // H <= 256

// out = LayerNorm_H(act(pre)) * gamma[p] + beta[p], act = SiLU or identity
__device__ void act_ln_fwd(const float* pre, int C, int H, const float* gamma, const float* beta, bool silu,
                           float* out) {
    const int lpc = H < 64 ? H : 64, gpw = 64 / lpc, epl = H / lpc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int grp = lane / lpc, gl = lane - grp * lpc;
    const float inv_h = 1.0f / H;
    for (int c0 = 0; c0 < C; c0 += nw * gpw) {
        const int c = c0 + wave * gpw + grp;
        const bool live = c < C;
        const float* x = pre + (live ? c : 0) * H;
        float y[LN_MAX_EPL];
        float s = 0.0f, ss = 0.0f;
#pragma unroll
        for (int e = 0; e < LN_MAX_EPL; ++e) {
            if (e < epl) {
                const float v = x[gl + e * lpc];
                y[e] = silu ? v * sigmoid_(v) : v;
                s += y[e];
                ss = fmaf(y[e], y[e], ss);
            }
        }
        group_sum2(s, ss, lpc);
        const float mean = s * inv_h;
        const float var = fmaxf(fmaf(-mean, mean, ss * inv_h), 0.0f);
        const float rstd = rsqrtf(var + LN_EPS);
        if (live) {
#pragma unroll
            for (int e = 0; e < LN_MAX_EPL; ++e) {
                if (e < epl) {
                    const int p = gl + e * lpc;
                    out[c * H + p] = fmaf((y[e] - mean) * rstd, gamma[p], beta[p]);
                }
            }
        }
    }
    __syncthreads();
}

// backward of act_ln_fwd: dpre from dout; accumulates ggamma / gbeta.  xhat_scratch: [C][H] work space.
template <class GP>
__device__ void act_ln_bwd(const float* dout, const float* pre, int C, int H, const float* gamma, bool silu,
                           float* dpre, float* xhat_scratch, GP ggamma, GP gbeta) {
    const int lpc = H < 64 ? H : 64, gpw = 64 / lpc, epl = H / lpc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int grp = lane / lpc, gl = lane - grp * lpc;
    const float inv_h = 1.0f / H;
    for (int c0 = 0; c0 < C; c0 += nw * gpw) {
        const int c = c0 + wave * gpw + grp;
        const bool live = c < C;
        const int cc = live ? c : 0;
        const float* x = pre + cc * H;
        float y[LN_MAX_EPL], v_[LN_MAX_EPL], dxh[LN_MAX_EPL];
        float s = 0.0f, ss = 0.0f;
#pragma unroll
        for (int e = 0; e < LN_MAX_EPL; ++e) {
            if (e < epl) {
                const int p = gl + e * lpc;
                v_[e] = x[p];
                y[e] = silu ? v_[e] * sigmoid_(v_[e]) : v_[e];
                dxh[e] = dout[cc * H + p] * gamma[p];
                s += y[e];
                ss = fmaf(y[e], y[e], ss);
            }
        }
        group_sum2(s, ss, lpc);
        const float mean = s * inv_h;
        const float var = fmaxf(fmaf(-mean, mean, ss * inv_h), 0.0f);
        const float rstd = rsqrtf(var + LN_EPS);
        float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
        for (int e = 0; e < LN_MAX_EPL; ++e) {
            if (e < epl) {
                y[e] = (y[e] - mean) * rstd;  // xhat
                m1 += dxh[e];
                m2 = fmaf(dxh[e], y[e], m2);
            }
        }
        group_sum2(m1, m2, lpc);
        m1 *= inv_h;
        m2 *= inv_h;
        if (live) {
#pragma unroll
            for (int e = 0; e < LN_MAX_EPL; ++e) {
                if (e < epl) {
                    const int p = gl + e * lpc;
                    float dy = rstd * (dxh[e] - m1 - y[e] * m2);
                    if (silu) {
                        const float sg = sigmoid_(v_[e]);
                        dy *= sg * (1.0f + v_[e] * (1.0f - sg));
                    }
                    xhat_scratch[c * H + p] = y[e];
                    dpre[c * H + p] = dy;
                }
            }
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < H; p += blockDim.x) {
        float gg0 = 0.0f, gg1 = 0.0f, gb0 = 0.0f, gb1 = 0.0f;
        int c = 0;
        for (; c + 2 <= C; c += 2) {
            const float d0 = dout[c * H + p], d1 = dout[(c + 1) * H + p];
            gg0 = fmaf(d0, xhat_scratch[c * H + p], gg0);
            gg1 = fmaf(d1, xhat_scratch[(c + 1) * H + p], gg1);
            gb0 += d0;
            gb1 += d1;
        }
        for (; c < C; ++c) {
            const float d0 = dout[c * H + p];
            gg0 = fmaf(d0, xhat_scratch[c * H + p], gg0);
            gb0 += d0;
        }
        ggamma[p] += gg0 + gg1;
        gbeta[p] += gb0 + gb1;
    }
    __syncthreads();
}

__device__ void lds_load(float* dst, const float* __restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}
// n4 float4 global -> LDS by LDS-DMA (global_load_lds_dwordx4: one wave instruction moves 64 x 16 B, no VGPRs in between), NOT
// waited for: any number of these may be in flight; lds_dma_wait() -- or any barrier behind an s_waitcnt vmcnt(0) -- completes
// them.  dst and src 16-byte aligned.
__device__ __forceinline__ void lds_dma_v4(float* dst, const float* __restrict__ src, int n4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int j = wave; j * 64 < n4; j += nw) {
        const int i4 = j * 64 + lane;
        if (i4 < n4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * i4),
                                             (__attribute__((address_space(3))) void*)(dst + 4 * i4), 16, 0, 0);
    }
}
__device__ __forceinline__ void lds_dma_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ void lds_store(float* __restrict__ dst, const float* src, int n) {
    if (dst)
        for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// parameter staging: weights of a module are copied into LDS once per workgroup; gradient
// accumulators live either in LDS (flushed into the workgroup's partial row at the end) or, when LDS
// is too small, directly in the partial row (owner-exclusive read-modify-write, no atomics).
// ---------------------------------------------------------------------------------------------
template <int NP>
struct ParamViews {
    const float* w[NP];  // LDS copies of the weights
    float* g[NP];        // gradient accumulators (LDS or partial row); unset in forward kernels
};

template <int NP>
__device__ void stage_weights(const float* const* gw, const int* size, float* lds_w, ParamViews<NP>& v) {
    int off = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        v.w[i] = lds_w + off;
        off += size[i];
    }
    // first TPB elements of every parameter (all of it for the small ones): NP independent loads in flight at once
    float head[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) head[i] = (int)threadIdx.x < size[i] ? gw[i][threadIdx.x] : 0.0f;
#pragma unroll
    for (int i = 0; i < NP; ++i)
        if ((int)threadIdx.x < size[i]) const_cast<float*>(v.w[i])[threadIdx.x] = head[i];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        float* dst = const_cast<float*>(v.w[i]);
        for (int j = threadIdx.x + TPB; j < size[i]; j += 4 * TPB) {   // the tails of the large ones, four loads per round
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = j + u * TPB < size[i] ? gw[i][j + u * TPB] : 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j + u * TPB < size[i]) dst[j + u * TPB] = t[u];
        }
    }
    __syncthreads();
}

template <int NP>
__device__ void setup_grads(const int* size, float* base, bool zero, ParamViews<NP>& v) {
    int off = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        v.g[i] = base + off;
        off += size[i];
    }
    if (zero)
        for (int j = threadIdx.x; j < off; j += blockDim.x) base[j] = 0.0f;
    __syncthreads();
}

// row[j] += acc[j] for the workgroup's partial-gradient row.  The global loads of a batch are all issued before
// the first add/store: as a plain loop this read-modify-write paid one HBM round trip per iteration (20-30 of them).
__device__ __forceinline__ void add_to_row(float* __restrict__ row, const float* acc, int psize) {
    constexpr int U = 8;
    for (int j0 = threadIdx.x; j0 < psize; j0 += U * TPB) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * TPB;
            v[u] = j < psize ? row[j] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * TPB;
            if (j < psize) row[j] = v[u] + acc[j];
        }
    }
}

template <int NP>
__host__ __device__ inline int psize_of(const int* size) {
    int t = 0;
    for (int i = 0; i < NP; ++i) t += size[i];
    return t;
}

// ---------------------------------------------------------------------------------------------
// encoder: three residual blocks
// ---------------------------------------------------------------------------------------------
struct RBBuf {  // LDS pointers of one block's forward intermediates
    float *in, *skip, *a1pre, *a1, *a2pre, *a2, *s, *out;
    int cin, cout, hin, hout, stride;
};

__device__ void rb_forward(const RBBuf& b, const float* const* w) {
    conv_fwd<1>(b.in, b.cin, b.hin, w[SUR_RB_SKIP], nullptr, b.cout, b.stride, 0, b.skip, false, lower_half(), false);
    conv_fwd<3>(b.in, b.cin, b.hin, w[SUR_RB_CONV1], nullptr, b.cout, b.stride, 1, b.a1pre, false, upper_half(), true);
    act_ln_fwd(b.a1pre, b.cout, b.hout, w[SUR_RB_LN1_W], w[SUR_RB_LN1_B], true, b.a1);
    conv_fwd<3>(b.a1, b.cout, b.hout, w[SUR_RB_CONV2], nullptr, b.cout, 1, 1, b.a2pre, false);
    act_ln_fwd(b.a2pre, b.cout, b.hout, w[SUR_RB_LN2_W], w[SUR_RB_LN2_B], true, b.a2);
    const int n = b.cout * b.hout;
    for (int i = threadIdx.x; i < n; i += blockDim.x) b.s[i] = b.a2[i] + b.skip[i];
    __syncthreads();
    act_ln_fwd(b.s, b.cout, b.hout, w[SUR_RB_LN3_W], w[SUR_RB_LN3_B], false, b.out);
}

// dout [cout][hout] -> din [cin][hin]; g1, g2, g3, xh: scratch of cout*hout floats each
// need_din = false (the encoder's first block: its input is data): the two data-gradient GEMMs onto the block input are
// skipped -- with cin = 1 they fill one row of every 16-row MFMA tile and were the longest phase of that block.
template <class GP>
__device__ void rb_backward(const RBBuf& b, const float* const* w, const GP* g, const float* dout, float* din,
                            float* g1, float* g2, float* g3, float* xh, int sb = -1, bool need_din = true) {
#ifdef SUR_STAMP
#define RB_STAMP(i) do { if (sb >= 0) STAMP(sb + (i)); } while (0)
#else
#define RB_STAMP(i) do { } while (0)
#endif
    act_ln_bwd(dout, b.s, b.cout, b.hout, w[SUR_RB_LN3_W], false, g1, xh, g[SUR_RB_LN3_W], g[SUR_RB_LN3_B]);
    RB_STAMP(0);
    // skip path
    if (need_din) {
        conv_bwd_weight<1>(g1, b.cout, b.in, b.cin, b.hin, b.stride, 0, g[SUR_RB_SKIP], GP(nullptr), lower_half(), false);
        conv_bwd_data<1>(g1, b.cout, b.hin, w[SUR_RB_SKIP], b.cin, b.stride, 0, din, false, upper_half(), true);
    } else {
        conv_bwd_weight<1>(g1, b.cout, b.in, b.cin, b.hin, b.stride, 0, g[SUR_RB_SKIP], GP(nullptr), all_waves(), false);
        // no barrier: the next phase reads g1 / a2pre and writes g2, xh and the LayerNorm accumulators only
    }
    RB_STAMP(1);
    // residual path
    act_ln_bwd(g1, b.a2pre, b.cout, b.hout, w[SUR_RB_LN2_W], true, g2, xh, g[SUR_RB_LN2_W], g[SUR_RB_LN2_B]);
    RB_STAMP(2);
    conv_bwd_weight<3>(g2, b.cout, b.a1, b.cout, b.hout, 1, 1, g[SUR_RB_CONV2], GP(nullptr), lower_half(), false);
    conv_bwd_data<3>(g2, b.cout, b.hout, w[SUR_RB_CONV2], b.cout, 1, 1, g3, false, upper_half(), true);
    RB_STAMP(3);
    act_ln_bwd(g3, b.a1pre, b.cout, b.hout, w[SUR_RB_LN1_W], true, g1, xh, g[SUR_RB_LN1_W], g[SUR_RB_LN1_B]);
    RB_STAMP(4);
    if (need_din) {
        conv_bwd_weight<3>(g1, b.cout, b.in, b.cin, b.hin, b.stride, 1, g[SUR_RB_CONV1], GP(nullptr), lower_half(), false);
        conv_bwd_data<3>(g1, b.cout, b.hin, w[SUR_RB_CONV1], b.cin, b.stride, 1, din, true, upper_half(), true);
    } else {
        conv_bwd_weight<3>(g1, b.cout, b.in, b.cin, b.hin, b.stride, 1, g[SUR_RB_CONV1], GP(nullptr), all_waves(), true);
    }
    RB_STAMP(5);
#undef RB_STAMP
}

// ---------------------------------------------------------------------------------------------
// Narrow residual block: every channel count <= 4 (the action encoder, 1 -> 2 -> 4 -> 4) -- on the VALU, ONE WAVE PER SAMPLE.
// The MFMA gather-GEMM path spends its instructions on tile bookkeeping, index functors and masks whatever the channel
// counts: ~11 k wave-instructions per sample and block backward for ~10 k useful multiply-adds, and these kernels are
// instruction-issue bound.  Here a lane owns the positions p = lane + 64 e of every channel, keeps
// its values in registers, reads convolution neighbours from a wave-private LDS copy; LayerNorm statistics of all channels
// go through one halving butterfly (nr_wave_totals), a convolution's weight gradient is ONE MFMA tile whose reduction
// dimension is the position (nr_wgrad); the four waves of a workgroup handle four samples independently (no workgroup
// barrier inside a pass), sharing the staged weights.  Records in `saved`, outputs and partial-gradient rows are laid out
// exactly as the MFMA path leaves them, so either path can consume the other's.
// ---------------------------------------------------------------------------------------------
constexpr int NR_C = 4;   // channels
// positions per lane: at most 2 (block outputs are <= 128 wide, inputs that need a gradient too: N <= 256)

__host__ __device__ inline bool enc_narrow(const sur_encoder_params& p) {
    // block 0: 1 -> <= 2 channels onto <= 128 positions (tile 2 x 2, no input gradient); block 1: <= 2 -> <= 4 channels, <= 128 ->
    // <= 64 positions; block 2: <= 4 -> <= 4 channels on <= 64 positions
    if (p.c[0] != 1 || p.c[1] > 2 || p.c[2] > NR_C || p.c[3] > NR_C) return false;
    for (int b = 0; b < 3; ++b)
        if (p.stride[b] != 1 && p.stride[b] != 2) return false;
    const int h1 = p.n / p.stride[0], h2 = h1 / p.stride[1], h3 = h2 / p.stride[2];
    return h1 <= 128 && h2 <= 64 && h3 <= 64 && h3 == h2 && (h1 & 7) == 0 && (h2 & 7) == 0;
}

// LDS traffic between the lanes of ONE wave: make the wave's own writes visible to its later reads
#define NR_WAVE_SYNC()                                           \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)

// Totals over the whole wave of M (4 or 8) per-lane values, returned wave-uniform (SGPRs).  A halving butterfly: in a stage two
// lanes that differ in one bit of the lane id split a PAIR of values between them -- each keeps one, hands the other over and
// adds what it receives -- so a lane carries half as many values afterwards; M values cost ~M exchanges instead of 6 M, in six
// dependent stages instead of 6 M / 2 (a wave that owns a whole sample has nothing to hide that chain behind; LayerNorm sums
// were half of the narrow kernels' instructions).  Lane bits 0 and 1 by DPP quad permutes, bit 4 by v_permlane16_swap (which
// IS the keep-one-hand-one-over exchange between even and odd rows of 16), then plain all-reduce steps on the one value left:
// bits 3, 2 by row rotations, bit 5 by v_permlane32_swap.  Every lane ends up with the total of the value its bits (4,1,0)
// name; v_readlane hands them out.
template <int CTRL>
__device__ __forceinline__ float nr_halve(bool upper, float lo, float hi) {
    return (upper ? hi : lo) + dpp_rot<CTRL>(upper ? lo : hi);
}
template <int M>
__device__ __forceinline__ void nr_wave_totals(float (&v)[M], int lane) {
    static_assert(M == 4 || M == 8, "two or four channels, two sums each");
    const bool b0 = lane & 1, b1 = lane & 2;
    float w[M / 2], x[M / 4];
#pragma unroll
    for (int i = 0; i < M / 2; ++i) w[i] = nr_halve<0xB1>(b0, v[2 * i], v[2 * i + 1]);     // quad_perm:[1,0,3,2]
#pragma unroll
    for (int i = 0; i < M / 4; ++i) x[i] = nr_halve<0x4E>(b1, w[2 * i], w[2 * i + 1]);     // quad_perm:[2,3,0,1]
    // rows 1, 3 of the first operand <-> rows 0, 2 of the second: even rows collect x[0], odd rows x[M / 4 - 1]
    const swap_u2 r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[0]), __float_as_uint(x[M / 4 - 1]), false, false);
    float t = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
    t += dpp_rot<0x128>(t);  // row_ror:8
    t += dpp_rot<0x124>(t);  // row_ror:4: with the step before, the four lanes of the row that share bits 0 and 1
    t = swap_add32(t);
#pragma unroll
    for (int m = 0; m < M; ++m)
        v[m] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), (m & 3) + (M == 8 ? 16 * (m >> 2) : 0)));
}

// A lane's values of a [C][H] activation: channels c < TC, positions lane + 64 e, e < TE.  TC / TE are compile-time bounds
// (the kernels instantiate the three shapes of the 1 -> 2 -> 4 -> 4 encoder and a generic 4 x 2 fallback): a tile is 4 to 8
// registers, so the residual block's working set stays far below the 128-VGPR budget the MFMA path of the same kernel has.
template <int TC, int TE>
struct NrTile {
    float v[TC][TE];
};

// a value every lane of the wave holds alike (a staged weight read from LDS): kept in an SGPR instead of a VGPR per weight
__device__ __forceinline__ float nr_uniform(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

template <int TC, int TE>
__device__ __forceinline__ void nr_load(const float* buf, int C, int H, int lane, NrTile<TC, TE>& t) {
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
        for (int e = 0; e < TE; ++e) {
            const int p = lane + 64 * e;
            t.v[c][e] = (c < C && p < H) ? buf[c * H + p] : 0.0f;
        }
}
template <int TC, int TE>
__device__ __forceinline__ void nr_store(float* buf, int C, int H, int lane, const NrTile<TC, TE>& t) {
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
        for (int e = 0; e < TE; ++e) {
            const int p = lane + 64 * e;
            if (c < C && p < H) buf[c * H + p] = t.v[c][e];
        }
}

// out[o][p] = sum_{ci,k} W[o][ci][k] * in[ci][(p*stride + k - pad) mod hin] for the lane's own output positions (IC >= cin)
template <int K, int IC, int TC, int TE>
__device__ __forceinline__ void nr_conv(const float* in, int cin, int hin, const float* W, int cout, int stride, int pad, int hout,
                                        int lane, NrTile<TC, TE>& out) {
#pragma unroll
    for (int e = 0; e < TE; ++e) {
        const int p = lane + 64 * e;
        int idx[K];
#pragma unroll
        for (int k = 0; k < K; ++k) idx[k] = wrapi((p < hout ? p : 0) * stride + k - pad, hin);
        float x[IC][K];
#pragma unroll
        for (int ci = 0; ci < IC; ++ci)
#pragma unroll
            for (int k = 0; k < K; ++k) x[ci][k] = ci < cin ? in[ci * hin + idx[k]] : 0.0f;
#pragma unroll
        for (int o = 0; o < TC; ++o) {
            float acc = 0.0f;
            if (o < cout) {
#pragma unroll
                for (int ci = 0; ci < IC; ++ci)
                    if (ci < cin) {
#pragma unroll
                        for (int k = 0; k < K; ++k) acc = fmaf(nr_uniform(W[(o * cin + ci) * K + k]), x[ci][k], acc);
                    }
            }
            out.v[o][e] = p < hout ? acc : 0.0f;
        }
    }
}

// y = LayerNorm_H(act(x)) * gamma + beta, rows = channels, statistics over the 64 lanes x TE positions
template <int TC, int TE>
__device__ __forceinline__ void nr_ln_fwd(const NrTile<TC, TE>& x, int C, int H, const float* gamma, const float* beta, bool silu,
                                          int lane, NrTile<TC, TE>& y) {
    const float inv_h = 1.0f / H;
    float a[TC][TE], st[2 * TC];
#pragma unroll
    for (int c = 0; c < TC; ++c) {
        float s = 0.0f, ss = 0.0f;
#pragma unroll
        for (int e = 0; e < TE; ++e) {
            const bool live = c < C && lane + 64 * e < H;
            const float v = x.v[c][e];
            a[c][e] = live ? (silu ? v * sigmoid_(v) : v) : 0.0f;
            s += a[c][e];
            ss = fmaf(a[c][e], a[c][e], ss);
        }
        st[2 * c] = s;
        st[2 * c + 1] = ss;
    }
    nr_wave_totals(st, lane);
#pragma unroll
    for (int c = 0; c < TC; ++c) {
        const float mean = st[2 * c] * inv_h;
        const float rstd = rsqrtf(fmaxf(fmaf(-mean, mean, st[2 * c + 1] * inv_h), 0.0f) + LN_EPS);
#pragma unroll
        for (int e = 0; e < TE; ++e) {
            const int p = lane + 64 * e;
            y.v[c][e] = (c < C && p < H) ? fmaf((a[c][e] - mean) * rstd, gamma[p], beta[p]) : 0.0f;
        }
    }
}

// backward of nr_ln_fwd: dpre from dout; the lane adds its own positions' gamma / beta gradients to the wave's accumulators
template <int TC, int TE>
__device__ __forceinline__ void nr_ln_bwd(const NrTile<TC, TE>& dout, const NrTile<TC, TE>& pre, int C, int H, const float* gamma,
                                          bool silu, int lane, NrTile<TC, TE>& dpre, float* ggamma, float* gbeta) {
    const float inv_h = 1.0f / H;
    float y[TC][TE], dxh[TC][TE], st[2 * TC], rstd[TC];
#pragma unroll
    for (int c = 0; c < TC; ++c) {
        float s = 0.0f, ss = 0.0f;
#pragma unroll
        for (int e = 0; e < TE; ++e) {
            const int p = lane + 64 * e;
            const bool live = c < C && p < H;
            const float v = pre.v[c][e];
            y[c][e] = live ? (silu ? v * sigmoid_(v) : v) : 0.0f;
            dxh[c][e] = live ? dout.v[c][e] * gamma[p] : 0.0f;
            s += y[c][e];
            ss = fmaf(y[c][e], y[c][e], ss);
        }
        st[2 * c] = s;
        st[2 * c + 1] = ss;
    }
    nr_wave_totals(st, lane);
#pragma unroll
    for (int c = 0; c < TC; ++c) {
        const float mean = st[2 * c] * inv_h;
        rstd[c] = rsqrtf(fmaxf(fmaf(-mean, mean, st[2 * c + 1] * inv_h), 0.0f) + LN_EPS);
        float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
        for (int e = 0; e < TE; ++e) {
            const bool live = c < C && lane + 64 * e < H;
            y[c][e] = live ? (y[c][e] - mean) * rstd[c] : 0.0f;     // xhat
            m1 += dxh[c][e];
            m2 = fmaf(dxh[c][e], y[c][e], m2);
        }
        st[2 * c] = m1;
        st[2 * c + 1] = m2;
    }
    nr_wave_totals(st, lane);
    float gg[TE], gb[TE];
#pragma unroll
    for (int e = 0; e < TE; ++e) gg[e] = gb[e] = 0.0f;
#pragma unroll
    for (int c = 0; c < TC; ++c) {
        const float m1 = st[2 * c] * inv_h, m2 = st[2 * c + 1] * inv_h;
#pragma unroll
        for (int e = 0; e < TE; ++e) {
            const bool live = c < C && lane + 64 * e < H;
            float dy = rstd[c] * (dxh[c][e] - m1 - y[c][e] * m2);
            if (silu) {
                const float v = pre.v[c][e], sg = sigmoid_(v);
                dy *= sg * (1.0f + v * (1.0f - sg));
            }
            dpre.v[c][e] = live ? dy : 0.0f;
            if (live) {
                gg[e] = fmaf(dout.v[c][e], y[c][e], gg[e]);
                gb[e] += dout.v[c][e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < TE; ++e) {
        const int p = lane + 64 * e;
        if (p < H) {
            ggamma[p] += gg[e];
            gbeta[p] += gb[e];
        }
    }
}

// gW[o][ci][k] += sum_p dout[o][p] * in[ci][(p*stride + k - pad) mod hin]   (dout [cout][hout] and in [cin][hin]: this wave's LDS)
// One 16 x 16 MFMA tile: rows = output channels (cout <= 4 of 16), columns = (ci, k) (cin * K <= 12 of 16), reduction over the
// positions four at a time.  Mostly empty, and still ~50 instructions where 64-lane reductions of every weight's partial
// products took ~700: the tile does the cross-lane summation.  hout is a multiple of 8 (enc_narrow).
template <int K>
__device__ __forceinline__ void nr_wgrad(const float* dout, const float* in, int cin, int hin, int cout, int stride, int pad, int hout,
                                         int lane, float* gW) {
    const int r = lane & 15, q = lane >> 4, ncol = cin * K;
    const bool a_ok = r < cout, b_ok = r < ncol;
    const int ci = b_ok ? r / K : 0, k = b_ok ? r - ci * K : 0;
    const float* ap = dout + (a_ok ? r : 0) * hout;
    const float* bp = in + ci * hin;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < hout; p0 += 8) {
        const int pa = p0 + q, pb = p0 + 4 + q;
        const float a0 = ap[pa], b0 = bp[wrapi(pa * stride + k - pad, hin)];
        const float a1 = ap[pb], b1 = bp[wrapi(pb * stride + k - pad, hin)];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? a0 : 0.f, b_ok ? b0 : 0.f, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? a1 : 0.f, b_ok ? b1 : 0.f, acc1, 0, 0, 0);
    }
    if (q == 0 && b_ok) {        // rows 0..3 of the tile live in lanes 0..15
#pragma unroll
        for (int o = 0; o < NR_C; ++o)
            if (o < cout) gW[o * ncol + r] += acc0[o] + acc1[o];
    }
}

// din[ci][j] (+)= sum_{o,k : (p*stride + k - pad) mod hin == j} W[o][ci][k] * dout[o][p]   (dout: LDS, din: own positions; OC >= cout)
template <int K, int OC, int IC, int IE>
__device__ __forceinline__ void nr_dgrad(const float* dout, int cout, int hin, const float* W, int cin, int stride, int pad, int hout,
                                         int lane, NrTile<IC, IE>& din, bool accumulate) {
#pragma unroll
    for (int e = 0; e < IE; ++e) {
        const int j = lane + 64 * e;
        const bool live = j < hin;
        float acc[IC];
#pragma unroll
        for (int ci = 0; ci < IC; ++ci) acc[ci] = accumulate ? din.v[ci][e] : 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int t = wrapi((live ? j : 0) - k + pad, hin);
            if ((t & (stride - 1)) == 0) {
                const int pp = t >> (stride - 1);
#pragma unroll
                for (int o = 0; o < OC; ++o) {
                    if (o >= cout) break;
                    const float d = dout[o * hout + pp];
#pragma unroll
                    for (int ci = 0; ci < IC; ++ci)
                        if (ci < cin) acc[ci] = fmaf(nr_uniform(W[(o * cin + ci) * K + k]), d, acc[ci]);
                }
            }
        }
#pragma unroll
        for (int ci = 0; ci < IC; ++ci) din.v[ci][e] = (ci < cin && live) ? acc[ci] : 0.0f;
    }
}

// floats of one wave's LDS region
__host__ __device__ inline int nr_fwd_wave_floats(int cin, int hin, int cout, int hout) { return cin * hin + 7 * cout * hout; }
__host__ __device__ inline int nr_bwd_wave_floats(int cin, int hin, int cout, int hout, int psize_blk) {
    return cin * hin + 6 * cout * hout + ((psize_blk + 3) & ~3);   // in | a1pre a1 a2pre | s | dout | gbuf | gradient accumulators
}

// forward of one block for one sample: `wl` = this wave's LDS region; record layout skip | a1pre | a1 | a2pre | a2 | s | out
template <int IC, int TC, int TE>
__device__ __forceinline__ void nr_block_forward_t(int cin, int hin, int cout, int hout, int stride, const float* const* w, float* wl,
                                                   const float* __restrict__ src, float* __restrict__ rec_out, float* __restrict__ z,
                                                   int lane) {
    typedef NrTile<TC, TE> T;
    const int a = cout * hout, nin = cin * hin;
    float* in = wl;
    float* skip = in + nin;
    float *a1pre = skip + a, *a1 = skip + 2 * a, *a2pre = skip + 3 * a, *a2 = skip + 4 * a, *sb = skip + 5 * a, *out = skip + 6 * a;
    for (int i = lane; i < nin; i += 64) in[i] = src[i];
    NR_WAVE_SYNC();
    T t_skip, t_pre, t_act;
    nr_conv<1, IC>(in, cin, hin, w[SUR_RB_SKIP], cout, stride, 0, hout, lane, t_skip);
    nr_conv<3, IC>(in, cin, hin, w[SUR_RB_CONV1], cout, stride, 1, hout, lane, t_pre);
    nr_store(skip, cout, hout, lane, t_skip);
    nr_store(a1pre, cout, hout, lane, t_pre);
    nr_ln_fwd(t_pre, cout, hout, w[SUR_RB_LN1_W], w[SUR_RB_LN1_B], true, lane, t_act);
    nr_store(a1, cout, hout, lane, t_act);
    NR_WAVE_SYNC();
    nr_conv<3, TC>(a1, cout, hout, w[SUR_RB_CONV2], cout, 1, 1, hout, lane, t_pre);
    nr_store(a2pre, cout, hout, lane, t_pre);
    nr_ln_fwd(t_pre, cout, hout, w[SUR_RB_LN2_W], w[SUR_RB_LN2_B], true, lane, t_act);
    nr_store(a2, cout, hout, lane, t_act);
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
        for (int e = 0; e < TE; ++e) t_act.v[c][e] += t_skip.v[c][e];
    nr_store(sb, cout, hout, lane, t_act);
    nr_ln_fwd(t_act, cout, hout, w[SUR_RB_LN3_W], w[SUR_RB_LN3_B], false, lane, t_pre);
    nr_store(out, cout, hout, lane, t_pre);
    NR_WAVE_SYNC();
    for (int i = lane; i < (7 * a) >> 2; i += 64) reinterpret_cast<float4*>(rec_out)[i] = reinterpret_cast<const float4*>(skip)[i];
    if (z)
        for (int i = lane; i < a; i += 64) z[i] = out[i];
    NR_WAVE_SYNC();      // the region is reused by this wave's next sample
}

__device__ __forceinline__ void nr_block_forward(int cin, int hin, int cout, int hout, int stride, const float* const* w, float* wl,
                                                 const float* __restrict__ src, float* __restrict__ rec_out, float* __restrict__ z, int lane) {
    if (cin <= 1 && cout <= 2) nr_block_forward_t<1, 2, 2>(cin, hin, cout, hout, stride, w, wl, src, rec_out, z, lane);
    else nr_block_forward_t<4, 4, 1>(cin, hin, cout, hout, stride, w, wl, src, rec_out, z, lane);      // hout <= 64 (enc_narrow)
}

// backward of one block for one sample; g[] = this wave's gradient accumulators (LDS)
template <int TC, int TE, int IC, int IE>
__device__ __forceinline__ void nr_block_backward_t(int cin, int hin, int cout, int hout, int stride, const float* const* w, float* const* g,
                                                    float* wl, const float* __restrict__ in_src, const float* __restrict__ rec_blk,
                                                    const float* __restrict__ dsrc, float* __restrict__ din_dst, bool need_din, int lane) {
    typedef NrTile<TC, TE> T;
    const int a = cout * hout, nin = cin * hin;
    float* in = wl;
    float* a1pre = in + nin;
    float *a1 = a1pre + a, *a2pre = a1pre + 2 * a, *sb = a1pre + 3 * a, *dout = a1pre + 4 * a, *gbuf = a1pre + 5 * a;
    for (int i = lane; i < nin; i += 64) in[i] = in_src[i];
    for (int i = lane; i < 3 * a; i += 64) a1pre[i] = rec_blk[a + i];            // a1pre | a1 | a2pre
    for (int i = lane; i < a; i += 64) {
        sb[i] = rec_blk[5 * a + i];
        dout[i] = dsrc[i];
    }
    NR_WAVE_SYNC();
    T t_d, t_x, t_g;
    NrTile<IC, IE> t_din;
    nr_load(dout, cout, hout, lane, t_d);
    nr_load(sb, cout, hout, lane, t_x);
    nr_ln_bwd(t_d, t_x, cout, hout, w[SUR_RB_LN3_W], false, lane, t_g, g[SUR_RB_LN3_W], g[SUR_RB_LN3_B]);       // g1
    nr_store(gbuf, cout, hout, lane, t_g);
    NR_WAVE_SYNC();
    nr_wgrad<1>(gbuf, in, cin, hin, cout, stride, 0, hout, lane, g[SUR_RB_SKIP]);
    if (need_din) nr_dgrad<1, TC>(gbuf, cout, hin, w[SUR_RB_SKIP], cin, stride, 0, hout, lane, t_din, false);
    nr_load(a2pre, cout, hout, lane, t_x);
    nr_ln_bwd(t_g, t_x, cout, hout, w[SUR_RB_LN2_W], true, lane, t_d, g[SUR_RB_LN2_W], g[SUR_RB_LN2_B]);         // g2 (in t_d)
    NR_WAVE_SYNC();       // every lane has read g1 from gbuf
    nr_store(gbuf, cout, hout, lane, t_d);
    NR_WAVE_SYNC();
    nr_wgrad<3>(gbuf, a1, cout, hout, cout, 1, 1, hout, lane, g[SUR_RB_CONV2]);
    nr_dgrad<3, TC>(gbuf, cout, hout, w[SUR_RB_CONV2], cout, 1, 1, hout, lane, t_g, false);                        // g3 (in t_g)
    nr_load(a1pre, cout, hout, lane, t_x);
    nr_ln_bwd(t_g, t_x, cout, hout, w[SUR_RB_LN1_W], true, lane, t_d, g[SUR_RB_LN1_W], g[SUR_RB_LN1_B]);         // g1' (in t_d)
    NR_WAVE_SYNC();
    nr_store(gbuf, cout, hout, lane, t_d);
    NR_WAVE_SYNC();
    nr_wgrad<3>(gbuf, in, cin, hin, cout, stride, 1, hout, lane, g[SUR_RB_CONV1]);
    if (need_din) {
        nr_dgrad<3, TC>(gbuf, cout, hin, w[SUR_RB_CONV1], cin, stride, 1, hout, lane, t_din, true);
#pragma unroll
        for (int c = 0; c < IC; ++c)
#pragma unroll
            for (int e = 0; e < IE; ++e) {
                const int j = lane + 64 * e;
                if (c < cin && j < hin) din_dst[c * hin + j] = t_din.v[c][e];
            }
    }
    NR_WAVE_SYNC();
}

__device__ __forceinline__ void nr_block_backward(int cin, int hin, int cout, int hout, int stride, const float* const* w, float* const* g,
                                                  float* wl, const float* __restrict__ in_src, const float* __restrict__ rec_blk,
                                                  const float* __restrict__ dsrc, float* __restrict__ din_dst, bool need_din, int lane) {
    // the three shapes of the 1 -> 2 -> 4 -> 4 encoder (N <= 256); enc_narrow() admits nothing else
    if (!need_din && cin <= 1 && cout <= 2)
        nr_block_backward_t<2, 2, 1, 1>(cin, hin, cout, hout, stride, w, g, wl, in_src, rec_blk, dsrc, din_dst, false, lane);
    else if (hout <= 64 && cin <= 2 && hin <= 128)
        nr_block_backward_t<4, 1, 2, 2>(cin, hin, cout, hout, stride, w, g, wl, in_src, rec_blk, dsrc, din_dst, need_din, lane);
    else      // hout <= 64, hin <= 64 (enc_narrow)
        nr_block_backward_t<4, 1, 4, 1>(cin, hin, cout, hout, stride, w, g, wl, in_src, rec_blk, dsrc, din_dst, need_din, lane);
}

struct EncLayout {
    RBBuf rb[3];
    float *g1, *g2, *g3, *xh, *dA, *dB, *end;
};

__host__ __device__ inline int enc_max_act(const sur_encoder_params& p) {
    int h = p.n, m = p.c[0] * p.n;
    for (int b = 0; b < 3; ++b) {
        h /= p.stride[b];
        const int a = p.c[b + 1] * h;
        m = a > m ? a : m;
    }
    return m;
}

__host__ __device__ inline int enc_act_floats(const sur_encoder_params& p, bool backward) {
    int h = p.n, total = p.c[0] * p.n;
    for (int b = 0; b < 3; ++b) {
        h /= p.stride[b];
        total += 7 * p.c[b + 1] * h;
    }
    if (backward) total += 6 * enc_max_act(p);
    return total;
}

__device__ void enc_layout(const sur_encoder_params& p, float* lds, bool backward, EncLayout& L) {
    float* cur = lds;
    int h = p.n;
    float* in = cur;
    cur += p.c[0] * p.n;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        RBBuf& r = L.rb[b];
        r.cin = p.c[b];
        r.cout = p.c[b + 1];
        r.hin = h;
        r.stride = p.stride[b];
        h /= p.stride[b];
        r.hout = h;
        const int a = r.cout * r.hout;
        r.in = in;
        r.skip = cur;
        r.a1pre = cur + a;
        r.a1 = cur + 2 * a;
        r.a2pre = cur + 3 * a;
        r.a2 = cur + 4 * a;
        r.s = cur + 5 * a;
        r.out = cur + 6 * a;
        cur += 7 * a;
        in = r.out;
    }
    if (backward) {
        const int m = enc_max_act(p);
        L.g1 = cur;
        L.g2 = cur + m;
        L.g3 = cur + 2 * m;
        L.xh = cur + 3 * m;
        L.dA = cur + 4 * m;
        L.dB = cur + 5 * m;
        cur += 6 * m;
    }
    L.end = cur;
}

// floats of one sample's forward intermediates (everything after the input in the LDS layout), as enc_fwd_kernel can
// save them for enc_bwd_kernel
__host__ __device__ inline int enc_saved_floats(const sur_encoder_params& p) { return enc_act_floats(p, false) - p.c[0] * p.n; }

// n4 float4 from global to LDS with all of a thread's loads of a round in flight before the first store
__device__ __forceinline__ void lds_load_v4(float* dst, const float* __restrict__ src, int n4) {
    constexpr int U = 4;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i0 = threadIdx.x; i0 < n4; i0 += U * TPB) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s4[i0 + u * TPB < n4 ? i0 + u * TPB : i0];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + u * TPB < n4) d4[i0 + u * TPB] = v[u];
    }
    __syncthreads();
}

__global__ void __launch_bounds__(TPB) enc_fwd_kernel(const sur_encoder_params p, const float* __restrict__ x, int m_total,
                                                      float* __restrict__ z, float* __restrict__ saved) {
    extern __shared__ __align__(16) float lds[];
    EncLayout L;
    enc_layout(p, lds, false, L);
    ParamViews<SUR_ENC_NPARAM> v;
    stage_weights<SUR_ENC_NPARAM>(p.w, p.size, L.end, v);
    const int nin = p.c[0] * p.n, nout = L.rb[2].cout * L.rb[2].hout;
    for (int m = blockIdx.x; m < m_total; m += gridDim.x) {
        lds_load(L.rb[0].in, x + (size_t)m * nin, nin);
#pragma unroll
        for (int b = 0; b < 3; ++b) rb_forward(L.rb[b], v.w + b * SUR_RB_NPARAM);
        if (saved) {  // every intermediate of the three blocks: the backward kernel then skips its forward recomputation
            const int n4 = enc_saved_floats(p) >> 2;
            float4* dst = reinterpret_cast<float4*>(saved + (size_t)m * (n4 << 2));
            const float4* src = reinterpret_cast<const float4*>(L.rb[0].skip);
            for (int i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
        }
        lds_store(z + (size_t)m * nout, L.rb[2].out, nout);
    }
}

// Two workgroups per CU (2 waves per SIMD): the kernel is bound by dependent-phase latency, so a second resident
// workgroup fills the issue slots the first one leaves empty.  Costs ~118 spilled dwords, measured +10 % on the step.
#ifndef ENC_BWD_OCC
#define ENC_BWD_OCC 2
#endif
// body of the encoder backward for the workgroup `wg` of `nwg` working on one encoder (job)
__device__ __forceinline__ void enc_bwd_body(const sur_encoder_params& p, const float* __restrict__ x,
                                             const float* __restrict__ dz, int m_total, float* __restrict__ dx, int grads_in_lds,
                                             int row_base, const float* __restrict__ saved, int wg, int nwg, float* lds) {
    STAMP(32);
    EncLayout L;
    enc_layout(p, lds, true, L);
    ParamViews<SUR_ENC_NPARAM> v;
    stage_weights<SUR_ENC_NPARAM>(p.w, p.size, L.end, v);
    STAMP(33);
    const int psize = psize_of<SUR_ENC_NPARAM>(p.size);
    float* row = p.partial + (size_t)(row_base + wg) * psize;
    float* gacc = grads_in_lds ? L.end + psize : row;
    setup_grads<SUR_ENC_NPARAM>(p.size, gacc, grads_in_lds != 0, v);
    STAMP(34);
    const int nin = p.c[0] * p.n, nout = L.rb[2].cout * L.rb[2].hout;
    for (int m = wg; m < m_total; m += nwg) {
        lds_load(L.rb[0].in, x + (size_t)m * nin, nin);
        STAMP(35);
        if (saved) {  // the second workgroup resident on this CU covers the load latency
            const int nsv = enc_saved_floats(p);
            lds_load_v4(L.rb[0].skip, saved + (size_t)m * nsv, nsv >> 2);
        } else {
#pragma unroll
            for (int b = 0; b < 3; ++b) rb_forward(L.rb[b], v.w + b * SUR_RB_NPARAM);
        }
        STAMP(36);
        lds_load(L.dA, dz + (size_t)m * nout, nout);
        STAMP(37);
        float *dout = L.dA, *din = L.dB;
#pragma unroll
        for (int b = 2; b >= 0; --b) {
            rb_backward(L.rb[b], v.w + b * SUR_RB_NPARAM, v.g + b * SUR_RB_NPARAM, dout, din, L.g1, L.g2, L.g3, L.xh);
            STAMP(38 + b);
            float* t = dout;
            dout = din;
            din = t;
        }
        if (dx) lds_store(dx + (size_t)m * nin, dout, nin);
        STAMP(41);
    }
    if (grads_in_lds) {
        add_to_row(row, gacc, psize);
    }
    STAMP(42);
}

__global__ void __launch_bounds__(TPB, ENC_BWD_OCC) enc_bwd_kernel(const sur_encoder_params p, const float* __restrict__ x,
                                                      const float* __restrict__ dz, int m_total, float* __restrict__ dx,
                                                      int grads_in_lds, int row_base, const float* __restrict__ saved) {
    extern __shared__ __align__(16) float lds[];
    enc_bwd_body(p, x, dz, m_total, dx, grads_in_lds, row_base, saved, blockIdx.x, gridDim.x, lds);
}

// Several encoder backward jobs (different encoders / inputs) in ONE launch: the workgroups of all jobs are
// dispatched together, so a long job never queues behind a short one on the same hardware queue.
struct EncBwdJob {
    sur_encoder_params p;
    const float* x;
    const float* dz;
    const float* saved;
    int m, row_base, wg_begin, wg_count, grads_in_lds;
};

__global__ void __launch_bounds__(TPB, ENC_BWD_OCC) enc_bwd_multi_kernel(const EncBwdJob j0, const EncBwdJob j1, const EncBwdJob j2,
                                                                         int njobs) {
    extern __shared__ __align__(16) float lds[];
    const int wg = blockIdx.x;
    if (njobs > 2 && wg >= j2.wg_begin)
        enc_bwd_body(j2.p, j2.x, j2.dz, j2.m, nullptr, j2.grads_in_lds, j2.row_base, j2.saved, wg - j2.wg_begin, j2.wg_count, lds);
    else if (njobs > 1 && wg >= j1.wg_begin)
        enc_bwd_body(j1.p, j1.x, j1.dz, j1.m, nullptr, j1.grads_in_lds, j1.row_base, j1.saved, wg - j1.wg_begin, j1.wg_count, lds);
    else
        enc_bwd_body(j0.p, j0.x, j0.dz, j0.m, nullptr, j0.grads_in_lds, j0.row_base, j0.saved, wg, j0.wg_count, lds);
}

// ---------------------------------------------------------------------------------------------
// Encoder backward, one residual block per launch (block 2, then 1, then 0): a third of the code and of the LDS
// of the whole-encoder kernel per launch, so it no longer sits at the 256-VGPR cap and more workgroups share a CU.
// The blocks' forward intermediates come from the `saved` buffer of enc_fwd_kernel; the gradient between blocks
// travels through a small workspace ([M, c1*h1 + c2*h2] floats).
// ---------------------------------------------------------------------------------------------
struct EncBlockGeom {
    int cin, cout, hin, hout, stride;
    int saved_off;      // offset of this block's 7 intermediates in a sample's saved record
    int in_saved_off;   // offset of this block's INPUT (= previous block's `out`) in the record, -1: the raw input x
    int param_off;      // column of this block's parameters in a partial-gradient row
    int psize_blk;      // their total size
};

__host__ __device__ inline EncBlockGeom enc_block_geom(const sur_encoder_params& p, int blk) {
    EncBlockGeom g{};
    int h = p.n, off = 0, prev_out = -1;
    for (int b = 0; b <= blk; ++b) {
        g.cin = p.c[b];
        g.cout = p.c[b + 1];
        g.hin = h;
        g.stride = p.stride[b];
        h /= p.stride[b];
        g.hout = h;
        g.saved_off = off;
        g.in_saved_off = prev_out;
        const int a = g.cout * g.hout;
        prev_out = off + 6 * a;
        off += 7 * a;
    }
    g.param_off = 0;
    for (int i = 0; i < SUR_RB_NPARAM * blk; ++i) g.param_off += p.size[i];
    g.psize_blk = 0;
    for (int i = 0; i < SUR_RB_NPARAM; ++i) g.psize_blk += p.size[SUR_RB_NPARAM * blk + i];
    return g;
}

// workspace floats per sample: the gradients wrt the inputs of blocks 1 and 2
__host__ __device__ inline int enc_ws_floats(const sur_encoder_params& p) {
    const int h1 = p.n / p.stride[0], h2 = h1 / p.stride[1];
    return p.c[1] * h1 + p.c[2] * h2;
}

// LDS of the block backward: only what rb_backward reads is resident -- the input, four of the seven saved intermediates
// (a1pre, a1, a2pre: contiguous in the record; s), and scratch that reuses dead buffers (g2 over dout, g3 over s).
// 2 nin + 7 a floats instead of 2 nin + 13 a: at N = 256 a third workgroup fits a CU.
__host__ __device__ inline int enc_block_act_floats(const EncBlockGeom& g) {
    const int a = g.cout * g.hout, nin = g.cin * g.hin;
    return nin + 4 * a + 2 * a + a + nin;   // in, {a1pre a1 a2pre s}, {g1 xh}, dout (= g2), din
}

struct EncBlockJob {
    sur_encoder_params p;
    const float* x;
    const float* dz;
    const float* saved;
    float* ws;
    int m, row_base, wg_begin, wg_count, grads_in_lds;
};

template <bool GL>     // GL: this job's gradient accumulators are in LDS (j.grads_in_lds), as LDS-typed pointers (conv_bwd_weight)
__device__ __forceinline__ void enc_block_bwd_body(const EncBlockJob& j, int blk, int wg, float* lds) {
    const sur_encoder_params& p = j.p;
    const EncBlockGeom gm = enc_block_geom(p, blk);
    const int a = gm.cout * gm.hout, nin = gm.cin * gm.hin;
    RBBuf rb{};
    rb.cin = gm.cin; rb.cout = gm.cout; rb.hin = gm.hin; rb.hout = gm.hout; rb.stride = gm.stride;
    float* cur = lds;
    rb.in = cur; cur += nin;
    rb.a1pre = cur; rb.a1 = cur + a; rb.a2pre = cur + 2 * a; rb.s = cur + 3 * a; cur += 4 * a;   // skip, a2, out: not read
    float *g1 = cur, *xh = cur + a;
    cur += 2 * a;
    float* dout = cur; cur += a;
    float* g2 = dout;     // dout is consumed by the first LayerNorm backward, before g2 is written
    float* g3 = rb.s;     // likewise s
    float* din = cur; cur += nin;
    ParamViews<SUR_RB_NPARAM> v;
    const int sbase = 64 + 16 * blk;   // SUR_STAMP ids of this block's phases
    (void)sbase;
    STAMP(sbase + 0);
    stage_weights<SUR_RB_NPARAM>(p.w + SUR_RB_NPARAM * blk, p.size + SUR_RB_NPARAM * blk, cur, v);
    const int psize = psize_of<SUR_ENC_NPARAM>(p.size);
    float* row = p.partial + (size_t)(j.row_base + wg) * psize + gm.param_off;
    typedef typename std::conditional<GL, lds_f*, float*>::type GP;
    GP gacc, g[SUR_RB_NPARAM];
    if constexpr (GL) gacc = (lds_f*)(cur + gm.psize_blk);
    else gacc = row;
    {
        int off = 0;
#pragma unroll
        for (int i = 0; i < SUR_RB_NPARAM; ++i) {
            g[i] = gacc + off;
            off += p.size[SUR_RB_NPARAM * blk + i];
        }
        if (GL)
            for (int t = threadIdx.x; t < off; t += blockDim.x) gacc[t] = 0.0f;
        __syncthreads();
    }
    STAMP(sbase + 1);
    const int nsv = enc_saved_floats(p), nws = enc_ws_floats(p);
    const int h1 = p.n / p.stride[0];
    const int ws_in_off = blk == 2 ? p.c[1] * h1 : 0;       // where this block writes d loss / d its input (blocks 2, 1)
    const int ws_out_off = blk == 1 ? p.c[1] * h1 : 0;      // where block 1 reads d loss / d its output (written by block 2)
    for (int m = wg; m < j.m; m += j.wg_count) {
        const float* rec = j.saved + (size_t)m * nsv;
        const float* dsrc = blk == 2 ? j.dz + (size_t)m * a : j.ws + (size_t)m * nws + (blk == 1 ? ws_out_off : 0);
        // this block's input, four of its seven intermediates, the gradient wrt its output.  record: skip | a1pre a1 a2pre | a2 | s | out
        // (one loader with every load of the sample in flight and a single barrier measured 3 % SLOWER on the whole step than
        // these four: 40 more live registers spill in a 128-VGPR kernel, and the co-resident workgroups cover the round trips)
        // by LDS-DMA: all four in flight at once, one wait + barrier (through registers each was a round trip of its own -- a sixth of
        // the block's time -- and holding them all in registers spilled; the DMA needs none)
        if (gm.in_saved_off < 0) lds_dma_v4(rb.in, j.x + (size_t)m * nin, nin >> 2);
        else lds_dma_v4(rb.in, rec + gm.in_saved_off, nin >> 2);
        lds_dma_v4(rb.a1pre, rec + gm.saved_off + a, (3 * a) >> 2);
        lds_dma_v4(rb.s, rec + gm.saved_off + 5 * a, a >> 2);
        lds_dma_v4(dout, dsrc, a >> 2);
        lds_dma_wait();
        STAMP(sbase + 2);
        rb_backward(rb, v.w, g, dout, din, g1, g2, g3, xh, sbase + 3, blk > 0);
        if (blk > 0) {
            float* dst = j.ws + (size_t)m * nws + (blk == 2 ? ws_in_off : 0);
            for (int i = threadIdx.x; i < nin; i += blockDim.x) dst[i] = din[i];
        }
        __syncthreads();
        STAMP(sbase + 9);
    }
    if constexpr (GL) add_to_row(row, (const float*)gacc, gm.psize_blk);
    STAMP(sbase + 10);
}


// A narrow job (one wave per sample, see "Narrow residual block") has a flat argument block whose arrays are only ever indexed
// by constants (as members of the wide jobs' structs -- indexed by the block number -- they went through scratch: 3 KB per
// lane).  Its workgroups ride BEHIND the wide jobs' in the same launch (`narrow_begin` of the multi kernels: as launches of
// their own, serialized behind the wide ones, the narrow kernels lost what they had gained); a launch set without a wide job
// uses the standalone kernels below.
struct NarrowJob {
    const float* w[SUR_RB_NPARAM];   // this block's weights (global)
    int size[SUR_RB_NPARAM];
    int cin, hin, cout, hout, stride;
    int nsv, nws;                    // floats per sample of the saved record / of the inter-block gradient workspace
    int saved_off, in_saved_off;     // this block's intermediates / its input inside a record (-1: the raw input x)
    int ws_in_off, ws_out_off;       // where the block writes d loss / d input, reads d loss / d output (blocks 1, 0)
    int blk, m, wg_count, psize_blk, need_din;
    const float* x;
    const float* dz;                 // backward, block 2: d loss / d z
    float* z;                        // forward, block 2: the encoder output
    float* saved;
    float* ws;
    float* rows;                     // backward: &partial[row_base][param_off]; workgroup wg adds to rows + wg * row_stride
    int row_stride;
};

__device__ __forceinline__ void narrow_stage(const NarrowJob& j, float* dst, const float** w) {
    int off = 0;
#pragma unroll
    for (int i = 0; i < SUR_RB_NPARAM; ++i) {
        w[i] = dst + off;
        for (int t = threadIdx.x; t < j.size[i]; t += blockDim.x) dst[off + t] = j.w[i][t];
        off += j.size[i];
    }
    __syncthreads();
}

__device__ __forceinline__ void enc_narrow_fwd_run(const NarrowJob& j, int wg, float* lds) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
    const int a = j.cout * j.hout, nin = j.cin * j.hin;
    const int wf = nr_fwd_wave_floats(j.cin, j.hin, j.cout, j.hout);
    const float* w[SUR_RB_NPARAM];
    narrow_stage(j, lds + nwv * wf, w);
    const int passes = (j.m + nwv - 1) / nwv;
    for (int mp = wg; mp < passes; mp += j.wg_count) {
        const int m = mp * nwv + wave;
        if (m < j.m) {
            float* rec = j.saved + (size_t)m * j.nsv;
            const float* src = j.in_saved_off < 0 ? j.x + (size_t)m * nin : rec + j.in_saved_off;
            nr_block_forward(j.cin, j.hin, j.cout, j.hout, j.stride, w, lds + wave * wf, src, rec + j.saved_off,
                             j.z ? j.z + (size_t)m * a : nullptr, lane);
        }
    }
}

__global__ void __launch_bounds__(TPB) enc_narrow_fwd_kernel(const NarrowJob j) {
    extern __shared__ __align__(16) float lds[];
    enc_narrow_fwd_run(j, blockIdx.x, lds);
}

__device__ __forceinline__ void enc_narrow_bwd_run(const NarrowJob& j, int wg, float* lds) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
    const int a = j.cout * j.hout, nin = j.cin * j.hin;
    const int wf = nr_bwd_wave_floats(j.cin, j.hin, j.cout, j.hout, j.psize_blk);
    float* wl = lds + wave * wf;
    float* gacc = wl + nin + 6 * a;
    const float* w[SUR_RB_NPARAM];
    float* g[SUR_RB_NPARAM];
    {
        int off = 0;
#pragma unroll
        for (int i = 0; i < SUR_RB_NPARAM; ++i) {
            g[i] = gacc + off;
            off += j.size[i];
        }
        for (int i = lane; i < j.psize_blk; i += 64) gacc[i] = 0.0f;
    }
    narrow_stage(j, lds + nwv * wf, w);
    const int passes = (j.m + nwv - 1) / nwv;
    for (int mp = wg; mp < passes; mp += j.wg_count) {
        const int m = mp * nwv + wave;
        if (m < j.m) {
            const float* rec = j.saved + (size_t)m * j.nsv;
            const float* in_src = j.in_saved_off < 0 ? j.x + (size_t)m * nin : rec + j.in_saved_off;
            const float* dsrc = j.blk == 2 ? j.dz + (size_t)m * a : j.ws + (size_t)m * j.nws + j.ws_out_off;
            float* din_dst = j.need_din ? j.ws + (size_t)m * j.nws + j.ws_in_off : nullptr;
            nr_block_backward(j.cin, j.hin, j.cout, j.hout, j.stride, w, g, wl, in_src, rec + j.saved_off, dsrc, din_dst, j.need_din != 0,
                              lane);
        }
    }
    __syncthreads();
    float* row = j.rows + (size_t)wg * j.row_stride;
    for (int i = threadIdx.x; i < j.psize_blk; i += blockDim.x) {
        float t = 0.0f;
        for (int wv = 0; wv < nwv; ++wv) t += lds[wv * wf + nin + 6 * a + i];
        row[i] += t;
    }
}

__global__ void __launch_bounds__(TPB) enc_narrow_bwd_kernel(const NarrowJob j) {
    extern __shared__ __align__(16) float lds[];
    enc_narrow_bwd_run(j, blockIdx.x, lds);
}

static NarrowJob narrow_job(const sur_encoder_params& p, int blk, int m) {
    NarrowJob j{};
    const EncBlockGeom gm = enc_block_geom(p, blk);
    for (int i = 0; i < SUR_RB_NPARAM; ++i) {
        j.w[i] = p.w[SUR_RB_NPARAM * blk + i];
        j.size[i] = p.size[SUR_RB_NPARAM * blk + i];
    }
    j.cin = gm.cin; j.hin = gm.hin; j.cout = gm.cout; j.hout = gm.hout; j.stride = gm.stride;
    j.nsv = enc_saved_floats(p);
    j.nws = enc_ws_floats(p);
    j.saved_off = gm.saved_off;
    j.in_saved_off = gm.in_saved_off;
    const int h1 = p.n / p.stride[0];
    j.ws_in_off = blk == 2 ? p.c[1] * h1 : 0;
    j.ws_out_off = blk == 1 ? p.c[1] * h1 : 0;
    j.blk = blk;
    j.m = m;
    j.psize_blk = gm.psize_blk;
    j.need_din = blk > 0 ? 1 : 0;
    return j;
}

#ifndef ENC_BLK_OCC
#define ENC_BLK_OCC 3   // 168 VGPRs, no spills.  Four per CU (128 VGPRs, ~30 spilled dwords) measured best while the action encoder's
                        // 640 samples went through these workgroups too; with those on the narrow path a launch is <= 480 workgroups
                        // and the spills cost more than the residency gives: 0.496 vs 0.511 ms per step at N = 256 (2 per CU: 0.499)
#endif
template <bool GL>     // every wide job keeps its accumulators in LDS (the host clears grads_in_lds of all of them otherwise)
__global__ void __launch_bounds__(TPB, ENC_BLK_OCC)
enc_block_bwd_multi_kernel(const EncBlockJob j0, const EncBlockJob j1, const EncBlockJob j2, int njobs, int blk, const NarrowJob nj,
                           int narrow_begin) {
    extern __shared__ __align__(16) float lds[];
    const int wg = blockIdx.x;
    if (wg >= narrow_begin) {      // the workgroups behind the wide jobs' carry the narrow job (one wave per sample)
        enc_narrow_bwd_run(nj, wg - narrow_begin, lds);
        return;
    }
    if (njobs > 2 && wg >= j2.wg_begin) enc_block_bwd_body<GL>(j2, blk, wg - j2.wg_begin, lds);
    else if (njobs > 1 && wg >= j1.wg_begin) enc_block_bwd_body<GL>(j1, blk, wg - j1.wg_begin, lds);
    else enc_block_bwd_body<GL>(j0, blk, wg, lds);
}

// Encoder forward, one residual block per launch (block 0, 1, 2), several jobs per launch: for large sample counts the
// same trade as in the backward pass -- a third of the registers and LDS per launch, more workgroups per CU.  Each
// block reads its input from the previous block's record in `saved` and writes its own seven intermediates there.
struct EncFwdJob {
    sur_encoder_params p;
    const float* x;
    float* z;
    float* saved;
    int m, wg_begin, wg_count;
};

__device__ __forceinline__ void enc_block_fwd_body(const EncFwdJob& j, int blk, int wg, float* lds) {
    const sur_encoder_params& p = j.p;
    const EncBlockGeom gm = enc_block_geom(p, blk);
    const int a = gm.cout * gm.hout, nin = gm.cin * gm.hin;
    RBBuf rb{};
    rb.cin = gm.cin; rb.cout = gm.cout; rb.hin = gm.hin; rb.hout = gm.hout; rb.stride = gm.stride;
    float* cur = lds;
    rb.in = cur; cur += nin;
    rb.skip = cur; rb.a1pre = cur + a; rb.a1 = cur + 2 * a; rb.a2pre = cur + 3 * a; rb.a2 = cur + 4 * a; rb.s = cur + 5 * a;
    rb.out = cur + 6 * a; cur += 7 * a;
    ParamViews<SUR_RB_NPARAM> v;
    stage_weights<SUR_RB_NPARAM>(p.w + SUR_RB_NPARAM * blk, p.size + SUR_RB_NPARAM * blk, cur, v);
    const int nsv = enc_saved_floats(p);
    for (int m = wg; m < j.m; m += j.wg_count) {
        float* rec = j.saved + (size_t)m * nsv;
        if (gm.in_saved_off < 0) lds_load(rb.in, j.x + (size_t)m * nin, nin);
        else lds_load_v4(rb.in, rec + gm.in_saved_off, nin >> 2);
        rb_forward(rb, v.w);
        float4* dst = reinterpret_cast<float4*>(rec + gm.saved_off);
        const float4* src = reinterpret_cast<const float4*>(rb.skip);
        for (int i = threadIdx.x; i < (7 * a) >> 2; i += blockDim.x) dst[i] = src[i];
        if (blk == 2)
            for (int i = threadIdx.x; i < a; i += blockDim.x) j.z[(size_t)m * a + i] = rb.out[i];
        __syncthreads();
    }
}


__global__ void __launch_bounds__(TPB, ENC_BLK_OCC)
enc_block_fwd_multi_kernel(const EncFwdJob j0, const EncFwdJob j1, int njobs, int blk, const NarrowJob nj, int narrow_begin) {
    extern __shared__ __align__(16) float lds[];
    const int wg = blockIdx.x;
    if (wg >= narrow_begin) {
        enc_narrow_fwd_run(nj, wg - narrow_begin, lds);
        return;
    }
    if (njobs > 1 && wg >= j1.wg_begin) enc_block_fwd_body(j1, blk, wg - j1.wg_begin, lds);
    else enc_block_fwd_body(j0, blk, wg, lds);
}

// ---------------------------------------------------------------------------------------------
// TBPTT delta-mode loss: one launch for what the reference spells as ~30 tiny torch ops
// (pdecontrol/surrogates/training.py:100-121): true deltas from the state sequence, the undscaling
// forward, the element-wise MSE, its time-resolved and overall means, the four logged statistics and
// the gradient of the mean loss wrt the predicted deltas.
//   grid = T x nsplit workgroups: row t < T-1 handles time step t of all samples, row T-1 only zeroes the
//   (unused) last step's gradient.  Partial sums are fp64 and reduced in a fixed order by the
//   workgroup that arrives last (ticket counter), so the result does not depend on scheduling.
// ---------------------------------------------------------------------------------------------
constexpr int LOSS_NSUM = 5;  // squared error, sum / sum of squares of predicted deltas, same of true deltas
constexpr int LOSS_UNROLL = 4;
constexpr int LOSS_MAX_SPLIT = 8;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ void delta_loss_finish(int B, int T, int N, int nsplit, float* __restrict__ hsteploss, float* __restrict__ loss,
                                  float* __restrict__ stats, double* __restrict__ partial, unsigned int* __restrict__ ticket);

__global__ void __launch_bounds__(TPB)
delta_loss_kernel(const float* __restrict__ states, long sb, long st, const float* __restrict__ d_all, int B, int T, int N, float delta,
                  float mean, float stdv, float* __restrict__ deltas, float* __restrict__ dd_all,
                  float* __restrict__ hsteploss, float* __restrict__ loss, float* __restrict__ stats,
                  double* __restrict__ partial, unsigned int* __restrict__ ticket, int t0, int take_ticket) {
    __shared__ double red[TPB / 64][LOSS_NSUM];
    __shared__ bool last;
    // blockIdx.x = time step, blockIdx.y = slice of the B*N elements of that step; LOSS_UNROLL elements per
    // thread per round with every load of the round issued before the first use
    // a launch covers the time steps [t0, t0 + gridDim.x) of the T rows; the ticket counts the workgroups of ALL T rows, so
    // the launches of one loss (one per TBPTT chunk, in any order, on any streams) share the final reduction
    const int t = t0 + blockIdx.x, per_t = B * N, nsplit = gridDim.y;
    const double count = (double)per_t * (T - 1);
    double acc[LOSS_NSUM] = {0.0, 0.0, 0.0, 0.0, 0.0};
    const int stride = nsplit * TPB;
    if (t < T - 1) {
        const float gscale = (float)(2.0 / count);
        for (int e0 = blockIdx.y * TPB + threadIdx.x; e0 < per_t; e0 += LOSS_UNROLL * stride) {
            float s0[LOSS_UNROLL], s1[LOSS_UNROLL], od[LOSS_UNROLL];
#pragma unroll
            for (int u = 0; u < LOSS_UNROLL; ++u) {
                const int e = e0 + u * stride, ec = e < per_t ? e : e0;
                const int b = ec / N, i = ec - b * N;
                const size_t sidx = (size_t)b * sb + (size_t)t * st + i;   // states[b, t] (any batch / time strides)
                s0[u] = states[sidx];
                s1[u] = states[sidx + st];
                od[u] = d_all[((size_t)t * B + b) * N + i];
            }
#pragma unroll
            for (int u = 0; u < LOSS_UNROLL; ++u) {
                const int e = e0 + u * stride;
                if (e >= per_t) continue;
                const int b = e / N, i = e - b * N;
                const float dl = ((s1[u] - s0[u]) / delta - mean) / stdv;
                deltas[((size_t)b * (T - 1) + t) * N + i] = dl;
                const float err = od[u] - dl;
                if (dd_all) dd_all[((size_t)t * B + b) * N + i] = gscale * err;
                acc[0] += (double)(err * err);
                acc[1] += od[u];
                acc[2] += (double)od[u] * od[u];
                acc[3] += dl;
                acc[4] += (double)dl * dl;
            }
        }
    } else if (dd_all) {
        for (int e = blockIdx.y * TPB + threadIdx.x; e < per_t; e += stride) dd_all[(size_t)t * per_t + e] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < LOSS_NSUM; ++j) {
        const double w = wave_sum(acc[j]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = w;
    }
    __syncthreads();
    if (threadIdx.x < LOSS_NSUM) {
        double v = 0.0;
        for (int w = 0; w < TPB / 64; ++w) v += red[w][threadIdx.x];
        partial[((size_t)t * nsplit + blockIdx.y) * LOSS_NSUM + threadIdx.x] = v;
    }
    if (!take_ticket) return;   // the caller finishes the loss with delta_loss_finalize_kernel, off its critical path
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = (atomicAdd(ticket, 1u) == (unsigned)(T * nsplit - 1));
    __syncthreads();
    if (!last) return;
    __threadfence();
    delta_loss_finish(B, T, N, nsplit, hsteploss, loss, stats, partial, ticket);
}

// fixed-order reduction of the (T-1) x nsplit partial sums: by the workgroup that arrived last, or by a launch of its own
__device__ void delta_loss_finish(int B, int T, int N, int nsplit, float* __restrict__ hsteploss, float* __restrict__ loss,
                                  float* __restrict__ stats, double* __restrict__ partial, unsigned int* __restrict__ ticket) {
    const int per_t = B * N;
    const double count = (double)per_t * (T - 1);
    // fixed-order reduction of the (T-1) x nsplit partial sums by the workgroup that arrived last.  The partials are
    // fetched with relaxed device-scope atomic loads, one partial per thread, so the loads overlap (a serial loop
    // of volatile loads cost ~20 us here).
    auto peek = [&](size_t i) { return __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    __shared__ double tsum[LOSS_NSUM];
    __shared__ double stage[LOSS_NSUM][TPB];
    for (int tt = threadIdx.x; tt < T - 1; tt += blockDim.x) {
        double se = 0.0;
        for (int y = 0; y < nsplit; ++y) se += peek(((size_t)tt * nsplit + y) * LOSS_NSUM);
        hsteploss[tt] = (float)(se / per_t);
    }
    const int nq = (T - 1) * nsplit;
    double tot = 0.0;
    for (int q0 = 0; q0 < nq; q0 += TPB) {
        const int q = q0 + threadIdx.x;
#pragma unroll
        for (int j = 0; j < LOSS_NSUM; ++j) stage[j][threadIdx.x] = q < nq ? peek((size_t)q * LOSS_NSUM + j) : 0.0;
        __syncthreads();
        if (threadIdx.x < LOSS_NSUM) {
            const int lim = nq - q0 < TPB ? nq - q0 : TPB;
            for (int i = 0; i < lim; ++i) tot += stage[threadIdx.x][i];
        }
        __syncthreads();
    }
    if (threadIdx.x < LOSS_NSUM) tsum[threadIdx.x] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        *loss = (float)(tsum[0] / count);
        const double m_od = tsum[1] / count, m_dl = tsum[3] / count;
        stats[0] = (float)m_od;
        stats[1] = (float)sqrt(fmax(tsum[2] - count * m_od * m_od, 0.0) / (count - 1.0));  // unbiased, like Tensor.std()
        stats[2] = (float)m_dl;
        stats[3] = (float)sqrt(fmax(tsum[4] - count * m_dl * m_dl, 0.0) / (count - 1.0));
        *ticket = 0u;  // ready for the next launch (graph replay)
    }
}

__global__ void __launch_bounds__(TPB)
delta_loss_finalize_kernel(int B, int T, int N, int nsplit, float* __restrict__ hsteploss, float* __restrict__ loss,
                           float* __restrict__ stats, double* __restrict__ partial, unsigned int* __restrict__ ticket) {
    delta_loss_finish(B, T, N, nsplit, hsteploss, loss, stats, partial, ticket);
}

// ---------------------------------------------------------------------------------------------
// TBPTT chunk: K x (ConvLSTM cell + decoder + integration), time loop inside the kernel
// ---------------------------------------------------------------------------------------------
struct StepLayout {
    float *x, *h, *c, *gates, *cnew, *hnew, *p0, *a0, *p1, *a1, *p2, *a2, *d, *outv;
    // backward only
    float *dgates, *dh, *gA, *gB, *xh, *dx, *dhin, *dh_carry, *dc_carry, *dout_carry;
    float* end;
    int n;  // N = 4*hq
};

__host__ __device__ inline int step_max_act(const sur_chunk_params& p) {
    const int a = p.cs * 2 * p.hq, b = p.c_mid * 4 * p.hq;
    return a > b ? a : b;
}

// Floats of one step's forward intermediates as the forward kernel can save them for the backward kernel:
// the LDS block [gates .. a2] (activated gates, c_k, h_k, decoder pre-/post-LayerNorm activations), padded to
// whole 1 KiB pieces because the backward kernel fetches it by LDS-DMA (one wave instruction = 64 x 16 B).
constexpr int DMA_PIECE = 256;  // floats per global_load_lds_dwordx4 wave instruction
__host__ __device__ inline int step_block_floats(const sur_chunk_params& p) {
    const int s = p.cs * p.hq, n = 4 * p.hq;
    return 6 * s + 2 * p.cs * 2 * p.hq + 2 * p.c_mid * n + 2 * n;
}
__host__ __device__ inline int step_saved_floats(const sur_chunk_params& p) {
    return (step_block_floats(p) + DMA_PIECE - 1) / DMA_PIECE * DMA_PIECE;
}

// ConvLSTM cell with the gate non-linearities in the GEMM epilogue (4 * T waves, cs = 16): wave group w = wave & 3 owns
// channels 4w .. 4w+3 and its 16 tile rows are (channel, gate) pairs, so after the K loop lane (q, n) holds the four gate
// pre-activations of channel 4w+q at position n in its four accumulator registers -- i, f, g, o, c' and h' are finished
// in registers: no second pass over LDS, one barrier per step instead of two.  The hq / 16 column tiles are spread over
// the T = blockDim / 256 wave sets (the chain runs ONE workgroup per sample on a quarter of the CUs: at hq = 64 sixteen
// waves finish the recurrent step's GEMM in one tile each instead of four in a row).
template <int NSETS>
__device__ void cell_forward_fused(const sur_chunk_params& p, const StepLayout& L, const float* const* w) {
    const int s = p.cs * p.hq, hq = p.hq, ca = p.ca, cs = p.cs;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, tset = NSETS > 1 ? threadIdx.x >> 8 : 0;
    constexpr int nsets = NSETS;
    const int r = lane & 15, q = lane >> 4;
    const int gate_stride = (int)(w[SUR_ST_WXF] - w[SUR_ST_WXI]);
    const int gate_r = r & 3, ch_r = 4 * wave + (r >> 2);                 // identity of this lane's A row
    const float* ax = w[SUR_ST_WXI] + gate_r * gate_stride + ch_r * (ca * 3);   // W_g[(o*cin + ci)*3 + tap]
    const float* ah = w[SUR_ST_WHI] + gate_r * gate_stride + ch_r * (cs * 3);
    const int ch = 4 * wave + q;                                           // channel of this lane's accumulators
    const float bi = w[SUR_ST_BXI][ch], bf = (w[SUR_ST_BXI] + gate_stride)[ch], bc = (w[SUR_ST_BXI] + 2 * gate_stride)[ch],
                bo = (w[SUR_ST_BXI] + 3 * gate_stride)[ch];
    for (int n0 = 16 * tset; n0 < hq; n0 += 16 * nsets) {
        const int n = n0 + r;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int col = wrapi((n < hq ? n : 0) + tap - 1, hq);
            {   // latent action channels (ca <= 4: one K step)
                const bool ok = q < ca;
                const float a = ax[(ok ? q : 0) * 3 + tap], bv = L.x[(ok ? q : 0) * hq + col];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? a : 0.f, ok ? bv : 0.f, acc0, 0, 0, 0);
            }
#pragma unroll
            for (int c0 = 0; c0 < 16; c0 += 8) {   // hidden channels (cs = 16): two independent accumulators
                const float a0 = ah[(c0 + q) * 3 + tap], b0 = L.h[(c0 + q) * hq + col];
                const float a1 = ah[(c0 + 4 + q) * 3 + tap], b1 = L.h[(c0 + 4 + q) * hq + col];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
            }
        }
        if (n < hq) {
            const int idx = ch * hq + n;
            const float gi = sigmoid_(acc0[0] + acc1[0] + bi), gf = sigmoid_(acc0[1] + acc1[1] + bf),
                        gg = tanh_(acc0[2] + acc1[2] + bc), go = sigmoid_(acc0[3] + acc1[3] + bo);
            L.gates[idx] = gi;
            L.gates[s + idx] = gf;
            L.gates[2 * s + idx] = gg;
            L.gates[3 * s + idx] = go;
            const float cn = fmaf(gf, L.c[idx], gi * gg);
            L.cnew[idx] = cn;
            L.hnew[idx] = go * tanh_(cn);
        }
    }
    __syncthreads();
}

// ConvLSTM cell on LDS-resident x, h, c: fills gates (activated), cnew, hnew.  BLOCK = blockDim.x when it is known at
// compile time (the chain kernels), 0 otherwise.
template <int BLOCK = 0>
__device__ void cell_forward(const sur_chunk_params& p, const StepLayout& L, const float* const* w) {
    const int s = p.cs * p.hq;
    if constexpr (BLOCK >= 256 && (BLOCK & 255) == 0) {
        if (p.cs == 16 && p.ca <= 4) {
            cell_forward_fused<BLOCK / 256>(p, L, w);
            return;
        }
    } else {
        if (blockDim.x == 256 && p.cs == 16 && p.ca <= 4) {
            cell_forward_fused<1>(p, L, w);
            return;
        }
    }
    STAMP(0);
    // all four gates' pre-activations in ONE gather-GEMM: rows = (gate, channel), K = 3*(ca + cs).
    // The four gates' weights are consecutive in the LDS copy (Wx_g, b_g, Wh_g per gate), so the
    // row stride between gates is constant.
    {
        const int gate_stride = (int)(w[SUR_ST_WXF] - w[SUR_ST_WXI]);
        const GemmSeg seg[2] = {{w[SUR_ST_WXI], p.ca * 3, 3, L.x, p.hq, p.ca}, {w[SUR_ST_WHI], p.cs * 3, 3, L.h, p.hq, p.cs}};
        const int cs = p.cs, hq = p.hq, ca = p.ca;
        // row m = g*cs + o lives at gate g's weight block: fold the gate offset into a_sm arithmetic by
        // running one GEMM per gate tile row-block (cs is a multiple of 16 for this model family)
        const int nwg = blockDim.x >> 6;
        for (int g = 0; g < 4; ++g) {
            const GemmSeg sg[2] = {{seg[0].a + g * gate_stride, ca * 3, 3, L.x, hq, ca},
                                   {seg[1].a + g * gate_stride, cs * 3, 3, L.h, hq, cs}};
            const float* bias = w[SUR_ST_BXI] + g * gate_stride;
            float* outg = L.gates + g * s;
            // gate g on its own wave quarter (all waves if the workgroup has fewer than 4)
            const WaveSet ws = nwg >= 4 ? WaveSet{g * (nwg / 4), nwg / 4} : WaveSet{0, nwg};
            gemm_taps<3, 2>(ws, false, cs, hq, sg, [&](int, int tap, int n) { return wrapi(n + tap - 1, hq); },
                            [&](int m, int n, float v) {
                                if (m < cs) outg[m * hq + n] = v + bias[m];
                            });
        }
        __syncthreads();
    }
    STAMP(1);
    for (int i = threadIdx.x; i < s; i += blockDim.x) {
        const float gi = sigmoid_(L.gates[i]), gf = sigmoid_(L.gates[s + i]), gg = tanh_(L.gates[2 * s + i]),
                    go = sigmoid_(L.gates[3 * s + i]);
        L.gates[i] = gi;
        L.gates[s + i] = gf;
        L.gates[2 * s + i] = gg;
        L.gates[3 * s + i] = go;
        const float cn = fmaf(gf, L.c[i], gi * gg);
        L.cnew[i] = cn;
        L.hnew[i] = go * tanh_(cn);
    }
    __syncthreads();
    STAMP(2);
}

// state decoder on LDS-resident hnew: fills p0, a0, p1, a1, p2, a2 and d
__device__ void decoder_forward(const sur_chunk_params& p, const StepLayout& L, const float* const* w) {
    deconv_fwd(L.hnew, p.cs, p.hq, w[SUR_ST_DC0_W], w[SUR_ST_DC0_B], p.cs, L.p0);
    STAMP(3);
    act_ln_fwd(L.p0, p.cs, 2 * p.hq, w[SUR_ST_LN0_W], w[SUR_ST_LN0_B], true, L.a0);
    STAMP(4);
    deconv_fwd(L.a0, p.cs, 2 * p.hq, w[SUR_ST_DC1_W], w[SUR_ST_DC1_B], p.c_mid, L.p1);
    STAMP(5);
    act_ln_fwd(L.p1, p.c_mid, L.n, w[SUR_ST_LN1_W], w[SUR_ST_LN1_B], true, L.a1);
    STAMP(6);
    conv_fwd<7>(L.a1, p.c_mid, L.n, w[SUR_ST_CV2_W], w[SUR_ST_CV2_B], 1, 1, 3, L.p2, false);
    STAMP(7);
    act_ln_fwd(L.p2, 1, L.n, w[SUR_ST_LN2_W], w[SUR_ST_LN2_B], true, L.a2);
    STAMP(8);
    conv_fwd<5>(L.a2, 1, L.n, w[SUR_ST_CV3_W], w[SUR_ST_CV3_B], 1, 1, 2, L.d, false);
    STAMP(9);
}

// decoder backward on LDS-resident activations: L.gA holds d loss / d d on entry, L.dh the gradient wrt hnew on exit.
// LDS is what limits the (step, sample)-parallel kernel's residency at N = 256, so nothing is allocated that a dead buffer
// can serve (dec_bwd_kernel lays the buffers out): the LayerNorm backward's normalised-activation scratch lives in the
// post-LayerNorm activation that was consumed just before (a2 for the last norm, a1 -- dead after the 7-tap weight
// gradient -- for the other two), L.dh aliases L.gB, and two activations arrive LATE from registers: p0 into p1's
// buffer once the middle norm's backward has read p1 (`late_p0`), h into a0's buffer once the second transposed
// convolution's weight gradient has read a0 (`late_h`).  Each is stored right after a barrier that retires the old
// contents and at least one barrier before its first reader.
template <class GP, typename StoreP0, typename StoreH>
__device__ void decoder_backward(const sur_chunk_params& p, const StepLayout& L, const float* const* w, const GP* g,
                                 StoreP0 late_p0, StoreH late_h) {
    const int n = L.n;
    // single-channel gradients fill the first n floats of gA: the rest of the buffer is the weight gradients' scratch
    float* const wg_scratch = L.gA + n;
    const int wg_floats = step_max_act(p) - n;
    conv_bwd_weight<5>(L.gA, 1, L.a2, 1, n, 1, 2, g[SUR_ST_CV3_W], g[SUR_ST_CV3_B], all_waves(), false, wg_scratch, wg_floats);
    STAMP(12);
    conv_bwd_data<5>(L.gA, 1, n, w[SUR_ST_CV3_W], 1, 1, 2, L.gB, false);
    STAMP(13);
    act_ln_bwd(L.gB, L.p2, 1, n, w[SUR_ST_LN2_W], true, L.gA, L.a2, g[SUR_ST_LN2_W], g[SUR_ST_LN2_B]);
    STAMP(14);
    conv_bwd_weight<7>(L.gA, 1, L.a1, p.c_mid, n, 1, 3, g[SUR_ST_CV2_W], g[SUR_ST_CV2_B], all_waves(), false, wg_scratch, wg_floats);
    STAMP(15);
    conv_bwd_data<7>(L.gA, 1, n, w[SUR_ST_CV2_W], p.c_mid, 1, 3, L.gB, false);
    STAMP(16);
    act_ln_bwd(L.gB, L.p1, p.c_mid, n, w[SUR_ST_LN1_W], true, L.gA, L.a1, g[SUR_ST_LN1_W], g[SUR_ST_LN1_B]);
    STAMP(17);
    late_p0();      // p1 is dead (act_ln_bwd ends with a barrier); first reader: the act_ln_bwd two barriers below
    deconv_bwd_weight(L.gA, p.c_mid, L.a0, p.cs, 2 * p.hq, g[SUR_ST_DC1_W], g[SUR_ST_DC1_B], lower_half(), false);
    STAMP(18);
    deconv_bwd_data(L.gA, p.c_mid, 2 * p.hq, w[SUR_ST_DC1_W], p.cs, L.gB, upper_half(), true);
    STAMP(19);
    late_h();       // a0 is dead; first reader: deconv_bwd_weight below, behind act_ln_bwd's barrier
    act_ln_bwd(L.gB, L.p0, p.cs, 2 * p.hq, w[SUR_ST_LN0_W], true, L.gA, L.a1, g[SUR_ST_LN0_W], g[SUR_ST_LN0_B]);
    STAMP(26);
    deconv_bwd_weight(L.gA, p.cs, L.hnew, p.cs, p.hq, g[SUR_ST_DC0_W], g[SUR_ST_DC0_B], lower_half(), false);
    STAMP(27);
    deconv_bwd_data(L.gA, p.cs, p.hq, w[SUR_ST_DC0_W], p.cs, L.dh, upper_half(), true);

}

// Cell backward, recurrent part: dh_in [cs][hq] = sum_g Wh_g^T * dG_g.  The K = 4 * cs * 3 contraction is split by gate
// over the wave quarters (each writes its partial tile to part[g]); the caller adds the four partials.
__device__ void cell_dh_gemm(const sur_chunk_params& p, const StepLayout& L, const float* const* w, float* part) {
    const int s = p.cs * p.hq, nwg = blockDim.x >> 6, cs = p.cs, hq = p.hq;
    const int gate_stride = (int)(w[SUR_ST_WXF] - w[SUR_ST_WXI]);
    auto tap_col = [&](int, int tap, int j) { return wrapi(j - tap + 1, hq); };  // stride 1, pad 1
    const float* wh = w[SUR_ST_WHI];
    for (int gt = 0; gt < 4; ++gt) {   // A[m=ci][c=o][tap] = W_g[(o*cin + ci)*3 + tap];  B = dG_g[o][col]
        const GemmSeg sh[1] = {{wh + gt * gate_stride, 3, cs * 3, L.dgates + gt * s, hq, cs}};
        float* dst = part + gt * s;
        const WaveSet ws = nwg >= 4 ? WaveSet{gt * (nwg / 4), nwg / 4} : all_waves();
        gemm_taps<3, 1>(ws, false, cs, hq, sh, tap_col, [&](int m, int j, float v) {
            if (m < cs) dst[m * hq + j] = v;
        });
    }
    __syncthreads();
}

// Cell backward, non-recurrent part for one (step, sample): dx [ca][hq] = sum_g Wx_g^T * dG_g and the LSTM weight /
// bias gradients dG_g x {x, h_in}.  dx on the first wave, the weight-gradient tiles on the others.
// `hs`: floats between consecutive rows of L.x, L.h and L.dgates (>= hq).  With hs = hq = 64 the sixteen tile rows a GEMM operand
// is gathered from start in the same LDS bank (SQ_LDS_BANK_CONFLICT was 79 % of the LDS-active cycles of this kernel);
// hs = hq + 4 staggers them by four banks and keeps rows 16-byte aligned.
template <class GP>
__device__ void cell_wgrad_gemms(const sur_chunk_params& p, const StepLayout& L, const float* const* w, const GP* g, int hs) {
    const int s = p.cs * hs, nwg = blockDim.x >> 6, cs = p.cs, ca = p.ca, hq = p.hq;
    const int gate_stride = (int)(w[SUR_ST_WXF] - w[SUR_ST_WXI]);     // between the gates' Wx as staged (cell_wgrad_kernel: back to back)
    const int ggate_stride = (int)(g[SUR_ST_WXF] - g[SUR_ST_WXI]);    // between the gates' accumulators (parameter order)
    const WaveSet w_dx = nwg >= 4 ? WaveSet{0, 1} : all_waves();
    const WaveSet w_gw = nwg >= 4 ? WaveSet{1, nwg - 1} : all_waves();
    auto tap_col = [&](int, int tap, int j) { return wrapi(j - tap + 1, hq); };
    {
        const float* wx = w[SUR_ST_WXI];
        const GemmSeg sx[4] = {{wx, 3, ca * 3, L.dgates, hs, cs},
                               {wx + gate_stride, 3, ca * 3, L.dgates + s, hs, cs},
                               {wx + 2 * gate_stride, 3, ca * 3, L.dgates + 2 * s, hs, cs},
                               {wx + 3 * gate_stride, 3, ca * 3, L.dgates + 3 * s, hs, cs}};
        float* dxp = L.dx;
        gemm_taps<3, 4>(w_dx, false, ca, hq, sx, tap_col, [&](int m, int j, float v) {
            if (m < ca) dxp[m * hq + j] = v;
        });
    }
    struct St { const float* row; int off; };
    {   // weight gradients, all gates in one GEMM each: rows m = (gate, o)
        const GP gx = g[SUR_ST_WXI];
        const float* xin = L.x;
        const int ncols = ca * 3;
        gemm_pos(w_gw, false, 4 * cs, ncols, hq, L.dgates, hs,
                 [&](int n) { const int ci = n / 3, k = n - ci * 3; return St{xin + ci * hs, k - 1}; },
                 [&](const St& st, int pp) { return st.row[wrapi(pp + st.off, hq)]; },
                 [&](int m, int n, float v) {
                     if (m < 4 * cs && n < ncols) {
                         const int gt = m / cs, o = m - gt * cs;
                         gx[gt * ggate_stride + o * ncols + n] += v;
                     }
                 });
        const GP gh = g[SUR_ST_WHI];
        const float* hin_ = L.h;
        const int ncols_h = cs * 3;
        gemm_pos(w_gw, false, 4 * cs, ncols_h, hq, L.dgates, hs,
                 [&](int n) { const int ci = n / 3, k = n - ci * 3; return St{hin_ + ci * hs, k - 1}; },
                 [&](const St& st, int pp) { return st.row[wrapi(pp + st.off, hq)]; },
                 [&](int m, int n, float v) {
                     if (m < 4 * cs && n < ncols_h) {
                         const int gt = m / cs, o = m - gt * cs;
                         gh[gt * ggate_stride + o * ncols_h + n] += v;
                     }
                 });
        // bias gradients: one 16-lane group per (gate, channel) row of dG
        const GP gbx = g[SUR_ST_BXI];
        const int group = threadIdx.x >> 4, gl = threadIdx.x & 15, ngroups = blockDim.x >> 4;
        for (int r0 = 0; r0 < 4 * cs; r0 += ngroups) {
            const int r = r0 + group;
            float a0 = 0.0f, a1 = 0.0f;
            if (r < 4 * cs)
                for (int pp = gl; pp < hq; pp += 32) {
                    a0 += L.dgates[r * hs + pp];
                    if (pp + 16 < hq) a1 += L.dgates[r * hs + pp + 16];
                }
            group_sum2(a0, a1, 16);
            if (gl == 0 && r < 4 * cs) {
                const int gt = r / cs, o = r - gt * cs;
                gbx[gt * ggate_stride + o] += a0 + a1;
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Split path: only the ConvLSTM cell is a recurrence.  The decoder of step k reads h_k and feeds nothing back
// into step k+1 (the reference's transition ignores the re-encoded prediction, transition.py:285-296; teacher
// forcing replaces H by the encoded true state), so it is evaluated for ALL (step, sample) pairs in parallel,
// forward and backward, by persistent workgroups on every CU, while the chain kernels (one workgroup per
// sample) carry only the cell.  Layout of a saved block (step_saved_floats per (step, sample)):
//   [ gates 4s | c_k s | h_k s | p0 | a0 | p1 | a1 | p2 | a2 ]     cell part = first 6s floats
// ---------------------------------------------------------------------------------------------
#ifndef PAR_OCC
#define PAR_OCC 2   // resident workgroups per CU the (step, sample)-parallel kernels are built for
#endif
constexpr int ST_NLSTM = SUR_ST_DC0_W;              // the LSTM parameters come first in the chunk parameter order
constexpr int ST_NDEC = SUR_ST_NPARAM - ST_NLSTM;

__host__ __device__ inline int cell_part_floats(const sur_chunk_params& p) { return 6 * p.cs * p.hq; }
// threads of a cell-chain workgroup: one 256-thread wave set per 16-wide column tile of the latent, at most four
inline int cell_chain_threads(const sur_chunk_params& p) {
    const int sets = p.hq / 16;
    return TPB * (sets < 1 ? 1 : (sets > 4 ? 4 : sets));
}
__host__ __device__ inline int dec_part_floats(const sur_chunk_params& p) { return step_block_floats(p) - cell_part_floats(p); }
__host__ __device__ inline int cell_fwd_act_floats(const sur_chunk_params& p) { return p.ca * p.hq + 8 * p.cs * p.hq; }
__host__ __device__ inline int cell_bwd_act_floats(const sur_chunk_params& p) {
    const int s = p.cs * p.hq;   // [gates | c_k] twice (ping-pong DMA targets), dgates, four dh partials
    return 2 * 5 * s + 4 * s + 4 * s;
}
__host__ __device__ inline int cell_wgrad_act_floats(const sur_chunk_params& p) {
    return p.ca * p.hq + (p.ca + 5 * p.cs) * (p.hq + 4);   // dx | x, h_in, dgates with rows hq + 4 apart (cell_wgrad_gemms)
}
__host__ __device__ inline int dec_act_floats(const sur_chunk_params& p, bool backward) {
    const int s = p.cs * p.hq, n = 4 * p.hq;
    // backward (dec_bwd_kernel): three activation buffers [a0 -> h | p1 -> p0 | a1], gA, gB (dh, and p2 / a2 in its tail)
    if (backward) return 5 * step_max_act(p);
    return s + dec_part_floats(p) + n;                                    // hnew, p0..a2, d
}

template <int NP>
__device__ __forceinline__ void stage_range(const sur_chunk_params& p, int first, float* lds_w, const float** w) {
    ParamViews<NP> v;
    stage_weights<NP>(p.w + first, p.size + first, lds_w, v);
#pragma unroll
    for (int i = 0; i < NP; ++i) w[first + i] = v.w[i];
}

// (the chain kernels are instantiated per workgroup size: the 256-thread build keeps its register budget)
// The whole forward chain of one sample with the state in registers (cs = 16, ca <= 4, hq = 16 * NSETS: exactly one
// (channel, position) element per thread).  Per step and lane: 15 MFMAs whose A operands -- the gate weights -- were read
// from global memory ONCE into 15 registers before the time loop, B operands gathered from the x / h buffers in LDS;
// i, f, g, o, c', h' finished in the accumulator registers (same row <-> (channel, gate) map and the same summation order
// as cell_forward_fused), c carried in a register, h' and the next step's x written to the OTHER of two LDS buffers and
// every output stored to HBM straight from registers: ONE barrier per step, no LDS copy of h / c / gates, no weights in
// LDS.  The next step's latent action and (teacher forcing) encoded state are fetched one step ahead.
template <int NSETS>
__device__ __forceinline__ void cell_chain_forward(const sur_chunk_params& p, float* lds, const float* __restrict__ xlat_t,
                                                   const float* __restrict__ lstates_t, const float* __restrict__ h0,
                                                   const float* __restrict__ c0, int hc_bstride, int K, int S, int B, int b,
                                                   float* __restrict__ h_all, float* __restrict__ c_all,
                                                   float* __restrict__ saved) {
    const int hq = p.hq, ca = p.ca, s = 16 * hq, nx = ca * hq;
    // LDS-typed pointers: through a generic `float*` picked from an array by (k & 1) the operand reads of the time loop were
    // flat_load_dword -- 64-bit address arithmetic per read and, because a flat access may be a global one, an
    // s_waitcnt vmcnt(0) in front of every MFMA group: each step waited for its own prefetch and the previous step's stores
    lds_f* const xbase = (lds_f*)lds;
    lds_f* const hbase = xbase + 2 * nx;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, tset = NSETS > 1 ? tid >> 8 : 0;
    const int r = lane & 15, q = lane >> 4;
    const int n = 16 * tset + r, ch = 4 * wave + q, idx = ch * hq + n;
    // A operands: row r of this wave's tile = (gate r & 3, channel 4 * wave + (r >> 2)); lane (r, q) feeds k = q of each K block
    const int gate_r = r & 3, ch_r = 4 * wave + (r >> 2);
    const bool xq = q < ca;
    float ax[3], ah[4][3];
    {
        const float* wx = p.w[SUR_ST_WXI + 3 * gate_r] + (ch_r * ca + (xq ? q : 0)) * 3;   // W_g[(o*cin + ci)*3 + tap]
        const float* wh = p.w[SUR_ST_WHI + 3 * gate_r] + ch_r * 16 * 3;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            ax[tap] = xq ? wx[tap] : 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) ah[c4][tap] = wh[(4 * c4 + q) * 3 + tap];
        }
    }
    const float bi = p.w[SUR_ST_BXI][ch], bf = p.w[SUR_ST_BXF][ch], bc = p.w[SUR_ST_BXC][ch], bo = p.w[SUR_ST_BXO][ch];
    float c_reg = c0[(size_t)b * hc_bstride + idx];
    hbase[idx] = S > 0 ? lstates_t[(size_t)b * s + idx] : h0[(size_t)b * hc_bstride + idx];
    if (tid < nx) xbase[tid] = xlat_t[(size_t)b * nx + tid];
    const size_t save_stride = step_saved_floats(p);
    int colx[3];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) colx[tap] = wrapi(n + tap - 1, hq);
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        const size_t kb = (size_t)k * B + b;
        const bool nxt = k + 1 < K, next_forced = nxt && k + 1 < S;
        float xnext = 0.0f, hforced = 0.0f;
        if (nxt && tid < nx) xnext = xlat_t[(kb + B) * nx + tid];
        if (next_forced) hforced = lstates_t[(kb + B) * s + idx];
        const lds_f* xc = xbase + (k & 1) * nx;
        const lds_f* hc = hbase + (k & 1) * s;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int col = colx[tap];
            const float bx = xc[(xq ? q : 0) * hq + col];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[tap], xq ? bx : 0.f, acc0, 0, 0, 0);
#pragma unroll
            for (int c0_ = 0; c0_ < 4; c0_ += 2) {
                const float b0 = hc[(4 * c0_ + q) * hq + col], b1 = hc[(4 * c0_ + 4 + q) * hq + col];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[c0_][tap], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[c0_ + 1][tap], b1, acc1, 0, 0, 0);
            }
        }
        const float gi = sigmoid_(acc0[0] + acc1[0] + bi), gf = sigmoid_(acc0[1] + acc1[1] + bf),
                    gg = tanh_(acc0[2] + acc1[2] + bc), go = sigmoid_(acc0[3] + acc1[3] + bo);
        const float cn = fmaf(gf, c_reg, gi * gg);
        const float hn = go * tanh_(cn);
        c_reg = cn;
        h_all[kb * s + idx] = hn;
        c_all[kb * s + idx] = cn;
        if (saved) {   // [gates | c_k | h_k]
            float* dst = saved + kb * save_stride + idx;
            dst[0] = gi;
            dst[s] = gf;
            dst[2 * s] = gg;
            dst[3 * s] = go;
            dst[4 * s] = cn;
            dst[5 * s] = hn;
        }
        if (nxt) {
            hbase[((k + 1) & 1) * s + idx] = next_forced ? hforced : hn;   // teacher forcing replaces H
            if (tid < nx) xbase[((k + 1) & 1) * nx + tid] = xnext;
        }
        __syncthreads();
    }
}

template <int MAXT>
__global__ void __launch_bounds__(MAXT)
cell_fwd_kernel(const sur_chunk_params p, const float* __restrict__ xlat_t, const float* __restrict__ lstates_t,
                const float* __restrict__ h0, const float* __restrict__ c0, int hc_bstride, int K, int S, int B,
                float* __restrict__ h_all, float* __restrict__ c_all, float* __restrict__ saved) {
    extern __shared__ __align__(16) float lds[];
    const int b = blockIdx.x, s = p.cs * p.hq, nx = p.ca * p.hq;
#ifndef SUR_STAMP
    if constexpr ((MAXT & 255) == 0) {
        if (p.cs == 16 && p.ca <= 4 && p.hq * 16 == MAXT) {   // one (channel, position) element per thread
            cell_chain_forward<MAXT / 256>(p, lds, xlat_t, lstates_t, h0, c0, hc_bstride, K, S, B, b, h_all, c_all, saved);
            return;
        }
    }
#endif
    StepLayout L{};
    L.x = lds;
    L.h = L.x + nx;
    L.c = L.h + s;
    L.gates = L.c + s;
    L.cnew = L.gates + 4 * s;
    L.hnew = L.cnew + s;
    const float* w[SUR_ST_NPARAM];
    stage_range<ST_NLSTM>(p, 0, L.hnew + s, w);
    const size_t save_stride = step_saved_floats(p);
    for (int i = threadIdx.x; i < s; i += blockDim.x) {
        L.hnew[i] = h0[(size_t)b * hc_bstride + i];   // hc_bstride = 0: one initial state shared by the batch
        L.cnew[i] = c0[(size_t)b * hc_bstride + i];
    }
    STAMP(109);
    // the next step's latent action is fetched while this step computes
    float xn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        xn[u] = threadIdx.x + u * MAXT < nx ? xlat_t[(size_t)b * nx + threadIdx.x + u * MAXT] : 0.0f;
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        const size_t kb = (size_t)k * B + b;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (threadIdx.x + u * MAXT < nx) L.x[threadIdx.x + u * MAXT] = xn[u];
        for (int i = threadIdx.x + 4 * MAXT; i < nx; i += MAXT) L.x[i] = xlat_t[kb * nx + i];
        for (int i = threadIdx.x; i < s; i += blockDim.x) {
            L.h[i] = (k < S) ? lstates_t[kb * s + i] : L.hnew[i];  // teacher forcing replaces H
            L.c[i] = L.cnew[i];
        }
        __syncthreads();
        STAMP(k < S ? 110 : 111);
        if (k + 1 < K) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (threadIdx.x + u * MAXT < nx) xn[u] = xlat_t[(kb + B) * nx + threadIdx.x + u * MAXT];
        }
        cell_forward<MAXT>(p, L, w);
        STAMP(112);
        for (int i = threadIdx.x; i < s; i += blockDim.x) {
            h_all[kb * s + i] = L.hnew[i];
            c_all[kb * s + i] = L.cnew[i];
        }
        if (saved) {   // [gates | c_k | h_k]
            float4* dst = reinterpret_cast<float4*>(saved + kb * save_stride);
            const float4* src = reinterpret_cast<const float4*>(L.gates);
            for (int i = threadIdx.x; i < (6 * s) >> 2; i += blockDim.x) dst[i] = src[i];
        }
        __syncthreads();
        STAMP(113);
    }
}

// decoder of every (step, sample) pair m: d_all[m] from h_all[m]; persistent workgroups
__global__ void __launch_bounds__(TPB, PAR_OCC)
dec_fwd_kernel(const sur_chunk_params p, const float* __restrict__ h_all, int M, float* __restrict__ d_all,
               float* __restrict__ saved) {
    extern __shared__ __align__(16) float lds[];
    const int s = p.cs * p.hq, n = 4 * p.hq;
    StepLayout L{};
    L.n = n;
    L.hnew = lds;
    L.p0 = L.hnew + s;
    L.a0 = L.p0 + p.cs * 2 * p.hq;
    L.p1 = L.a0 + p.cs * 2 * p.hq;
    L.a1 = L.p1 + p.c_mid * n;
    L.p2 = L.a1 + p.c_mid * n;
    L.a2 = L.p2 + n;
    L.d = L.a2 + n;
    const float* w[SUR_ST_NPARAM];
    stage_range<ST_NDEC>(p, ST_NLSTM, L.d + n, w);
    const size_t save_stride = step_saved_floats(p);
    const int ndec4 = dec_part_floats(p) >> 2;
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        lds_load_v4(L.hnew, h_all + (size_t)m * s, s >> 2);
        decoder_forward(p, L, w);
        for (int i = threadIdx.x; i < n; i += blockDim.x) d_all[(size_t)m * n + i] = L.d[i];
        if (saved) {
            float4* dst = reinterpret_cast<float4*>(saved + (size_t)m * save_stride + 6 * s);
            const float4* src = reinterpret_cast<const float4*>(L.p0);
            for (int i = threadIdx.x; i < ndec4; i += blockDim.x) dst[i] = src[i];
        }
        __syncthreads();
    }
}

// out_k = base_k + delta * (d_k * mul + add); base_k = the given state while teacher forcing, else out_{k-1}
__global__ void __launch_bounds__(TPB)
integrate_kernel(const float* __restrict__ states_t, const float* __restrict__ d_all, int K, int S, int B, int n, float delta,
                 float mul, float add, float* __restrict__ out_all) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * n) return;
    float prev = 0.0f;
    for (int k = 0; k < K; ++k) {
        const size_t idx = (size_t)k * B * n + e;
        const float base = k < S ? states_t[idx] : prev;
        prev = base + delta * fmaf(d_all[idx], mul, add);
        out_all[idx] = prev;
    }
}

// total gradient wrt d_k: direct + through out_k (and through the outputs after it while free running)
__global__ void __launch_bounds__(TPB)
dgrad_scan_kernel(const float* __restrict__ dd_all, const float* __restrict__ dout_all, int K, int S, int B, int n, float scale,
                  float* __restrict__ ga_all) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * n) return;
    float carry = 0.0f;
    for (int k = K - 1; k >= 0; --k) {
        const size_t idx = (size_t)k * B * n + e;
        const float go = (dout_all ? dout_all[idx] : 0.0f) + carry;
        ga_all[idx] = fmaf(scale, go, dd_all ? dd_all[idx] : 0.0f);
        carry = (k >= S) ? go : 0.0f;   // out_{k-1} is the base of step k only while free running
    }
}

// decoder backward of every (step, sample) pair: dh_dec[m] = d loss / d h_m through the decoder; decoder parameter
// gradients into this workgroup's partial row.  Built for DEC_BWD_OCC workgroups per CU: at N = 256 the LDS layout below
// is 52 KB (round 2: 75 KB, two per CU), so three workgroups share a CU and the 640 pairs of a 10-step chunk are ONE round
// of workgroups.  Saved block of a pair: [gates 4s | c s | h s | p0 | a0 | p1 | a1 | p2 | a2].
//   LDS: X2 = a0 (later h) | X1 = p1 (later p0) | X0 = a1 | gA | gB (dh; p2, a2 in its last 2n floats: phase 1 only uses
//        the first n floats of gB, and p2 / a2 are dead when the 7-tap data gradient fills it)
// p0 and h wait in registers (2 + 1 float4 per thread at N = 256) until their buffers are free (decoder_backward).
#ifndef DEC_BWD_OCC
#define DEC_BWD_OCC 3
#endif
constexpr int DEC_LATE_P0 = 2;   // float4 registers per thread for the late p0: cs * 2 hq <= 2 * 4 * TPB
constexpr int DEC_LATE_H = 1;    //                                  ... for the late h:  cs * hq <= 4 * TPB
// GL: the gradient accumulators are in LDS (behind the staged weights) -- known at compile time, so that they are LDS-typed
// pointers (see conv_bwd_weight); GL = false accumulates straight into the workgroup's partial row.
template <bool GL>
__global__ void __launch_bounds__(TPB, DEC_BWD_OCC)
dec_bwd_kernel(const sur_chunk_params p, const float* __restrict__ saved, const float* __restrict__ ga_all, int M,
               float* __restrict__ dh_dec, int row_base) {
    extern __shared__ __align__(16) float lds[];
    const int s = p.cs * p.hq, n = 4 * p.hq, mx = step_max_act(p);
    const int a0f = p.cs * 2 * p.hq, a1f = p.c_mid * n;
    StepLayout L{};
    L.n = n;
    float* X2 = lds;
    float* X1 = X2 + mx;
    float* X0 = X1 + mx;
    L.a0 = X2;
    L.hnew = X2;
    L.p1 = X1;
    L.p0 = X1;
    L.a1 = X0;
    L.gA = X0 + mx;
    L.gB = L.gA + mx;
    L.dh = L.gB;              // s <= mx: the first LayerNorm's backward has consumed gB before dh is written
    L.p2 = L.gB + mx - 2 * n;
    L.a2 = L.p2 + n;
    float* wbase = L.gB + mx;
    const float* w[SUR_ST_NPARAM];
    stage_range<ST_NDEC>(p, ST_NLSTM, wbase, w);
    int off_dec = 0, psize_dec = 0;
    for (int i = 0; i < ST_NLSTM; ++i) off_dec += p.size[i];
    for (int i = ST_NLSTM; i < SUR_ST_NPARAM; ++i) psize_dec += p.size[i];
    const int psize = off_dec + psize_dec;
    float* row = p.partial + (size_t)(row_base + blockIdx.x) * psize + off_dec;
    typedef typename std::conditional<GL, lds_f*, float*>::type GP;
    GP gacc;
    if constexpr (GL) gacc = (lds_f*)(wbase + psize_dec);
    else gacc = row;
    GP g[SUR_ST_NPARAM];
    {
        int off = 0;
        for (int i = ST_NLSTM; i < SUR_ST_NPARAM; ++i) {
            g[i] = gacc + off;
            off += p.size[i];
        }
        if (GL)
            for (int j = threadIdx.x; j < psize_dec; j += blockDim.x) gacc[j] = 0.0f;
    }
    __syncthreads();
    const size_t save_stride = step_saved_floats(p);
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        const float* blk = saved + (size_t)m * save_stride + 5 * s;        // [h | p0 | a0 | p1 | a1 | p2 | a2]
        // the late activations start their trip now and land in registers
        float4 rp0[DEC_LATE_P0], rh[DEC_LATE_H];
#pragma unroll
        for (int u = 0; u < DEC_LATE_P0; ++u) {
            const int i = threadIdx.x + u * TPB;
            rp0[u] = i < (a0f >> 2) ? reinterpret_cast<const float4*>(blk + s)[i] : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < DEC_LATE_H; ++u) {
            const int i = threadIdx.x + u * TPB;
            rh[u] = i < (s >> 2) ? reinterpret_cast<const float4*>(blk)[i] : float4{0.f, 0.f, 0.f, 0.f};
        }
        for (int i = threadIdx.x; i < n; i += blockDim.x) L.gA[i] = ga_all[(size_t)m * n + i];
        for (int i = threadIdx.x; i < (2 * n) >> 2; i += blockDim.x)
            reinterpret_cast<float4*>(L.p2)[i] = reinterpret_cast<const float4*>(blk + s + 2 * a0f + 2 * a1f)[i];
        for (int i = threadIdx.x; i < (a0f >> 2); i += blockDim.x)
            reinterpret_cast<float4*>(X2)[i] = reinterpret_cast<const float4*>(blk + s + a0f)[i];
        lds_load_v4(X1, blk + s + 2 * a0f, a1f >> 2);                      // p1 (ends with a barrier)
        lds_load_v4(X0, blk + s + 2 * a0f + a1f, a1f >> 2);                // a1
        decoder_backward(p, L, w, g,
                         [&] {
#pragma unroll
                             for (int u = 0; u < DEC_LATE_P0; ++u) {
                                 const int i = threadIdx.x + u * TPB;
                                 if (i < (a0f >> 2)) reinterpret_cast<float4*>(X1)[i] = rp0[u];
                             }
                         },
                         [&] {
#pragma unroll
                             for (int u = 0; u < DEC_LATE_H; ++u) {
                                 const int i = threadIdx.x + u * TPB;
                                 if (i < (s >> 2)) reinterpret_cast<float4*>(X2)[i] = rh[u];
                             }
                         });
        for (int i = threadIdx.x; i < s; i += blockDim.x) dh_dec[(size_t)m * s + i] = L.dh[i];
        __syncthreads();
    }
    if constexpr (GL) add_to_row(row, (const float*)gacc, psize_dec);
}

// BPTT through the cell chain of one sample: consumes dh_dec (decoder) and the upstream dh / dc gradients, emits the
// gate gradients dG_k of every step (for cell_wgrad_kernel) and the gradient wrt the teacher-forced hidden inputs.
// Only what is recurrent stays here: the gate derivative and dh_{k-1} = sum_g Wh_g^T dG_g.  Two barriers per step:
//   A  element-wise: every thread owns the same elements in every step, so dc_carry lives in registers; the incoming
//      dh is the sum of the four partial tiles the previous step's GEMM left in LDS; the next step's c_{k-1} and
//      dh_dec were prefetched into registers, its [gates | c] block arrived by LDS-DMA in the other block buffer
//   B  issue the prefetches for step k-1, then the dh GEMM (K split by gate over the wave quarters)
constexpr int CELL_EPT = 4;   // elements per thread: cs * hq <= 4 * 256
struct ChunkSpans {
    sur_chunk_span sp[SUR_MAX_SPANS];
    int n;
};
// The backward chain of one (chunk, sample) for cs = 16, hq = 16 * NSETS (one (channel, position) element per thread), built
// like cell_chain_forward: the recurrent weights Wh_g^T are read from global memory ONCE into 12 registers per lane (wave
// group g = wave / NSETS contracts gate g, K = 48, for the column tile wave % NSETS), LDS holds only the gate derivatives
// and the four partial dh tiles, and everything a step reads from HBM -- its activated gates and c_k (saved by the forward),
// c_{k-1}, the decoder's dh -- is fetched TWO steps ahead into registers, so no step waits for memory (with one step of
// lead, LDS-DMA + prefetch, the loads landed ~1 us after they were needed).  Two barriers per step remain: the gate
// derivatives of all channels feed every wave's GEMM, and the four gates' partial tiles are summed by the next step.
struct CellBwdIn {
    float g[5], cp, dhd;   // activated gates i f g o, c_k; c_{k-1}; d loss / d h_k from the decoder
};

template <int NSETS>
__device__ __forceinline__ void cell_chain_backward(const sur_chunk_params& p, float* lds, const sur_chunk_span& sp, int b, int B,
                                                    const float* __restrict__ c_all, const float* __restrict__ saved,
                                                    const float* __restrict__ dh_dec, const float* __restrict__ dh_all,
                                                    const float* __restrict__ dc_all, float* __restrict__ dg_all,
                                                    float* __restrict__ dh0, float* __restrict__ dc0) {
    const int hq = p.hq, s = 16 * hq;
    const int K0 = sp.k0, K = sp.k1, S = sp.k0 + sp.s;
    const float* __restrict__ c0 = sp.c0;
    const int hc_bstride = sp.hc_bstride;
    float* __restrict__ dlstates_t = sp.dlstates_t;
    lds_f* const dgates = (lds_f*)lds;    // [4][s]   LDS-typed: as generic pointers these were flat accesses, whose
    lds_f* const part = dgates + 4 * s;   // [4][s]   s_waitcnt vmcnt(0) drained the two-steps-ahead prefetch every step
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int gt = wid / NSETS, n0 = 16 * (wid % NSETS);
    const size_t save_stride = step_saved_floats(p);
    // A[m = ci][k = o][tap] = Wh_g[(o * 16 + ci) * 3 + tap]; lane (r, q): row ci = r, k = q of each block of four o
    float a[3][4];
    {
        const float* wh = p.w[SUR_ST_WHI + 3 * gt];
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) a[tap][blk] = wh[((4 * blk + q) * 16 + r) * 3 + tap];
    }
    int colj[3];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) colj[tap] = wrapi(n0 + r - tap + 1, hq);
    const int i = tid;   // this thread's element of [cs][hq]
    auto fetch = [&](int kk, CellBwdIn& in) {
        const float* rec = saved + ((size_t)kk * B + b) * save_stride + i;
#pragma unroll
        for (int j = 0; j < 5; ++j) in.g[j] = rec[(size_t)j * s];
        in.cp = kk > K0 ? c_all[((size_t)(kk - 1) * B + b) * s + i] : c0[(size_t)b * hc_bstride + i];
        in.dhd = dh_dec[((size_t)kk * B + b) * s + i];
    };
    CellBwdIn cur{}, nx1{}, nx2{};
    fetch(K - 1, cur);
    if (K - 2 >= K0) fetch(K - 2, nx1);
    float dcc = 0.0f;
    for (int k = K - 1; k >= K0 - 1; --k) {
        if (k - 2 >= K0) fetch(k - 2, nx2);
        // ---- A: what step k+1's GEMM produced, then (k >= K0) this step's gate derivative ----
        const bool have_prev = k + 1 < K, prev_forced = have_prev && k + 1 < S;
        float dh_in = 0.0f;
        if (have_prev) {
            const float v_ = (part[i] + part[s + i]) + (part[2 * s + i] + part[3 * s + i]);
            if (prev_forced) {
                if (dlstates_t) dlstates_t[((size_t)(k + 1 - K0) * B + b) * s + i] = v_;
            } else {
                dh_in = v_;
            }
        }
        if (k < K0) {   // past the first step: what is left goes to the initial state
            if (dh0) dh0[(size_t)b * s + i] = dh_in;
            if (dc0) dc0[(size_t)b * s + i] = dcc;
            break;
        }
        const size_t kb = (size_t)k * B + b;
        const float dhn = cur.dhd + dh_in + (dh_all ? dh_all[kb * s + i] : 0.0f);
        const float gi = cur.g[0], gf = cur.g[1], gg = cur.g[2], go = cur.g[3];
        const float tc = tanh_(cur.g[4]);
        const float dcn = dcc + (dc_all ? dc_all[kb * s + i] : 0.0f) + dhn * go * (1.0f - tc * tc);
        const float d0 = dcn * gg * gi * (1.0f - gi), d1 = dcn * cur.cp * gf * (1.0f - gf),
                    d2 = dcn * gi * (1.0f - gg * gg), d3 = dhn * tc * go * (1.0f - go);
        dgates[i] = d0;
        dgates[s + i] = d1;
        dgates[2 * s + i] = d2;
        dgates[3 * s + i] = d3;
        float* dg = dg_all + kb * 4 * s;
        dg[i] = d0;
        dg[s + i] = d1;
        dg[2 * s + i] = d2;
        dg[3 * s + i] = d3;
        dcc = dcn * gf;   // gradient wrt c_{k-1}
        __syncthreads();
        // ---- B: dh_{k-1} partial of gate gt, column tile n0: sum_{o, tap} Wh_gt[o][ci][tap] * dG_gt[o][j - tap + 1] ----
        {
            const lds_f* bsrc = dgates + gt * s;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const lds_f* bp = bsrc + colj[tap];
#pragma unroll
                for (int blk = 0; blk < 4; blk += 2) {
                    const float b0 = bp[(4 * blk + q) * hq], b1 = bp[(4 * blk + 4 + q) * hq];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tap][blk], b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tap][blk + 1], b1, acc1, 0, 0, 0);
                }
            }
            lds_f* dst = part + gt * s + n0 + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(4 * q + e) * hq] = acc0[e] + acc1[e];
        }
        cur = nx1;
        nx1 = nx2;
        __syncthreads();
    }
}

template <int MAXT>
__global__ void __launch_bounds__(MAXT)
cell_bwd_kernel(const sur_chunk_params p, const ChunkSpans spans, const float* __restrict__ c_all,
                const float* __restrict__ saved, const float* __restrict__ dh_dec, const float* __restrict__ dh_all,
                const float* __restrict__ dc_all, int B, float* __restrict__ dg_all, float* __restrict__ dh0,
                float* __restrict__ dc0) {
    extern __shared__ __align__(16) float lds[];
    // workgroup -> (chunk of the time axis, sample): the chunks' chains are independent (TBPTT cuts the graph)
    const sur_chunk_span sp = spans.sp[blockIdx.x / B];
    const int b = blockIdx.x % B, s = p.cs * p.hq;
#ifndef SUR_STAMP
    if constexpr ((MAXT & 255) == 0) {
        if (p.cs == 16 && p.hq * 16 == MAXT) {   // one (channel, position) element per thread
            cell_chain_backward<MAXT / 256>(p, lds, sp, b, B, c_all, saved, dh_dec, dh_all, dc_all, dg_all, dh0, dc0);
            return;
        }
    }
#endif
    const int K0 = sp.k0, K = sp.k1, S = sp.k0 + sp.s;   // steps K0 .. K-1 (global time index), the first sp.s teacher forced
    const float* __restrict__ c0 = sp.c0;
    const int hc_bstride = sp.hc_bstride;
    float* __restrict__ dlstates_t = sp.dlstates_t;
    StepLayout L{};
    float* blk[2] = {lds, lds + 5 * s};   // [gates | c_k] of the step at hand / of the next one (ping-pong)
    L.dgates = lds + 10 * s;
    float* part = L.dgates + 4 * s;        // four partial dh tiles
    float* wbase = part + 4 * s;
    const size_t save_stride = step_saved_floats(p);
    const int npieces = (5 * s) / DMA_PIECE;   // a whole number: checked on the host
    auto fetch_block = [&](int kk) {
        const float* src = saved + ((size_t)kk * B + b) * save_stride;
        float* dst = blk[kk & 1];
        const int lane = threadIdx.x & 63;
        for (int j = threadIdx.x >> 6; j < npieces; j += blockDim.x >> 6)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * DMA_PIECE + lane * 4),
                                             (__attribute__((address_space(3))) void*)(dst + j * DMA_PIECE + lane * 4), 16, 0, 0);
    };
    float cp[CELL_EPT], dhd[CELL_EPT], dcc[CELL_EPT];
    auto prefetch = [&](int kk) {   // c_{kk-1} and the decoder's dh of step kk
#pragma unroll
        for (int e = 0; e < CELL_EPT; ++e) {
            const int i = threadIdx.x + e * MAXT;
            if (i < s) {
                cp[e] = (kk > K0) ? c_all[((size_t)(kk - 1) * B + b) * s + i] : c0[(size_t)b * hc_bstride + i];
                dhd[e] = dh_dec[((size_t)kk * B + b) * s + i];
            }
        }
    };
    fetch_block(K - 1);
    prefetch(K - 1);
    const float* w[SUR_ST_NPARAM];
    stage_range<ST_NLSTM>(p, 0, wbase, w);
#pragma unroll
    for (int e = 0; e < CELL_EPT; ++e) dcc[e] = 0.0f;
    __syncthreads();

    for (int k = K - 1; k >= K0 - 1; --k) {
        // ---- A: what step k+1's GEMM produced, then (k >= K0) this step's gate derivative ----
        const bool have_prev = k + 1 < K;               // part[] holds d loss / d h_in of step k+1
        const bool prev_forced = have_prev && k + 1 < S; // ... which was the encoded given state, not h_k
        const float* g_ = k >= K0 ? blk[k & 1] : nullptr;
#pragma unroll
        for (int e = 0; e < CELL_EPT; ++e) {
            const int i = threadIdx.x + e * MAXT;
            if (i >= s) continue;
            float dh_in = 0.0f;
            if (have_prev) {
                const float v_ = (part[i] + part[s + i]) + (part[2 * s + i] + part[3 * s + i]);
                if (prev_forced) {
                    if (dlstates_t) dlstates_t[((size_t)(k + 1 - K0) * B + b) * s + i] = v_;
                } else {
                    dh_in = v_;
                }
            }
            if (k < K0) {   // past the first step: what is left goes to the initial state
                if (dh0) dh0[(size_t)b * s + i] = dh_in;  // non-zero only if step 0 was free running (S == 0)
                if (dc0) dc0[(size_t)b * s + i] = dcc[e];
                continue;
            }
            const size_t kb = (size_t)k * B + b;
            const float dhn = dhd[e] + dh_in + (dh_all ? dh_all[kb * s + i] : 0.0f);
            const float gi = g_[i], gf = g_[s + i], gg = g_[2 * s + i], go = g_[3 * s + i];
            const float tc = tanh_(g_[4 * s + i]);
            const float dcn = dcc[e] + (dc_all ? dc_all[kb * s + i] : 0.0f) + dhn * go * (1.0f - tc * tc);
            const float d0 = dcn * gg * gi * (1.0f - gi), d1 = dcn * cp[e] * gf * (1.0f - gf),
                        d2 = dcn * gi * (1.0f - gg * gg), d3 = dhn * tc * go * (1.0f - go);
            L.dgates[i] = d0;
            L.dgates[s + i] = d1;
            L.dgates[2 * s + i] = d2;
            L.dgates[3 * s + i] = d3;
            float* dg = dg_all + kb * 4 * s;
            dg[i] = d0;
            dg[s + i] = d1;
            dg[2 * s + i] = d2;
            dg[3 * s + i] = d3;
            dcc[e] = dcn * gf;  // gradient wrt c_{k-1}
        }
        if (k < K0) break;
        __syncthreads();
        STAMP(120);
        // ---- B: prefetches for step k-1, then dh_{k-1} partials ----
        if (k > K0) {
            fetch_block(k - 1);
            prefetch(k - 1);
        }
        STAMP(121);
        cell_dh_gemm(p, L, w, part);   // ends with the barrier that also retires the DMA
        STAMP(122);
    }
}

// Everything of the cell backward that is not recurrent, for all (step, sample) pairs in parallel: the gradient wrt
// the latent action and the LSTM weight / bias gradients (into this workgroup's partial row).
template <bool GL>      // as dec_bwd_kernel
__global__ void __launch_bounds__(TPB, PAR_OCC)
cell_wgrad_kernel(const sur_chunk_params p, const ChunkSpans spans, const float* __restrict__ xlat_t,
                  const float* __restrict__ h_all, const float* __restrict__ dg_all, int K, int B,
                  float* __restrict__ dxlat_t, int row_base) {
    extern __shared__ __align__(16) float lds[];
    const int s = p.cs * p.hq, nx = p.ca * p.hq, M = K * B, hq = p.hq, hs = hq + 4;
    StepLayout L{};
    L.dx = lds;
    L.x = L.dx + nx;
    L.h = L.x + p.ca * hs;
    L.dgates = L.h + p.cs * hs;
    float* wbase = L.dgates + 4 * p.cs * hs;
    // of the LSTM's weights only the four Wx_g are read here (dx); staged back to back.  (All twelve tensors were 15.6 KB: with
    // the padded rows one workgroup too many for three per CU -- measured: 0.571 instead of 0.541 ms per step.)
    const float* w[SUR_ST_NPARAM] = {};
    int wx_floats = 0;
    for (int gt = 0; gt < 4; ++gt) {
        const int i = SUR_ST_WXI + 3 * gt;     // parameter order: Wx_g, b_g, Wh_g per gate
        for (int j = threadIdx.x; j < p.size[i]; j += blockDim.x) wbase[wx_floats + j] = p.w[i][j];
        w[i] = wbase + wx_floats;
        wx_floats += p.size[i];
    }
    int psize_lstm = 0;
    for (int i = 0; i < ST_NLSTM; ++i) psize_lstm += p.size[i];
    int psize = psize_lstm;
    for (int i = ST_NLSTM; i < SUR_ST_NPARAM; ++i) psize += p.size[i];
    float* row = p.partial + (size_t)(row_base + blockIdx.x) * psize;
    typedef typename std::conditional<GL, lds_f*, float*>::type GP;
    GP gacc;
    if constexpr (GL) gacc = (lds_f*)(wbase + wx_floats);
    else gacc = row;
    GP g[SUR_ST_NPARAM];
    {
        int off = 0;
        for (int i = 0; i < ST_NLSTM; ++i) {
            g[i] = gacc + off;
            off += p.size[i];
        }
        if (GL)
            for (int j = threadIdx.x; j < psize_lstm; j += blockDim.x) gacc[j] = 0.0f;
    }
    __syncthreads();
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        const int k = m / B, b = m - k * B;
        int si = 0;
        for (int j = 1; j < spans.n; ++j) si = k >= spans.sp[j].k0 ? j : si;
        const sur_chunk_span sp = spans.sp[si];
        const float* hin = (k - sp.k0 < sp.s) ? sp.lstates_t + ((size_t)(k - sp.k0) * B + b) * s
                                               : (k > sp.k0 ? h_all + ((size_t)(k - 1) * B + b) * s : sp.h0 + (size_t)b * sp.hc_bstride);
        // rows land hs floats apart (float4 granules: hq is a multiple of 4)
        const int q4 = hq >> 2;
        auto padded = [&](int i4) { const int row = i4 / q4; return row * hs + 4 * (i4 - row * q4); };
        {
            const float4* xs = reinterpret_cast<const float4*>(xlat_t + (size_t)m * nx);
            const float4* hsrc = reinterpret_cast<const float4*>(hin);
            const float4* gs = reinterpret_cast<const float4*>(dg_all + (size_t)m * 4 * s);
            for (int i = threadIdx.x; i < (nx >> 2); i += blockDim.x) *reinterpret_cast<float4*>(L.x + padded(i)) = xs[i];
            for (int i = threadIdx.x; i < (s >> 2); i += blockDim.x) *reinterpret_cast<float4*>(L.h + padded(i)) = hsrc[i];
            constexpr int U = 4;    // all of a thread's loads of a round in flight before the first store (lds_load_v4)
            for (int i0 = threadIdx.x; i0 < s; i0 += U * TPB) {
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = gs[i0 + u * TPB < s ? i0 + u * TPB : i0];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (i0 + u * TPB < s) *reinterpret_cast<float4*>(L.dgates + padded(i0 + u * TPB)) = v[u];
            }
            __syncthreads();
        }
        cell_wgrad_gemms(p, L, w, g, hs);
        if (dxlat_t)
            for (int i = threadIdx.x; i < nx; i += blockDim.x) dxlat_t[(size_t)m * nx + i] = L.dx[i];
        __syncthreads();
    }
    if constexpr (GL) add_to_row(row, (const float*)gacc, psize_lstm);
}

// g[i][j] += sum_r partial[r][off_i + j]; the partial rows are re-zeroed.  With an Adam descriptor the reduced gradient is
// consumed on the spot: g = sum (not accumulated), then the torch.optim.Adam update of the parameter element
// (no weight decay, no amsgrad; same formula as torch's fused kernel) -- the optimizer costs no extra launch and the
// gradients need no zeroing pass.
#ifndef FLUSH_TPB
#define FLUSH_TPB 1024   // the reduction over rows is a latency chain (rows / FLUSH_RG / 8 rounds of loads): 1024 threads = 32 row groups
#endif
constexpr int FLUSH_COLS = 32, FLUSH_RG = FLUSH_TPB / FLUSH_COLS;   // 16 columns (twice the blocks) measured slower: 34 vs 26 us
__host__ __device__ inline int flush_blocks(int psize) { return (psize + FLUSH_COLS - 1) / FLUSH_COLS; }

template <int NP, typename Params>
__device__ __forceinline__ void flush_grads_body(const Params& p, int psize, const sur_adam& adam, bool overwrite, int blk,
                                                 int nblk) {
    // block = FLUSH_COLS columns x FLUSH_RG row groups: each thread sums every FLUSH_RG-th row of its column, LDS combines
    // the partials (a wave reads 2 rows x 128 contiguous bytes per load instruction).  The kernel is a chain of memory
    // round trips (rows -> LDS -> Adam state -> ticket), so whatever does not depend on the sum is fetched FIRST: the row
    // group 0 threads -- the ones that finish a column -- locate their parameter and load its moments / weight before
    // the row loop.
    __shared__ float part[FLUSH_RG][FLUSH_COLS + 1];
    const int col = threadIdx.x & (FLUSH_COLS - 1), rg = threadIdx.x / FLUSH_COLS;
    const int t = blk * FLUSH_COLS + col;
    const int step = adam.m ? *adam.step + 1 : 0;   // read before this block takes its ticket (see below)
    const float lr = adam.m ? *adam.lr : 0.0f;      // device scalar: a scheduler can change it between graph replays
    const bool finisher = rg == 0 && t < psize;
    float* gdst = nullptr;
    float* wdst = nullptr;
    float m_old = 0.0f, v_old = 0.0f, w_old = 0.0f, bc1 = 1.0f, bc2 = 1.0f;
    if (finisher) {
        int off = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (t >= off && t < off + p.size[i]) {
                gdst = p.g[i] + (t - off);
                wdst = const_cast<float*>(p.w[i]) + (t - off);
            }
            off += p.size[i];
        }
        if (adam.m) {
            m_old = adam.m[t];
            v_old = adam.v[t];
            w_old = *wdst;
            bc1 = 1.0f - powf(adam.beta1, (float)step);
            bc2 = 1.0f - powf(adam.beta2, (float)step);
        }
    }
    float acc = 0.0f;
    if (t < psize) {
        constexpr int U = 8;   // loads of a round all in flight before the first add / re-zero (same summation order)
        for (int r0 = rg; r0 < p.rows; r0 += FLUSH_RG * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + FLUSH_RG * u;
                v[u] = r < p.rows ? p.partial[(size_t)r * psize + t] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + FLUSH_RG * u;
                if (r < p.rows) {
                    acc += v[u];
                    p.partial[(size_t)r * psize + t] = 0.0f;
                }
            }
        }
    }
    part[rg][col] = acc;
    __syncthreads();
    if (finisher) {
        float tot = 0.0f;
#pragma unroll
        for (int i = 0; i < FLUSH_RG; ++i) tot += part[i][col];
        if (adam.m) {
            *gdst = tot;
            const float m = adam.beta1 * m_old + (1.0f - adam.beta1) * tot;
            const float v = adam.beta2 * v_old + (1.0f - adam.beta2) * tot * tot;
            adam.m[t] = m;
            adam.v[t] = v;
            const float denom = sqrtf(v) / sqrtf(bc2) + adam.eps;
            *wdst = w_old - (lr / bc1) * (m / denom);
        } else if (overwrite) {
            *gdst = tot;
        } else {
            *gdst += tot;
        }
    }
    if (adam.m) {   // the workgroup that takes the last ticket has, like every other, already read the step count
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(adam.ticket, 1u) == (unsigned)(nblk - 1)) {
            *adam.step = step;
            *adam.ticket = 0u;
        }
    }
}

template <int NP, typename Params>
__global__ void __launch_bounds__(FLUSH_TPB) flush_grads_kernel(const Params p, int psize, const sur_adam adam, int overwrite) {
    flush_grads_body<NP, Params>(p, psize, adam, overwrite != 0, blockIdx.x, gridDim.x);
}

// the three gradient reductions of a surrogate (state encoder, action encoder, chunk) in one launch
__global__ void __launch_bounds__(FLUSH_TPB)
flush_all_kernel(const sur_encoder_params e0, const sur_adam a0, int n0, const sur_encoder_params e1, const sur_adam a1, int n1,
                 const sur_chunk_params c2, const sur_adam a2, int n2, int overwrite_mask) {
    const int b0 = flush_blocks(n0), b1 = flush_blocks(n1), b2 = flush_blocks(n2);
    const int blk = blockIdx.x;
    if (blk < b0) flush_grads_body<SUR_ENC_NPARAM, sur_encoder_params>(e0, n0, a0, overwrite_mask & 1, blk, b0);
    else if (blk < b0 + b1) flush_grads_body<SUR_ENC_NPARAM, sur_encoder_params>(e1, n1, a1, overwrite_mask & 2, blk - b0, b1);
    else flush_grads_body<SUR_ST_NPARAM, sur_chunk_params>(c2, n2, a2, overwrite_mask & 4, blk - b0 - b1, b2);
}

// Fold the partial-gradient rows [base, base + count) of up to three packs into ONE row each (dst += their sum, fixed order)
// and re-zero them: what a backward branch that finishes early does with its own rows, so that the reduction at the end
// of the step (flush) reads one row per branch instead of hundreds (hipops.fused_tbptt_train).
struct FoldJob {
    float* partial;
    int psize, base, count, dst;
};

__device__ __forceinline__ void fold_rows_body(const FoldJob& j, int blk) {
    __shared__ float part[FLUSH_RG][FLUSH_COLS + 1];
    const int col = threadIdx.x & (FLUSH_COLS - 1), rg = threadIdx.x / FLUSH_COLS;
    const int t = blk * FLUSH_COLS + col;
    float acc = 0.0f;
    if (t < j.psize) {
        constexpr int U = 8;
        float* src = j.partial + (size_t)j.base * j.psize + t;
        for (int r0 = rg; r0 < j.count; r0 += FLUSH_RG * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + FLUSH_RG * u;
                v[u] = r < j.count ? src[(size_t)r * j.psize] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + FLUSH_RG * u;
                if (r < j.count) {
                    acc += v[u];
                    src[(size_t)r * j.psize] = 0.0f;
                }
            }
        }
    }
    part[rg][col] = acc;
    __syncthreads();
    if (rg == 0 && t < j.psize) {
        float tot = 0.0f;
#pragma unroll
        for (int i = 0; i < FLUSH_RG; ++i) tot += part[i][col];
        j.partial[(size_t)j.dst * j.psize + t] += tot;
    }
}

__global__ void __launch_bounds__(FLUSH_TPB) fold_rows_kernel(const FoldJob j0, const FoldJob j1, const FoldJob j2) {
    const int b0 = flush_blocks(j0.psize), b1 = flush_blocks(j1.psize);
    const int blk = blockIdx.x;
    if (blk < b0) fold_rows_body(j0, blk);
    else if (blk < b0 + b1) fold_rows_body(j1, blk - b0);
    else fold_rows_body(j2, blk - b0 - b1);
}

// torch.optim.Adam (no weight decay / amsgrad) of a whole parameter pack from its gradient tensors: the optimizer of the
// EAGER fused path (pdecontrol.surrogates.hipops.PackAdam) -- one launch instead of torch's per-step Python bookkeeping.
// Same arithmetic as the Adam branch of flush_grads_body.
template <int NP, typename Params>
__device__ __forceinline__ void adam_apply_body(const Params& p, int psize, const sur_adam& adam, int blk, int nblk) {
    const int t = blk * TPB + threadIdx.x;
    const int step = *adam.step + 1;   // read before this block takes its ticket
    const float lr = *adam.lr;
    if (t < psize) {
        int off = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (t >= off && t < off + p.size[i]) {
                const float g = p.g[i][t - off];
                const float m = adam.beta1 * adam.m[t] + (1.0f - adam.beta1) * g;
                const float v = adam.beta2 * adam.v[t] + (1.0f - adam.beta2) * g * g;
                adam.m[t] = m;
                adam.v[t] = v;
                const float bc1 = 1.0f - powf(adam.beta1, (float)step), bc2 = 1.0f - powf(adam.beta2, (float)step);
                const float denom = sqrtf(v) / sqrtf(bc2) + adam.eps;
                float* w = const_cast<float*>(p.w[i]);
                w[t - off] -= (lr / bc1) * (m / denom);
            }
            off += p.size[i];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(adam.ticket, 1u) == (unsigned)(nblk - 1)) {
        *adam.step = step;
        *adam.ticket = 0u;
    }
}

__global__ void __launch_bounds__(TPB)
adam_all_kernel(const sur_encoder_params e0, const sur_adam a0, int n0, const sur_encoder_params e1, const sur_adam a1, int n1,
                const sur_chunk_params c2, const sur_adam a2, int n2) {
    const int b0 = (n0 + TPB - 1) / TPB, b1 = (n1 + TPB - 1) / TPB, b2 = (n2 + TPB - 1) / TPB;
    const int blk = blockIdx.x;
    if (blk < b0) adam_apply_body<SUR_ENC_NPARAM, sur_encoder_params>(e0, n0, a0, blk, b0);
    else if (blk < b0 + b1) adam_apply_body<SUR_ENC_NPARAM, sur_encoder_params>(e1, n1, a1, blk - b0, b1);
    else adam_apply_body<SUR_ST_NPARAM, sur_chunk_params>(c2, n2, a2, blk - b0 - b1, b2);
}

template <typename F>
int launch_checked(F&& f, const char* what) {
    f();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "%s launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

constexpr size_t LDS_LIMIT = 160 * 1024;
// Gradient accumulators live in LDS when they fit next to the activations; otherwise the kernels accumulate straight into the
// workgroup's partial row (the GL = false instantiations).  No supported geometry needs that today, so the tests reach those
// instantiations through this switch (SUR_ACCUMULATE_IN_ROWS=1, read at every launch).
static bool accumulators_fit(size_t bytes_with_accumulators) {
    const char* force = getenv("SUR_ACCUMULATE_IN_ROWS");
    return bytes_with_accumulators <= LDS_LIMIT && !(force && force[0] == '1');
}

template <typename K>
int set_lds(K kernel, size_t bytes, const char* what) {
    if (bytes > LDS_LIMIT) return fail(-4, "%s needs %zu B of LDS (> 160 KiB): N too large for the fused path", what, bytes);
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return 0;
}

}  // namespace

extern "C" {

const char* sur_last_error(void) { return g_err; }

#ifdef SUR_STAMP
int sur_debug_stamps(long long* out128, int reset) {
    if (out128 && hipMemcpyFromSymbol(out128, HIP_SYMBOL(sur_stamp_buf), sizeof(long long) * 128) != hipSuccess) return -2;
    if (reset) {
        long long z[128] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sur_stamp_buf), z, sizeof(z)) != hipSuccess) return -2;
    }
    return 0;
}
#endif

// The fused LayerNorm reduces a row with groups of 16, 32 or 64 lanes holding H / lanes elements each (act_ln_fwd): H is 16, 32
// or a multiple of 64 up to 64 * LN_MAX_EPL.  Any other width would be normalised over a part of the row -- silently.
static bool ln_width_ok(int h) { return h == 16 || h == 32 || (h >= 64 && h <= 64 * LN_MAX_EPL && (h & 63) == 0); }

static int enc_geometry(const sur_encoder_params* p, const char* who) {
    int h = p->n;
    for (int b = 0; b < 3; ++b) {
        if (p->stride[b] < 1 || h % p->stride[b]) return fail(-4, "%s: width %d is not a multiple of stride %d (block %d)", who, h, p->stride[b], b);
        h /= p->stride[b];
        if (!ln_width_ok(h))
            return fail(-4, "%s: N = %d gives a LayerNorm row of %d in block %d; the fused kernels take 16, 32, 64, 128, 192, 256", who,
                        p->n, h, b);
    }
    return 0;
}

static int chunk_geometry(const sur_chunk_params* p, const char* who) {
    if (!ln_width_ok(2 * p->hq) || !ln_width_ok(4 * p->hq))
        return fail(-4, "%s: N = %d gives LayerNorm rows of %d and %d in the decoder; the fused kernels take 16, 32, 64, 128, 192, 256", who,
                    4 * p->hq, 2 * p->hq, 4 * p->hq);
    return 0;
}

int sur_geometry_supported(const sur_encoder_params* state_enc, const sur_encoder_params* action_enc, const sur_chunk_params* chunk) {
    if (state_enc) if (int rc = enc_geometry(state_enc, "state encoder")) return rc;
    if (action_enc) if (int rc = enc_geometry(action_enc, "action encoder")) return rc;
    if (chunk) if (int rc = chunk_geometry(chunk, "cell / decoder")) return rc;
    return 0;
}

int sur_encoder_saved_floats(const sur_encoder_params* p) {
    if (!p) return 0;
    const int total = enc_saved_floats(*p);
    return ((total & 3) || ((p->c[0] * p->n) & 3)) ? 0 : total;  // float4 granularity of the block and of its LDS position
}

int sur_encoder_forward(void* stream, const sur_encoder_params* p, const float* x, int m, float* z, float* saved) {
    if (!p || !x || !z || m <= 0) return fail(-1, "sur_encoder_forward: bad argument");
    if (saved && sur_encoder_saved_floats(p) == 0)
        return fail(-4, "sur_encoder_forward: this geometry has no saved-activation path (pass saved = NULL)");
    if (int rc = enc_geometry(p, "sur_encoder_forward")) return rc;
    const int psize = psize_of<SUR_ENC_NPARAM>(p->size);
    const size_t lds = sizeof(float) * (enc_act_floats(*p, false) + psize);
    if (int rc = set_lds(enc_fwd_kernel, lds, "encoder forward")) return rc;
    const int grid = m < 1024 ? m : 1024;
    return launch_checked([&] { hipLaunchKernelGGL(enc_fwd_kernel, dim3(grid), dim3(TPB), lds, (hipStream_t)stream, *p, x, m, z, saved); },
                          "enc_fwd");
}

int sur_encoder_backward(void* stream, const sur_encoder_params* p, const float* x, const float* dz, int m, float* dx,
                         int row_base, int row_count, const float* saved) {
    if (!p || !x || !dz || m <= 0) return fail(-1, "sur_encoder_backward: bad argument");
    if (int rc = enc_geometry(p, "sur_encoder_backward")) return rc;
    if (saved && sur_encoder_saved_floats(p) == 0)
        return fail(-4, "sur_encoder_backward: this geometry has no saved-activation path (pass saved = NULL)");
    if (!p->partial || row_count <= 0 || row_base < 0 || row_base + row_count > p->rows)
        return fail(-1, "sur_encoder_backward: partial rows [%d, %d) outside the buffer of %d rows", row_base,
                    row_base + row_count, p ? p->rows : 0);
    const int psize = psize_of<SUR_ENC_NPARAM>(p->size);
    const size_t base = sizeof(float) * (enc_act_floats(*p, true) + psize);
    int grads_in_lds = accumulators_fit(base + sizeof(float) * psize) ? 1 : 0;
    const size_t lds = base + (grads_in_lds ? sizeof(float) * psize : 0);
    if (int rc = set_lds(enc_bwd_kernel, lds, "encoder backward")) return rc;
    const int grid = m < row_count ? m : row_count;
    return launch_checked([&] {
        hipLaunchKernelGGL(enc_bwd_kernel, dim3(grid), dim3(TPB), lds, (hipStream_t)stream, *p, x, dz, m, dx, grads_in_lds,
                           row_base, saved);
    }, "enc_bwd");
}

int sur_encoder_forward_multi(void* stream, int njobs, const sur_encoder_params* const* ps, const float* const* xs, const int* ms,
                              float* const* zs, float* const* saveds, int max_workgroups) {
    if (njobs < 1 || njobs > 2 || !ps || !xs || !ms || !zs || !saveds || max_workgroups <= 0)
        return fail(-1, "sur_encoder_forward_multi: bad argument (1 or 2 jobs)");
    for (int blk = 0; blk < 3; ++blk) {
        // the jobs of this block, wide (MFMA kernels) and narrow (one wave per sample) ones in a launch each
        EncFwdJob wide[2] = {};
        NarrowJob narrow[2] = {};
        size_t lds_w = 0;
        int grid_w = 0, nw = 0, nn = 0;
        for (int j = 0; j < njobs; ++j) {
            const sur_encoder_params* p = ps[j];
            if (!p || !xs[j] || !zs[j] || !saveds[j] || ms[j] <= 0)
                return fail(-1, "sur_encoder_forward_multi: job %d: bad argument (the `saved` buffer is required)", j);
            if (sur_encoder_saved_floats(p) == 0) return fail(-4, "sur_encoder_forward_multi: job %d: geometry not float4-granular", j);
            if (int rc = enc_geometry(p, "sur_encoder_forward_multi")) return rc;
            const EncBlockGeom gm = enc_block_geom(*p, blk);
            if (enc_narrow(*p)) {      // one wave per sample, a launch of its own
                const int per_wg = TPB / 64, passes = (ms[j] + per_wg - 1) / per_wg;
                NarrowJob nj = narrow_job(*p, blk, ms[j]);
                nj.x = xs[j];
                nj.z = blk == 2 ? zs[j] : nullptr;
                nj.saved = saveds[j];
                nj.wg_count = passes < max_workgroups ? passes : max_workgroups;
                narrow[nn++] = nj;
            } else {
                const size_t need = sizeof(float) * (gm.cin * gm.hin + 7 * gm.cout * gm.hout + gm.psize_blk);
                lds_w = need > lds_w ? need : lds_w;
                const int wgs = ms[j] < max_workgroups ? ms[j] : max_workgroups;
                wide[nw++] = EncFwdJob{*p, xs[j], zs[j], saveds[j], ms[j], grid_w, wgs};
                grid_w += wgs;
            }
        }
        // ONE launch: the wide jobs' workgroups first, the (first) narrow job's behind them; further narrow jobs on their own
        auto narrow_lds = [](const NarrowJob& nj) {
            return sizeof(float) * ((size_t)(TPB / 64) * nr_fwd_wave_floats(nj.cin, nj.hin, nj.cout, nj.hout) + nj.psize_blk);
        };
        if (nw) {
            const bool ride = nn > 0;
            const size_t lds = ride && narrow_lds(narrow[0]) > lds_w ? narrow_lds(narrow[0]) : lds_w;
            const int grid = grid_w + (ride ? narrow[0].wg_count : 0);
            if (int rc = set_lds(enc_block_fwd_multi_kernel, lds, "encoder block forward")) return rc;
            if (int rc = launch_checked([&] {
                    hipLaunchKernelGGL(enc_block_fwd_multi_kernel, dim3(grid), dim3(TPB), lds, (hipStream_t)stream, wide[0], wide[1], nw,
                                       blk, narrow[0], ride ? grid_w : grid);
                }, "enc_block_fwd")) return rc;
        }
        for (int k = nw ? 1 : 0; k < nn; ++k) {
            const NarrowJob& nj = narrow[k];
            const size_t lds_n = narrow_lds(nj);
            if (int rc = set_lds(enc_narrow_fwd_kernel, lds_n, "narrow encoder block forward")) return rc;
            if (int rc = launch_checked([&] {
                    hipLaunchKernelGGL(enc_narrow_fwd_kernel, dim3(nj.wg_count), dim3(TPB), lds_n, (hipStream_t)stream, nj);
                }, "enc_narrow_fwd")) return rc;
        }
    }
    return 0;
}

int sur_encoder_workspace_floats(const sur_encoder_params* p, int m) {
    if (!p || m <= 0) return 0;
    return m * enc_ws_floats(*p);
}

int sur_encoder_backward_multi(void* stream, int njobs, const sur_encoder_params* const* ps, const float* const* xs,
                               const float* const* dzs, const int* ms, const int* row_bases, const int* row_counts,
                               const float* const* saveds, float* const* workspaces) {
    if (njobs < 1 || njobs > 3 || !ps || !xs || !dzs || !ms || !row_bases || !row_counts || !saveds)
        return fail(-1, "sur_encoder_backward_multi: bad argument (1 to 3 jobs)");
    for (int j = 0; j < njobs; ++j) {
        if (!ps[j]) return fail(-1, "sur_encoder_backward_multi: job %d: no parameters", j);
        if (int rc = enc_geometry(ps[j], "sur_encoder_backward_multi")) return rc;
    }
    bool split = workspaces != nullptr;
    for (int j = 0; j < njobs && split; ++j) split = saveds[j] && workspaces[j] && ps[j] && sur_encoder_saved_floats(ps[j]) > 0;
    if (split) {
        // one residual block per launch, last block first; every launch carries all jobs
        for (int blk = 2; blk >= 0; --blk) {
            EncBlockJob wide[3] = {};
            NarrowJob narrow[3] = {};
            size_t lds_w = 0;
            int grid_w = 0, nw = 0, nn = 0;
            for (int j = 0; j < njobs; ++j) {
                const sur_encoder_params* p = ps[j];
                if (!xs[j] || !dzs[j] || ms[j] <= 0) return fail(-1, "sur_encoder_backward_multi: job %d: bad argument", j);
                if (!p->partial || row_counts[j] <= 0 || row_bases[j] < 0 || row_bases[j] + row_counts[j] > p->rows)
                    return fail(-1, "sur_encoder_backward_multi: job %d: partial rows [%d, %d) outside the buffer of %d rows", j,
                                row_bases[j], row_bases[j] + row_counts[j], p->rows);
                const EncBlockGeom gm = enc_block_geom(*p, blk);
                if ((gm.cout * gm.hout) & 3 || ((gm.cin * gm.hin) & 3 && blk > 0))
                    return fail(-4, "sur_encoder_backward_multi: job %d: block %d is not float4-granular", j, blk);
                if (enc_narrow(*p)) {       // one wave per sample, gradient accumulators per wave in LDS, a launch of its own
                    const int per_wg = TPB / 64, passes = (ms[j] + per_wg - 1) / per_wg;
                    NarrowJob nj = narrow_job(*p, blk, ms[j]);
                    nj.x = xs[j];
                    nj.dz = dzs[j];
                    nj.saved = const_cast<float*>(saveds[j]);
                    nj.ws = workspaces[j];
                    nj.wg_count = passes < row_counts[j] ? passes : row_counts[j];
                    nj.row_stride = psize_of<SUR_ENC_NPARAM>(p->size);
                    nj.rows = p->partial + (size_t)row_bases[j] * nj.row_stride + gm.param_off;
                    narrow[nn++] = nj;
                } else {
                    const size_t base = sizeof(float) * (enc_block_act_floats(gm) + gm.psize_blk);
                    const int gl = accumulators_fit(base + sizeof(float) * gm.psize_blk) ? 1 : 0;
                    const size_t need = base + (gl ? sizeof(float) * gm.psize_blk : 0);
                    lds_w = need > lds_w ? need : lds_w;
                    const int wgs = ms[j] < row_counts[j] ? ms[j] : row_counts[j];
                    wide[nw++] = EncBlockJob{*p, xs[j], dzs[j], saveds[j], workspaces[j], ms[j], row_bases[j], grid_w, wgs, gl};
                    grid_w += wgs;
                }
            }
            auto narrow_lds = [](const NarrowJob& nj) {
                return sizeof(float) * ((size_t)(TPB / 64) * nr_bwd_wave_floats(nj.cin, nj.hin, nj.cout, nj.hout, nj.psize_blk) + nj.psize_blk);
            };
            if (nw) {      // ONE launch: the wide jobs' workgroups first, the (first) narrow job's behind them
                const bool ride = nn > 0;
                const size_t lds = ride && narrow_lds(narrow[0]) > lds_w ? narrow_lds(narrow[0]) : lds_w;
                const int grid = grid_w + (ride ? narrow[0].wg_count : 0);
                bool all_gl = true;       // LDS accumulators for every wide job, or for none (one instantiation serves the launch)
                for (int k = 0; k < nw; ++k) all_gl = all_gl && wide[k].grads_in_lds;
                if (!all_gl)
                    for (int k = 0; k < nw; ++k) wide[k].grads_in_lds = 0;
                auto kernel = all_gl ? enc_block_bwd_multi_kernel<true> : enc_block_bwd_multi_kernel<false>;
                if (int rc = set_lds(kernel, lds, "encoder block backward")) return rc;
                if (int rc = launch_checked([&] {
                        hipLaunchKernelGGL(kernel, dim3(grid), dim3(TPB), lds, (hipStream_t)stream, wide[0], wide[1], wide[2], nw, blk, narrow[0],
                                           ride ? grid_w : grid);
                    }, "enc_block_bwd")) return rc;
            }
            for (int k = nw ? 1 : 0; k < nn; ++k) {
                const NarrowJob& nj = narrow[k];
                const size_t lds_n = narrow_lds(nj);
                if (int rc = set_lds(enc_narrow_bwd_kernel, lds_n, "narrow encoder block backward")) return rc;
                if (int rc = launch_checked([&] {
                        hipLaunchKernelGGL(enc_narrow_bwd_kernel, dim3(nj.wg_count), dim3(TPB), lds_n, (hipStream_t)stream, nj);
                    }, "enc_narrow_bwd")) return rc;
            }
        }
        return 0;
    }
    EncBwdJob jobs[3] = {};
    size_t lds = 0;
    int grid = 0;
    for (int j = 0; j < njobs; ++j) {
        const sur_encoder_params* p = ps[j];
        if (!p || !xs[j] || !dzs[j] || ms[j] <= 0) return fail(-1, "sur_encoder_backward_multi: job %d: bad argument", j);
        if (!p->partial || row_counts[j] <= 0 || row_bases[j] < 0 || row_bases[j] + row_counts[j] > p->rows)
            return fail(-1, "sur_encoder_backward_multi: job %d: partial rows [%d, %d) outside the buffer of %d rows", j, row_bases[j],
                        row_bases[j] + row_counts[j], p->rows);
        if (saveds[j] && sur_encoder_saved_floats(p) == 0)
            return fail(-4, "sur_encoder_backward_multi: job %d: this geometry has no saved-activation path", j);
        const int psize = psize_of<SUR_ENC_NPARAM>(p->size);
        const size_t base = sizeof(float) * (enc_act_floats(*p, true) + psize);
        const int gl = accumulators_fit(base + sizeof(float) * psize) ? 1 : 0;
        const size_t need = base + (gl ? sizeof(float) * psize : 0);
        lds = need > lds ? need : lds;
        const int wgs = ms[j] < row_counts[j] ? ms[j] : row_counts[j];
        jobs[j] = EncBwdJob{*p, xs[j], dzs[j], saveds[j], ms[j], row_bases[j], grid, wgs, gl};
        grid += wgs;
    }
    if (int rc = set_lds(enc_bwd_multi_kernel, lds, "encoder backward (multi)")) return rc;
    return launch_checked([&] {
        hipLaunchKernelGGL(enc_bwd_multi_kernel, dim3(grid), dim3(TPB), lds, (hipStream_t)stream, jobs[0], jobs[1], jobs[2], njobs);
    }, "enc_bwd_multi");
}

int sur_flush_encoder_grads(void* stream, const sur_encoder_params* p, const sur_adam* adam, int overwrite) {
    if (!p || !p->partial) return fail(-1, "sur_flush_encoder_grads: bad argument");
    for (int i = 0; i < SUR_ENC_NPARAM; ++i)
        if (!p->g[i]) return fail(-1, "sur_flush_encoder_grads: gradient tensor %d is NULL", i);
    const int psize = psize_of<SUR_ENC_NPARAM>(p->size);
    sur_adam ad{};
    if (adam) {
        if (!adam->m || !adam->v || !adam->step || !adam->ticket || !adam->lr)
            return fail(-1, "flush: incomplete Adam descriptor");
        ad = *adam;
    }
    return launch_checked([&] {
        hipLaunchKernelGGL((flush_grads_kernel<SUR_ENC_NPARAM, sur_encoder_params>), dim3(flush_blocks(psize)), dim3(FLUSH_TPB), 0,
                           (hipStream_t)stream, *p, psize, ad, overwrite);
    }, "flush_enc");
}

int sur_chunk_saved_floats(const sur_chunk_params* p) {
    if (!p) return 0;
    // the GEMM tiles want whole 16-wide latent rows; the cell backward moves [gates | c] in whole 1 KiB DMA pieces
    if ((p->hq & 15) || ((p->ca * p->hq) & 3) || (5 * p->cs * p->hq) % DMA_PIECE || p->cs * p->hq > CELL_EPT * TPB) return 0;
    // the decoder backward (dec_bwd_kernel) keeps every activation in buffers of step_max_act floats, p2 / a2 in the tail of
    // the gradient ping-pong buffer behind the first n floats its first phase uses, and two late activations in
    // DEC_LATE_P0 / DEC_LATE_H float4 registers per thread
    if (p->cs * p->hq > step_max_act(*p) || 12 * p->hq > step_max_act(*p) || p->cs * 2 * p->hq > 4 * DEC_LATE_P0 * TPB ||
        p->cs * p->hq > 4 * DEC_LATE_H * TPB)
        return 0;
    return step_saved_floats(*p);
}

int sur_chunk_forward(void* stream, const sur_chunk_params* p, const float* xlat_t, const float* lstates_t,
                      const float* states_t, const float* h0, const float* c0, int hc_bstride, int k, int s, int b,
                      float* h_all, float* c_all, float* d_all, float* out_all, float* saved) {
    if (!p || !xlat_t || !h0 || !c0 || !h_all || !c_all || !d_all || k <= 0 || b <= 0 || s < 1 ||
        !lstates_t || !states_t || hc_bstride < 0)
        return fail(-1, "sur_chunk_forward: bad argument (need K > 0, B > 0, S >= 1)");
    if (int rc = chunk_geometry(p, "sur_chunk_forward")) return rc;
    if (p->hq & 15) return fail(-4, "sur_chunk_forward: latent width N/4 = %d must be a multiple of 16", p->hq);
    if (saved && sur_chunk_saved_floats(p) == 0)
        return fail(-4, "sur_chunk_forward: hq = %d, ca = %d, cs = %d: no `saved` buffer for this geometry", p->hq, p->ca, p->cs);
    // split path: the cell chain (one workgroup per sample), then the decoders of all (step, sample) pairs in
    // parallel, then the integration of the predicted deltas
    int psize_lstm = 0, psize_dec = 0;
    for (int i = 0; i < ST_NLSTM; ++i) psize_lstm += p->size[i];
    for (int i = ST_NLSTM; i < SUR_ST_NPARAM; ++i) psize_dec += p->size[i];
    const size_t lds_cell = sizeof(float) * (cell_fwd_act_floats(*p) + psize_lstm);
    const size_t lds_dec = sizeof(float) * (dec_act_floats(*p, false) + psize_dec);
    if (int rc = set_lds(dec_fwd_kernel, lds_dec, "decoder forward")) return rc;
    const int m = k * b;
    const int chain_threads = cell_chain_threads(*p);
    auto launch_cell_fwd = [&](auto kernel) -> int {
        if (int rc = set_lds(kernel, lds_cell, "cell forward")) return rc;
        return launch_checked([&] {
            hipLaunchKernelGGL(kernel, dim3(b), dim3(chain_threads), lds_cell, (hipStream_t)stream, *p, xlat_t, lstates_t, h0, c0,
                               hc_bstride, k, s, b, h_all, c_all, saved);
        }, "cell_fwd");
    };
    if (int rc = chain_threads == TPB ? launch_cell_fwd(cell_fwd_kernel<TPB>)
                                      : (chain_threads == 2 * TPB ? launch_cell_fwd(cell_fwd_kernel<2 * TPB>)
                                                                  : launch_cell_fwd(cell_fwd_kernel<4 * TPB>)))
        return rc;
    if (int rc = launch_checked([&] {
            hipLaunchKernelGGL(dec_fwd_kernel, dim3(m < 1024 ? m : 1024), dim3(TPB), lds_dec, (hipStream_t)stream, *p, h_all, m,
                               d_all, saved);
        }, "dec_fwd")) return rc;
    if (!out_all) return 0;   // the caller integrates later / elsewhere (sur_chunk_integrate)
    return sur_chunk_integrate(stream, p, states_t, d_all, k, s, b, out_all);
}

int sur_chunk_integrate(void* stream, const sur_chunk_params* p, const float* states_t, const float* d_all, int k, int s, int b,
                        float* out_all) {
    if (!p || !states_t || !d_all || !out_all || k <= 0 || b <= 0 || s < 1) return fail(-1, "sur_chunk_integrate: bad argument");
    const int n = 4 * p->hq;
    return launch_checked([&] {
        hipLaunchKernelGGL(integrate_kernel, dim3((b * n + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)stream, states_t, d_all, k,
                           s, b, n, p->delta, p->mul, p->add, out_all);
    }, "integrate");
}

int sur_chunk_workspace_floats(const sur_chunk_params* p, int k, int b) {
    if (!p || k <= 0 || b <= 0) return 0;
    return k * b * (5 * p->cs * p->hq + 4 * p->hq);   // dh_dec [K,B,cs,hq] + dg_all [K,B,4,cs,hq] + ga_all [K,B,1,N]
}

static int chunks_backward_impl(void* stream, const sur_chunk_params* p, const ChunkSpans& spans, const float* xlat_t,
                                const float* h_all, const float* c_all, const float* dd_all, const float* dout_all,
                                const float* dh_all, const float* dc_all, int k_total, int b, float* dxlat_t, float* dh0,
                                float* dc0, int row_base, int row_count, const float* saved, float* workspace, const char* who) {
    if (int rc = chunk_geometry(p, who)) return rc;
    if (!saved || !workspace) return fail(-1, "%s: needs the `saved` buffer the forward filled and a workspace", who);
    if (sur_chunk_saved_floats(p) == 0) return fail(-4, "%s: hq = %d, ca = %d, cs = %d: geometry not supported", who, p->hq, p->ca, p->cs);
    if (!p->partial || row_base < 0 || row_count < spans.n * b || p->rows < row_base + row_count)
        return fail(-1, "%s: partial gradient buffer has %d rows, need [%d, %d) with at least %d of them", who, p->rows, row_base,
                    row_base + row_count, spans.n * b);
    // decoder backward of all (step, sample) pairs in parallel, then the cell chains (one workgroup per chunk and
    // sample), then dx and the LSTM weight gradients of all pairs in parallel
    int psize_lstm = 0, psize_dec = 0;
    for (int i = 0; i < ST_NLSTM; ++i) psize_lstm += p->size[i];
    for (int i = ST_NLSTM; i < SUR_ST_NPARAM; ++i) psize_dec += p->size[i];
    const int m = k_total * b, n = 4 * p->hq, sl = p->cs * p->hq;
    float* dh_dec = workspace;
    float* dg_all = workspace + (size_t)m * sl;
    float* ga_all = workspace + (size_t)m * 5 * sl;
    const float* ga = dd_all;
    if (dout_all || !dd_all) {
        if (spans.n != 1) return fail(-1, "%s: gradients wrt the outputs are supported for one chunk at a time", who);
        if (int rc = launch_checked([&] {
                hipLaunchKernelGGL(dgrad_scan_kernel, dim3((b * n + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)stream, dd_all,
                                   dout_all, k_total, spans.sp[0].s, b, n, p->delta * p->mul, ga_all);
            }, "dgrad_scan")) return rc;
        ga = ga_all;
    }
    const size_t dec_base = sizeof(float) * (dec_act_floats(*p, true) + psize_dec);
    const int dec_gl = accumulators_fit(dec_base + sizeof(float) * psize_dec) ? 1 : 0;
    const size_t lds_dec = dec_base + (dec_gl ? sizeof(float) * psize_dec : 0);
    const size_t lds_cell = sizeof(float) * (cell_bwd_act_floats(*p) + psize_lstm);
    const size_t wg_base = sizeof(float) * (cell_wgrad_act_floats(*p) + p->size[SUR_ST_WXI] + p->size[SUR_ST_WXF] + p->size[SUR_ST_WXC] +
                                            p->size[SUR_ST_WXO]);      // activations + the staged Wx_g
    const int wg_gl = accumulators_fit(wg_base + sizeof(float) * psize_lstm) ? 1 : 0;
    const size_t lds_wg = wg_base + (wg_gl ? sizeof(float) * psize_lstm : 0);
    if (int rc = dec_gl ? set_lds(dec_bwd_kernel<true>, lds_dec, "decoder backward") : set_lds(dec_bwd_kernel<false>, lds_dec, "decoder backward"))
        return rc;
    if (int rc = wg_gl ? set_lds(cell_wgrad_kernel<true>, lds_wg, "cell weight gradients") : set_lds(cell_wgrad_kernel<false>, lds_wg, "cell weight gradients"))
        return rc;
    const int grid = m < row_count ? m : row_count;
    if (int rc = launch_checked([&] {
            if (dec_gl) hipLaunchKernelGGL(dec_bwd_kernel<true>, dim3(grid), dim3(TPB), lds_dec, (hipStream_t)stream, *p, saved, ga, m, dh_dec, row_base);
            else hipLaunchKernelGGL(dec_bwd_kernel<false>, dim3(grid), dim3(TPB), lds_dec, (hipStream_t)stream, *p, saved, ga, m, dh_dec, row_base);
        }, "dec_bwd")) return rc;
    const int chain_threads = cell_chain_threads(*p);
    auto launch_cell_bwd = [&](auto kernel) -> int {
        if (int rc = set_lds(kernel, lds_cell, "cell backward")) return rc;
        return launch_checked([&] {
            hipLaunchKernelGGL(kernel, dim3(spans.n * b), dim3(chain_threads), lds_cell, (hipStream_t)stream, *p, spans, c_all, saved,
                               dh_dec, dh_all, dc_all, b, dg_all, dh0, dc0);
        }, "cell_bwd");
    };
    if (int rc = chain_threads == TPB ? launch_cell_bwd(cell_bwd_kernel<TPB>)
                                      : (chain_threads == 2 * TPB ? launch_cell_bwd(cell_bwd_kernel<2 * TPB>)
                                                                  : launch_cell_bwd(cell_bwd_kernel<4 * TPB>)))
        return rc;
    return launch_checked([&] {
        if (wg_gl) hipLaunchKernelGGL(cell_wgrad_kernel<true>, dim3(grid), dim3(TPB), lds_wg, (hipStream_t)stream, *p, spans, xlat_t, h_all, dg_all, k_total, b, dxlat_t, row_base);
        else hipLaunchKernelGGL(cell_wgrad_kernel<false>, dim3(grid), dim3(TPB), lds_wg, (hipStream_t)stream, *p, spans, xlat_t, h_all, dg_all, k_total, b, dxlat_t, row_base);
    }, "cell_wgrad");
}

int sur_chunk_backward(void* stream, const sur_chunk_params* p, const float* xlat_t, const float* lstates_t,
                       const float* h0, const float* c0, int hc_bstride, const float* h_all, const float* c_all,
                       const float* dd_all, const float* dout_all, const float* dh_all, const float* dc_all, int k, int s, int b,
                       float* dxlat_t, float* dlstates_t, float* dh0, float* dc0, int row_base, int row_count,
                       const float* saved, float* workspace) {
    if (!p || !xlat_t || !lstates_t || !h0 || !c0 || !h_all || !c_all || k <= 0 || b <= 0 || s < 1)
        return fail(-1, "sur_chunk_backward: bad argument");
    ChunkSpans spans{};
    spans.n = 1;
    spans.sp[0] = sur_chunk_span{0, k, s < k ? s : k, lstates_t, h0, c0, hc_bstride, dlstates_t};
    return chunks_backward_impl(stream, p, spans, xlat_t, h_all, c_all, dd_all, dout_all, dh_all, dc_all, k, b, dxlat_t, dh0, dc0,
                                row_base, row_count, saved, workspace, "sur_chunk_backward");
}

int sur_chunks_backward(void* stream, const sur_chunk_params* p, int nspans, const sur_chunk_span* spans_in, const float* xlat_t,
                        const float* h_all, const float* c_all, const float* dd_all, int k_total, int b, float* dxlat_t,
                        int row_base, int row_count, const float* saved, float* workspace) {
    if (!p || !spans_in || nspans < 1 || nspans > SUR_MAX_SPANS || !xlat_t || !h_all || !c_all || !dd_all || k_total <= 0 || b <= 0)
        return fail(-1, "sur_chunks_backward: bad argument (1 to %d chunks)", SUR_MAX_SPANS);
    ChunkSpans spans{};
    spans.n = nspans;
    int expect = 0;
    for (int j = 0; j < nspans; ++j) {
        const sur_chunk_span& sp = spans_in[j];
        if (sp.k0 != expect || sp.k1 <= sp.k0 || sp.s < 1 || sp.s > sp.k1 - sp.k0 || !sp.lstates_t || !sp.h0 || !sp.c0 || sp.hc_bstride < 0)
            return fail(-1, "sur_chunks_backward: chunk %d: spans must tile [0, K) in order, with 1 <= s <= k1 - k0", j);
        spans.sp[j] = sp;
        expect = sp.k1;
    }
    if (expect != k_total) return fail(-1, "sur_chunks_backward: the chunks cover [0, %d), not [0, %d)", expect, k_total);
    return chunks_backward_impl(stream, p, spans, xlat_t, h_all, c_all, dd_all, nullptr, nullptr, nullptr, k_total, b, dxlat_t, nullptr,
                                nullptr, row_base, row_count, saved, workspace, "sur_chunks_backward");
}

int sur_flush_chunk_grads(void* stream, const sur_chunk_params* p, const sur_adam* adam, int overwrite) {
    if (!p || !p->partial) return fail(-1, "sur_flush_chunk_grads: bad argument");
    for (int i = 0; i < SUR_ST_NPARAM; ++i)
        if (!p->g[i]) return fail(-1, "sur_flush_chunk_grads: gradient tensor %d is NULL", i);
    const int psize = psize_of<SUR_ST_NPARAM>(p->size);
    sur_adam ad{};
    if (adam) {
        if (!adam->m || !adam->v || !adam->step || !adam->ticket || !adam->lr)
            return fail(-1, "flush: incomplete Adam descriptor");
        ad = *adam;
    }
    return launch_checked([&] {
        hipLaunchKernelGGL((flush_grads_kernel<SUR_ST_NPARAM, sur_chunk_params>), dim3(flush_blocks(psize)), dim3(FLUSH_TPB), 0,
                           (hipStream_t)stream, *p, psize, ad, overwrite);
    }, "flush_chunk");
}

int sur_flush_all_grads(void* stream, const sur_encoder_params* e0, const sur_adam* a0, const sur_encoder_params* e1,
                        const sur_adam* a1, const sur_chunk_params* c2, const sur_adam* a2, int overwrite_mask) {
    if (!e0 || !e1 || !c2 || !e0->partial || !e1->partial || !c2->partial) return fail(-1, "sur_flush_all_grads: bad argument");
    for (int i = 0; i < SUR_ENC_NPARAM; ++i)
        if (!e0->g[i] || !e1->g[i]) return fail(-1, "sur_flush_all_grads: encoder gradient tensor %d is NULL", i);
    for (int i = 0; i < SUR_ST_NPARAM; ++i)
        if (!c2->g[i]) return fail(-1, "sur_flush_all_grads: chunk gradient tensor %d is NULL", i);
    const sur_adam* in[3] = {a0, a1, a2};
    sur_adam ad[3] = {};
    for (int j = 0; j < 3; ++j)
        if (in[j]) {
            if (!in[j]->m || !in[j]->v || !in[j]->step || !in[j]->ticket || !in[j]->lr)
                return fail(-1, "sur_flush_all_grads: incomplete Adam descriptor %d", j);
            ad[j] = *in[j];
        }
    const int n0 = psize_of<SUR_ENC_NPARAM>(e0->size), n1 = psize_of<SUR_ENC_NPARAM>(e1->size), n2 = psize_of<SUR_ST_NPARAM>(c2->size);
    const int grid = flush_blocks(n0) + flush_blocks(n1) + flush_blocks(n2);
    return launch_checked([&] {
        hipLaunchKernelGGL(flush_all_kernel, dim3(grid), dim3(FLUSH_TPB), 0, (hipStream_t)stream, *e0, ad[0], n0, *e1, ad[1], n1, *c2, ad[2],
                           n2, overwrite_mask);
    }, "flush_all");
}

int sur_fold_rows(void* stream, const sur_encoder_params* e0, const sur_encoder_params* e1, const sur_chunk_params* c2,
                  const int* bases, const int* counts, const int* dsts) {
    if (!e0 || !e1 || !c2 || !bases || !counts || !dsts) return fail(-1, "sur_fold_rows: bad argument");
    float* partial[3] = {e0->partial, e1->partial, c2->partial};
    const int rows[3] = {e0->rows, e1->rows, c2->rows};
    const int psize[3] = {psize_of<SUR_ENC_NPARAM>(e0->size), psize_of<SUR_ENC_NPARAM>(e1->size), psize_of<SUR_ST_NPARAM>(c2->size)};
    FoldJob jobs[3];
    int grid = 0;
    for (int j = 0; j < 3; ++j) {
        if (!partial[j] || bases[j] < 0 || counts[j] <= 0 || bases[j] + counts[j] > rows[j] || dsts[j] < 0 || dsts[j] >= rows[j] ||
            (dsts[j] >= bases[j] && dsts[j] < bases[j] + counts[j]))
            return fail(-1, "sur_fold_rows: pack %d: rows [%d, %d) -> row %d do not fit the buffer of %d rows", j, bases[j],
                        bases[j] + counts[j], dsts[j], rows[j]);
        jobs[j] = FoldJob{partial[j], psize[j], bases[j], counts[j], dsts[j]};
        grid += flush_blocks(psize[j]);
    }
    return launch_checked([&] {
        hipLaunchKernelGGL(fold_rows_kernel, dim3(grid), dim3(FLUSH_TPB), 0, (hipStream_t)stream, jobs[0], jobs[1], jobs[2]);
    }, "fold_rows");
}

int sur_adam_apply(void* stream, const sur_encoder_params* e0, const sur_adam* a0, const sur_encoder_params* e1,
                   const sur_adam* a1, const sur_chunk_params* c2, const sur_adam* a2) {
    // a pack without a descriptor is skipped (its size counts as 0)
    const sur_adam* in[3] = {a0, a1, a2};
    sur_adam ad[3] = {};
    for (int j = 0; j < 3; ++j)
        if (in[j]) {
            if (!in[j]->m || !in[j]->v || !in[j]->step || !in[j]->ticket || !in[j]->lr)
                return fail(-1, "sur_adam_apply: incomplete Adam descriptor %d", j);
            ad[j] = *in[j];
        }
    if ((a0 && !e0) || (a1 && !e1) || (a2 && !c2)) return fail(-1, "sur_adam_apply: descriptor without its parameter pack");
    sur_encoder_params pe0{}, pe1{};
    sur_chunk_params pc2{};
    int n0 = 0, n1 = 0, n2 = 0;
    if (a0) {
        pe0 = *e0;
        n0 = psize_of<SUR_ENC_NPARAM>(e0->size);
        for (int i = 0; i < SUR_ENC_NPARAM; ++i)
            if (!e0->g[i] || !e0->w[i]) return fail(-1, "sur_adam_apply: encoder 0 tensor %d is NULL", i);
    }
    if (a1) {
        pe1 = *e1;
        n1 = psize_of<SUR_ENC_NPARAM>(e1->size);
        for (int i = 0; i < SUR_ENC_NPARAM; ++i)
            if (!e1->g[i] || !e1->w[i]) return fail(-1, "sur_adam_apply: encoder 1 tensor %d is NULL", i);
    }
    if (a2) {
        pc2 = *c2;
        n2 = psize_of<SUR_ST_NPARAM>(c2->size);
        for (int i = 0; i < SUR_ST_NPARAM; ++i)
            if (!c2->g[i] || !c2->w[i]) return fail(-1, "sur_adam_apply: chunk tensor %d is NULL", i);
    }
    const int grid = (n0 + TPB - 1) / TPB + (n1 + TPB - 1) / TPB + (n2 + TPB - 1) / TPB;
    if (grid == 0) return 0;
    return launch_checked([&] {
        hipLaunchKernelGGL(adam_all_kernel, dim3(grid), dim3(TPB), 0, (hipStream_t)stream, pe0, ad[0], n0, pe1, ad[1], n1, pc2, ad[2], n2);
    }, "adam_all");
}

static int delta_loss_launch(void* stream, const float* states, long states_bstride, long states_tstride, const float* d_all, int b,
                             int t, int n, float delta, float mean, float stdv, float* deltas, float* dd_all, float* hsteploss,
                             float* loss, float* stats, double* partial, unsigned int* ticket, int t_begin, int t_end,
                             const char* who, int take_ticket = 1) {
    if (!states || !d_all || !deltas || !hsteploss || !loss || !stats || !partial || !ticket || b <= 0 || t < 2 || n <= 0)
        return fail(-1, "%s: bad argument (need B > 0, T >= 2, N > 0)", who);
    if (t_begin < 0 || t_end > t || t_begin >= t_end) return fail(-1, "%s: time range [%d, %d) outside [0, %d)", who, t_begin, t_end, t);
    if (states_bstride < n || states_tstride < n) return fail(-1, "%s: state strides must be at least N", who);
    if (!(delta != 0.0f) || !(stdv > 0.0f)) return fail(-1, "%s: delta must be non-zero and std positive", who);
    int nsplit = (b * n + LOSS_UNROLL * TPB - 1) / (LOSS_UNROLL * TPB);
    nsplit = nsplit < 1 ? 1 : (nsplit > LOSS_MAX_SPLIT ? LOSS_MAX_SPLIT : nsplit);
    return launch_checked([&] {
        hipLaunchKernelGGL(delta_loss_kernel, dim3(t_end - t_begin, nsplit), dim3(TPB), 0, (hipStream_t)stream, states, states_bstride,
                           states_tstride, d_all, b, t, n, delta, mean,
                           stdv, deltas, dd_all, hsteploss, loss, stats, partial, ticket, t_begin, take_ticket);
    }, "delta_loss");
}

static int loss_nsplit(int b, int n) {
    const int nsplit = (b * n + LOSS_UNROLL * TPB - 1) / (LOSS_UNROLL * TPB);
    return nsplit < 1 ? 1 : (nsplit > LOSS_MAX_SPLIT ? LOSS_MAX_SPLIT : nsplit);
}

int sur_tbptt_delta_loss(void* stream, const float* states, long states_bstride, long states_tstride, const float* d_all, int b,
                         int t, int n, float delta, float mean,
                         float stdv, float* deltas, float* dd_all, float* hsteploss, float* loss, float* stats,
                         double* partial, unsigned int* ticket) {
    return delta_loss_launch(stream, states, states_bstride, states_tstride, d_all, b, t, n, delta, mean, stdv, deltas, dd_all,
                             hsteploss, loss, stats, partial, ticket, 0, t, "sur_tbptt_delta_loss");
}

int sur_tbptt_delta_loss_range(void* stream, const float* states, long states_bstride, long states_tstride, const float* d_all,
                               int b, int t, int n, float delta, float mean, float stdv, float* deltas, float* dd_all,
                               float* hsteploss, float* loss, float* stats, double* partial, unsigned int* ticket, int t_begin,
                               int t_end) {
    return delta_loss_launch(stream, states, states_bstride, states_tstride, d_all, b, t, n, delta, mean, stdv, deltas, dd_all,
                             hsteploss, loss, stats, partial, ticket, t_begin, t_end, "sur_tbptt_delta_loss_range");
}

int sur_tbptt_delta_loss_rows(void* stream, const float* states, long states_bstride, long states_tstride, const float* d_all,
                              int b, int t, int n, float delta, float mean, float stdv, float* deltas, float* dd_all,
                              float* hsteploss, float* loss, float* stats, double* partial, unsigned int* ticket, int t_begin,
                              int t_end) {
    return delta_loss_launch(stream, states, states_bstride, states_tstride, d_all, b, t, n, delta, mean, stdv, deltas, dd_all,
                             hsteploss, loss, stats, partial, ticket, t_begin, t_end, "sur_tbptt_delta_loss_rows", 0);
}

int sur_tbptt_delta_loss_finalize(void* stream, int b, int t, int n, float* hsteploss, float* loss, float* stats, double* partial,
                                  unsigned int* ticket) {
    if (!hsteploss || !loss || !stats || !partial || !ticket || b <= 0 || t < 2 || n <= 0)
        return fail(-1, "sur_tbptt_delta_loss_finalize: bad argument");
    const int nsplit = loss_nsplit(b, n);
    return launch_checked([&] {
        hipLaunchKernelGGL(delta_loss_finalize_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream, b, t, n, nsplit, hsteploss, loss,
                           stats, partial, ticket);
    }, "delta_loss_finalize");
}

}  // extern "C"
