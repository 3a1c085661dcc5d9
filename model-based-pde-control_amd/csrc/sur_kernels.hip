// sur_kernels.hip -- fused gfx950 kernels for the surrogate's TBPTT step (C ABI: include/surrogate_hip.h).
//
// Reference arithmetic (paths relative to the reference root):
//   pdecontrol/surrogates/models/cnn.py:126-145   ResidualBlock.forward (conv3 -> SiLU -> LN, twice; 1x1 skip; LN)
//   pdecontrol/surrogates/models/cnn.py:35-41,64-70  ConvBlock / DeConvolutionBlock (conv -> act -> LN)
//   pdecontrol/surrogates/transition.py:218-226   CNNLSTMCell.forward
//   pdecontrol/surrogates/surrogate.py:97-107     one rollout step: cell -> decoder -> integrate
//   pdecontrol/architectures/autoreg.py:51-94     channel / kernel / stride / padding choices
//
// Execution model: ONE workgroup (256 threads) per sample, every activation of the module in LDS,
// parameters read through L1/L2 (the whole model is 39 KB).  At the reference's sizes (channels <= 16,
// widths <= 64) a layer is a few hundred to a few thousand MACs: far too small for MFMA tiles and
// bound by launch latency when run as separate kernels, so a whole module (3 residual blocks, or
// LSTM cell + 4-layer decoder + integration) is one launch.  Backward kernels first recompute the
// forward intermediates from the module inputs (cheaper than writing/reading them through HBM),
// then back-propagate.  The time loop of a TBPTT chunk runs inside the kernel (weights and hidden
// state stay in LDS across steps).  Parameter gradients are summed over space, time and the
// workgroup's samples in LDS, added to the workgroup's own row of a partial buffer (no atomics, so
// results are deterministic) and reduced over rows by a flush kernel.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/surrogate_hip.h"

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

__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ int wrapi(int j, int n) { return j < 0 ? j + n : (j >= n ? j - n : j); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// block-cooperative layer primitives on LDS-resident activations [C][H] (row-major, H contiguous).
// Every primitive ends with __syncthreads().
// ---------------------------------------------------------------------------------------------

// The loops below are latency-bound (dependent LDS read -> FMA chains), not bandwidth-bound, so they are
// written for instruction-level parallelism: kernel width K is a template parameter (fully unrolled,
// wrapped indices hoisted out of the channel loop), channel loops are unrolled by 4 with independent
// accumulators, and layers with fewer outputs than threads split their reduction over SPLIT lanes.

// circular Conv1d: out[o][p] (+)= bias[o] + sum_{ci,k} W[o][ci][k] * in[ci][(p*stride + k - pad) mod hin]
template <int K>
__device__ void conv_fwd(const float* in, int cin, int hin, const float* W, const float* bias, int cout, int stride,
                         int pad, float* out, bool accumulate) {
    const int hout = hin / stride, total = cout * hout;
    // split the input-channel reduction over `split` adjacent lanes when the layer is small
    int split = 1;
    while (split < 8 && total * split * 2 <= (int)blockDim.x && cin % (split * 2) == 0) split *= 2;
    const int cper = cin / split;
    for (int base = 0; base < total * split; base += blockDim.x) {
        const int t = base + threadIdx.x;
        const bool live = t < total * split;
        const int idx = live ? t / split : 0, part = t % split;
        const int o = idx / hout, p = idx - o * hout;
        int jj[K];
#pragma unroll
        for (int k = 0; k < K; ++k) jj[k] = wrapi(p * stride + k - pad, hin);
        const float* w = W + ((size_t)o * cin + part * cper) * K;
        const float* row = in + (part * cper) * hin;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        int ci = 0;
        for (; ci + 4 <= cper; ci += 4) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                a0 = fmaf(w[(ci + 0) * K + k], row[(ci + 0) * hin + jj[k]], a0);
                a1 = fmaf(w[(ci + 1) * K + k], row[(ci + 1) * hin + jj[k]], a1);
                a2 = fmaf(w[(ci + 2) * K + k], row[(ci + 2) * hin + jj[k]], a2);
                a3 = fmaf(w[(ci + 3) * K + k], row[(ci + 3) * hin + jj[k]], a3);
            }
        }
        for (; ci < cper; ++ci) {
#pragma unroll
            for (int k = 0; k < K; ++k) a0 = fmaf(w[ci * K + k], row[ci * hin + jj[k]], a0);
        }
        float acc = (a0 + a1) + (a2 + a3);
        for (int m = 1; m < split; m <<= 1) acc += __shfl_xor(acc, m, 64);
        if (live && part == 0) {
            if (bias) acc += bias[o];
            out[idx] = accumulate ? out[idx] + acc : acc;
        }
    }
    __syncthreads();
}

// din[ci][j] (+)= sum_{o,k : (p*stride + k - pad) mod hin == j} W[o][ci][k] * dout[o][p]
template <int K>
__device__ void conv_bwd_data(const float* dout, int cout, int hin, const float* W, int cin, int stride, int pad,
                              float* din, bool accumulate) {
    const int hout = hin / stride;
    for (int idx = threadIdx.x; idx < cin * hin; idx += blockDim.x) {
        const int ci = idx / hin, j = idx - ci * hin;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int t = wrapi(j - k + pad, hin);
            if (t % stride) continue;
            const int p = t / stride;
            const float* w = W + (size_t)ci * K + k;
            const float* d = dout + p;
            int o = 0;
            for (; o + 4 <= cout; o += 4) {
                a0 = fmaf(w[(size_t)(o + 0) * cin * K], d[(o + 0) * hout], a0);
                a1 = fmaf(w[(size_t)(o + 1) * cin * K], d[(o + 1) * hout], a1);
                a2 = fmaf(w[(size_t)(o + 2) * cin * K], d[(o + 2) * hout], a2);
                a3 = fmaf(w[(size_t)(o + 3) * cin * K], d[(o + 3) * hout], a3);
            }
            for (; o < cout; ++o) a0 = fmaf(w[(size_t)o * cin * K], d[o * hout], a0);
        }
        const float acc = (a0 + a1) + (a2 + a3);
        din[idx] = accumulate ? din[idx] + acc : acc;
    }
    __syncthreads();
}

// gW[o][ci][k] += sum_p dout[o][p] * in[ci][(p*stride + k - pad) mod hin];  gb[o] += sum_p dout[o][p]
template <int K>
__device__ void conv_bwd_weight(const float* dout, int cout, const float* in, int cin, int hin, int stride, int pad,
                                float* gW, float* gb) {
    const int hout = hin / stride;
    for (int idx = threadIdx.x; idx < cout * cin * K; idx += blockDim.x) {
        const int o = idx / (cin * K), r = idx - o * cin * K, ci = r / K, k = r - ci * K;
        const float* d = dout + o * hout;
        const float* row = in + ci * hin;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (int p = 0; p < hout; p += 4) {  // hout is a multiple of 4 for every layer of this model family
            a0 = fmaf(d[p + 0], row[wrapi((p + 0) * stride + k - pad, hin)], a0);
            a1 = fmaf(d[p + 1], row[wrapi((p + 1) * stride + k - pad, hin)], a1);
            a2 = fmaf(d[p + 2], row[wrapi((p + 2) * stride + k - pad, hin)], a2);
            a3 = fmaf(d[p + 3], row[wrapi((p + 3) * stride + k - pad, hin)], a3);
        }
        gW[idx] += (a0 + a1) + (a2 + a3);
    }
    if (gb) {
        for (int o = threadIdx.x; o < cout; o += blockDim.x) {
            float a0 = 0.0f, a1 = 0.0f;
            for (int p = 0; p < hout; p += 2) {
                a0 += dout[o * hout + p];
                a1 += dout[o * hout + p + 1];
            }
            gb[o] += a0 + a1;
        }
    }
    __syncthreads();
}

// ConvTranspose1d(k=3, stride=2, padding=1, output_padding=1), zero padded; W[ci][o][k]; hout = 2*hin
// out[o][j] = b[o] + sum_{ci,k : j = 2i - 1 + k} W[ci][o][k] * in[ci][i]
__device__ void deconv_fwd(const float* in, int cin, int hin, const float* W, const float* bias, int cout,
                           float* out) {
    const int hout = 2 * hin;
    for (int idx = threadIdx.x; idx < cout * hout; idx += blockDim.x) {
        const int o = idx / hout, j = idx - o * hout;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int t = j + 1 - k;
            if (t < 0 || (t & 1)) continue;
            const int i = t >> 1;
            if (i >= hin) continue;
            const float* w = W + (size_t)o * 3 + k;
            const float* x = in + i;
            int ci = 0;
            for (; ci + 4 <= cin; ci += 4) {
                a0 = fmaf(w[(size_t)(ci + 0) * cout * 3], x[(ci + 0) * hin], a0);
                a1 = fmaf(w[(size_t)(ci + 1) * cout * 3], x[(ci + 1) * hin], a1);
                a2 = fmaf(w[(size_t)(ci + 2) * cout * 3], x[(ci + 2) * hin], a2);
                a3 = fmaf(w[(size_t)(ci + 3) * cout * 3], x[(ci + 3) * hin], a3);
            }
            for (; ci < cin; ++ci) a0 = fmaf(w[(size_t)ci * cout * 3], x[ci * hin], a0);
        }
        out[idx] = bias[o] + ((a0 + a1) + (a2 + a3));
    }
    __syncthreads();
}

__device__ void deconv_bwd_data(const float* dout, int cout, int hin, const float* W, int cin, float* din) {
    const int hout = 2 * hin;
    for (int idx = threadIdx.x; idx < cin * hin; idx += blockDim.x) {
        const int ci = idx / hin, i = idx - ci * hin;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = 2 * i - 1 + k;
            if (j < 0 || j >= hout) continue;
            const float* w = W + (size_t)ci * cout * 3 + k;
            const float* d = dout + j;
            int o = 0;
            for (; o + 4 <= cout; o += 4) {
                a0 = fmaf(w[(o + 0) * 3], d[(o + 0) * hout], a0);
                a1 = fmaf(w[(o + 1) * 3], d[(o + 1) * hout], a1);
                a2 = fmaf(w[(o + 2) * 3], d[(o + 2) * hout], a2);
                a3 = fmaf(w[(o + 3) * 3], d[(o + 3) * hout], a3);
            }
            for (; o < cout; ++o) a0 = fmaf(w[o * 3], d[o * hout], a0);
        }
        din[idx] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
}

__device__ void deconv_bwd_weight(const float* dout, int cout, const float* in, int cin, int hin, float* gW, float* gb) {
    const int hout = 2 * hin;
    for (int idx = threadIdx.x; idx < cin * cout * 3; idx += blockDim.x) {
        const int ci = idx / (cout * 3), r = idx - ci * cout * 3, o = r / 3, k = r - o * 3;
        const float* x = in + ci * hin;
        const float* d = dout + o * hout + (k - 1);  // j = 2i - 1 + k
        float a0 = 0.0f, a1 = 0.0f;
        // i = 0 (k = 0) and i = hin-1 (k = 2 -> j = hout) fall outside the output: handle the ends explicitly
        const int i_lo = (k == 0) ? 1 : 0, i_hi = hin;  // j = 2i + k - 1 < hout always holds for k <= 2, i < hin
        int i = i_lo;
        for (; i + 2 <= i_hi; i += 2) {
            a0 = fmaf(x[i], d[2 * i], a0);
            a1 = fmaf(x[i + 1], d[2 * i + 2], a1);
        }
        for (; i < i_hi; ++i) a0 = fmaf(x[i], d[2 * i], a0);
        gW[idx] += a0 + a1;
    }
    for (int o = threadIdx.x; o < cout; o += blockDim.x) {
        float a0 = 0.0f, a1 = 0.0f;
        for (int j = 0; j < hout; j += 2) {
            a0 += dout[o * hout + j];
            a1 += dout[o * hout + j + 1];
        }
        gb[o] += a0 + a1;
    }
    __syncthreads();
}

// out = LayerNorm_H(act(pre)) * gamma[p] + beta[p], act = SiLU or identity; one wave per channel
__device__ void act_ln_fwd(const float* pre, int C, int H, const float* gamma,
                           const float* beta, bool silu, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int c = wave; c < C; c += nw) {
        const float* x = pre + c * H;
        float s = 0.0f;
        for (int p = lane; p < H; p += 64) {
            const float v = x[p];
            s += silu ? v * sigmoid_(v) : v;
        }
        const float mean = wave_sum(s) / H;
        float ss = 0.0f;
        for (int p = lane; p < H; p += 64) {
            const float v = x[p];
            const float y = (silu ? v * sigmoid_(v) : v) - mean;
            ss = fmaf(y, y, ss);
        }
        const float rstd = rsqrtf(wave_sum(ss) / H + LN_EPS);
        for (int p = lane; p < H; p += 64) {
            const float v = x[p];
            const float y = silu ? v * sigmoid_(v) : v;
            out[c * H + p] = fmaf((y - mean) * rstd, gamma[p], beta[p]);
        }
    }
    __syncthreads();
}

// backward of act_ln_fwd: dpre from dout; accumulates ggamma / gbeta.  xhat_scratch: [C][H] work space.
__device__ void act_ln_bwd(const float* dout, const float* pre, int C, int H, const float* gamma,
                           bool silu, float* dpre, float* xhat_scratch, float* ggamma, float* gbeta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int c = wave; c < C; c += nw) {
        const float* x = pre + c * H;
        float s = 0.0f;
        for (int p = lane; p < H; p += 64) {
            const float v = x[p];
            s += silu ? v * sigmoid_(v) : v;
        }
        const float mean = wave_sum(s) / H;
        float ss = 0.0f;
        for (int p = lane; p < H; p += 64) {
            const float v = x[p];
            const float y = (silu ? v * sigmoid_(v) : v) - mean;
            ss = fmaf(y, y, ss);
        }
        const float rstd = rsqrtf(wave_sum(ss) / H + LN_EPS);
        float m1 = 0.0f, m2 = 0.0f;
        for (int p = lane; p < H; p += 64) {
            const float v = x[p];
            const float xh = ((silu ? v * sigmoid_(v) : v) - mean) * rstd;
            const float dxh = dout[c * H + p] * gamma[p];
            xhat_scratch[c * H + p] = xh;
            m1 += dxh;
            m2 = fmaf(dxh, xh, m2);
        }
        m1 = wave_sum(m1) / H;
        m2 = wave_sum(m2) / H;
        for (int p = lane; p < H; p += 64) {
            const float v = x[p];
            const float xh = xhat_scratch[c * H + p];
            const float dxh = dout[c * H + p] * gamma[p];
            float dy = rstd * (dxh - m1 - xh * m2);
            if (silu) {
                const float sg = sigmoid_(v);
                dy *= sg * (1.0f + v * (1.0f - sg));
            }
            dpre[c * H + p] = dy;
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < H; p += blockDim.x) {
        float gg = 0.0f, gb = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float d = dout[c * H + p];
            gg = fmaf(d, xhat_scratch[c * H + p], gg);
            gb += d;
        }
        ggamma[p] += gg;
        gbeta[p] += gb;
    }
    __syncthreads();
}

__device__ void lds_load(float* dst, const float* __restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
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
        float* dst = lds_w + off;
        for (int j = threadIdx.x; j < size[i]; j += blockDim.x) dst[j] = gw[i][j];
        v.w[i] = dst;
        off += size[i];
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
    conv_fwd<1>(b.in, b.cin, b.hin, w[SUR_RB_SKIP], nullptr, b.cout, b.stride, 0, b.skip, false);
    conv_fwd<3>(b.in, b.cin, b.hin, w[SUR_RB_CONV1], nullptr, b.cout, b.stride, 1, b.a1pre, false);
    act_ln_fwd(b.a1pre, b.cout, b.hout, w[SUR_RB_LN1_W], w[SUR_RB_LN1_B], true, b.a1);
    conv_fwd<3>(b.a1, b.cout, b.hout, w[SUR_RB_CONV2], nullptr, b.cout, 1, 1, b.a2pre, false);
    act_ln_fwd(b.a2pre, b.cout, b.hout, w[SUR_RB_LN2_W], w[SUR_RB_LN2_B], true, b.a2);
    const int n = b.cout * b.hout;
    for (int i = threadIdx.x; i < n; i += blockDim.x) b.s[i] = b.a2[i] + b.skip[i];
    __syncthreads();
    act_ln_fwd(b.s, b.cout, b.hout, w[SUR_RB_LN3_W], w[SUR_RB_LN3_B], false, b.out);
}

// dout [cout][hout] -> din [cin][hin]; g1, g2, g3, xh: scratch of cout*hout floats each
__device__ void rb_backward(const RBBuf& b, const float* const* w, float* const* g, const float* dout, float* din,
                            float* g1, float* g2, float* g3, float* xh) {
    act_ln_bwd(dout, b.s, b.cout, b.hout, w[SUR_RB_LN3_W], false, g1, xh, g[SUR_RB_LN3_W], g[SUR_RB_LN3_B]);
    // skip path
    conv_bwd_weight<1>(g1, b.cout, b.in, b.cin, b.hin, b.stride, 0, g[SUR_RB_SKIP], nullptr);
    conv_bwd_data<1>(g1, b.cout, b.hin, w[SUR_RB_SKIP], b.cin, b.stride, 0, din, false);
    // residual path
    act_ln_bwd(g1, b.a2pre, b.cout, b.hout, w[SUR_RB_LN2_W], true, g2, xh, g[SUR_RB_LN2_W], g[SUR_RB_LN2_B]);
    conv_bwd_weight<3>(g2, b.cout, b.a1, b.cout, b.hout, 1, 1, g[SUR_RB_CONV2], nullptr);
    conv_bwd_data<3>(g2, b.cout, b.hout, w[SUR_RB_CONV2], b.cout, 1, 1, g3, false);
    act_ln_bwd(g3, b.a1pre, b.cout, b.hout, w[SUR_RB_LN1_W], true, g1, xh, g[SUR_RB_LN1_W], g[SUR_RB_LN1_B]);
    conv_bwd_weight<3>(g1, b.cout, b.in, b.cin, b.hin, b.stride, 1, g[SUR_RB_CONV1], nullptr);
    conv_bwd_data<3>(g1, b.cout, b.hin, w[SUR_RB_CONV1], b.cin, b.stride, 1, din, true);
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

__global__ void __launch_bounds__(TPB) enc_fwd_kernel(const sur_encoder_params p, const float* __restrict__ x, int m_total,
                                                      float* __restrict__ z) {
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
        lds_store(z + (size_t)m * nout, L.rb[2].out, nout);
    }
}

__global__ void __launch_bounds__(TPB) enc_bwd_kernel(const sur_encoder_params p, const float* __restrict__ x,
                                                      const float* __restrict__ dz, int m_total, float* __restrict__ dx,
                                                      int grads_in_lds) {
    extern __shared__ __align__(16) float lds[];
    EncLayout L;
    enc_layout(p, lds, true, L);
    ParamViews<SUR_ENC_NPARAM> v;
    stage_weights<SUR_ENC_NPARAM>(p.w, p.size, L.end, v);
    const int psize = psize_of<SUR_ENC_NPARAM>(p.size);
    float* row = p.partial + (size_t)blockIdx.x * psize;
    float* gacc = grads_in_lds ? L.end + psize : row;
    setup_grads<SUR_ENC_NPARAM>(p.size, gacc, grads_in_lds != 0, v);
    const int nin = p.c[0] * p.n, nout = L.rb[2].cout * L.rb[2].hout;
    for (int m = blockIdx.x; m < m_total; m += gridDim.x) {
        lds_load(L.rb[0].in, x + (size_t)m * nin, nin);
#pragma unroll
        for (int b = 0; b < 3; ++b) rb_forward(L.rb[b], v.w + b * SUR_RB_NPARAM);
        lds_load(L.dA, dz + (size_t)m * nout, nout);
        float *dout = L.dA, *din = L.dB;
#pragma unroll
        for (int b = 2; b >= 0; --b) {
            rb_backward(L.rb[b], v.w + b * SUR_RB_NPARAM, v.g + b * SUR_RB_NPARAM, dout, din, L.g1, L.g2, L.g3, L.xh);
            float* t = dout;
            dout = din;
            din = t;
        }
        if (dx) lds_store(dx + (size_t)m * nin, dout, nin);
    }
    if (grads_in_lds) {
        for (int j = threadIdx.x; j < psize; j += blockDim.x) row[j] += gacc[j];
    }
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

__host__ __device__ inline int step_act_floats(const sur_chunk_params& p, bool backward) {
    const int s = p.cs * p.hq, n = 4 * p.hq;
    int total = p.ca * p.hq + 2 * s + 4 * s + 2 * s + 2 * p.cs * 2 * p.hq + 2 * p.c_mid * n + 4 * n;
    if (backward) total += 4 * s + s + 3 * step_max_act(p) + p.ca * p.hq + s + 2 * s + n;
    return total;
}

__device__ void step_layout(const sur_chunk_params& p, float* lds, bool backward, StepLayout& L) {
    const int s = p.cs * p.hq, n = 4 * p.hq;
    float* cur = lds;
    auto take = [&](int k) { float* r = cur; cur += k; return r; };
    L.n = n;
    L.x = take(p.ca * p.hq);
    L.h = take(s);
    L.c = take(s);
    L.gates = take(4 * s);
    L.cnew = take(s);
    L.hnew = take(s);
    L.p0 = take(p.cs * 2 * p.hq);
    L.a0 = take(p.cs * 2 * p.hq);
    L.p1 = take(p.c_mid * n);
    L.a1 = take(p.c_mid * n);
    L.p2 = take(n);
    L.a2 = take(n);
    L.d = take(n);
    L.outv = take(n);
    if (backward) {
        const int m = step_max_act(p);
        L.dgates = take(4 * s);
        L.dh = take(s);
        L.gA = take(m);
        L.gB = take(m);
        L.xh = take(m);
        L.dx = take(p.ca * p.hq);
        L.dhin = take(s);
        L.dh_carry = take(s);
        L.dc_carry = take(s);
        L.dout_carry = take(n);
    }
    L.end = cur;
}

// one rollout step on LDS-resident x, h, c: fills gates (activated), cnew, hnew, decoder activations, d
__device__ void step_forward_body(const sur_chunk_params& p, const StepLayout& L, const float* const* w) {
    const int s = p.cs * p.hq;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        conv_fwd<3>(L.x, p.ca, p.hq, w[SUR_ST_WXI + 3 * g], w[SUR_ST_BXI + 3 * g], p.cs, 1, 1, L.gates + g * s, false);
        conv_fwd<3>(L.h, p.cs, p.hq, w[SUR_ST_WHI + 3 * g], nullptr, p.cs, 1, 1, L.gates + g * s, true);
    }
    for (int i = threadIdx.x; i < s; i += blockDim.x) {
        const float gi = sigmoid_(L.gates[i]), gf = sigmoid_(L.gates[s + i]), gg = tanhf(L.gates[2 * s + i]),
                    go = sigmoid_(L.gates[3 * s + i]);
        L.gates[i] = gi;
        L.gates[s + i] = gf;
        L.gates[2 * s + i] = gg;
        L.gates[3 * s + i] = go;
        const float cn = fmaf(gf, L.c[i], gi * gg);
        L.cnew[i] = cn;
        L.hnew[i] = go * tanhf(cn);
    }
    __syncthreads();
    deconv_fwd(L.hnew, p.cs, p.hq, w[SUR_ST_DC0_W], w[SUR_ST_DC0_B], p.cs, L.p0);
    act_ln_fwd(L.p0, p.cs, 2 * p.hq, w[SUR_ST_LN0_W], w[SUR_ST_LN0_B], true, L.a0);
    deconv_fwd(L.a0, p.cs, 2 * p.hq, w[SUR_ST_DC1_W], w[SUR_ST_DC1_B], p.c_mid, L.p1);
    act_ln_fwd(L.p1, p.c_mid, L.n, w[SUR_ST_LN1_W], w[SUR_ST_LN1_B], true, L.a1);
    conv_fwd<7>(L.a1, p.c_mid, L.n, w[SUR_ST_CV2_W], w[SUR_ST_CV2_B], 1, 1, 3, L.p2, false);
    act_ln_fwd(L.p2, 1, L.n, w[SUR_ST_LN2_W], w[SUR_ST_LN2_B], true, L.a2);
    conv_fwd<5>(L.a2, 1, L.n, w[SUR_ST_CV3_W], w[SUR_ST_CV3_B], 1, 1, 2, L.d, false);
}

__global__ void __launch_bounds__(TPB)
chunk_fwd_kernel(const sur_chunk_params p, const float* __restrict__ xlat_t, const float* __restrict__ lstates_t,
                 const float* __restrict__ states_t, const float* __restrict__ h0, const float* __restrict__ c0, int K,
                 int S, int B, float* __restrict__ h_all, float* __restrict__ c_all, float* __restrict__ d_all,
                 float* __restrict__ out_all) {
    extern __shared__ __align__(16) float lds[];
    StepLayout L;
    step_layout(p, lds, false, L);
    ParamViews<SUR_ST_NPARAM> v;
    stage_weights<SUR_ST_NPARAM>(p.w, p.size, L.end, v);
    const int b = blockIdx.x, s = p.cs * p.hq, nx = p.ca * p.hq, n = L.n;
    for (int i = threadIdx.x; i < s; i += blockDim.x) {
        L.hnew[i] = h0[(size_t)b * s + i];  // "previous" hidden state
        L.cnew[i] = c0[(size_t)b * s + i];
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        const size_t kb = (size_t)k * B + b;
        for (int i = threadIdx.x; i < nx; i += blockDim.x) L.x[i] = xlat_t[kb * nx + i];
        for (int i = threadIdx.x; i < s; i += blockDim.x) {
            L.h[i] = (k < S) ? lstates_t[kb * s + i] : L.hnew[i];  // teacher forcing replaces H
            L.c[i] = L.cnew[i];
        }
        // base of the integration: the given state while teacher forcing, else the previous output
        if (k < S)
            for (int i = threadIdx.x; i < n; i += blockDim.x) L.outv[i] = states_t[kb * n + i];
        __syncthreads();
        step_forward_body(p, L, v.w);
        for (int i = threadIdx.x; i < s; i += blockDim.x) {
            h_all[kb * s + i] = L.hnew[i];
            c_all[kb * s + i] = L.cnew[i];
        }
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float d = L.d[i];
            const float o = L.outv[i] + p.delta * fmaf(d, p.mul, p.add);
            d_all[kb * n + i] = d;
            out_all[kb * n + i] = o;
            L.outv[i] = o;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(TPB)
chunk_bwd_kernel(const sur_chunk_params p, const float* __restrict__ xlat_t, const float* __restrict__ lstates_t,
                 const float* __restrict__ h0, const float* __restrict__ c0, const float* __restrict__ h_all,
                 const float* __restrict__ c_all, const float* __restrict__ dd_all, const float* __restrict__ dout_all,
                 const float* __restrict__ dh_all, const float* __restrict__ dc_all, int K, int S, int B,
                 float* __restrict__ dxlat_t, float* __restrict__ dlstates_t, float* __restrict__ dh0,
                 float* __restrict__ dc0, int grads_in_lds) {
    extern __shared__ __align__(16) float lds[];
    StepLayout L;
    step_layout(p, lds, true, L);
    ParamViews<SUR_ST_NPARAM> v;
    stage_weights<SUR_ST_NPARAM>(p.w, p.size, L.end, v);
    const int psize = psize_of<SUR_ST_NPARAM>(p.size);
    float* row = p.partial + (size_t)blockIdx.x * psize;
    float* gacc = grads_in_lds ? L.end + psize : row;
    setup_grads<SUR_ST_NPARAM>(p.size, gacc, grads_in_lds != 0, v);
    const float* const* w = v.w;
    float* const* g = v.g;

    const int b = blockIdx.x, s = p.cs * p.hq, nx = p.ca * p.hq, n = L.n;
    for (int i = threadIdx.x; i < s; i += blockDim.x) L.dh_carry[i] = L.dc_carry[i] = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) L.dout_carry[i] = 0.0f;
    __syncthreads();

    for (int k = K - 1; k >= 0; --k) {
        const size_t kb = (size_t)k * B + b;
        // ---- reload this step's inputs and recompute its forward intermediates ----
        for (int i = threadIdx.x; i < nx; i += blockDim.x) L.x[i] = xlat_t[kb * nx + i];
        for (int i = threadIdx.x; i < s; i += blockDim.x) {
            const size_t prev = ((size_t)(k - 1) * B + b) * s + i;
            L.h[i] = (k < S) ? lstates_t[kb * s + i] : (k > 0 ? h_all[prev] : h0[(size_t)b * s + i]);
            L.c[i] = (k > 0) ? c_all[prev] : c0[(size_t)b * s + i];
        }
        __syncthreads();
        step_forward_body(p, L, w);

        // ---- total gradient wrt d_k: direct + through out_k = base + delta*(d*mul + add) ----
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float go = (dout_all ? dout_all[kb * n + i] : 0.0f) + L.dout_carry[i];
            L.gA[i] = fmaf(p.delta * p.mul, go, dd_all ? dd_all[kb * n + i] : 0.0f);
            // out_{k-1} is the base of step k only while free running
            L.dout_carry[i] = (k >= S) ? go : 0.0f;
        }
        __syncthreads();
        // ---- decoder backward ----
        conv_bwd_weight<5>(L.gA, 1, L.a2, 1, n, 1, 2, g[SUR_ST_CV3_W], g[SUR_ST_CV3_B]);
        conv_bwd_data<5>(L.gA, 1, n, w[SUR_ST_CV3_W], 1, 1, 2, L.gB, false);
        act_ln_bwd(L.gB, L.p2, 1, n, w[SUR_ST_LN2_W], true, L.gA, L.xh, g[SUR_ST_LN2_W], g[SUR_ST_LN2_B]);
        conv_bwd_weight<7>(L.gA, 1, L.a1, p.c_mid, n, 1, 3, g[SUR_ST_CV2_W], g[SUR_ST_CV2_B]);
        conv_bwd_data<7>(L.gA, 1, n, w[SUR_ST_CV2_W], p.c_mid, 1, 3, L.gB, false);
        act_ln_bwd(L.gB, L.p1, p.c_mid, n, w[SUR_ST_LN1_W], true, L.gA, L.xh, g[SUR_ST_LN1_W], g[SUR_ST_LN1_B]);
        deconv_bwd_weight(L.gA, p.c_mid, L.a0, p.cs, 2 * p.hq, g[SUR_ST_DC1_W], g[SUR_ST_DC1_B]);
        deconv_bwd_data(L.gA, p.c_mid, 2 * p.hq, w[SUR_ST_DC1_W], p.cs, L.gB);
        act_ln_bwd(L.gB, L.p0, p.cs, 2 * p.hq, w[SUR_ST_LN0_W], true, L.gA, L.xh, g[SUR_ST_LN0_W], g[SUR_ST_LN0_B]);
        deconv_bwd_weight(L.gA, p.cs, L.hnew, p.cs, p.hq, g[SUR_ST_DC0_W], g[SUR_ST_DC0_B]);
        deconv_bwd_data(L.gA, p.cs, p.hq, w[SUR_ST_DC0_W], p.cs, L.dh);

        // ---- cell backward ----
        for (int i = threadIdx.x; i < s; i += blockDim.x) {
            const float dhn = L.dh[i] + L.dh_carry[i] + (dh_all ? dh_all[kb * s + i] : 0.0f);
            const float gi = L.gates[i], gf = L.gates[s + i], gg = L.gates[2 * s + i], go = L.gates[3 * s + i];
            const float tc = tanhf(L.cnew[i]);
            const float dcn = L.dc_carry[i] + (dc_all ? dc_all[kb * s + i] : 0.0f) + dhn * go * (1.0f - tc * tc);
            L.dgates[i] = dcn * gg * gi * (1.0f - gi);
            L.dgates[s + i] = dcn * L.c[i] * gf * (1.0f - gf);
            L.dgates[2 * s + i] = dcn * gi * (1.0f - gg * gg);
            L.dgates[3 * s + i] = dhn * tc * go * (1.0f - go);
            L.dc_carry[i] = dcn * gf;  // gradient wrt c_{k-1}
        }
        __syncthreads();
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) {
            const float* dg = L.dgates + gt * s;
            conv_bwd_weight<3>(dg, p.cs, L.x, p.ca, p.hq, 1, 1, g[SUR_ST_WXI + 3 * gt], g[SUR_ST_BXI + 3 * gt]);
            conv_bwd_weight<3>(dg, p.cs, L.h, p.cs, p.hq, 1, 1, g[SUR_ST_WHI + 3 * gt], nullptr);
            conv_bwd_data<3>(dg, p.cs, p.hq, w[SUR_ST_WXI + 3 * gt], p.ca, 1, 1, L.dx, gt > 0);
            conv_bwd_data<3>(dg, p.cs, p.hq, w[SUR_ST_WHI + 3 * gt], p.cs, 1, 1, L.dhin, gt > 0);
        }
        if (dxlat_t)
            for (int i = threadIdx.x; i < nx; i += blockDim.x) dxlat_t[kb * nx + i] = L.dx[i];
        for (int i = threadIdx.x; i < s; i += blockDim.x) {
            const float v_ = L.dhin[i];
            if (k < S) {  // h_in was the encoded given state: gradient goes to the state encoder
                if (dlstates_t) dlstates_t[kb * s + i] = v_;
                L.dh_carry[i] = 0.0f;
            } else {      // h_in was h_{k-1}
                L.dh_carry[i] = v_;
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < s; i += blockDim.x) {
        if (dh0) dh0[(size_t)b * s + i] = L.dh_carry[i];  // non-zero only if step 0 was free running (S == 0)
        if (dc0) dc0[(size_t)b * s + i] = L.dc_carry[i];
    }
    if (grads_in_lds) {
        __syncthreads();
        for (int j = threadIdx.x; j < psize; j += blockDim.x) row[j] += gacc[j];
    }
}

// g[i][j] += sum_r partial[r][off_i + j]; the partial rows are re-zeroed
template <int NP, typename Params>
__global__ void __launch_bounds__(TPB) flush_grads_kernel(const Params p, int psize) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= psize) return;
    float acc = 0.0f;
    for (int r = 0; r < p.rows; ++r) {
        float* q = p.partial + (size_t)r * psize + t;
        acc += *q;
        *q = 0.0f;
    }
    int off = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (t >= off && t < off + p.size[i]) p.g[i][t - off] += acc;
        off += p.size[i];
    }
}

template <typename F>
int launch_checked(F&& f, const char* what) {
    f();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "%s launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

constexpr size_t LDS_LIMIT = 160 * 1024;

template <typename K>
int set_lds(K kernel, size_t bytes, const char* what) {
    if (bytes > LDS_LIMIT) return fail(-4, "%s needs %zu B of LDS (> 160 KiB): N too large for the fused path", what, bytes);
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return 0;
}

}  // namespace

extern "C" {

const char* sur_last_error(void) { return g_err; }

int sur_encoder_forward(void* stream, const sur_encoder_params* p, const float* x, int m, float* z) {
    if (!p || !x || !z || m <= 0) return fail(-1, "sur_encoder_forward: bad argument");
    const int psize = psize_of<SUR_ENC_NPARAM>(p->size);
    const size_t lds = sizeof(float) * (enc_act_floats(*p, false) + psize);
    if (int rc = set_lds(enc_fwd_kernel, lds, "encoder forward")) return rc;
    const int grid = m < 1024 ? m : 1024;
    return launch_checked([&] { hipLaunchKernelGGL(enc_fwd_kernel, dim3(grid), dim3(TPB), lds, (hipStream_t)stream, *p, x, m, z); },
                          "enc_fwd");
}

int sur_encoder_backward(void* stream, const sur_encoder_params* p, const float* x, const float* dz, int m, float* dx) {
    if (!p || !x || !dz || m <= 0) return fail(-1, "sur_encoder_backward: bad argument");
    if (!p->partial || p->rows <= 0) return fail(-1, "sur_encoder_backward: no partial gradient buffer");
    const int psize = psize_of<SUR_ENC_NPARAM>(p->size);
    const size_t base = sizeof(float) * (enc_act_floats(*p, true) + psize);
    int grads_in_lds = (base + sizeof(float) * psize <= LDS_LIMIT) ? 1 : 0;
    const size_t lds = base + (grads_in_lds ? sizeof(float) * psize : 0);
    if (int rc = set_lds(enc_bwd_kernel, lds, "encoder backward")) return rc;
    const int grid = m < p->rows ? m : p->rows;
    return launch_checked([&] {
        hipLaunchKernelGGL(enc_bwd_kernel, dim3(grid), dim3(TPB), lds, (hipStream_t)stream, *p, x, dz, m, dx, grads_in_lds);
    }, "enc_bwd");
}

int sur_flush_encoder_grads(void* stream, const sur_encoder_params* p) {
    if (!p || !p->partial) return fail(-1, "sur_flush_encoder_grads: bad argument");
    for (int i = 0; i < SUR_ENC_NPARAM; ++i)
        if (!p->g[i]) return fail(-1, "sur_flush_encoder_grads: gradient tensor %d is NULL", i);
    const int psize = psize_of<SUR_ENC_NPARAM>(p->size);
    return launch_checked([&] {
        hipLaunchKernelGGL((flush_grads_kernel<SUR_ENC_NPARAM, sur_encoder_params>), dim3((psize + TPB - 1) / TPB), dim3(TPB), 0,
                           (hipStream_t)stream, *p, psize);
    }, "flush_enc");
}

int sur_chunk_forward(void* stream, const sur_chunk_params* p, const float* xlat_t, const float* lstates_t,
                      const float* states_t, const float* h0, const float* c0, int k, int s, int b, float* h_all,
                      float* c_all, float* d_all, float* out_all) {
    if (!p || !xlat_t || !h0 || !c0 || !h_all || !c_all || !d_all || !out_all || k <= 0 || b <= 0 || s < 1 ||
        !lstates_t || !states_t)
        return fail(-1, "sur_chunk_forward: bad argument (need K > 0, B > 0, S >= 1)");
    const int psize = psize_of<SUR_ST_NPARAM>(p->size);
    const size_t lds = sizeof(float) * (step_act_floats(*p, false) + psize);
    if (int rc = set_lds(chunk_fwd_kernel, lds, "chunk forward")) return rc;
    return launch_checked([&] {
        hipLaunchKernelGGL(chunk_fwd_kernel, dim3(b), dim3(TPB), lds, (hipStream_t)stream, *p, xlat_t, lstates_t, states_t, h0,
                           c0, k, s, b, h_all, c_all, d_all, out_all);
    }, "chunk_fwd");
}

int sur_chunk_backward(void* stream, const sur_chunk_params* p, const float* xlat_t, const float* lstates_t,
                       const float* h0, const float* c0, const float* h_all, const float* c_all, const float* dd_all,
                       const float* dout_all, const float* dh_all, const float* dc_all, int k, int s, int b,
                       float* dxlat_t, float* dlstates_t, float* dh0, float* dc0) {
    if (!p || !xlat_t || !lstates_t || !h0 || !c0 || !h_all || !c_all || k <= 0 || b <= 0 || s < 1)
        return fail(-1, "sur_chunk_backward: bad argument");
    if (!p->partial || p->rows < b) return fail(-1, "sur_chunk_backward: partial gradient buffer has %d rows, need %d", p->rows, b);
    const int psize = psize_of<SUR_ST_NPARAM>(p->size);
    const size_t base = sizeof(float) * (step_act_floats(*p, true) + psize);
    int grads_in_lds = (base + sizeof(float) * psize <= LDS_LIMIT) ? 1 : 0;
    const size_t lds = base + (grads_in_lds ? sizeof(float) * psize : 0);
    if (int rc = set_lds(chunk_bwd_kernel, lds, "chunk backward")) return rc;
    return launch_checked([&] {
        hipLaunchKernelGGL(chunk_bwd_kernel, dim3(b), dim3(TPB), lds, (hipStream_t)stream, *p, xlat_t, lstates_t, h0, c0, h_all,
                           c_all, dd_all, dout_all, dh_all, dc_all, k, s, b, dxlat_t, dlstates_t, dh0, dc0, grads_in_lds);
    }, "chunk_bwd");
}

int sur_flush_chunk_grads(void* stream, const sur_chunk_params* p) {
    if (!p || !p->partial) return fail(-1, "sur_flush_chunk_grads: bad argument");
    for (int i = 0; i < SUR_ST_NPARAM; ++i)
        if (!p->g[i]) return fail(-1, "sur_flush_chunk_grads: gradient tensor %d is NULL", i);
    const int psize = psize_of<SUR_ST_NPARAM>(p->size);
    return launch_checked([&] {
        hipLaunchKernelGGL((flush_grads_kernel<SUR_ST_NPARAM, sur_chunk_params>), dim3((psize + TPB - 1) / TPB), dim3(TPB), 0,
                           (hipStream_t)stream, *p, psize);
    }, "flush_chunk");
}

}  // extern "C"
