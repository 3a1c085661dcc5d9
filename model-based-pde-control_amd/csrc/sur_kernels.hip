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
// then back-propagate; parameter gradients are reduced over the spatial axis per workgroup and added
// to the caller's fp32 gradient buffers with one atomic per parameter element per sample.
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

// circular Conv1d: out[o][p] (+)= bias[o] + sum_{ci,k} W[o][ci][k] * in[ci][(p*stride + k - pad) mod hin]
__device__ void conv_fwd(const float* in, int cin, int hin, const float* __restrict__ W,
                         const float* __restrict__ bias, int cout, int K, int stride, int pad, float* out,
                         bool accumulate) {
    const int hout = hin / stride;
    for (int idx = threadIdx.x; idx < cout * hout; idx += blockDim.x) {
        const int o = idx / hout, p = idx - o * hout;
        float acc = bias ? bias[o] : 0.0f;
        const float* w = W + (size_t)o * cin * K;
        for (int ci = 0; ci < cin; ++ci) {
            const float* row = in + ci * hin;
            for (int k = 0; k < K; ++k) acc = fmaf(w[ci * K + k], row[wrapi(p * stride + k - pad, hin)], acc);
        }
        out[idx] = accumulate ? out[idx] + acc : acc;
    }
    __syncthreads();
}

// din[ci][j] (+)= sum_{o,k : (p*stride + k - pad) mod hin == j} W[o][ci][k] * dout[o][p]
__device__ void conv_bwd_data(const float* dout, int cout, int hin, const float* __restrict__ W, int cin, int K,
                              int stride, int pad, float* din, bool accumulate) {
    const int hout = hin / stride;
    for (int idx = threadIdx.x; idx < cin * hin; idx += blockDim.x) {
        const int ci = idx / hin, j = idx - ci * hin;
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) {
            const int t = wrapi(j - k + pad, hin);
            if (t % stride) continue;
            const int p = t / stride;
            for (int o = 0; o < cout; ++o) acc = fmaf(W[((size_t)o * cin + ci) * K + k], dout[o * hout + p], acc);
        }
        din[idx] = accumulate ? din[idx] + acc : acc;
    }
    __syncthreads();
}

// gW[o][ci][k] += sum_p dout[o][p] * in[ci][(p*stride + k - pad) mod hin];  gb[o] += sum_p dout[o][p]
__device__ void conv_bwd_weight(const float* dout, int cout, const float* in, int cin, int hin, int K, int stride,
                                int pad, float* __restrict__ gW, float* __restrict__ gb) {
    const int hout = hin / stride;
    for (int idx = threadIdx.x; idx < cout * cin * K; idx += blockDim.x) {
        const int o = idx / (cin * K), r = idx - o * cin * K, ci = r / K, k = r - ci * K;
        const float* d = dout + o * hout;
        const float* row = in + ci * hin;
        float acc = 0.0f;
        for (int p = 0; p < hout; ++p) acc = fmaf(d[p], row[wrapi(p * stride + k - pad, hin)], acc);
        atomicAdd(gW + idx, acc);
    }
    if (gb) {
        for (int o = threadIdx.x; o < cout; o += blockDim.x) {
            float acc = 0.0f;
            for (int p = 0; p < hout; ++p) acc += dout[o * hout + p];
            atomicAdd(gb + o, acc);
        }
    }
    __syncthreads();
}

// ConvTranspose1d(k=3, stride=2, padding=1, output_padding=1), zero padded; W[ci][o][k]; hout = 2*hin
// out[o][j] = b[o] + sum_{ci,k : j = 2i - 1 + k} W[ci][o][k] * in[ci][i]
__device__ void deconv_fwd(const float* in, int cin, int hin, const float* __restrict__ W,
                           const float* __restrict__ bias, int cout, float* out) {
    const int hout = 2 * hin;
    for (int idx = threadIdx.x; idx < cout * hout; idx += blockDim.x) {
        const int o = idx / hout, j = idx - o * hout;
        float acc = bias[o];
        for (int k = 0; k < 3; ++k) {
            const int t = j + 1 - k;
            if (t < 0 || (t & 1)) continue;
            const int i = t >> 1;
            if (i >= hin) continue;
            for (int ci = 0; ci < cin; ++ci) acc = fmaf(W[((size_t)ci * cout + o) * 3 + k], in[ci * hin + i], acc);
        }
        out[idx] = acc;
    }
    __syncthreads();
}

__device__ void deconv_bwd_data(const float* dout, int cout, int hin, const float* __restrict__ W, int cin,
                                float* din) {
    const int hout = 2 * hin;
    for (int idx = threadIdx.x; idx < cin * hin; idx += blockDim.x) {
        const int ci = idx / hin, i = idx - ci * hin;
        float acc = 0.0f;
        for (int k = 0; k < 3; ++k) {
            const int j = 2 * i - 1 + k;
            if (j < 0 || j >= hout) continue;
            for (int o = 0; o < cout; ++o) acc = fmaf(W[((size_t)ci * cout + o) * 3 + k], dout[o * hout + j], acc);
        }
        din[idx] = acc;
    }
    __syncthreads();
}

__device__ void deconv_bwd_weight(const float* dout, int cout, const float* in, int cin, int hin,
                                  float* __restrict__ gW, float* __restrict__ gb) {
    const int hout = 2 * hin;
    for (int idx = threadIdx.x; idx < cin * cout * 3; idx += blockDim.x) {
        const int ci = idx / (cout * 3), r = idx - ci * cout * 3, o = r / 3, k = r - o * 3;
        float acc = 0.0f;
        for (int i = 0; i < hin; ++i) {
            const int j = 2 * i - 1 + k;
            if (j < 0 || j >= hout) continue;
            acc = fmaf(in[ci * hin + i], dout[o * hout + j], acc);
        }
        atomicAdd(gW + idx, acc);
    }
    for (int o = threadIdx.x; o < cout; o += blockDim.x) {
        float acc = 0.0f;
        for (int j = 0; j < hout; ++j) acc += dout[o * hout + j];
        atomicAdd(gb + o, acc);
    }
    __syncthreads();
}

// out = LayerNorm_H(act(pre)) * gamma[p] + beta[p], act = SiLU or identity; one wave per channel
__device__ void act_ln_fwd(const float* pre, int C, int H, const float* __restrict__ gamma,
                           const float* __restrict__ beta, bool silu, float* out) {
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
__device__ void act_ln_bwd(const float* dout, const float* pre, int C, int H, const float* __restrict__ gamma,
                           bool silu, float* dpre, float* xhat_scratch, float* __restrict__ ggamma,
                           float* __restrict__ gbeta) {
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
        atomicAdd(ggamma + p, gg);
        atomicAdd(gbeta + p, gb);
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
// encoder: three residual blocks
// ---------------------------------------------------------------------------------------------
struct RBBuf {  // LDS pointers of one block's forward intermediates
    float *in, *skip, *a1pre, *a1, *a2pre, *a2, *s, *out;
    int cin, cout, hin, hout, stride;
};

__device__ void rb_forward(const RBBuf& b, const float* const* w) {
    conv_fwd(b.in, b.cin, b.hin, w[SUR_RB_SKIP], nullptr, b.cout, 1, b.stride, 0, b.skip, false);
    conv_fwd(b.in, b.cin, b.hin, w[SUR_RB_CONV1], nullptr, b.cout, 3, b.stride, 1, b.a1pre, false);
    act_ln_fwd(b.a1pre, b.cout, b.hout, w[SUR_RB_LN1_W], w[SUR_RB_LN1_B], true, b.a1);
    conv_fwd(b.a1, b.cout, b.hout, w[SUR_RB_CONV2], nullptr, b.cout, 3, 1, 1, b.a2pre, false);
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
    conv_bwd_weight(g1, b.cout, b.in, b.cin, b.hin, 1, b.stride, 0, g[SUR_RB_SKIP], nullptr);
    conv_bwd_data(g1, b.cout, b.hin, w[SUR_RB_SKIP], b.cin, 1, b.stride, 0, din, false);
    // residual path
    act_ln_bwd(g1, b.a2pre, b.cout, b.hout, w[SUR_RB_LN2_W], true, g2, xh, g[SUR_RB_LN2_W], g[SUR_RB_LN2_B]);
    conv_bwd_weight(g2, b.cout, b.a1, b.cout, b.hout, 3, 1, 1, g[SUR_RB_CONV2], nullptr);
    conv_bwd_data(g2, b.cout, b.hout, w[SUR_RB_CONV2], b.cout, 3, 1, 1, g3, false);
    act_ln_bwd(g3, b.a1pre, b.cout, b.hout, w[SUR_RB_LN1_W], true, g1, xh, g[SUR_RB_LN1_W], g[SUR_RB_LN1_B]);
    conv_bwd_weight(g1, b.cout, b.in, b.cin, b.hin, 3, b.stride, 1, g[SUR_RB_CONV1], nullptr);
    conv_bwd_data(g1, b.cout, b.hin, w[SUR_RB_CONV1], b.cin, 3, b.stride, 1, din, true);
}

struct EncLayout {
    RBBuf rb[3];
    float *g1, *g2, *g3, *xh, *dA, *dB;
    int total;  // floats
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

__host__ __device__ inline int enc_lds_floats(const sur_encoder_params& p, bool backward) {
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
    }
}

__global__ void __launch_bounds__(TPB) enc_fwd_kernel(const sur_encoder_params p, const float* __restrict__ x,
                                                      float* __restrict__ z) {
    extern __shared__ __align__(16) float lds[];
    EncLayout L;
    enc_layout(p, lds, false, L);
    const int m = blockIdx.x;
    lds_load(L.rb[0].in, x + (size_t)m * p.c[0] * p.n, p.c[0] * p.n);
#pragma unroll
    for (int b = 0; b < 3; ++b) rb_forward(L.rb[b], p.w + b * SUR_RB_NPARAM);
    const int nout = L.rb[2].cout * L.rb[2].hout;
    lds_store(z + (size_t)m * nout, L.rb[2].out, nout);
}

__global__ void __launch_bounds__(TPB) enc_bwd_kernel(const sur_encoder_params p, const float* __restrict__ x,
                                                      const float* __restrict__ dz, float* __restrict__ dx) {
    extern __shared__ __align__(16) float lds[];
    EncLayout L;
    enc_layout(p, lds, true, L);
    const int m = blockIdx.x;
    lds_load(L.rb[0].in, x + (size_t)m * p.c[0] * p.n, p.c[0] * p.n);
#pragma unroll
    for (int b = 0; b < 3; ++b) rb_forward(L.rb[b], p.w + b * SUR_RB_NPARAM);
    const int nout = L.rb[2].cout * L.rb[2].hout;
    lds_load(L.dA, dz + (size_t)m * nout, nout);
    float *dout = L.dA, *din = L.dB;
#pragma unroll
    for (int b = 2; b >= 0; --b) {
        rb_backward(L.rb[b], p.w + b * SUR_RB_NPARAM, p.g + b * SUR_RB_NPARAM, dout, din, L.g1, L.g2, L.g3, L.xh);
        float* t = dout;
        dout = din;
        din = t;
    }
    if (dx) lds_store(dx + (size_t)m * p.c[0] * p.n, dout, p.c[0] * p.n);
}

// ---------------------------------------------------------------------------------------------
// rollout step: ConvLSTM cell + decoder + integration
// ---------------------------------------------------------------------------------------------
struct StepLayout {
    float *x, *h, *c, *gates, *cnew, *hnew, *p0, *a0, *p1, *a1, *p2, *a2, *d;
    // backward only
    float *dgates, *dh, *gA, *gB, *gC, *xh, *dx, *dhin;
    int n;  // N = 4*hq
};

__host__ __device__ inline int step_max_act(const sur_step_params& p) {
    const int a = p.cs * 2 * p.hq, b = p.c_mid * 4 * p.hq;
    return a > b ? a : b;
}

__host__ __device__ inline int step_lds_floats(const sur_step_params& p, bool backward) {
    const int s = p.cs * p.hq, n = 4 * p.hq;
    int total = p.ca * p.hq + 2 * s + 4 * s + 2 * s + 2 * p.cs * 2 * p.hq + 2 * p.c_mid * n + 3 * n;
    if (backward) total += 4 * s + s + 4 * step_max_act(p) + p.ca * p.hq + s;
    return total;
}

__device__ void step_layout(const sur_step_params& p, float* lds, bool backward, StepLayout& L) {
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
    if (backward) {
        const int m = step_max_act(p);
        L.dgates = take(4 * s);
        L.dh = take(s);
        L.gA = take(m);
        L.gB = take(m);
        L.gC = take(m);
        L.xh = take(m);
        L.dx = take(p.ca * p.hq);
        L.dhin = take(s);
    }
}

__device__ void step_forward_body(const sur_step_params& p, const StepLayout& L) {
    const int s = p.cs * p.hq;
    // gate pre-activations: Wx*x + b + Wh*h  (k = 3, circular)
    for (int g = 0; g < 4; ++g) {
        conv_fwd(L.x, p.ca, p.hq, p.w[SUR_ST_WXI + 3 * g], p.w[SUR_ST_BXI + 3 * g], p.cs, 3, 1, 1, L.gates + g * s, false);
        conv_fwd(L.h, p.cs, p.hq, p.w[SUR_ST_WHI + 3 * g], nullptr, p.cs, 3, 1, 1, L.gates + g * s, true);
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
    // decoder
    deconv_fwd(L.hnew, p.cs, p.hq, p.w[SUR_ST_DC0_W], p.w[SUR_ST_DC0_B], p.cs, L.p0);
    act_ln_fwd(L.p0, p.cs, 2 * p.hq, p.w[SUR_ST_LN0_W], p.w[SUR_ST_LN0_B], true, L.a0);
    deconv_fwd(L.a0, p.cs, 2 * p.hq, p.w[SUR_ST_DC1_W], p.w[SUR_ST_DC1_B], p.c_mid, L.p1);
    act_ln_fwd(L.p1, p.c_mid, L.n, p.w[SUR_ST_LN1_W], p.w[SUR_ST_LN1_B], true, L.a1);
    conv_fwd(L.a1, p.c_mid, L.n, p.w[SUR_ST_CV2_W], p.w[SUR_ST_CV2_B], 1, 7, 1, 3, L.p2, false);
    act_ln_fwd(L.p2, 1, L.n, p.w[SUR_ST_LN2_W], p.w[SUR_ST_LN2_B], true, L.a2);
    conv_fwd(L.a2, 1, L.n, p.w[SUR_ST_CV3_W], p.w[SUR_ST_CV3_B], 1, 5, 1, 2, L.d, false);
}

__global__ void __launch_bounds__(TPB)
step_fwd_kernel(const sur_step_params p, const float* __restrict__ xlat, const float* __restrict__ h_in,
                const float* __restrict__ c_prev, const float* __restrict__ base, float* __restrict__ h_out,
                float* __restrict__ c_out, float* __restrict__ d_out, float* __restrict__ out) {
    extern __shared__ __align__(16) float lds[];
    StepLayout L;
    step_layout(p, lds, false, L);
    const int b = blockIdx.x, s = p.cs * p.hq, nx = p.ca * p.hq;
    for (int i = threadIdx.x; i < nx; i += blockDim.x) L.x[i] = xlat[(size_t)b * nx + i];
    for (int i = threadIdx.x; i < s; i += blockDim.x) {
        L.h[i] = h_in[(size_t)b * s + i];
        L.c[i] = c_prev[(size_t)b * s + i];
    }
    __syncthreads();
    step_forward_body(p, L);
    for (int i = threadIdx.x; i < s; i += blockDim.x) {
        h_out[(size_t)b * s + i] = L.hnew[i];
        c_out[(size_t)b * s + i] = L.cnew[i];
    }
    for (int i = threadIdx.x; i < L.n; i += blockDim.x) {
        const float d = L.d[i];
        d_out[(size_t)b * L.n + i] = d;
        out[(size_t)b * L.n + i] = base[(size_t)b * L.n + i] + p.delta * fmaf(d, p.mul, p.add);
    }
}

__global__ void __launch_bounds__(TPB)
step_bwd_kernel(const sur_step_params p, const float* __restrict__ xlat, const float* __restrict__ h_in,
                const float* __restrict__ c_prev, const float* __restrict__ dd, const float* __restrict__ dout,
                const float* __restrict__ dh, const float* __restrict__ dc, float* __restrict__ dxlat,
                float* __restrict__ dh_in, float* __restrict__ dc_prev, float* __restrict__ dbase) {
    extern __shared__ __align__(16) float lds[];
    StepLayout L;
    step_layout(p, lds, true, L);
    const int b = blockIdx.x, s = p.cs * p.hq, nx = p.ca * p.hq, n = L.n;
    for (int i = threadIdx.x; i < nx; i += blockDim.x) L.x[i] = xlat[(size_t)b * nx + i];
    for (int i = threadIdx.x; i < s; i += blockDim.x) {
        L.h[i] = h_in[(size_t)b * s + i];
        L.c[i] = c_prev[(size_t)b * s + i];
    }
    __syncthreads();
    step_forward_body(p, L);  // recompute the intermediates

    // total gradient wrt the decoded delta d: direct + through out = base + delta*(d*mul + add)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float g = dd ? dd[(size_t)b * n + i] : 0.0f;
        if (dout) {
            const float go = dout[(size_t)b * n + i];
            g = fmaf(p.delta * p.mul, go, g);
            if (dbase) dbase[(size_t)b * n + i] = go;
        } else if (dbase) {
            dbase[(size_t)b * n + i] = 0.0f;
        }
        L.gA[i] = g;
    }
    __syncthreads();
    // ---- decoder backward ----
    conv_bwd_weight(L.gA, 1, L.a2, 1, n, 5, 1, 2, p.g[SUR_ST_CV3_W], p.g[SUR_ST_CV3_B]);
    conv_bwd_data(L.gA, 1, n, p.w[SUR_ST_CV3_W], 1, 5, 1, 2, L.gB, false);                       // d a2
    act_ln_bwd(L.gB, L.p2, 1, n, p.w[SUR_ST_LN2_W], true, L.gA, L.xh, p.g[SUR_ST_LN2_W], p.g[SUR_ST_LN2_B]);  // d p2
    conv_bwd_weight(L.gA, 1, L.a1, p.c_mid, n, 7, 1, 3, p.g[SUR_ST_CV2_W], p.g[SUR_ST_CV2_B]);
    conv_bwd_data(L.gA, 1, n, p.w[SUR_ST_CV2_W], p.c_mid, 7, 1, 3, L.gB, false);                 // d a1
    act_ln_bwd(L.gB, L.p1, p.c_mid, n, p.w[SUR_ST_LN1_W], true, L.gA, L.xh, p.g[SUR_ST_LN1_W], p.g[SUR_ST_LN1_B]);  // d p1
    deconv_bwd_weight(L.gA, p.c_mid, L.a0, p.cs, 2 * p.hq, p.g[SUR_ST_DC1_W], p.g[SUR_ST_DC1_B]);
    deconv_bwd_data(L.gA, p.c_mid, 2 * p.hq, p.w[SUR_ST_DC1_W], p.cs, L.gB);                     // d a0
    act_ln_bwd(L.gB, L.p0, p.cs, 2 * p.hq, p.w[SUR_ST_LN0_W], true, L.gA, L.xh, p.g[SUR_ST_LN0_W], p.g[SUR_ST_LN0_B]);  // d p0
    deconv_bwd_weight(L.gA, p.cs, L.hnew, p.cs, p.hq, p.g[SUR_ST_DC0_W], p.g[SUR_ST_DC0_B]);
    deconv_bwd_data(L.gA, p.cs, p.hq, p.w[SUR_ST_DC0_W], p.cs, L.dh);                            // d hnew (decoder part)

    // ---- cell backward ----
    for (int i = threadIdx.x; i < s; i += blockDim.x) {
        const float dhn = L.dh[i] + (dh ? dh[(size_t)b * s + i] : 0.0f);
        const float gi = L.gates[i], gf = L.gates[s + i], gg = L.gates[2 * s + i], go = L.gates[3 * s + i];
        const float tc = tanhf(L.cnew[i]);
        const float dcn = (dc ? dc[(size_t)b * s + i] : 0.0f) + dhn * go * (1.0f - tc * tc);
        L.dgates[i] = dcn * gg * gi * (1.0f - gi);
        L.dgates[s + i] = dcn * L.c[i] * gf * (1.0f - gf);
        L.dgates[2 * s + i] = dcn * gi * (1.0f - gg * gg);
        L.dgates[3 * s + i] = dhn * tc * go * (1.0f - go);
        if (dc_prev) dc_prev[(size_t)b * s + i] = dcn * gf;
    }
    __syncthreads();
    for (int g = 0; g < 4; ++g) {
        const float* dg = L.dgates + g * s;
        conv_bwd_weight(dg, p.cs, L.x, p.ca, p.hq, 3, 1, 1, p.g[SUR_ST_WXI + 3 * g], p.g[SUR_ST_BXI + 3 * g]);
        conv_bwd_weight(dg, p.cs, L.h, p.cs, p.hq, 3, 1, 1, p.g[SUR_ST_WHI + 3 * g], nullptr);
        conv_bwd_data(dg, p.cs, p.hq, p.w[SUR_ST_WXI + 3 * g], p.ca, 3, 1, 1, L.dx, g > 0);
        conv_bwd_data(dg, p.cs, p.hq, p.w[SUR_ST_WHI + 3 * g], p.cs, 3, 1, 1, L.dhin, g > 0);
    }
    if (dxlat)
        for (int i = threadIdx.x; i < nx; i += blockDim.x) dxlat[(size_t)b * nx + i] = L.dx[i];
    if (dh_in)
        for (int i = threadIdx.x; i < s; i += blockDim.x) dh_in[(size_t)b * s + i] = L.dhin[i];
}

template <typename F>
int launch_checked(F&& f, const char* what) {
    f();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "%s launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

int check_lds(size_t bytes, const char* what) {
    if (bytes > 160 * 1024) return fail(-4, "%s needs %zu B of LDS (> 160 KiB): N too large for the fused path", what, bytes);
    return 0;
}

}  // namespace

extern "C" {

const char* sur_last_error(void) { return g_err; }

int sur_encoder_forward(void* stream, const sur_encoder_params* p, const float* x, int m, float* z) {
    if (!p || !x || !z || m <= 0) return fail(-1, "sur_encoder_forward: bad argument");
    const size_t lds = sizeof(float) * enc_lds_floats(*p, false);
    if (int rc = check_lds(lds, "encoder forward")) return rc;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)enc_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return launch_checked([&] { hipLaunchKernelGGL(enc_fwd_kernel, dim3(m), dim3(TPB), lds, (hipStream_t)stream, *p, x, z); },
                          "enc_fwd");
}

int sur_encoder_backward(void* stream, const sur_encoder_params* p, const float* x, const float* dz, int m, float* dx) {
    if (!p || !x || !dz || m <= 0) return fail(-1, "sur_encoder_backward: bad argument");
    for (int i = 0; i < 3 * SUR_RB_NPARAM; ++i)
        if (!p->g[i]) return fail(-1, "sur_encoder_backward: gradient buffer %d is NULL", i);
    const size_t lds = sizeof(float) * enc_lds_floats(*p, true);
    if (int rc = check_lds(lds, "encoder backward")) return rc;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)enc_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return launch_checked([&] { hipLaunchKernelGGL(enc_bwd_kernel, dim3(m), dim3(TPB), lds, (hipStream_t)stream, *p, x, dz, dx); },
                          "enc_bwd");
}

int sur_step_forward(void* stream, const sur_step_params* p, const float* xlat, const float* h_in, const float* c_prev,
                     const float* base, int b, float* h_out, float* c_out, float* d_out, float* out) {
    if (!p || !xlat || !h_in || !c_prev || !base || !h_out || !c_out || !d_out || !out || b <= 0)
        return fail(-1, "sur_step_forward: bad argument");
    const size_t lds = sizeof(float) * step_lds_floats(*p, false);
    if (int rc = check_lds(lds, "step forward")) return rc;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)step_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return launch_checked([&] {
        hipLaunchKernelGGL(step_fwd_kernel, dim3(b), dim3(TPB), lds, (hipStream_t)stream, *p, xlat, h_in, c_prev, base,
                           h_out, c_out, d_out, out);
    }, "step_fwd");
}

int sur_step_backward(void* stream, const sur_step_params* p, const float* xlat, const float* h_in, const float* c_prev,
                      const float* dd, const float* dout, const float* dh, const float* dc, int b, float* dxlat,
                      float* dh_in, float* dc_prev, float* dbase) {
    if (!p || !xlat || !h_in || !c_prev || b <= 0) return fail(-1, "sur_step_backward: bad argument");
    for (int i = 0; i < SUR_ST_NPARAM; ++i)
        if (!p->g[i]) return fail(-1, "sur_step_backward: gradient buffer %d is NULL", i);
    const size_t lds = sizeof(float) * step_lds_floats(*p, true);
    if (int rc = check_lds(lds, "step backward")) return rc;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)step_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return launch_checked([&] {
        hipLaunchKernelGGL(step_bwd_kernel, dim3(b), dim3(TPB), lds, (hipStream_t)stream, *p, xlat, h_in, c_prev, dd, dout,
                           dh, dc, dxlat, dh_in, dc_prev, dbase);
    }, "step_bwd");
}

}  // extern "C"
