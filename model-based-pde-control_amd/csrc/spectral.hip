// spectral.hip -- fused truncated-DFT spectral convolution for gfx950 (C ABI: include/spectral_hip.h).
//
// One workgroup (4 wavefronts) per sample.  LDS: the sample [C][N] (row stride N + 4 floats: the MFMA A-operand gather
// reads 16 rows at once), one cosine table cos(2 pi k / N), the two small spectra.  Three phases, one barrier each:
//   A  spectrum S[c][k] = sum_n in[c][n] * Bf[n][k],   k < 2M:  Bf[n][m] = cos(th), Bf[n][M + m] = -sin(th), th = 2 pi m n / N
//      (the truncated rfft as a [C x N] @ [N x 2M] GEMM on v_mfma_f32_16x16x4_f32; twiddles gathered from the table)
//   B  complex mode mixing with the weights (read through L2; they are shared by every workgroup):
//        forward :  Y[o][m] = s_m * sum_i S[i][m] * W[i][o][m]           s_0 = 1/N, s_m = 2/N (irfft's weights)
//        backward:  G = s (.) S is saved;  GX[i][m] = sum_o G[o][m] * conj(W[i][o][m])
//   C  out[c][n] = sum_k Z[c][k] * Bi[k][n]:  Bi[m][n] = cos(th), Bi[M + m][n] = -sin(th)   ([C x 2M] @ [2M x N] GEMM)
// When it fits (it does at BASELINE configs[4]: C = 32, N = 512, 16 modes -> 142 KB of the 160 KB), the twiddle matrix
// Bf[n][k] (= Bi[k][n]) is materialised in LDS once per workgroup, so the GEMM loops gather two operands and do no index
// arithmetic; larger geometries gather from the N-entry cosine table instead (DENSE = false).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/spectral_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TPB = 256;

thread_local char g_err[256] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

struct Args {
    const float* in;    // [B, Cin_, N]   (x, or dy)
    const float* wr;    // [Cin, Cout, M]
    const float* wi;
    float* out;         // [B, Cout_, N]  (y, or dx)
    float* spec;        // [B, Cin_, 2, M] saved spectrum (forward: raw S; backward: scaled G) or nullptr
    int cin, cout;      // of the FORWARD operator
    int n, m;
    int backward;
};

// twiddle of column / row k (0 <= k < 2M) at position pos: k < M -> cos(2 pi k pos / N), else -sin(2 pi (k - M) pos / N)
__device__ __forceinline__ float twiddle(const float* tab, int k, int pos, int M, int nmask, int quarter) {
    const bool is_sin = k >= M;
    const int mode = is_sin ? k - M : k;
    const int idx = (mode * pos) & nmask;
    // sin(t) = cos(t - pi/2): index - N/4
    return is_sin ? -tab[(idx - quarter) & nmask] : tab[idx];
}

template <bool DENSE>
__global__ void __launch_bounds__(TPB) spec_conv_kernel(const Args a) {
    extern __shared__ __align__(16) float lds[];
    const int N = a.n, M = a.m, K2 = 2 * M;
    const int c_in = a.backward ? a.cout : a.cin;     // channels of this launch's input / output tensors
    const int c_out = a.backward ? a.cin : a.cout;
    const int NP = N + 4, KP = K2 + 4;
    float* xs = lds;                       // [c_in][NP]
    float* tab = xs + c_in * NP;           // DENSE: Bf [N][KP]; else cos table [N]
    float* S = tab + (DENSE ? N * KP : N); // [c_in][KP]
    float* Z = S + c_in * KP;              // [c_out][KP]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nmask = N - 1, quarter = N >> 2;

    // ---- load the sample and the table -------------------------------------------------------------------------
    const float* src = a.in + (size_t)b * c_in * N;
    for (int i = threadIdx.x; i < (c_in * N) >> 2; i += blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        const int e = i << 2, c = e / N, p = e - c * N;
        *reinterpret_cast<float4*>(xs + c * NP + p) = v;
    }
    if constexpr (DENSE) {
        // N cosines per workgroup (two per thread), then the [N][2M] matrix by table look-up: evaluating 2 M N
        // transcendental functions here cost 16 of the kernel's 40 us
        float* ctab = S;                                  // S / Z are not live yet: N <= (c_in + c_out) * KP is checked on the host
        for (int k = threadIdx.x; k < N; k += blockDim.x) ctab[k] = cospif(2.0f * (float)k / (float)N);
        __syncthreads();
        if ((K2 & (K2 - 1)) == 0) {           // the usual 16 modes: shifts instead of a runtime division per entry
            const int sh = __ffs(K2) - 1;
            for (int i = threadIdx.x; i < N * K2; i += blockDim.x) {
                const int pos = i >> sh, k = i & (K2 - 1);
                tab[pos * KP + k] = twiddle(ctab, k, pos, M, nmask, quarter);
            }
        } else {
            for (int i = threadIdx.x; i < N * K2; i += blockDim.x) {
                const int pos = i / K2, k = i - pos * K2;
                tab[pos * KP + k] = twiddle(ctab, k, pos, M, nmask, quarter);
            }
        }
    } else {
        for (int k = threadIdx.x; k < N; k += blockDim.x) tab[k] = cospif(2.0f * (float)k / (float)N);
    }
    __syncthreads();
    auto tw = [&](int k, int pos) -> float {   // twiddle of spectrum column / row k at grid position pos
        if constexpr (DENSE) return tab[pos * KP + k];
        else return twiddle(tab, k, pos, M, nmask, quarter);
    };

    // ---- A: truncated DFT, S [c_in][2M] ------------------------------------------------------------------------
    {
        const int rtiles = c_in >> 4, ctiles = K2 >> 4;
        for (int t = wave; t < rtiles * ctiles; t += nwaves) {
            const int rt = t / ctiles, ct = t - rt * ctiles;
            const int col = 16 * ct + r;
            const float* arow = xs + (16 * rt + r) * NP;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < N; k0 += 16) {     // four independent accumulators, eight operand loads in flight (N % 32 == 0)
                const int p0 = k0 + q, p1 = k0 + 4 + q, p2 = k0 + 8 + q, p3 = k0 + 12 + q;
                const float a0 = arow[p0], a1 = arow[p1], a2 = arow[p2], a3 = arow[p3];
                const float b0 = tw(col, p0), b1 = tw(col, p1), b2 = tw(col, p2), b3 = tw(col, p3);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc3, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) S[(16 * rt + 4 * q + j) * KP + col] = (acc0[j] + acc1[j]) + (acc2[j] + acc3[j]);
        }
    }
    __syncthreads();

    // ---- B: scaling + complex mode mixing ----------------------------------------------------------------------
    const float s0 = 1.0f / (float)N, s1 = 2.0f / (float)N;
    if (!a.backward) {
        if (a.spec) {   // raw truncated spectrum of x: [c_in][2][M]
            float* dst = a.spec + (size_t)b * c_in * K2;
            for (int i = threadIdx.x; i < c_in * K2; i += blockDim.x) {
                const int c = i / K2, k = i - c * K2;
                dst[i] = S[c * KP + k];
            }
        }
        // The weights ([Cin][Cout][M] x 2, 128 KB at width 32 / 16 modes) stream through the sample's LDS buffer -- dead
        // after phase A -- in slabs of `ch` input channels: coalesced 16-byte loads, every thread busy, instead of
        // 4-byte gathers whose latency four waves cannot cover (that gather WAS the kernel: 42 of its 44 us).
        const int slab = a.cout * M;                       // floats of one input channel's [Cout][M] weight plane
        int ch = 8;
        while (ch > 1 && (2 * ch * slab > c_in * NP || (c_in % ch))) ch >>= 1;
        const bool staged = 2 * ch * slab <= c_in * NP && !(slab & 3);
        float* wbuf_r = xs;
        float* wbuf_i = xs + ch * slab;
        constexpr int EPT = 4;                             // outputs per thread: Cout * M <= 4 * 256 (checked on the host)
        float yr[EPT], yi[EPT];
#pragma unroll
        for (int u = 0; u < EPT; ++u) yr[u] = yi[u] = 0.0f;
        if (staged) {
            for (int i0 = 0; i0 < c_in; i0 += ch) {
                __syncthreads();
                const float4* gr = reinterpret_cast<const float4*>(a.wr + (size_t)i0 * slab);
                const float4* gi = reinterpret_cast<const float4*>(a.wi + (size_t)i0 * slab);
                for (int j = threadIdx.x; j < (ch * slab) >> 2; j += blockDim.x) {
                    reinterpret_cast<float4*>(wbuf_r)[j] = gr[j];
                    reinterpret_cast<float4*>(wbuf_i)[j] = gi[j];
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    const int e = threadIdx.x + u * TPB;
                    if (e < c_out * M) {
                        const int m = e % M;
                        for (int c = 0; c < ch; ++c) {
                            const float xr = S[(i0 + c) * KP + m], xi = S[(i0 + c) * KP + M + m];
                            const float wr = wbuf_r[c * slab + e], wi = wbuf_i[c * slab + e];
                            yr[u] = fmaf(xr, wr, fmaf(-xi, wi, yr[u]));
                            yi[u] = fmaf(xr, wi, fmaf(xi, wr, yi[u]));
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e = threadIdx.x + u * TPB;
                if (e < c_out * M) {
                    const int m = e % M;
                    for (int i = 0; i < c_in; ++i) {
                        const float xr = S[i * KP + m], xi = S[i * KP + M + m];
                        const float wr = a.wr[(size_t)i * slab + e], wi = a.wi[(size_t)i * slab + e];
                        yr[u] = fmaf(xr, wr, fmaf(-xi, wi, yr[u]));
                        yi[u] = fmaf(xr, wi, fmaf(xi, wr, yi[u]));
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int e = threadIdx.x + u * TPB;
            if (e < c_out * M) {
                const int o = e / M, m = e - o * M;
                const float s = m == 0 ? s0 : s1;
                Z[o * KP + m] = s * yr[u];
                Z[o * KP + M + m] = s * yi[u];
            }
        }
    } else {
        for (int i = threadIdx.x; i < c_in * K2; i += blockDim.x) {   // G = s (.) S, in place
            const int c = i / K2, k = i - c * K2;
            const int m = k < M ? k : k - M;
            S[c * KP + k] *= (m == 0 ? s0 : s1);
        }
        __syncthreads();
        if (a.spec) {
            float* dst = a.spec + (size_t)b * c_in * K2;
            for (int i = threadIdx.x; i < c_in * K2; i += blockDim.x) {
                const int c = i / K2, k = i - c * K2;
                dst[i] = S[c * KP + k];
            }
        }
        // GX[i][m] = sum_o G[o][m] * conj(W[i][o][m]); c_out = Cin, c_in = Cout of the forward operator.  The weights stream
        // through the (dead) sample buffer in slabs of `ch` forward-output channels: for every i a contiguous run of ch * M
        // floats at W[i][o0 .. o0 + ch][:]
        int ch = 8;
        while (ch > 1 && (2 * ch * M * c_out > c_in * NP || (c_in % ch))) ch >>= 1;
        const int run = ch * M;                             // floats per i in a slab
        const bool staged = 2 * run * c_out <= c_in * NP && !(run & 3);
        float* wbuf_r = xs;
        float* wbuf_i = xs + run * c_out;
        constexpr int EPT = 4;
        float gr[EPT], gi[EPT];
#pragma unroll
        for (int u = 0; u < EPT; ++u) gr[u] = gi[u] = 0.0f;
        if (staged) {
            const int run4 = run >> 2;
            for (int o0 = 0; o0 < c_in; o0 += ch) {
                __syncthreads();
                for (int j = threadIdx.x; j < c_out * run4; j += blockDim.x) {
                    const int i = j / run4, t = j - i * run4;
                    const size_t g4 = (((size_t)i * a.cout + o0) * M) / 4 + t;
                    reinterpret_cast<float4*>(wbuf_r)[j] = reinterpret_cast<const float4*>(a.wr)[g4];
                    reinterpret_cast<float4*>(wbuf_i)[j] = reinterpret_cast<const float4*>(a.wi)[g4];
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    const int e = threadIdx.x + u * TPB;
                    if (e < c_out * M) {
                        const int i = e / M, m = e - i * M;
                        for (int c = 0; c < ch; ++c) {
                            const float yr = S[(o0 + c) * KP + m], yi = S[(o0 + c) * KP + M + m];
                            const float wr = wbuf_r[i * run + c * M + m], wi = wbuf_i[i * run + c * M + m];
                            gr[u] = fmaf(yr, wr, fmaf(yi, wi, gr[u]));
                            gi[u] = fmaf(yi, wr, fmaf(-yr, wi, gi[u]));
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e = threadIdx.x + u * TPB;
                if (e < c_out * M) {
                    const int i = e / M, m = e - i * M;
                    for (int o = 0; o < c_in; ++o) {
                        const float yr = S[o * KP + m], yi = S[o * KP + M + m];
                        const size_t wi_ = ((size_t)i * a.cout + o) * M + m;
                        const float wr = a.wr[wi_], wi = a.wi[wi_];
                        gr[u] = fmaf(yr, wr, fmaf(yi, wi, gr[u]));
                        gi[u] = fmaf(yi, wr, fmaf(-yr, wi, gi[u]));
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int e = threadIdx.x + u * TPB;
            if (e < c_out * M) {
                const int i = e / M, m = e - i * M;
                Z[i * KP + m] = gr[u];
                Z[i * KP + M + m] = gi[u];
            }
        }
    }
    __syncthreads();

    // ---- C: inverse truncated DFT, out [c_out][N] --------------------------------------------------------------
    {
        float* dst = a.out + (size_t)b * c_out * N;
        const int rtiles = c_out >> 4, ctiles = N >> 4;
        for (int t = wave; t < rtiles * ctiles; t += nwaves) {
            const int rt = t / ctiles, ct = t - rt * ctiles;
            const int pos = 16 * ct + r;
            const float* arow = Z + (16 * rt + r) * KP;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < K2; k0 += 8) {     // 2 M is a multiple of 16
                const int ka = k0 + q, kb = k0 + 4 + q;
                const float a0 = arow[ka], a1 = arow[kb];
                const float b0 = tw(ka, pos), b1 = tw(kb, pos);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[(size_t)(16 * rt + 4 * q + j) * N + pos] = acc0[j] + acc1[j];
        }
    }
}

int launch(void* stream, const Args& a, int b, const char* who) {
    const int c_in = a.backward ? a.cout : a.cin, c_out = a.backward ? a.cin : a.cout;
    const size_t kp = 2 * a.m + 4;
    const size_t base = sizeof(float) * ((size_t)c_in * (a.n + 4) + (size_t)(c_in + c_out) * kp);
    const size_t lds_dense = base + sizeof(float) * (size_t)a.n * kp, lds_table = base + sizeof(float) * (size_t)a.n;
    const bool dense = lds_dense <= 160 * 1024 && (size_t)a.n <= (size_t)(c_in + c_out) * kp;
    const size_t lds = dense ? lds_dense : lds_table;
    if (lds > 160 * 1024) return fail(-4, "%s: needs %zu B of LDS (> 160 KiB): channels x N too large", who, lds);
    auto go = [&](auto kernel) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3(b), dim3(TPB), lds, (hipStream_t)stream, a);
    };
    if (dense) go(spec_conv_kernel<true>);
    else go(spec_conv_kernel<false>);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "%s launch failed: %s", who, hipGetErrorString(e));
    return 0;
}

int check_geometry(const char* who, int b, int cin, int cout, int n, int modes) {
    if (b <= 0 || cin <= 0 || cout <= 0) return fail(-1, "%s: bad argument", who);
    if ((cin & 15) || (cout & 15)) return fail(-4, "%s: channel counts (%d, %d) must be multiples of 16", who, cin, cout);
    if (n < 32 || n > 2048 || (n & (n - 1))) return fail(-4, "%s: N = %d must be a power of two in [32, 2048]", who, n);
    if (modes <= 0 || (modes & 7) || 2 * modes >= n) return fail(-4, "%s: modes = %d must be a multiple of 8 below N/2", who, modes);
    if (cin * modes > 4 * TPB || cout * modes > 4 * TPB)
        return fail(-4, "%s: channels x modes (%d, %d) exceed %d mixing outputs per workgroup", who, cin * modes, cout * modes, 4 * TPB);
    return 0;
}

}  // namespace

extern "C" {

const char* spec_last_error(void) { return g_err; }

int spec_conv_forward(void* stream, const float* x, const float* wr, const float* wi, int b, int cin, int cout, int n, int modes,
                      float* y, float* xft) {
    if (!x || !wr || !wi || !y) return fail(-1, "spec_conv_forward: bad argument");
    if (int rc = check_geometry("spec_conv_forward", b, cin, cout, n, modes)) return rc;
    Args a{x, wr, wi, y, xft, cin, cout, n, modes, 0};
    return launch(stream, a, b, "spec_conv_forward");
}

int spec_conv_backward(void* stream, const float* dy, const float* wr, const float* wi, int b, int cin, int cout, int n, int modes,
                       float* dx, float* gyft) {
    if (!dy || !wr || !wi || !dx) return fail(-1, "spec_conv_backward: bad argument");
    if (int rc = check_geometry("spec_conv_backward", b, cin, cout, n, modes)) return rc;
    Args a{dy, wr, wi, dx, gyft, cin, cout, n, modes, 1};
    return launch(stream, a, b, "spec_conv_backward");
}

}  // extern "C"
