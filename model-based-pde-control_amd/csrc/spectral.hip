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

__global__ void __launch_bounds__(TPB, 2) spec_conv_kernel(const Args a) {
    extern __shared__ __align__(16) float lds[];
    const int N = a.n, M = a.m, K2 = 2 * M;
    const int c_in = a.backward ? a.cout : a.cin;     // channels of this launch's input / output tensors
    const int c_out = a.backward ? a.cin : a.cout;
    const int NP = N + 4, KP = K2 + 4;
    float* xs = lds;                       // [c_in][NP]
    float* tab = xs + c_in * NP;           // [N]
    float* S = tab + N;                    // [c_in][KP]
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
    for (int k = threadIdx.x; k < N; k += blockDim.x) tab[k] = cospif(2.0f * (float)k / (float)N);
    __syncthreads();

    // ---- A: truncated DFT, S [c_in][2M] ------------------------------------------------------------------------
    {
        const int rtiles = c_in >> 4, ctiles = K2 >> 4;
        for (int t = wave; t < rtiles * ctiles; t += nwaves) {
            const int rt = t / ctiles, ct = t - rt * ctiles;
            const int col = 16 * ct + r;
            const float* arow = xs + (16 * rt + r) * NP;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < N; k0 += 8) {      // two independent accumulators
                const int p0 = k0 + q, p1 = k0 + 4 + q;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[p0], twiddle(tab, col, p0, M, nmask, quarter), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[p1], twiddle(tab, col, p1, M, nmask, quarter), acc1, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) S[(16 * rt + 4 * q + j) * KP + col] = acc0[j] + acc1[j];
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
        for (int e = threadIdx.x; e < c_out * M; e += blockDim.x) {
            const int o = e / M, m = e - o * M;
            float yr = 0.0f, yi = 0.0f;
            const size_t wstep = (size_t)a.cout * M;
            const float* pr = a.wr + (size_t)o * M + m;
            const float* pi = a.wi + (size_t)o * M + m;
            for (int i0 = 0; i0 < c_in; i0 += 16) {       // channels are multiples of 16: 32 weight loads in flight
                float wr[16], wi[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    wr[u] = pr[(size_t)(i0 + u) * wstep];
                    wi[u] = pi[(size_t)(i0 + u) * wstep];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float xr = S[(i0 + u) * KP + m], xi = S[(i0 + u) * KP + M + m];
                    yr = fmaf(xr, wr[u], fmaf(-xi, wi[u], yr));
                    yi = fmaf(xr, wi[u], fmaf(xi, wr[u], yi));
                }
            }
            const float s = m == 0 ? s0 : s1;
            Z[o * KP + m] = s * yr;
            Z[o * KP + M + m] = s * yi;
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
        for (int e = threadIdx.x; e < c_out * M; e += blockDim.x) {   // c_out = Cin of the forward operator
            const int i = e / M, m = e - i * M;
            float gr = 0.0f, gi = 0.0f;
            const float* pr = a.wr + (size_t)i * a.cout * M + m;
            const float* pi = a.wi + (size_t)i * a.cout * M + m;
            for (int o0 = 0; o0 < c_in; o0 += 16) {
                float wr[16], wi[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    wr[u] = pr[(size_t)(o0 + u) * M];
                    wi[u] = pi[(size_t)(o0 + u) * M];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float yr = S[(o0 + u) * KP + m], yi = S[(o0 + u) * KP + M + m];
                    gr = fmaf(yr, wr[u], fmaf(yi, wi[u], gr));
                    gi = fmaf(yi, wr[u], fmaf(-yr, wi[u], gi));
                }
            }
            Z[i * KP + m] = gr;
            Z[i * KP + M + m] = gi;
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
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < K2; k0 += 4) {
                const int k = k0 + q;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[k], twiddle(tab, k, pos, M, nmask, quarter), acc, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[(size_t)(16 * rt + 4 * q + j) * N + pos] = acc[j];
        }
    }
}

int launch(void* stream, const Args& a, int b, const char* who) {
    const int c_in = a.backward ? a.cout : a.cin, c_out = a.backward ? a.cin : a.cout;
    const size_t lds = sizeof(float) * ((size_t)c_in * (a.n + 4) + a.n + (size_t)(c_in + c_out) * (2 * a.m + 4));
    if (lds > 160 * 1024) return fail(-4, "%s: needs %zu B of LDS (> 160 KiB): channels x N too large", who, lds);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)spec_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(spec_conv_kernel, dim3(b), dim3(TPB), lds, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-2, "%s launch failed: %s", who, hipGetErrorString(e));
    return 0;
}

int check_geometry(const char* who, int b, int cin, int cout, int n, int modes) {
    if (b <= 0 || cin <= 0 || cout <= 0) return fail(-1, "%s: bad argument", who);
    if ((cin & 15) || (cout & 15)) return fail(-4, "%s: channel counts (%d, %d) must be multiples of 16", who, cin, cout);
    if (n < 32 || n > 2048 || (n & (n - 1))) return fail(-4, "%s: N = %d must be a power of two in [32, 2048]", who, n);
    if (modes <= 0 || (modes & 7) || 2 * modes >= n) return fail(-4, "%s: modes = %d must be a multiple of 8 below N/2", who, modes);
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
