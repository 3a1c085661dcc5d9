/*
 * spectral_hip.h -- C ABI of libspectral_hip.so: fused 1-D spectral convolution (FNO layer core) for gfx950
 * (BASELINE.json configs[4]: "FNO-style surrogate ... spectral-conv kernel").
 *
 * The reference contains no FNO; the operator is the published one (Li et al., "Fourier Neural Operator"):
 *     y = irfft( W . rfft(x)[:modes] ),   W complex [Cin, Cout, modes]
 * i.e. what torch spells as rfft -> einsum("bim,iom->bom") -> zero-padded irfft -- three library calls and two
 * [B, C, N/2+1] complex round trips through HBM.  Only `modes` << N/2 frequencies are kept, so the transforms are
 * TRUNCATED DFTs = small GEMMs ([C x N] @ [N x 2 modes] and [C x 2 modes] @ [2 modes x N]); the kernel does
 * DFT-GEMM -> complex mode mixing -> inverse-DFT-GEMM for one sample per workgroup with everything in LDS
 * (v_mfma_f32_16x16x4_f32, twiddles gathered from one cos table), reading x once and writing y once.
 *
 * The backward pass has the same shape: dx = iDFT'( conj(W) . s (.) DFT(dy) ); the weight gradient is a contraction over
 * the batch of two tiny tensors the launches save ([B, C, 2, modes] each) and is left to the caller (one einsum).
 *
 * All pointers are DEVICE pointers of contiguous fp32 tensors; launches are asynchronous on the given hipStream_t.
 * Return 0 on success, negative on error (spec_last_error()).  Constraints: Cin, Cout multiples of 16; modes a
 * multiple of 8 with modes <= N/2 - 1... (modes < N/2); N a power of two, 32 <= N <= 2048.
 */
#ifndef SPECTRAL_HIP_H
#define SPECTRAL_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* x [B,Cin,N]; wr, wi [Cin,Cout,modes]; y [B,Cout,N]; xft [B,Cin,2,modes] (real | imaginary rows of the truncated
 * rfft of x, saved for the weight gradient) or NULL. */
int spec_conv_forward(void* stream, const float* x, const float* wr, const float* wi, int b, int cin, int cout, int n, int modes,
                      float* y, float* xft);

/* dy [B,Cout,N] -> dx [B,Cin,N]; gyft [B,Cout,2,modes] = d loss / d (mixed spectrum) (real | imaginary), or NULL.
 * Weight gradients: gWr[i,o,m] = sum_b gyr*xr + gyi*xi,  gWi[i,o,m] = sum_b -gyr*xi + gyi*xr. */
int spec_conv_backward(void* stream, const float* dy, const float* wr, const float* wi, int b, int cin, int cout, int n, int modes,
                       float* dx, float* gyft);

const char* spec_last_error(void);

/* ---- whole-network FNO kernels (csrc/fno.hip): one launch per model evaluation ---------------------------------------
 *
 * The FNO-style surrogate of BASELINE configs[4] (pdecontrol/architectures/fno.py::FNO1d: lift -> 4 x [spectral conv +
 * pointwise conv, GELU] -> project; no counterpart in the reference) evaluated for `pairs` (time step, sample)
 * pairs at once, one workgroup per pair, activations in LDS.  This replaces the ~170 kernels per time step that the
 * reference's rollout loop structure (pdecontrol/surrogates/surrogate.py:79-133) would issue for it.
 *
 * Built for width 32, 16 modes, 4 layers (any other geometry: negative return, the caller keeps the per-operator path),
 * N a power of two in [64, 512].  Pair p = t * nb + b reads its state row at u + t * u_stride_t + b * u_stride_b (same
 * for act; strides in floats) and owns row p of every [pairs][...] output. */
typedef struct fno_weights {
    const float* lift_w;        /* [32][2]  (state, action field) -> width */
    const float* lift_b;        /* [32] */
    const float* spec_wr[4];    /* [32 in][32 out][16] real / imaginary parts of the mode-mixing weights */
    const float* spec_wi[4];
    const float* pw_w[4];       /* [32 out][32 in] pointwise convolutions */
    const float* pw_b[4];       /* [32] */
    const float* p1_w;          /* [32][32] project, first layer */
    const float* p1_b;          /* [32] */
    const float* p2_w;          /* [32]     project, second layer (width -> 1) */
    const float* p2_b;          /* [1] */
} fno_weights;

/* delta[p] = model(u[p], act[p]);  out[p] = u[p] + cscale * delta[p] + cshift (the rollout's integration
 * next = prev + dt * dscaling(delta) with an affine dscaling), out may be NULL.
 * pre [pairs][4][32][N] and xspec [4][32][spec_pairs][32] are what the backward pass needs (both NULL: inference); the
 * spectra buffer may span more pairs than this launch (a whole rollout): pair p of the launch is its pair spec_pair0 + p. */
int fno_forward(void* stream, const fno_weights* w, int width, int modes, int layers, int n, int nb, int pairs, const float* u,
                long u_stride_t, long u_stride_b, const float* act, long a_stride_t, long a_stride_b, float cscale, float cshift,
                float* delta, float* out, float* pre, float* xspec, int spec_pairs, int spec_pair0);

/* Backward of fno_forward for the same pairs.  gdelta [pairs][N] = d loss / d delta; gout [nb][N] (or NULL) = d loss / d out
 * of time step gout_t (the gradient a LATER step sent to the prediction it started from).  Writes gspec
 * [4][32][spec_pairs][32] (scaled spectra of the layer gradients; window as in fno_forward), rows [pairs][fno_row_width()] (every parameter gradient except the spectral
 * weights, one row per pair) and, unless NULL, dbase [pairs][N] = d loss / d u. */
int fno_backward(void* stream, const fno_weights* w, int width, int modes, int layers, int n, int nb, int pairs, const float* u,
                 long u_stride_t, long u_stride_b, const float* act, long a_stride_t, long a_stride_b, float cscale,
                 const float* gdelta, const float* gout, int gout_t, const float* pre, float* gspec, int spec_pairs, int spec_pair0,
                 float* rows, float* dbase);

/* Floats per gradient row: lift_w 64 | lift_b 32 | 4 x (pw_w 1024 | pw_b 32) | p1_w 1024 | p1_b 32 | p2_w 32 | p2_b 1 | pad. */
int fno_row_width(void);
/* out[fno_row_width()] = sum over pairs of rows (fixed order). */
int fno_reduce_rows(void* stream, const float* rows, int pairs, float* out);
/* Spectral weight gradients of all 4 layers from the saved spectra ([4][32][pairs][32] each): dwr / dwi are HOST arrays of 4
 * device pointers to [32][32][16] outputs (overwritten). */
int fno_spec_wgrad(void* stream, const float* xspec, const float* gspec, int pairs, float* const* dwr, float* const* dwi);
const char* fno_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
