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

#ifdef __cplusplus
}
#endif
#endif
