/*
 * surrogate_hip.h -- C ABI of libsurrogate_hip.so: fused gfx950 kernels for the surrogate's TBPTT step.
 *
 * The reference runs the surrogate as ~60 tiny torch CPU ops per time step
 * (pdecontrol/surrogates/surrogate.py:97-119 -> transition.py:218-226 + models/cnn.py:126-145,35-70).
 * On an MI355X that shape is launch-latency bound, so the per-sample work of whole modules is fused:
 *
 *   sur_encoder_forward / _backward   3 x ResidualBlock (models/cnn.py:73-145) as built by
 *                                     architectures/autoreg.py:51-73: [M,1,N] -> [M,C3,N/4]
 *   sur_step_forward / _backward      one rollout step (surrogate.py:97-107): CNNLSTMCell
 *                                     (transition.py:218-226) + state decoder (autoreg.py:79-94)
 *                                     + integration  out = base + delta * (d * mul + add)
 *
 * One workgroup handles one sample with every activation in LDS; the backward kernels recompute
 * the forward intermediates from the saved inputs (nothing but module inputs/outputs touches HBM)
 * and accumulate the parameter gradients straight into the caller's gradient buffers with fp32
 * atomics (the buffers must be zeroed by the caller each step).
 *
 * All pointers are DEVICE pointers of contiguous fp32 tensors; launches are asynchronous on the
 * given hipStream_t.  Return 0 on success, negative on error (sur_last_error()).
 */
#ifndef SURROGATE_HIP_H
#define SURROGATE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* parameter order inside one ResidualBlock */
enum { SUR_RB_CONV1 = 0, SUR_RB_LN1_W, SUR_RB_LN1_B, SUR_RB_CONV2, SUR_RB_LN2_W, SUR_RB_LN2_B, SUR_RB_SKIP,
       SUR_RB_LN3_W, SUR_RB_LN3_B, SUR_RB_NPARAM };

typedef struct sur_encoder_params {
    const float* w[3 * SUR_RB_NPARAM]; /* weights, block-major                                  */
    float* g[3 * SUR_RB_NPARAM];       /* gradient accumulators (same order) or NULL in forward  */
    int c[4];                          /* channels: in, block0, block1, block2 (1,8,16,16)       */
    int stride[3];                     /* 2, 2, 1                                                */
    int n;                             /* input width N                                          */
} sur_encoder_params;

/* parameter order of the step kernel */
enum { SUR_ST_WXI = 0, SUR_ST_BXI, SUR_ST_WHI, SUR_ST_WXF, SUR_ST_BXF, SUR_ST_WHF, SUR_ST_WXC, SUR_ST_BXC,
       SUR_ST_WHC, SUR_ST_WXO, SUR_ST_BXO, SUR_ST_WHO,
       SUR_ST_DC0_W, SUR_ST_DC0_B, SUR_ST_LN0_W, SUR_ST_LN0_B, SUR_ST_DC1_W, SUR_ST_DC1_B, SUR_ST_LN1_W,
       SUR_ST_LN1_B, SUR_ST_CV2_W, SUR_ST_CV2_B, SUR_ST_LN2_W, SUR_ST_LN2_B, SUR_ST_CV3_W, SUR_ST_CV3_B,
       SUR_ST_NPARAM };

typedef struct sur_step_params {
    const float* w[SUR_ST_NPARAM];
    float* g[SUR_ST_NPARAM];
    int ca, cs;        /* latent action / state channels (4, 16)                         */
    int hq;            /* latent width N/4                                                */
    int c_mid;         /* channels after the 2nd transposed conv (8)                      */
    float delta;       /* integration step                                                */
    float mul, add;    /* dscaling(d) = d * mul + add  (Normalize.Inverse with scalar stats) */
} sur_step_params;

int sur_encoder_forward(void* stream, const sur_encoder_params* p, const float* x, int m, float* z);
/* dx may be NULL (raw data input).  Accumulates into p->g[]. */
int sur_encoder_backward(void* stream, const sur_encoder_params* p, const float* x, const float* dz, int m,
                         float* dx);

int sur_step_forward(void* stream, const sur_step_params* p, const float* xlat, const float* h_in,
                     const float* c_prev, const float* base, int b, float* h_out, float* c_out, float* d_out,
                     float* out);
/* dd / dout / dh / dc: upstream gradients wrt d_out / out / h_out / c_out, each may be NULL (= 0).
 * dxlat / dh_in / dc_prev / dbase: outputs, each may be NULL.  Accumulates into p->g[]. */
int sur_step_backward(void* stream, const sur_step_params* p, const float* xlat, const float* h_in,
                      const float* c_prev, const float* dd, const float* dout, const float* dh, const float* dc,
                      int b, float* dxlat, float* dh_in, float* dc_prev, float* dbase);

const char* sur_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
