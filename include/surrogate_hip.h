/*
 * surrogate_hip.h -- C ABI of libsurrogate_hip.so: fused gfx950 kernels for the surrogate's TBPTT step.
 *
 * The reference runs the surrogate as ~60 tiny torch CPU ops per time step
 * (pdecontrol/surrogates/surrogate.py:97-119 -> transition.py:218-226 + models/cnn.py:126-145,35-70).
 * On an MI355X that shape is launch-latency bound, so whole modules are fused per sample:
 *
 *   sur_encoder_forward / _backward   3 x ResidualBlock (models/cnn.py:73-145) as built by
 *                                     architectures/autoreg.py:51-73: [M,1,N] -> [M,C3,N/4]
 *   sur_chunk_forward / _backward     a whole TBPTT chunk of K rollout steps (surrogate.py:97-119):
 *                                     per step CNNLSTMCell (transition.py:218-226) + state decoder
 *                                     (autoreg.py:79-94) + integration out = base + delta*(d*mul+add);
 *                                     teacher forcing on the first S steps (transition.py:274-279),
 *                                     free running afterwards (:285-296).  Only the cell is a recurrence: its
 *                                     time loop runs INSIDE one kernel (weights and hidden state in LDS, one
 *                                     workgroup per sample); the decoders of all (step, sample) pairs run in
 *                                     parallel on every CU, forward and backward.
 *   sur_flush_*_grads                 reduces the per-workgroup partial gradient rows into the
 *                                     parameter gradient tensors (deterministic, no atomics)
 *
 * One workgroup handles one sample (or one (step, sample) pair) at a time with every activation in LDS; forward
 * launches can save their intermediates (`saved` buffers) and the backward launches read them back instead of
 * recomputing.  Parameter gradients are summed over space and time inside the workgroup and added to that
 * workgroup's own row of `partial` ([rows][sum(size)] fp32, caller-allocated, zero-initialised); the flush
 * kernels sum the rows into g[] (optionally applying the Adam update, `sur_adam`) and re-zero them.
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
#define SUR_ENC_NPARAM (3 * SUR_RB_NPARAM)

typedef struct sur_encoder_params {
    const float* w[SUR_ENC_NPARAM]; /* weights, block-major                                     */
    float* g[SUR_ENC_NPARAM];       /* gradient tensors (targets of sur_flush_encoder_grads)    */
    int size[SUR_ENC_NPARAM];       /* element count of each parameter                           */
    int c[4];                       /* channels: in, block0, block1, block2 (1,8,16,16)          */
    int stride[3];                  /* 2, 2, 1                                                   */
    int n;                          /* input width N                                             */
    float* partial;                 /* [rows][sum(size)] partial gradient rows                   */
    int rows;
} sur_encoder_params;

/* parameter order of the chunk kernel */
enum { SUR_ST_WXI = 0, SUR_ST_BXI, SUR_ST_WHI, SUR_ST_WXF, SUR_ST_BXF, SUR_ST_WHF, SUR_ST_WXC, SUR_ST_BXC,
       SUR_ST_WHC, SUR_ST_WXO, SUR_ST_BXO, SUR_ST_WHO,
       SUR_ST_DC0_W, SUR_ST_DC0_B, SUR_ST_LN0_W, SUR_ST_LN0_B, SUR_ST_DC1_W, SUR_ST_DC1_B, SUR_ST_LN1_W,
       SUR_ST_LN1_B, SUR_ST_CV2_W, SUR_ST_CV2_B, SUR_ST_LN2_W, SUR_ST_LN2_B, SUR_ST_CV3_W, SUR_ST_CV3_B,
       SUR_ST_NPARAM };

typedef struct sur_chunk_params {
    const float* w[SUR_ST_NPARAM];
    float* g[SUR_ST_NPARAM];
    int size[SUR_ST_NPARAM];
    int ca, cs;        /* latent action / state channels (4, 16)                            */
    int hq;            /* latent width N/4                                                   */
    int c_mid;         /* channels after the 2nd transposed conv (8)                         */
    float delta;       /* integration step                                                   */
    float mul, add;    /* dscaling(d) = d * mul + add  (Normalize.Inverse with scalar stats) */
    float* partial;    /* [rows][sum(size)]                                                  */
    int rows;
} sur_chunk_params;

/* 0 when the fused kernels implement this geometry (any pointer may be NULL), negative otherwise (sur_last_error()).
 * The fused LayerNorm takes rows of 16, 32, 64, 128, 192 or 256 values: with the reference's strides (2, 2, 1) and the
 * decoder's two transposed convolutions that is N in {64, 128, 256}.  Every launch checks the same and returns -4. */
int sur_geometry_supported(const sur_encoder_params* state_enc, const sur_encoder_params* action_enc, const sur_chunk_params* chunk);

/* Floats per sample of the forward intermediates sur_encoder_forward can save for sur_encoder_backward (0 = this
 * geometry has no saved-activation path).  With `saved` [M, sur_encoder_saved_floats] the backward kernel loads
 * the three blocks' intermediates instead of recomputing them (a third of its time); results are bit-identical. */
int sur_encoder_saved_floats(const sur_encoder_params* p);

/* Optional optimizer step inside the flush launches: torch.optim.Adam(lr, betas, eps) without weight decay /
 * amsgrad (reference: pdecontrol/surrogates/training.py:273-278).  m, v: flat [sum(size)] moment buffers in the
 * pack's parameter order, zero-initialised; step: device counter of completed updates, incremented by the launch;
 * ticket: one zero-initialised unsigned the launch leaves at zero.  With a descriptor the flush writes
 * g = (sum of the partial rows) instead of accumulating into g, and updates the parameters in place. */
typedef struct sur_adam {
    float* m;
    float* v;
    int* step;
    unsigned int* ticket;
    const float* lr;   /* DEVICE scalar: a learning-rate scheduler updates it between replays of a captured step */
    float beta1, beta2, eps;
} sur_adam;

int sur_encoder_forward(void* stream, const sur_encoder_params* p, const float* x, int m, float* z,
                        float* saved /* may be NULL */);
/* dx may be NULL (raw data input).  Accumulates parameter gradients into rows
 * [row_base, row_base + row_count) of p->partial (one row per workgroup; launches that may run
 * concurrently on different streams must be given disjoint row ranges). */
int sur_encoder_backward(void* stream, const sur_encoder_params* p, const float* x, const float* dz, int m,
                         float* dx, int row_base, int row_count, const float* saved /* or NULL = recompute */);
/* Up to three encoder backward jobs (arrays of length njobs; dx is not produced) per launch: all workgroups are
 * dispatched together, so the long action-encoder job does not queue behind the state-encoder jobs.  When every job
 * has a `saved` buffer and a workspace (sur_encoder_workspace_floats(p, m) floats), the backward runs one residual
 * block per launch (three launches, each with a third of the registers and LDS: more workgroups per CU); otherwise one
 * whole-encoder launch. */
int sur_encoder_workspace_floats(const sur_encoder_params* p, int m);
/* One or two encoder forward jobs with `saved` buffers, one residual block per launch (three launches, each carrying
 * all jobs, at most max_workgroups workgroups per job): the large-sample-count form of sur_encoder_forward. */
int sur_encoder_forward_multi(void* stream, int njobs, const sur_encoder_params* const* ps, const float* const* xs, const int* ms,
                              float* const* zs, float* const* saveds, int max_workgroups);
int sur_encoder_backward_multi(void* stream, int njobs, const sur_encoder_params* const* ps, const float* const* xs,
                               const float* const* dzs, const int* ms, const int* row_bases, const int* row_counts,
                               const float* const* saveds, float* const* workspaces /* array may be NULL */);
/* overwrite != 0 (and no Adam descriptor): g = sum of the rows instead of g += sum -- for gradient tensors that start
 * undefined (optimizer.zero_grad(set_to_none=True)), saves the zero-fill. */
int sur_flush_encoder_grads(void* stream, const sur_encoder_params* p, const sur_adam* adam /* may be NULL */, int overwrite);

/* Floats per (step, sample) of the forward intermediates sur_chunk_forward saves for sur_chunk_backward (activated
 * gates, c_k, h_k, decoder pre-/post-LayerNorm activations; 14.8 KB at N = 64), or 0 if the latent sizes are not
 * float4-granular (then only the forward is available).  The backward kernels read the intermediates back from HBM
 * instead of recomputing the steps -- HBM capacity and bandwidth are free on this device, dependent-phase latency
 * is not. */
int sur_chunk_saved_floats(const sur_chunk_params* p);

/* Floats of the scratch buffer sur_chunk_backward needs for its split path (0: bad arguments). */
int sur_chunk_workspace_floats(const sur_chunk_params* p, int k, int b);

/* Only the ConvLSTM cell is a recurrence: sur_chunk_forward runs the cell chain with one workgroup per sample, then
 * the decoders of ALL (step, sample) pairs in parallel on every CU, then the integration; sur_chunk_backward (given
 * `saved` and `workspace`) runs the decoder backward of all pairs in parallel, then the cell chain's BPTT.
 *
 * Time-major tensors: xlat_t [K,B,ca,hq]; lstates_t [S,B,cs,hq] (encoded given states, S >= 1);
 * states_t [S,B,1,N] (the given states: bases of the teacher-forced steps); h0, c0 [B,cs,hq] with hc_bstride =
 * cs*hq elements between samples, or ONE [cs,hq] initial state shared by the batch with hc_bstride = 0 (the
 * reference's H0 / C0 parameters, transition.py:254-259).
 * Outputs: h_all, c_all [K,B,cs,hq]; d_all, out_all [K,B,1,N]. */
int sur_chunk_forward(void* stream, const sur_chunk_params* p, const float* xlat_t, const float* lstates_t,
                      const float* states_t, const float* h0, const float* c0, int hc_bstride, int k, int s, int b, float* h_all,
                      float* c_all, float* d_all, float* out_all, float* saved /* NULL: forward only, no backward later */);
/* Upstream gradients (each may be NULL = 0): dd_all / dout_all [K,B,1,N] wrt d_all / out_all;
 * dh_all / dc_all [K,B,cs,hq] wrt h_all / c_all.  Outputs (each may be NULL): dxlat_t [K,B,ca,hq],
 * dlstates_t [S,B,cs,hq], dh0, dc0 [B,cs,hq].  Accumulates parameter gradients into rows
 * [row_base, row_base + row_count) of p->partial (row_count >= B; the parallel decoder backward uses one row
 * per workgroup, up to row_count of them). */
int sur_chunk_backward(void* stream, const sur_chunk_params* p, const float* xlat_t, const float* lstates_t,
                       const float* h0, const float* c0, int hc_bstride, const float* h_all, const float* c_all,
                       const float* dd_all, const float* dout_all, const float* dh_all, const float* dc_all, int k,
                       int s, int b, float* dxlat_t, float* dlstates_t, float* dh0, float* dc0, int row_base, int row_count,
                       const float* saved /* what sur_chunk_forward wrote */, float* workspace /* sur_chunk_workspace_floats */);
/* The backward pass of SEVERAL consecutive TBPTT chunks in the same three launches (TBPTT cuts the graph between
 * chunks, so their cell chains are independent: one workgroup per (chunk, sample); the parallel kernels do not care
 * about chunk borders).  The chunks tile the time axis [0, k_total) of time-major tensors shared by all of them:
 * xlat_t [K,B,ca,hq], h_all / c_all [K,B,cs,hq], saved [K,B,sur_chunk_saved_floats], dd_all [K,B,1,N],
 * dxlat_t [K,B,ca,hq], workspace sur_chunk_workspace_floats(p, k_total, b).  Per chunk: its first `s` steps are
 * teacher forced with lstates_t [s,B,cs,hq] (gradient -> dlstates_t, may be NULL), its initial state is h0 / c0. */
#define SUR_MAX_SPANS 4
typedef struct sur_chunk_span {
    int k0, k1, s;
    const float* lstates_t;
    const float* h0;
    const float* c0;
    int hc_bstride;
    float* dlstates_t;
} sur_chunk_span;
int sur_chunks_backward(void* stream, const sur_chunk_params* p, int nspans, const sur_chunk_span* spans, const float* xlat_t,
                        const float* h_all, const float* c_all, const float* dd_all, int k_total, int b, float* dxlat_t,
                        int row_base, int row_count, const float* saved, float* workspace);
int sur_flush_chunk_grads(void* stream, const sur_chunk_params* p, const sur_adam* adam /* may be NULL */, int overwrite);
/* The reductions of a surrogate's three parameter packs (two encoders, chunk) in ONE launch; bit j of overwrite_mask
 * is the `overwrite` flag of pack j. */
int sur_flush_all_grads(void* stream, const sur_encoder_params* e0, const sur_adam* a0, const sur_encoder_params* e1,
                        const sur_adam* a1, const sur_chunk_params* c2, const sur_adam* a2, int overwrite_mask);

/* torch.optim.Adam step of up to three packs FROM their gradient tensors g[] in one launch (the optimizer of the eager
 * fused path: reference pdecontrol/surrogates/training.py:273-278 driven by pl.Trainer.fit, mbrl.py:593).  A NULL
 * descriptor skips its pack.  Moments / step counter / device learning rate as in sur_adam above. */
int sur_adam_apply(void* stream, const sur_encoder_params* e0, const sur_adam* a0, const sur_encoder_params* e1,
                   const sur_adam* a1, const sur_chunk_params* c2, const sur_adam* a2);

/* Delta-mode TBPTT loss in one launch (reference: pdecontrol/surrogates/training.py:100-121):
 *   deltas[b,t]  = ((states[b,t+1] - states[b,t]) / delta - mean) / stdv          t < T-1   (undscaling forward)
 *   loss         = mean over (b, t < T-1, i) of (d_all[t,b,i] - deltas[b,t,i])^2  (MSE, reduction "none" + mean)
 *   hsteploss[t] = the same mean per time step;  stats = {mean, unbiased std} of the predicted deltas
 *                  d_all[:T-1], then of the true deltas (the four "Train ... Delta" metrics)
 *   dd_all       = d loss / d d_all  [T,B,1,N] (last step zero), may be NULL
 * states [B,T,1,N] with element strides (states_bstride, states_tstride) between samples / time steps -- contiguous
 * batch-major (T*N, N) or a view of time-major storage (N, B*N); d_all [T,B,1,N] time-major as written by sur_chunk_forward; deltas [B,T-1,1,N];
 * hsteploss [T-1]; loss [1]; stats [4]; partial: scratch of 40*T doubles; ticket: one zero-initialised
 * unsigned the kernel leaves at zero.  Sums are fp64 and reduced in a fixed order (deterministic). */
int sur_tbptt_delta_loss(void* stream, const float* states, long states_bstride, long states_tstride, const float* d_all, int b,
                         int t, int n, float delta, float mean,
                         float stdv, float* deltas, float* dd_all, float* hsteploss, float* loss, float* stats,
                         double* partial, unsigned int* ticket);

/* The integration pass of sur_chunk_forward on its own (out_k = base_k + delta * (d_k * mul + add), surrogate.py:108-117):
 * sur_chunk_forward skips it when called with out_all = NULL, so a caller that needs the integrated predictions of a chunk
 * only for reporting (the last TBPTT chunk: nothing is rolled out from them) can queue it off its critical path. */
int sur_chunk_integrate(void* stream, const sur_chunk_params* p, const float* states_t, const float* d_all, int k, int s, int b,
                        float* out_all);

/* sur_tbptt_delta_loss_range in two halves: _rows writes the `deltas` / `dd_all` rows [t_begin, t_end) and their partial sums
 * and takes no ticket -- the backward pass can start from dd_all right away --, _finalize (one workgroup, after every row
 * launch of the loss) reduces the partial sums into loss / hsteploss / stats.  A loss may mix _range launches (early
 * chunks) with ONE _rows launch as long as _finalize follows them all. */
int sur_tbptt_delta_loss_rows(void* stream, const float* states, long states_bstride, long states_tstride, const float* d_all,
                              int b, int t, int n, float delta, float mean, float stdv, float* deltas, float* dd_all,
                              float* hsteploss, float* loss, float* stats, double* partial, unsigned int* ticket, int t_begin,
                              int t_end);
int sur_tbptt_delta_loss_finalize(void* stream, int b, int t, int n, float* hsteploss, float* loss, float* stats, double* partial,
                                  unsigned int* ticket);

/* Fold partial-gradient rows [bases[j], bases[j] + counts[j]) of pack j (0: e0, 1: e1, 2: c2) into row dsts[j] of the same
 * buffer (dst += sum, fixed order; dst outside the folded range) and re-zero them.  A backward branch that ends early
 * (an early TBPTT chunk) folds its own rows, so the step's final sur_flush_all_grads reads one row per branch. */
int sur_fold_rows(void* stream, const sur_encoder_params* e0, const sur_encoder_params* e1, const sur_chunk_params* c2,
                  const int* bases, const int* counts, const int* dsts);

/* The same loss over the time steps [t_begin, t_end) of the T rows only: the launches of one loss -- e.g. one per TBPTT
 * chunk, issued as soon as that chunk's predictions exist, on any streams and in any order -- must cover [0, T) exactly
 * once between them; `loss`, `hsteploss` and `stats` are written by whichever launch finishes last (the ticket counts
 * the workgroups of all T rows), `deltas` / `dd_all` rows by the launch that covers them.  Lets the backward pass of an
 * early chunk start while later chunks are still rolling out (training.py:71-121: gradients are cut between chunks). */
int sur_tbptt_delta_loss_range(void* stream, const float* states, long states_bstride, long states_tstride, const float* d_all,
                               int b, int t, int n, float delta, float mean, float stdv, float* deltas, float* dd_all,
                               float* hsteploss, float* loss, float* stats, double* partial, unsigned int* ticket, int t_begin,
                               int t_end);

const char* sur_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
