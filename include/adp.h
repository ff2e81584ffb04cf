/*
 * adp.h -- C-ABI of libadp_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * 1-D U-Net denoising hot path of archinetai/audio-diffusion-pytorch.
 *
 * The reference has NO native / FFI boundary for this path (it is pure Python):
 * every entry point below replaces a chain of ATen ops reached through the third-party
 * `a_unet` blocks that /root/reference/audio_diffusion_pytorch/components.py:79-105 composes,
 * or the tensor math of /root/reference/audio_diffusion_pytorch/diffusion.py:77-95, :172-190.
 * The citation on each function names the reference call site it stands in for.
 *
 * Conventions (all functions):
 *   - plain device pointers (fp32 unless noted), int64 sizes, a hipStream_t passed as void*;
 *   - returns 0 (ADP_OK) or a negative ADP_ERR_* code; never throws, never allocates,
 *     never synchronises: stream-ordered, re-entrant, hipGraph-capturable;
 *   - tensors are contiguous, activations laid out [B, C, L] with L fastest.
 */
#ifndef ADP_H
#define ADP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADP_OK 0
#define ADP_ERR_SHAPE (-1)
#define ADP_ERR_UNSUPPORTED (-2)
#define ADP_ERR_ALIGN (-3)
#define ADP_ERR_LAUNCH (-4)
#define ADP_ERR_NULL (-5)

int adp_version(void);

/* Profiling introspection (host side only, per calling thread): copies the list of kernels launched by this
 * thread's adp_* calls since the previous adp_launch_trace call into buf (";"-separated
 * "<kernel expression>@<launcher signature incl. template arguments>", NUL terminated, at most cap bytes), clears
 * it, and switches recording on (enable != 0) or off.  Returns the number of bytes written.  bench.py uses it to
 * label each timed launch with the kernel instantiation name rocprofv3 reports. */
int64_t adp_launch_trace(int64_t enable, char* buf, int64_t cap);
/* While recording is on, every kernel launch is bracketed by two HIP events recorded on the stream it is launched
 * on.  adp_launch_times waits for them, writes the elapsed milliseconds of each launch since the previous call
 * (launch order, the order adp_launch_trace lists the kernels) into ms[0..cap) and returns how many it wrote. */
int64_t adp_launch_times(float* ms, int64_t cap);

/* ------------------------------------------------------------------------------------------
 * Fused implicit-GEMM Conv1d on the f32 matrix cores (v_mfma_f32_32x32x2_f32).
 *
 *   out[b, m, n] = e_scale[b,m] * ( bias[m] + sum_{r,t} A(m,r,t) * Xv[b, r, n*stride + t*dil - pad] ) + res[...]
 *
 *   Xv = prologue(x) nearest-upsampled by `up` and zero padded (padding applies AFTER the
 *        prologue, exactly like nn.Conv1d padding after GroupNorm+SiLU).
 *   A(m,r,t) = w[m][r][t]              (transposed = 0: forward conv, w is [M, R, KT])
 *            = w[r][m][KT-1-t]         (transposed = 1: data gradient, w is [R, M, KT])
 *   prologue: 0 none | 1 GroupNorm(groups)+SiLU using pro_stats[b,g,{mean,rstd}], pro_gamma/beta[r]
 *             | 2 LayerNorm over channels using pro_stats[b,l,{mean,rstd}], pro_gamma/beta[r]
 *   store:    0 out[b][m][n]
 *             | 1 pixel-shuffle out[b][m / sp][n*sp + m % sp]   (gradient of a kernel=stride=sp conv)
 *             | 2 pooled       out[b][m][n / sp] = sum of sp adjacent n (gradient of nearest upsample)
 *   x2 / R1: optional channel concat -- channels r >= R1 are read from x2[b][r-R1][.] (R1 = R: unused).
 *
 * Replaces (reference call sites): ResnetItem ConvBlocks (components.py:89), DownsampleItem /
 * UpsampleItem (implicit XBlock defaults, components.py:84), the Linear projections of
 * AttentionItem / CrossAttentionItem (components.py:92-93) run as 1x1 convs, InjectChannelsItem
 * and AppendChannelsPlugin's torch.cat (components.py:91, :174-176), SkipModulate's merge
 * (components.py:99), and their autograd data gradients.
 * ------------------------------------------------------------------------------------------ */
typedef struct adp_conv_desc {
  const float* x;          /* [B, R1, Lin] */
  const float* x2;         /* [B, R-R1, Lin] or NULL */
  const float* w;          /* see A(m,r,t) */
  const float* bias;       /* [M] or NULL */
  const float* pro_stats;  /* prologue statistics or NULL */
  const float* pro_gamma;  /* [R] or NULL (=1) */
  const float* pro_beta;   /* [R] or NULL (=0) */
  const float* e_scale;    /* e_scale[b*e_bstride + m] or NULL (=1) */
  const float* res;        /* same layout as out, or NULL */
  float* out;
  float* out_pre;          /* optional (store 0 only): bias + conv BEFORE e_scale / res, kept for the backward pass */
  int64_t B, R, R1, Lin, M, N; /* N = output positions per batch element BEFORE the store transform */
  int64_t KT, stride, dil, pad, up;
  int64_t transposed, prologue, groups, store, sp;
  int64_t e_bstride;       /* batch stride of e_scale in floats (0 -> M) */
  float* ws;               /* scratch of adp_conv1d_ws_bytes(d) bytes (NULL when that is 0): cross-workgroup split-K
                              partial tiles of small-grid problems (batch-1 deep layers) */
  float* gn_part;          /* optional: GroupNorm partial statistics of the OUTPUT tensor, written by the epilogue:
                              gn_part[((b*M/4 + q)*E + e)*3 + {0,1,2}] = (mean, M2, count) of the e-th slice of the
                              4-channel row quad q of batch element b, E = adp_conv1d_gn_entries(d) (0: this shape /
                              kernel family cannot, pass NULL).  The consumer's GroupNorm then needs no pass of its
                              own over the tensor: adp_gn_finalize. */
  /* optional (round 6): this launch is the data gradient that feeds the backward of a = SiLU(GroupNorm(gnb_x)) -- its output
     is da.  The epilogue then also leaves the first stage of that backward (what adp_gn_silu_bwd_reduce computes from a pass
     over x and da): gnb_ab[((b*M + m)*E + e)*2 + {0,1}] = (sum ds*xhat, sum ds) over the e-th position slice of row m,
     ds = da * silu'(gamma*xhat + beta), E = adp_conv1d_gnb_entries(d) (0: this launch cannot; leave gnb_ab NULL).
     The second stage is adp_gn_silu_bwd_apply_ab(..., NSab = E).  components.py:89 (ConvBlock), backward. */
  const float* gnb_x;      /* [B, M, N], the GroupNorm's input */
  const float* gnb_stats;  /* [B, gnb_groups, 2] (mean, rstd) */
  const float* gnb_gamma;  /* [M] */
  const float* gnb_beta;   /* [M] */
  float* gnb_ab;
  int64_t gnb_groups;
} adp_conv_desc;

/* Scratch the launch wants (0 for most shapes).  With ws == NULL the call still succeeds on the unsplit path. */
int64_t adp_conv1d_ws_bytes(const adp_conv_desc* d);
/* Slices per output row the epilogue would report statistics for (see gn_part); 0 = not available for this launch.
   Depends on whether the launch will K-split: fill in d->ws (adp_conv1d_ws_bytes) BEFORE asking. */
int64_t adp_conv1d_gn_entries(const adp_conv_desc* d);
/* Position slices per output row of gnb_ab (see there); 0 = the kernel this launch dispatches to has no such epilogue
   (fill in d->ws first, as above). */
int64_t adp_conv1d_gnb_entries(const adp_conv_desc* d);
int adp_conv1d(const adp_conv_desc* d, void* stream);
/* tile the dispatcher selects for this problem, BM*1000+BN (introspection for profiling / roofline reports) */
int64_t adp_conv1d_tile(const adp_conv_desc* d);

/* Weight (+bias) gradient of the same convolution, deterministic two-stage reduction:
 *   dw[m][r][t] = sum_{b,n} dy[b,m,n] * Xv[b, r, n*stride + t*dil - pad]     dbias[m] = sum_{b,n} dy[b,m,n]
 * `x`/prologue fields describe Xv exactly as in adp_conv1d (the activated conv input is
 * recomputed on load, never materialised).  ws must hold adp_conv1d_wgrad_ws_bytes().
 * Replaces autograd's conv weight gradient for every Conv1d above. */
typedef struct adp_wgrad_desc {
  const float* x;
  const float* x2;
  const float* dy;         /* [B, M, N] */
  const float* pro_stats;
  const float* pro_gamma;
  const float* pro_beta;
  float* dw;               /* [M, R, KT] (overwritten, or accumulated when accumulate != 0) */
  float* dbias;            /* [M] or NULL */
  float* ws;
  int64_t B, R, R1, Lin, M, N;
  int64_t KT, stride, dil, pad, up;
  int64_t prologue, groups;
  int64_t accumulate;      /* bit 0: dw / dbias += instead of =.  bit 1 (value 2): PARK the second stage -- the launch leaves its
                              adp_conv1d_wgrad_partials(d) partial slices in ws and the caller sums them later with
                              adp_wgrad_reduce_batch (set it only when that query returns > 1; ws must stay alive until then) */
} adp_wgrad_desc;

int64_t adp_conv1d_wgrad_ws_bytes(const adp_wgrad_desc* d);
int adp_conv1d_wgrad(const adp_wgrad_desc* d, void* stream);
/* n weight gradients of ONE shape (descriptors equal in every integer field and in which optional pointers are set; each with
   its own ws) -- the ConvBlock convs of one side of a U-Net block.  The matrix-core family runs them as ONE launch per 8 (the
   items' workgroups follow each other on a CU without a kernel boundary) + one batched second stage; other families run item by
   item.  Nothing on the backward chain waits for a weight gradient, so the caller may collect them until the block side is done. */
int adp_conv1d_wgrad_batch(const adp_wgrad_desc* descs, int64_t n, void* stream);
/* Partial slices a PARKED launch of this problem leaves in ws (see adp_wgrad_desc.accumulate); 1 = the shape has no parked form
   (the call always finishes dw itself).  Layout of ws: [partials][M*R*KT] followed by [partials][M] (dbias). */
int64_t adp_conv1d_wgrad_partials(const adp_wgrad_desc* d);
/* Second stage of n parked weight gradients of ONE shape (partials, cnt = M*R*KT, M) in one launch per 8:
   dw_i[j] (+)= sum_k ws_i[k*cnt + j], dbias_i[m] (+)= sum_k ws_i[partials*cnt + k*M + m]  (fixed summation order).
   ws / dw / dbias are HOST arrays of device pointers; dbias may be NULL (no bias gradients).  The ConvBlock convs of one U-Net
   depth share their shape: their second stages are batched per side of the block. */
int adp_wgrad_reduce_batch(const float* const* ws, float* const* dw, float* const* dbias, int64_t n, int64_t partials,
                           int64_t cnt, int64_t M, int64_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm statistics (nn.GroupNorm inside a_unet ConvBlock; components.py:89, resnet_groups :46)
 * stats[b, g, 0] = mean, stats[b, g, 1] = rstd = 1/sqrt(var + eps)  (biased variance).
 * ws: adp_gn_stats_ws_bytes(B, C, L, G) bytes of scratch.
 * ------------------------------------------------------------------------------------------ */
int64_t adp_gn_stats_ws_bytes(int64_t B, int64_t C, int64_t L, int64_t G);
int adp_gn_stats(const float* x, int64_t B, int64_t C, int64_t L, int64_t G, float eps, float* stats, float* ws,
                 void* stream);

/* Statistics + materialised activation in the same two launches: stats as adp_gn_stats, and
 * act[b,c,l] = SiLU((x - mean) * rstd * gamma[c] + beta[c]).  For the wide layers (C >= 512), where 8-16 workgroups of
 * the conv kernels would each recompute the activation of the tile they stage; the narrow layers keep the activation
 * fused into the conv loaders (adp_conv_desc.prologue = 1).  ws: adp_gn_stats_ws_bytes. */
int adp_gn_stats_act(const float* x, int64_t B, int64_t C, int64_t L, int64_t G, float eps, const float* gamma,
                     const float* beta, float* stats, float* act, float* ws, void* stream);

/* GroupNorm statistics from producer-side partials (conv epilogues): part[((b*C/4 + q)*E + e)*3 + {0,1,2}] =
 * (mean, M2, count) of slice e of row quad q; Chan's combination over the (C/G)/4 quads x E slices of each group, one
 * wave per (b, g); (C/G) % 4 == 0.  stats as adp_gn_stats.  Replaces the statistics pass over the activation. */
int adp_gn_finalize(const float* part, int64_t B, int64_t C, int64_t E, int64_t G, float eps, float* stats,
                    void* stream);
/* adp_gn_finalize + adp_gn_act in ONE launch for rows with few slices (wide layers fed by a conv epilogue): every
 * workgroup combines its group's C/G * E partial entries itself. */
int adp_gn_finalize_act(const float* x, const float* part, int64_t B, int64_t C, int64_t L, int64_t E, int64_t G,
                        float eps, const float* gamma, const float* beta, float* stats, float* act, void* stream);
/* act[b,c,l] = SiLU((x - mean) * rstd * gamma[c] + beta[c]) from finished statistics (the wide-layer path of
 * adp_gn_stats_act when the statistics came from adp_gn_finalize). */
int adp_gn_act(const float* x, const float* stats, const float* gamma, const float* beta, int64_t B, int64_t C,
               int64_t L, int64_t G, float* act, void* stream);

/* Split count used by the row-wise two-stage reductions below (rows = B*C rows of length L). */
int64_t adp_row_nsplit(int64_t rows, int64_t L);

/* Backward of y = SiLU(GroupNorm(x)) given dact = dL/dy (the conv data gradient), NS = adp_row_nsplit(B*C, L):
 *   adp_gn_silu_bwd_reduce: ab[b,c,s,0] = sum_l ds*xhat, ab[b,c,s,1] = sum_l ds over slice s of L, ds = dact*silu'(h)
 *   adp_gn_silu_bwd_apply : dx = rstd*(gamma*ds - m1 - xhat*m2) (+ dres); with dgamma/dbeta != NULL it also writes
 *                           the parameter gradients (same arithmetic as adp_gn_param_grad, no third launch)
 *   adp_gn_param_grad     : dgamma[c] = sum_{b,s} ab[..0], dbeta[c] = sum_{b,s} ab[..1]
 * ab holds B*C*NS*2 floats. */
int adp_gn_silu_bwd_reduce(const float* x, const float* dact, const float* stats, const float* gamma,
                           const float* beta, int64_t B, int64_t C, int64_t L, int64_t G, int64_t NS, float* ab,
                           void* stream);
int adp_gn_silu_bwd_apply(const float* x, const float* dact, const float* stats, const float* gamma,
                          const float* beta, const float* ab, const float* dres, int64_t B, int64_t C, int64_t L,
                          int64_t G, int64_t NS, float* dx, float* dgamma, float* dbeta, int64_t accumulate,
                          void* stream);
/* adp_gn_silu_bwd_apply with the first stage's sums laid out for NSab slices per row (written by a data-gradient conv's
   epilogue: adp_conv_desc.gnb_ab) while the launch itself splits its rows NS ways; NSab == NS is the call above. */
int adp_gn_silu_bwd_apply_ab(const float* x, const float* dact, const float* stats, const float* gamma,
                             const float* beta, const float* ab, const float* dres, int64_t B, int64_t C, int64_t L,
                             int64_t G, int64_t NS, int64_t NSab, float* dx, float* dgamma, float* dbeta,
                             int64_t accumulate, void* stream);
int adp_gn_param_grad(const float* ab, int64_t B, int64_t C, int64_t NS, float* dgamma, float* dbeta,
                      int64_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Modulation (a_unet ModulationItem, components.py:90): per position LayerNorm over channels
 * without affine, then * (1 + scale[b,c]) + shift[b,c]; scale = ss[b*ss_bstride + c],
 * shift = ss[b*ss_bstride + C + c] (a slice of the conditioning bank, see adp_linear_fwd).
 * stats[b, l, {mean, rstd}] is written for the backward pass.
 * ------------------------------------------------------------------------------------------ */
int adp_modulation_fwd(const float* x, const float* ss, int64_t ss_bstride, int64_t B, int64_t C, int64_t L,
                       float eps, float* y, float* stats, void* stream);
/* ModulationItem followed by an AttentionItem / CrossAttentionItem (components.py:90-93) in one launch: y and stats as
 * adp_modulation_fwd, then the attention's LayerNorm of y while the tile is still in registers: xn = LN(y) * gamma + beta,
 * xn2 = LN(y) * gamma2 + beta2 (norm_context of a self-attention item; may be NULL), ln_stats[b, l, {mean, rstd}] of y
 * (what adp_ln_affine_fwd(y, ...) would return). */
int adp_modulation_ln_fwd(const float* x, const float* ss, int64_t ss_bstride, int64_t B, int64_t C, int64_t L, float eps,
                          float* y, float* stats, float eps_ln, const float* gamma, const float* beta, float* xn,
                          const float* gamma2, const float* beta2, float* xn2, float* ln_stats, void* stream);
/* dx, and dss[b*dss_bstride + {c | C + c}] = {sum_l dy*xhat | sum_l dy} (overwritten).
 * ws: adp_chan_ln_bwd_ws_bytes(B, C, L). */
int64_t adp_chan_ln_bwd_ws_bytes(int64_t B, int64_t C, int64_t L);
int adp_modulation_bwd(const float* x, const float* dy, const float* ss, int64_t ss_bstride, const float* stats,
                       int64_t B, int64_t C, int64_t L, float* dx, float* dss, int64_t dss_bstride, float* ws,
                       void* stream);
/* The two stages of adp_modulation_bwd separately.  adp_modulation_bwd_partial: dx and the per-tile channel sums in ws
 * (adp_chan_ln_bwd_ws_bytes); returns the tile count NT (> 0) or a negative error.  adp_modulation_bwd_reduce: the second
 * stage of n Modulation backwards of ONE shape (B, C, NT) in one launch per 8 -- ws[i] / dss[i] are host arrays of device
 * pointers; dss_i[b*dss_bstride + {c | C + c}] as above.  The Modulation items of a U-Net depth share their shape, and their
 * scale / shift gradients are first read when the depth's conditioning rows are formed. */
int64_t adp_modulation_bwd_partial(const float* x, const float* dy, const float* ss, int64_t ss_bstride,
                                   const float* stats, int64_t B, int64_t C, int64_t L, float* dx, float* ws,
                                   void* stream);
int adp_modulation_bwd_reduce(const float* const* ws, float* const* dss, int64_t n, int64_t B, int64_t C, int64_t NT,
                              int64_t dss_bstride, void* stream);
/* Backward of adp_modulation_ln_fwd's pair for the `xn` output in one pass: d(xn) -> d(y) (LayerNorm backward with affine
 * `gamma`, statistics ln_stats, + the residual gradient dres, may be NULL) -> d(x) (the Modulation's backward as
 * adp_modulation_bwd_partial: per-tile channel sums to ws, tile count returned, summed later by adp_modulation_bwd_reduce).
 * [dgamma | dbeta] of the LayerNorm to dgamma_dbeta (accumulate != 0: added).  y = the Modulation's output (read only when the
 * tensors do not allow the 16-byte form); ws and ws_ln: adp_chan_ln_bwd_ws_bytes(B, C, L) each. */
int64_t adp_modulation_ln_bwd_partial(const float* x, const float* ss, int64_t ss_bstride, const float* stats,
                                      const float* y, const float* dxn, const float* gamma, const float* ln_stats,
                                      const float* dres, int64_t B, int64_t C, int64_t L, int64_t accumulate, float* dx,
                                      float* ws, float* dgamma_dbeta, float* ws_ln, void* stream);

/* LayerNorm-over-channels statistics only (LayerNorm prologue of the attention projections, components.py:92-93) */
int adp_ln_stats(const float* x, int64_t B, int64_t C, int64_t L, float eps, float* stats, void* stream);
/* Affine LayerNorm over channels, materialised: y = LayerNorm_C(x) * gamma + beta (and, from the same statistics,
 * y2 with gamma2 / beta2 when y2 != NULL: self attention normalises x twice, for q and for k/v -- a_unet Attention's
 * `norm` and `norm_context`, components.py:92).  The projections then run as plain 1x1 convs on the MFMA kernel. */
int adp_ln_affine_fwd(const float* x, int64_t B, int64_t C, int64_t L, float eps, const float* gamma,
                      const float* beta, float* y, const float* gamma2, const float* beta2, float* y2, float* stats,
                      void* stream);
/* Backward of xn = LayerNorm_C(x) * gamma + beta given dxn: dx = ... (+ dres); dgamma_dbeta = [dgamma | dbeta] (2C).
 * ws: adp_chan_ln_bwd_ws_bytes(B, C, L). */
int adp_ln_bwd(const float* x, const float* dxn, const float* stats, const float* gamma, const float* dres,
               int64_t B, int64_t C, int64_t L, int64_t accumulate, float* dx, float* dgamma_dbeta, float* ws,
               void* stream);

/* ------------------------------------------------------------------------------------------
 * Small-batch Linear layers of the conditioning path (TimeConditioningPlugin components.py:74-76,
 * Modulation / MergeModulate `Linear(SiLU(features))`): y[b*y_bstride + n] = post(bias[n] + sum_k act(x[b,k]) w[n,k])
 * act: 0 identity | 1 SiLU | 2 GELU(erf).  post: 0 none | 2 GELU(erf).  B <= 16 rows; the weight matrix is
 * streamed once.  Every Modulation/SkipModulate Linear of a model lives in ONE contiguous weight bank so a
 * single call serves all of them (y = the conditioning bank ss_all [B, NTOT]).
 * ------------------------------------------------------------------------------------------ */
int adp_linear_fwd(const float* x, const float* w, const float* bias, int64_t B, int64_t K, int64_t N, int64_t act,
                   int64_t post, float* y, int64_t y_bstride, void* stream);
/* dxa[b,k] (+)= sum_n dy[b*dy_bstride + n] w[n,k]  (gradient w.r.t. act(x)); ws: adp_linear_bwd_data_ws_bytes */
int64_t adp_linear_bwd_data_ws_bytes(int64_t B, int64_t K, int64_t N);
int adp_linear_bwd_data(const float* dy, int64_t dy_bstride, const float* w, int64_t B, int64_t K, int64_t N,
                        int64_t accumulate, float* dxa, float* ws, void* stream);
/* dw[n,k] (+)= sum_b dy[b,n] act(x[b,k]) ; dbias[n] (+)= sum_b dy[b,n] */
int adp_linear_bwd_weight(const float* dy, int64_t dy_bstride, const float* x, int64_t B, int64_t K, int64_t N,
                          int64_t act, int64_t accumulate, float* dw, float* dbias, void* stream);

/* NumberEmbedder of TimeConditioningPlugin: four[b, :] = [t, sin(2 pi t w), cos(2 pi t w)], w in R^H */
int adp_time_fourier_fwd(const float* t, const float* w, int64_t B, int64_t H, float* four, void* stream);
int adp_time_fourier_bwd(const float* t, const float* w, const float* dfour, int64_t B, int64_t H, int64_t accumulate,
                         float* dw, void* stream);
/* elementwise activation helpers on small [n] vectors: y = act(x); dx (+)= dy * act'(x) */
int adp_act_fwd(const float* x, int64_t n, int64_t act, float* y, void* stream);
int adp_act_bwd(const float* x, const float* dy, int64_t n, int64_t act, int64_t accumulate, float* dx, void* stream);

/* ------------------------------------------------------------------------------------------
 * SkipModulate backward (components.py:99): out = skip + scale[b,c] * x
 *   dx = scale * g ; dscale[b*dscale_bstride + c] = sum_l g * x ; ws: adp_skipmod_bwd_ws_bytes
 * ------------------------------------------------------------------------------------------ */
int64_t adp_skipmod_bwd_ws_bytes(int64_t B, int64_t C, int64_t L);
int adp_skipmod_bwd(const float* g, const float* x, const float* scale, int64_t scale_bstride, int64_t B, int64_t C,
                    int64_t L, float* dx, float* dscale, int64_t dscale_bstride, float* ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * v-objective math (diffusion.py:77-95 and :183-187).  sigma -> alpha = cos(sigma*pi/2), beta = sin(.)
 * ------------------------------------------------------------------------------------------ */
/* x_noisy = a x + b n ; v_target = a n - b x   (sigma per batch element; per = C*L elements each) */
int adp_v_noise(const float* x, const float* noise, const float* sigma, int64_t B, int64_t per, float* x_noisy,
                float* v_target, void* stream);
/* loss = mean((v_pred - v_target)^2) (deterministic two-stage) and dv = 2 (v_pred - v_target) / n * gscale */
int64_t adp_mse_ws_bytes(int64_t n);
int adp_mse_fwd(const float* v_pred, const float* v_target, int64_t n, float* loss, float* ws, void* stream);
int adp_mse_bwd(const float* v_pred, const float* v_target, const float* gloss, int64_t n, float* dv, void* stream);
/* one VSampler step: x <- a1*(a0 x - b0 v) + b1*(b0 x + a0 v); ab4 = device [a0, b0, a1, b1] */
int adp_v_step(const float* x, const float* v, const float* ab4, int64_t n, float* x_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-head attention core (a_unet AttentionBase; components.py:92-93): channel-major operands
 *   q [B, H*D, n], k,v [B, H*D, m]  ->  o [B, H*D, n] = softmax(q^T k * D^-0.5) v   per head
 * lse [B, H, n] (log-sum-exp) is written for the backward pass.
 * ------------------------------------------------------------------------------------------ */
/* ws: scratch of adp_attn_fwd_ws_bytes() bytes or NULL.  With it, small grids (batch 1) split the key range over
 * several waves per query tile and merge the partial softmax results in a second launch (fixed order). */
int adp_attn_fwd(const float* q, const float* k, const float* v, int64_t B, int64_t H, int64_t D, int64_t n,
                 int64_t m, int64_t q_bstride, int64_t kv_bstride, float* o, float* lse, float* ws, void* stream);
int64_t adp_attn_fwd_ws_bytes(int64_t B, int64_t H, int64_t D, int64_t n, int64_t m);
int adp_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                 const float* lse, int64_t B, int64_t H, int64_t D, int64_t n, int64_t m, int64_t q_bstride,
                 int64_t kv_bstride, float* dq, float* dk, float* dv, float* ws, void* stream);
int64_t adp_attn_bwd_ws_bytes(int64_t B, int64_t H, int64_t D, int64_t n, int64_t m);

/* The context side of all CrossAttentionItems of a U-Net as one weight bank (components.py:93: every item projects the SAME
 * embedding through its own LayerNorm + to_kv).  With xhat = LayerNorm-without-affine(context) computed once,
 * kv_i = (W_i diag(gamma_i)) xhat + W_i beta_i: adp_ctx_fold_fwd writes the folded bank w_all [I*M2, E] and bias_all [I*M2]
 * from I per-item tensors (w, gamma, beta: DEVICE arrays of I pointers; W_i is [M2, E] row-major), so that ONE adp_conv1d
 * yields every item's k | v.  adp_ctx_fold_bwd turns the bank's gradients (dw_all, dbias_all from ONE adp_conv1d_wgrad) into
 * dW_i = dW'_i diag(gamma_i) + db'_i beta_i^T at flat[dw_off[i]] and [dgamma_i | dbeta_i] (2E floats) at flat[dgb_off[i]]
 * (dw_off / dgb_off: device arrays of element offsets into the caller's flat gradient buffer). */
int adp_ctx_fold_fwd(const float* const* w, const float* const* gamma, const float* const* beta, int64_t I, int64_t M2,
                     int64_t E, float* w_all, float* bias_all, void* stream);
int adp_ctx_fold_bwd(const float* const* w, const float* const* gamma, const float* const* beta, const float* dw_all,
                     const float* dbias_all, int64_t I, int64_t M2, int64_t E, float* flat, const int64_t* dw_off,
                     const int64_t* dgb_off, void* stream);

/* one VInpainter resample step (diffusion.py:339-350): x_new = mask ? a1*source + b1*noise
 *                                                                  : a1*(a0 x - b0 v) + b1*(b0 x + a0 v);
 * ab4 = device [a_i, b_i, a_j, b_j] (j = i + 1 on the last resample of a step, else i); mask is 1 byte / element */
int adp_v_inpaint_step(const float* x, const float* v, const float* source, const float* noise, const uint8_t* mask,
                       const float* ab4, int64_t n, float* x_out, void* stream);

/* ClassifierFreeGuidancePlugin (components.py:66-69): the guided and the masked evaluation run as ONE [2B, ...]
 * U-Net call; out = y[half:] + (y[:half] - y[half:]) * scale.  adp_select_rows: out[r,:] = pick[r] ? a[r,:] : b[r,:]
 * (training-time embedding mask). */
int adp_cfg_mix(const float* y, int64_t half, float scale, float* out, void* stream);
int adp_select_rows(const float* a, const float* b, const uint8_t* pick, int64_t rows, int64_t per, float* out,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Windowed-sinc polyphase resampler (utils.py:82-109: F.pad + strided conv1d + "(b c) k l -> b c (l k)" + crop):
 *   out[row, l*fo + k] = sum_{j<J} kern[k*J + j] * xpad[row, l*fi + j],  xpad = x with `width` zeros on the left
 *   and width + fi on the right.  rows = B*C, kern = the reference's [fo, 1, J] kernel bank (built by the host with
 *   the reference's own formula), out_len = int(fo * length / fi) (the crop), out is [rows, out_len].
 * ------------------------------------------------------------------------------------------ */
int adp_resample(const float* x, const float* kern, int64_t rows, int64_t length, int64_t fi, int64_t fo, int64_t J,
                 int64_t width, int64_t out_len, float* out, void* stream);

/* y = a + b (n elements); used where two gradient streams meet */
int adp_add(const float* a, const float* b, int64_t n, float* y, void* stream);

/* out = a * x + b * y (y may be NULL: out = a * x).  SkipCat's 2^-1/2 skip branch (a_unet SkipCat, selected by
 * components.py:99 when use_modulation is False) and the accumulation of the embedding gradient over the
 * CrossAttentionItems (components.py:93). */
int adp_axpby(float a, const float* x, float b, const float* y, int64_t n, float* out, void* stream);

/* dst[r*dst_stride + c] = src[r*src_stride + c], r < rows, c < cols: torch.cat / split along channels of [B, C, L]
 * tensors (AppendChannelsPlugin around a net that is not a UNetV0, components.py:174-176). */
int adp_copy2d(const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int64_t rows, int64_t cols,
               void* stream);

/* DownsampleItem / UpsampleItem with a factor other than 1, 2, 4 (a_unet accepts any integer factor,
 * components.py:38; README.md:27) run through two index-only helpers around the 1x1 / k3 convs:
 *   adp_unshuffle: out[(row*f + k), l] = x[row, l*f + k]        ([rows, L] -> [rows*f, L/f], space-to-depth)
 *   adp_pool_sum : out[row, l] = sum_{k<f} x[row, l*f + k] (+ res[row, l])   (gradient of the nearest upsample) */
int adp_unshuffle(const float* x, int64_t rows, int64_t L, int64_t f, float* out, void* stream);
int adp_pool_sum(const float* x, int64_t rows, int64_t Lout, int64_t f, const float* res, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Calibration probes (csrc/probe.hip).  Measurement infrastructure: bench.py times them in the benchmarked process
 * before the timed window so that the line says how fast THIS box streams, multiplies and launches (the reference has
 * no counterpart; GPU boxes of one pool differ by up to 12 % on the whole step).
 *   adp_probe_copy   : dst[i] = src[i], n floats (n % 4 == 0, 16-byte aligned), 16-byte accesses  -> HBM GB/s
 *   adp_probe_mfma   : 512 workgroups x 4 waves, each `iters` rounds of four independent v_mfma_f32_32x32x2_f32 with
 *                      register operands; out receives one float per thread (out_elems >= 512 * 256); returns the
 *                      launch's flops (negative: error code)                                        -> f32 matrix TFLOP/s
 *   adp_probe_launch : an empty kernel of `workgroups` single-wave workgroups                       -> launch gap
 *   adp_probe_chase  : ONE lane follows `steps` dependent loads i = chain[i] from i = 0 (chain: int32 indices forming a cycle
 *                      the host laid out over the working set it wants to probe), out[0] = the final index
 *                                                                                       -> load-to-use latency (L2 / MALL / HBM)
 * ------------------------------------------------------------------------------------------ */
int adp_probe_copy(const float* src, float* dst, int64_t n, void* stream);
int adp_probe_chase(const int32_t* chain, int64_t steps, int32_t* out, void* stream);
int64_t adp_probe_mfma(int64_t iters, float* out, int64_t out_elems, void* stream);
int adp_probe_launch(int64_t workgroups, void* stream);
/* Ceiling search of the two rate probes (bench.py reports the best variant and names it): copy variants 0-7 differ in grid
 * (persistent 1024-8192 workgroups), 16-byte loads in flight per lane (1-8) and nontemporal access, 8 = read only (sums; n
 * floats read), 9 = write only (n floats written); MFMA variants 0-4 in waves per
 * SIMD (1 / 2 / 4) and MFMAs per loop trip (4 / 16 / 32; `iters` % 8 == 0).  Same return conventions as the plain probes. */
int adp_probe_copy_v(const float* src, float* dst, int64_t n, int variant, void* stream);
int64_t adp_probe_mfma_v(int64_t iters, float* out, int64_t out_elems, int variant, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADP_H */
