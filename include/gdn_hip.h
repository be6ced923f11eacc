/*
 * gdn_hip.h — C ABI of libgdn_hip.so, the MI355X (gfx950) implementation of GDN's
 * graph-attention hot path.
 *
 * The reference (SchlomoFeng/GDN) has no FFI: its boundary is the Python nn.Module
 * `GDN.forward(data, org_edge_index)` (models/GDN.py:122-187).  gdn_amd/model.py keeps
 * that Python API and calls the entry points below through ctypes; each entry point
 * replaces the reference lines cited at its declaration.  INTEGRATION.md shows the
 * binding a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - the caller allocates every output; the library owns nothing and keeps no state
 *     between calls (it caches only immutable device properties);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls only
 *     enqueue work, never synchronise, never allocate: they are hipGraph-capturable;
 *   - return value: GDN_OK (0) or a negative GDN_ERR_* code; nothing is thrown;
 *   - tensors are dense, row-major, fp32 unless stated; "BN rows" means batch*n rows in
 *     window-major order (row = b*n + sensor), the layout of models/GDN.py:130;
 *   - re-entrant across distinct streams.
 *
 * Supported shapes: d in {16, 32, 64, 128}; 1 <= w <= 64; 1 <= k <= n with k+1 <= 1024,
 * and the window's working set must fit the 160 KB of LDS of one CU:
 *   forward (staged and fused): the xlin tile (n+1)*dc*4 bytes, dc = d (d = 128: 64, two
 *     column slices) — n up to ~600 at d = 64/128, ~1000 at d = 32, ~2000 at d = 16;
 *   backward (gdn_attn_aggregate_bwd): that tile at full d PLUS two [n, pitch] fp32 tables and the lists
 *     in LDS — n up to ~250 at d = 64 with k = 30 (127-sensor WADI, 51-sensor SWaT, the 25-55-sensor
 *     MSL/SMAP/PSM sets); beyond that the tables go through the caller's workspace in global memory and
 *     only the tile must fit (n <= ~600 at d = 64: the 512-sensor / k = 64 stress shape trains; d = 128 walks
 *     two 64-column slices when the full tile does not fit: n <= ~600 there too);
 *   matrix-core ("dense") kernels — gdn_forward_fused, gdn_project_fwd, gdn_attn_aggregate_fwd pick them
 *     by themselves for n <= 127, d = 64, w <= 32, k <= 63 (gdn_forward_fused also at d = 128); the staged
 *     bf16-storage entry points exist only there;
 * anything else returns GDN_ERR_UNSUPPORTED (never a silent fallback).
 */
#ifndef GDN_HIP_H
#define GDN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDN_OK 0
#define GDN_ERR_ARG (-1)          /* null pointer / non-positive dimension                 */
#define GDN_ERR_LAUNCH (-2)       /* hipGetLastError() != hipSuccess after the launch      */
#define GDN_ERR_UNSUPPORTED (-3)  /* shape outside the supported set above                 */

#define GDN_ABI_VERSION 21
int gdn_abi_version(void);

/* Number of u16 slots per neighbour-list row for a given k: (k+1) rounded up to 16. */
int gdn_nbr_pitch(int k);

/* ---- sensor graph -----------------------------------------------------------------
 * Replaces models/GDN.py:148-159 (cosine matrix, top-k) and the per-forward edge-list
 * build :161-165 + get_batch_edge_index :15-24 + the self-loop strip/append of
 * models/graph_layer.py:61-63.  One neighbour list per sensor, shared by every window:
 *   topk_idx[n,k]  int64, descending cosine, ties -> lower index, NaN ranks highest
 *                  (= model.learned_graph);
 *   nbr[n,pitch]   u16: the top-k entries != i in rank order, then i itself, then
 *                  padding = the sentinel index n; pitch = gdn_nbr_pitch(k);
 *   deg[n]         int32: number of valid entries (k, or k+1 when i is not in its own
 *                  top-k);
 *   cos_out[n,n]   optional (may be NULL) cosine matrix for inspection.               */
int gdn_topk_graph(const float* emb, int n, int d, int k,
                   int64_t* topk_idx, uint16_t* nbr, int32_t* deg, float* cos_out, void* stream);

/* Same neighbour-list build from a GIVEN top-k table (kernel-level parity with an
 * injected graph; also lets a caller reuse a graph learned elsewhere).                */
int gdn_graph_from_topk(const int64_t* topk_idx, int n, int k,
                        uint16_t* nbr, int32_t* deg, void* stream);

/* ---- per-forward constants ----------------------------------------------------------
 * The attention logit of models/graph_layer.py:91-103 is separable:
 *   pi(i<-j) = [xlin_i.att_i + v_i.att_em_i] + [xlin_j.att_j + v_j.att_em_j]
 * and xlin = x.lin^T, so each bracket is  x_row . a + c[sensor]  with
 *   a_i = lin^T att_i  [w],  c_i[s] = v_s . att_em_i  [n]   (same for _j).
 * node_terms receives [a_i(64) | a_j(64) | c_i(n) | c_j(n)] (a_* zero padded to 64).   */
int gdn_node_terms(const float* lin_w, const float* att_i, const float* att_j,
                   const float* att_em_i, const float* att_em_j, const float* emb,
                   int n, int d, int w, float* node_terms, void* stream);

/* gdn_topk_graph + gdn_node_terms as ONE launch (a training step rebuilds both from the parameters it is about
 * to use; they are independent of each other): same outputs, no cos_out.                              */
int gdn_topk_graph_terms(const float* emb, int n, int d, int k, int64_t* topk_idx, uint16_t* nbr,
                         int32_t* deg, const float* lin_w, const float* att_i, const float* att_j,
                         const float* att_em_i, const float* att_em_j, int w, float* node_terms,
                         void* stream);

/* Eval-mode BatchNorm1d folded to y = x*scale + shift (models/GDN.py:77, :179 under
 * model.eval()): scale = weight/sqrt(var+eps), shift = bias - mean*scale.
 * affine receives [scale(c) | shift(c)].                                               */
int gdn_bn_fold(const float* weight, const float* bias, const float* running_mean,
                const float* running_var, float eps, int c, float* affine, void* stream);

/* ---- staged forward (training + inspection path) ------------------------------------
 * gdn_project_fwd: models/graph_layer.py:56 (xlin = x lin^T) plus the two per-node
 * attention scalars.  x[BN,w] -> xlin[BN,d], s_i[BN], s_j[BN].                         */
int gdn_project_fwd(const float* x, const float* lin_w, const float* node_terms,
                    int batch, int n, int w, int d,
                    float* xlin, float* s_i, float* s_j, void* stream);

/* gdn_attn_aggregate_fwd: models/graph_layer.py:65-74,82-117 + PyG propagate / softmax:
 * LeakyReLU(0.2) logits, softmax over each target's incoming edges (max-subtract, exp,
 * /(sum+1e-16)), alpha-weighted sum of source rows, + bias.
 *   z[BN,d]          aggregate incl. bias (the GraphLayer output);
 *   alpha[BN,pitch]  optional (NULL to skip): attention weight of nbr slot p of each
 *                    target row (0 in padding) — att_weight_1 in dense form.           */
int gdn_attn_aggregate_fwd(const float* xlin, const float* s_i, const float* s_j,
                           const uint16_t* nbr, const int32_t* deg, const float* bias,
                           int batch, int n, int d, int k,
                           float* z, float* alpha, void* stream);

/* gdn_head_fwd: models/GDN.py:77-79 (BN+ReLU), :175-180 (x embedding, BN+ReLU), eval
 * dropout = identity (:182), OutLayer with out_layer_num == 1 (:27-56) = Linear(d->1).
 * bn1_affine / bn2_affine come from gdn_bn_fold.  z[BN,d] -> out[BN].
 * h2 (optional, NULL to skip) receives the [BN,d] input of the OutLayer (needed when
 * out_layer_num > 1: the MLP then runs in gdn_mlp_fwd).                                  */
int gdn_head_fwd(const float* z, const float* emb, const float* bn1_affine,
                 const float* bn2_affine, const float* out_w, const float* out_b,
                 int batch, int n, int d, float* out, float* h2, void* stream);

/* ---- OutLayer MLP, out_layer_num > 1 (models/GDN.py:27-56,:183), eval mode -------------
 * h2[rows, d_in] (the h2 output of gdn_head_fwd) -> [Linear(K->hidden), BatchNorm1d(hidden) with
 * running statistics, ReLU] x (layers-1) -> Linear(hidden->1) -> out[rows], one launch, on the
 * 16-bit matrix cores with the fp32-grade two-term split (activations never leave registers).
 * The weights enter through a PLAN (BatchNorm folded, split, reordered), built once per parameter
 * update: gdn_mlp_plan_layer for hidden layer `layer` = 0 .. layers-2 (weight[hidden, K],
 * K = d_in for layer 0, hidden otherwise; bias[hidden]; the BatchNorm1d behind it), then
 * gdn_mlp_plan_out for the final Linear (weight[hidden], bias[1]).
 * Supported: layers 2..8, hidden <= 256, d_in in {16,32,64,128}; gdn_mlp_plan_bytes returns 0
 * otherwise.                                                                                */
long long gdn_mlp_plan_bytes(int d_in, int hidden, int layers);
int gdn_mlp_plan_layer(const float* weight, const float* bias, const float* bn_weight,
                       const float* bn_bias, const float* bn_mean, const float* bn_var, float eps,
                       int d_in, int hidden, int layers, int layer, void* plan, void* stream);
int gdn_mlp_plan_out(const float* weight, const float* bias, int d_in, int hidden, int layers,
                     void* plan, void* stream);
int gdn_mlp_fwd(const float* h2, const void* plan, int rows, int d_in, int hidden, int layers,
                float* out, void* stream);

/* ---- train-mode head (out_layer_num == 1) -------------------------------------------
 * gdn_head_train_fwd: the same chain as gdn_head_fwd under model.train(): both BatchNorms
 * normalise by the statistics of this batch (models/GDN.py:77-79 GNNLayer.bn, :178-180
 * bn_outlayer_in — biased variance over all batch*n rows) and update running_mean /
 * running_var (momentum, unbiased variance) / num_batches_tracked; dropout (:182) is applied
 * as the caller's mask: either mask[BN,d] fp32 multipliers (0 or 1/(1-p)) or keep[BN,d] bytes
 * (1 kept / 0 dropped, multiplier = keep * keep_scale: a quarter of the traffic); both NULL =
 * no dropout; OutLayer Linear(d->1).
 *   stats (out)  gdn_head_train_stats_bytes(d) bytes: the column sums of z, z^2, h1, h1^2 — opaque,
 *                kept for the backward.  Sums across workgroups are EXACT (260-bit fixed point, 64-bit
 *                integer atomics, see gdn_exact_sum): statistics and gradients do not depend on the
 *                order the workgroups finish in.
 *   running_* / batches*     may be NULL (track_running_stats off).
 * Three streaming passes over z; nothing [BN,d]-sized is stored.  batch*n >= 2.             */
long long gdn_head_train_stats_bytes(int d);
/* The accumulator of the training statistics on its own: out[0] = the sum of x[0..count) (device fp64, count <=
 * 2048), exact up to the final conversion to fp64 (values are held to 2^-130, NaN / inf / |x| >= 2^130 give NaN).
 * workspace: gdn_exact_sum_workspace_bytes() bytes, zero on entry, left zero.  Exported for tests.          */
long long gdn_exact_sum_workspace_bytes(void);
int gdn_exact_sum(const double* x, int count, void* workspace, double* out, void* stream);
int gdn_head_train_fwd(const float* z, const float* emb, const float* bn1_w, const float* bn1_b,
                       const float* bn2_w, const float* bn2_b, const float* lin_w,
                       const float* lin_b, const float* mask, const uint8_t* keep,
                       float keep_scale, int batch, int n, int d,
                       float eps1, float eps2, float momentum1, float momentum2,
                       float* running_mean1, float* running_var1, long long* batches1,
                       float* running_mean2, float* running_var2, long long* batches2,
                       double* stats, float* out, void* stream);

/* gdn_head_train_bwd: gradients of gdn_head_train_fwd given d_out[BN] (what autograd derives
 * for models/GDN.py:77-79,:175-184 in training): d_z[BN,d], d_emb[n,d] (the head's share of the
 * embedding gradient), BatchNorm weight/bias gradients [d], OutLayer Linear d_lin_w[d] /
 * d_lin_b[1].  workspace: gdn_head_train_workspace_bytes(n, d) bytes of scratch.            */
long long gdn_head_train_workspace_bytes(int n, int d);
int gdn_head_train_bwd(const float* d_out, const float* z, const float* emb, const float* bn1_w,
                       const float* bn1_b, const float* bn2_w, const float* bn2_b,
                       const float* lin_w, const float* mask, const uint8_t* keep,
                       float keep_scale, const double* stats,
                       int batch, int n, int d, float eps1, float eps2, double* workspace,
                       float* d_z, float* d_emb, float* d_bn1_w, float* d_bn1_b, float* d_bn2_w,
                       float* d_bn2_b, float* d_lin_w, float* d_lin_b, void* stream);

/* The same two passes with the dropout mask of models/GDN.py:114,182 DRAWN INSIDE the kernels instead of
 * read from a tensor: element e of training step t is dropped iff mix32(e, seed, t) < p_drop * 2^32, kept
 * values are scaled by 1/(1-p_drop).  rng_seed_step = {seed, t} (two int64 in device memory); the forward
 * and the backward of one step must see the same pair (gdn_adam_step increments t after the backward).
 * Stateless draw: nothing [BN,d]-sized is written or read for the mask.  It is NOT torch's Philox stream:
 * reproducing a given torch mask needs the mask/keep arguments of the plain entry points.             */
int gdn_head_train_fwd_rng(const float* z, const float* emb, const float* bn1_w, const float* bn1_b,
                           const float* bn2_w, const float* bn2_b, const float* lin_w,
                           const float* lin_b, const long long* rng_seed_step, float p_drop,
                           int batch, int n, int d, float eps1, float eps2, float momentum1,
                           float momentum2, float* running_mean1, float* running_var1,
                           long long* batches1, float* running_mean2, float* running_var2,
                           long long* batches2, double* stats, float* out, int buffers_zeroed,
                           void* stream);
int gdn_head_train_bwd_rng(const float* d_out, const float* z, const float* emb, const float* bn1_w,
                           const float* bn1_b, const float* bn2_w, const float* bn2_b,
                           const float* lin_w, const long long* rng_seed_step, float p_drop,
                           const double* stats, int batch, int n, int d, float eps1, float eps2,
                           double* workspace, float* d_z, float* d_emb, float* d_bn1_w, float* d_bn1_b,
                           float* d_bn2_w, float* d_bn2_b, float* d_lin_w, float* d_lin_b,
                           int buffers_zeroed, void* stream);
/* gdn_head_train_fwd_rng with the loss folded in (train.py:20-23,70-72: F.mse_loss + the first step of
 * loss.backward()): the last forward pass also writes d_out = 2 (out - y) / (batch n) and loss[0] =
 * mean((out - y)^2) — per-workgroup fp64 partials added in a fixed order by the last workgroup to finish, as
 * gdn_mse_loss_grad does, without its launch.  mse_workspace: gdn_head_mse_workspace_bytes(), zero-filled once.
 * (Measured on MI355X at 512 windows: 5 us slower than the separate launch — every one of the 512 workgroups
 * pays the block reduction and the ticket; harness.NativeTrainStep uses it only with GDN_FUSE_MSE=1.)        */
long long gdn_head_mse_workspace_bytes(void);
int gdn_head_train_fwd_rng_mse(const float* z, const float* emb, const float* bn1_w, const float* bn1_b,
                               const float* bn2_w, const float* bn2_b, const float* lin_w,
                               const float* lin_b, const long long* rng_seed_step, float p_drop,
                               int batch, int n, int d, float eps1, float eps2, float momentum1,
                               float momentum2, float* running_mean1, float* running_var1,
                               long long* batches1, float* running_mean2, float* running_var2,
                               long long* batches2, double* stats, float* out, const float* y,
                               double* mse_workspace, float* loss, float* d_out, int buffers_zeroed,
                               void* stream);
/* buffers_zeroed (the _rng and _act entry points; bit 1 of gdn_head_train_bwd_rng's: see gdn_train_finish): 0 = the call zero-fills its accumulators itself (a
 * memset launch each in forward and backward); 1 = the caller guarantees `stats` (forward) / the first
 * gdn_head_train_stats_bytes-style block of `workspace` (backward) are zero on entry, and the backward
 * leaves BOTH zeroed again when it finishes — a training loop that zero-fills them once never pays the two
 * memset launches (~5 us each at 512 windows).                                                          */

/* ---- train-mode OutLayer MLP, out_layer_num > 1 (models/GDN.py:27-56 under model.train()) ----------
 * The head passes above end at the [B*n, d] activation after dropout (forward: `act`) / start from its
 * gradient (backward: `d_act`) instead of the fused Linear(d -> 1); mask / keep as in gdn_head_train_fwd,
 * or (both null, rng_seed_step given, p_drop > 0) the in-kernel draw of the _rng entry points.          */
int gdn_head_train_fwd_act(const float* z, const float* emb, const float* bn1_w, const float* bn1_b,
                           const float* bn2_w, const float* bn2_b, const float* mask,
                           const uint8_t* keep, float keep_scale, const long long* rng_seed_step,
                           float p_drop, int batch, int n, int d, float eps1,
                           float eps2, float momentum1, float momentum2, float* running_mean1,
                           float* running_var1, long long* batches1, float* running_mean2,
                           float* running_var2, long long* batches2, double* stats, float* act,
                           int buffers_zeroed, void* stream);
int gdn_head_train_bwd_act(const float* d_act, const float* z, const float* emb, const float* bn1_w,
                           const float* bn1_b, const float* bn2_w, const float* bn2_b,
                           const float* mask, const uint8_t* keep, float keep_scale,
                           const long long* rng_seed_step, float p_drop,
                           const double* stats, int batch, int n, int d, float eps1, float eps2,
                           double* workspace, float* d_z, float* d_emb, float* d_bn1_w, float* d_bn1_b,
                           float* d_bn2_w, float* d_bn2_b, int buffers_zeroed, void* stream);
/* The MLP itself: Y_l = A_l W_l^T + b_l, A_{l+1} = relu(BatchNorm_train(Y_l)) for l = 0..layers-2 (A_0 = act
 * [rows, d_in]), out = A_{layers-1} w_o + b_o — replaces OutLayer.forward (models/GDN.py:47-56) and the
 * autograd graph behind train.py:72.  fp32 matrix cores (exact fp32 products), batch statistics and every
 * reduction in fp64 in a fixed order (bitwise reproducible).  params[4*l .. 4*l+3] = {W_l [hidden, K_l],
 * b_l, gamma_l, beta_l} (K_0 = d_in, K_l = hidden), running[2*l .. 2*l+1] = {running_mean, running_var} (null
 * = not tracked), batches[l] = num_batches_tracked or null; eps[l], momentum[l]: host floats.  The arrays
 * of pointers live in HOST memory and hold device pointers.  `saved` (gdn_mlp_train_saved_bytes) carries the
 * pre-BatchNorm outputs and the batch constants from the forward to the backward; `workspace`
 * (gdn_mlp_train_workspace_bytes) is scratch.  grads[4*l .. 4*l+3] = gradients in the layout of params.
 * Supported: layers 2..8, d_in a multiple of 4 up to 256, hidden 1..512 (widths that are not a multiple of 4
 * stage their operands element by element); else GDN_ERR_UNSUPPORTED and byte counts of 0.                                                                                   */
long long gdn_mlp_train_saved_bytes(int rows, int d_in, int hidden, int layers);
long long gdn_mlp_train_workspace_bytes(int rows, int d_in, int hidden, int layers);
int gdn_mlp_train_fwd(const float* act, const float* const* params, float* const* running,
                      long long* const* batches, const float* eps, const float* momentum,
                      const float* out_w, const float* out_b, int rows, int d_in, int hidden,
                      int layers, void* saved, void* workspace, float* out, void* stream);
int gdn_mlp_train_bwd(const float* d_out, const float* act, const float* const* params,
                      const float* out_w, int rows, int d_in, int hidden, int layers,
                      const void* saved, void* workspace, float* const* grads, float* d_out_w,
                      float* d_out_b, float* d_act, void* stream);

/* gdn_mlp_eval_fwd: the OutLayer MLP under model.eval() (models/GDN.py:45-56: Linear, BatchNorm on the RUNNING
 * statistics, ReLU per hidden layer, Linear(hidden -> 1)) for the widths the one-launch chain gdn_mlp_fwd does not
 * take (hidden > 256; up to 512 = the reference class's default inter_num): one fp32 matrix-core GEMM per hidden
 * layer, the BatchNorm + ReLU folded into the next GEMM's operand staging, a column pass for the last Linear.
 * params / running as gdn_mlp_train_fwd (host arrays of device pointers; running must be non-null here).
 * hidden 1..512, d_in a multiple of 4; workspace: gdn_mlp_eval_workspace_bytes (0 = unsupported shape).    */
long long gdn_mlp_eval_workspace_bytes(int rows, int d_in, int hidden, int layers);
int gdn_mlp_eval_fwd(const float* act, const float* const* params, const float* const* running,
                     const float* eps, const float* out_w, const float* out_b,
                     int rows, int d_in, int hidden, int layers, void* workspace, float* out, void* stream);

/* One launch less at the end of a training step: gdn_head_train_bwd_rng with (buffers_zeroed | 2) leaves out its
 * small finishing launch, gdn_project_bwd_partials is gdn_project_bwd without its reduce launch (*rows_out =
 * partial rows written to `workspace`), and gdn_train_finish runs both reductions as ONE launch (independent
 * workgroups, disjoint outputs).  Same results as the separate forms.                                       */
int gdn_project_bwd_partials(const float* x, const float* d_xlin, const float* d_si, const float* d_sj,
                             int batch, int n, int w, int d, float* workspace, int* rows_out,
                             void* stream);
int gdn_train_finish(double* head_workspace, double* stats, int head_zeroed, int batch, int n, int d,
                     float* d_emb, float* d_bn1_w, float* d_bn1_b, float* d_bn2_w, float* d_bn2_b,
                     float* d_lin_w, float* d_lin_b, const float* proj_workspace, int proj_rows, int w,
                     float* d_proj_w, float* d_a, float* d_c, void* stream);

/* gdn_adam_step: torch.optim.Adam(lr, betas, eps, weight_decay) of train.py:31,73 over ONE flat fp32
 * buffer holding every parameter back to back (params / grads / exp_avg / exp_avg_sq: [count]); step[0] =
 * steps taken so far (device memory), incremented by the call.  grads are multiplied by grad_scale first
 * (1/ranks after a summing all-reduce).  Same update, in the same operation order, as torch's
 * single-tensor Adam (amsgrad and maximize off).  grads[zero_from, zero_from + zero_count) is cleared
 * after use (zero_count = 0: nothing is cleared; gdn_attn_aggregate_bwd WRITES its d_bias since ABI 20). */
int gdn_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                  long long* step, int count, double lr, double beta1, double beta2, double eps,
                  double weight_decay, double grad_scale, int zero_from, int zero_count, void* stream);

/* gdn_mse_loss_grad: train.py:20-23 `F.mse_loss(out, y, reduction='mean')` and the gradient
 * autograd derives for it, d_out = 2 (out - y) / count, in one launch (fp64 accumulation,
 * bitwise reproducible).  workspace: gdn_mse_workspace_bytes() bytes, ZEROED ONCE by the
 * caller when allocated; every call leaves it zeroed.  Calls sharing a workspace must be
 * stream-ordered.  loss[1], d_out[count].                                                   */
long long gdn_mse_workspace_bytes(void);
int gdn_mse_loss_grad(const float* out, const float* y, long long count, double* workspace,
                      float* loss, float* d_out, void* stream);

/* ---- fused eval forward (the throughput path) ---------------------------------------
 * Everything from x[batch,n,w] to out[batch,n] in one launch (one workgroup per window,
 * xlin tile and neighbour lists resident in LDS): models/GDN.py:122-187 under
 * model.eval() with out_layer_num == 1.  HBM traffic = x in + out.                      */
int gdn_forward_fused(const float* x, const float* lin_w, const float* node_terms,
                      const uint16_t* nbr, const int32_t* deg, const float* gnn_bias,
                      const float* emb, const float* bn1_affine, const float* bn2_affine,
                      const float* out_w, const float* out_b,
                      int batch, int n, int w, int d, int k, float* out, void* stream);

/* Same forward fed from the RAW series instead of materialised windows: series[n, series_len]
 * fp32 (the [node, time] layout of datasets/TimeDataset.py:42); window b of the launch is
 * series[:, first+b : first+b+w] (TimeDataset.py:46-49 with stride 1, the test-mode loop), its
 * target column is first+b+w.  Requires first + batch - 1 + w <= series_len.  Needs d >= 32 and
 * w <= 32 (the matrix-core projection variants); otherwise GDN_ERR_UNSUPPORTED.          */
int gdn_forward_fused_series(const float* series, int series_len, int first, const float* lin_w,
                             const float* node_terms, const uint16_t* nbr, const int32_t* deg,
                             const float* gnn_bias, const float* emb, const float* bn1_affine,
                             const float* bn2_affine, const float* out_w, const float* out_b,
                             int batch, int n, int w, int d, int k, float* out, void* stream);

/* gdn_graph_bank_order: a copy of the neighbour lists with every row's entries permuted inside its two halves
 * (slots [0, pitch/2) and [pitch/2, pitch)) so that the staged matrix-core kernels' per-slot LDS accesses (the
 * s_j gather, the two scatters into the attention image) spread over the banks: softmax and aggregation do not
 * depend on the order of a target's slots, so gdn_attn_aggregate_fwd[_bf16] may be given this table instead of
 * `nbr` whenever the caller does not read alpha in rank order (z is the same to the summation order of the
 * softmax denominator).  Matrix-core shapes only (n <= 127, k <= 63); once per graph, one small launch.      */
int gdn_graph_bank_order(const uint16_t* nbr, int n, int k, uint16_t* nbr_ordered, void* stream);

/* ---- plans: the fused forward with its per-launch constants precomputed -------------
 * For shapes on the matrix-core path (n <= 127, d = 64 or 128, w <= 32, k <= 63) everything a
 * workgroup of gdn_forward_fused derives from the parameters and the sensor graph (list
 * offsets, split weight operands, folded BatchNorm / embedding factors) can be computed
 * once per parameter update into a caller-owned device buffer, the PLAN; launches that are
 * given it skip that prologue (~10 us per workgroup), which is most of the time of a
 * single-minibatch launch.  The plan is read-only for the launches and holds no pointers.
 *   gdn_fused_plan_bytes   size of the plan for this shape (0 = shape not on this path:
 *                          use gdn_forward_fused);
 *   gdn_fused_plan_build   same parameter arguments as gdn_forward_fused;
 *   gdn_forward_fused_plan / _series_plan   = gdn_forward_fused(_bf16) /
 *                          gdn_forward_fused_series on the plan's parameters; x is
 *                          fp32 [batch,n,w] (bf16_storage = 0) or bf16 (1, must match the
 *                          plan).  Results are bit-identical to the plan-less calls.        */
long long gdn_fused_plan_bytes(int n, int w, int d, int k, int bf16_storage);
int gdn_fused_plan_build(const float* lin_w, const float* node_terms, const uint16_t* nbr,
                         const int32_t* deg, const float* gnn_bias, const float* emb,
                         const float* bn1_affine, const float* bn2_affine, const float* out_w,
                         const float* out_b, int n, int w, int d, int k, int bf16_storage,
                         void* plan, void* stream);
int gdn_forward_fused_plan(const void* x, const void* plan, int batch, int n, int w, int d, int k,
                           int bf16_storage, float* out, int* range_guard, void* stream);
int gdn_forward_fused_series_plan(const float* series, int series_len, int first, const void* plan,
                                  int batch, int n, int w, int d, int k, float* out, int* range_guard,
                                  void* stream);

/* ---- range guard ---------------------------------------------------------------------
 * The reference computes in fp32 on whatever the caller feeds it (models/graph_layer.py:56).  The fp32-storage
 * matrix-core kernels carry x, and the BatchNorm-folded projected tile times 8, as two f16 terms each: values
 * of 65504 and beyond do not exist there.  Every plan therefore holds its X LIMIT, the largest |x| for which
 * both are representable whatever the window: min(60000, (60000 - max_c |C-in[c]|) / max_c sum_w |lin'[c,w]|)
 * (a float at byte gdn_fused_plan_limit_offset(...) of the plan; +inf for bf16 storage, whose terms have
 * fp32's exponent range).  Normalised data (the reference's scripts/process_*.py: MinMax to [0, 1]) sits four
 * orders of magnitude below it; raw engineering units may not.  Two ways to stay exact for ANY input:
 *   (a) known data (a resident series): compare max|x| with the limit once and call the `_gated` entry points
 *       below with guard = null — the fp32 row-gather kernels, fp32's own range — when it is exceeded;
 *   (b) unknown data (GDN.forward on a caller's tensor): pass `range_guard` (2 ints, ZERO before the first
 *       call) to the planned launch — it stores 1 in range_guard[0] when a window holds |x| >= limit or a
 *       NaN — and follow it on the same stream with the gated launch on the same guard: a no-op (every
 *       workgroup returns at once) unless the flag is up, in which case it recomputes the launch's windows
 *       in fp32 and leaves the guard zeroed.  No host synchronisation, hipGraph-capturable; one guard per
 *       stream.  range_guard = null: no detection (the `_keys` launches never detect: use (a)).
 * The staged kernels have `_wide` twins that always take the row-gather path (training on raw-unit data:
 * harness.train / python -m gdn_amd.main compare the data they hold with GDN_WIDE_LIMIT once).          */
long long gdn_fused_plan_limit_offset(int n, int w, int d, int k, int bf16_storage);
int gdn_forward_fused_gated(int* guard, const float* x, const float* lin_w, const float* node_terms,
                            const uint16_t* nbr, const int32_t* deg, const float* gnn_bias,
                            const float* emb, const float* bn1_affine, const float* bn2_affine,
                            const float* out_w, const float* out_b,
                            int batch, int n, int w, int d, int k, float* out, void* stream);
int gdn_forward_fused_series_gated(int* guard, const float* series, int series_len, int first,
                                   const float* lin_w, const float* node_terms, const uint16_t* nbr,
                                   const int32_t* deg, const float* gnn_bias, const float* emb,
                                   const float* bn1_affine, const float* bn2_affine, const float* out_w,
                                   const float* out_b, int batch, int n, int w, int d, int k, float* out,
                                   void* stream);
int gdn_project_fwd_wide(const float* x, const float* lin_w, const float* node_terms,
                         int batch, int n, int w, int d,
                         float* xlin, float* s_i, float* s_j, void* stream);
int gdn_attn_aggregate_fwd_wide(const float* xlin, const float* s_i, const float* s_j,
                                const uint16_t* nbr, const int32_t* deg, const float* bias,
                                int batch, int n, int d, int k,
                                float* z, float* alpha, void* stream);
int gdn_attn_aggregate_bwd_wide(const float* d_z, const float* xlin, const float* alpha,
                                const float* s_i, const float* s_j,
                                const uint16_t* nbr, const uint32_t* rent, const int32_t* rlen,
                                int batch, int n, int d, int k,
                                float* d_xlin, float* d_si, float* d_sj, float* d_bias,
                                float* workspace, void* stream);

/* The two planned launches with the scoring hand-off folded into their epilogue: besides out[B, n] they leave
 * keys[sensor * key_pitch + b] = |out[b][sensor] - gt[b][sensor]| in float64 — the radix keys
 * gdn_score_select consumes (evaluate.py:48-50, util/data.py:75-82), which gdn_score_keys would otherwise
 * produce in a launch of its own (transposing 2 x 4 bytes in, 8 bytes out per value).  key_pitch >= batch;
 * slots batch..key_pitch-1 of every row are left untouched (pre-fill them with the all-ones filler when
 * key_pitch > batch).  Measured on MI355X: the 127 scattered 8-byte stores per window cost the launch more
 * (+40 us per 32768 windows) than the separate transposing kernel (14 us); harness.SeriesEvaluator uses
 * them only on request (GDN_FUSE_KEYS=1).                                                                 */
int gdn_forward_fused_plan_keys(const void* x, const void* plan, const float* gt, double* keys,
                                int key_pitch, int batch, int n, int w, int d, int k, int bf16_storage,
                                float* out, void* stream);
int gdn_forward_fused_series_plan_keys(const float* series, int series_len, int first, const void* plan,
                                       const float* gt, double* keys, int key_pitch, int batch, int n,
                                       int w, int d, int k, float* out, void* stream);

/* ---- bf16 STORAGE variants (BASELINE.json configs[2] / configs[4]) --------------------
 * Same arithmetic as the four forward entry points above with the windowed inputs x, the
 * projected features xlin and the aggregate z held in bfloat16 IN HBM (uint16_t = the raw
 * bf16 bit pattern, round-to-nearest-even where a value is stored); attention scalars,
 * logits, softmax, every accumulation, BatchNorm and the outputs stay fp32
 * (models/graph_layer.py:56,87-117 with 2-byte features).  Semantics, exactly:
 *   x     is given in bf16 (exact);
 *   xlin  = bf16( x . lin^T accumulated in fp32 )            [stored / LDS resident];
 *   s_i, s_j from the UNROUNDED projection (x_row . a + c, fp32);
 *   z     = sum_j alpha_ij xlin_j + bias in fp32; gdn_attn_aggregate_fwd_bf16 stores bf16(z),
 *           gdn_forward_fused_bf16 keeps z on chip in fp32;
 *   head  as gdn_head_fwd.
 * The three staged entry points: matrix-core path only (n <= 127, d = 64, w <= 32,
 * k <= 63; other shapes return GDN_ERR_UNSUPPORTED).  gdn_forward_fused_bf16 takes every
 * shape gdn_forward_fused takes: outside the matrix-core path the fp32 row-gather kernel
 * reads the bf16 windows and rounds its LDS-resident projected tile to bf16 (configs[4]:
 * 512 sensors, top-k 64, W = 30).
 * RANGE: with fp32 STORAGE the matrix-core kernels carry x, and the BatchNorm-folded features
 * 8*(scale1*xlin + shift), as two f16 terms each: values of 65504 and beyond do not exist there.
 * The reference takes any fp32 (models/graph_layer.py:56; its main.py normalises nothing, the
 * offline scripts/process_*.py do), so see "range guard" above: planned launches detect
 * out-of-range windows on the device, the `_wide` / `_gated` entry points are the fp32 row-gather
 * kernels with fp32's own range.  With bf16 storage x is ONE exact bf16 term with fp32's exponent
 * range and the projected tile is rounded to bf16: no such limit.                            */
int gdn_project_fwd_bf16(const uint16_t* x, const float* lin_w, const float* node_terms,
                         int batch, int n, int w, int d,
                         uint16_t* xlin, float* s_i, float* s_j, void* stream);
int gdn_attn_aggregate_fwd_bf16(const uint16_t* xlin, const float* s_i, const float* s_j,
                                const uint16_t* nbr, const int32_t* deg, const float* bias,
                                int batch, int n, int d, int k,
                                uint16_t* z, float* alpha, void* stream);
int gdn_head_fwd_bf16(const uint16_t* z, const float* emb, const float* bn1_affine,
                      const float* bn2_affine, const float* out_w, const float* out_b,
                      int batch, int n, int d, float* out, float* h2, void* stream);
int gdn_forward_fused_bf16(const uint16_t* x, const float* lin_w, const float* node_terms,
                           const uint16_t* nbr, const int32_t* deg, const float* gnn_bias,
                           const float* emb, const float* bn1_affine, const float* bn2_affine,
                           const float* out_w, const float* out_b,
                           int batch, int n, int w, int d, int k, float* out, void* stream);

/* ---- backward (training) -------------------------------------------------------------
 * Gradients of gdn_attn_aggregate_fwd and gdn_project_fwd; the autograd graph of
 * `loss.backward()` at train.py:72 restricted to the GraphLayer.  No gradient flows into
 * the neighbour lists (the graph is built from a detached embedding, models/GDN.py:145).
 *   d_xlin[BN,d]  gradient of the message term (gathered through the reverse lists: no
 *                 scatter atomics, bitwise reproducible);
 *   d_si, d_sj    grads of the per-node scalars;
 *   d_bias[d]     column sums of d_z, WRITTEN (not accumulated): every workgroup leaves one partial row in
 *                 the workspace and the last one to finish adds the rows in row order — no floating-point
 *                 atomics anywhere in the backward, so a training step is bitwise reproducible (the true
 *                 gradient of a bias in front of a train-mode BatchNorm is 0: what arrives here is rounding
 *                 noise, which Adam turns into +-lr steps — with atomics those steps differed run to run).
 *   workspace     gdn_attn_aggregate_bwd_workspace_bytes(batch, n, d, k) bytes, REQUIRED: [16 bytes: the
 *                 ticket, ZERO before the first call, left zero by every call][1024 x d floats: the d_bias
 *                 rows][batch*n*pitch floats of d_pi when the tile plus the two [n, pitch] tables exceed LDS
 *                 (n ~> 250 at d = 64, k = 30; the 512-sensor / k = 64 stress shape) — the tables then go
 *                 through global memory and only the tile must fit: (n+1)*d*4 bytes <= ~155 KB].  One
 *                 workspace per stream: two concurrent calls must not share one.                      */
long long gdn_attn_aggregate_bwd_workspace_bytes(int batch, int n, int d, int k);
int gdn_attn_aggregate_bwd(const float* d_z, const float* xlin, const float* alpha,
                           const float* s_i, const float* s_j,
                           const uint16_t* nbr, const uint32_t* rent, const int32_t* rlen,
                           int batch, int n, int d, int k,
                           float* d_xlin, float* d_si, float* d_sj, float* d_bias,
                           float* workspace, void* stream);

/* Shapes on the matrix-core path (n <= 127, d = 64, k <= 63) run the backward as two dense products per
 * window (G = dZ . X^T for d alpha, dX = A^T . dZ for the reverse gather; d_z scaled per window by a power of
 * two, every factor as two f16 terms) and do not read the reverse lists: gdn_attn_aggregate_bwd_uses_reverse
 * returns 0 there, rent / rlen may be null and gdn_graph_reverse need not run.                         */
int gdn_attn_aggregate_bwd_uses_reverse(int n, int d, int k);

/* 1 when every kernel of a training step (gdn_project_fwd, gdn_attn_aggregate_fwd, gdn_attn_aggregate_bwd,
 * gdn_project_bwd, gdn_terms_bwd) takes the shape, 0 otherwise: ask before capturing a step.           */
int gdn_train_supported(int n, int w, int d, int k);

/* Reverse neighbour lists for the backward gather: rent[n, gdn_rev_pitch(n)] u32 holds, for
 * source j, (target << 16 | slot) of every list entry that names j, ascending target;
 * rlen[n] their counts.  Built once per graph.                                           */
int gdn_rev_pitch(int n);
int gdn_graph_reverse(const uint16_t* nbr, const int32_t* deg, int n, int k,
                      uint32_t* rent, int32_t* rlen, void* stream);

/* Backward of gdn_project_fwd, i.e. of `x = self.lin(x)` (models/graph_layer.py:56) and of the
 * folded logit scalars (graph_layer.py:94-104):
 * x[BN,w], d_xlin[BN,d], d_si/d_sj[BN] -> d_lin_w[d,w] (direct term), d_a[2,64]
 * (grads of a_i, a_j), d_c[2,n] (grads of c_i, c_j); outputs are written, not accumulated.
 * workspace: gdn_project_bwd_workspace_bytes(n, w, d) bytes (one partial row per workgroup). */
long long gdn_project_bwd_workspace_bytes(int n, int w, int d);
int gdn_project_bwd(const float* x, const float* d_xlin, const float* d_si, const float* d_sj,
                    int batch, int n, int w, int d, float* workspace,
                    float* d_lin_w, float* d_a, float* d_c, void* stream);

/* gdn_terms_bwd: chain rule through gdn_node_terms (a = lin^T att, c = emb . att_em — the
 * separable form of models/graph_layer.py:90-104): from gdn_project_bwd's d_a[2,64] / d_c[2,n]
 *   d_lin_w[d,w] += att_i (x) d_a[0] + att_j (x) d_a[1]      (in/out: holds the direct term)
 *   d_att_i/j[d]  = lin_w d_a[0/1]     d_att_em_i/j[d] = emb^T d_c[0/1]
 *   d_emb[n,d]    = d_c[0] (x) att_em_i + d_c[1] (x) att_em_j                               */
int gdn_terms_bwd(const float* lin_w, const float* att_i, const float* att_j,
                  const float* att_em_i, const float* att_em_j, const float* emb,
                  const float* d_a, const float* d_c, int n, int d, int w, float* d_lin_w,
                  float* d_att_i, float* d_att_j, float* d_att_em_i, float* d_att_em_j,
                  float* d_emb, void* stream);

/* ---- anomaly scoring -----------------------------------------------------------------
 * evaluate.py:48-68 + util/data.py:75-82 + the max over sensors of evaluate.py:131-139,
 * in float64 like the reference.  pred, gt: fp32 [t,n] (time-major, as test.py returns).
 *   gdn_score_quantiles: per sensor, median and IQR (numpy 'linear' percentiles 25/75)
 *     of |pred-gt| over all t ticks -> med_iqr[n,2] float64.  workspace: device buffer of
 *     gdn_score_workspace_bytes(t, n) bytes (8-byte aligned).
 *   gdn_score_smooth_max: a=(|pred-gt|-med)/(|iqr|+1e-2); 4-tap causal mean (first 3
 *     ticks of the SERIES 0); `first_tick` = series index of row 0 of this shard; when it is
 *     > 0, halo_pred/halo_gt [3,n] hold the 3 rows before it (right aligned; rows that would
 *     precede tick 0 are never read); scores[n,t] float64 (optional, NULL to skip) and
 *     anomaly[t] float64 = max over sensors.                                            */
long long gdn_score_workspace_bytes(int t, int n);
/* The two halves of gdn_score_quantiles, separable for the multi-GPU exchange (each rank builds the
 * keys of its own ticks, rows are exchanged by sensor, the owner selects over every rank's block):
 *   gdn_score_keys:   keys[n, pitch] float64 = |pred-gt| transposed; slots t..pitch-1 hold a filler
 *                     (bit pattern of all ones) that the select ignores.  pitch >= t.
 *   gdn_score_select: keys[blocks, n, pitch] (block r = the rows received from rank r), `total` real
 *                     keys per sensor -> med_iqr[n,2].  The input is not modified.  blocks > 1
 *                     needs pitch % 2048 == 0.  workspace: gdn_score_select_workspace_bytes bytes. */
long long gdn_score_select_workspace_bytes(int blocks, int n, int pitch);
int gdn_score_keys(const float* pred, const float* gt, int t, int n, int pitch, double* keys, void* stream);
int gdn_score_select(const double* keys, int blocks, int n, int pitch, long long total,
                     double* workspace, double* med_iqr, void* stream);
int gdn_score_quantiles(const float* pred, const float* gt, int t, int n,
                        double* workspace, double* med_iqr, void* stream);
int gdn_score_smooth_max(const float* pred, const float* gt, const double* med_iqr,
                         int t, int n, int first_tick, const float* halo_pred,
                         const float* halo_gt, double* scores, double* anomaly, void* stream);

/* gdn_terms_bwd with accumulate_emb != 0: d_emb += (instead of =) — the head's share of the embedding
 * gradient already sits in d_emb (gdn_head_train_bwd), so both land in one gradient slot without an add
 * kernel.                                                                                              */
int gdn_terms_bwd_acc(const float* lin_w, const float* att_i, const float* att_j,
                      const float* att_em_i, const float* att_em_j, const float* emb,
                      const float* d_a, const float* d_c, int n, int d, int w, float* d_lin_w,
                      float* d_att_i, float* d_att_j, float* d_att_em_i, float* d_att_em_j,
                      float* d_emb, int accumulate_emb, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDN_HIP_H */
