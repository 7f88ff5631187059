/*
 * istnet_pw.h -- C ABI of the fused per-point MLP ("SharedMLP") kernels in libistnet_pn2.so.
 *
 * These entry points have NO counterpart in the reference's native extension: there the dense
 * stack Conv2d(1x1, bias=False) -> BatchNorm2d -> ReLU (model/pointnet2/pytorch_utils.py:25-50,
 * 80-134) and the max over nsample (pointnet2_modules.py:65-68) run as separate cuDNN / ATen
 * kernels.  Here one layer is one fp32-MFMA GEMM over the raw (pre-BN) activations; see
 * csrc/pw_mlp.hip and DESIGN.md section 5.
 *
 * Conventions: as istnet_pn2.h (device pointers, int status, async on `stream`, stateless).
 * Layout: activations (B, C, P) f32 with P (= npoint * nsample) contiguous and P % 4 == 0;
 * conv weight w (Cout, Cin) row-major (the Conv2d weight viewed 2-D).
 * BN constant blocks:  bn   = [4][C]: scale = gamma*invstd, shift = beta - mean*scale, mean, invstd
 *                      bwdc = [3][C]: dY = bwdc[0]*g + bwdc[1] + bwdc[2]*y,  g = dA * [y*scale+shift > 0]
 * Gradient w.r.t. the layer output is given either dense (d_dense (B,C,P)) or, after the fused
 * max-pool, pooled (d_pooled (B,C,P/nsample) + arg (B,C,P/nsample) u8, nsample % 4 == 0).
 * pooled_bstride = elements between consecutive clouds of d_pooled (0 = C*P/nsample): a channel slice of the
 * concatenated MSG output gradient (B, Ctot, npoint) is consumed in place, d_pooled pointing at its first row.
 */
#ifndef ISTNET_PW_H_
#define ISTNET_PW_H_

#include "istnet_pn2.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tile configuration the forward / dgrad launch picks for M output rows: M_T * 1000 + N_T (for reporting) */
ISTNET_PN2_API int istnet_pw_tile_cfg(int b, int m, int p);
ISTNET_PN2_API int istnet_pw_dgrad_tile_cfg(int b, int m, int p);
ISTNET_PN2_API int istnet_pw_wgrad_tile_cfg(int b, int cin, int cout, int p);
/* launch heuristics (experiments): key 0 = point count up to which wgrad uses 64x64 tiles, 1 / 2 = split-K
 * workgroup targets for large / small outputs, 5 = workgroups of the fused small-layer backward, 7 = minimum workgroup
 * count for the tall dgrad tiles, 8 / 9 = workgroups / enable of the fused mid-size-layer backward, 11 / 12 = enable /
 * workgroups of the role-split wgrad kernel (dense input, cin and cout >= 64; istnet_pw_wgrad_tile_cfg then reports
 * 1000000 + M_T * 1000 + N_T).  Returns ISTNET_PN2_EINVAL for an unknown key. */
ISTNET_PN2_API int istnet_pw_set_tuning(int key, int value);
/* current value of a tuning key (the keys that are plain numbers: 5, 7, 8, 12, 14, 16, 18, 20, 21), or -1 */
ISTNET_PN2_API int istnet_pw_get_tuning(int key);

/* number of per-channel partial-statistics slots istnet_pw_forward writes for this shape */
ISTNET_PN2_API int istnet_pw_stat_tiles(int b, int cout, int p);

/* y[b][co][p] = sum_ci w[co][ci] * act(x[b][ci][p]); act = relu(v*in_scale[ci]+in_shift[ci]) or identity
 * when in_scale == NULL.  If part_sum != NULL: part_sum/part_sq [cout][tiles] receive per-tile sum(y), sum(y*y). */
ISTNET_PN2_API int istnet_pw_forward(int b, int cin, int cout, int p, const float *x, const float *w,
                                     const float *in_scale, const float *in_shift, float *y,
                                     float *part_sum, float *part_sq, void *stream);
/* istnet_pw_forward takes the direct-operand kernel (pw_fwd2_kernel: the activation operand goes from global memory to
 * the MFMA without an LDS tile) when cin % 16 == 0, cin <= 1024, p % 128 == 0 and cout >= 16 (istnet_pw_set_tuning
 * key 13 = 0 disables it).  istnet_pw_forward_tiles = number of statistics partials per channel that launch writes
 * (istnet_pw_stat_tiles for the other forward entry points); istnet_pw_forward_cfg = 0 for pw_fwd_kernel, else
 * TMW * 1000 + WM * 100 + WN * 10 + (K chunk == 32) of the pw_fwd2_kernel instance (for reporting). */
ISTNET_PN2_API int istnet_pw_forward_tiles(int b, int cin, int cout, int p);
ISTNET_PN2_API int istnet_pw_forward_cfg(int b, int cin, int cout, int p);
/* Small launches (at most 1024 tiles of 32 x 128, cin % 8 == 0, cin >= 64, p % 128 == 0, cout >= 32; keys 15 / 16 of
 * istnet_pw_set_tuning) of istnet_pw_forward, istnet_pw_forward_ld and istnet_pw_forward_acc run pw_fwd_sk_kernel (no LDS
 * operands, K split over the four waves of a workgroup; istnet_pw_forward_cfg reports 1).  Number of statistics partials
 * per channel of an istnet_pw_forward_ld / istnet_pw_forward_acc launch: */
ISTNET_PN2_API int istnet_pw_forward_ld_tiles(int b, int cin, int cout, int p);
/* Layer 0 of a feature-propagation stack in one launch: y = three_interpolate(zk, idx, weight) + w . x, with
 * zk (b, cout, m) the product over the known points, idx / weight (b, p, 3) as three_nn returns them (reference
 * pointnet2_utils.py:249-273 for the interpolation).  Only for shapes the split-K kernel takes (istnet_pw_forward_cfg
 * == 1, else ISTNET_PN2_EINVAL: interpolate first and call istnet_pw_forward_acc); partials: istnet_pw_forward_ld_tiles. */
ISTNET_PN2_API int istnet_pw_forward_acc_interp(int b, int cin, int cout, int p, const float *x, const float *w, int ldw,
                                                const float *zk, int m, const int *idx, const float *weight, float *y,
                                                float *part_sum, float *part_sq, void *stream);
/* The same for istnet_pw_dgrad with a dense gradient source (pw_dgrad_sk_kernel; cout % 8 == 0, 256 <= cout <= 2048 (key 18),
 * p % 128 == 0, m_rows >= 32, at most 1024 tiles; key 17 disables): istnet_pw_dgrad_sk = 1 when that kernel runs,
 * istnet_pw_dgrad_tiles = statistics partials per input channel the launch writes (dense: 1 = dense gradient source). */
ISTNET_PN2_API int istnet_pw_dgrad_sk(int b, int m_rows, int cout, int p);
/* Row blocks of 32 per workgroup (1 or 2) of a split-K FORWARD launch with `rows` output rows over b * p points: 2 =
 * 64 x 128 tiles, taken when the launch still has >= 256 workgroups (istnet_pw_set_tuning key 24; 0 = never).  Same sums in
 * the same order as the 32 x 128 form: bit-identical results.  (The template argument of pw_fwd_sk_kernel<TM> in a trace;
 * the dgrad kernel stays at 32 x 128: pw_dgrad_sk_kernel<1>.) */
ISTNET_PN2_API int istnet_pw_sk_tm(int b, int rows, int p);
ISTNET_PN2_API int istnet_pw_dgrad_tiles(int b, int m_rows, int cout, int p, int dense);
/* 1 when istnet_pw_dgrad runs the loader / MFMA-wave kernel (pw_bwd_mid_kernel<8, 4, POOLED, false>: cout = 256, all 128
 * input channels (ci_off = 0, cin_total = m_rows), p % 128 == 0, statistics requested; key 19 disables) */
ISTNET_PN2_API int istnet_pw_dgrad_rs(int b, int m_rows, int cout, int p, int dense);

/* istnet_pw_forward with w a column slice of a wider row-major matrix: row stride ldw >= cin */
ISTNET_PN2_API int istnet_pw_forward_ld(int b, int cin, int cout, int p, const float *x, const float *w, int ldw,
                                        const float *in_scale, const float *in_shift, float *y,
                                        float *part_sum, float *part_sq, void *stream);

/* y = c_init + w . x (no input activation): the accumulators start from the (b, cout, p) tensor c_init.  Used by the
 * feature-propagation layer 0, y0 = interpolate(Wa . known) + Wb . skip. */
ISTNET_PN2_API int istnet_pw_forward_acc(int b, int cin, int cout, int p, const float *x, const float *w, int ldw,
                                         const float *c_init, float *y, float *part_sum, float *part_sq,
                                         void *stream);

/* BatchNorm statistics partials of an arbitrary (b, c, p) tensor: part_sum / part_sq [c][istnet_pw_bwd_stat_tiles(b, p)] */
ISTNET_PN2_API int istnet_pw_channel_stats(int b, int c, int p, const float *y, float *part_sum, float *part_sq,
                                           void *stream);
/* three_interpolate (reference interpolate_gpu.cu:77-106, pointnet2_utils.py:249-273) of points (b, c, m) to n points
 * together with the BatchNorm statistics partials of the result, one launch: out (b, c, n), bit-identical to
 * istnet_pn2_three_interpolate; part_sum / part_sq [c][istnet_pw_interp_stats_tiles(b, n)].  Layer 0 of a feature-
 * propagation level without skip features (reference pointnet2_modules.py:196-209 with unknow_feats = None). */
ISTNET_PN2_API int istnet_pw_interp_stats_tiles(int b, int n);
ISTNET_PN2_API int istnet_pw_interp_stats(int b, int c, int m, int n, const float *points, const int *idx,
                                          const float *weight, float *out, float *part_sum, float *part_sq, void *stream);

/* out = dY = bwdc[0]*(d_dense*[relu(bn(y)) > 0]) + bwdc[1] + bwdc[2]*y, materialised (b, c, p) */
ISTNET_PN2_API int istnet_pw_dy(int b, int c, int p, const float *y, const float *d_dense, const float *bn,
                                const float *bwdc, float *out, void *stream);

/* Layer 0 of a set-abstraction scale split by linearity: with z = W0[:, 3:] . feat (b, cout, n) computed over the
 * n SOURCE points (istnet_pw_forward_ld), y[b][co][p] = z[b][co][idx[p]] + W0[co][0:3] . (xyz[idx[p]] - new_xyz[p / nsample]),
 * plus the per-channel partials of sum(y), sum(y*y) ([cout][istnet_pw_gather_add_tiles(b, p)], may be NULL).
 * w0 is (cout, ldw) row-major, its first three columns are the xyz weights.  z == NULL: xyz-only layer (no features). */
ISTNET_PN2_API int istnet_pw_gather_add_tiles(int b, int p);
ISTNET_PN2_API int istnet_pw_gather_add(int b, int n, int npoint, int nsample, int cout, const float *xyz,
                                        const float *new_xyz, const int *idx, const float *z, const float *w0,
                                        int ldw, float *y, float *part_sum, float *part_sq, void *stream);

/* (Round 5: the un-split gathered layer-0 forward, istnet_pw_forward_gather, is gone.  Layer 0 of a set-abstraction scale
 * always runs in the split form -- istnet_pw_forward_ld over the n source points + istnet_pw_gather_add -- which does
 * nsample * npoint / n times fewer MACs; the gathered operand loader survives in the weight-gradient kernel below.) */

/* partials -> bn[4][c]; updates running_mean / running_var (unbiased) unless they are NULL.  `momentum` is a DEVICE
 * pointer to one float, read when the kernel runs: a step captured in a HIP graph then follows the per-iteration
 * BNMomentumScheduler.step of the reference (utils/solver.py:91-92, pytorch_utils.py:303-330) instead of replaying the
 * value it was captured with. */
ISTNET_PN2_API int istnet_bn_finalize_fwd(int c, int nt, double count, const float *part_sum,
                                          const float *part_sq, const float *gamma, const float *beta,
                                          float eps, const float *momentum, float *running_mean,
                                          float *running_var, float *bn, void *stream);
/* the same launch also counts the batch: *num_batches_tracked += 1 (the module buffer torch.nn.BatchNorm keeps, an int64 on the
 * device; NULL: not counted) -- the framework's add for it then needs no launch of its own on the forward chain */
ISTNET_PN2_API int istnet_bn_finalize_fwd_nbt(int c, int nt, double count, const float *part_sum,
                                              const float *part_sq, const float *gamma, const float *beta,
                                              float eps, const float *momentum, float *running_mean,
                                              float *running_var, float *bn, long long *num_batches_tracked,
                                              void *stream);

/* out[b][c][g] = max_s relu(y[b][c][g][s]*scale+shift), arg = index of the first maximum (s > 1); cloud b of
 * `out` starts at out + b*out_bstride (0 = c*g), so a scale writes straight into its channel slice of the
 * concatenated MSG output;
 * s == 1: out = relu(y*scale+shift), arg unused (may be NULL);
 * ymax (optional, (b, c, g)): raw y at the arg-max, consumed by istnet_pw_bwd_stats_pooled */
ISTNET_PN2_API int istnet_bn_relu_pool(int b, int c, int g, int s, const float *y, const float *bn,
                                       float *out, long long out_bstride, unsigned char *arg, float *ymax,
                                       void *stream);

/* bn[4][c] = scale, shift, mean, invstd of a FIXED affine normalisation: eval-mode BatchNorm
 * (scale = gamma / sqrt(var + eps), shift = beta - mean * scale) or, with gamma = mean = var = NULL, a plain conv
 * bias (scale 1, shift beta).  The forward / backward kernels then treat the layer like any other. */
ISTNET_PN2_API int istnet_affine_consts(int c, const float *gamma, const float *beta, const float *mean,
                                        const float *var, float eps, float *bn, void *stream);

/* the same for n <= 8 layers in ONE launch (host arrays of length n; gamma[l] / mean[l] / var[l] may be NULL as
 * above): the constants depend on parameters only, so a stack computes all of them before its first GEMM */
ISTNET_PN2_API int istnet_affine_consts_multi(int n, const int *c, const float *const *gamma,
                                              const float *const *beta, const float *const *mean,
                                              const float *const *var, const float *eps, float *const *bn,
                                              void *stream);

/* out = y * bn[0] + bn[1] (per channel), followed by ReLU when relu != 0 -- final layer of a bias stack */
ISTNET_PN2_API int istnet_affine_apply(int b, int c, int p, int relu, const float *y, const float *bn,
                                       float *out, void *stream);

/* backward statistics: partial sums of g and g*y per channel -> [c][tiles] */
ISTNET_PN2_API int istnet_pw_bwd_stat_tiles(int b, int p);
ISTNET_PN2_API int istnet_pw_bwd_stats(int b, int c, int p, int nsample, const float *y,
                                       const float *d_dense, const float *d_pooled, long long pooled_bstride,
                                       const unsigned char *arg, const float *bn, float *part_g,
                                       float *part_gy, void *stream);
/* the same partial sums for a max-pooled gradient source, from the (b, c, g) tensors only: d_pooled (cloud
 * stride pooled_bstride, 0 = c*g) and the raw maxima `ymax` of istnet_bn_relu_pool; partials [c][b] (nt = b) */
ISTNET_PN2_API int istnet_pw_bwd_stats_pooled(int b, int c, int g, const float *d_pooled,
                                              long long pooled_bstride, const float *ymax, const float *bn,
                                              float *part_g, float *part_gy, void *stream);
/* istnet_pw_bwd_stats_pooled followed by istnet_bn_finalize_bwd in one launch (count = b * g * nsample points per
 * channel): dgamma, dbeta and the three BN-backward constants bwdc [3][c] of the last layer of a set-abstraction scale */
/* istnet_pw_bwd_stats (dense gradient source) followed by istnet_bn_finalize_bwd in one launch, count = b * p: the first
 * layer met by the backward pass of a stack without a max-pool (feature propagation) */
ISTNET_PN2_API int istnet_bn_bwd_dense_finalize(int b, int c, int p, double count, int training, const float *y,
                                                const float *d_dense, const float *gamma, const float *bn,
                                                float *dgamma, float *dbeta, float *bwdc, void *stream);
ISTNET_PN2_API int istnet_bn_bwd_pooled_finalize(int b, int c, int g, double count, int training, const float *d_pooled,
                                                 long long pooled_bstride, const float *ymax, const float *gamma,
                                                 const float *bn, float *dgamma, float *dbeta, float *bwdc,
                                                 void *stream);
/* partials -> dgamma, dbeta, bwdc[3][c]; training = 0 treats BN as a fixed affine map (eval mode) */
ISTNET_PN2_API int istnet_bn_finalize_bwd(int c, int nt, double count, int training, const float *part_g,
                                          const float *part_gy, const float *gamma, const float *bn,
                                          float *dgamma, float *dbeta, float *bwdc, void *stream);

/* dx[b][m][p] = sum_co w[co][ci_off+m] * dY[b][co][p], m < m_rows  (w is (cout, cin_total)).
 * Optional fused statistics for the layer that PRODUCED this layer's input (dx is its dA): with
 * part_g != NULL the epilogue also writes per-tile sums of g and g*y_in (g = dx * [y_in*scale+shift > 0],
 * y_in (b, m_rows, p) raw activation, bn_in its bn block) into part_g / part_gy [m_rows][tiles],
 * tiles = istnet_pw_dgrad_stat_tiles(b, m_rows, p) -- a drop-in for istnet_pw_bwd_stats on that layer. */
ISTNET_PN2_API int istnet_pw_dgrad_stat_tiles(int b, int m_rows, int p);
ISTNET_PN2_API int istnet_pw_dgrad(int b, int cin_total, int ci_off, int m_rows, int cout, int p,
                                   int nsample, const float *w, const float *y, const float *d_dense,
                                   const float *d_pooled, long long pooled_bstride, const unsigned char *arg,
                                   const float *bn, const float *bwdc, float *dx, const float *y_in,
                                   const float *bn_in, float *part_g, float *part_gy, void *stream);

/* out[b][co][i] = sum_{p : idx[b][p] == i} dY[b][co][p], i < n  (the group_points_grad scatter applied to the
 * layer's dY instead of to the layer-0 input gradient; the scatter commutes with the channel mixing, so the
 * feature gradient of a set-abstraction scale is  W0[:, 3:]^T . out[b]  -- a GEMM over n instead of p columns).
 * idx (b, p) i32 with values in [0, n); needs n <= 4096.  Cloud b of `out` starts at out + b*out_bstride
 * (0 = cout*n).
 * The same `out` also gives the layer-0 weight gradient without a pass over the grouped points:
 *   dW0[:, 3:] = sum_b out[b] . feat[b]^T  (a wgrad over the n source points), and, when dwx != NULL, this call
 *   also writes dwx[b][co][k] = sum_p dY[b][co][p] * (xyz[b][idx[p]][k] - new_xyz[b][p / group_nsample][k]),
 *   whose sum over b is dW0[:, 0:3].
 * out == NULL (xyz-only layer 0, no feature gradient wanted): only dwx is produced, no scatter; the points of a
 * cloud are then split over istnet_pw_dwx_chunks(b, cout, p) workgroups and dwx is (b * chunks, cout, 3). */
ISTNET_PN2_API int istnet_pw_dwx_chunks(int b, int cout, int p);
ISTNET_PN2_API int istnet_pw_scatter_dy(int b, int cout, int n, int p, int nsample, const float *y,
                                        const float *d_dense, const float *d_pooled, long long pooled_bstride,
                                        const unsigned char *arg, const float *bn, const float *bwdc,
                                        const int *idx, float *out, long long out_bstride, const float *xyz,
                                        const float *new_xyz, int group_nsample, float *dwx, void *stream);
/* The same scatter, atomic-free and deterministic, over inverse lists (offsets (b,n+1), entries (b,p)) of the
 * ball-query indices built by istnet_pn2_csr_build: source point i sums dY0 over the slots that picked it in
 * ascending slot order.  Dense gradient source only.  dwx (b*chunks, cout, 3) with chunks =
 * istnet_pw_scatter_csr_chunks(n) receives the per-workgroup partials of dW0[:, 0:3] (NULL to skip). */
ISTNET_PN2_API int istnet_pw_scatter_csr_chunks(int n);
ISTNET_PN2_API int istnet_pw_scatter_dy_csr(int b, int cout, int n, int p, const float *y, const float *d_dense,
                                            const float *bn, const float *bwdc, const int *offsets,
                                            const int *entries, float *out, long long out_bstride,
                                            const float *xyz, const float *new_xyz, int group_nsample,
                                            float *dwx, void *stream);
/* istnet_bn_finalize_bwd of the layer folded into istnet_pw_scatter_dy_csr: every workgroup derives the BN-backward
 * constants of its few channels from the nt statistics partials (part_g / part_gy [cout][nt]); the workgroups of cloud 0
 * store dgamma, dbeta and bwdc [3][cout] (bwdc is an OUTPUT here). */
ISTNET_PN2_API int istnet_pw_scatter_dy_csr_fin(int b, int cout, int n, int p, const float *y, const float *d_dense,
                                                const float *bn, int nt, double count, int training,
                                                const float *part_g, const float *part_gy, const float *gamma,
                                                float *dgamma, float *dbeta, float *bwdc, const int *offsets,
                                                const int *entries, float *out, long long out_bstride,
                                                const float *xyz, const float *new_xyz, int group_nsample, float *dwx,
                                                void *stream);

/* split-K weight gradient (requires p % 32 == 0): dw_part[split][co][ci] = sum_{p in split} dY[b][co][p] * act(x[b][ci][p]) with
 * istnet_pw_wgrad_splits(...) splits; istnet_pw_wgrad_reduce sums the partials in a fixed order:
 * dw[i] = sum_s dw_part[s][i], count = cout*cin */
ISTNET_PN2_API int istnet_pw_wgrad_splits(int b, int cin, int cout, int p);
ISTNET_PN2_API int istnet_pw_wgrad(int b, int cin, int cout, int p, int nsample, const float *x,
                                   const float *in_scale, const float *in_shift, const float *y,
                                   const float *d_dense, const float *d_pooled, long long pooled_bstride,
                                   const unsigned char *arg, const float *bn, const float *bwdc, float *dw_part,
                                   void *stream);
/* Fused backward of a small layer (cin <= 32, cout <= 32, input = raw output x of the previous layer with its
 * BN block bn_in): one pass writes dx (b, cin, p) = W^T . dY, the statistics partials part_g / part_gy
 * [cin][splits] of g = dx * [relu(bn_in(x)) > 0] (for istnet_bn_finalize_bwd of the previous layer,
 * nt = splits = istnet_pw_bwd_small_splits(b, p)) and the split-K partials dw_part [splits][cout][cin] of dW --
 * i.e. istnet_pw_dgrad + istnet_pw_wgrad in one read of (y, gradient source, x). */
ISTNET_PN2_API int istnet_pw_bwd_small_ok(int cin, int cout, int p);
ISTNET_PN2_API int istnet_pw_bwd_small_splits(int b, int p);
ISTNET_PN2_API int istnet_pw_bwd_small(int b, int cin, int cout, int p, int nsample, const float *w, const float *x,
                                       const float *bn_in, const float *y, const float *d_dense,
                                       const float *d_pooled, long long pooled_bstride, const unsigned char *arg,
                                       const float *bn, const float *bwdc, float *dx, float *part_g,
                                       float *part_gy, float *dw_part, void *stream);

/* wgrad with the layer-0 input of a set-abstraction scale gathered on the fly (the grouped tensor of QueryAndGroup,
 * pointnet2_utils.py:348-358, is never materialised): input channel k < 3 is xyz[b][idx[b][p]][k] - new_xyz[b][p / nsample][k],
 * channel k >= 3 is feat[b][k-3][idx[b][p]]; cin = 3 + cfeat, p = npoint * nsample, nsample % 4 == 0, feat may be NULL when
 * cfeat == 0.  grad_nsample = nsample of the pooled gradient source (0 when d_dense is given) */
ISTNET_PN2_API int istnet_pw_wgrad_gather(int b, int n, int npoint, int nsample, int cfeat, int cout,
                                          int grad_nsample, const float *xyz, const float *new_xyz,
                                          const float *feat, const int *idx, const float *y,
                                          const float *d_dense, const float *d_pooled, long long pooled_bstride,
                                          const unsigned char *arg, const float *bn, const float *bwdc,
                                          float *dw_part, void *stream);
ISTNET_PN2_API int istnet_pw_wgrad_reduce(int count, int splits, const float *dw_part, float *dw,
                                          void *stream);
/* the same reduction for n <= 8 layers in ONE launch (host arrays of length n) */
ISTNET_PN2_API int istnet_pw_wgrad_reduce_multi(int n, const int *counts, const int *splits,
                                                const float *const *parts, float *const *dws, void *stream);

/* istnet_bn_finalize_fwd of a stack's LAST layer folded into its tail (istnet_bn_relu_pool): one launch that finishes the
 * batch statistics from the partials (writes bn[4][c], updates the running statistics; `momentum` is a device pointer),
 * then applies out = max_s relu(scale y + shift) (+ arg / ymax) exactly as istnet_bn_relu_pool does -- bit-identical to
 * the two launches.  s == 1: out = relu(scale y + shift) (arg / ymax unused).  [ref pytorch_utils.py:25-50 +
 * pointnet2_modules.py:65-68] */
ISTNET_PN2_API int istnet_bn_fin_relu_pool(int b, int c, int g, int s, int nt, double count, const float *part_sum,
                                           const float *part_sq, const float *gamma, const float *beta, float eps,
                                           const float *momentum, float *running_mean, float *running_var, float *bn,
                                           const float *y, float *out, long long out_bstride, unsigned char *arg,
                                           float *ymax, void *stream);
ISTNET_PN2_API int istnet_bn_fin_relu_pool_nbt(int b, int c, int g, int s, int nt, double count,
                                               const float *part_sum, const float *part_sq, const float *gamma,
                                               const float *beta, float eps, const float *momentum,
                                               float *running_mean, float *running_var, float *bn, const float *y,
                                               float *out, long long out_bstride, unsigned char *arg, float *ymax,
                                               long long *num_batches_tracked, void *stream);   /* counts the batch too */

/* three_interpolate_grad over the inverse lists of the taps (istnet_pn2_three_interpolate_grad_csr; reference
 * interpolate_gpu.cu:115-148) of dY = ca * relu'(bn(y)) * d_dense + cb + cc * y, formed on the fly from the raw pair
 * (y, d_dense) (b, c, n) and the constant blocks bn[4][c] / bwdc[3][c]: grad_points (b, c, m).  Bit-identical to
 * istnet_pw_dy followed by istnet_pn2_three_interpolate_grad_csr. */
ISTNET_PN2_API int istnet_interp_grad_csr_dy(int b, int c, int n, int m, const float *y, const float *d_dense,
                                             const float *bn, const float *bwdc, const float *weight,
                                             const int *offsets, const int *entries, float *grad_points,
                                             void *stream);

/* out (b, c) = mean over the p points of relu(scale_c * y[b][c][:] + shift_c)  (bn = [scale | shift | ...] as everywhere):
 * the AdaptiveAvgPool1d(1) that ends pose_mlp2 of both estimators (model/ist_net.py:246,314) from the raw output of the
 * stack's last layer; istnet_expand_rows is its adjoint, out[row][:] = g[row] / p for rows = b * c.  p % 4 == 0. */
ISTNET_PN2_API int istnet_bn_relu_mean(int b, int c, int p, const float *y, const float *bn, float *out, void *stream);
ISTNET_PN2_API int istnet_expand_rows(int rows, int p, const float *g, float *out, void *stream);


/* ... where item l may be a column block of a wider destination and a row block of wider partials: element i of the
 * item (i < counts[l], counts[l] % cols[l] == 0) is the sum over k < splits[l] of parts[l][k * pstrides[l] + i] and goes
 * to dws[l][(i / cols[l]) * lds[l] + i % cols[l]].  The pieces of one weight gradient produced by different launches
 * (xyz / feature columns of a set-abstraction layer 0, interpolated / skip columns of a feature-propagation layer 0)
 * are written straight into the parameter's gradient -- what the reference gets from autograd's cat backward. */
ISTNET_PN2_API int istnet_pw_wgrad_reduce_multi_ld(int n, const int *counts, const int *splits,
                                                   const float *const *parts, float *const *dws, const int *cols,
                                                   const int *lds, const long long *pstrides, void *stream);
/* n <= 64 device buffers of words[l] 4-byte words each copied back to back into dst, one launch (stacking the layer-0
 * weights of a level's scales; refilling the flat storage of a geometry slot). */
ISTNET_PN2_API int istnet_pack_words(int n, const void *const *srcs, const long long *words, void *dst,
                                     void *stream);

/* Layer 0 of a stack whose input is the channel concatenation of nsrc <= 6 tensors srcs[s] (b, chans[s], p), without
 * building the concatenation (the IST / pose heads concatenate 3-5 feature tensors in front of every per-point MLP,
 * model/ist_net.py:167-171,253,322): y (b, cout, p) = w[:, 0:sum chans] . [srcs...] (+ row_init[b][co], an optional
 * per-cloud bias (b, cout) -- the rank-1 term W[:, C:] . mean(feat) that replaces the reference's expand + concat of
 * the global mean feature, :174-175,256-257,324-325).  chans[s] % 16 == 0; w is (cout, ldw) row-major. */
ISTNET_PN2_API int istnet_pw_forward_multi(int b, int nsrc, const float *const *srcs, const int *chans, int cout,
                                           int p, const float *w, int ldw, const float *row_init, float *y,
                                           void *stream);

/* ---- compact-column form of a set-abstraction scale (csrc/sa_compact.hip) ------------------------------------------
 * The reference pads every ball-query row to nsample slots by repeating its first hit (ball_query_gpu.cu:38-45) and
 * pushes the repeats through SharedMLP + max_pool2d like any other slot (pointnet2_modules.py:61-68).  A repeat has the
 * activations of slot 0 of its group; what it adds is its MULTIPLICITY in every sum over points (BatchNorm statistics
 * forward and backward, weight gradients, gradient scatter).  These entry points evaluate a scale on compact columns:
 * per group its distinct neighbours, then one representative of the repeats carrying their number as a column weight.
 * All groups of all clouds share one point axis of static capacity cap = b*g*s (a multiple of 256); the number of valid
 * columns lives in device memory (gstart[b*g]), so grids are static and workgroups past it return at once.
 *
 * istnet_sa_compact: idx (b, g, s) ball-query indices over n source points per cloud ->
 *   glen (b*g) columns per group, gstart (b*g + 1) their exclusive scan (gstart[b*g] = T, the valid column count),
 *   cidx (cap) global source point b*n + i of a column, meta (cap) = group * 64 + position in the group,
 *   colw (cap) multiplicity (1, or nsample - cnt for the representative; 0 on the null columns T .. roundup(T, 256)). */
ISTNET_PN2_API int istnet_sa_compact(int b, int g, int s, int n, const int *idx, int *glen, int *gstart, int *cidx,
                                     int *meta, float *colw, void *stream);
/* the same for the two scales of one level in 2 launches (3 without counts) instead of 6; have_glen: glen_a / glen_b were
 * written by istnet_pn2_query_ball_point_pair */
ISTNET_PN2_API int istnet_sa_compact_pair(int b, int g, int n, int s_a, const int *idx_a, int *glen_a, int *gstart_a,
                                          int *cidx_a, int *meta_a, float *colw_a, int s_b, const int *idx_b, int *glen_b,
                                          int *gstart_b, int *cidx_b, int *meta_b, float *colw_b, int have_glen,
                                          void *stream);
/* layer 0 on compact columns: y (cout, cap) = z[cloud][:, point] + W0[:, 0:3] . (xyz[source] - new_xyz[group]), weighted
 * partials [cout][cap / 256]; z (b, cout, n) or NULL (xyz-only level) */
ISTNET_PN2_API int istnet_pw_gather_add_cols(int b, int n, int g, long long cap, int cout, const float *xyz,
                                             const float *new_xyz, const int *cidx, const int *meta, const float *colw,
                                             const int *ncols, const float *z, const float *w0, int ldw, float *y,
                                             float *part_sum, float *part_sq, void *stream);
/* istnet_pw_forward on compact columns: x (cin, cap) -> y (cout, cap), weighted partials [cout][istnet_pw_stat_tiles(1, cout, cap)] */
ISTNET_PN2_API int istnet_pw_forward_cols(int cin, int cout, long long cap, const float *x, const float *w,
                                          const float *in_scale, const float *in_shift, float *y, float *part_sum,
                                          float *part_sq, const int *ncols, const float *colw, void *stream);
/* istnet_bn_relu_pool over the ragged groups: out (b, c, g) slice, arg = position of the first maximum inside the group */
ISTNET_PN2_API int istnet_bn_relu_pool_cols(int b, int c, int g, long long cap, const float *y, const float *bn,
                                            const int *gstart, float *out, long long out_bstride, unsigned char *arg,
                                            float *ymax, void *stream);
/* gradient through the max-pool as a dense compact tensor (c, cap): d_pooled at each group's arg-max column, else 0 */
ISTNET_PN2_API int istnet_pw_pooled_grad_cols(int b, int c, int g, long long cap, const float *d_pooled,
                                              long long pooled_bstride, const unsigned char *arg, const int *meta,
                                              const int *ncols, float *out, void *stream);
/* The same contract for a mid-size layer (cout in {64, 128}, cin in {32, 64, 128} with cin <= cout, p % 128 == 0; replaces
 * the istnet_pw_dgrad + istnet_pw_wgrad pair of such a layer -- reference: the autograd backward of one
 * Conv2d(1x1) + BatchNorm2d + ReLU block of pt_utils.SharedMLP, /root/reference/lib/pointnet2/pytorch_utils.py:10-60):
 * dx (b, cin, p), statistics partials [cin][splits], dw_part [splits][cout][cin], splits = istnet_pw_bwd_mid_splits(...).
 * istnet_pw_set_tuning key 8 = target workgroup count, key 9 = 0 disables (istnet_pw_bwd_mid_ok then returns 0). */
ISTNET_PN2_API int istnet_pw_bwd_mid_ok(int cin, int cout, int p);
ISTNET_PN2_API int istnet_pw_bwd_mid_splits(int b, int cin, int cout, int p);
ISTNET_PN2_API int istnet_pw_bwd_mid(int b, int cin, int cout, int p, int nsample, const float *w, const float *x,
                                     const float *bn_in, const float *y, const float *d_dense, const float *d_pooled,
                                     long long pooled_bstride, const unsigned char *arg, const float *bn,
                                     const float *bwdc, float *dx, float *part_g, float *part_gy, float *dw_part,
                                     void *stream);

/* istnet_pw_bwd_small on compact columns (dense gradient source); partials: [cin][splits], dw_part [splits][cout][cin],
 * splits = istnet_pw_bwd_small_cols_splits() */
ISTNET_PN2_API int istnet_pw_bwd_small_cols_splits(void);
ISTNET_PN2_API int istnet_pw_bwd_small_cols(int cin, int cout, long long cap, const float *w, const float *x,
                                            const float *bn_in, const float *y, const float *d_dense, const float *bn,
                                            const float *bwdc, float *dx, float *part_g, float *part_gy,
                                            float *dw_part, const int *ncols, const float *colw, void *stream);
/* xyz weight gradient of layer 0 on compact columns: dwx (chunks, cout, 3) partials, chunks = istnet_pw_dwx_cols_chunks(cout) */
ISTNET_PN2_API int istnet_pw_dwx_cols_chunks(int cout);
ISTNET_PN2_API int istnet_pw_dwx_cols(int cout, long long cap, const float *y, const float *d_dense, const float *bn,
                                      const float *bwdc, const int *cidx, const int *meta, const float *colw,
                                      const int *ncols, const float *xyz, const float *new_xyz, float *dwx,
                                      void *stream);


/* dense-gradient dgrad / wgrad on compact columns (layers too wide for the fused small-layer backward): same
 * arithmetic as istnet_pw_dgrad / istnet_pw_wgrad with b = 1, p = cap; statistics partials [m_rows][istnet_pw_dgrad_stat_tiles(1, m_rows, cap)]
 * and weight-gradient partials [istnet_pw_wgrad_cols_splits(cin, cout)][cout][cin] carry the column multiplicities */
ISTNET_PN2_API int istnet_pw_dgrad_cols(int cin_total, int ci_off, int m_rows, int cout, long long cap, const float *w,
                                        const float *y, const float *d_dense, const float *bn, const float *bwdc,
                                        float *dx, const float *y_in, const float *bn_in, float *part_g,
                                        float *part_gy, const int *ncols, const float *colw, void *stream);
ISTNET_PN2_API int istnet_pw_wgrad_cols_splits(int cin, int cout);
ISTNET_PN2_API int istnet_pw_wgrad_cols(int cin, int cout, long long cap, const float *x, const float *in_scale,
                                        const float *in_shift, const float *y, const float *d_dense, const float *bn,
                                        const float *bwdc, float *dw_part, const int *ncols, const float *colw,
                                        void *stream);
/* istnet_pw_scatter_dy_csr on compact columns: inverse lists from istnet_pn2_csr_build_segmented over cidx (seg = gstart
 * taken every g groups, key_sub = n); out (b, rows, n) / dwx (b, cout, 3) as in the padded form */
ISTNET_PN2_API int istnet_pw_scatter_dy_csr_cols(int b, int cout, int n, int g, long long cap, const float *y,
                                                 const float *d_dense, const float *bn, const float *bwdc,
                                                 const int *gstart, const int *offsets, const int *entries,
                                                 const int *meta, const float *colw, float *out,
                                                 long long out_bstride, const float *xyz, const float *new_xyz,
                                                 float *dwx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ISTNET_PW_H_ */
