/*
 * istnet_rgb.h -- C ABI (library libistnet_pn2.so, conventions of istnet_pn2.h) of the RGB-branch decoder pieces that
 * are not convolutions (the PSPNet decoder of the reference, model/modules.py:36-81).  The convolutions
 * and batch-norms of the branch run on MIOpen; profiles/r02_rgb_branch_breakdown.txt showed 24 % of the branch's
 * training step in the framework's PReLU backward and 7 % in its bilinear-upsample backward.
 */
#ifndef ISTNET_RGB_H_
#define ISTNET_RGB_H_

#include "istnet_pn2.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Backward of nn.PReLU() with ONE slope parameter (modules.py:42, :64-68) over n contiguous f32 elements (any memory
 * format: x and dy only have to share it):  dx = dy * (x > 0 ? 1 : a);  part[j] = partial sums of dy * x * [x <= 0]
 * (the caller sums the `nparts` = istnet_prelu_bwd_parts(n) partials in a fixed order -> d(slope)).  a: device scalar. */
ISTNET_PN2_API int istnet_prelu_bwd_parts(long long n);
ISTNET_PN2_API int istnet_prelu_bwd(long long n, const float *x, const float *dy, const float *a, float *dx,
                                    float *part, void *stream);

/* Backward of F.interpolate(mode="bilinear", align_corners=True) (nn.Upsample(scale_factor=2) of PSPUpsample,
 * modules.py:39) for channels-last tensors: dx (b, hin, win, c) from dy (b, hout, wout, c), c % 4 == 0.  Gather form
 * (every input pixel sums its <= 5 x 5 weighted output pixels; the weights follow the framework's forward
 * arithmetic: src = o * (in-1)/(out-1), i0 = floor(src), w1 = src - i0) -- no atomics, deterministic. */
ISTNET_PN2_API int istnet_upsample_bilinear_ac_bwd_nhwc(int b, int c, int hin, int win, int hout, int wout,
                                                        const float *dy, float *dx, void *stream);

/* Forward of the same op: y (b, hout, wout, c) from x (b, hin, win, c), channels-last, c % 4 == 0; the blend in the
 * framework's order h0*(w0*a + w1*b) + h1*(w0*c + w1*d).  One thread per output pixel and channel quad; the
 * framework's kernel writes the decoder's 192x192 map at ~0.8 TB/s, this one is bound by the store stream. */
ISTNET_PN2_API int istnet_upsample_bilinear_ac_fwd_nhwc(int b, int c, int hin, int win, int hout, int wout,
                                                        const float *x, float *y, void *stream);

/* PSPUpsample (modules.py:36-49) = bilinear 2x upsample (align_corners) -> 3x3 convolution (padding 1) without the
 * convolution at full size: conv3x3(U p) = sum_taps shift_tap(U (W_tap p)).  The caller forms q = p . Wr on the small
 * map (one GEMM, (b*hin*win, cin) x (cin, 9*c), Wr[ci][(ky*3+kx)*c + co] = W[co][ci][ky][kx]: a quarter of the
 * convolution's flops, forward and both backward products); these two kernels do what is left at full size:
 *   fwd: y[b,oy,ox,co] = bias[co] + sum_{ky,kx} inside(oy+ky-1, ox+kx-1) * bilinear(q[b,:,:,ky*3+kx,co]; oy+ky-1, ox+kx-1)
 *   bwd: dq = transpose of that map applied to dy (gather form, no atomics).
 * q / dq: (b, hin, win, 9, c) f32; y / dy: (b, hout, wout, c) channels-last; c % 4 == 0; bias may be NULL. */
ISTNET_PN2_API int istnet_upconv3_fwd_nhwc(int b, int c, int hin, int win, int hout, int wout, const float *q,
                                           const float *bias, float *y, void *stream);
ISTNET_PN2_API int istnet_upconv3_bwd_nhwc(int b, int c, int hin, int win, int hout, int wout, const float *dy,
                                           float *dq, void *stream);

/* BatchNorm2d with batch statistics + PReLU (one slope) + optional Dropout2d mask of a channels-last map y (b, hw, c), the
 * tail of every decoder stage (model/modules.py:25-34,63-65), as two streaming passes per direction:
 *   forward   istnet_nhwc_channel_stats: part_sum / part_sq [c][istnet_nhwc_stat_parts(b * hw)] (finalised by
 *             istnet_bn_finalize_fwd of istnet_pw.h, which also updates the running statistics and writes
 *             bn = [scale | shift | mean | invstd]);  istnet_nhwc_bn_prelu_apply: z = prelu(scale y + shift) * mask[b][c]
 *   backward  istnet_nhwc_bn_prelu_bwd_stats: with g = dz * mask * prelu'(scale y + shift): part_g = sum g,
 *             part_gy = sum g y per channel (finalised by istnet_bn_finalize_bwd -> dgamma, dbeta, bwdc = [ca | cb | cc]),
 *             part_slope[parts] = partial sums of the slope gradient;  istnet_nhwc_bn_prelu_bwd_apply: dy = ca g + cb + cc y.
 * c % 4 == 0, c <= 1024; mask (b, c) or NULL. */
ISTNET_PN2_API int istnet_nhwc_stat_parts(long long rows);
ISTNET_PN2_API int istnet_nhwc_channel_stats(long long rows, int c, const float *y, float *part_sum, float *part_sq,
                                             void *stream);
ISTNET_PN2_API int istnet_nhwc_bn_prelu_apply(int b, long long hw, int c, const float *y, const float *bn,
                                              const float *slope, const float *mask, float *z, void *stream);
ISTNET_PN2_API int istnet_nhwc_bn_prelu_bwd_stats(int b, long long hw, int c, const float *y, const float *dz,
                                                  const float *bn, const float *slope, const float *mask, float *part_g,
                                                  float *part_gy, float *part_slope, void *stream);
/* finalize of the backward statistics of a decoder stage: istnet_bn_finalize_bwd's algebra (training mode) plus colsum[c] =
 * sum_p dy (the preceding convolution's bias gradient, from the statistics: no pass over the map; NULL: not formed) and
 * *dslope = sum of part_slope (NULL: not formed). */
ISTNET_PN2_API int istnet_nhwc_bn_prelu_bwd_finalize(int c, int nparts, double count, const float *part_g,
                                                     const float *part_gy, const float *part_slope, const float *gamma,
                                                     const float *bn, float *dgamma, float *dbeta, float *bwdc,
                                                     float *colsum, float *dslope, void *stream);
ISTNET_PN2_API int istnet_nhwc_bn_prelu_bwd_apply(int b, long long hw, int c, const float *y, const float *dz,
                                                  const float *bn, const float *bwdc, const float *slope, const float *mask,
                                                  float *dy, void *stream);

/* The same two passes for the tail of a ResNet basic block (reference model/resnet.py:36-67: out = relu(bn2(conv2(.)) +
 * identity), torch's BatchNorm2d + add + ReLU kernels), slope = 0 for ReLU (any slope >= 0 works; the slope gradient is
 * not formed):
 *   forward   istnet_nhwc_bn_act_res_apply: z = act(scale y + shift + res)
 *   backward  istnet_nhwc_bn_act_res_bwd_stats: g = dz * act'(.) with the sign taken from the saved output z; writes g (the
 *             identity branch's gradient) and the per-channel partials of sum g, sum g y; after istnet_bn_finalize_bwd the
 *             input gradient is istnet_nhwc_bn_prelu_bwd_apply(y, g, ..., slope = 1, mask = NULL).
 * A plain BatchNorm2d + ReLU (bn1 of a block, the stem) is istnet_nhwc_bn_prelu_* with slope = 0. */
ISTNET_PN2_API int istnet_nhwc_bn_act_res_apply(int b, long long hw, int c, const float *y, const float *bn,
                                                const float *slope, const float *res, float *z, void *stream);
ISTNET_PN2_API int istnet_nhwc_bn_act_res_bwd_stats(int b, long long hw, int c, const float *y, const float *dz,
                                                    const float *z, const float *bn, const float *slope, float *g,
                                                    float *part_g, float *part_gy, float *part_slope, void *stream);

/* The decoder's `final` stage (Conv2d 1x1 -> BatchNorm2d -> PReLU, modules.py:63-67) at the chosen pixels only
 * (ist_net.py:41-45), training mode: its batch statistics follow from the moments of the stage's input u, a
 * channels-last map of `rows` = b*h*w pixels x 64 channels.
 *   istnet_nhwc_gram64: s1[64] = sum_p u_p, s2[64][64] = sum_p u_p u_p^T as FLOAT64 (per-workgroup fp32 MFMA partials
 *     part_s2 [istnet_nhwc_gram64_parts(rows)][64][64], part_s1 [parts][64], summed in a fixed order in float64) -- one
 *     pass over u;
 *   istnet_nhwc_rowmix64: out[p][j] = c0[j] + sum_i u[p][i] * a[j][i], the dense part of the stage's input gradient
 *     (the statistics couple every pixel: dL/du_p = A u_p + c0, A = W^T diag(k) W). */
ISTNET_PN2_API int istnet_nhwc_gram64_parts(long long rows);
ISTNET_PN2_API int istnet_nhwc_gram64(long long rows, const float *u, float *part_s2, float *part_s1, double *s2,
                                      double *s1, void *stream);
ISTNET_PN2_API int istnet_nhwc_rowmix64(long long rows, const float *u, const float *a, const float *c0, float *out,
                                        void *stream);

/* The whole stage in two calls (round 4; rgb_branch._FinalAtChosenFn, reference model/modules.py:63-67 evaluated at the pixels
 * of model/ist_net.py:41-45).  u: (b, hw, 64) channels-last input of the stage, choose: (b, n) int64 flat pixel indices,
 * w: (cout, 64) the 1x1 convolution, cout <= 512, one PReLU slope.  forward: moments (as above) -> batch statistics
 * stat[3][cout] float64 (mean, 1/sqrt(var + eps), var; running statistics updated with *momentum_p when given) -> y and
 * zhat (b, cout, n).  backward: dy (b, cout, n) -> du (b, hw, 64) (every pixel: A u_p + c0; the chosen ones also dz W, added
 * atomically because a pixel may be chosen twice), dw, db, dgamma, dbeta, dslope.  Work buffers: part [3][cout][b] float,
 * bwdc [4][cout] double, amat [64][64], c0 [64], dwp [istnet_final_chosen_workgroups(b*n)][cout][64] float. */
ISTNET_PN2_API int istnet_final_chosen_workgroups(long long rows);
ISTNET_PN2_API int istnet_final_chosen_forward(int b, long long hw, int n, int cout, const float *u, const long long *choose,
                                               const float *w, const float *bias, const float *gamma, const float *beta,
                                               const float *slope, float *running_mean, float *running_var,
                                               const float *momentum_p, double eps, float *part_s2, float *part_s1, double *s2,
                                               double *s1, double *stat, float *y, float *zhat, void *stream);
ISTNET_PN2_API int istnet_final_chosen_backward(int b, long long hw, int n, int cout, const float *u, const long long *choose,
                                                const float *w, const float *bias, const float *gamma, const float *beta,
                                                const float *slope, const double *s2, const double *s1, const double *stat,
                                                const float *dy, const float *zhat, float *part, double *bwdc, float *amat,
                                                float *c0, float *dwp, float *du, float *dw, float *db, float *dgamma,
                                                float *dbeta, float *dslope, void *stream);

#ifdef __cplusplus
}
#endif
#endif
