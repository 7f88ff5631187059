/*
 * istnet_heads.h -- C ABI (library libistnet_pn2.so, conventions of istnet_pn2.h) of the tail of IST-Net's pose
 * estimators and of the pose / NOCS losses: SURVEY.md section 8 rows a13 (HeavyEstimator / LightEstimator heads,
 * model/ist_net.py:250-264,318-332) and a14 (Ortho6d2Mat, utils/rotation_utils.py:4-28), and the loss terms of
 * model/losses.py:3-49 that the training step evaluates on their outputs.  All tensors contiguous float32 on the device;
 * every function enqueues on `stream` and returns 0 or a hipError_t / ISTNET_PN2_EINVAL.
 */
#ifndef ISTNET_HEADS_H_
#define ISTNET_HEADS_H_

#include "istnet_pn2.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One nn.Linear (+ ReLU when relu != 0) layer of nheads <= 4 heads in ONE launch -- the three pose heads of an estimator
 * (rotation_estimator / translation_estimator / size_estimator, ist_net.py:232-248) run layer by layer side by side:
 *     y[h] (b, n[h]) = act(x[h] (b, k) . w[h]^T + bias[h]),   w[h] (n[h], k) as torch stores Linear.weight.
 * b <= 64 (the batch of instances), k % 4 == 0.  x[h] may be the same pointer for every head (the pooled feature). */
ISTNET_PN2_API int istnet_fc_forward(int nheads, int b, int k, const int *n, const float *const *x, const float *const *w,
                                     const float *const *bias, float *const *y, int relu, void *stream);
/* Backward of the same layer.  dz = dy . [y > 0] when relu != 0 (y = the layer's output), else dy.
 *   dx[h] (b, k) = dz[h] . w[h]                (skipped when dx == NULL; shared_x != 0: x is shared, the heads' input
 *                                               gradients are SUMMED into dx[0])
 *   dw[h] (n[h], k) = dz[h]^T . x[h],  db[h] (n[h]) = column sums of dz[h]      (skipped when dw == NULL) */
ISTNET_PN2_API int istnet_fc_backward(int nheads, int b, int k, const int *n, const float *const *dy, const float *const *y,
                                      const float *const *w, const float *const *x, float *const *dx, float *const *dw,
                                      float *const *db, int relu, int shared_x, void *stream);

/* Ortho6d2Mat (utils/rotation_utils.py:4-28): r6 (b, 6) = [x_raw | y_raw] -> r (b, 3, 3) with columns
 * [y x z, y = norm(y_raw), z = norm(x_raw x y)]; norms clamped at 1e-8.  The backward is the closed form of the two
 * normalisations and two cross products (d_r (b, 3, 3) -> d_r6 (b, 6)). */
ISTNET_PN2_API int istnet_ortho6d_forward(int b, const float *r6, float *r, void *stream);
ISTNET_PN2_API int istnet_ortho6d_backward(int b, const float *r6, const float *d_r, float *d_r6, void *stream);

/* PoseDis (model/losses.py:37-49): loss (1) = mean over (b, 3) of |r1 - r2| taken over dim 1, + mean_b |t1 - t2| +
 * mean_b |s1 - s2|.  norms (b, 5) keeps the five norms per sample for the backward, which writes the gradients with
 * respect to r1 (b, 3, 3), t1 (b, 3), s1 (b, 3) scaled by gout (1). */
ISTNET_PN2_API int istnet_pose_dis_forward(int b, const float *r1, const float *t1, const float *s1, const float *r2,
                                           const float *t2, const float *s2, float *loss, float *norms, void *stream);
ISTNET_PN2_API int istnet_pose_dis_backward(int b, const float *gout, const float *r1, const float *t1, const float *s1,
                                            const float *r2, const float *t2, const float *s2, const float *norms,
                                            float *dr, float *dt, float *ds, void *stream);

/* SmoothL1Dis (model/losses.py:3-22) of p1, p2 (rows, 3): mean over rows of the sum over xyz of
 * (|d| > threshold ? |d| - threshold / 2 : d^2 / (2 threshold)).  part: istnet_smooth_l1_parts(rows) floats of scratch
 * (per-workgroup partial sums, added in a fixed order).  The backward writes d p1 (rows, 3). */
ISTNET_PN2_API int istnet_smooth_l1_parts(long long rows);
ISTNET_PN2_API int istnet_smooth_l1_forward(long long rows, float threshold, const float *p1, const float *p2, float *part,
                                            float *loss, void *stream);
ISTNET_PN2_API int istnet_smooth_l1_backward(long long rows, float threshold, const float *gout, const float *p1,
                                             const float *p2, float *dp1, void *stream);

/* Mean squared error of two dense float32 tensors of n elements -- the feature-alignment term nn.MSELoss of the
 * reference's SupervisedLoss (model/ist_net.py:99) -- together with its gradient, one pass over the operands:
 * loss = mean (a - b)^2 (per-workgroup partial sums in part[istnet_mse_parts(n)], added in a fixed order in float64) and
 * da = 2 (a - b) / n  (the gradient with respect to b is -da).  b may be NULL (b = 0: the mean of squares).  a, b, da
 * 16-byte aligned. */
ISTNET_PN2_API int istnet_mse_parts(long long n);
ISTNET_PN2_API int istnet_mse_value_grad(long long n, const float *a, const float *b, float *da, float *part, float *loss,
                                         void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ISTNET_HEADS_H_ */
