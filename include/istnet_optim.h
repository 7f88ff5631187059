/*
 * istnet_optim.h -- C ABI (library libistnet_pn2.so, conventions of istnet_pn2.h) of the optimizer step over one
 * flat parameter buffer.  Replaces torch.optim.Adam.step of the reference's training loop
 * (utils/solver.py:88-99, optimizer built at train.py:101-107).
 */
#ifndef ISTNET_OPTIM_H_
#define ISTNET_OPTIM_H_

#include "istnet_pn2.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One Adam update (no amsgrad, L2 weight decay added to the gradient, as torch.optim.Adam) of n contiguous f32
 * parameters:   g' = grad*grad_scale + wd*param;  m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2;
 *               param -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),   t = *step (device f32, already
 * incremented by the caller; read-only here, so the launch can sit in a captured graph).  lr_dev, when not NULL,
 * is a device f32 that overrides `lr`: a per-iteration schedule (the reference steps CyclicLR every iteration,
 * utils/solver.py:46-47,88-89) then only writes that scalar and a captured step needs no re-capture.  grad_scale folds the
 * 1/world_size of a data-parallel sum all-reduce into the update.  One element per lane, float4 accesses,
 * n/1024 workgroups -- the point of not using a multi-tensor-apply kernel on a single tensor (20 workgroups for
 * the encoder's 1.31 M parameters). */
ISTNET_PN2_API int istnet_adam_step(long long n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                    const float *step, const float *lr_dev, double lr, double beta1, double beta2,
                                    double eps, double weight_decay, double grad_scale, void *stream);

/* The same update with the step count advanced BY the launch: t = *step + 1 is what the update uses, and the last workgroup
 * to finish stores it back (ticket: a device u32 that is 0 between launches and 0 again afterwards).  Saves the framework's
 * `step += 1` kernel in front of every update (reference utils/solver.py:98, torch.optim.Adam.step keeps its count the
 * same way: state['step'] += 1, then the update with the new value).  n == 0 only counts. */
ISTNET_PN2_API int istnet_adam_step_counting(long long n, float *param, const float *grad, float *exp_avg,
                                             float *exp_avg_sq, float *step, unsigned *ticket, const float *lr_dev,
                                             double lr, double beta1, double beta2, double eps, double weight_decay,
                                             double grad_scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif
