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

#ifdef __cplusplus
}
#endif
#endif
