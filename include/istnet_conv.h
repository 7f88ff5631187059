/* C ABI of the RGB trunk's convolutions: channels-last (NHWC) float32 2-D convolution as an
 * implicit GEMM on the fp32 matrix cores of gfx950, forward / backward-data / backward-weights.
 *
 * Replaces, for the layers of the ResNet-18 trunk (reference model/resnet.py:18-67,109-202: 3x3 and 1x1 convolutions,
 * stride 1 or 2, padding k / 2, no dilation, no groups, no bias), torch.nn.Conv2d's forward and the two products of its
 * backward, which the reference runs on cuDNN and PyTorch-ROCm on MIOpen.
 *
 * Layouts (all float32, device pointers, 16-byte aligned):
 *   in   (b, h, w, cin)            -- a channels-last (b, cin, h, w) tensor
 *   wgt  (cout, kh, kw, cin)       -- a channels-last (cout, cin, kh, kw) weight
 *   out  (b, oh, ow, cout),  oh = (h + 2 pad - kh) / stride + 1
 * cin % 64 == 0 and cout % 64 == 0 (every layer of the trunk but the 7x7 stem); other shapes return ISTNET_PN2_EINVAL and
 * the host keeps the framework's convolution for them.
 *
 * All entry points return 0 or a hipError_t / ISTNET_PN2_EINVAL and never synchronise. */
#ifndef ISTNET_CONV_H
#define ISTNET_CONV_H
#include "istnet_pn2.h"
#ifdef __cplusplus
extern "C" {
#endif

/* 1 when the three entry points below take this layer */
ISTNET_PN2_API int istnet_conv_supported(int cin, int cout, int kh, int kw, int stride, int pad);
/* floats of work space the forward (backward_data = 0) / backward-data (1) launch of this shape needs (0: none): the launcher
 * splits K over several workgroups per output tile when the tile count does not fill the chip's workgroup slots in whole
 * rounds (576 tiles on 512 slots), each split accumulating into its own slab, summed in a fixed order; -1: unsupported shape */
ISTNET_PN2_API int istnet_conv_workspace_floats(int backward_data, int b, int h, int w, int cin, int cout, int kh, int kw,
                                                int stride, int pad);
ISTNET_PN2_API int istnet_conv_forward(int b, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad,
                                       const float *in, const float *wgt, float *out, float *ws, void *stream);
/* din (b, h, w, cin) = the gradient of `in` given dout (b, oh, ow, cout) */
ISTNET_PN2_API int istnet_conv_backward_data(int b, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad,
                                             const float *dout, const float *wgt, float *din, float *ws, void *stream);
/* dwgt (cout, kh, kw, cin) = the gradient of `wgt`: split-K over the pixels, partials part [istnet_conv_wrw_splits()][cout kh kw cin]
 * summed in a fixed order (deterministic) */
ISTNET_PN2_API int istnet_conv_wrw_splits(int b, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad);
ISTNET_PN2_API int istnet_conv_backward_weights(int b, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad,
                                                const float *in, const float *dout, float *part, float *dwgt, void *stream);

/* process-wide launch options.  key 1: arithmetic of istnet_conv_forward -- 0 (default) exact fp32 matrix cores
 * (v_mfma_f32_32x32x2_f32); 1 split precision, an OPT-IN experiment: every fp32 operand is split exactly into three bf16
 * terms and the product evaluated as six v_mfma_f32_32x32x16_bf16 products with fp32 accumulation (the dropped terms are
 * <= 2^-24 relative: fp32-class accuracy at ~2.6x the fp32 matrix rate).  Returns 0, or ISTNET_PN2_EINVAL. */
ISTNET_PN2_API int istnet_conv_set_tuning(int key, int value);
ISTNET_PN2_API int istnet_conv_get_tuning(int key);

#ifdef __cplusplus
}
#endif
#endif
