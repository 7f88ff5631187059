/*
 * istnet_preproc.h -- C ABI (same library, libistnet_pn2.so, and same conventions as istnet_pn2.h) of the
 * per-instance input preparation that the reference does in numpy inside its Dataset classes
 * (SURVEY.md 8f rank 2): depth back-projection at the sampled pixels and the remap of the `choose`
 * indices from crop coordinates to the resized img_size x img_size crop.
 */
#ifndef ISTNET_PREPROC_H_
#define ISTNET_PREPROC_H_

#include "istnet_pn2.h"

#ifdef __cplusplus
extern "C" {
#endif

/* replaces provider/dataset.py:203-210 + :226-231 (TrainingDataset.__getitem__) and :348-355 + :392,401-405
 * (TestDataset.__getitem__), which build the full (h,w,3) float64 point map per image on the host and then
 * keep n of its pixels.  Here only the n sampled pixels of each instance are back-projected.
 *
 *   depth        (h,w) image per instance, device memory; depth_kind 0 = uint16 millimetres (load_depth,
 *                utils/data_utils.py:6-22), 1 = float32 (the output of fill_missing, data_utils.py:516-540).
 *                Instance i reads depth + i*depth_stride ELEMENTS (0 = all instances share one image, as the
 *                instances of one test image do).
 *   bbox         (count,4) int32: rmin, rmax, cmin, cmax of get_bbox (crop = [rmin:rmax, cmin:cmax]).
 *   choose       (count,n) int32: flat row-major indices into the crop (width cmax-cmin), dataset.py:191-199.
 *   fx,fy,cx,cy  camera intrinsics; norm_scale = 1000.0; img_size = side of the resized crop (192).
 *   pts          (count,n,3) f32 out: ((x-cx)*z/fx, (y-cy)*z/fy, z), z = depth/norm_scale, evaluated in
 *                float64 in numpy's promotion order and rounded to f32 once (bit-exact with the host code;
 *                for depth_kind 1, z itself is the f32 quotient, as numpy keeps float32/python-float in f32).
 *   choose_out   (count,n) int64 out: floor(row*ratio)*img_size + floor(col*ratio), row = choose / crop_w,
 *                col = choose % crop_w, crop_w = rmax-rmin, ratio = img_size/crop_w in float64 (the reference
 *                uses the crop HEIGHT for both, get_bbox crops are square).
 * Returns ISTNET_PN2_EINVAL for negative sizes, depth_kind outside {0,1} or img_size < 1.  A `choose` index
 * outside its crop or a crop outside the image is the caller's error (the reference raises IndexError); the kernel
 * clamps the pixel coordinate into the image instead of reading out of bounds. */
ISTNET_PN2_API int istnet_backproject_choose(int count, int n, int h, int w, const void *depth, int depth_kind,
                                             long long depth_stride, const int *bbox, const int *choose,
                                             double fx, double fy, double cx, double cy, double norm_scale,
                                             int img_size, float *pts, long long *choose_out, void *stream);

/* replaces provider/dataset.py:213-219 (TrainingDataset, without the random colour jitter of :218) and :397-399
 * (TestDataset): crop [rmin:rmax, cmin:cmax] of the (h, w, 3) uint8 image, cv2.resize(..., (img_size, img_size),
 * INTER_LINEAR), ToTensor and Normalize(mean, std) -- one thread per output pixel.
 *   image            device memory, (h, w, 3) uint8; instance i reads image + i * image_stride BYTES (0: one image for all
 *                    instances, as the detections of one test image)
 *   reverse_channels 1: channel c of the output is channel 2 - c of the image (cv2.imread returns BGR, dataset.py:214)
 *   bbox             (count, 4) int32 rmin, rmax, cmin, cmax (get_bbox); clamped into the image
 *   mean, std        HOST pointers, 3 floats each (dataset.py:103-105: ImageNet statistics)
 *   out_u8           (count, img_size, img_size, 3) uint8 or NULL: the resized crop (what the colour jitter would take)
 *   out              (count, 3, img_size, img_size) float32 or NULL: ((u8 / 255) - mean) / std in float32, IEEE division
 * The resize is OpenCV's generic 8-bit path: taps and weights from fx = (float)((dx + 0.5) * scale - 0.5) in 11-bit
 * fixed point (round half to even), horizontal pass in int32, vertical pass
 * (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  Bit-exact with the numpy restatement
 * oracle/preproc_oracle.py (parity unpinned: cv2 is not importable in this image). */
ISTNET_PN2_API int istnet_crop_resize_normalize(int count, int h, int w, const unsigned char *image,
                                                long long image_stride, int reverse_channels, const int *bbox,
                                                int img_size, const float *mean, const float *std,
                                                unsigned char *out_u8, float *out, void *stream);

/* replaces fill_in_multiscale of utils/data_utils.py:357-510 as fill_missing calls it (:516-540: fill_type 'multiscale',
 * extrapolate False, blur_type 'bilateral'; provider/dataset.py:172-173,361-362): morphological depth completion of b
 * float32 depth images (b, h, w) already scaled to the unit the thresholds are written in (metres).  Pass by pass the
 * reference's own sequence -- inversion max_depth - d of the valid pixels (d > 0.01), cross-kernel dilations of the
 * far / medium / near bins (3 / 5 / 7), 5x5 closing, masked 5x5 median, 9x9 hole fill under the top mask, six masked 5x5
 * fills, masked 5x5 median, bilateral filter (d = 5, sigmaColor 0.5, sigmaSpace 2.0) on the pixels valid before the
 * blurs, inversion back -- with OpenCV's documented border rules (csrc/depth_fill.hip).  scratch:
 * istnet_depth_fill_scratch_floats(b, h, w) floats.  out (b, h, w). */
ISTNET_PN2_API int istnet_depth_fill_scratch_floats(int b, int h, int w);
ISTNET_PN2_API int istnet_depth_fill_multiscale(int b, int h, int w, const float *depth, float max_depth, float *scratch,
                                                float *out, void *stream);
/* fill_missing itself (utils/data_utils.py:516-540; the Dataset classes call fill_missing(depth, norm_scale, 1)): the raw
 * depth images -- uint16 millimetres as cv2.imread returns them (raw_is_float 0), float32 (1) or float64 (2) -- are scaled by
 * scale_2_80m / cam_scale in float64 and rounded to float32 once (what numpy does) inside the first kernel, completed as
 * above, and scaled back (float32 / scale_2_80m * cam_scale) by the last one: no conversion passes on the caller's side. */
ISTNET_PN2_API int istnet_depth_fill_missing(int b, int h, int w, const void *depth_raw, int raw_is_float, double cam_scale,
                                             double scale_2_80m, float max_depth, float *scratch, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
