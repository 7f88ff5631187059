// Input preparation kernels (include/istnet_preproc.h).  HBM/latency-trivial: one thread per sampled pixel.
#include <hip/hip_runtime.h>

#include "../../include/istnet_preproc.h"

namespace {

constexpr int kThreads = 256;

template <typename DepthT>
__global__ void backproject_choose_kernel(int n, int h, int w, const DepthT* __restrict__ depth, long long depth_stride,
                                          const int* __restrict__ bbox, const int* __restrict__ choose, double fx,
                                          double fy, double cx, double cy, double norm_scale, int img_size,
                                          float* __restrict__ pts, long long* __restrict__ choose_out) {
  const int inst = blockIdx.y;
  const int j = blockIdx.x * kThreads + threadIdx.x;
  if (j >= n) return;
  const int rmin = bbox[inst * 4 + 0], rmax = bbox[inst * 4 + 1], cmin = bbox[inst * 4 + 2], cmax = bbox[inst * 4 + 3];
  const int crop_cols = max(cmax - cmin, 1), crop_w = max(rmax - rmin, 1);
  const int flat = choose[(size_t)inst * n + j];
  // pts[rmin:rmax, cmin:cmax].reshape(-1, 3)[choose]   [dataset.py:209]
  const int y = min(max(rmin + flat / crop_cols, 0), h - 1), x = min(max(cmin + flat % crop_cols, 0), w - 1);
  const DepthT d = depth[(size_t)inst * depth_stride + (size_t)y * w + x];
  double z;
  float z_out;
  if constexpr (sizeof(DepthT) == 2) {   // uint16 / python float -> float64   [dataset.py:205]
    z = (double)d / norm_scale;
    z_out = (float)z;
  } else {                               // float32 / python float stays float32 in numpy
    z_out = (float)d / (float)norm_scale;
    z = (double)z_out;
  }
  float* out = pts + ((size_t)inst * n + j) * 3;
  out[0] = (float)(((double)x - cx) * z / fx);   // (xmap - cam_cx) * pts2 / cam_fx   [dataset.py:206]
  out[1] = (float)(((double)y - cy) * z / fy);   // [dataset.py:207]
  out[2] = z_out;
  // crop coordinates -> resized crop   [dataset.py:226-231]
  const double ratio = (double)img_size / (double)crop_w;
  const double row = (double)(flat / crop_w), col = (double)(flat % crop_w);
  choose_out[(size_t)inst * n + j] = (long long)(floor(row * ratio) * (double)img_size + floor(col * ratio));
}

// cv::resize(INTER_LINEAR) of an 8-bit image as OpenCV's generic C++ path evaluates it (imgproc/resize.cpp): source tap and
// two weights in 11-bit fixed point per destination coordinate
struct LinTap { int s0, s1, w0, w1; };
__device__ inline LinTap linear_tap(int d, int ssize, int dsize) {
  const double scale = (double)ssize / (double)dsize;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  LinTap t;
  t.s0 = s;
  t.s1 = min(s + 1, ssize - 1);
  t.w1 = (int)rintf(f * 2048.f);              // saturate_cast<short>(f * INTER_RESIZE_COEF_SCALE): round half to even
  t.w0 = (int)rintf((1.f - f) * 2048.f);
  return t;
}

// one thread per destination pixel: crop [rmin:rmax, cmin:cmax] of the (h, w, 3) uint8 image, resize to S x S, optional
// channel reversal, then ToTensor + Normalize ((u8 / 255 - mean) / std, float32, IEEE division)
__global__ void crop_resize_normalize_kernel(int h, int w, const unsigned char* __restrict__ image, long long image_stride,
                                             int reverse, const int* __restrict__ bbox, int S, float m0, float m1, float m2,
                                             float d0, float d1, float d2, unsigned char* __restrict__ out_u8,
                                             float* __restrict__ out) {
  const int inst = blockIdx.y;
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= S * S) return;
  const int dy = p / S, dx = p % S;
  const int rmin = min(max(bbox[inst * 4 + 0], 0), h - 1), rmax = min(max(bbox[inst * 4 + 1], rmin + 1), h);
  const int cmin = min(max(bbox[inst * 4 + 2], 0), w - 1), cmax = min(max(bbox[inst * 4 + 3], cmin + 1), w);
  const LinTap tx = linear_tap(dx, cmax - cmin, S), ty = linear_tap(dy, rmax - rmin, S);
  const unsigned char* img = image + (size_t)inst * image_stride;
  const unsigned char* r0 = img + ((size_t)(rmin + ty.s0) * w + cmin) * 3;
  const unsigned char* r1 = img + ((size_t)(rmin + ty.s1) * w + cmin) * 3;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {d0, d1, d2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int cs = reverse ? 2 - c : c;
    const int h0 = (int)r0[tx.s0 * 3 + cs] * tx.w0 + (int)r0[tx.s1 * 3 + cs] * tx.w1;      // HResizeLinear
    const int h1 = (int)r1[tx.s0 * 3 + cs] * tx.w0 + (int)r1[tx.s1 * 3 + cs] * tx.w1;
    int v = (((ty.w0 * (h0 >> 4)) >> 16) + ((ty.w1 * (h1 >> 4)) >> 16) + 2) >> 2;          // VResizeLinear, 8-bit
    v = min(max(v, 0), 255);
    if (out_u8 != nullptr) out_u8[((size_t)inst * S * S + p) * 3 + c] = (unsigned char)v;
    if (out != nullptr) out[((size_t)inst * 3 + c) * S * S + p] = ((float)v / 255.f - mean[c]) / stdv[c];
  }
}

}  // namespace

extern "C" int istnet_crop_resize_normalize(int count, int h, int w, const unsigned char* image, long long image_stride,
                                            int reverse_channels, const int* bbox, int img_size, const float* mean,
                                            const float* std, unsigned char* out_u8, float* out, void* stream) {
  if (count < 0 || h < 1 || w < 1 || img_size < 1 || image_stride < 0 || !image || !bbox || !mean || !std ||
      (!out_u8 && !out))
    return ISTNET_PN2_EINVAL;
  if (count == 0) return 0;
  const dim3 grid((img_size * img_size + kThreads - 1) / kThreads, count);
  crop_resize_normalize_kernel<<<grid, kThreads, 0, (hipStream_t)stream>>>(h, w, image, image_stride, reverse_channels != 0,
                                                                            bbox, img_size, mean[0], mean[1], mean[2], std[0],
                                                                            std[1], std[2], out_u8, out);
  return (int)hipGetLastError();
}

extern "C" int istnet_backproject_choose(int count, int n, int h, int w, const void* depth, int depth_kind,
                                         long long depth_stride, const int* bbox, const int* choose, double fx,
                                         double fy, double cx, double cy, double norm_scale, int img_size, float* pts,
                                         long long* choose_out, void* stream) {
  if (count < 0 || n < 0 || h < 1 || w < 1 || img_size < 1 || depth_stride < 0 || (depth_kind != 0 && depth_kind != 1))
    return ISTNET_PN2_EINVAL;
  if (count == 0 || n == 0) return 0;
  const dim3 grid((n + kThreads - 1) / kThreads, count);
  hipStream_t s = (hipStream_t)stream;
  if (depth_kind == 0)
    backproject_choose_kernel<unsigned short><<<grid, kThreads, 0, s>>>(
        n, h, w, (const unsigned short*)depth, depth_stride, bbox, choose, fx, fy, cx, cy, norm_scale, img_size, pts, choose_out);
  else
    backproject_choose_kernel<float><<<grid, kThreads, 0, s>>>(n, h, w, (const float*)depth, depth_stride, bbox, choose,
                                                                fx, fy, cx, cy, norm_scale, img_size, pts, choose_out);
  return (int)hipGetLastError();
}
