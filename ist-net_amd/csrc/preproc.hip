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

}  // namespace

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
