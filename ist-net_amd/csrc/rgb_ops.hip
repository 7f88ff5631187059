// RGB-branch decoder pieces that are not convolutions (include/istnet_rgb.h): streaming kernels, HBM-bound.
#include <hip/hip_runtime.h>

#include "../../include/istnet_rgb.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxParts = 2048;

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// grid-stride over float4 groups; one partial of sum dy*x*[x<=0] per workgroup, summed in a fixed order
__global__ __launch_bounds__(kThreads) void prelu_bwd_kernel(long long n4, long long n, const float* __restrict__ x,
                                                             const float* __restrict__ dy, const float* __restrict__ a,
                                                             float* __restrict__ dx, float* __restrict__ part) {
  const float slope = *a;
  float acc = 0.f;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    const float4 gv = reinterpret_cast<const float4*>(dy)[i];
    float4 o;
    o.x = xv.x > 0.f ? gv.x : gv.x * slope;  acc += xv.x > 0.f ? 0.f : gv.x * xv.x;
    o.y = xv.y > 0.f ? gv.y : gv.y * slope;  acc += xv.y > 0.f ? 0.f : gv.y * xv.y;
    o.z = xv.z > 0.f ? gv.z : gv.z * slope;  acc += xv.z > 0.f ? 0.f : gv.z * xv.z;
    o.w = xv.w > 0.f ? gv.w : gv.w * slope;  acc += xv.w > 0.f ? 0.f : gv.w * xv.w;
    reinterpret_cast<float4*>(dx)[i] = o;
  }
  if (blockIdx.x == 0) {       // tail (n % 4 elements)
    const long long i = 4 * n4 + threadIdx.x;
    if (i < n) {
      const float xv = x[i], gv = dy[i];
      dx[i] = xv > 0.f ? gv : gv * slope;
      acc += xv > 0.f ? 0.f : gv * xv;
    }
  }
  __shared__ float red[kThreads / 64];
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one thread per (input pixel, channel quad).  src(o) = o * r, r = (in-1)/(out-1) in f32 as the forward computes it;
// output o touches inputs i0 = floor(src) (weight 1 - l) and min(i0 + 1, in - 1) (weight l), l = src - i0.
__device__ __forceinline__ float tap_weight(int o, int i, float r, int in) {
  const float src = r * (float)o;
  const int i0 = (int)src;
  const float l = src - (float)i0;
  const int i1 = min(i0 + 1, in - 1);
  return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}
__global__ __launch_bounds__(kThreads) void upsample_ac_bwd_nhwc_kernel(int c4, int hin, int win, int hout, int wout,
                                                                        float rh, float rw, float inv_rh, float inv_rw,
                                                                        const float* __restrict__ dy,
                                                                        float* __restrict__ dx, long long total) {
  const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (t >= total) return;
  const int q = (int)(t % c4);
  long long pix = t / c4;
  const int ix = (int)(pix % win);
  pix /= win;
  const int iy = (int)(pix % hin);
  const int b = (int)(pix / hin);
  // candidate outputs: src in (i - 1, i + 1); one extra on each side absorbs the rounding of the f32 products
  const int oy0 = max((int)floorf((float)(iy - 1) * inv_rh) - 1, 0), oy1 = min((int)ceilf((float)(iy + 1) * inv_rh) + 1, hout - 1);
  const int ox0 = max((int)floorf((float)(ix - 1) * inv_rw) - 1, 0), ox1 = min((int)ceilf((float)(ix + 1) * inv_rw) + 1, wout - 1);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int oy = oy0; oy <= oy1; ++oy) {
    const float wy = tap_weight(oy, iy, rh, hin);
    if (wy == 0.f) continue;
    const float4* row = reinterpret_cast<const float4*>(dy) + ((size_t)b * hout + oy) * wout * c4 + q;
    for (int ox = ox0; ox <= ox1; ++ox) {
      const float w = wy * tap_weight(ox, ix, rw, win);
      if (w == 0.f) continue;
      const float4 g = row[(size_t)ox * c4];
      acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
    }
  }
  reinterpret_cast<float4*>(dx)[t] = acc;
}

}  // namespace

extern "C" {

int istnet_prelu_bwd_parts(long long n) {
  const long long blocks = (n / 4 + kThreads * 4 - 1) / (kThreads * 4);   // >= 4 float4 per thread
  return (int)(blocks < 1 ? 1 : (blocks > kMaxParts ? kMaxParts : blocks));
}

int istnet_prelu_bwd(long long n, const float* x, const float* dy, const float* a, float* dx, float* part, void* stream) {
  if (n <= 0 || !x || !dy || !a || !dx || !part) return ISTNET_PN2_EINVAL;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(istnet_prelu_bwd_parts(n)), dim3(kThreads), 0, (hipStream_t)stream, n / 4, n,
                     x, dy, a, dx, part);
  return (int)hipGetLastError();
}

int istnet_upsample_bilinear_ac_bwd_nhwc(int b, int c, int hin, int win, int hout, int wout, const float* dy, float* dx,
                                         void* stream) {
  if (b <= 0 || c <= 0 || (c & 3) || hin < 2 || win < 2 || hout < 2 || wout < 2 || !dy || !dx) return ISTNET_PN2_EINVAL;
  if (((uintptr_t)dy | (uintptr_t)dx) & 15) return ISTNET_PN2_EINVAL;
  const float rh = (float)(hin - 1) / (float)(hout - 1), rw = (float)(win - 1) / (float)(wout - 1);
  const long long total = (long long)b * hin * win * (c / 4);
  hipLaunchKernelGGL(upsample_ac_bwd_nhwc_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, c / 4, hin, win, hout, wout, rh, rw, 1.f / rh, 1.f / rw, dy, dx, total);
  return (int)hipGetLastError();
}

}  // extern "C"
