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

// Forward of the same upsample: one thread per (output pixel, channel quad), the four taps as float4 loads (the
// 2x-smaller source stays in L2), the blend in the framework's order
//   h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d),   h1 = src_y - floor(src_y), src_y = oy * (hin-1)/(hout-1).
__global__ __launch_bounds__(kThreads) void upsample_ac_fwd_nhwc_kernel(int c4, int hin, int win, int hout, int wout,
                                                                        float rh, float rw, const float* __restrict__ x,
                                                                        float* __restrict__ y, long long total) {
  const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (t >= total) return;
  const int q = (int)(t % c4);
  long long p = t / c4;
  const int ox = (int)(p % wout); p /= wout;
  const int oy = (int)(p % hout);
  const int b = (int)(p / hout);
  const float sy = rh * (float)oy, sx = rw * (float)ox;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = min(y0 + 1, hin - 1), x1 = min(x0 + 1, win - 1);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float4* base = reinterpret_cast<const float4*>(x) + (size_t)b * hin * win * c4 + q;
  const float4 a = base[((size_t)y0 * win + x0) * c4], bb = base[((size_t)y0 * win + x1) * c4];
  const float4 c = base[((size_t)y1 * win + x0) * c4], d = base[((size_t)y1 * win + x1) * c4];
  float4 o;
  o.x = hy * (hx * a.x + lx * bb.x) + ly * (hx * c.x + lx * d.x);
  o.y = hy * (hx * a.y + lx * bb.y) + ly * (hx * c.y + lx * d.y);
  o.z = hy * (hx * a.z + lx * bb.z) + ly * (hx * c.z + lx * d.z);
  o.w = hy * (hx * a.w + lx * bb.w) + ly * (hx * c.w + lx * d.w);
  reinterpret_cast<float4*>(y)[t] = o;
}

// ---- bilinear 2x upsample (align_corners) followed by a 3x3 convolution, without the convolution at full size ----
// conv3x3(U p) = sum_taps shift_tap(U (W_tap p)): the channel mixing W_tap (Cin -> Cout per tap) is a 1x1 product
// on the SMALL map -- one GEMM p (B h w, Cin) x Wr (Cin, 9 Cout) = q, a quarter of the convolution's flops -- and
// what is left at full size is linear interpolation and nine shifted adds, done here:
//   out[b, y, x, co] = bias[co] + sum_{ky, kx} [ (y+ky-1, x+kx-1) inside ] * bilinear(q[b, :, :, ky*3+kx, co]; y+ky-1, x+kx-1)
// q: (b, hin, win, 9, c) f32, out: (b, hout, wout, c) channels-last; one thread per output pixel and channel quad;
// the 36 taps of a thread hit the L2-resident small map.  Fixed summation order.
__global__ __launch_bounds__(kThreads) void upconv3_fwd_kernel(int c4, int hin, int win, int hout, int wout, float rh,
                                                               float rw, const float* __restrict__ q,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               long long total) {
  const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (t >= total) return;
  const int cq = (int)(t % c4);
  long long p = t / c4;
  const int ox = (int)(p % wout); p /= wout;
  const int oy = (int)(p % hout);
  const int b = (int)(p / hout);
  // (scalar reads: a parameter that lives in a flat optimizer buffer is only 4-byte aligned)
  float4 acc = bias != nullptr ? make_float4(bias[4 * cq], bias[4 * cq + 1], bias[4 * cq + 2], bias[4 * cq + 3])
                               : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* qb = reinterpret_cast<const float4*>(q) + (size_t)b * hin * win * 9 * c4 + cq;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = oy + ky - 1;
    if (yy < 0 || yy >= hout) continue;
    const float sy = rh * (float)yy;
    const int y0 = (int)sy, y1 = min(y0 + 1, hin - 1);
    const float ly = sy - (float)y0, hy = 1.f - ly;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xx = ox + kx - 1;
      if (xx < 0 || xx >= wout) continue;
      const float sx = rw * (float)xx;
      const int x0 = (int)sx, x1 = min(x0 + 1, win - 1);
      const float lx = sx - (float)x0, hx = 1.f - lx;
      const int tap = ky * 3 + kx;
      const float4 a = qb[(((size_t)y0 * win + x0) * 9 + tap) * c4], bb = qb[(((size_t)y0 * win + x1) * 9 + tap) * c4];
      const float4 c = qb[(((size_t)y1 * win + x0) * 9 + tap) * c4], d = qb[(((size_t)y1 * win + x1) * 9 + tap) * c4];
      acc.x += hy * (hx * a.x + lx * bb.x) + ly * (hx * c.x + lx * d.x);
      acc.y += hy * (hx * a.y + lx * bb.y) + ly * (hx * c.y + lx * d.y);
      acc.z += hy * (hx * a.z + lx * bb.z) + ly * (hx * c.z + lx * d.z);
      acc.w += hy * (hx * a.w + lx * bb.w) + ly * (hx * c.w + lx * d.w);
    }
  }
  reinterpret_cast<float4*>(y)[t] = acc;
}

// gradient of the above w.r.t. q, gather form: q[b, iy, ix, tap, :] collects, over the full-size positions (yy, xx)
// whose interpolation reads (iy, ix), weight * dy[b, yy - ky + 1, xx - kx + 1, :] when that output exists.
// One thread per (input pixel, channel quad) for ALL nine taps: the <= 7 candidate rows / columns and their weights
// are formed once (they depend on the pixel only), every dy position of the 9 x 9 window around them is loaded once
// and feeds the up to nine taps it belongs to (dy[oy][ox] is the tap-(ky, kx) term of position (oy + ky - 1, ox + kx - 1)).
// The first version (one thread per tap, a 7 x 7 candidate loop each) was bound by that loop's arithmetic: 1.6 ms per
// training step for the three decoder stages against 0.4 ms of HBM time.
__global__ __launch_bounds__(kThreads) void upconv3_bwd_kernel(int c4, int hin, int win, int hout, int wout, float rh,
                                                               float rw, float inv_rh, float inv_rw,
                                                               const float* __restrict__ dy, float* __restrict__ dq,
                                                               long long total) {
  const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (t >= total) return;
  const int cq = (int)(t % c4);
  long long p = t / c4;
  const int ix = (int)(p % win); p /= win;
  const int iy = (int)(p % hin);
  const int b = (int)(p / hin);
  // full-size rows / columns that can read input row iy / column ix: src in (i - 1, i + 1), at most 5 of them at a 2x
  // ratio; 7 candidates starting one early absorb the rounding of the f32 products (as in upsample_ac_bwd_nhwc_kernel)
  constexpr int NC = 7;
  const int yy0 = (int)floorf((float)(iy - 1) * inv_rh) - 1, xx0 = (int)floorf((float)(ix - 1) * inv_rw) - 1;
  float wy[NC], wx[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int yy = yy0 + j, xx = xx0 + j;
    wy[j] = (yy >= 0 && yy < hout) ? tap_weight(yy, iy, rh, hin) : 0.f;
    wx[j] = (xx >= 0 && xx < wout) ? tap_weight(xx, ix, rw, win) : 0.f;
  }
  float4 acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* gb = reinterpret_cast<const float4*>(dy) + (size_t)b * hout * wout * c4 + cq;
#pragma unroll
  for (int r = 0; r < NC + 2; ++r) {                 // dy row oy = yy0 - 1 + r serves candidate rows j = r + ky - 2
    const int oy = yy0 - 1 + r;
    float wrow[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) { const int j = r + ky - 2; wrow[ky] = (j >= 0 && j < NC) ? wy[j] : 0.f; }
    if (oy < 0 || oy >= hout || (wrow[0] + wrow[1] + wrow[2]) == 0.f) continue;       // weights are >= 0
#pragma unroll
    for (int c = 0; c < NC + 2; ++c) {
      const int ox = xx0 - 1 + c;
      float wcol[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) { const int j = c + kx - 2; wcol[kx] = (j >= 0 && j < NC) ? wx[j] : 0.f; }
      if (ox < 0 || ox >= wout || (wcol[0] + wcol[1] + wcol[2]) == 0.f) continue;
      const float4 g = gb[((size_t)oy * wout + ox) * c4];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float w = wrow[ky] * wcol[kx];
          float4& a = acc[ky * 3 + kx];
          a.x += w * g.x; a.y += w * g.y; a.z += w * g.z; a.w += w * g.w;
        }
    }
  }
  float4* out = reinterpret_cast<float4*>(dq) + (((size_t)b * hin + iy) * win + ix) * 9 * c4 + cq;
#pragma unroll
  for (int k = 0; k < 9; ++k) out[(size_t)k * c4] = acc[k];
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

int istnet_upsample_bilinear_ac_fwd_nhwc(int b, int c, int hin, int win, int hout, int wout, const float* x, float* y,
                                         void* stream) {
  if (b <= 0 || c <= 0 || (c & 3) || hin < 2 || win < 2 || hout < 2 || wout < 2 || !x || !y) return ISTNET_PN2_EINVAL;
  if (((uintptr_t)x | (uintptr_t)y) & 15) return ISTNET_PN2_EINVAL;
  const float rh = (float)(hin - 1) / (float)(hout - 1), rw = (float)(win - 1) / (float)(wout - 1);
  const long long total = (long long)b * hout * wout * (c / 4);
  hipLaunchKernelGGL(upsample_ac_fwd_nhwc_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, c / 4, hin, win, hout, wout, rh, rw, x, y, total);
  return (int)hipGetLastError();
}

int istnet_upconv3_fwd_nhwc(int b, int c, int hin, int win, int hout, int wout, const float* q, const float* bias,
                            float* y, void* stream) {
  if (b <= 0 || c <= 0 || (c & 3) || hin < 2 || win < 2 || hout < 2 || wout < 2 || !q || !y) return ISTNET_PN2_EINVAL;
  if (((uintptr_t)q | (uintptr_t)y) & 15) return ISTNET_PN2_EINVAL;
  const float rh = (float)(hin - 1) / (float)(hout - 1), rw = (float)(win - 1) / (float)(wout - 1);
  const long long total = (long long)b * hout * wout * (c / 4);
  hipLaunchKernelGGL(upconv3_fwd_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, c / 4, hin, win, hout, wout, rh, rw, q, bias, y, total);
  return (int)hipGetLastError();
}

int istnet_upconv3_bwd_nhwc(int b, int c, int hin, int win, int hout, int wout, const float* dy, float* dq,
                            void* stream) {
  if (b <= 0 || c <= 0 || (c & 3) || hin < 2 || win < 2 || hout < 2 || wout < 2 || !dy || !dq) return ISTNET_PN2_EINVAL;
  if (((uintptr_t)dy | (uintptr_t)dq) & 15) return ISTNET_PN2_EINVAL;
  const float rh = (float)(hin - 1) / (float)(hout - 1), rw = (float)(win - 1) / (float)(wout - 1);
  const long long total = (long long)b * hin * win * (c / 4);
  hipLaunchKernelGGL(upconv3_bwd_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, c / 4, hin, win, hout, wout, rh, rw, 1.f / rh, 1.f / rw, dy, dq, total);
  return (int)hipGetLastError();
}

}  // extern "C"
