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

// ============================================================================================
// BatchNorm2d (training statistics) + PReLU (+ Dropout2d mask) of a channels-last map as two streaming passes per
// direction (model/modules.py:25-34: every decoder stage ends conv -> BatchNorm2d -> PReLU, and Dropout2d follows two of
// them, :63-65).  The framework runs MIOpen's two-kernel BatchNorm, a PReLU kernel and a dropout multiply: 4 reads + 3
// writes of the map forward, ~7 reads + 3 writes backward; here 2 + 1 and 4 + 1.
// Layout: rows = B * H * W pixels of C channels (C % 4 == 0, C <= 1024); a workgroup walks a range of pixels with
// thread = (pixel lane, channel quad), so a wave reads whole contiguous pixels.
// ============================================================================================
// pixels per workgroup in the statistics passes: 512 for the decoder's maps (>= 1 M pixels per batch), fewer for the trunk's
// (18 432 pixels at 24 x 24 x 32 would be 36 workgroups): a power of two aiming at ~2 048 workgroups, at least 32
__host__ __device__ inline int stat_pixels(long long rows) {
  int pix = 32;
  while (pix < 512 && (long long)pix * 2048 < rows) pix *= 2;
  return pix;
}

// pass 1 forward: per-channel sum and sum of squares of the pixels [blockIdx.x * pix, +pix), pix = stat_pixels(rows)
__global__ __launch_bounds__(kThreads) void nhwc_stats_kernel(long long rows, int c4, const float4* __restrict__ y,
                                                              float* __restrict__ part_sum, float* __restrict__ part_sq,
                                                              int nparts) {
  extern __shared__ float4 red4[];                  // [2][lanes_p][c4]
  const int lanes_p = kThreads / c4;                // pixel lanes (host: c4 divides 256 or the tail threads idle)
  const int q = threadIdx.x % c4, pl = threadIdx.x / c4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sq = s;
  if (pl < lanes_p) {
    const int pix = stat_pixels(rows);
    const long long p0 = (long long)blockIdx.x * pix, p1 = min(p0 + pix, rows);
    for (long long p = p0 + pl; p < p1; p += lanes_p) {
      const float4 v = y[p * c4 + q];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w;
    }
    red4[pl * c4 + q] = s;
    red4[(lanes_p + pl) * c4 + q] = sq;
  }
  __syncthreads();
  if (threadIdx.x < c4) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    for (int k = 0; k < lanes_p; ++k) {
      const float4 u = red4[k * c4 + q], w = red4[(lanes_p + k) * c4 + q];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
    }
    const int c = 4 * q;
    part_sum[(size_t)(c + 0) * nparts + blockIdx.x] = a.x; part_sum[(size_t)(c + 1) * nparts + blockIdx.x] = a.y;
    part_sum[(size_t)(c + 2) * nparts + blockIdx.x] = a.z; part_sum[(size_t)(c + 3) * nparts + blockIdx.x] = a.w;
    part_sq[(size_t)(c + 0) * nparts + blockIdx.x] = b.x; part_sq[(size_t)(c + 1) * nparts + blockIdx.x] = b.y;
    part_sq[(size_t)(c + 2) * nparts + blockIdx.x] = b.z; part_sq[(size_t)(c + 3) * nparts + blockIdx.x] = b.w;
  }
}

__device__ __forceinline__ float prelu1(float u, float a) { return u > 0.f ? u : a * u; }

// pass 2 forward: z = prelu(scale_c y + shift_c [+ res]) * mask[b][c]   (res: the identity branch of a residual block)
__global__ __launch_bounds__(kThreads) void nhwc_bn_prelu_apply_kernel(long long n4, int c4, long long hw_c4,
                                                                       const float4* __restrict__ y,
                                                                       const float4* __restrict__ scale,
                                                                       const float4* __restrict__ shift,
                                                                       const float* __restrict__ slope,
                                                                       const float4* __restrict__ mask,
                                                                       const float4* __restrict__ res,
                                                                       float4* __restrict__ z) {
  const float a = *slope;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kThreads) {
    const int q = (int)(i % c4);
    const float4 v = y[i], sc = scale[q], sh = shift[q];
    float4 u = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
    if (res != nullptr) {
      const float4 r = res[i];
      u.x += r.x; u.y += r.y; u.z += r.z; u.w += r.w;
    }
    float4 o;
    o.x = prelu1(u.x, a); o.y = prelu1(u.y, a); o.z = prelu1(u.z, a); o.w = prelu1(u.w, a);
    if (mask != nullptr) {
      const float4 m = mask[(i / hw_c4) * c4 + q];
      o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
    }
    z[i] = o;
  }
}

// pass 1 backward: g = dz * mask * prelu'(u), u = scale y + shift: per-channel sum g, sum g y; the slope gradient
// sum dz * mask * min(u, 0) over everything (one partial per workgroup)
__global__ __launch_bounds__(kThreads) void nhwc_bn_prelu_bwd_stats_kernel(
    long long rows, int c4, long long hw, const float4* __restrict__ y, const float4* __restrict__ dz,
    const float4* __restrict__ scale, const float4* __restrict__ shift, const float* __restrict__ slope,
    const float4* __restrict__ mask, const float4* __restrict__ zs, float4* __restrict__ gout,
    float* __restrict__ part_g, float* __restrict__ part_gy, float* __restrict__ part_slope, int nparts) {
  extern __shared__ float4 red4[];
  __shared__ float sred[kThreads / 64];
  const int lanes_p = kThreads / c4;
  const int q = threadIdx.x % c4, pl = threadIdx.x / c4;
  const float a = *slope;
  float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sgy = sg;
  float ds = 0.f;
  if (pl < lanes_p) {
    const float4 sc = scale[q], sh = shift[q];
    const int pix = stat_pixels(rows);
    const long long p0 = (long long)blockIdx.x * pix, p1 = min(p0 + pix, rows);
    for (long long p = p0 + pl; p < p1; p += lanes_p) {
      const float4 v = y[p * c4 + q];
      float4 d = dz[p * c4 + q];
      if (mask != nullptr) {
        const float4 m = mask[(p / hw) * c4 + q];
        d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w;
      }
      float ux = v.x * sc.x + sh.x, uy = v.y * sc.y + sh.y, uz = v.z * sc.z + sh.z, uw = v.w * sc.w + sh.w;
      if (zs != nullptr) {        // residual block: the activation's argument is u + res; its sign is the saved output's
        const float4 zv = zs[p * c4 + q];      // (slope >= 0; the slope gradient is not formed on this route)
        ux = zv.x; uy = zv.y; uz = zv.z; uw = zv.w;
      } else {
        ds += (d.x * fminf(ux, 0.f) + d.y * fminf(uy, 0.f)) + (d.z * fminf(uz, 0.f) + d.w * fminf(uw, 0.f));
      }
      const float gx = ux > 0.f ? d.x : a * d.x, gy = uy > 0.f ? d.y : a * d.y;
      const float gz = uz > 0.f ? d.z : a * d.z, gw = uw > 0.f ? d.w : a * d.w;
      if (gout != nullptr) gout[p * c4 + q] = make_float4(gx, gy, gz, gw);
      sg.x += gx; sg.y += gy; sg.z += gz; sg.w += gw;
      sgy.x += gx * v.x; sgy.y += gy * v.y; sgy.z += gz * v.z; sgy.w += gw * v.w;
    }
    red4[pl * c4 + q] = sg;
    red4[(lanes_p + pl) * c4 + q] = sgy;
  }
  ds = wave_sum64(ds);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = ds;
  __syncthreads();
  if (threadIdx.x == 0) part_slope[blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  if (threadIdx.x < c4) {
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    for (int k = 0; k < lanes_p; ++k) {
      const float4 u = red4[k * c4 + q], w = red4[(lanes_p + k) * c4 + q];
      s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
      s1.x += w.x; s1.y += w.y; s1.z += w.z; s1.w += w.w;
    }
    const int c = 4 * q;
    part_g[(size_t)(c + 0) * nparts + blockIdx.x] = s0.x; part_g[(size_t)(c + 1) * nparts + blockIdx.x] = s0.y;
    part_g[(size_t)(c + 2) * nparts + blockIdx.x] = s0.z; part_g[(size_t)(c + 3) * nparts + blockIdx.x] = s0.w;
    part_gy[(size_t)(c + 0) * nparts + blockIdx.x] = s1.x; part_gy[(size_t)(c + 1) * nparts + blockIdx.x] = s1.y;
    part_gy[(size_t)(c + 2) * nparts + blockIdx.x] = s1.z; part_gy[(size_t)(c + 3) * nparts + blockIdx.x] = s1.w;
  }
}

// pass 2 backward: dy = ca_c g + cb_c + cc_c y   (ca, cb, cc from the finalize of the statistics)
__global__ __launch_bounds__(kThreads) void nhwc_bn_prelu_bwd_apply_kernel(
    long long n4, int c4, long long hw_c4, const float4* __restrict__ y, const float4* __restrict__ dz,
    const float4* __restrict__ scale, const float4* __restrict__ shift, const float* __restrict__ slope,
    const float4* __restrict__ mask, const float4* __restrict__ ca, const float4* __restrict__ cb,
    const float4* __restrict__ cc, float4* __restrict__ dy) {
  const float a = *slope;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kThreads) {
    const int q = (int)(i % c4);
    const float4 v = y[i], sc = scale[q], sh = shift[q], fa = ca[q], fb = cb[q], fc = cc[q];
    float4 d = dz[i];
    if (mask != nullptr) {
      const float4 m = mask[(i / hw_c4) * c4 + q];
      d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w;
    }
    float4 o;
    o.x = fa.x * ((v.x * sc.x + sh.x > 0.f) ? d.x : a * d.x) + fb.x + fc.x * v.x;
    o.y = fa.y * ((v.y * sc.y + sh.y > 0.f) ? d.y : a * d.y) + fb.y + fc.y * v.y;
    o.z = fa.z * ((v.z * sc.z + sh.z > 0.f) ? d.z : a * d.z) + fb.z + fc.z * v.z;
    o.w = fa.w * ((v.w * sc.w + sh.w > 0.f) ? d.w : a * d.w) + fb.w + fc.w * v.w;
    dy[i] = o;
  }
}

}  // namespace

// ============================================================================================
// The decoder's `final` stage evaluated at the chosen pixels only (rgb_branch._FinalAtChosenFn; reference
// model/modules.py:63-67 + model/ist_net.py:41-45): training-mode BatchNorm statistics of z = W u + b over ALL pixels
// follow from the first and second moments of the stage's input u (rows, 64) -- ONE pass over u here (fp32 MFMA,
// v_mfma_f32_32x32x2_f32: the 64 x 64 Gram matrix as 2 x 2 tiles, one per wave, K = pixels) instead of a framework
// column sum (0.9 TB/s) plus a batched library product over a second pass -- and its backward reaches every pixel
// through those statistics as the affine map  du_p = A u_p + c0  (nhwc_rowmix64_kernel: M = pixels, K = N = 64).
// Both kernels are HBM-bound streaming passes over the 302 MB map of the training batch.
// ============================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

constexpr int kGramC = 64;
// grid: nparts workgroups, each a contiguous range of `rows_per` rows (a multiple of 32); partials
// part_s2[nparts][64][64], part_s1[nparts][64] (rows past `rows` count as zero)
__global__ __launch_bounds__(kThreads) void nhwc_gram64_kernel(long long rows, long long rows_per,
                                                               const float* __restrict__ u, float* __restrict__ part_s2,
                                                               float* __restrict__ part_s1) {
  __shared__ __attribute__((aligned(16))) float tile[2][32][kGramC];      // [buffer][k = pixel][m = channel]: k-major for both operands
  __shared__ float colred[16][kGramC];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int tm = wv >> 1, tn = wv & 1;
  const long long r0 = (long long)blockIdx.x * rows_per;
  const long long r1 = r0 + rows_per < rows ? r0 + rows_per : rows;
  const int nchunks = r1 > r0 ? (int)((r1 - r0 + 31) / 32) : 0;
  // a thread stages two float4 per chunk: element e = tid + 256 i -> pixel e / 16, channels 4 (e % 16) .. +3 (the same
  // channels in every chunk: its running column sums are four registers per i)
  float4 stage[2], colsum[2];
  colsum[0] = colsum[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + kThreads * i;
      const long long row = r0 + (long long)t * 32 + e / 16;
      stage[i] = row < r1 ? *reinterpret_cast<const float4*>(u + row * kGramC + (e % 16) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + kThreads * i;
      *reinterpret_cast<float4*>(&tile[buf][e / 16][(e % 16) * 4]) = stage[i];
      colsum[i].x += stage[i].x; colsum[i].y += stage[i].y; colsum[i].z += stage[i].z; colsum[i].w += stage[i].w;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (nchunks > 0) { load(0); put(0); }
  __syncthreads();
  for (int t = 0; t < nchunks; ++t) {
    const int buf = t & 1;
    load(t + 1 < nchunks ? t + 1 : t);             // unconditional issue (the tail re-reads its chunk): exact wait counts
    const float* ap = &tile[buf][half][32 * tm + l31];
    const float* bp = &tile[buf][half][32 * tn + l31];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s2 * kGramC], bp[2 * s2 * kGramC], acc, 0, 0, 0);
    if (t + 1 < nchunks) put(buf ^ 1);
    __syncthreads();
  }
  float* o2 = part_s2 + (size_t)blockIdx.x * kGramC * kGramC;
#pragma unroll
  for (int r = 0; r < 16; ++r) o2[(32 * tm + mfma_row(r, lane)) * kGramC + 32 * tn + l31] = acc[r];
  // column sums: 16 threads share a channel quad (tid % 16; the same quad for both of a thread's elements), fixed-order
  // sum through LDS
  const float4 cs = make_float4(colsum[0].x + colsum[1].x, colsum[0].y + colsum[1].y, colsum[0].z + colsum[1].z,
                                colsum[0].w + colsum[1].w);
  *reinterpret_cast<float4*>(&colred[tid / 16][(tid % 16) * 4]) = cs;
  __syncthreads();
  if (tid < kGramC) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a += colred[k][tid];
    part_s1[(size_t)blockIdx.x * kGramC + tid] = a;
  }
}

// fixed-order float64 sum of the partials: s2 (64 x 64) and s1 (64) as float64.  One workgroup = 64 consecutive output
// elements x 16 slices of the partial index (1024 threads): a thread sums every 16th partial with eight loads in flight,
// the 16 slice sums meet in LDS in slice order.  (A thread per element walking all partials alone was 250 us of dependent
// loads -- three times the streaming pass it finishes.)
__global__ __launch_bounds__(1024) void nhwc_gram64_reduce_kernel(int nparts, const float* __restrict__ part_s2,
                                                                   const float* __restrict__ part_s1,
                                                                   double* __restrict__ s2, double* __restrict__ s1) {
  __shared__ double red[16][64];
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;                      // 0 .. 4095: s2; 4096 .. 4159: s1
  const bool is2 = e < kGramC * kGramC;
  const float* src = is2 ? part_s2 + e : part_s1 + (e - kGramC * kGramC);
  const size_t pitch = is2 ? (size_t)kGramC * kGramC : (size_t)kGramC;
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int k = sl;
  for (; k + 7 * 16 < nparts; k += 8 * 16) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(k + 16 * j) * pitch];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += (double)v[j];
  }
  for (; k < nparts; k += 16) a[0] += (double)src[(size_t)k * pitch];
  red[sl][el] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (sl == 0) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][el];
    if (is2) s2[e] = t; else s1[e - kGramC * kGramC] = t;
  }
}

// out[p][j] = c0[j] + sum_i u[p][i] * a[j][i]   (rows x 64) x (64 x 64)^T: one wave = 32 pixels x 64 output channels per
// chunk, the A^T fragments (64 registers) loaded once; u staged through LDS ([pixel][channel], odd pitch), the output
// leaves as 128-byte runs (a half-wave writes 32 consecutive channels of one pixel)
__global__ __launch_bounds__(kThreads) void nhwc_rowmix64_kernel(long long rows, const float* __restrict__ u,
                                                                 const float* __restrict__ a, const float* __restrict__ c0,
                                                                 float* __restrict__ out) {
  constexpr int LD = kGramC + 1;
  __shared__ float tile[2][128][LD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  float bfrag[2][32];                       // B[k = i][n = j] = a[j][i]: j = 32 t + l31, i = 2 s + half
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) bfrag[t][s2] = a[(size_t)(32 * t + l31) * kGramC + 2 * s2 + half];
  const float cj0 = c0[l31], cj1 = c0[32 + l31];
  const long long nblk = (rows + 127) / 128;
  float4 stage[8];                          // 128 x 64 floats = 2048 float4 = 8 per thread
  auto load = [&](long long blk) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + kThreads * i;
      const long long row = blk * 128 + e / 16;
      stage[i] = row < rows ? *reinterpret_cast<const float4*>(u + row * kGramC + (e % 16) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + kThreads * i;
      float* d = &tile[buf][e / 16][(e % 16) * 4];
      d[0] = stage[i].x; d[1] = stage[i].y; d[2] = stage[i].z; d[3] = stage[i].w;
    }
  };
  long long blk = blockIdx.x;
  if (blk < nblk) { load(blk); put(0); }
  __syncthreads();
  int buf = 0;
  for (; blk < nblk; blk += gridDim.x, buf ^= 1) {
    const long long nxt = blk + gridDim.x < nblk ? blk + gridDim.x : blk;
    load(nxt);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const float* ap = &tile[buf][32 * wv + l31][half];      // A[m = pixel][k = i]
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) {
      const float av = ap[2 * s2];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bfrag[0][s2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bfrag[1][s2], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long row = blk * 128 + 32 * wv + mfma_row(r, lane);
      if (row < rows) {
        out[row * kGramC + l31] = acc0[r] + cj0;
        out[row * kGramC + 32 + l31] = acc1[r] + cj1;
      }
    }
    if (blk + gridDim.x < nblk) put(buf ^ 1);
    __syncthreads();
  }
}

// ---- the rest of the final-at-chosen stage (rgb_branch._FinalAtChosenFn): everything between the moments and the
// stage's outputs / gradients, which was ~85 framework launches of small float64 algebra per step ----
constexpr int kFinC = 64;          // input channels of the stage (the decoder's last PSPUpsample)
constexpr int kFinMaxOut = 512;    // output channels handled (the model has 128)

__device__ __forceinline__ double wave_sum64d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// finalize of the backward statistics pass of a decoder stage (the float64 algebra of istnet_bn_finalize_bwd) that also
// leaves (i) the column sums of dy = ca g + cb + cc y, which the preceding convolution's bias gradient is:
// sum_p dy = ca sum g + rows (cb + cc mean) -- analytically zero, BatchNorm removes the bias; the reference computes the same
// round-off -- and (ii) the PReLU slope gradient (sum of the per-workgroup partials).  grid = C workgroups.
__global__ __launch_bounds__(kThreads) void nhwc_bn_prelu_bwd_finalize_kernel(int C, int nt, double count,
                                                                              const float* __restrict__ part_g,
                                                                              const float* __restrict__ part_gy,
                                                                              const float* __restrict__ part_slope,
                                                                              const float* __restrict__ gamma,
                                                                              const float* __restrict__ bn,
                                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                              float* __restrict__ bwdc, float* __restrict__ colsum,
                                                                              float* __restrict__ dslope) {
  __shared__ double sh[3][kThreads / 64];
  const int c = blockIdx.x, tid = threadIdx.x;
  const float* pa = part_g + (size_t)c * nt;
  const float* pb = part_gy + (size_t)c * nt;
  const bool slope_wg = dslope != nullptr && c == 0;
  double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0}, e = 0.0;
  int i = tid;
  for (; i + 3 * kThreads < nt; i += 4 * kThreads) {
    float x[4], y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { x[k] = pa[i + k * kThreads]; y[k] = pb[i + k * kThreads]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] += (double)x[k]; b[k] += (double)y[k]; }
  }
  for (; i < nt; i += kThreads) { a[0] += (double)pa[i]; b[0] += (double)pb[i]; }
  if (slope_wg)
    for (int k = tid; k < nt; k += kThreads) e += (double)part_slope[k];
  double sa = (a[0] + a[1]) + (a[2] + a[3]), sb = (b[0] + b[1]) + (b[2] + b[3]);
  sa = wave_sum64d(sa); sb = wave_sum64d(sb); e = wave_sum64d(e);
  if ((tid & 63) == 0) { sh[0][tid >> 6] = sa; sh[1][tid >> 6] = sb; sh[2][tid >> 6] = e; }
  __syncthreads();
  if (tid == 0) {
    const double sg = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]), sgy = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    const double mean = bn[2 * C + c], istd = bn[3 * C + c];
    const double dg = (sgy - mean * sg) * istd;
    dgamma[c] = (float)dg;
    dbeta[c] = (float)sg;
    const double gsc = (double)gamma[c] * istd;
    const double c1 = sg / count, c2 = dg / count;
    const float ca = (float)gsc, cb = (float)(-gsc * c1 + gsc * mean * istd * c2), cc = (float)(-gsc * istd * c2);
    bwdc[0 * C + c] = ca; bwdc[1 * C + c] = cb; bwdc[2 * C + c] = cc;
    if (colsum) colsum[c] = (float)((double)ca * sg + count * ((double)cb + (double)cc * mean));
    if (slope_wg) *dslope = (float)((sh[2][0] + sh[2][1]) + (sh[2][2] + sh[2][3]));
  }
}

// batch statistics of z = W u + b over all `npix` pixels from the moments of u; grid = cout workgroups of 64 threads.
// stat[0][c] = mean, stat[1][c] = 1 / sqrt(var + eps), stat[2][c] = biased var (float64); running statistics updated as
// BatchNorm2d does (momentum read from device memory, unbiased variance).
__global__ __launch_bounds__(64) void final_stats_kernel(int cout, double npix, double eps, const double* __restrict__ s1,
                                                         const double* __restrict__ s2, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ momentum_p,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                                         double* __restrict__ stat) {
  __shared__ double wc[kFinC], mu_u[kFinC];
  const int c = blockIdx.x, j = threadIdx.x;
  wc[j] = (double)w[(size_t)c * kFinC + j];
  mu_u[j] = s1[j] / npix;
  __syncthreads();
  double t = 0.0;                    // (Cov w_c)[j]
  for (int i = 0; i < kFinC; ++i) t += (s2[(size_t)i * kFinC + j] / npix - mu_u[j] * mu_u[i]) * wc[i];    // S2 is symmetric
  double var = wave_sum64d(wc[j] * t);
  double mean = wave_sum64d(wc[j] * mu_u[j]);
  if (j == 0) {
    mean += (double)bias[c];
    var = var > 0.0 ? var : 0.0;
    stat[c] = mean;
    stat[cout + c] = 1.0 / sqrt(var + eps);
    stat[2 * cout + c] = var;
    if (running_mean) {
      const float m = *momentum_p;
      const float unb = (float)(var * (npix / (npix > 1.0 ? npix - 1.0 : 1.0)));
      running_mean[c] = running_mean[c] + m * ((float)mean - running_mean[c]);      // lerp(start, end, w), as the framework
      running_var[c] = running_var[c] + m * (unb - running_var[c]);
    }
  }
}

// z = W u_sel + b at the chosen pixels -> zhat = (z - mean) istd, y = prelu(zhat gamma + beta); y and zhat leave as
// (B, Cout, N).  One workgroup = 64 rows (b, n); a thread = one row x 32 output channels per pass of 128 channels.
__global__ __launch_bounds__(kThreads) void final_chosen_fwd_kernel(
    int n, long long hw, int cout, long long rows, const float* __restrict__ u, const long long* __restrict__ choose,
    const float* __restrict__ w, const float* __restrict__ bias, const double* __restrict__ stat,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ slope_p,
    float* __restrict__ y, float* __restrict__ zhat) {
  __shared__ float ut[64][kFinC + 1];
  __shared__ __attribute__((aligned(16))) float wt[kFinC][128];          // W^T of the current 128-channel block
  const int tid = threadIdx.x, r = tid & 63, cg = tid >> 6;
  const long long row0 = (long long)blockIdx.x * 64;
  for (int e = tid; e < 64 * 16; e += kThreads) {                        // gather 64 rows x 64 channels (float4 pieces)
    const long long row = row0 + e / 16;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows) {
      const long long b = row / n, pix = choose[row];
      v = *reinterpret_cast<const float4*>(u + ((size_t)b * hw + pix) * kFinC + (e % 16) * 4);
    }
    float* d = &ut[e / 16][(e % 16) * 4];
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  const float slope = *slope_p;
  const long long row = row0 + r;
  const long long b = row / n, nn = row % n;
  for (int cb = 0; cb < cout; cb += 128) {
    __syncthreads();
    for (int e = tid; e < 128 * kFinC; e += kThreads) {                  // wt[k][c] = w[cb + c][k] (32 KB, L2-resident)
      const int k = e / 128, c = e % 128;
      wt[k][c] = cb + c < cout ? w[(size_t)(cb + c) * kFinC + k] : 0.f;
    }
    __syncthreads();
    float acc[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) acc[q] = 0.f;
#pragma unroll 4
    for (int k = 0; k < kFinC; ++k) {
      const float uv = ut[r][k];
      const float4* wr = reinterpret_cast<const float4*>(&wt[k][cg * 32]);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 wv = wr[q];
        acc[4 * q + 0] = fmaf(uv, wv.x, acc[4 * q + 0]);
        acc[4 * q + 1] = fmaf(uv, wv.y, acc[4 * q + 1]);
        acc[4 * q + 2] = fmaf(uv, wv.z, acc[4 * q + 2]);
        acc[4 * q + 3] = fmaf(uv, wv.w, acc[4 * q + 3]);
      }
    }
    if (row < rows) {
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int c = cb + cg * 32 + q;
        if (c < cout) {
          const float z = acc[q] + bias[c];
          const float zh = (z - (float)stat[c]) * (float)stat[cout + c];
          const float v = zh * gamma[c] + beta[c];
          const size_t o = ((size_t)b * cout + c) * n + nn;
          zhat[o] = zh;
          y[o] = v > 0.f ? v : v * slope;
        }
      }
    }
  }
}

// per (channel, batch entry): sums over the N chosen pixels of gv, gv zhat and g v [v <= 0]  (g = dL/dy, v = zhat gamma +
// beta, gv = g (v > 0 ? 1 : slope)); part[3][cout][b]
__global__ __launch_bounds__(kThreads) void final_chosen_bwd_sums_kernel(int n, int cout, int nb, const float* __restrict__ dy,
                                                                         const float* __restrict__ zhat,
                                                                         const float* __restrict__ gamma,
                                                                         const float* __restrict__ beta,
                                                                         const float* __restrict__ slope_p,
                                                                         float* __restrict__ part) {
  __shared__ float red[3][kThreads / 64];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float ga = gamma[c], be = beta[c], slope = *slope_p;
  const size_t base = ((size_t)b * cout + c) * n;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int i = tid; i < n; i += kThreads) {
    const float g = dy[base + i], zh = zhat[base + i];
    const float v = zh * ga + be;
    const bool neg = v <= 0.f;
    const float gv = neg ? g * slope : g;
    s0 += gv; s1 += gv * zh; s2 += neg ? g * v : 0.f;
  }
  s0 = wave_sum64(s0); s1 = wave_sum64(s1); s2 = wave_sum64(s2);
  if ((tid & 63) == 0) { red[0][tid >> 6] = s0; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
  __syncthreads();
  if (tid < 3) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < kThreads / 64; ++k) t += red[tid][k];
    part[((size_t)tid * cout + c) * nb + b] = t;
  }
}

// the constants of the backward pass; grid = 64 workgroups (row i of A) of 64 threads (column j).  Every workgroup reduces
// the per-batch partial sums itself (fixed order, float64).  bwdc (float64) [4][cout] = a, k, bias - mean, dz scale
// (gamma istd); A = W^T diag(k) W and c0 = W^T (a + k (bias - mean)) as float32 for the dense pass; workgroup 0 writes
// dgamma, dbeta, dslope.
__global__ __launch_bounds__(64) void final_bwd_consts_kernel(int cout, int nb, double npix, const float* __restrict__ part,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ gamma, const double* __restrict__ stat,
                                                              double* __restrict__ bwdc, float* __restrict__ amat,
                                                              float* __restrict__ c0, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ dslope) {
  __shared__ double kk[kFinMaxOut], off[kFinMaxOut];
  const int i = blockIdx.x, j = threadIdx.x;
  double dsl = 0.0;
  for (int c = j; c < cout; c += 64) {
    double p0 = 0.0, p1 = 0.0, p2 = 0.0;
    for (int b = 0; b < nb; ++b) {
      p0 += (double)part[((size_t)0 * cout + c) * nb + b];
      p1 += (double)part[((size_t)1 * cout + c) * nb + b];
      p2 += (double)part[((size_t)2 * cout + c) * nb + b];
    }
    dsl += p2;
    const double istd = stat[cout + c], ga = (double)gamma[c];
    const double a = -(istd * (ga * p0)) / npix;                  // dL/dmean / P
    const double k = -(istd * istd) * (ga * p1) / npix;           // 2 dL/dvar / P
    const double d = (double)bias[c] - stat[c];
    kk[c] = k;
    off[c] = a + k * d;
    if (i == 0) {
      bwdc[c] = a; bwdc[cout + c] = k; bwdc[2 * cout + c] = d; bwdc[3 * cout + c] = ga * istd;
      dbeta[c] = (float)p0; dgamma[c] = (float)p1;
    }
  }
  dsl = wave_sum64d(dsl);
  if (i == 0 && j == 0) *dslope = (float)dsl;
  __syncthreads();
  double t = 0.0, t0 = 0.0;
  for (int c = 0; c < cout; ++c) {
    const double wi = (double)w[(size_t)c * kFinC + i], wj = (double)w[(size_t)c * kFinC + j];
    t += wi * kk[c] * wj;
    t0 += wj * off[c];
  }
  amat[(size_t)i * kFinC + j] = (float)t;
  if (i == 0) c0[j] = (float)t0;
}

// the direct path: dz = gv gamma istd at the chosen pixels; du[pixel] += dz W (atomic: a pixel may be chosen twice) and
// the partial weight gradient dwp[workgroup][c][i] = sum_rows dz[row][c] u_sel[row][i].  A workgroup walks tiles of 64
// rows; cout <= 128 per pass of the channel loop.
__global__ __launch_bounds__(kThreads) void final_chosen_bwd_rows_kernel(
    int n, long long hw, int cout, long long rows, const float* __restrict__ u, const long long* __restrict__ choose,
    const float* __restrict__ w, const float* __restrict__ dy, const float* __restrict__ zhat,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ slope_p,
    const double* __restrict__ bwdc, float* __restrict__ du, float* __restrict__ dwp) {
  __shared__ float ut[64][kFinC + 1];
  __shared__ float dzt[128][65];                                          // [c][row]
  __shared__ __attribute__((aligned(16))) float wl[128][kFinC];           // W rows of the current channel block
  const int tid = threadIdx.x, lo = tid & 63, hi = tid >> 6;
  const float slope = *slope_p;
  const long long ntiles = (rows + 63) / 64;
  for (int cb = 0; cb < cout; cb += 128) {
    __syncthreads();
    for (int e = tid; e < 128 * kFinC; e += kThreads) wl[e / kFinC][e % kFinC] = cb + e / kFinC < cout ? w[(size_t)cb * kFinC + e] : 0.f;
    float dwa[32];                                                        // thread: i = lo, channels hi*32 .. +31
#pragma unroll
    for (int q = 0; q < 32; ++q) dwa[q] = 0.f;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const long long row0 = tile * 64;
      __syncthreads();
      for (int e = tid; e < 64 * 16; e += kThreads) {
        const long long row = row0 + e / 16;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows) v = *reinterpret_cast<const float4*>(u + ((size_t)(row / n) * hw + choose[row]) * kFinC + (e % 16) * 4);
        float* d = &ut[e / 16][(e % 16) * 4];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
      {                                                                   // dz tile: lanes = rows (contiguous n), 32 channels per wave
        const long long row = row0 + lo;
        const long long b = row / n, nn = row % n;
#pragma unroll 4
        for (int q = 0; q < 32; ++q) {
          const int cl = hi * 32 + q, c = cb + cl;
          float dz = 0.f;
          if (row < rows && c < cout) {
            const size_t o = ((size_t)b * cout + c) * n + nn;
            const float g = dy[o], zh = zhat[o];
            const float v = zh * gamma[c] + beta[c];
            dz = (v <= 0.f ? g * slope : g) * (float)bwdc[3 * cout + c];
          }
          dzt[cl][lo] = dz;
        }
      }
      __syncthreads();
      {                                                                   // du_sel[row][i]: row = lo, i = hi*16 .. +15
        float acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll 4
        for (int c = 0; c < 128; ++c) {
          const float dv = dzt[c][lo];
          const float4* wr = reinterpret_cast<const float4*>(&wl[c][hi * 16]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 wv = wr[q];
            acc[4 * q + 0] = fmaf(dv, wv.x, acc[4 * q + 0]);
            acc[4 * q + 1] = fmaf(dv, wv.y, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(dv, wv.z, acc[4 * q + 2]);
            acc[4 * q + 3] = fmaf(dv, wv.w, acc[4 * q + 3]);
          }
        }
        const long long row = row0 + lo;
        if (row < rows) {
          float* d = du + ((size_t)(row / n) * hw + choose[row]) * kFinC + hi * 16;
#pragma unroll
          for (int q = 0; q < 16; ++q) atomicAdd(d + q, acc[q]);
        }
      }
#pragma unroll 2
      for (int rr = 0; rr < 64; ++rr) {                                   // dw partial: i = lo, channels hi*32 .. +31
        const float uv = ut[rr][lo];
#pragma unroll
        for (int q = 0; q < 32; ++q) dwa[q] = fmaf(dzt[hi * 32 + q][rr], uv, dwa[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const int c = cb + hi * 32 + q;
      if (c < cout) dwp[((size_t)blockIdx.x * cout + c) * kFinC + lo] = dwa[q];
    }
  }
}

// dW = sum of the partials + a s1^T + k (W S2 + (bias - mean) s1^T),  db = sum dz + P a + k (W s1 + P (bias - mean));
// grid = cout workgroups of 64 threads (column i)
__global__ __launch_bounds__(64) void final_bwd_params_kernel(int cout, int nwg, int nb, double npix, const float* __restrict__ dwp,
                                                              const float* __restrict__ part, const double* __restrict__ bwdc,
                                                              const float* __restrict__ w, const double* __restrict__ s1,
                                                              const double* __restrict__ s2, float* __restrict__ dw,
                                                              float* __restrict__ db) {
  const int c = blockIdx.x, i = threadIdx.x;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int g = 0;
  for (; g + 8 <= nwg; g += 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = dwp[((size_t)(g + q) * cout + c) * kFinC + i];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] += (double)v[q];
  }
  for (; g < nwg; ++g) acc[0] += (double)dwp[((size_t)g * cout + c) * kFinC + i];
  double direct = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  const double a = bwdc[c], k = bwdc[cout + c], d = bwdc[2 * cout + c];
  double ws2 = 0.0;
  for (int j = 0; j < kFinC; ++j) ws2 += (double)w[(size_t)c * kFinC + j] * s2[(size_t)j * kFinC + i];
  dw[(size_t)c * kFinC + i] = (float)(direct + a * s1[i] + k * (ws2 + d * s1[i]));
  const double ws1 = wave_sum64d((double)w[(size_t)c * kFinC + i] * s1[i]);
  if (i == 0) {
    double p0 = 0.0;
    for (int b = 0; b < nb; ++b) p0 += (double)part[(size_t)c * nb + b];
    db[c] = (float)(bwdc[3 * cout + c] * p0 + npix * a + k * (ws1 + npix * d));
  }
}

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

int istnet_nhwc_stat_parts(long long rows) {
  if (rows <= 0) return 0;
  const int pix = stat_pixels(rows);
  return (int)((rows + pix - 1) / pix);
}

static bool nhwc_ok(long long rows, int c) { return rows > 0 && c >= 4 && c <= 1024 && c % 4 == 0; }

int istnet_nhwc_channel_stats(long long rows, int c, const float* y, float* part_sum, float* part_sq, void* stream) {
  if (!nhwc_ok(rows, c) || c / 4 > kThreads || !y || !part_sum || !part_sq) return ISTNET_PN2_EINVAL;
  const int c4 = c / 4, nparts = istnet_nhwc_stat_parts(rows), lanes_p = kThreads / c4;
  hipLaunchKernelGGL(nhwc_stats_kernel, dim3(nparts), dim3(kThreads), (size_t)2 * lanes_p * c4 * sizeof(float4),
                     (hipStream_t)stream, rows, c4, reinterpret_cast<const float4*>(y), part_sum, part_sq, nparts);
  return (int)hipGetLastError();
}

int istnet_nhwc_bn_prelu_apply(int b, long long hw, int c, const float* y, const float* bn, const float* slope,
                               const float* mask, float* z, void* stream) {
  if (b <= 0 || !nhwc_ok(hw, c) || !y || !bn || !slope || !z) return ISTNET_PN2_EINVAL;
  const long long n4 = (long long)b * hw * (c / 4);
  long long blocks = (n4 + kThreads * 4 - 1) / (kThreads * 4);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(nhwc_bn_prelu_apply_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, n4, c / 4,
                     hw * (c / 4), reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(bn),
                     reinterpret_cast<const float4*>(bn + c), slope, reinterpret_cast<const float4*>(mask),
                     (const float4*)nullptr, reinterpret_cast<float4*>(z));
  return (int)hipGetLastError();
}

int istnet_nhwc_bn_act_res_apply(int b, long long hw, int c, const float* y, const float* bn, const float* slope,
                                 const float* res, float* z, void* stream) {
  if (b <= 0 || !nhwc_ok(hw, c) || !y || !bn || !slope || !res || !z) return ISTNET_PN2_EINVAL;
  const long long n4 = (long long)b * hw * (c / 4);
  long long blocks = (n4 + kThreads * 4 - 1) / (kThreads * 4);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(nhwc_bn_prelu_apply_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, n4, c / 4,
                     hw * (c / 4), reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(bn),
                     reinterpret_cast<const float4*>(bn + c), slope, (const float4*)nullptr,
                     reinterpret_cast<const float4*>(res), reinterpret_cast<float4*>(z));
  return (int)hipGetLastError();
}

int istnet_nhwc_bn_act_res_bwd_stats(int b, long long hw, int c, const float* y, const float* dz, const float* z,
                                     const float* bn, const float* slope, float* g, float* part_g, float* part_gy,
                                     float* part_slope, void* stream) {
  const long long rows = (long long)b * hw;
  if (b <= 0 || !nhwc_ok(hw, c) || c / 4 > kThreads || !y || !dz || !z || !bn || !slope || !g || !part_g || !part_gy ||
      !part_slope)
    return ISTNET_PN2_EINVAL;
  const int c4 = c / 4, nparts = istnet_nhwc_stat_parts(rows), lanes_p = kThreads / c4;
  hipLaunchKernelGGL(nhwc_bn_prelu_bwd_stats_kernel, dim3(nparts), dim3(kThreads), (size_t)2 * lanes_p * c4 * sizeof(float4),
                     (hipStream_t)stream, rows, c4, hw, reinterpret_cast<const float4*>(y),
                     reinterpret_cast<const float4*>(dz), reinterpret_cast<const float4*>(bn),
                     reinterpret_cast<const float4*>(bn + c), slope, (const float4*)nullptr,
                     reinterpret_cast<const float4*>(z), reinterpret_cast<float4*>(g), part_g, part_gy, part_slope, nparts);
  return (int)hipGetLastError();
}

int istnet_nhwc_bn_prelu_bwd_stats(int b, long long hw, int c, const float* y, const float* dz, const float* bn,
                                   const float* slope, const float* mask, float* part_g, float* part_gy, float* part_slope,
                                   void* stream) {
  const long long rows = (long long)b * hw;
  if (b <= 0 || !nhwc_ok(hw, c) || c / 4 > kThreads || !y || !dz || !bn || !slope || !part_g || !part_gy || !part_slope)
    return ISTNET_PN2_EINVAL;
  const int c4 = c / 4, nparts = istnet_nhwc_stat_parts(rows), lanes_p = kThreads / c4;
  hipLaunchKernelGGL(nhwc_bn_prelu_bwd_stats_kernel, dim3(nparts), dim3(kThreads), (size_t)2 * lanes_p * c4 * sizeof(float4),
                     (hipStream_t)stream, rows, c4, hw, reinterpret_cast<const float4*>(y),
                     reinterpret_cast<const float4*>(dz), reinterpret_cast<const float4*>(bn),
                     reinterpret_cast<const float4*>(bn + c), slope, reinterpret_cast<const float4*>(mask),
                     (const float4*)nullptr, (float4*)nullptr, part_g, part_gy, part_slope, nparts);
  return (int)hipGetLastError();
}

int istnet_nhwc_bn_prelu_bwd_apply(int b, long long hw, int c, const float* y, const float* dz, const float* bn,
                                   const float* bwdc, const float* slope, const float* mask, float* dy, void* stream) {
  if (b <= 0 || !nhwc_ok(hw, c) || !y || !dz || !bn || !bwdc || !slope || !dy) return ISTNET_PN2_EINVAL;
  const long long n4 = (long long)b * hw * (c / 4);
  long long blocks = (n4 + kThreads * 4 - 1) / (kThreads * 4);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(nhwc_bn_prelu_bwd_apply_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, n4,
                     c / 4, hw * (c / 4), reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(dz),
                     reinterpret_cast<const float4*>(bn), reinterpret_cast<const float4*>(bn + c), slope,
                     reinterpret_cast<const float4*>(mask), reinterpret_cast<const float4*>(bwdc),
                     reinterpret_cast<const float4*>(bwdc + c), reinterpret_cast<const float4*>(bwdc + 2 * c),
                     reinterpret_cast<float4*>(dy));
  return (int)hipGetLastError();
}

int istnet_nhwc_bn_prelu_bwd_finalize(int c, int nparts, double count, const float* part_g, const float* part_gy,
                                      const float* part_slope, const float* gamma, const float* bn, float* dgamma,
                                      float* dbeta, float* bwdc, float* colsum, float* dslope, void* stream) {
  if (c <= 0 || nparts <= 0 || count <= 0 || !part_g || !part_gy || !gamma || !bn || !dgamma || !dbeta || !bwdc ||
      (dslope && !part_slope))
    return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(nhwc_bn_prelu_bwd_finalize_kernel, dim3(c), dim3(kThreads), 0, (hipStream_t)stream, c, nparts, count, part_g,
                     part_gy, part_slope, gamma, bn, dgamma, dbeta, bwdc, colsum, dslope);
  return (int)hipGetLastError();
}

int istnet_nhwc_gram64_parts(long long rows) {
  long long parts = (rows + 32 * 36 - 1) / (32 * 36);            // >= 36 chunks of 32 pixels per workgroup
  return (int)(parts < 1 ? 1 : (parts > 512 ? 512 : parts));
}

int istnet_nhwc_gram64(long long rows, const float* u, float* part_s2, float* part_s1, double* s2, double* s1, void* stream) {
  if (rows <= 0 || !u || !part_s2 || !part_s1 || !s2 || !s1 || ((uintptr_t)u & 15)) return ISTNET_PN2_EINVAL;
  const int nparts = istnet_nhwc_gram64_parts(rows);
  const long long rows_per = ((rows + nparts - 1) / nparts + 31) / 32 * 32;
  hipLaunchKernelGGL(nhwc_gram64_kernel, dim3(nparts), dim3(kThreads), 0, (hipStream_t)stream, rows, rows_per, u, part_s2,
                     part_s1);
  hipLaunchKernelGGL(nhwc_gram64_reduce_kernel, dim3((kGramC * kGramC + kGramC) / 64), dim3(1024), 0, (hipStream_t)stream,
                     nparts, part_s2, part_s1, s2, s1);
  return (int)hipGetLastError();
}

int istnet_nhwc_rowmix64(long long rows, const float* u, const float* a, const float* c0, float* out, void* stream) {
  if (rows <= 0 || !u || !a || !c0 || !out || (((uintptr_t)u | (uintptr_t)out) & 15)) return ISTNET_PN2_EINVAL;
  long long blocks = (rows + 127) / 128;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(nhwc_rowmix64_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, rows, u, a, c0, out);
  return (int)hipGetLastError();
}

int istnet_final_chosen_workgroups(long long rows) {
  long long tiles = (rows + 63) / 64;
  return (int)(tiles < 1 ? 1 : (tiles > 256 ? 256 : tiles));
}

int istnet_final_chosen_forward(int b, long long hw, int n, int cout, const float* u, const long long* choose, const float* w,
                                const float* bias, const float* gamma, const float* beta, const float* slope,
                                float* running_mean, float* running_var, const float* momentum_p, double eps, float* part_s2,
                                float* part_s1, double* s2, double* s1, double* stat, float* y, float* zhat, void* stream) {
  if (b <= 0 || hw <= 0 || n <= 0 || cout <= 0 || cout > kFinMaxOut || !u || !choose || !w || !bias || !gamma || !beta ||
      !slope || !part_s2 || !part_s1 || !s2 || !s1 || !stat || !y || !zhat || ((uintptr_t)u & 15) ||
      ((running_mean || running_var) && (!running_mean || !running_var || !momentum_p)))
    return ISTNET_PN2_EINVAL;
  const long long npix = (long long)b * hw, rows = (long long)b * n;
  const int rc = istnet_nhwc_gram64(npix, u, part_s2, part_s1, s2, s1, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(final_stats_kernel, dim3(cout), dim3(64), 0, (hipStream_t)stream, cout, (double)npix, eps, s1, s2, w, bias,
                     momentum_p, running_mean, running_var, stat);
  hipLaunchKernelGGL(final_chosen_fwd_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(kThreads), 0, (hipStream_t)stream, n, hw,
                     cout, rows, u, choose, w, bias, stat, gamma, beta, slope, y, zhat);
  return (int)hipGetLastError();
}

int istnet_final_chosen_backward(int b, long long hw, int n, int cout, const float* u, const long long* choose, const float* w,
                                 const float* bias, const float* gamma, const float* beta, const float* slope, const double* s2,
                                 const double* s1, const double* stat, const float* dy, const float* zhat, float* part,
                                 double* bwdc, float* amat, float* c0, float* dwp, float* du, float* dw, float* db,
                                 float* dgamma, float* dbeta, float* dslope, void* stream) {
  if (b <= 0 || hw <= 0 || n <= 0 || cout <= 0 || cout > kFinMaxOut || !u || !choose || !w || !bias || !gamma || !beta ||
      !slope || !s2 || !s1 || !stat || !dy || !zhat || !part || !bwdc || !amat || !c0 || !dwp || !du || !dw || !db ||
      !dgamma || !dbeta || !dslope || (((uintptr_t)u | (uintptr_t)du) & 15))
    return ISTNET_PN2_EINVAL;
  const long long npix = (long long)b * hw, rows = (long long)b * n;
  const int nwg = istnet_final_chosen_workgroups(rows);
  hipLaunchKernelGGL(final_chosen_bwd_sums_kernel, dim3(cout, b), dim3(kThreads), 0, (hipStream_t)stream, n, cout, b, dy, zhat,
                     gamma, beta, slope, part);
  hipLaunchKernelGGL(final_bwd_consts_kernel, dim3(kFinC), dim3(64), 0, (hipStream_t)stream, cout, b, (double)npix, part, w, bias,
                     gamma, stat, bwdc, amat, c0, dgamma, dbeta, dslope);
  const int rc = istnet_nhwc_rowmix64(npix, u, amat, c0, du, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(final_chosen_bwd_rows_kernel, dim3(nwg), dim3(kThreads), 0, (hipStream_t)stream, n, hw, cout, rows, u, choose,
                     w, dy, zhat, gamma, beta, slope, bwdc, du, dwp);
  hipLaunchKernelGGL(final_bwd_params_kernel, dim3(cout), dim3(64), 0, (hipStream_t)stream, cout, nwg, b, (double)npix, dwp, part,
                     bwdc, w, s1, s2, dw, db);
  return (int)hipGetLastError();
}

}  // extern "C"
