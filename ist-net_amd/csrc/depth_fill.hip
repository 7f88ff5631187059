// depth_fill.hip -- `fill_missing` of the reference's data pipeline on the GPU (gfx950).
//
// Reference: utils/data_utils.py:516-540 (fill_missing) -> :357-510 (fill_in_multiscale, the 'multiscale' / 'bilateral'
// branch every Dataset class uses, provider/dataset.py:172-173,361-362): depth completion by morphology -- inversion,
// three binned cross-kernel dilations, a 5x5 closing, 5x5 median blurs, masked hole fills with full 9x9 / 5x5 dilations,
// a bilateral filter -- as ~25 cv2 / numpy passes over a 480 x 640 float image on the host, per image.
// Here the same passes are stencil kernels over a BATCH of images.  Round 5 (profiles/r05_fill_missing_kernels.txt: the
// thread-per-pixel version spent 1.39 ms on 32 images, a quarter of it in 7 full-image passes of the hole filling and a
// seventh in dtype conversions on the framework side): the closing, the medians and the six masked 5 x 5 fills run on LDS
// tiles with halos (the six fills as ONE kernel: halo 12, one read and one write of the image instead of six), the median
// is a 99-exchange selection network (verified on all 2^25 binary inputs, tools/verify_median25.py) instead of a 234-exchange
// partial sort, and the unit conversions of fill_missing happen in the first and the last kernel.  Border rules are OpenCV's documented ones: dilate / erode ignore pixels outside the image
// (morphologyDefaultBorderValue), medianBlur replicates the border, bilateralFilter reflects without repeating the edge
// pixel (BORDER_REFLECT_101) and weights exp(-d^2 / (2 sigma_space^2)) exp(-(dI)^2 / (2 sigma_color^2)) over the disc of
// radius 2 (OpenCV's float path evaluates the colour term through a 4096-bin interpolated table of the same function).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/istnet_preproc.h"

namespace {

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

constexpr float kValid = 0.01f;
constexpr int TW = 64, TH = 32, kT = 256;      // output tile of the LDS kernels: 32 rows x 64 columns, 256 threads

// value of `img` dilated (max) at (r, c) with a full k x k structuring element, pixels outside the image ignored
__device__ __forceinline__ float dilate_at(const float* __restrict__ img, int h, int w, int r, int c, int k) {
  const int rad = k / 2;
  float best = -__builtin_inff();
  for (int dr = -rad; dr <= rad; ++dr) {
    const int rr = r + dr;
    if (rr < 0 || rr >= h) continue;
    for (int dc = -rad; dc <= rad; ++dc) {
      const int cc = c + dc;
      if (cc < 0 || cc >= w) continue;
      best = fmaxf(best, img[(size_t)rr * w + cc]);
    }
  }
  return best;
}

// fill_missing's `dpt / cam_scale * scale_2_80m` (numpy: float64, rounded to float32 once by fill_in_multiscale) and
// s1: inversion of the valid depths.  RAW: 0 = uint16 millimetres, 1 = float32, 2 = float64 (numpy keeps float64 / int32
// input in float64 through the scaling and rounds to float32 once: converting it to float32 first would round twice).
template <int RAW>
__global__ void convert_invert_kernel(long long n, double cam_scale, double scale, float max_depth, const void* __restrict__ raw,
                                      float* __restrict__ depth, float* __restrict__ s1) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double v = RAW == 0   ? (double)reinterpret_cast<const unsigned short*>(raw)[i]
                   : RAW == 1 ? (double)reinterpret_cast<const float*>(raw)[i]
                              : reinterpret_cast<const double*>(raw)[i];
  const float d = (float)(v / cam_scale * scale);
  depth[i] = d;
  s1[i] = d > kValid ? max_depth - d : d;
}

// s2: cross-kernel dilations of the far / medium / near bins (3 / 5 / 7), combined farthest to nearest.  Tile + halo 3 of the
// input depth and of s1 in LDS (positions outside the image: depth 0 = in no bin, never selected).
__global__ __launch_bounds__(kT) void binned_dilate_kernel(int h, int w, const float* __restrict__ depth_all,
                                                            const float* __restrict__ s1_all, float* __restrict__ s2_all) {
  constexpr int H0 = 3, W0 = TW + 2 * H0, R0 = TH + 2 * H0;
  __shared__ float sd[R0 * W0];
  __shared__ float sv[R0 * W0];
  __shared__ unsigned char sin[R0 * W0];
  const int c0 = blockIdx.x * TW, r0 = blockIdx.y * TH, b = blockIdx.z;
  const float* depth = depth_all + (size_t)b * h * w;
  const float* s1 = s1_all + (size_t)b * h * w;
  for (int i = threadIdx.x; i < R0 * W0; i += kT) {
    const int r = r0 - H0 + i / W0, c = c0 - H0 + i % W0;
    const bool in = r >= 0 && r < h && c >= 0 && c < w;
    sd[i] = in ? depth[(size_t)r * w + c] : 0.f;
    sv[i] = in ? s1[(size_t)r * w + c] : 0.f;
    sin[i] = in ? 1 : 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TH * TW; i += kT) {
    const int lr = i / TW + H0, lc = i % TW + H0;
    const int r = r0 + lr - H0, c = c0 + lc - H0;
    if (r >= h || c >= w) continue;
    // dilation of (s1 * bin mask): a neighbour outside its bin contributes 0, one outside the image nothing
    float far = -__builtin_inff(), med = -__builtin_inff(), near = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < 13; ++t) {
      const int dr = t < 7 ? t - 3 : 0, dc = t < 7 ? 0 : (t < 10 ? t - 10 : t - 9);      // column taps, then the row's other six
      const int j = (lr + dr) * W0 + lc + dc;
      if (!sin[j]) continue;
      const float d = sd[j], v = sv[j];
      const int dist = dr != 0 ? (dr < 0 ? -dr : dr) : (dc < 0 ? -dc : dc);
      if (dist <= 1) far = fmaxf(far, d > 2.0f ? v : 0.f);
      if (dist <= 2) med = fmaxf(med, (d > 1.0f && d <= 2.0f) ? v : 0.f);
      near = fmaxf(near, (d > kValid && d <= 1.0f) ? v : 0.f);
    }
    float out = sv[lr * W0 + lc];
    if (far > kValid) out = far;
    if (med > kValid) out = med;
    if (near > kValid) out = near;
    s2_all[((size_t)b * h + r) * w + c] = out;
  }
}

// s3 = MORPH_CLOSE with the full 5 x 5 element: dilate, then erode; both ignore the outside of the image.  One tile per
// workgroup: the source with a halo of 4 in LDS, the dilation on the halo-2 region into a second LDS tile (positions outside
// the image hold +inf there, so the erosion ignores them), the erosion on the tile.
__global__ __launch_bounds__(kT) void close5_kernel(int h, int w, const float* __restrict__ src_all, float* __restrict__ dst_all) {
  constexpr int H0 = 4, W0 = TW + 2 * H0, R0 = TH + 2 * H0;      // source region
  constexpr int H1 = 2, W1 = TW + 2 * H1, R1 = TH + 2 * H1;      // dilated region
  __shared__ float s0[R0 * W0];
  __shared__ float s1[R1 * W1];
  const int c0 = blockIdx.x * TW, r0 = blockIdx.y * TH, b = blockIdx.z;
  const float* src = src_all + (size_t)b * h * w;
  for (int i = threadIdx.x; i < R0 * W0; i += kT) {
    const int r = r0 - H0 + i / W0, c = c0 - H0 + i % W0;
    s0[i] = (r >= 0 && r < h && c >= 0 && c < w) ? src[(size_t)r * w + c] : -__builtin_inff();
  }
  __syncthreads();
  for (int i = threadIdx.x; i < R1 * W1; i += kT) {
    const int lr = i / W1, lc = i % W1;
    const int r = r0 - H1 + lr, c = c0 - H1 + lc;
    float best = __builtin_inff();                 // outside the image: ignored by the erosion
    if (r >= 0 && r < h && c >= 0 && c < w) {
      best = -__builtin_inff();
#pragma unroll
      for (int dr = 0; dr < 5; ++dr)
#pragma unroll
        for (int dc = 0; dc < 5; ++dc) best = fmaxf(best, s0[(lr + dr) * W0 + lc + dc]);
    }
    s1[i] = best;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TH * TW; i += kT) {
    const int lr = i / TW, lc = i % TW;
    const int r = r0 + lr, c = c0 + lc;
    if (r >= h || c >= w) continue;
    float best = __builtin_inff();
#pragma unroll
    for (int dr = 0; dr < 5; ++dr)
#pragma unroll
      for (int dc = 0; dc < 5; ++dc) best = fminf(best, s1[(lr + dr) * W1 + lc + dc]);
    dst_all[((size_t)b * h + r) * w + c] = best;
  }
}

// median of 25 values: N. Devillard's 99-exchange selection network (after it v[12] is the 13th smallest); checked against
// a sort on all 2^25 binary inputs (0-1 principle), tools/verify_median25.py
#define ISTNET_CE(a, b) { const float lo_ = fminf(v[a], v[b]); v[b] = fmaxf(v[a], v[b]); v[a] = lo_; }
__device__ __forceinline__ float median25(float (&v)[25]) {
  ISTNET_CE(0, 1) ISTNET_CE(3, 4) ISTNET_CE(2, 4) ISTNET_CE(2, 3) ISTNET_CE(6, 7) ISTNET_CE(5, 7) ISTNET_CE(5, 6) ISTNET_CE(9, 10)
  ISTNET_CE(8, 10) ISTNET_CE(8, 9) ISTNET_CE(12, 13) ISTNET_CE(11, 13) ISTNET_CE(11, 12) ISTNET_CE(15, 16) ISTNET_CE(14, 16)
  ISTNET_CE(14, 15) ISTNET_CE(18, 19) ISTNET_CE(17, 19) ISTNET_CE(17, 18) ISTNET_CE(21, 22) ISTNET_CE(20, 22) ISTNET_CE(20, 21)
  ISTNET_CE(23, 24) ISTNET_CE(2, 5) ISTNET_CE(3, 6) ISTNET_CE(0, 6) ISTNET_CE(0, 3) ISTNET_CE(4, 7) ISTNET_CE(1, 7) ISTNET_CE(1, 4)
  ISTNET_CE(11, 14) ISTNET_CE(8, 14) ISTNET_CE(8, 11) ISTNET_CE(12, 15) ISTNET_CE(9, 15) ISTNET_CE(9, 12) ISTNET_CE(13, 16)
  ISTNET_CE(10, 16) ISTNET_CE(10, 13) ISTNET_CE(20, 23) ISTNET_CE(17, 23) ISTNET_CE(17, 20) ISTNET_CE(21, 24) ISTNET_CE(18, 24)
  ISTNET_CE(18, 21) ISTNET_CE(19, 22) ISTNET_CE(8, 17) ISTNET_CE(9, 18) ISTNET_CE(0, 18) ISTNET_CE(0, 9) ISTNET_CE(10, 19)
  ISTNET_CE(1, 19) ISTNET_CE(1, 10) ISTNET_CE(11, 20) ISTNET_CE(2, 20) ISTNET_CE(2, 11) ISTNET_CE(12, 21) ISTNET_CE(3, 21)
  ISTNET_CE(3, 12) ISTNET_CE(13, 22) ISTNET_CE(4, 22) ISTNET_CE(4, 13) ISTNET_CE(14, 23) ISTNET_CE(5, 23) ISTNET_CE(5, 14)
  ISTNET_CE(15, 24) ISTNET_CE(6, 24) ISTNET_CE(6, 15) ISTNET_CE(7, 16) ISTNET_CE(7, 19) ISTNET_CE(13, 21) ISTNET_CE(15, 23)
  ISTNET_CE(7, 13) ISTNET_CE(7, 15) ISTNET_CE(1, 9) ISTNET_CE(3, 11) ISTNET_CE(5, 17) ISTNET_CE(11, 17) ISTNET_CE(9, 17)
  ISTNET_CE(4, 10) ISTNET_CE(6, 12) ISTNET_CE(7, 14) ISTNET_CE(4, 6) ISTNET_CE(4, 7) ISTNET_CE(12, 14) ISTNET_CE(10, 14)
  ISTNET_CE(6, 7) ISTNET_CE(10, 12) ISTNET_CE(6, 10) ISTNET_CE(6, 17) ISTNET_CE(12, 17) ISTNET_CE(7, 17) ISTNET_CE(7, 10)
  ISTNET_CE(12, 18) ISTNET_CE(7, 12) ISTNET_CE(10, 18) ISTNET_CE(12, 20) ISTNET_CE(10, 20) ISTNET_CE(10, 12)
  return v[12];
}
#undef ISTNET_CE

// dst = valid(src) ? median5x5(src) : src, valid = src > 0.01 (and inside the top mask when given); border replicated
// (cv2.medianBlur).  The tile with its halo of 2 is staged in LDS with clamped coordinates.
__global__ __launch_bounds__(kT) void median_masked_kernel(int h, int w, const float* __restrict__ src_all,
                                                            const int* __restrict__ top_row, float* __restrict__ dst_all) {
  constexpr int H0 = 2, W0 = TW + 2 * H0, R0 = TH + 2 * H0;
  __shared__ float s0[R0 * W0];
  const int c0 = blockIdx.x * TW, r0 = blockIdx.y * TH, b = blockIdx.z;
  const float* src = src_all + (size_t)b * h * w;
  for (int i = threadIdx.x; i < R0 * W0; i += kT) {
    const int r = min(max(r0 - H0 + i / W0, 0), h - 1), c = min(max(c0 - H0 + i % W0, 0), w - 1);
    s0[i] = src[(size_t)r * w + c];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TH * TW; i += kT) {
    const int lr = i / TW, lc = i % TW;
    const int r = r0 + lr, c = c0 + lc;
    if (r >= h || c >= w) continue;
    const float x = s0[(lr + H0) * W0 + lc + H0];
    float out = x;
    if (x > kValid && (top_row == nullptr || r >= top_row[b * w + c])) {
      float v[25];
#pragma unroll
      for (int dr = 0; dr < 5; ++dr)
#pragma unroll
        for (int dc = 0; dc < 5; ++dc) v[dr * 5 + dc] = s0[(lr + dr) * W0 + lc + dc];
      out = median25(v);
    }
    dst_all[((size_t)b * h + r) * w + c] = out;
  }
}

// first row of each column with a valid pixel (np.argmax of the boolean column: 0 when the column has none)
__global__ void top_row_kernel(int h, int w, const float* __restrict__ img_all, int* __restrict__ top_row) {
  const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y;
  if (c >= w) return;
  const float* img = img_all + (size_t)b * h * w;
  int top = 0;
  for (int r = 0; r < h; ++r)
    if (img[(size_t)r * w + c] > kValid) { top = r; break; }
  top_row[b * w + c] = top;
}

// s5: dst = (empty pixel inside the top mask) ? dilate_full_9x9(src) : src;  empty: !(src > 0.01)
__global__ __launch_bounds__(kT) void fill_empty9_kernel(int h, int w, const float* __restrict__ src_all,
                                                          const int* __restrict__ top_row, float* __restrict__ dst_all) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
  if (c >= w || r >= h) return;
  const float* src = src_all + (size_t)b * h * w;
  const float v = src[(size_t)r * w + c];
  const bool empty = !(v > kValid) && r >= top_row[b * w + c];
  dst_all[((size_t)b * h + r) * w + c] = empty ? dilate_at(src, h, w, r, c, 9) : v;
}

// s7: SIX rounds of  x = (x < 0.01 inside the top mask) ? dilate_full_5x5(x) : x  (data_utils.py:464-468) in one kernel: the
// tile with a halo of 12 in LDS, round i valid on the region of halo 12 - 2 i, two LDS buffers in turn.  Positions outside
// the image hold -inf and are never updated, so every dilation ignores them as cv2.dilate does.
__global__ __launch_bounds__(kT) void fill6_kernel(int h, int w, const float* __restrict__ src_all,
                                                    const int* __restrict__ top_row, float* __restrict__ dst_all) {
  constexpr int H0 = 12, W0 = TW + 2 * H0, R0 = TH + 2 * H0;
  __shared__ float buf[2][R0 * W0];
  __shared__ int top[W0];
  const int c0 = blockIdx.x * TW, r0 = blockIdx.y * TH, b = blockIdx.z;
  const float* src = src_all + (size_t)b * h * w;
  for (int i = threadIdx.x; i < W0; i += kT) {
    const int c = c0 - H0 + i;
    top[i] = (c >= 0 && c < w) ? top_row[b * w + c] : 0;
  }
  int any_empty = 0;
  for (int i = threadIdx.x; i < R0 * W0; i += kT) {
    const int r = r0 - H0 + i / W0, c = c0 - H0 + i % W0;
    const bool in = r >= 0 && r < h && c >= 0 && c < w;
    const float x = in ? src[(size_t)r * w + c] : -__builtin_inff();
    buf[0][i] = x;
    buf[1][i] = x;           // the rim that a round does not recompute keeps its (never read again) value
    any_empty |= (in && x < kValid) ? 1 : 0;
  }
  // a tile (with its halo) without an empty pixel is a fixed point of all six rounds
  if (!__syncthreads_or(any_empty)) {
    for (int i = threadIdx.x; i < TH * TW; i += kT) {
      const int lr = i / TW, lc = i % TW;
      const int r = r0 + lr, c = c0 + lc;
      if (r < h && c < w) dst_all[((size_t)b * h + r) * w + c] = buf[0][(lr + H0) * W0 + lc + H0];
    }
    return;
  }
  int cur = 0;
  for (int it = 1; it <= 6; ++it) {
    const int halo = H0 - 2 * it;                      // region this round computes
    const int rw = TW + 2 * halo, rh = TH + 2 * halo, off = H0 - halo;
    const float* s = buf[cur];
    float* d = buf[cur ^ 1];
    for (int i = threadIdx.x; i < rw * rh; i += kT) {
      const int lr = off + i / rw, lc = off + i % rw;
      const int r = r0 - H0 + lr, c = c0 - H0 + lc;
      float x = s[lr * W0 + lc];
      if (r >= 0 && r < h && c >= 0 && c < w && x < kValid && r >= top[lc]) {
        float best = -__builtin_inff();
#pragma unroll
        for (int dr = -2; dr <= 2; ++dr)
#pragma unroll
          for (int dc = -2; dc <= 2; ++dc) best = fmaxf(best, s[(lr + dr) * W0 + lc + dc]);
        x = best;
      }
      d[lr * W0 + lc] = x;
    }
    __syncthreads();
    cur ^= 1;
  }
  for (int i = threadIdx.x; i < TH * TW; i += kT) {
    const int lr = i / TW, lc = i % TW;
    const int r = r0 + lr, c = c0 + lc;
    if (r < h && c < w) dst_all[((size_t)b * h + r) * w + c] = buf[cur][(lr + H0) * W0 + lc + H0];
  }
}

// bilateral filter d = 5 (disc of radius 2), BORDER_REFLECT_101; applied where valid, then the final inversion and
// fill_missing's `/ scale_2_80m * cam_scale` (float32 operations in the reference: final_dpt is float32).  Tile + halo 2
// in LDS with reflected coordinates; the three spatial weights (r^2 = 1, 2, 4) are computed once per thread.
__global__ __launch_bounds__(kT) void bilateral_invert_kernel(int h, int w, float sigma_color, float sigma_space, float max_depth,
                                                               float scale, float cam_scale, const float* __restrict__ pre_all,
                                                               const float* __restrict__ src_all, const int* __restrict__ top_row,
                                                               float* __restrict__ dst_all) {
  constexpr int H0 = 2, W0 = TW + 2 * H0, R0 = TH + 2 * H0;
  __shared__ float s0[R0 * W0];
  const int c0 = blockIdx.x * TW, r0 = blockIdx.y * TH, b = blockIdx.z;
  const float* src = src_all + (size_t)b * h * w;
  for (int i = threadIdx.x; i < R0 * W0; i += kT) {
    int r = r0 - H0 + i / W0, c = c0 - H0 + i % W0;
    r = r < 0 ? -r : (r >= h ? 2 * h - 2 - r : r);
    c = c < 0 ? -c : (c >= w ? 2 * w - 2 - c : c);
    r = min(max(r, 0), h - 1);                       // (only positions no in-image pixel reads)
    c = min(max(c, 0), w - 1);
    s0[i] = src[(size_t)r * w + c];
  }
  __syncthreads();
  const float gc = -0.5f / (sigma_color * sigma_color), gs = -0.5f / (sigma_space * sigma_space);
  const float ws[5] = {1.0f, expf(1.0f * gs), expf(2.0f * gs), 0.f, expf(4.0f * gs)};      // by r^2; expf(0) == 1 exactly
  for (int i = threadIdx.x; i < TH * TW; i += kT) {
    const int lr = i / TW + H0, lc = i % TW + H0;
    const int r = r0 + lr - H0, c = c0 + lc - H0;
    if (r >= h || c >= w) continue;
    const float v = s0[lr * W0 + lc];
    // `valid` of the reference is the mask computed BEFORE the second median blur (data_utils.py:477-478 reused at :487)
    const float pre = pre_all[((size_t)b * h + r) * w + c];
    float out = v;
    if (pre > kValid && r >= top_row[b * w + c]) {
      float sum = 0.f, wsum = 0.f;
#pragma unroll
      for (int dr = -2; dr <= 2; ++dr)
#pragma unroll
        for (int dc = -2; dc <= 2; ++dc) {
          if (dr * dr + dc * dc > 4) continue;
          const float u = s0[(lr + dr) * W0 + lc + dc];
          const float wgt = ws[dr * dr + dc * dc] * expf((u - v) * (u - v) * gc);
          sum += u * wgt;
          wsum += wgt;
        }
      out = sum / wsum;
    }
    out = out > kValid ? max_depth - out : out;
    dst_all[((size_t)b * h + r) * w + c] = out / scale * cam_scale;
  }
}

static int fill_impl(int b, int h, int w, const void* raw, int raw_kind, double cam_scale, double scale, float max_depth,
                     float* scratch, float* out, void* stream) {
  if (b <= 0 || h < 5 || w < 5 || !raw || !scratch || !out || max_depth <= 0.f || !(cam_scale > 0.0) || !(scale > 0.0))
    return ISTNET_PN2_EINVAL;
  hipStream_t st = as_stream(stream);
  const long long n = (long long)b * h * w;
  float* a = scratch;
  float* bb = scratch + n;
  float* depth = scratch + 2 * n;
  int* top = reinterpret_cast<int*>(scratch + 3 * n);
  const dim3 rows4(ceil_div(w, 64), ceil_div(h, 4), b), tiles(ceil_div(w, TW), ceil_div(h, TH), b), blk(kT);
  const dim3 g1((unsigned)((n + 255) / 256));
  if (raw_kind == 0)
    hipLaunchKernelGGL(convert_invert_kernel<0>, g1, dim3(256), 0, st, n, cam_scale, scale, max_depth, raw, depth, a);        // s1
  else if (raw_kind == 1)
    hipLaunchKernelGGL(convert_invert_kernel<1>, g1, dim3(256), 0, st, n, cam_scale, scale, max_depth, raw, depth, a);
  else
    hipLaunchKernelGGL(convert_invert_kernel<2>, g1, dim3(256), 0, st, n, cam_scale, scale, max_depth, raw, depth, a);
  hipLaunchKernelGGL(binned_dilate_kernel, tiles, blk, 0, st, h, w, depth, a, bb);                                          // s2
  hipLaunchKernelGGL(close5_kernel, tiles, blk, 0, st, h, w, bb, a);                                                        // s3
  hipLaunchKernelGGL(median_masked_kernel, tiles, blk, 0, st, h, w, a, (const int*)nullptr, bb);                            // s4
  hipLaunchKernelGGL(top_row_kernel, dim3(ceil_div(w, 64), b), dim3(64), 0, st, h, w, bb, top);
  hipLaunchKernelGGL(fill_empty9_kernel, rows4, blk, 0, st, h, w, bb, top, a);                                              // s5
  hipLaunchKernelGGL(top_row_kernel, dim3(ceil_div(w, 64), b), dim3(64), 0, st, h, w, a, top);                              // mask of s5
  hipLaunchKernelGGL(fill6_kernel, tiles, blk, 0, st, h, w, a, top, bb);                                                    // s7: six masked 5x5 fills
  // bb = s7 before the blurs (kept: its validity mask gates both blurs); median into a, bilateral + inversion into out
  hipLaunchKernelGGL(median_masked_kernel, tiles, blk, 0, st, h, w, bb, top, a);
  hipLaunchKernelGGL(bilateral_invert_kernel, tiles, blk, 0, st, h, w, 0.5f, 2.0f, max_depth, (float)scale, (float)cam_scale, bb,
                     a, top, out);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int istnet_depth_fill_scratch_floats(int b, int h, int w) { return b > 0 && h > 0 && w > 0 ? 3 * b * h * w + b * w : 0; }

int istnet_depth_fill_multiscale(int b, int h, int w, const float* depth, float max_depth, float* scratch, float* out,
                                 void* stream) {
  return fill_impl(b, h, w, depth, 1, 1.0, 1.0, max_depth, scratch, out, stream);
}

int istnet_depth_fill_missing(int b, int h, int w, const void* depth_raw, int raw_is_float, double cam_scale, double scale_2_80m,
                              float max_depth, float* scratch, float* out, void* stream) {
  if (raw_is_float < 0 || raw_is_float > 2) return ISTNET_PN2_EINVAL;
  return fill_impl(b, h, w, depth_raw, raw_is_float, cam_scale, scale_2_80m, max_depth, scratch, out, stream);
}

}  // extern "C"
