// depth_fill.hip -- `fill_missing` of the reference's data pipeline on the GPU (gfx950).
//
// Reference: utils/data_utils.py:516-540 (fill_missing) -> :357-510 (fill_in_multiscale, the 'multiscale' / 'bilateral'
// branch every Dataset class uses, provider/dataset.py:172-173,361-362): depth completion by morphology -- inversion,
// three binned cross-kernel dilations, a 5x5 closing, 5x5 median blurs, masked hole fills with full 9x9 / 5x5 dilations,
// a bilateral filter -- as ~25 cv2 / numpy passes over a 480 x 640 float image on the host, per image.
// Here the same passes are stencil kernels over a BATCH of images (thread per pixel; the images are 1.2 MB each and stay
// in L2 between passes).  Border rules are OpenCV's documented ones: dilate / erode ignore pixels outside the image
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

// value of `img` dilated (max) / eroded (min) at (r, c) with a full k x k or a cross (row + column) structuring element
template <bool ERODE>
__device__ __forceinline__ float morph_at(const float* __restrict__ img, int h, int w, int r, int c, int k, bool cross) {
  const int rad = k / 2;
  float best = ERODE ? __builtin_inff() : -__builtin_inff();
  for (int dr = -rad; dr <= rad; ++dr) {
    const int rr = r + dr;
    if (rr < 0 || rr >= h) continue;
    for (int dc = -rad; dc <= rad; ++dc) {
      if (cross && dr != 0 && dc != 0) continue;
      const int cc = c + dc;
      if (cc < 0 || cc >= w) continue;
      const float v = img[(size_t)rr * w + cc];
      best = ERODE ? fminf(best, v) : fmaxf(best, v);
    }
  }
  return best;
}

// s1: inversion of the valid depths; the three depth bins are masks of the INPUT depth
__global__ void invert_kernel(long long n, float max_depth, const float* __restrict__ depth, float* __restrict__ s1) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float d = depth[i];
  s1[i] = d > kValid ? max_depth - d : d;
}

// s2: cross-kernel dilations of the far / medium / near bins (3 / 5 / 7), combined farthest to nearest
__global__ void binned_dilate_kernel(int h, int w, float max_depth, const float* __restrict__ depth_all,
                                     const float* __restrict__ s1_all, float* __restrict__ s2_all) {
  const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  if (c >= w) return;
  const float* depth = depth_all + (size_t)b * h * w;
  const float* s1 = s1_all + (size_t)b * h * w;
  // dilation of (s1 * bin mask): a neighbour outside its bin contributes 0
  float far = -__builtin_inff(), med = -__builtin_inff(), near = -__builtin_inff();
  for (int dr = -3; dr <= 3; ++dr) {
    const int rr = r + dr;
    if (rr < 0 || rr >= h) continue;
    for (int dc = -3; dc <= 3; ++dc) {
      if (dr != 0 && dc != 0) continue;
      const int cc = c + dc;
      if (cc < 0 || cc >= w) continue;
      const float d = depth[(size_t)rr * w + cc], v = s1[(size_t)rr * w + cc];
      const int dist = dr != 0 ? (dr < 0 ? -dr : dr) : (dc < 0 ? -dc : dc);
      if (dist <= 1) far = fmaxf(far, d > 2.0f ? v : 0.f);
      if (dist <= 2) med = fmaxf(med, (d > 1.0f && d <= 2.0f) ? v : 0.f);
      near = fmaxf(near, (d > kValid && d <= 1.0f) ? v : 0.f);
    }
  }
  float out = s1[(size_t)r * w + c];
  if (far > kValid) out = far;
  if (med > kValid) out = med;
  if (near > kValid) out = near;
  s2_all[((size_t)b * h + r) * w + c] = out;
}

// plain dilate / erode with a full k x k element
template <bool ERODE>
__global__ void morph_full_kernel(int h, int w, int k, const float* __restrict__ src, float* __restrict__ dst) {
  const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  if (c >= w) return;
  dst[((size_t)b * h + r) * w + c] = morph_at<ERODE>(src + (size_t)b * h * w, h, w, r, c, k, false);
}

// median of the 5 x 5 window, border replicated
__device__ __forceinline__ float median25(const float* __restrict__ img, int h, int w, int r, int c) {
  float v[25];
#pragma unroll
  for (int dr = -2; dr <= 2; ++dr)
#pragma unroll
    for (int dc = -2; dc <= 2; ++dc) {
      const int rr = min(max(r + dr, 0), h - 1), cc = min(max(c + dc, 0), w - 1);
      v[(dr + 2) * 5 + dc + 2] = img[(size_t)rr * w + cc];
    }
  // partial selection: after 13 passes v[12] is the 13th smallest
#pragma unroll
  for (int i = 0; i < 13; ++i) {
#pragma unroll
    for (int j = i + 1; j < 25; ++j) {
      const float a = v[i], bb = v[j];
      v[i] = fminf(a, bb);
      v[j] = fmaxf(a, bb);
    }
  }
  return v[12];
}
// dst = valid(src) ? median(src) : src, valid = src > 0.01 (and inside the top mask when given)
__global__ void median_masked_kernel(int h, int w, const float* __restrict__ src_all, const int* __restrict__ top_row,
                                     float* __restrict__ dst_all) {
  const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  if (c >= w) return;
  const float* src = src_all + (size_t)b * h * w;
  const float v = src[(size_t)r * w + c];
  const bool valid = v > kValid && (top_row == nullptr || r >= top_row[b * w + c]);
  dst_all[((size_t)b * h + r) * w + c] = valid ? median25(src, h, w, r, c) : v;
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

// dst = (empty pixel inside the top mask) ? dilate_full_k(src) : src;  empty: src <= 0.01 (strict = 0) or src < 0.01 (strict = 1)
__global__ void fill_empty_kernel(int h, int w, int k, int strict, const float* __restrict__ src_all,
                                  const int* __restrict__ top_row, float* __restrict__ dst_all) {
  const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  if (c >= w) return;
  const float* src = src_all + (size_t)b * h * w;
  const float v = src[(size_t)r * w + c];
  const bool empty = (strict ? v < kValid : !(v > kValid)) && r >= top_row[b * w + c];
  dst_all[((size_t)b * h + r) * w + c] = empty ? morph_at<false>(src, h, w, r, c, k, false) : v;
}

// bilateral filter d = 5 (disc of radius 2), BORDER_REFLECT_101; applied where valid, then the final inversion
__global__ void bilateral_invert_kernel(int h, int w, float sigma_color, float sigma_space, float max_depth,
                                        const float* __restrict__ pre_all, const float* __restrict__ src_all,
                                        const int* __restrict__ top_row, float* __restrict__ dst_all) {
  const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  if (c >= w) return;
  const float* src = src_all + (size_t)b * h * w;
  const float v = src[(size_t)r * w + c];
  // `valid` of the reference is the mask computed BEFORE the second median blur (data_utils.py:477-478 reused at :487)
  const float pre = pre_all[((size_t)b * h + r) * w + c];
  const bool valid = pre > kValid && r >= top_row[b * w + c];
  float out = v;
  if (valid) {
    const float gc = -0.5f / (sigma_color * sigma_color), gs = -0.5f / (sigma_space * sigma_space);
    float sum = 0.f, wsum = 0.f;
    for (int dr = -2; dr <= 2; ++dr)
      for (int dc = -2; dc <= 2; ++dc) {
        if (dr * dr + dc * dc > 4) continue;
        int rr = r + dr, cc = c + dc;
        rr = rr < 0 ? -rr : (rr >= h ? 2 * h - 2 - rr : rr);
        cc = cc < 0 ? -cc : (cc >= w ? 2 * w - 2 - cc : cc);
        const float u = src[(size_t)rr * w + cc];
        const float wgt = expf((float)(dr * dr + dc * dc) * gs) * expf((u - v) * (u - v) * gc);
        sum += u * wgt;
        wsum += wgt;
      }
    out = sum / wsum;
  }
  dst_all[((size_t)b * h + r) * w + c] = out > kValid ? max_depth - out : out;
}

}  // namespace

extern "C" {

int istnet_depth_fill_scratch_floats(int b, int h, int w) { return b > 0 && h > 0 && w > 0 ? 3 * b * h * w + b * w : 0; }

int istnet_depth_fill_multiscale(int b, int h, int w, const float* depth, float max_depth, float* scratch, float* out,
                                 void* stream) {
  if (b <= 0 || h < 5 || w < 5 || !depth || !scratch || !out || max_depth <= 0.f) return ISTNET_PN2_EINVAL;
  hipStream_t st = as_stream(stream);
  const long long n = (long long)b * h * w;
  float* a = scratch;
  float* bb = scratch + n;
  float* cc = scratch + 2 * n;
  int* top = reinterpret_cast<int*>(scratch + 3 * n);
  const dim3 grid(ceil_div(w, 64), h, b), blk(64);
  hipLaunchKernelGGL(invert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, max_depth, depth, a);           // s1
  hipLaunchKernelGGL(binned_dilate_kernel, grid, blk, 0, st, h, w, max_depth, depth, a, bb);                                // s2
  hipLaunchKernelGGL(morph_full_kernel<false>, grid, blk, 0, st, h, w, 5, bb, a);                                           // close: dilate
  hipLaunchKernelGGL(morph_full_kernel<true>, grid, blk, 0, st, h, w, 5, a, bb);                                            //        erode -> s3
  hipLaunchKernelGGL(median_masked_kernel, grid, blk, 0, st, h, w, bb, (const int*)nullptr, a);                             // s4
  hipLaunchKernelGGL(top_row_kernel, dim3(ceil_div(w, 64), b), blk, 0, st, h, w, a, top);
  hipLaunchKernelGGL(fill_empty_kernel, grid, blk, 0, st, h, w, 9, 0, a, top, bb);                                          // s5
  hipLaunchKernelGGL(top_row_kernel, dim3(ceil_div(w, 64), b), blk, 0, st, h, w, bb, top);                                  // mask of s5
  float* cur = bb;
  float* nxt = a;
  for (int i = 0; i < 6; ++i) {                                                                                             // s7: six masked 5x5 fills
    hipLaunchKernelGGL(fill_empty_kernel, grid, blk, 0, st, h, w, 5, 1, cur, top, nxt);
    float* t = cur; cur = nxt; nxt = t;
  }
  // cur = s7 before the blurs (kept: its validity mask gates both blurs); median into nxt, bilateral + inversion into out
  hipLaunchKernelGGL(median_masked_kernel, grid, blk, 0, st, h, w, cur, top, nxt);
  hipLaunchKernelGGL(bilateral_invert_kernel, grid, blk, 0, st, h, w, 0.5f, 2.0f, max_depth, cur, nxt, top, out);
  (void)cc;
  return (int)hipGetLastError();
}

}  // extern "C"
