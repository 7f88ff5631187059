// sa_compact.hip -- compact-column form of a set-abstraction scale (gfx950).
//
// The reference pads every ball-query row to nsample slots by repeating the row's first hit
// (ball_query_gpu.cu:38-45), and QueryAndGroup / SharedMLP / max_pool2d then process the padded slots like any
// other (pointnet2_utils.py:348-358, pointnet2_modules.py:61-68).  With the radii of IST-Net (model/ist_net.py:16)
// most slots of the fine levels are such repeats: 83 % / 67 % of level 1 and 68 % / 36 % of level 2 on the benchmark
// clouds (tools/ball_padding_stats.py).  A repeated slot carries the same input as slot 0 of its group, hence the
// same activations in every layer of the per-point MLP; its only footprint is its MULTIPLICITY in the sums taken
// over points -- BatchNorm statistics forward and backward, weight gradients, the layer-0 gradient scatter.
//
// So a scale can be evaluated on COMPACT columns: per group its `cnt` distinct neighbours followed, when the row is
// padded, by ONE representative of the nsample - cnt repeats, which carries that multiplicity as a column weight
// (the representative keeps its own column because its gradient differs from slot 0's: the max-pool routes to the
// first maximum, never to a repeat).  Columns of all groups of all clouds are concatenated on one point axis
// of static capacity B * npoint * nsample; the number of valid columns T lives in device memory, so launches keep
// static grids (HIP-graph capturable) and workgroups past T leave at once.  Results equal the padded evaluation up
// to fp32 summation order.
//
// This file: the compaction (column tables from the ball-query indices) and the kernels of a scale that are not
// GEMMs (layer-0 gather-add, BN + ReLU + max over ragged groups, pooled gradient -> compact dense gradient, xyz
// weight gradient).  The GEMM kernels of csrc/pw_mlp.hip take the same (T, column weight) pair.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/istnet_pw.h"

namespace {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0x111>(v);
  v += dpp_f<0x112>(v);
  v += dpp_f<0x114>(v);
  v += dpp_f<0x118>(v);
  v += dpp_f<0x142, 0xa>(v);
  v += dpp_f<0x143, 0xc>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- compaction ------------------------------------------------------------------------------------------------
// glen[g] = columns of group g = cnt + (cnt < S), cnt = distinct leading entries of the row (a valid entry never
// equals the row's first hit again: valid entries ascend strictly, ball_query_gpu.cu:33-47).
__global__ __launch_bounds__(256) void compact_count_kernel(int NG, int S, const int* __restrict__ idx,
                                                            int* __restrict__ glen) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= NG) return;
  const int* row = idx + (size_t)g * S;
  const int first = row[0];
  int cnt = 1;
  for (int s = 1; s < S; ++s) cnt += row[s] != first ? 1 : 0;
  glen[g] = cnt + (cnt < S ? 1 : 0);
}
// gstart = exclusive scan of glen over all NG groups (one workgroup; NG <= 1024 * 64), gstart[NG] = T.
// Each thread owns a contiguous slice; slice sums are scanned inside the wave with shuffles and across the 16 waves
// through LDS: two barriers in all.
__global__ __launch_bounds__(1024) void compact_scan_kernel(int NG, const int* __restrict__ glen,
                                                            int* __restrict__ gstart) {
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = ((NG + 1023) / 1024 + 3) & ~3;          // multiple of 4: int4 loads
  const int lo = min(tid * per, NG), hi = min(lo + per, NG);
  int s = 0;
  int i = lo;
  for (; i + 3 < hi; i += 4) {
    const int4 v = *reinterpret_cast<const int4*>(glen + i);
    s += (v.x + v.y) + (v.z + v.w);
  }
  for (; i < hi; ++i) s += glen[i];
  int incl = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) base += w < wv ? wave_tot[w] : 0;
  int run = base + incl - s;
  for (i = lo; i < hi; ++i) { gstart[i] = run; run += glen[i]; }
  if (tid == 1023) gstart[NG] = base + incl;
}
// column tables: cidx[p] = global source point (cloud * n + point), meta[p] = group * 64 + position in the group,
// colw[p] = multiplicity; columns T .. roundup(T, 256) - 1 are null columns (weight 0, a valid address) so that tiles
// straddling T compute finite values
__global__ __launch_bounds__(256) void compact_fill_kernel(int NG, int G, int S, int n, const int* __restrict__ idx,
                                                           const int* __restrict__ gstart, int* __restrict__ cidx,
                                                           int* __restrict__ meta, float* __restrict__ colw) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)NG * S;
  if (e < 256) {                                  // tail (also covers T == 0)
    const int T = gstart[NG];
    const long long p = T + e;
    if (p < ((long long)T + 255) / 256 * 256 && p < total) { cidx[p] = 0; meta[p] = 0; colw[p] = 0.f; }
  }
  if (e >= total) return;
  const int g = (int)(e / S), s = (int)(e - (long long)g * S);
  const int* row = idx + (size_t)g * S;
  const int first = row[0], len = gstart[g + 1] - gstart[g];
  const int cnt = len < S ? len - 1 : (row[S - 1] != first || S == 1 ? S : S - 1);
  // cnt: len == cnt + 1 when padded; len == S means either cnt == S, or cnt == S - 1 with one repeat
  const int b = g / G;
  const int base = gstart[g];
  if (s < cnt) {
    cidx[base + s] = b * n + row[s];
    meta[base + s] = g * 64 + s;
    colw[base + s] = 1.f;
  } else if (s == cnt) {                          // the representative of the S - cnt repeats of slot 0
    cidx[base + s] = b * n + first;
    meta[base + s] = g * 64 + s;
    colw[base + s] = (float)(S - cnt);
  }
}

// ---- the two scales of an MSG level in one launch each (blockIdx.y = scale) -------------------------------------------
struct CompactOne {
  int S;
  const int* idx;
  int* glen;
  int* gstart;
  int* cidx;
  int* meta;
  float* colw;
};
struct CompactPair { CompactOne s[2]; };

__global__ __launch_bounds__(256) void compact_count_pair_kernel(int NG, CompactPair cp) {
  const CompactOne c = cp.s[blockIdx.y];
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= NG) return;
  const int* row = c.idx + (size_t)g * c.S;
  const int first = row[0];
  int cnt = 1;
  for (int s = 1; s < c.S; ++s) cnt += row[s] != first ? 1 : 0;
  c.glen[g] = cnt + (cnt < c.S ? 1 : 0);
}

// compact_scan_kernel for two tables; a thread's slice (<= 64 entries, int4 loads) stays in registers between the sum and
// the write-back (the stand-alone kernel re-reads glen entry by entry in a dependent loop: 18 us for 16 384 groups)
__global__ __launch_bounds__(1024) void compact_scan_pair_kernel(int NG, CompactPair cp) {
  const CompactOne c = cp.s[blockIdx.x];
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = ((NG + 1023) / 1024 + 3) & ~3;          // multiple of 4, <= 64 (NG <= 1024 * 64)
  const int lo = min(tid * per, NG), hi = min(lo + per, NG);
  int4 v[16];
  int s = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = lo + 4 * q;
    v[q] = make_int4(0, 0, 0, 0);
    if (4 * q < per) {
      if (i + 3 < hi) v[q] = *reinterpret_cast<const int4*>(c.glen + i);
      else {
        if (i < hi) v[q].x = c.glen[i];
        if (i + 1 < hi) v[q].y = c.glen[i + 1];
        if (i + 2 < hi) v[q].z = c.glen[i + 2];
      }
      s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
    }
  }
  int incl = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) base += w < wv ? wave_tot[w] : 0;
  int run = base + incl - s;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = lo + 4 * q;
    if (4 * q < per && i < hi) {
      int4 o;
      o.x = run; run += v[q].x;
      o.y = run; run += v[q].y;
      o.z = run; run += v[q].z;
      o.w = run; run += v[q].w;
      if (i + 3 < hi) *reinterpret_cast<int4*>(c.gstart + i) = o;
      else {
        c.gstart[i] = o.x;
        if (i + 1 < hi) c.gstart[i + 1] = o.y;
        if (i + 2 < hi) c.gstart[i + 2] = o.z;
      }
    }
  }
  if (tid == 1023) c.gstart[NG] = base + incl;
}

__global__ __launch_bounds__(256) void compact_fill_pair_kernel(int NG, int G, int n, CompactPair cp) {
  const CompactOne c = cp.s[blockIdx.y];
  const int S = c.S;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)NG * S;
  if (e < 256) {                                  // tail (also covers T == 0)
    const int T = c.gstart[NG];
    const long long p = T + e;
    if (p < ((long long)T + 255) / 256 * 256 && p < total) { c.cidx[p] = 0; c.meta[p] = 0; c.colw[p] = 0.f; }
  }
  if (e >= total) return;
  const int g = (int)(e / S), s = (int)(e - (long long)g * S);
  const int* row = c.idx + (size_t)g * S;
  const int first = row[0], len = c.gstart[g + 1] - c.gstart[g];
  const int cnt = len < S ? len - 1 : (row[S - 1] != first || S == 1 ? S : S - 1);
  const int b = g / G;
  const int base = c.gstart[g];
  if (s < cnt) {
    c.cidx[base + s] = b * n + row[s];
    c.meta[base + s] = g * 64 + s;
    c.colw[base + s] = 1.f;
  } else if (s == cnt) {                          // the representative of the S - cnt repeats of slot 0
    c.cidx[base + s] = b * n + first;
    c.meta[base + s] = g * 64 + s;
    c.colw[base + s] = (float)(S - cnt);
  }
}

// ---- layer 0: y0[c][p] = z[cloud][c][point] + W0x[c] . (xyz[source] - centre[group]), weighted statistics ------------
constexpr int kGatherAddCO = 32;
__global__ __launch_bounds__(256) void gather_add_cols_kernel(int n, int G, long long cap, int cout, int ldw,
                                                              const float* __restrict__ xyz,
                                                              const float* __restrict__ new_xyz,
                                                              const int* __restrict__ cidx,
                                                              const int* __restrict__ meta,
                                                              const float* __restrict__ colw,
                                                              const int* __restrict__ ncols,
                                                              const float* __restrict__ z,
                                                              const float* __restrict__ w0, float* __restrict__ y,
                                                              float* __restrict__ part_sum,
                                                              float* __restrict__ part_sq, int nt_total) {
  __shared__ float wx[kGatherAddCO * 3];
  __shared__ float red[4][kGatherAddCO][2];
  const int tid = threadIdx.x;
  const int c0 = blockIdx.y * kGatherAddCO;
  const int nco = min(kGatherAddCO, cout - c0);
  const bool stats = part_sum != nullptr;
  const long long p = (long long)blockIdx.x * 256 + tid;
  if ((long long)blockIdx.x * 256 >= *ncols) {      // tile past the valid columns
    if (stats && tid < nco) {
      part_sum[(size_t)(c0 + tid) * nt_total + blockIdx.x] = 0.f;
      part_sq[(size_t)(c0 + tid) * nt_total + blockIdx.x] = 0.f;
    }
    return;
  }
  if (tid < nco * 3) wx[tid] = w0[(size_t)(c0 + tid / 3) * ldw + (tid % 3)];
  const int src = cidx[p], g = meta[p] >> 6;
  const float wcol = colw[p];
  const float* xs = xyz + (size_t)src * 3;
  const float* xc = new_xyz + (size_t)g * 3;
  const float dx = xs[0] - xc[0], dy = xs[1] - xc[1], dz = xs[2] - xc[2];
  const int cloud = g / G;
  const float* zb = z != nullptr ? z + ((size_t)cloud * cout + c0) * n + (src - cloud * n) : nullptr;
  float* yb = y + (size_t)c0 * cap + p;
  __syncthreads();
  const int wv = tid >> 6;
  for (int co = 0; co < nco; co += 4) {
    float zv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) zv[j] = zb != nullptr ? zb[(size_t)min(co + j, nco - 1) * n] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (co + j < nco) {
        const int c = co + j;
        const float v = zv[j] + ((wx[3 * c] * dx + wx[3 * c + 1] * dy) + wx[3 * c + 2] * dz);
        yb[(size_t)c * cap] = v;
        if (stats) {
          const float s = wave_sum(wcol * v), q = wave_sum(wcol * v * v);
          if ((tid & 63) == 0) { red[wv][c][0] = s; red[wv][c][1] = q; }
        }
      }
    }
  }
  if (stats) {
    __syncthreads();
    if (tid < nco) {
      part_sum[(size_t)(c0 + tid) * nt_total + blockIdx.x] = (red[0][tid][0] + red[1][tid][0]) + (red[2][tid][0] + red[3][tid][0]);
      part_sq[(size_t)(c0 + tid) * nt_total + blockIdx.x] = (red[0][tid][1] + red[1][tid][1]) + (red[2][tid][1] + red[3][tid][1]);
    }
  }
}

// ---- tail: out[cloud][c][group] = max over the group's columns of relu(bn(y)); arg = position of the first maximum --
__global__ __launch_bounds__(256) void bn_relu_pool_cols_kernel(int C, int G, int NG, long long cap,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const int* __restrict__ gstart,
                                                                float* __restrict__ out, long long out_bstride,
                                                                uint8_t* __restrict__ arg, float* __restrict__ ymax) {
  const int c = blockIdx.y;
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= NG) return;
  const float s = scale[c], h = shift[c];
  const int a = gstart[g], z = gstart[g + 1];
  const float* row = y + (size_t)c * cap;
  float best = -1.f, raw = 0.f;
  int besti = 0;
  // Round 6: eight columns loaded before the first is compared (clamped, issued unconditionally).  With one load per trip of a
  // loop whose length is data (up to nsample + 1 columns) every column waited for its own memory round trip: 27 us for the
  // nsample-32 scale of level 1, at the end of its forward chain.  Same comparisons in the same order.
  for (int p0 = a; p0 < z; p0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = row[min(p0 + u, z - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (p0 + u < z) {
        const float act = fmaxf(v[u] * s + h, 0.f);
        if (act > best) { best = act; besti = p0 + u - a; raw = v[u]; }
      }
    }
  }
  const int cloud = g / G, j = g - cloud * G;
  out[(size_t)cloud * out_bstride + (size_t)c * G + j] = best;
  arg[((size_t)cloud * C + c) * G + j] = (uint8_t)besti;
  if (ymax != nullptr) ymax[((size_t)cloud * C + c) * G + j] = raw;
}

// ---- gradient through the max-pool as a dense compact tensor: dA[c][p] = dO[cloud][c][group] at the arg-max column ----
__global__ __launch_bounds__(256) void pooled_grad_cols_kernel(int C, int G, long long cap,
                                                               const float* __restrict__ pooled,
                                                               long long pooled_bstride,
                                                               const uint8_t* __restrict__ arg,
                                                               const int* __restrict__ meta,
                                                               const int* __restrict__ ncols,
                                                               float* __restrict__ out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if ((long long)blockIdx.x * 256 >= *ncols) return;
  const int m = meta[p], g = m >> 6, k = m & 63;
  const int cloud = g / G, j = g - cloud * G;
  const int c0 = blockIdx.y * 8;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = c0 + u;
    if (c < C) {
      const int a = arg[((size_t)cloud * C + c) * G + j];
      out[(size_t)c * cap + p] = a == k ? pooled[(size_t)cloud * pooled_bstride + (size_t)c * G + j] : 0.f;
    }
  }
}

// ---- xyz weight gradient of layer 0: dwx[chunk][c][k] = sum_p w_p dY0[c][p] (xyz[source_p] - centre_p)[k] ----------
constexpr int kDwxCH = 4;
__global__ __launch_bounds__(256) void dwx_cols_kernel(int cout, long long cap, const float* __restrict__ y,
                                                       const float* __restrict__ d, const float* __restrict__ bn,
                                                       const float* __restrict__ bwdc,
                                                       const int* __restrict__ cidx, const int* __restrict__ meta,
                                                       const float* __restrict__ colw,
                                                       const int* __restrict__ ncols, const float* __restrict__ xyz,
                                                       const float* __restrict__ new_xyz, float* __restrict__ dwx) {
  const int c0 = blockIdx.x * kDwxCH;
  const int nch = min(kDwxCH, cout - c0);
  const int T = *ncols;
  const int chunks = gridDim.y;
  const long long per = (((long long)T + chunks - 1) / chunks + 255) / 256 * 256;
  const long long beg = (long long)blockIdx.y * per, end = min(beg + per, (long long)T);
  float acc[kDwxCH][3];
#pragma unroll
  for (int ch = 0; ch < kDwxCH; ++ch) acc[ch][0] = acc[ch][1] = acc[ch][2] = 0.f;
  float rs[kDwxCH], rh[kDwxCH], ca[kDwxCH], cb[kDwxCH], cc[kDwxCH];
#pragma unroll
  for (int ch = 0; ch < kDwxCH; ++ch) {
    const int co = min(c0 + ch, cout - 1);
    rs[ch] = bn[co]; rh[ch] = bn[cout + co];
    ca[ch] = bwdc[co]; cb[ch] = bwdc[cout + co]; cc[ch] = bwdc[2 * cout + co];
  }
  for (long long p = beg + threadIdx.x; p < end; p += 256) {
    const int src = cidx[p], g = meta[p] >> 6;
    const float wcol = colw[p];
    const float* xs = xyz + (size_t)src * 3;
    const float* xc = new_xyz + (size_t)g * 3;
    const float x0 = (xs[0] - xc[0]) * wcol, x1 = (xs[1] - xc[1]) * wcol, x2 = (xs[2] - xc[2]) * wcol;
#pragma unroll
    for (int ch = 0; ch < kDwxCH; ++ch) {
      const size_t o = (size_t)min(c0 + ch, cout - 1) * cap + p;
      const float yv = y[o];
      const float v = ca[ch] * ((yv * rs[ch] + rh[ch] > 0.f) ? d[o] : 0.f) + cb[ch] + cc[ch] * yv;
      acc[ch][0] += v * x0; acc[ch][1] += v * x1; acc[ch][2] += v * x2;
    }
  }
  __shared__ float wred[4][kDwxCH * 3];
#pragma unroll
  for (int ch = 0; ch < kDwxCH; ++ch)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float t = wave_sum(acc[ch][k]);
      if (lane_id() == 0) wred[threadIdx.x >> 6][ch * 3 + k] = t;
    }
  __syncthreads();
  if (threadIdx.x < nch * 3)
    dwx[((size_t)blockIdx.y * cout + c0) * 3 + threadIdx.x] =
        (wred[0][threadIdx.x] + wred[1][threadIdx.x]) + (wred[2][threadIdx.x] + wred[3][threadIdx.x]);
}

// ---- layer-0 gradient scatter over inverse lists of the compact columns (csrc/pw_mlp.hip: pw_scatter_csr_kernel) ----
// G[cloud][c][i] = sum over the columns p of the cloud whose source point is i of w_p dY0[c][p]; a workgroup owns CH
// channels of one cloud, stages the weighted dY0 of the cloud's columns in LDS (coalesced), then every source point sums
// its list (ascending columns, four lanes per list combined in a fixed order): no atomics, deterministic.
// dwx[cloud][c][0:3] = sum_i xyz[i] G[c][i] - sum_p w_p dY0[c][p] centre(group of p).
template <int CH>
__global__ __launch_bounds__(256) void scatter_csr_cols_kernel(int cout, int n, int G, long long cap,
                                                               const float* __restrict__ y,
                                                               const float* __restrict__ d,
                                                               const float* __restrict__ bn,
                                                               const float* __restrict__ bwdc,
                                                               const int* __restrict__ gstart,
                                                               const int* __restrict__ off_all,
                                                               const int* __restrict__ ent_all,
                                                               const int* __restrict__ meta,
                                                               const float* __restrict__ colw,
                                                               float* __restrict__ out, long long out_bstride,
                                                               const float* __restrict__ xyz,
                                                               const float* __restrict__ new_xyz,
                                                               float* __restrict__ dwx) {
  extern __shared__ __attribute__((aligned(16))) float dy[];   // [columns of the cloud][CH]
  const int b = blockIdx.y, c0 = blockIdx.x * CH;
  const int nch = min(CH, cout - c0);
  const int base = gstart[b * G], Pb = gstart[(b + 1) * G] - base;
  float rs[CH], rh[CH], ca[CH], cb[CH], cc[CH];
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    const int co = min(c0 + ch, cout - 1);
    rs[ch] = bn[co]; rh[ch] = bn[cout + co];
    ca[ch] = bwdc[co]; cb[ch] = bwdc[cout + co]; cc[ch] = bwdc[2 * cout + co];
  }
  for (int p = threadIdx.x; p < Pb; p += 256) {
    const float wcol = colw[base + p];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      const size_t o = (size_t)min(c0 + ch, cout - 1) * cap + base + p;
      const float yv = y[o];
      dy[(size_t)p * CH + ch] = wcol * (ca[ch] * ((yv * rs[ch] + rh[ch] > 0.f) ? d[o] : 0.f) + cb[ch] + cc[ch] * yv);
    }
  }
  __syncthreads();
  const int* off = off_all + (size_t)b * (n + 1);
  const int* ent = ent_all + base;
  float wx[CH][3];
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) wx[ch][0] = wx[ch][1] = wx[ch][2] = 0.f;
  const int part = threadIdx.x & 3;
  const int n_round = (n + 63) / 64 * 64;
  for (int i = threadIdx.x >> 2; i < n_round; i += 64) {
    const bool valid = i < n;
    const int a0 = valid ? off[i] : 0, z0 = valid ? off[i + 1] : 0;
    const int len = z0 - a0, q = (len + 3) >> 2;
    const int a = min(a0 + part * q, z0), z = min(a + q, z0);
    float sum[CH], wc[CH][3];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) { sum[ch] = 0.f; wc[ch][0] = wc[ch][1] = wc[ch][2] = 0.f; }
    for (int u = a; u < z; ++u) {
      const int e = ent[u];
      float c3[3] = {0.f, 0.f, 0.f};
      if (dwx != nullptr) {
        const float* cp = new_xyz + (size_t)(meta[base + e] >> 6) * 3;
        c3[0] = cp[0]; c3[1] = cp[1]; c3[2] = cp[2];
      }
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        const float v = dy[(size_t)e * CH + ch];
        sum[ch] += v;
        wc[ch][0] += v * c3[0]; wc[ch][1] += v * c3[1]; wc[ch][2] += v * c3[2];
      }
    }
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      sum[ch] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sum[ch]), 0xB1, 0xf, 0xf, false));
      sum[ch] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sum[ch]), 0x4E, 0xf, 0xf, false));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        wc[ch][k] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(wc[ch][k]), 0xB1, 0xf, 0xf, false));
        wc[ch][k] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(wc[ch][k]), 0x4E, 0xf, 0xf, false));
      }
    }
    if (valid && part == 0) {
#pragma unroll
      for (int ch = 0; ch < CH; ++ch)
        if (ch < nch) out[(size_t)b * out_bstride + (size_t)(c0 + ch) * n + i] = sum[ch];
      if (dwx != nullptr) {
        const float* xs = xyz + ((size_t)b * n + i) * 3;
        const float px = xs[0], py = xs[1], pz = xs[2];
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
          wx[ch][0] += px * sum[ch] - wc[ch][0]; wx[ch][1] += py * sum[ch] - wc[ch][1]; wx[ch][2] += pz * sum[ch] - wc[ch][2];
        }
      }
    }
  }
  if (dwx != nullptr) {
    __syncthreads();
    float* wred = dy;     // [4 waves][CH*3]
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float t = wave_sum(wx[ch][k]);
        if (lane_id() == 0) wred[(threadIdx.x >> 6) * CH * 3 + ch * 3 + k] = t;
      }
    __syncthreads();
    if (threadIdx.x < nch * 3)
      dwx[((size_t)b * cout + c0) * 3 + threadIdx.x] = (wred[threadIdx.x] + wred[CH * 3 + threadIdx.x]) +
                                                       (wred[2 * CH * 3 + threadIdx.x] + wred[3 * CH * 3 + threadIdx.x]);
  }
}

}  // namespace

extern "C" {

int istnet_sa_compact(int b, int g, int s, int n, const int* idx, int* glen, int* gstart, int* cidx, int* meta,
                      float* colw, void* stream) {
  if (b <= 0 || g <= 0 || s <= 0 || s > 64 || n <= 0 || !idx || !glen || !gstart || !cidx || !meta || !colw)
    return ISTNET_PN2_EINVAL;
  const long long ng = (long long)b * g;
  if (ng > 1024 * 64 || ng * s >= (1LL << 31) || ng >= (1 << 25)) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(compact_count_kernel, dim3(ceil_div((int)ng, 256)), dim3(256), 0, as_stream(stream), (int)ng, s,
                     idx, glen);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, as_stream(stream), (int)ng, glen, gstart);
  hipLaunchKernelGGL(compact_fill_kernel, dim3((unsigned)((ng * s + 255) / 256)), dim3(256), 0, as_stream(stream),
                     (int)ng, g, s, n, idx, gstart, cidx, meta, colw);
  return (int)hipGetLastError();
}

// istnet_sa_compact for the two scales of one level (same b, g, n): 2 launches (3 when the counts are not supplied)
// instead of 6.  have_glen != 0: glen_a / glen_b already hold the column counts (istnet_pn2_query_ball_point_pair).
int istnet_sa_compact_pair(int b, int g, int n, int s_a, const int* idx_a, int* glen_a, int* gstart_a, int* cidx_a,
                           int* meta_a, float* colw_a, int s_b, const int* idx_b, int* glen_b, int* gstart_b,
                           int* cidx_b, int* meta_b, float* colw_b, int have_glen, void* stream) {
  if (b <= 0 || g <= 0 || n <= 0 || s_a <= 0 || s_a > 64 || s_b <= 0 || s_b > 64 || !idx_a || !glen_a || !gstart_a ||
      !cidx_a || !meta_a || !colw_a || !idx_b || !glen_b || !gstart_b || !cidx_b || !meta_b || !colw_b)
    return ISTNET_PN2_EINVAL;
  const long long ng = (long long)b * g;
  const int smax = s_a > s_b ? s_a : s_b;
  if (ng > 1024 * 64 || ng * smax >= (1LL << 31) || ng >= (1 << 25)) return ISTNET_PN2_EINVAL;
  CompactPair cp;
  cp.s[0] = CompactOne{s_a, idx_a, glen_a, gstart_a, cidx_a, meta_a, colw_a};
  cp.s[1] = CompactOne{s_b, idx_b, glen_b, gstart_b, cidx_b, meta_b, colw_b};
  if (!have_glen)
    hipLaunchKernelGGL(compact_count_pair_kernel, dim3(ceil_div((int)ng, 256), 2), dim3(256), 0, as_stream(stream), (int)ng, cp);
  hipLaunchKernelGGL(compact_scan_pair_kernel, dim3(2), dim3(1024), 0, as_stream(stream), (int)ng, cp);
  hipLaunchKernelGGL(compact_fill_pair_kernel, dim3((unsigned)((ng * smax + 255) / 256), 2), dim3(256), 0, as_stream(stream),
                     (int)ng, g, n, cp);
  return (int)hipGetLastError();
}

int istnet_pw_gather_add_cols(int b, int n, int g, long long cap, int cout, const float* xyz, const float* new_xyz,
                              const int* cidx, const int* meta, const float* colw, const int* ncols, const float* z,
                              const float* w0, int ldw, float* y, float* part_sum, float* part_sq, void* stream) {
  if (b <= 0 || n <= 0 || g <= 0 || cap <= 0 || (cap & 255) || cout <= 0 || ldw < 3 || !cidx || !meta || !colw || !ncols)
    return ISTNET_PN2_EINVAL;
  const dim3 grid((unsigned)(cap / 256), ceil_div(cout, kGatherAddCO));
  hipLaunchKernelGGL(gather_add_cols_kernel, grid, dim3(256), 0, as_stream(stream), n, g, cap, cout, ldw, xyz, new_xyz,
                     cidx, meta, colw, ncols, z, w0, y, part_sum, part_sq, (int)(cap / 256));
  return (int)hipGetLastError();
}

int istnet_bn_relu_pool_cols(int b, int c, int g, long long cap, const float* y, const float* bn, const int* gstart,
                             float* out, long long out_bstride, unsigned char* arg, float* ymax, void* stream) {
  if (b <= 0 || c <= 0 || g <= 0 || cap <= 0 || !y || !bn || !gstart || !out || !arg) return ISTNET_PN2_EINVAL;
  if (out_bstride <= 0) out_bstride = (long long)c * g;
  const int ng = b * g;
  hipLaunchKernelGGL(bn_relu_pool_cols_kernel, dim3(ceil_div(ng, 256), c), dim3(256), 0, as_stream(stream), c, g, ng,
                     cap, y, bn, bn + c, gstart, out, out_bstride, arg, ymax);
  return (int)hipGetLastError();
}

int istnet_pw_pooled_grad_cols(int b, int c, int g, long long cap, const float* d_pooled, long long pooled_bstride,
                               const unsigned char* arg, const int* meta, const int* ncols, float* out, void* stream) {
  if (b <= 0 || c <= 0 || g <= 0 || cap <= 0 || (cap & 255) || !d_pooled || !arg || !meta || !ncols || !out)
    return ISTNET_PN2_EINVAL;
  if (pooled_bstride <= 0) pooled_bstride = (long long)c * g;
  hipLaunchKernelGGL(pooled_grad_cols_kernel, dim3((unsigned)(cap / 256), ceil_div(c, 8)), dim3(256), 0,
                     as_stream(stream), c, g, cap, d_pooled, pooled_bstride, arg, meta, ncols, out);
  return (int)hipGetLastError();
}

int istnet_pw_dwx_cols_chunks(int cout) {
  const int wgs = ceil_div(cout, kDwxCH);
  return ceil_div(1024, wgs > 0 ? wgs : 1);
}

int istnet_pw_dwx_cols(int cout, long long cap, const float* y, const float* d_dense, const float* bn,
                       const float* bwdc, const int* cidx, const int* meta, const float* colw, const int* ncols,
                       const float* xyz, const float* new_xyz, float* dwx, void* stream) {
  if (cout <= 0 || cap <= 0 || !y || !d_dense || !bn || !bwdc || !cidx || !meta || !colw || !ncols || !xyz ||
      !new_xyz || !dwx)
    return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(dwx_cols_kernel, dim3(ceil_div(cout, kDwxCH), istnet_pw_dwx_cols_chunks(cout)), dim3(256), 0,
                     as_stream(stream), cout, cap, y, d_dense, bn, bwdc, cidx, meta, colw, ncols, xyz, new_xyz, dwx);
  return (int)hipGetLastError();
}

int istnet_pw_scatter_dy_csr_cols(int b, int cout, int n, int g, long long cap, const float* y, const float* d_dense,
                                  const float* bn, const float* bwdc, const int* gstart, const int* offsets,
                                  const int* entries, const int* meta, const float* colw, float* out,
                                  long long out_bstride, const float* xyz, const float* new_xyz, float* dwx,
                                  void* stream) {
  if (b <= 0 || cout <= 0 || n <= 0 || g <= 0 || cap <= 0 || (cap % b) || !y || !d_dense || !bn || !bwdc || !gstart ||
      !offsets || !entries || !meta || !colw || !out)
    return ISTNET_PN2_EINVAL;
  if (dwx != nullptr && (xyz == nullptr || new_xyz == nullptr)) return ISTNET_PN2_EINVAL;
  const long long pc = cap / b;                         // most columns one cloud can hold
  int ch = 16;
  while (ch > 1 && ((size_t)ch * pc * 4 > 64 * 1024 || (long long)b * ceil_div(cout, ch) < 512)) ch >>= 1;
  if ((size_t)ch * pc * 4 > 64 * 1024) return ISTNET_PN2_EINVAL;
  const size_t lds = (size_t)ch * pc * 4 < 4 * 16 * 3 * 4 ? 4 * 16 * 3 * 4 : (size_t)ch * pc * 4;
  const dim3 grid(ceil_div(cout, ch), b);
  const long long obs = out_bstride > 0 ? out_bstride : (long long)cout * n;
#define ISTNET_SCSRC(CH)                                                                                          \
  hipLaunchKernelGGL(scatter_csr_cols_kernel<CH>, grid, dim3(256), lds, as_stream(stream), cout, n, g, cap, y,     \
                     d_dense, bn, bwdc, gstart, offsets, entries, meta, colw, out, obs, xyz, new_xyz, dwx)
  switch (ch) {
    case 16: ISTNET_SCSRC(16); break;
    case 8: ISTNET_SCSRC(8); break;
    case 4: ISTNET_SCSRC(4); break;
    case 2: ISTNET_SCSRC(2); break;
    default: ISTNET_SCSRC(1); break;
  }
#undef ISTNET_SCSRC
  return (int)hipGetLastError();
}

}  // extern "C"
