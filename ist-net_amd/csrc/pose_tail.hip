// pose_tail.hip -- the tail of IST-Net's pose estimators and the pose / NOCS losses as a handful of launches (gfx950).
//
// Reference: model/ist_net.py:250-264,318-332 (mean-pooled 512-d feature -> three heads Linear 512-512-256-k with ReLU ->
// Ortho6d2Mat), utils/rotation_utils.py:4-28 (6-D rotation -> SO(3) by Gram-Schmidt), model/losses.py:3-49 (PoseDis,
// SmoothL1Dis).  With B = 32 every tensor here is a few KB; as framework ops the three estimators of a training step are
// ~1 000 launches of 2-5 us kernels in dependent chains (forward + autograd backward of ~45 ops per Ortho6d2Mat, ~10 per
// PoseDis, 9 Linear + 6 ReLU per estimator) -- a quarter of the point branch's wall time.  Here:
//   fc_fwd_kernel          one layer of ALL heads of an estimator: Y[h] = act(X[h] W[h]^T + b[h]), M = B <= 64 rows
//   fc_bwd_dx_kernel       dX[h] = (dY[h] . [Y[h] > 0]) W[h]           (summed over the heads that share X when asked)
//   fc_bwd_dw_kernel       dW[h] = (dY . mask)^T X,  db[h] = column sums of dY . mask
//   ortho6d_fwd / _bwd     thread per sample, closed-form backward of the two normalisations and the two cross products
//   pose_dis_fwd / _bwd    scalar loss and its gradient in one launch each
//   smooth_l1_fwd / _bwd   the NOCS-coordinate loss on (B, N, 3)
// VALU only: these are M = 32 products whose weights stream once; MFMA tiles would be 3 % full.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/istnet_heads.h"

namespace {

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

constexpr int kMaxHeads = 4;
struct FcBatch {
  const float* x[kMaxHeads];     // (B, K)
  const float* w[kMaxHeads];     // (N_h, K) row-major (torch Linear.weight)
  const float* bias[kMaxHeads];  // (N_h)
  float* y[kMaxHeads];           // (B, N_h)
  int n[kMaxHeads];
  int tile_begin[kMaxHeads + 1]; // prefix sum of ceil(N_h / NJ)
  int nheads;
};

// ---- forward: workgroup = 8 output neurons of one head (4 waves x 2 neurons; lane = batch row + 32 * neuron-in-wave for
// B <= 32, one neuron per wave for B <= 64).  X is staged per 128-column chunk TRANSPOSED in LDS ([k / 4][b] float4: the
// lanes of a wave read consecutive 16-byte words), the 8 weight rows beside it (broadcast reads).
template <int BPAD>
__global__ __launch_bounds__(256) void fc_fwd_kernel(FcBatch fb, int B, int K, int relu) {
  constexpr int KC = 128, NPW = 64 / BPAD, NJ = 4 * NPW;      // neurons per wave / per workgroup
  __shared__ __attribute__((aligned(16))) float xt[KC / 4][BPAD][4];
  __shared__ __attribute__((aligned(16))) float wl[NJ][KC];
  int h = 0;
  while (h + 1 < fb.nheads && (int)blockIdx.x >= fb.tile_begin[h + 1]) ++h;
  const int j0 = ((int)blockIdx.x - fb.tile_begin[h]) * NJ;
  const int N = fb.n[h];
  const float* __restrict__ x = fb.x[h];
  const float* __restrict__ w = fb.w[h];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = lane % BPAD, jw = wv * NPW + lane / BPAD;     // this thread's batch row and neuron (within the tile)
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += KC) {
    __syncthreads();
    for (int e = tid; e < (KC / 4) * BPAD; e += 256) {         // consecutive threads: consecutive rows of one float4 column
      const int bb = e % BPAD, kq = e / BPAD;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bb < B && k0 + 4 * kq < K) v = *reinterpret_cast<const float4*>(x + (size_t)bb * K + k0 + 4 * kq);
      *reinterpret_cast<float4*>(&xt[kq][bb][0]) = v;
    }
    for (int e = tid; e < NJ * (KC / 4); e += 256) {
      const int jj = e / (KC / 4), kq = e % (KC / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j0 + jj < N && k0 + 4 * kq < K) v = *reinterpret_cast<const float4*>(w + (size_t)(j0 + jj) * K + k0 + 4 * kq);
      *reinterpret_cast<float4*>(&wl[jj][4 * kq]) = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int kq = 0; kq < KC / 4; ++kq) {
      const float4 xv = *reinterpret_cast<const float4*>(&xt[kq][b][0]);
      const float4 wv4 = *reinterpret_cast<const float4*>(&wl[jw][4 * kq]);
      acc = __builtin_fmaf(xv.x, wv4.x, acc);
      acc = __builtin_fmaf(xv.y, wv4.y, acc);
      acc = __builtin_fmaf(xv.z, wv4.z, acc);
      acc = __builtin_fmaf(xv.w, wv4.w, acc);
    }
  }
  const int j = j0 + jw;
  if (b < B && j < N) {
    float v = acc + fb.bias[h][j];
    if (relu) v = fmaxf(v, 0.f);
    fb.y[h][(size_t)b * N + j] = v;
  }
}

// ---- backward, input gradient: dX[b][k] (+)= sum_h sum_j dZ_h[b][j] W_h[j][k],  dZ = dY . [Y > 0] (relu) or dY.
// Lane = input column (coalesced weight rows); dZ is staged transposed in LDS and read as broadcasts.
struct FcBwdBatch {
  const float* dy[kMaxHeads];   // (B, N_h)
  const float* y[kMaxHeads];    // (B, N_h) post-activation output of the layer (relu mask), or null
  const float* w[kMaxHeads];    // (N_h, K)
  const float* x[kMaxHeads];    // (B, K) layer input (dW)
  float* dx[kMaxHeads];         // (B, K); heads with the same pointer are summed in one pass (shared input)
  float* dw[kMaxHeads];         // (N_h, K)
  float* db[kMaxHeads];         // (N_h)
  int n[kMaxHeads];
  int tile_begin[kMaxHeads + 1];
  int nheads;
  int shared_x;                 // 1: all heads read the same X and their input gradients are summed into dx[0]
};

// Launch shape: the product is 1.5-3 MB of weights against a 32-row operand, i.e. pure latency -- a lane's weight loads are
// a dependent-looking chain the compiler does not pipeline.  So the chain is cut three ways: a workgroup owns 64 columns and
// RB = 8 batch rows (grid = column tiles x row groups [x heads]); its SIXTEEN waves each take 1/16 of every 512-neuron chunk
// with all 8 rows in registers, eight independent weight loads in flight per lane; the 16 partial sums meet in LDS and are
// added in wave order (deterministic).  512 -> 512 x 3 heads: 96 loads per lane instead of 1 536.
constexpr int kFcDxRows = 8, kFcDxWaves = 16, kFcDxChunk = 512;
__global__ __launch_bounds__(64 * kFcDxWaves) void fc_bwd_dx_kernel(FcBwdBatch fb, int B, int K, int relu) {
  constexpr int RB = kFcDxRows, NW = kFcDxWaves, JC = kFcDxChunk, JW = JC / NW;
  __shared__ __attribute__((aligned(16))) float dzt[JC][RB];        // [neuron][row]: a wave reads two broadcast float4
  __shared__ float red[NW][RB][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ktiles = ceil_div(K, 64), rgroups = ceil_div(B, RB);
  const int kt = (int)blockIdx.x % ktiles, rg = ((int)blockIdx.x / ktiles) % rgroups;
  const int hsel = fb.shared_x ? -1 : (int)blockIdx.x / (ktiles * rgroups);      // -1: loop over all heads
  const int k = kt * 64 + lane, kc = min(k, K - 1), b0 = rg * RB;
  float acc[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) acc[r] = 0.f;
  const int h_lo = hsel < 0 ? 0 : hsel, h_hi = hsel < 0 ? fb.nheads : hsel + 1;
  for (int h = h_lo; h < h_hi; ++h) {
    const int N = fb.n[h];
    const float* __restrict__ w = fb.w[h] + kc;
    for (int j0 = 0; j0 < N; j0 += JC) {
      __syncthreads();
      for (int e = tid; e < JC * RB; e += 64 * NW) {
        const int jj = e % JC, r = e / JC;           // consecutive threads: consecutive neurons (coalesced dY rows)
        float v = 0.f;
        if (b0 + r < B && j0 + jj < N) {
          v = fb.dy[h][(size_t)(b0 + r) * N + j0 + jj];
          if (relu && !(fb.y[h][(size_t)(b0 + r) * N + j0 + jj] > 0.f)) v = 0.f;
        }
        dzt[jj][r] = v;
      }
      __syncthreads();
      const int ja = wv * JW, jn = min(JW, N - j0 - ja);            // this wave's neurons of the chunk
      for (int jj = 0; jj < jn; jj += 8) {
        float wr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wr[u] = w[(size_t)(j0 + ja + min(jj + u, jn - 1)) * K];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float wu = jj + u < jn ? wr[u] : 0.f;
          const float4 d0 = *reinterpret_cast<const float4*>(&dzt[ja + min(jj + u, jn - 1)][0]);
          const float4 d1 = *reinterpret_cast<const float4*>(&dzt[ja + min(jj + u, jn - 1)][4]);
          acc[0] = __builtin_fmaf(d0.x, wu, acc[0]); acc[1] = __builtin_fmaf(d0.y, wu, acc[1]);
          acc[2] = __builtin_fmaf(d0.z, wu, acc[2]); acc[3] = __builtin_fmaf(d0.w, wu, acc[3]);
          acc[4] = __builtin_fmaf(d1.x, wu, acc[4]); acc[5] = __builtin_fmaf(d1.y, wu, acc[5]);
          acc[6] = __builtin_fmaf(d1.z, wu, acc[6]); acc[7] = __builtin_fmaf(d1.w, wu, acc[7]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) red[wv][r][lane] = acc[r];
  __syncthreads();
  if (tid < RB * 64) {
    const int r = tid / 64, c = tid % 64;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) sum += red[q][r][c];
    const int kk = kt * 64 + c;
    if (kk < K && b0 + r < B) fb.dx[hsel < 0 ? 0 : hsel][(size_t)(b0 + r) * K + kk] = sum;
  }
}

// ---- backward, weight / bias gradients: workgroup = 8 neurons of one head x 256 input columns; thread = column, loop
// over the batch rows (X read coalesced from L2, the eight dZ values of a row as LDS broadcasts).
__global__ __launch_bounds__(256) void fc_bwd_dw_kernel(FcBwdBatch fb, int B, int K, int relu) {
  constexpr int NJ = 8;
  __shared__ float dz[64][NJ];                       // [batch row][neuron]
  const int tid = threadIdx.x;
  const int ktiles = ceil_div(K, 256);
  const int tile = (int)blockIdx.x / ktiles, kt = (int)blockIdx.x % ktiles;
  int h = 0;
  while (h + 1 < fb.nheads && tile >= fb.tile_begin[h + 1]) ++h;
  const int j0 = (tile - fb.tile_begin[h]) * NJ;
  const int N = fb.n[h];
  for (int e = tid; e < 64 * NJ; e += 256) {
    const int bb = e / NJ, jj = e % NJ;
    float v = 0.f;
    if (bb < B && j0 + jj < N) {
      v = fb.dy[h][(size_t)bb * N + j0 + jj];
      if (relu && !(fb.y[h][(size_t)bb * N + j0 + jj] > 0.f)) v = 0.f;
    }
    dz[bb][jj] = v;
  }
  __syncthreads();
  const int k = kt * 256 + tid;
  float acc[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) acc[jj] = 0.f;
  if (k < K) {
    const float* __restrict__ x = fb.x[h];
    for (int bb = 0; bb < B; ++bb) {
      const float xv = x[(size_t)bb * K + k];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) acc[jj] = __builtin_fmaf(dz[bb][jj], xv, acc[jj]);
    }
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
      if (j0 + jj < N) fb.dw[h][(size_t)(j0 + jj) * K + k] = acc[jj];
  }
  if (kt == 0 && tid < NJ && j0 + tid < N && fb.db[h] != nullptr) {
    float s = 0.f;
    for (int bb = 0; bb < B; ++bb) s += dz[bb][tid];
    fb.db[h][j0 + tid] = s;
  }
}

// ---- Ortho6d2Mat (utils/rotation_utils.py:4-28): y = norm(y_raw), z = norm(x_raw x y), x = y x z, R = [x y z] columns;
// norms clamped at 1e-8 (clamp passes the gradient where mag >= 1e-8).  r6 (B, 6) = [x_raw | y_raw] -> R (B, 3, 3).
__device__ __forceinline__ void cross3(const float* u, const float* v, float* o) {
  o[0] = u[1] * v[2] - u[2] * v[1];
  o[1] = u[2] * v[0] - u[0] * v[2];
  o[2] = u[0] * v[1] - u[1] * v[0];
}
__device__ __forceinline__ float normalize3(const float* v, float* o) {   // returns the unclamped magnitude
  const float mag = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  const float m = fmaxf(mag, 1e-8f);
  o[0] = v[0] / m; o[1] = v[1] / m; o[2] = v[2] / m;
  return mag;
}
// gradient through o = v / max(|v|, 1e-8) given g = dL/do
__device__ __forceinline__ void normalize3_bwd(const float* o, float mag, const float* g, float* gv) {
  if (mag >= 1e-8f) {
    const float d = (o[0] * g[0] + o[1] * g[1]) + o[2] * g[2];
    gv[0] = (g[0] - o[0] * d) / mag; gv[1] = (g[1] - o[1] * d) / mag; gv[2] = (g[2] - o[2] * d) / mag;
  } else {
    gv[0] = g[0] / 1e-8f; gv[1] = g[1] / 1e-8f; gv[2] = g[2] / 1e-8f;
  }
}
__global__ void ortho6d_fwd_kernel(int B, const float* __restrict__ r6, float* __restrict__ R) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* xr = r6 + (size_t)b * 6;
  float y[3], c[3], z[3], x[3];
  normalize3(xr + 3, y);
  cross3(xr, y, c);
  normalize3(c, z);
  cross3(y, z, x);
  float* o = R + (size_t)b * 9;
#pragma unroll
  for (int i = 0; i < 3; ++i) { o[3 * i + 0] = x[i]; o[3 * i + 1] = y[i]; o[3 * i + 2] = z[i]; }
}
__global__ void ortho6d_bwd_kernel(int B, const float* __restrict__ r6, const float* __restrict__ dR,
                                   float* __restrict__ dr6) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* xr = r6 + (size_t)b * 6;
  float y[3], c[3], z[3];
  const float ymag = normalize3(xr + 3, y);
  cross3(xr, y, c);
  const float cmag = normalize3(c, z);
  const float* g = dR + (size_t)b * 9;
  float gx[3], gy[3], gz[3], t[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { gx[i] = g[3 * i + 0]; gy[i] = g[3 * i + 1]; gz[i] = g[3 * i + 2]; }
  // x = y x z:  dL/dy += z x gx,  dL/dz += gx x y
  cross3(z, gx, t); gy[0] += t[0]; gy[1] += t[1]; gy[2] += t[2];
  cross3(gx, y, t); gz[0] += t[0]; gz[1] += t[1]; gz[2] += t[2];
  float gc[3];
  normalize3_bwd(z, cmag, gz, gc);
  // c = x_raw x y:  dL/dx_raw = y x gc,  dL/dy += gc x x_raw
  float gxr[3];
  cross3(y, gc, gxr);
  cross3(gc, xr, t); gy[0] += t[0]; gy[1] += t[1]; gy[2] += t[2];
  float gyr[3];
  normalize3_bwd(y, ymag, gy, gyr);
  float* o = dr6 + (size_t)b * 6;
  o[0] = gxr[0]; o[1] = gxr[1]; o[2] = gxr[2]; o[3] = gyr[0]; o[4] = gyr[1]; o[5] = gyr[2];
}

// ---- PoseDis (model/losses.py:37-49): mean over (B, 3) of the norm over dim 1 of R1 - R2, + mean_b |t1 - t2| + mean_b
// |s1 - s2|.  One workgroup; terms[b] keeps what the backward needs: 3 column norms, |dt|, |ds|.
__global__ __launch_bounds__(256) void pose_dis_fwd_kernel(int B, const float* __restrict__ r1, const float* __restrict__ t1,
                                                           const float* __restrict__ s1, const float* __restrict__ r2,
                                                           const float* __restrict__ t2, const float* __restrict__ s2,
                                                           float* __restrict__ loss, float* __restrict__ norms) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    float n5[5];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) { const float d = r1[b * 9 + 3 * i + j] - r2[b * 9 + 3 * i + j]; q += d * d; }
      n5[j] = sqrtf(q);
    }
    float qt = 0.f, qs = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float dt = t1[b * 3 + i] - t2[b * 3 + i], ds = s1[b * 3 + i] - s2[b * 3 + i];
      qt += dt * dt; qs += ds * ds;
    }
    n5[3] = sqrtf(qt); n5[4] = sqrtf(qs);
#pragma unroll
    for (int j = 0; j < 5; ++j) norms[b * 5 + j] = n5[j];
    acc += ((double)n5[0] + (double)n5[1] + (double)n5[2]) / (3.0 * B) + ((double)n5[3] + (double)n5[4]) / (double)B;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = (float)red[0];
}
__global__ void pose_dis_bwd_kernel(int B, const float* __restrict__ gout, const float* __restrict__ r1,
                                    const float* __restrict__ t1, const float* __restrict__ s1,
                                    const float* __restrict__ r2, const float* __restrict__ t2, const float* __restrict__ s2,
                                    const float* __restrict__ norms, float* __restrict__ dr, float* __restrict__ dt,
                                    float* __restrict__ ds) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float g = *gout;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float n = norms[b * 5 + j];
    const float sc = n > 0.f ? g / (3.f * B) / n : 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) dr[b * 9 + 3 * i + j] = (r1[b * 9 + 3 * i + j] - r2[b * 9 + 3 * i + j]) * sc;
  }
  const float nt = norms[b * 5 + 3], ns = norms[b * 5 + 4];
  const float st = nt > 0.f ? g / B / nt : 0.f, ss = ns > 0.f ? g / B / ns : 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dt[b * 3 + i] = (t1[b * 3 + i] - t2[b * 3 + i]) * st;
    ds[b * 3 + i] = (s1[b * 3 + i] - s2[b * 3 + i]) * ss;
  }
}

// ---- SmoothL1Dis (model/losses.py:3-22) on (rows, 3): mean over rows of the sum over xyz of
// |d| > thr ? |d| - thr / 2 : d^2 / (2 thr).  Partials per workgroup, fixed-order final sum by the last launch stage.
__global__ __launch_bounds__(256) void smooth_l1_fwd_kernel(long long n3, float thr, const float* __restrict__ p1,
                                                            const float* __restrict__ p2, float* __restrict__ part) {
  __shared__ float red[256];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n3; i += (long long)gridDim.x * 256) {
    const float d = fabsf(p1[i] - p2[i]);
    acc += d > thr ? d - thr * 0.5f : d * d / (2.f * thr);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void smooth_l1_finish_kernel(int nparts, double rows, const float* __restrict__ part,
                                                               float* __restrict__ loss) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += (double)part[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = (float)(red[0] / rows);
}
// mean squared error of two dense tensors (the feature-alignment term of SupervisedLoss, model/ist_net.py:99) with its
// gradient in the same pass: part[block] = sum (a - b)^2, da = (2 / n) (a - b) [* gscale]; b == nullptr: b = 0
__global__ __launch_bounds__(256) void mse_value_grad_kernel(long long n4, long long n, float two_over_n,
                                                             const float* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ da, float* __restrict__ part) {
  __shared__ float red[256];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 x = reinterpret_cast<const float4*>(a)[i];
    float4 d = x;
    if (b != nullptr) {
      const float4 y = reinterpret_cast<const float4*>(b)[i];
      d.x -= y.x; d.y -= y.y; d.z -= y.z; d.w -= y.w;
    }
    acc += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    reinterpret_cast<float4*>(da)[i] = make_float4(d.x * two_over_n, d.y * two_over_n, d.z * two_over_n, d.w * two_over_n);
  }
  if (blockIdx.x == 0)
    for (long long i = 4 * n4 + threadIdx.x; i < n; i += 256) {           // tail of a length that is not a multiple of 4
      const float d = a[i] - (b != nullptr ? b[i] : 0.f);
      acc += d * d;
      da[i] = d * two_over_n;
    }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void smooth_l1_bwd_kernel(long long n3, float thr, float inv_rows,
                                                            const float* __restrict__ gout, const float* __restrict__ p1,
                                                            const float* __restrict__ p2, float* __restrict__ dp1) {
  const float g = *gout * inv_rows;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n3; i += (long long)gridDim.x * 256) {
    const float d = p1[i] - p2[i];
    const float a = fabsf(d);
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    dp1[i] = g * (a > thr ? sgn : d / thr);
  }
}

int fill_tiles(int nheads, const int* n, int nj, int* tile_begin) {
  tile_begin[0] = 0;
  for (int h = 0; h < nheads; ++h) {
    if (n[h] <= 0) return -1;
    tile_begin[h + 1] = tile_begin[h] + ceil_div(n[h], nj);
  }
  return tile_begin[nheads];
}

}  // namespace

extern "C" {

int istnet_fc_forward(int nheads, int b, int k, const int* n, const float* const* x, const float* const* w,
                      const float* const* bias, float* const* y, int relu, void* stream) {
  if (nheads <= 0 || nheads > kMaxHeads || b <= 0 || b > 64 || k <= 0 || (k & 3)) return ISTNET_PN2_EINVAL;
  FcBatch fb;
  fb.nheads = nheads;
  const int nj = b <= 32 ? 8 : 4;
  for (int h = 0; h < nheads; ++h) {
    if (!x[h] || !w[h] || !bias[h] || !y[h]) return ISTNET_PN2_EINVAL;
    fb.x[h] = x[h]; fb.w[h] = w[h]; fb.bias[h] = bias[h]; fb.y[h] = y[h]; fb.n[h] = n[h];
  }
  const int tiles = fill_tiles(nheads, n, nj, fb.tile_begin);
  if (tiles <= 0) return ISTNET_PN2_EINVAL;
  if (b <= 32) hipLaunchKernelGGL(fc_fwd_kernel<32>, dim3(tiles), dim3(256), 0, as_stream(stream), fb, b, k, relu);
  else hipLaunchKernelGGL(fc_fwd_kernel<64>, dim3(tiles), dim3(256), 0, as_stream(stream), fb, b, k, relu);
  return (int)hipGetLastError();
}

int istnet_fc_backward(int nheads, int b, int k, const int* n, const float* const* dy, const float* const* y,
                       const float* const* w, const float* const* x, float* const* dx, float* const* dw,
                       float* const* db, int relu, int shared_x, void* stream) {
  if (nheads <= 0 || nheads > kMaxHeads || b <= 0 || b > 64 || k <= 0) return ISTNET_PN2_EINVAL;
  FcBwdBatch fb;
  fb.nheads = nheads;
  fb.shared_x = shared_x ? 1 : 0;
  bool want_dx = false, want_dw = false;
  for (int h = 0; h < nheads; ++h) {
    if (!dy[h] || !w[h] || (relu && !y[h])) return ISTNET_PN2_EINVAL;
    fb.dy[h] = dy[h]; fb.y[h] = y ? y[h] : nullptr; fb.w[h] = w[h]; fb.x[h] = x ? x[h] : nullptr;
    fb.dx[h] = dx ? dx[h] : nullptr; fb.dw[h] = dw ? dw[h] : nullptr; fb.db[h] = db ? db[h] : nullptr; fb.n[h] = n[h];
    want_dx = want_dx || fb.dx[h] != nullptr;
    want_dw = want_dw || fb.dw[h] != nullptr;
  }
  if (want_dx) {
    for (int h = 0; h < (shared_x ? 1 : nheads); ++h)
      if (fb.dx[h] == nullptr) return ISTNET_PN2_EINVAL;
    const int grid = (shared_x ? 1 : nheads) * ceil_div(k, 64) * ceil_div(b, kFcDxRows);
    hipLaunchKernelGGL(fc_bwd_dx_kernel, dim3(grid), dim3(64 * kFcDxWaves), 0, as_stream(stream), fb, b, k, relu);
  }
  if (want_dw) {
    for (int h = 0; h < nheads; ++h)
      if (fb.dw[h] == nullptr || fb.x[h] == nullptr) return ISTNET_PN2_EINVAL;
    const int tiles = fill_tiles(nheads, n, 8, fb.tile_begin);
    if (tiles <= 0) return ISTNET_PN2_EINVAL;
    hipLaunchKernelGGL(fc_bwd_dw_kernel, dim3(tiles * ceil_div(k, 256)), dim3(256), 0, as_stream(stream), fb, b, k, relu);
  }
  return (int)hipGetLastError();
}

int istnet_ortho6d_forward(int b, const float* r6, float* r, void* stream) {
  if (b <= 0 || !r6 || !r) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(ortho6d_fwd_kernel, dim3(ceil_div(b, 64)), dim3(64), 0, as_stream(stream), b, r6, r);
  return (int)hipGetLastError();
}
int istnet_ortho6d_backward(int b, const float* r6, const float* d_r, float* d_r6, void* stream) {
  if (b <= 0 || !r6 || !d_r || !d_r6) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(ortho6d_bwd_kernel, dim3(ceil_div(b, 64)), dim3(64), 0, as_stream(stream), b, r6, d_r, d_r6);
  return (int)hipGetLastError();
}

int istnet_pose_dis_forward(int b, const float* r1, const float* t1, const float* s1, const float* r2, const float* t2,
                            const float* s2, float* loss, float* norms, void* stream) {
  if (b <= 0 || !r1 || !t1 || !s1 || !r2 || !t2 || !s2 || !loss || !norms) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(pose_dis_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), b, r1, t1, s1, r2, t2, s2, loss, norms);
  return (int)hipGetLastError();
}
int istnet_pose_dis_backward(int b, const float* gout, const float* r1, const float* t1, const float* s1, const float* r2,
                             const float* t2, const float* s2, const float* norms, float* dr, float* dt, float* ds,
                             void* stream) {
  if (b <= 0 || !gout || !r1 || !t1 || !s1 || !r2 || !t2 || !s2 || !norms || !dr || !dt || !ds) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(pose_dis_bwd_kernel, dim3(ceil_div(b, 64)), dim3(64), 0, as_stream(stream), b, gout, r1, t1, s1, r2,
                     t2, s2, norms, dr, dt, ds);
  return (int)hipGetLastError();
}

int istnet_smooth_l1_parts(long long rows) {
  const long long blocks = (rows * 3 + 4095) / 4096;
  return (int)(blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks));
}
int istnet_smooth_l1_forward(long long rows, float threshold, const float* p1, const float* p2, float* part, float* loss,
                             void* stream) {
  if (rows <= 0 || threshold <= 0.f || !p1 || !p2 || !part || !loss) return ISTNET_PN2_EINVAL;
  const int parts = istnet_smooth_l1_parts(rows);
  hipLaunchKernelGGL(smooth_l1_fwd_kernel, dim3(parts), dim3(256), 0, as_stream(stream), rows * 3, threshold, p1, p2, part);
  hipLaunchKernelGGL(smooth_l1_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), parts, (double)rows, part, loss);
  return (int)hipGetLastError();
}
int istnet_smooth_l1_backward(long long rows, float threshold, const float* gout, const float* p1, const float* p2,
                              float* dp1, void* stream) {
  if (rows <= 0 || threshold <= 0.f || !gout || !p1 || !p2 || !dp1) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(smooth_l1_bwd_kernel, dim3(istnet_smooth_l1_parts(rows)), dim3(256), 0, as_stream(stream), rows * 3,
                     threshold, (float)(1.0 / (double)rows), gout, p1, p2, dp1);
  return (int)hipGetLastError();
}

int istnet_mse_parts(long long n) {
  const long long blocks = (n / 4 + 256 * 4 - 1) / (256 * 4);           // >= 4 float4 per thread
  return (int)(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks));
}
int istnet_mse_value_grad(long long n, const float* a, const float* b, float* da, float* part, float* loss, void* stream) {
  if (n <= 0 || !a || !da || !part || !loss) return ISTNET_PN2_EINVAL;
  if (((uintptr_t)a | (uintptr_t)da | (uintptr_t)b) & 15) return ISTNET_PN2_EINVAL;
  const int parts = istnet_mse_parts(n);
  hipLaunchKernelGGL(mse_value_grad_kernel, dim3(parts), dim3(256), 0, as_stream(stream), n / 4, n, (float)(2.0 / (double)n), a,
                     b, da, part);
  hipLaunchKernelGGL(smooth_l1_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), parts, (double)n, part, loss);
  return (int)hipGetLastError();
}

}  // extern "C"
