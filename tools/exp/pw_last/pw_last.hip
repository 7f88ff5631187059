// pw_last.hip -- the LAST layer of a set-abstraction scale without its activation (gfx950, fp32 MFMA).
//
// Reference: every scale of a set-abstraction level ends in  conv1x1 -> BatchNorm2d -> ReLU -> max_pool2d over the
// nsample slots of a ball (model/pointnet2/pointnet2_modules.py:61-71, pytorch_utils.py:25-50).  The raw output y_L of
// that last convolution is the widest tensor of the stack (B x C_out x npoint x nsample: 34-67 MB per scale at B = 32);
// written by the GEMM, read by the max-pool, read again by the backward pass it was 1.2 GB of the 6.3 GB an encoder step
// moves.  It does not have to exist:
//
//  forward   the max-pool commutes with the monotone map y -> relu(scale y + shift), so pw_fwd2_kernel<..., POOL>
//            (pw_mlp.hip) keeps, per (channel, ball), the raw extremum y* and its slot; bn_finalize_pool_apply_kernel
//            (here) finishes the batch statistics and applies relu(scale y* + shift) to the (B, C, G) values.
//  backward  with g' the max-pool gradient (one non-zero per (channel, ball)) BatchNorm's backward is
//            dY = ca g' + cb + cc y,  y = W a,  a = act(y_{L-1})   -- dense only through y.  Hence
//              dA = W^T dY   = (W^T diag(cc) W) a + W^T cb + W^T (ca g')     = M a + c0 + S
//              dW = dY a^T   = diag(cc) W (a a^T) + cb (sum a)^T + (ca g') a^T
//            M (C_in x C_in) and c0 are tiny; M a is a GEMM with K = C_in instead of C_out (half the flops of the
//            dgrad it replaces); S and (ca g') a^T have ONE term per (channel, ball) and are evaluated as such --
//            S by the loader waves of pw_bwd_last_kernel with register accumulators indexed by the slot
//            (s_set_gpr_idx), the weight-gradient terms by pw_dw_last_kernel off the critical chain.
//            Bytes per point: C_in read + C_in written, against 2 C_out + 2 C_in before.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/istnet_pw.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

constexpr int kLastThreads = 512;   // waves 0..3 issue MFMAs, waves 4..7 load / do the per-ball work

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

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
__device__ __forceinline__ void split_point(long long q, int P, int& b, int& p) {
  const unsigned uq = (unsigned)q, up = (unsigned)P;
  const unsigned ub = uq / up;
  b = (int)ub;
  p = (int)(uq - ub * up);
}
// barrier that orders LDS traffic only (no vmcnt drain: the loaders' prefetch stays in flight); see pw_mlp.hip
__device__ __forceinline__ void lds_barrier() {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}
// MFMA C/D layout of v_mfma_f32_32x32x2_f32: reg r of lane l holds row (r&3)+8*(r>>2)+4*(l>>5), col l&31.
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ============================================================================================
// forward tail: BatchNorm statistics from the GEMM's partials, then out = relu(scale y* + shift) on the (B, C, G)
// extrema the pooled epilogue left -- the finalize and the (former) max-pool launch in one.  One workgroup per channel.
// ============================================================================================
constexpr int kFinThreads = 256;
__global__ __launch_bounds__(kFinThreads) void bn_finalize_pool_apply_kernel(
    int C, int B, int G, int nt, double count, const float* __restrict__ part_sum, const float* __restrict__ part_sq,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ bn,
    const float* __restrict__ gval, float* __restrict__ out, long long out_bstride) {
  const int c = blockIdx.x;
  const float* pa = part_sum + (size_t)c * nt;
  const float* pb = part_sq + (size_t)c * nt;
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  int i = threadIdx.x;
  for (; i + kFinThreads < nt; i += 2 * kFinThreads) {
    a0 += (double)pa[i]; a1 += (double)pa[i + kFinThreads];
    b0 += (double)pb[i]; b1 += (double)pb[i + kFinThreads];
  }
  for (; i < nt; i += kFinThreads) { a0 += (double)pa[i]; b0 += (double)pb[i]; }
  double a = a0 + a1, q = b0 + b1;
  for (int off = 32; off >= 1; off >>= 1) {
    a += __shfl_xor(a, off);
    q += __shfl_xor(q, off);
  }
  __shared__ double sh[2][kFinThreads / 64];
  __shared__ float s_aff[2];
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    const double sq = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    const double mean = s / count;
    double var = sq / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * istd;
    const float shf = beta[c] - (float)mean * sc;
    bn[0 * C + c] = sc;
    bn[1 * C + c] = shf;
    bn[2 * C + c] = (float)mean;
    bn[3 * C + c] = istd;
    if (running_mean != nullptr) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    s_aff[0] = sc;
    s_aff[1] = shf;
  }
  __syncthreads();
  const float sc = s_aff[0], shf = s_aff[1];
  for (int e = threadIdx.x; e < B * G; e += kFinThreads) {
    const int b = e / G, g = e - b * G;
    const float v = gval[((size_t)b * C + c) * G + g];
    out[(size_t)b * out_bstride + (size_t)c * G + g] = fmaxf(v * sc + shf, 0.f);
  }
}

// eval-mode / fixed-affine variant of the tail: constants given, only the apply
__global__ __launch_bounds__(256) void pool_apply_kernel(int C, int G, int rows, const float* __restrict__ bn,
                                                         const float* __restrict__ gval, float* __restrict__ out,
                                                         long long out_bstride) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= G) return;
  for (int bc = blockIdx.y; bc < rows; bc += gridDim.y) {
    const int b = bc / C, c = bc - b * C;
    out[(size_t)b * out_bstride + (size_t)c * G + g] = fmaxf(gval[(size_t)bc * G + g] * bn[c] + bn[C + c], 0.f);
  }
}

// ============================================================================================
// backward, step 1 (after the pooled BatchNorm-backward finalize produced ca / cb / cc):
//   role A  e[b][c][g] = ca_c dO[b][c][g] [scale_c y*[b][c][g] + shift_c > 0]   (natural layout, for pw_dw_last_kernel)
//           eT[b][g][c], slotT[b][g][c]: the same and the arg slots with the CHANNEL index contiguous -- the per-ball
//           lists pw_bwd_last_kernel walks with scalar loads
//   role B  M[ci][k] = sum_c cc_c W[c][ci] W[c][k]  (float64 accumulation),  c0[ci] = sum_c cb_c W[c][ci]
// ============================================================================================
__global__ __launch_bounds__(256) void pw_last_prep_kernel(
    int B, int COUT, int CIN, int G, int nA, const float* __restrict__ w, const float* __restrict__ bn,
    const float* __restrict__ bwdc, const float* __restrict__ pooled, long long pooled_bstride,
    const float* __restrict__ gval, const uint8_t* __restrict__ arg, float* __restrict__ e_nat,
    float* __restrict__ eT, uint8_t* __restrict__ slotT, float* __restrict__ M, float* __restrict__ c0) {
  __shared__ float tile[64][65];
  __shared__ uint8_t stile[64][68];
  __shared__ float wa[256][9], wb[256][9];       // role B: W[:, 8 ci] and W[:, 8 k] of up to 256 channels
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < nA) {
    const int gt = (G + 63) / 64, ct = COUT / 64;
    int t = blockIdx.x;
    const int cti = t % ct; t /= ct;
    const int gti = t % gt;
    const int b = t / gt;
    const int c_base = cti * 64, g_base = gti * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int el = tid + 256 * i, cl = el >> 6, gl = el & 63;     // consecutive threads: consecutive balls
      const int c = c_base + cl, g = g_base + gl;
      float ev = 0.f;
      uint8_t sl = 0;
      if (g < G) {
        const size_t o = ((size_t)b * COUT + c) * G + g;
        const float yv = gval[o];
        const float d = pooled[(size_t)b * pooled_bstride + (size_t)c * G + g];
        ev = (yv * bn[c] + bn[COUT + c] > 0.f) ? bwdc[c] * d : 0.f;
        sl = arg[o];
        e_nat[o] = ev;
      }
      tile[cl][gl] = ev;
      stile[cl][gl] = sl;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int el = tid + 256 * i, gl = el >> 6, cl = el & 63;     // consecutive threads: consecutive channels
      const int g = g_base + gl;
      if (g < G) {
        const size_t o = ((size_t)b * G + g) * COUT + c_base + cl;
        eT[o] = tile[cl][gl];
        slotT[o] = stile[cl][gl];
      }
    }
    return;
  }
  // ---- role B: an 8 x 8 tile of M (and 8 entries of c0 when the k tile is the first); the four waves of the workgroup
  // take quarters of the output channels and meet in LDS ----
  const int nk = CIN / 8;
  const int t = blockIdx.x - nA;
  const int ti = t / nk, tk = t % nk;
  float (*wa8)[9] = reinterpret_cast<float (*)[9]>(&wa[0][0]);     // [COUT][8 + 1]
  float (*wb8)[9] = reinterpret_cast<float (*)[9]>(&wb[0][0]);
  for (int el = tid; el < COUT * 8; el += 256) {
    const int c = el >> 3, j = el & 7;
    wa8[c][j] = w[(size_t)c * CIN + ti * 8 + j];
    wb8[c][j] = w[(size_t)c * CIN + tk * 8 + j];
  }
  __syncthreads();
  const int part = tid >> 6, i = (tid >> 3) & 7, k = tid & 7;
  const int cq = COUT / 4;
  double acc = 0.0, a0 = 0.0;
  for (int c = part * cq; c < (part + 1) * cq; ++c) {
    const double wi = (double)wa8[c][i];
    acc += (double)bwdc[2 * COUT + c] * wi * (double)wb8[c][k];
    if (k == 0) a0 += (double)bwdc[COUT + c] * wi;
  }
  double* dred = reinterpret_cast<double*>(&tile[0][0]);            // [4][64] M partials, [4][8] c0 partials
  dred[part * 64 + (tid & 63)] = acc;
  if (k == 0) dred[256 + part * 8 + i] = a0;
  __syncthreads();
  if (tid < 64) {
    const double v = (dred[tid] + dred[64 + tid]) + (dred[128 + tid] + dred[192 + tid]);
    M[(size_t)(ti * 8 + (tid >> 3)) * CIN + tk * 8 + (tid & 7)] = (float)v;
  }
  if (tk == 0 && tid < 8)
    c0[ti * 8 + tid] = (float)((dred[256 + tid] + dred[264 + tid]) + (dred[272 + tid] + dred[280 + tid]));
}

// ============================================================================================
// backward, step 2 (the critical chain):  dA_{L-1} = M a + c0 + S  with its BatchNorm-backward statistics partials.
// Structure of pw_bwd_mid_kernel (pw_mlp.hip): eight waves in two roles, one workgroup per CU, a chunk of PT points at
// a time through a double-buffered LDS, one LDS-only barrier per chunk.
//   loaders (waves 4..7)  x chunk -> Xs (raw: the statistics need it) and As = act(x) in the tensors' own layout; then
//       the sparse term: a work item is (ball, 64 input channels, a range of output channels); lane = input channel ci;
//       acc[slot] += e_c W[c][ci] over the item's channels with the slot a wave-uniform register index, the (e, slot)
//       lists read with scalar loads, W rows from L2; the S slots of the ball are then written to the Ss tile.  Two
//       partial Ss tiles when a chunk has fewer than four (ball, channel-half) items.
//   compute (waves 0..3)  one 32 x 32 tile of dA^T each: accumulators start from c0 + Ss, CIN / 2 MFMAs with the M
//       fragments held in registers, then dA is stored and the statistics of layer L-1 (sum g, sum g y_{L-1},
//       g = dA [relu active]) are accumulated from the raw tile.
// ============================================================================================
template <int CIT, int S>
struct LastCfg {
  static constexpr int CIN = 32 * CIT;
  static constexpr int PT = 128 / CIT;                 // points per chunk: 4 dA^T tiles, one per compute wave
  static constexpr int LD = PT + 4;
  static constexpr int F4 = PT / 4;
  static constexpr int NX = CIN * F4 / 256;            // float4 per loader thread (= 4)
  static constexpr int NG = PT / S;                    // balls per chunk
  static constexpr int NCIH = CIT == 4 ? 2 : 1;        // halves of the input channels (a lane holds ONE input channel)
  static constexpr int UNITS = NG * NCIH;
  static constexpr int NCS = UNITS >= 4 ? 1 : 4 / UNITS;   // output-channel splits so that every loader wave has an item
  static constexpr int ITEMS = UNITS * NCS;            // 4 or 8
  static constexpr int TILE = (2 + NCS) * CIN * LD;    // floats per buffer: Xs, As, Ss[NCS]
  static constexpr size_t LDS_BYTES = (2 * TILE + 3 * CIN) * sizeof(float);
  static_assert(NX == 4 && NG >= 1 && ITEMS % 4 == 0 && (NCS == 1 || NCS == 2), "unsupported shape");
};

// acc[slot] += e * w with the slot a wave-uniform REGISTER index (s_set_gpr_idx): the compiler's form toggles the
// VGPR-index mode four times per entry.  MEASURED (profiles/r03_last_layer_activation_free.txt): ~150 cycles per entry,
// which makes the loader waves 3-5x slower than the MFMA waves and the kernel slower than the stored-activation backward
// it was meant to replace; a hand-written block that keeps the mode on and swaps the index with s_set_gpr_idx_idx (one
// indexed v_fma_f32 per entry, accumulators pinned to v[64:95]) was slower still.  VGPR indexing is not a fast path on
// gfx950; a per-ball list sorted by slot (plain FMAs, one LDS store per slot) is the untried alternative.
template <int S> struct SlotAcc;
template <> struct SlotAcc<16> { typedef f32x16 type; };
template <> struct SlotAcc<32> { typedef f32x32 type; };

template <int CIT, int S>
__global__ __launch_bounds__(kLastThreads) void pw_bwd_last_kernel(
    int P, int COUT, long long total, int split_len, const float* __restrict__ w, const float* __restrict__ x,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, const float* __restrict__ M,
    const float* __restrict__ c0, const float* __restrict__ eT, const uint8_t* __restrict__ slotT,
    float* __restrict__ dx, float* __restrict__ part_g, float* __restrict__ part_gy, int nt_total) {
  using C = LastCfg<CIT, S>;
  typedef typename SlotAcc<S>::type acc_t;
  constexpr int CIN = C::CIN, PT = C::PT, LD = C::LD, F4 = C::F4, NX = C::NX, NCS = C::NCS;
  extern __shared__ __attribute__((aligned(16))) float last_lds[];
  float* const s_in = last_lds + 2 * C::TILE;     // [2][CIN] scale / shift of layer L-1's BatchNorm, [CIN] c0
  const int lane = lane_id();
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = wv >= 4;
  const int cw = wv & 3;
  const int tid = threadIdx.x & 255;
  const int l31 = lane & 31, half = lane >> 5;
  const long long qbeg = (long long)blockIdx.x * split_len;
  const long long qend = max(qbeg, min(qbeg + (long long)split_len, total));
  const int nchunks = (int)((qend - qbeg) / PT);
  const int G = P / S;
  for (int c = threadIdx.x; c < CIN; c += kLastThreads) {
    s_in[c] = in_scale[c]; s_in[CIN + c] = in_shift[c]; s_in[2 * CIN + c] = c0[c];
  }
  __syncthreads();

  if (loader) {
    struct Raw { float4 x[NX]; };
    Raw raw0, raw1;
    auto load_chunk = [&](Raw& rw, long long qk) {
      int b, pk;
      split_point(qk, P, b, pk);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int e = tid + 256 * i, row = e / F4, p = pk + (e % F4) * 4;
        rw.x[i] = *reinterpret_cast<const float4*>(x + ((size_t)b * CIN + row) * P + p);
      }
    };
    auto store_chunk = [&](const Raw& rw, float* buf) {
      float* Xs = buf;
      float* As = buf + CIN * LD;
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int e = tid + 256 * i, row = e / F4, k = (e % F4) * 4;
        const float sc = s_in[row], sh = s_in[CIN + row];
        const float4 v = rw.x[i];
        *reinterpret_cast<float4*>(&Xs[row * LD + k]) = v;
        float4 a;
        a.x = fmaxf(v.x * sc + sh, 0.f); a.y = fmaxf(v.y * sc + sh, 0.f);
        a.z = fmaxf(v.z * sc + sh, 0.f); a.w = fmaxf(v.w * sc + sh, 0.f);
        *reinterpret_cast<float4*>(&As[row * LD + k]) = a;
      }
    };
    // the sparse term of the chunk starting at flattened point qk, into the Ss tile(s) of `buf`
    auto sparse_chunk = [&](float* buf, long long qk) {
      int b, pk;
      split_point(qk, P, b, pk);
      const int g0 = pk / S;
#pragma unroll
      for (int it0 = 0; it0 < C::ITEMS; it0 += 4) {
        const int it = it0 + cw;
        const int cs = it % NCS, u = it / NCS;
        const int gidx = u / C::NCIH, cih = u % C::NCIH;
        const int ci = cih * 64 + lane;
        const bool act = ci < CIN;                         // CIN = 32: the upper half of the wave idles
        const int nch = COUT / NCS, c_lo = cs * nch;
        const size_t lo = ((size_t)b * G + g0 + gidx) * COUT + c_lo;      // wave-uniform
        const float* er = eT + lo;
        const unsigned* sr = reinterpret_cast<const unsigned*>(slotT + lo);
        const float* wr = w + (size_t)c_lo * CIN + (act ? ci : 0);
        // the item's (e, slot) lists as vector registers -- lane l holds e[64 q + l] and the slot word l (four slots) --
        // read back with v_readlane: no scalar-memory wait inside the loop
        float evq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) evq[q] = (64 * q + lane < nch) ? er[64 * q + lane] : 0.f;
        const unsigned sw = (4 * lane < nch) ? sr[lane] : 0u;
        acc_t acc;
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = 0.f;
        float wn[16];                                  // W rows of the NEXT 16 channels (L2 latency >> 16 entries' work)
#pragma unroll
        for (int j = 0; j < 16; ++j) wn[j] = wr[(size_t)j * CIN];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (64 * q < nch) {
            for (int cc = 0; cc < 64; cc += 16) {
              const int c = 64 * q + cc;
              float wc[16];
#pragma unroll
              for (int j = 0; j < 16; ++j) wc[j] = wn[j];
              if (c + 16 < nch) {
#pragma unroll
                for (int j = 0; j < 16; ++j) wn[j] = wr[(size_t)(c + 16 + j) * CIN];
              }
#pragma unroll
              for (int j = 0; j < 16; ++j) {
                const int ebits = __builtin_amdgcn_readlane(__float_as_int(evq[q]), cc + j);
                const unsigned word = (unsigned)__builtin_amdgcn_readlane((int)sw, 16 * q + (cc >> 2) + (j >> 2));
                const int sl = (int)((word >> (8 * (j & 3))) & 0xffu);
                acc[sl] = __builtin_fmaf(__int_as_float(ebits), wc[j], acc[sl]);
              }
            }
          }
        }
        if (act) {
          float* Ss = buf + (2 + cs) * CIN * LD + ci * LD + gidx * S;
#pragma unroll
          for (int s = 0; s < S; s += 4)
            *reinterpret_cast<float4*>(Ss + s) = make_float4(acc[s], acc[s + 1], acc[s + 2], acc[s + 3]);
        }
      }
    };

    if (nchunks > 0) load_chunk(raw0, qbeg);
    if (nchunks > 1) load_chunk(raw1, qbeg + PT);
    for (int t = 0; t < nchunks; t += 2) {
      const long long qk = qbeg + (long long)t * PT;
      store_chunk(raw0, last_lds);
      if (t + 2 < nchunks) load_chunk(raw0, qk + 2 * PT);
      sparse_chunk(last_lds, qk);
      lds_barrier();
      if (t + 1 < nchunks) {
        store_chunk(raw1, last_lds + C::TILE);
        if (t + 3 < nchunks) load_chunk(raw1, qk + 3 * PT);
        sparse_chunk(last_lds + C::TILE, qk + PT);
        lds_barrier();
      }
    }
    __syncthreads();      // the two barriers of the compute waves' epilogue
    __syncthreads();
    return;
  }

  // ---------------- compute waves ----------------
  const int pb = cw / CIT, cb = cw % CIT;      // dA^T tile: points 32 pb .., input channels 32 cb ..
  const int ci = 32 * cb + l31;
  float mfrag[CIN / 2];                        // B[k][j] = M[k][32 cb + j], k = 2 kk + half
#pragma unroll
  for (int kk = 0; kk < CIN / 2; ++kk) mfrag[kk] = M[(size_t)(2 * kk + half) * CIN + ci];
  const float dsc = s_in[ci], dsh = s_in[CIN + ci], c0v = s_in[2 * CIN + ci];
  float sg = 0.f, sgy = 0.f;

  for (int t = 0; t < nchunks; ++t) {
    lds_barrier();
    const float* buf = last_lds + (t & 1) * C::TILE;
    const float* Xs = buf;
    const float* As = buf + CIN * LD;
    const float* Ss = buf + 2 * CIN * LD;
    const long long qk = qbeg + (long long)t * PT;
    f32x16 accd;
    // accumulators start from c0 + S: register 4 j + i of a lane is point 32 pb + 8 j + 4 half + i, input channel ci
    {
      const float* sp = Ss + ci * LD + 32 * pb + 4 * half;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 v = *reinterpret_cast<const float4*>(sp + 8 * j);
        if (NCS == 2) {
          const float4 v2 = *reinterpret_cast<const float4*>(sp + CIN * LD + 8 * j);
          v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
        }
        accd[4 * j + 0] = v.x + c0v; accd[4 * j + 1] = v.y + c0v; accd[4 * j + 2] = v.z + c0v; accd[4 * j + 3] = v.w + c0v;
      }
    }
    const float* ap = As + half * LD + 32 * pb + l31;   // a[k = 2 kk + half][pt = 32 pb + l31]
    constexpr int kDG = CIN / 2 < 8 ? CIN / 2 : 8;
    float fa[2][kDG];
#pragma unroll
    for (int u = 0; u < kDG; ++u) fa[0][u] = ap[2 * u * LD];
#pragma unroll
    for (int gk = 0; gk < CIN / 2 / kDG; ++gk) {
      const int cur = gk & 1, nxt = cur ^ 1;
      if (gk + 1 < CIN / 2 / kDG) {
#pragma unroll
        for (int u = 0; u < kDG; ++u) fa[nxt][u] = ap[2 * ((gk + 1) * kDG + u) * LD];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kDG; ++u)
        accd = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][u], mfrag[gk * kDG + u], accd, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    int b, pk;
    split_point(qk, P, b, pk);
    float* dxb = dx + ((size_t)b * CIN + ci) * P + pk + 32 * pb + 4 * half;
    const float* xr = Xs + ci * LD + 32 * pb + 4 * half;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 o;
      o.x = accd[4 * j + 0]; o.y = accd[4 * j + 1]; o.z = accd[4 * j + 2]; o.w = accd[4 * j + 3];
      *reinterpret_cast<float4*>(dxb + 8 * j) = o;
      const float4 yin = *reinterpret_cast<const float4*>(xr + 8 * j);
      const float g0 = (yin.x * dsc + dsh > 0.f) ? o.x : 0.f, g1 = (yin.y * dsc + dsh > 0.f) ? o.y : 0.f;
      const float g2 = (yin.z * dsc + dsh > 0.f) ? o.z : 0.f, g3 = (yin.w * dsc + dsh > 0.f) ? o.w : 0.f;
      sg += g0; sgy += g0 * yin.x;
      sg += g1; sgy += g1 * yin.y;
      sg += g2; sgy += g2 * yin.z;
      sg += g3; sgy += g3 * yin.w;
    }
  }
  __syncthreads();      // the compute waves are done with the tiles
  float* sred = last_lds;    // [4 waves][2 halves][32][2]
  sred[((cw * 2 + half) * 32 + l31) * 2 + 0] = sg;
  sred[((cw * 2 + half) * 32 + l31) * 2 + 1] = sgy;
  __syncthreads();
  if (threadIdx.x < CIN) {
    const int cbk = threadIdx.x >> 5, l = threadIdx.x & 31;
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 4 / CIT; ++k) {
      const int wsrc = k * CIT + cbk;
      a += sred[((wsrc * 2 + 0) * 32 + l) * 2 + 0] + sred[((wsrc * 2 + 1) * 32 + l) * 2 + 0];
      c += sred[((wsrc * 2 + 0) * 32 + l) * 2 + 1] + sred[((wsrc * 2 + 1) * 32 + l) * 2 + 1];
    }
    part_g[(size_t)threadIdx.x * nt_total + blockIdx.x] = a;
    part_gy[(size_t)threadIdx.x * nt_total + blockIdx.x] = c;
  }
}

// ============================================================================================
// backward, step 3 (off the critical chain, on the deferred weight-gradient stream): the three ingredients of dW per
// workgroup range --  Gram = sum_p a a^T (MFMA; both operands rows of the act tile read as float4 along the points: the
// reduction index is a dummy index),  sa = sum_p a,  dWs[c][ci] = sum over balls of e[c][ball] a[ci][slot point]  (lane =
// output channel c, CIN register accumulators, one conflict-free LDS column read per (ball, ci)).
// 512 threads: all load, waves 0..3 then run the Gram tiles and the row sums, waves 4..7 the per-ball gather.
// ============================================================================================
template <int CIT>
struct DwCfg {
  static constexpr int CIN = 32 * CIT, PT = 128 / CIT, LD = PT + 4, F4 = PT / 4;
  static constexpr int NX = CIN * F4 / 256;              // float4 per loader thread (= 4)
  static constexpr int NT = CIT * CIT;                   // Gram tiles
  static constexpr int TPW = NT >= 4 ? NT / 4 : 1;       // tiles per compute wave
  static constexpr int KSPLIT = NT >= 4 ? 1 : 4 / NT;    // waves sharing a tile split the points (CIT = 1: 4)
  static_assert(NX == 4, "chunk is 4096 floats");
};

// Waves 4..7 load a chunk, apply act() and fill the (double-buffered) tile, then run the per-ball gather on it; waves 0..3
// run the Gram tiles and the row sums.  Chunk t: tile t & 1 is written BEFORE barrier t and read after it by both roles;
// the loaders refill it (chunk t + 2) only after barrier t + 1, which the compute waves reach after their reads.
template <int CIT, int S>
__global__ __launch_bounds__(kLastThreads) void pw_dw_last_kernel(
    int P, int COUT, long long total, int split_len, const float* __restrict__ x, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, const float* __restrict__ e_nat, const uint8_t* __restrict__ arg,
    float* __restrict__ gram_part, float* __restrict__ sa_part, float* __restrict__ dws_part) {
  using C = DwCfg<CIT>;
  constexpr int CIN = C::CIN, PT = C::PT, LD = C::LD, F4 = C::F4, NX = C::NX, TPW = C::TPW, KSPLIT = C::KSPLIT;
  __shared__ __attribute__((aligned(16))) float As2[2][CIN * LD];
  __shared__ float s_in[2 * CIN];
  __shared__ float red[KSPLIT > 1 ? 4 * 1024 : 1];
  const int lane = lane_id();
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tid = threadIdx.x & 255;
  const int l31 = lane & 31, half = lane >> 5;
  const int G = P / S;
  const long long qbeg = (long long)blockIdx.x * split_len;
  const long long qend = max(qbeg, min(qbeg + (long long)split_len, total));
  const int nchunks = (int)((qend - qbeg) / PT);
  for (int c = threadIdx.x; c < CIN; c += kLastThreads) { s_in[c] = in_scale[c]; s_in[CIN + c] = in_shift[c]; }
  __syncthreads();

  if (wv >= 4) {
    // ---------------- loaders + per-ball gather: lane = output channel of slice gw % NSL, balls gidx = gw / NSL mod (4 / NSL)
    const int NSL = COUT / 64;                           // 1, 2 or 4 channel slices
    const int gw = wv - 4;
    const int slice = gw % NSL, gsub = gw / NSL, nsub = 4 / NSL;
    const int cch = slice * 64 + lane;
    float accw[CIN];
#pragma unroll
    for (int i = 0; i < CIN; ++i) accw[i] = 0.f;
    float4 raw[NX];
    auto load_chunk = [&](long long qk) {
      int b, pk;
      split_point(qk, P, b, pk);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int el = tid + 256 * i, row = el / F4, p = pk + (el % F4) * 4;
        raw[i] = *reinterpret_cast<const float4*>(x + ((size_t)b * CIN + row) * P + p);
      }
    };
    if (nchunks > 0) load_chunk(qbeg);
    for (int t = 0; t < nchunks; ++t) {
      const long long qk = qbeg + (long long)t * PT;
      float* As = As2[t & 1];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int el = tid + 256 * i, row = el / F4, k = (el % F4) * 4;
        const float sc = s_in[row], sh = s_in[CIN + row];
        float4 a;
        a.x = fmaxf(raw[i].x * sc + sh, 0.f); a.y = fmaxf(raw[i].y * sc + sh, 0.f);
        a.z = fmaxf(raw[i].z * sc + sh, 0.f); a.w = fmaxf(raw[i].w * sc + sh, 0.f);
        *reinterpret_cast<float4*>(&As[row * LD + k]) = a;
      }
      if (t + 1 < nchunks) load_chunk(qk + PT);
      lds_barrier();
      int b, pk;
      split_point(qk, P, b, pk);
      const int g0 = pk / S;
      for (int gidx = gsub; gidx < PT / S; gidx += nsub) {
        const size_t o = ((size_t)b * COUT + cch) * G + g0 + gidx;
        const float ev = e_nat[o];
        const float* col = As + gidx * S + (int)arg[o];
#pragma unroll
        for (int i = 0; i < CIN; ++i) accw[i] = __builtin_fmaf(ev, col[i * LD], accw[i]);
      }
    }
    if (KSPLIT > 1) { __syncthreads(); }
    float* wout = dws_part + ((size_t)blockIdx.x * nsub + gsub) * COUT * CIN + (size_t)cch * CIN;
#pragma unroll
    for (int i = 0; i < CIN; i += 4)
      *reinterpret_cast<float4*>(wout + i) = make_float4(accw[i], accw[i + 1], accw[i + 2], accw[i + 3]);
    return;
  }
  // ---------------- compute waves: Gram tiles t = wv * TPW + k (row block t / CIT, column block t % CIT); CIT = 1: the four
  // waves take point quarters of the one tile.  Row sums: lane <-> row wv * 64 + lane.
  f32x16 accg[TPW];
#pragma unroll
  for (int k = 0; k < TPW; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) accg[k][r] = 0.f;
  float rowsum = 0.f;
  for (int t = 0; t < nchunks; ++t) {
    lds_barrier();
    const float* As = As2[t & 1];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int tix = KSPLIT > 1 ? 0 : wv * TPW + k;
      const int ti = tix / CIT, tj = tix % CIT;
      constexpr int KP = PT / KSPLIT;
      const int k0 = KSPLIT > 1 ? wv * KP : 0;
      const float* ap = As + (32 * ti + l31) * LD + k0 + 4 * half;
      const float* bp = As + (32 * tj + l31) * LD + k0 + 4 * half;
#pragma unroll
      for (int j = 0; j < KP / 8; ++j) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + 8 * j);
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * j);
        accg[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, accg[k], 0, 0, 0);
        accg[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, accg[k], 0, 0, 0);
        accg[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, accg[k], 0, 0, 0);
        accg[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, accg[k], 0, 0, 0);
      }
    }
    const int row = wv * 64 + lane;
    if (row < CIN) {
      const float* rp = As + row * LD;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < F4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(rp + 4 * j);
        s += (v.x + v.y) + (v.z + v.w);
      }
      rowsum += s;
    }
  }
  // ---- per-workgroup partials ----
  float* gout = gram_part + (size_t)blockIdx.x * CIN * CIN;
  if (KSPLIT > 1) {       // CIT = 1: four waves hold point quarters of the one tile
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv * 1024 + r * 64 + lane] = accg[0][r];
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (red[r * 64 + lane] + red[1024 + r * 64 + lane]) + (red[2048 + r * 64 + lane] + red[3072 + r * 64 + lane]);
        gout[(size_t)mfma_row(r, lane) * CIN + l31] = v;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int tix = wv * TPW + k, ti = tix / CIT, tj = tix % CIT;
#pragma unroll
      for (int r = 0; r < 16; ++r) gout[(size_t)(32 * ti + mfma_row(r, lane)) * CIN + 32 * tj + l31] = accg[k][r];
    }
  }
  const int row = wv * 64 + lane;
  if (row < CIN) sa_part[(size_t)blockIdx.x * CIN + row] = rowsum;
}

// ============================================================================================
// backward, step 4:  dW[c][ci] = dWs[c][ci] + cb_c sa[ci] + cc_c sum_k W[c][k] Gram[k][ci]   (dWs, sa, Gram already summed
// over the workgroup partials by wgrad_reduce_multi_kernel).  float64 for the two dense terms: they cancel against each
// other to the extent the layer's output is not centred.  grid (ceil(CIN / 64), COUT), 64 threads.
// ============================================================================================
__global__ __launch_bounds__(64) void pw_dw_last_finish_kernel(int COUT, int CIN, const float* __restrict__ w,
                                                               const float* __restrict__ bwdc,
                                                               const float* __restrict__ gram,
                                                               const float* __restrict__ sa,
                                                               const float* __restrict__ dws, float* __restrict__ dw) {
  const int c = blockIdx.y, ci = blockIdx.x * 64 + threadIdx.x;
  if (ci >= CIN) return;
  const double cb = bwdc[COUT + c], cc = bwdc[2 * COUT + c];
  const float* wr = w + (size_t)c * CIN;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;      // CIN % 32 == 0: four independent chains keep the loads in flight
#pragma unroll 2
  for (int k = 0; k < CIN; k += 4) {
    a0 += (double)wr[k + 0] * (double)gram[(size_t)(k + 0) * CIN + ci];
    a1 += (double)wr[k + 1] * (double)gram[(size_t)(k + 1) * CIN + ci];
    a2 += (double)wr[k + 2] * (double)gram[(size_t)(k + 2) * CIN + ci];
    a3 += (double)wr[k + 3] * (double)gram[(size_t)(k + 3) * CIN + ci];
  }
  const double acc = (a0 + a1) + (a2 + a3);
  dw[(size_t)c * CIN + ci] = (float)((double)dws[(size_t)c * CIN + ci] + cb * (double)sa[ci] + cc * acc);
}

int g_last_target = 128;      // workgroups of pw_bwd_last_kernel / pw_dw_last_kernel (istnet_pw_last_set_tuning)
int g_last_enable = 1;

bool last_ok(int cin, int cout, int p, int nsample) {
  if (!g_last_enable || p <= 0 || p % 128) return false;
  if (nsample != 16 && nsample != 32) return false;
  if (!(cin == 32 || cin == 64 || cin == 128)) return false;
  return cout == 64 || cout == 128 || cout == 256;
}
int last_pt(int cin) { return 128 / (cin / 32); }
int last_len(int b, int cin, int p) {
  const long long total = (long long)b * p;
  const int pt = last_pt(cin);
  long long len = (total + g_last_target - 1) / g_last_target;
  len = (len + pt - 1) / pt * pt;
  if (len < 2 * pt) len = 2 * pt;
  return (int)len;
}

}  // namespace

extern "C" {

int istnet_pw_last_set_tuning(int key, int value) {
  if (key == 0) { g_last_target = value > 0 ? value : 128; return 0; }
  if (key == 1) { g_last_enable = value != 0; return 0; }
  return ISTNET_PN2_EINVAL;
}

int istnet_pw_bwd_last_ok(int cin, int cout, int p, int nsample) { return last_ok(cin, cout, p, nsample) ? 1 : 0; }

int istnet_pw_bwd_last_splits(int b, int cin, int cout, int p, int nsample) {
  if (!last_ok(cin, cout, p, nsample) || b <= 0) return 0;
  const long long total = (long long)b * p;
  const int len = last_len(b, cin, p);
  return (int)((total + len - 1) / len);
}

int istnet_bn_finalize_pool_apply(int b, int c, int g, int nt, double count, const float* part_sum, const float* part_sq,
                                  const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                  float* running_var, float* bn, const float* gval, float* out, long long out_bstride,
                                  void* stream) {
  if (b <= 0 || c <= 0 || g <= 0 || nt <= 0 || count <= 0.0 || !part_sum || !part_sq || !gamma || !beta || !bn || !gval || !out)
    return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(bn_finalize_pool_apply_kernel, dim3(c), dim3(kFinThreads), 0, as_stream(stream), c, b, g, nt, count,
                     part_sum, part_sq, gamma, beta, eps, momentum, running_mean, running_var, bn, gval, out,
                     out_bstride > 0 ? out_bstride : (long long)c * g);
  return (int)hipGetLastError();
}

int istnet_pool_apply(int b, int c, int g, const float* bn, const float* gval, float* out, long long out_bstride,
                      void* stream) {
  if (b <= 0 || c <= 0 || g <= 0 || !bn || !gval || !out) return ISTNET_PN2_EINVAL;
  const long long rows = (long long)b * c;
  hipLaunchKernelGGL(pool_apply_kernel, dim3(ceil_div(g, 256), (unsigned)(rows < 65535 ? rows : 65535)), dim3(256), 0,
                     as_stream(stream), c, g, (int)rows, bn, gval, out, out_bstride > 0 ? out_bstride : (long long)c * g);
  return (int)hipGetLastError();
}

int istnet_pw_last_prep(int b, int cin, int cout, int p, int nsample, const float* w, const float* bn, const float* bwdc,
                        const float* d_pooled, long long pooled_bstride, const float* gval, const unsigned char* arg,
                        float* e_nat, float* e_t, unsigned char* slot_t, float* m, float* c0, void* stream) {
  if (b <= 0 || !last_ok(cin, cout, p, nsample)) return ISTNET_PN2_EINVAL;
  if (!w || !bn || !bwdc || !d_pooled || !gval || !arg || !e_nat || !e_t || !slot_t || !m || !c0) return ISTNET_PN2_EINVAL;
  const int G = p / nsample;
  const int nA = b * ceil_div(G, 64) * (cout / 64);
  const int nB = (cin / 8) * (cin / 8);
  hipLaunchKernelGGL(pw_last_prep_kernel, dim3(nA + nB), dim3(256), 0, as_stream(stream), b, cout, cin, G, nA, w, bn, bwdc,
                     d_pooled, pooled_bstride > 0 ? pooled_bstride : (long long)cout * G, gval, arg, e_nat, e_t, slot_t, m,
                     c0);
  return (int)hipGetLastError();
}

int istnet_pw_bwd_last(int b, int cin, int cout, int p, int nsample, const float* w, const float* x, const float* bn_in,
                       const float* m, const float* c0, const float* e_t, const unsigned char* slot_t, float* dx,
                       float* part_g, float* part_gy, void* stream) {
  if (b <= 0 || !last_ok(cin, cout, p, nsample) || (long long)b * p >= (1LL << 31)) return ISTNET_PN2_EINVAL;
  if (!w || !x || !bn_in || !m || !c0 || !e_t || !slot_t || !dx || !part_g || !part_gy) return ISTNET_PN2_EINVAL;
  const int len = last_len(b, cin, p);
  const int splits = istnet_pw_bwd_last_splits(b, cin, cout, p, nsample);
#define ISTNET_BWD_LAST(CIT, S)                                                                                    \
  do {                                                                                                             \
    constexpr size_t lds = LastCfg<CIT, S>::LDS_BYTES;                                                             \
    static bool attr_set = false;                                                                                  \
    if (!attr_set) {                                                                                               \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_bwd_last_kernel<CIT, S>),                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                \
        return ISTNET_PN2_EINVAL;                                                                                  \
      attr_set = true;                                                                                             \
    }                                                                                                              \
    hipLaunchKernelGGL((pw_bwd_last_kernel<CIT, S>), dim3(splits), dim3(kLastThreads), lds, as_stream(stream), p,  \
                       cout, (long long)b * p, len, w, x, bn_in, bn_in + cin, m, c0, e_t, slot_t, dx, part_g, part_gy, \
                       splits);                                                                                    \
  } while (0)
  if (cin == 32) { if (nsample == 16) ISTNET_BWD_LAST(1, 16); else ISTNET_BWD_LAST(1, 32); }
  else if (cin == 64) { if (nsample == 16) ISTNET_BWD_LAST(2, 16); else ISTNET_BWD_LAST(2, 32); }
  else { if (nsample == 16) ISTNET_BWD_LAST(4, 16); else ISTNET_BWD_LAST(4, 32); }
#undef ISTNET_BWD_LAST
  return (int)hipGetLastError();
}

int istnet_pw_dw_last_parts(int b, int cin, int cout, int p, int nsample) {
  // partial sets of dWs per workgroup: the gather waves that share a channel slice take alternate balls
  return istnet_pw_bwd_last_splits(b, cin, cout, p, nsample) * (4 / (cout / 64));
}

int istnet_pw_dw_last(int b, int cin, int cout, int p, int nsample, const float* x, const float* bn_in,
                      const float* e_nat, const unsigned char* arg, float* gram_part, float* sa_part, float* dws_part,
                      void* stream) {
  if (b <= 0 || !last_ok(cin, cout, p, nsample) || (long long)b * p >= (1LL << 31)) return ISTNET_PN2_EINVAL;
  if (!x || !bn_in || !e_nat || !arg || !gram_part || !sa_part || !dws_part) return ISTNET_PN2_EINVAL;
  const int len = last_len(b, cin, p);
  const int splits = istnet_pw_bwd_last_splits(b, cin, cout, p, nsample);
#define ISTNET_DW_LAST(CIT, S)                                                                                     \
  hipLaunchKernelGGL((pw_dw_last_kernel<CIT, S>), dim3(splits), dim3(kLastThreads), 0, as_stream(stream), p, cout, \
                     (long long)b * p, len, x, bn_in, bn_in + cin, e_nat, arg, gram_part, sa_part, dws_part)
  if (cin == 32) { if (nsample == 16) ISTNET_DW_LAST(1, 16); else ISTNET_DW_LAST(1, 32); }
  else if (cin == 64) { if (nsample == 16) ISTNET_DW_LAST(2, 16); else ISTNET_DW_LAST(2, 32); }
  else { if (nsample == 16) ISTNET_DW_LAST(4, 16); else ISTNET_DW_LAST(4, 32); }
#undef ISTNET_DW_LAST
  return (int)hipGetLastError();
}

int istnet_pw_dw_last_finish(int cin, int cout, const float* w, const float* bwdc, const float* gram, const float* sa,
                             const float* dws, float* dw, void* stream) {
  if (cin <= 0 || cout <= 0 || !w || !bwdc || !gram || !sa || !dws || !dw) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(pw_dw_last_finish_kernel, dim3(ceil_div(cin, 64), cout), dim3(64), 0, as_stream(stream), cout, cin,
                     w, bwdc, gram, sa, dws, dw);
  return (int)hipGetLastError();
}

}  // extern "C"
