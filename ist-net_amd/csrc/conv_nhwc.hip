// Channels-last float32 2-D convolution as an implicit GEMM on the fp32 matrix cores (include/istnet_conv.h): the 3x3 / 1x1
// convolutions of the RGB branch's ResNet-18 trunk (reference model/resnet.py:18-67,109-202), forward, backward-data and
// backward-weights.  gfx950 only, wave64, v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation).
//
// One GEMM view serves all three products (M x N += A (M x K) . B (K x N), K walked in chunks of 32 channels of one filter tap):
//   forward         M = output pixels, N = cout, K = taps x cin    A = in at the tap-shifted pixel (zero outside the image)
//   backward-data   M = input pixels,  N = cin,  K = taps x cout   A = dout at the pixel that tap connects to this one
//   backward-weights M = cout, N = cin (per tap), K = output pixels, split over workgroups
// A workgroup = 4 waves = one 128 x 128 (or 128 x 64 / 64 x 128 / 64 x 64) output tile; both operands go global -> registers
// -> LDS (double buffered, one barrier per chunk), the next chunk's global loads are issued before the current chunk's 64
// MFMAs per wave.  NHWC rows are K-contiguous, so the pixel operand is staged [row][32 + 4] and read as one ds_read_b128 per
// four k; operands whose contiguous direction is the OUTPUT index (the weights in backward-data, both operands in
// backward-weights) are staged [k][cols + 4] and read as scalars with consecutive lanes on consecutive words.
//
// Measured on the way (512 -> 512 at 24 x 24, B = 32, forward; profiles/r04_conv_microbench.txt holds the final table):
//   first version (prefetch arrays captured by lambdas -> scratch, conditional loads)            62 TFLOP/s
//   named prefetch registers, unconditional clamped loads, stride as a template parameter        87
//   two chunks of loads in flight / eight waves per workgroup / tiles dealt to the XCDs by row   87 / 85 / 84  (not the bound)
//   K split 8 ways for every tile (576 tiles on 512 slots are 1.125 rounds)                     103
//   per-tap row addresses formed once per tap                                                   108
//   whole rounds unsplit, only the remaining 64 tiles split 8 ways                              121  (MIOpen: 111-117)
//   K chunks of 16 instead of 32 (half the LDS: -DISTNET_CONV_KC=16)                            104, and +1 ms on the step
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>

#include <type_traits>

#include "../../include/istnet_conv.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kThreads = 256;
#ifndef ISTNET_CONV_KC
#define ISTNET_CONV_KC 32
#endif
constexpr int KC = ISTNET_CONV_KC;       // K chunk (32, or 16 for half the LDS)
constexpr int LPR = KC / 4;  // lanes per K-contiguous tile row (one float4 each)
constexpr int LDK = KC + 4;  // pitch of a K-contiguous LDS row: 144 bytes keeps the 16-byte reads of 32 rows off each other's banks

__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ float f4_at(const float4& v, int t) { return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w)); }

struct ConvGeom {
  int B, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
};

// Prefetch registers of one K chunk.  Named members, no arrays: an array of float4 reached through a lambda or a reference
// is placed in scratch by this compiler (ISA of the first version of this file: four scratch stores + loads per chunk).
struct Stage {
  float4 a0, a1, a2, a3, b0, b1, b2, b3;
  unsigned ok;        // bit i: A row i is inside the image (bits 8 + i: B row i, backward-weights only)
};
template <int I> __device__ __forceinline__ float4& st_a(Stage& s) {
  if constexpr (I == 0) return s.a0; else if constexpr (I == 1) return s.a1; else if constexpr (I == 2) return s.a2; else return s.a3;
}
template <int I> __device__ __forceinline__ float4& st_b(Stage& s) {
  if constexpr (I == 0) return s.b0; else if constexpr (I == 1) return s.b1; else if constexpr (I == 2) return s.b2; else return s.b3;
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// value select per component (a select between two float4 OBJECTS becomes a select between their addresses: scratch)
__device__ __forceinline__ float4 keep_if(bool ok, const float4& v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// MODE 0: forward (c = out, a_src = in); MODE 1: backward-data (c = din, a_src = dout).  STRIDE 1 or 2.
template <int MT, int NT, int WM, int WN, int MODE, int STRIDE>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 2) void conv_igemm_kernel(ConvGeom g, const float* __restrict__ a_src,
                                                                const float* __restrict__ wgt, float* __restrict__ c_dst,
                                                                float* __restrict__ ws, int chunks_per_split, int nsplits,
                                                                int tile_m_first, int tile_m_count) {
  constexpr int TM = MT / (32 * WM), TN = NT / (32 * WN);
  constexpr int NTHR = 64 * WM * WN;   // 4 or 8 waves per workgroup
  constexpr int RPP = NTHR / LPR;      // tile rows covered by one pass of the workgroup (LPR lanes x float4 = one row of the chunk)
  constexpr int AR = MT / RPP;         // float4 of the A tile per thread: rows tid / LPR + RPP i, columns 4 (tid % LPR) .. + 3
  constexpr int BR = NT / RPP;         // float4 of the B tile per thread (the same count in both modes)
  constexpr int LDB1 = NT + 4;         // MODE 1: B tile [KC][NT + 4]
  constexpr int A_STAGE = MT * LDK;
  constexpr int B_STAGE = MODE == 0 ? NT * LDK : KC * LDB1;
  static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1 && AR >= 1 && AR <= 4 && BR >= 1 && BR <= 4, "4 or 8 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][A_STAGE]
  float* Bs = smem + 2 * A_STAGE;      // [2][B_STAGE]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wv / WN, wn = wv % WN;
  const int taps = g.KH * g.KW;
  const int Ka = MODE == 0 ? g.Cin : g.Cout;                            // channels of the A source
  const int Ncols = MODE == 0 ? g.Cout : g.Cin;
  const int MH = MODE == 0 ? g.OH : g.H, MW = MODE == 0 ? g.OW : g.W;   // pixel grid of the output rows
  const int SH = MODE == 0 ? g.H : g.OH, SW = MODE == 0 ? g.W : g.OW;   // pixel grid of the A source
  const long long M = (long long)g.B * MH * MW;
  // Tile order: the hardware deals consecutive workgroups round-robin to the 8 XCDs, each with its own L2.  Workgroup id ->
  // (XCD x = id % 8, j = id / 8); XCD x walks the row tiles x, x + 8, ... and, for each, every column tile in turn, so the
  // workgroups resident on an XCD at one time are ~16 row tiles x all column tiles: an A tile is fetched into that L2 once
  // for its (cout / 128) users and a B tile once for its 16 (row-major order over the whole grid sent the users of one
  // tile to eight different L2s: 2.7 TB/s of operand traffic from beyond L2 at 87 TFLOP/s).
  // K may be split over `nsplits` workgroups per tile (split fastest): 576 tiles on 512 workgroup slots are two rounds, the
  // second one an eighth full; 576 x 8 units of an eighth of K are nine full rounds.  Split s accumulates chunks
  // [s, s + 1) * chunks_per_split into its own slab of `ws`, summed afterwards in a fixed order (deterministic, no atomics).
  const int tiles_n = Ncols / NT;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int split = jj % nsplits, jt = jj / nsplits;
  const int tile_local = (jt / tiles_n) * 8 + xcd;                       // row tile inside this launch's window
  if (tile_local >= tile_m_count) return;                                // (the grid is padded to a multiple of 8 row tiles)
  const long long m0 = (long long)(tile_m_first + tile_local) * MT;
  const int n0 = (jt % tiles_n) * NT;
  const int acol = (tid % LPR) * 4;
  // this thread's A rows: image base row (b * SH), y, x of the output pixel; rv: the row exists (pm < M)
  int rb[AR], ry[AR], rx[AR];
  unsigned rvalid = 0;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const long long pm = m0 + (tid / LPR) + RPP * i;
    const long long pc = pm < M ? pm : M - 1;
    const int b = (int)(pc / (MH * MW));
    const int rem = (int)(pc - (long long)b * MH * MW);
    ry[i] = rem / MW;
    rx[i] = rem - ry[i] * MW;
    rb[i] = b * SH;
    rvalid |= (pm < M ? 1u : 0u) << i;
  }
  const int nb = Ka / KC;
  const int c_first = split * chunks_per_split;
  const int nchunks = min(chunks_per_split, taps * nb - c_first);       // of this split (>= 1: the host sizes the splits so)

  // global -> registers, branch-free: every load is issued from a valid (clamped) address and rows outside the image are
  // zeroed when they go to LDS -- a load under a condition makes the wait-count pass drain the whole prefetch.  The row
  // addresses and the inside-the-image bits depend on the tap only: set_tap() forms them once per tap (every cin / 32 chunks).
  size_t aoff[AR];
  unsigned tap_ok = 0;
  auto set_tap = [&](int ky, int kx) {
    tap_ok = 0;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int sy, sx;
      bool ok = (rvalid >> i) & 1u;
      if (MODE == 0) {
        sy = ry[i] * STRIDE + ky - g.pad;
        sx = rx[i] * STRIDE + kx - g.pad;
      } else {
        const int ty = ry[i] + g.pad - ky, tx = rx[i] + g.pad - kx;
        if (STRIDE == 1) { sy = ty; sx = tx; }
        else { ok = ok && ((ty | tx) & 1) == 0; sy = ty >> 1; sx = tx >> 1; }
      }
      ok = ok && sy >= 0 && sy < SH && sx >= 0 && sx < SW;
      aoff[i] = ((size_t)(rb[i] + clampi(sy, 0, SH - 1)) * SW + clampi(sx, 0, SW - 1)) * Ka + acol;
      tap_ok |= (ok ? 1u : 0u) << i;
    }
  };
  auto issue = [&](Stage& s, int ky, int kx, int cb) {
    const int tap = ky * g.KW + kx;
    s.ok = tap_ok;
    static_for<AR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      st_a<i>(s) = *reinterpret_cast<const float4*>(a_src + aoff[i] + (size_t)cb * KC);
    });
    static_for<BR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (MODE == 0) {
        const int n = n0 + (tid / LPR) + RPP * i;
        st_b<i>(s) = *reinterpret_cast<const float4*>(wgt + ((size_t)n * taps + tap) * g.Cin + (size_t)cb * KC + acol);
      } else {
        const int e = tid + NTHR * i;
        const int k = e / (NT / 4), c4 = e % (NT / 4);
        st_b<i>(s) = *reinterpret_cast<const float4*>(wgt + ((size_t)(cb * KC + k) * taps + tap) * g.Cin + n0 + 4 * c4);
      }
    });
  };
  auto commit = [&](Stage& s, int buf) {
    float* as = As + buf * A_STAGE;
    float* bs = Bs + buf * B_STAGE;
    static_for<AR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      *reinterpret_cast<float4*>(as + ((tid / LPR) + RPP * i) * LDK + acol) = keep_if((s.ok >> i) & 1u, st_a<i>(s));
    });
    static_for<BR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (MODE == 0) {
        *reinterpret_cast<float4*>(bs + ((tid / LPR) + RPP * i) * LDK + acol) = st_b<i>(s);
      } else {
        const int e = tid + NTHR * i;
        *reinterpret_cast<float4*>(bs + (e / (NT / 4)) * LDB1 + 4 * (e % (NT / 4))) = st_b<i>(s);
      }
    });
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // Two chunks of global loads in flight (stages sa, sb alternate): a chunk's loads are issued two MFMA phases before they
  // are written to LDS, so the write waits only for the OLDER stage (`vmcnt(8)`) and HBM / L2 latency of 2 - 3 us is covered
  // by 2 x 64 MFMAs per wave.  LDS stays double buffered, one barrier per chunk.
  Stage sa, sb;
  int ky = (c_first / nb) / g.KW, kx = (c_first / nb) % g.KW, cb = c_first % nb;   // the chunk being PREFETCHED (uniform)
  auto advance = [&]() {
    if (++cb == nb) {
      cb = 0;
      if (++kx == g.KW) { kx = 0; ++ky; }
      if (ky == g.KH) { ky = g.KH - 1; kx = g.KW - 1; cb = nb - 1; }    // past the end: stay on the last chunk (loaded, unused)
      else set_tap(ky, kx);
    }
  };
  set_tap(ky, kx);
  auto mma = [&](int buf) {
    const float* as = As + buf * A_STAGE + ((wm * TM) * 32 + l31) * LDK + 4 * half;
    const float* bs = MODE == 0 ? Bs + buf * B_STAGE + ((wn * TN) * 32 + l31) * LDK + 4 * half
                                : Bs + buf * B_STAGE + (4 * half) * LDB1 + (wn * TN) * 32 + l31;
#pragma unroll
    for (int g8 = 0; g8 < KC / 8; ++g8) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) af[mi] = *reinterpret_cast<const float4*>(as + mi * 32 * LDK + 8 * g8);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        if (MODE == 0) {
          bf[ni] = *reinterpret_cast<const float4*>(bs + ni * 32 * LDK + 8 * g8);
        } else {
          const float* bp = bs + (8 * g8) * LDB1 + ni * 32;
          bf[ni] = make_float4(bp[0], bp[LDB1], bp[2 * LDB1], bp[3 * LDB1]);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_at(af[mi], t), f4_at(bf[ni], t), acc[mi][ni], 0, 0, 0);
    }
  };
  issue(sa, ky, kx, cb);               // chunk 0
  commit(sa, 0);
  if constexpr (WM * WN == 4) {
    advance();
    issue(sa, ky, kx, cb);             // chunk 1 (or the last chunk again)
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
      advance();
      issue(sb, ky, kx, cb);           // chunk c + 2
      __builtin_amdgcn_sched_barrier(0);
      mma(0);                          // chunk c
      __builtin_amdgcn_sched_barrier(0);
      commit(sa, 1);                   // chunk c + 1
      __syncthreads();
      if (c + 1 >= nchunks) break;
      advance();
      issue(sa, ky, kx, cb);           // chunk c + 3
      __builtin_amdgcn_sched_barrier(0);
      mma(1);                          // chunk c + 1
      __builtin_amdgcn_sched_barrier(0);
      commit(sb, 0);                   // chunk c + 2
      __syncthreads();
    }
  } else {
    // eight waves (four per SIMD with two workgroups per CU, 128 VGPRs each): one stage in flight, the other waves of the
    // SIMD cover the latency
    __syncthreads();
    int buf = 0;
    for (int c = 0; c < nchunks; ++c) {
      advance();
      issue(sa, ky, kx, cb);           // chunk c + 1
      __builtin_amdgcn_sched_barrier(0);
      mma(buf);
      __builtin_amdgcn_sched_barrier(0);
      commit(sa, buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- epilogue: rows m0 + ..., columns n0 + ...; a lane holds column l31 of 16 rows per 32 x 32 tile ----
  // (a split launch covers the rows of its window only: slabs of (M - first row) x Ncols, indexed from the window's first row)
  const long long m_first = (long long)tile_m_first * MT;
  float* dst = nsplits == 1 ? c_dst : ws + ((size_t)split * (M - m_first) - m_first) * Ncols;
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long row = m0 + (wm * TM + mi) * 32 + mfma_row(r, lane);
      if (row < M) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) dst[(size_t)row * Ncols + n0 + (wn * TN + ni) * 32 + l31] = acc[mi][ni][r];
      }
    }
}

// ============================================================================================
// Split-precision forward (opt-in experiment, istnet_conv_set_tuning(1, 1); DESIGN.md "split precision").
// Every fp32 operand is split EXACTLY into three bf16 terms x = hi + mid + lo (each takes the top 8 significand bits of what
// is left: truncation, so the remainders are exact fp32 subtractions), and a.b is evaluated as the six products
//   a_lo b_hi + a_hi b_lo + a_mid b_mid + a_mid b_hi + a_hi b_mid + a_hi b_hi
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Every bf16 x bf16 product is exact in fp32; the three dropped terms
// (mid lo, lo mid, lo lo) are <= 2^-24 |a b| each -- the size of ONE fp32 rounding of the product.  The bf16 matrix pipe runs
// 16x the fp32 one, so six products cost 6/16 of v_mfma_f32_32x32x2_f32: the roof moves from 157 to ~417 TFLOP/s.
// Operands are split once, when a chunk goes from registers to LDS (three bf16 planes per operand tile, rows K-contiguous,
// pitch KCS + 8 elements: 16-byte aligned, conflict-free 16-byte reads); a lane's MFMA fragment is one ds_read_b128 per plane.
// 128 x NT tiles, eight waves.
// ============================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Split3 { uint2 hi, mid, lo; };      // 4 consecutive k as packed bf16 pairs
__device__ __forceinline__ Split3 split3(const float4& v) {
  const unsigned x0 = __float_as_uint(v.x), x1 = __float_as_uint(v.y), x2 = __float_as_uint(v.z), x3 = __float_as_uint(v.w);
  // remainders after the top 8 significand bits: exact (hi shares sign and exponent with x)
  const float r0 = v.x - __uint_as_float(x0 & 0xffff0000u), r1 = v.y - __uint_as_float(x1 & 0xffff0000u);
  const float r2 = v.z - __uint_as_float(x2 & 0xffff0000u), r3 = v.w - __uint_as_float(x3 & 0xffff0000u);
  const unsigned m0 = __float_as_uint(r0), m1 = __float_as_uint(r1), m2 = __float_as_uint(r2), m3 = __float_as_uint(r3);
  const float q0 = r0 - __uint_as_float(m0 & 0xffff0000u), q1 = r1 - __uint_as_float(m1 & 0xffff0000u);
  const float q2 = r2 - __uint_as_float(m2 & 0xffff0000u), q3 = r3 - __uint_as_float(m3 & 0xffff0000u);
  Split3 s;
  // v_perm_b32: the upper halves of two dwords side by side (element k in the low half: little-endian bf16 order)
  s.hi = make_uint2(__builtin_amdgcn_perm(x1, x0, 0x07060302u), __builtin_amdgcn_perm(x3, x2, 0x07060302u));
  s.mid = make_uint2(__builtin_amdgcn_perm(m1, m0, 0x07060302u), __builtin_amdgcn_perm(m3, m2, 0x07060302u));
  s.lo = make_uint2(__builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u),
                    __builtin_amdgcn_perm(__float_as_uint(q3), __float_as_uint(q2), 0x07060302u));
  return s;
}

template <int NT, int STRIDE, int WM, int KCS, int MINW>      // WM = 4: eight waves of 32 x NT/2 (WM = 2: four of 64 x NT/2, measured slower)
__global__ __launch_bounds__(128 * WM, MINW) void conv_igemm_split_kernel(ConvGeom g, const float* __restrict__ a_src,
                                                                       const float* __restrict__ wgt, float* __restrict__ c_dst,
                                                                       float* __restrict__ ws, int chunks_per_split, int nsplits,
                                                                       int tile_m_first, int tile_m_count) {
  constexpr int MT = 128, WN = 2, NTHR = 64 * WM * WN;
  constexpr int KC_ = KCS, LPR_ = KCS / 4, LDKH_ = KCS + 8;      // K chunk of THIS kernel (32, or 16: half the LDS, two workgroups per CU)
  constexpr int TM = MT / (32 * WM), TN = NT / (32 * WN);
  constexpr int RPP = NTHR / LPR_;                       // tile rows per pass of the workgroup
  constexpr int AR = MT / RPP, BR = NT / RPP;           // float4 per thread and operand
  constexpr int A_PLANE = MT * LDKH_, B_PLANE = NT * LDKH_;               // bf16 elements
  constexpr int STAGE = 3 * (A_PLANE + B_PLANE);
  static_assert(AR >= 1 && AR <= 4 && BR >= 1 && BR <= 4 && TM >= 1 && TN >= 1, "128 x 128 or 128 x 64 tiles, 4 or 8 waves");
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];       // [2][STAGE]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wv / WN, wn = wv % WN;
  const int taps = g.KH * g.KW;
  const int Ka = g.Cin, Ncols = g.Cout;
  const int MH = g.OH, MW = g.OW, SH = g.H, SW = g.W;
  const long long M = (long long)g.B * MH * MW;
  const int tiles_n = Ncols / NT;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;                  // XCD-aware tile order: see conv_igemm_kernel
  const int split = jj % nsplits, jt = jj / nsplits;
  const int tile_local = (jt / tiles_n) * 8 + xcd;
  if (tile_local >= tile_m_count) return;
  const long long m0 = (long long)(tile_m_first + tile_local) * MT;
  const int n0 = (jt % tiles_n) * NT;
  const int acol = (tid % LPR_) * 4;
  int rb[AR], ry[AR], rx[AR];
  unsigned rvalid = 0;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const long long pm = m0 + (tid / LPR_) + RPP * i;
    const long long pc = pm < M ? pm : M - 1;
    const int b = (int)(pc / (MH * MW));
    const int rem = (int)(pc - (long long)b * MH * MW);
    ry[i] = rem / MW;
    rx[i] = rem - ry[i] * MW;
    rb[i] = b * SH;
    rvalid |= (pm < M ? 1u : 0u) << i;
  }
  const int nb = Ka / KC_;
  const int c_first = split * chunks_per_split;
  const int nchunks = min(chunks_per_split, taps * nb - c_first);
  size_t aoff[AR];
  unsigned tap_ok = 0;
  auto set_tap = [&](int ky, int kx) {
    tap_ok = 0;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int sy = ry[i] * STRIDE + ky - g.pad, sx = rx[i] * STRIDE + kx - g.pad;
      const bool ok = ((rvalid >> i) & 1u) && sy >= 0 && sy < SH && sx >= 0 && sx < SW;
      aoff[i] = ((size_t)(rb[i] + clampi(sy, 0, SH - 1)) * SW + clampi(sx, 0, SW - 1)) * Ka + acol;
      tap_ok |= (ok ? 1u : 0u) << i;
    }
  };
  auto issue = [&](Stage& st, int ky, int kx, int cb) {
    const int tap = ky * g.KW + kx;
    st.ok = tap_ok;
    static_for<AR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      st_a<i>(st) = *reinterpret_cast<const float4*>(a_src + aoff[i] + (size_t)cb * KC_);
    });
    static_for<BR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int n = n0 + (tid / LPR_) + RPP * i;
      st_b<i>(st) = *reinterpret_cast<const float4*>(wgt + ((size_t)n * taps + tap) * g.Cin + (size_t)cb * KC_ + acol);
    });
  };
  auto put = [&](unsigned short* plane0, int plane_elems, int row, const float4& v) {
    const Split3 s = split3(v);
    unsigned short* p = plane0 + row * LDKH_ + acol;
    *reinterpret_cast<uint2*>(p) = s.hi;
    *reinterpret_cast<uint2*>(p + plane_elems) = s.mid;
    *reinterpret_cast<uint2*>(p + 2 * plane_elems) = s.lo;
  };
  auto commit = [&](Stage& st, int buf) {
    unsigned short* as = smem_h + buf * STAGE;
    unsigned short* bs = as + 3 * A_PLANE;
    static_for<AR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      put(as, A_PLANE, (tid / LPR_) + RPP * i, keep_if((st.ok >> i) & 1u, st_a<i>(st)));
    });
    static_for<BR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      put(bs, B_PLANE, (tid / LPR_) + RPP * i, st_b<i>(st));
    });
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  auto mma = [&](int buf) {
    const unsigned short* as = smem_h + buf * STAGE + ((wm * TM) * 32 + l31) * LDKH_ + 8 * half;
    const unsigned short* bs = smem_h + buf * STAGE + 3 * A_PLANE + ((wn * TN) * 32 + l31) * LDKH_ + 8 * half;
#pragma unroll
    for (int g16 = 0; g16 < KC_ / 16; ++g16) {
      bf16x8 ah[TM], am[TM], al[TM];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const unsigned short* ap = as + mi * 32 * LDKH_ + 16 * g16;
        ah[mi] = *reinterpret_cast<const bf16x8*>(ap);
        am[mi] = *reinterpret_cast<const bf16x8*>(ap + A_PLANE);
        al[mi] = *reinterpret_cast<const bf16x8*>(ap + 2 * A_PLANE);
      }
      bf16x8 bh[TN], bm[TN], bl[TN];
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const unsigned short* bp = bs + ni * 32 * LDKH_ + 16 * g16;
        bh[ni] = *reinterpret_cast<const bf16x8*>(bp);
        bm[ni] = *reinterpret_cast<const bf16x8*>(bp + B_PLANE);
        bl[ni] = *reinterpret_cast<const bf16x8*>(bp + 2 * B_PLANE);
      }
      // products outer, accumulators inner: consecutive MFMAs never share an accumulator (a dependent bf16 MFMA waits for
      // its predecessor's result; with TM * TN >= 2 tiles per wave the chain of one tile hides behind the other's)
#define ISTNET_P(A, B)                                                                         \
  _Pragma("unroll") for (int mi = 0; mi < TM; ++mi)                                             \
    _Pragma("unroll") for (int ni = 0; ni < TN; ++ni)                                           \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mi], B[ni], acc[mi][ni], 0, 0, 0);
      ISTNET_P(al, bh) ISTNET_P(ah, bl) ISTNET_P(am, bm) ISTNET_P(am, bh) ISTNET_P(ah, bm) ISTNET_P(ah, bh)
#undef ISTNET_P
    }
  };
  int ky = (c_first / nb) / g.KW, kx = (c_first / nb) % g.KW, cb = c_first % nb;
  auto advance = [&]() {
    if (++cb == nb) {
      cb = 0;
      if (++kx == g.KW) { kx = 0; ++ky; }
      if (ky == g.KH) { ky = g.KH - 1; kx = g.KW - 1; cb = nb - 1; }
      else set_tap(ky, kx);
    }
  };
  set_tap(ky, kx);
  // Two stages of registers in flight (a chunk's loads are issued two MFMA phases before they are split and written to LDS;
  // four stages measured no gain: not bound by load latency), LDS double buffered, one barrier per chunk.  Past the end the
  // prefetch repeats the last chunk (loaded, never read).
  // (Interleaving mma(buf) with the independent commit(next chunk -> buf ^ 1) through sched_group_barrier hints, four
  // register stages, and four waves per workgroup all measured no gain or a loss: tools/exp/split_precision/.  The kernel is
  // bound by the SUM of its non-MFMA work -- LDS reads 178, split + LDS writes 133, global loads 117 of 532 us on layer4 3x3,
  // against 209 us of matrix-pipe time -- which the waves of ONE workgroup, all in the same phase, cannot overlap; with K chunks
  // of 16 two workgroups fit a CU and their phases interleave: 532 -> 476 us.)
  Stage sa, sb;
  issue(sa, ky, kx, cb); advance();
  commit(sa, 0);
  issue(sa, ky, kx, cb); advance();       // chunk 1
  __syncthreads();
  auto body = [&](Stage& snew, Stage& scommit, int buf) {
    issue(snew, ky, kx, cb); advance();
    __builtin_amdgcn_sched_barrier(0);
    mma(buf);
    __builtin_amdgcn_sched_barrier(0);
    commit(scommit, buf ^ 1);
    __syncthreads();
  };
  for (int c = 0; c < nchunks; c += 2) {
    body(sb, sa, 0);                      // chunk c from buffer 0; chunk c + 1 -> buffer 1; loads of chunk c + 2
    if (c + 1 >= nchunks) break;
    body(sa, sb, 1);
  }
  const long long m_first = (long long)tile_m_first * MT;
  float* dst = nsplits == 1 ? c_dst : ws + ((size_t)split * (M - m_first) - m_first) * Ncols;
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long row = m0 + (wm * TM + mi) * 32 + mfma_row(r, lane);
      if (row < M) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) dst[(size_t)row * Ncols + n0 + (wn * TN + ni) * 32 + l31] = acc[mi][ni][r];
      }
    }
}

// Backward-data at stride 1 IS a forward convolution of dout with the weights transposed and rotated by 180 degrees:
//   din[b, y, x, ci] = sum_{ky', kx', co} dout[b, y - (KH - 1 - pad) + ky', x - (KW - 1 - pad) + kx', co] w'[ci][ky'][kx'][co],
//   w'[ci][ky'][kx'][co] = w[co][KH - 1 - ky'][KW - 1 - kx'][ci]
// so the split-precision forward kernel serves it once the weights are laid out that way (one small pass per call).
__global__ __launch_bounds__(kThreads) void rotate_weights_kernel(int cout, int cin, int kh, int kw, const float* __restrict__ w,
                                                                  float* __restrict__ wt) {
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;      // index into wt [cin][kh][kw][cout]
  const long long n = (long long)cout * cin * kh * kw;
  if (i >= n) return;
  const int co = (int)(i % cout);
  long long r = i / cout;
  const int kx = (int)(r % kw); r /= kw;
  const int ky = (int)(r % kh);
  const int ci = (int)(r / kh);
  wt[i] = w[(((size_t)co * kh + (kh - 1 - ky)) * kw + (kw - 1 - kx)) * cin + ci];
}

// backward-weights: part[split][co][tap][ci] = sum over this split's output pixels of dout[p][co] * in[src(p, tap)][ci]
// grid (splits, co tiles x ci tiles, taps)
template <int MT, int NT, int WM, int WN, int STRIDE>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 2) void conv_wrw_kernel(ConvGeom g, const float* __restrict__ in,
                                                              const float* __restrict__ dout, float* __restrict__ part,
                                                              int chunks_per_split, int nsplits) {
  constexpr int TM = MT / (32 * WM), TN = NT / (32 * WN);
  constexpr int NTHR = 64 * WM * WN;
  constexpr int AR = KC * MT / 4 / NTHR, BR = KC * NT / 4 / NTHR;     // float4 per thread of the [KC][MT] / [KC][NT] tiles
  constexpr int LDA = MT + 4, LDB = NT + 4;
  constexpr int A_STAGE = KC * LDA, B_STAGE = KC * LDB;
  static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1 && AR >= 1 && AR <= 4 && BR >= 1 && BR <= 4, "4 or 8 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wv / WN, wn = wv % WN;
  // workgroup id -> (XCD x = id % 8, j = id / 8).  The taps of one (split, output tile) read the same dout rows and the same
  // input rows (shifted): they are consecutive workgroups of ONE XCD, so those rows come into that L2 once for the nine of
  // them (in launch order the hardware deals the nine to eight different L2s).  Groups (split, tile) are dealt to the XCDs
  // round-robin: 7 splits x 16 tiles = 112 groups = 14 per XCD.
  const int taps = g.KH * g.KW;
  const int ntn = g.Cin / NT, ntiles = (g.Cout / MT) * ntn;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int group = (jj / taps) * 8 + xcd, tap = jj % taps;
  if (group >= nsplits * ntiles) return;
  const int split = group / ntiles, tile = group % ntiles;
  const int ky = tap / g.KW, kx = tap - ky * g.KW;
  const int co0 = (tile / ntn) * MT, ci0 = (tile % ntn) * NT;
  const long long P = (long long)g.B * g.OH * g.OW;
  const long long nchunks_all = P / KC;
  const long long c_begin = (long long)split * chunks_per_split;
  long long c_end = c_begin + chunks_per_split;
  if (c_end > nchunks_all) c_end = nchunks_all;
  // Source-pixel table of this workgroup's pixel range for its tap: tab[j] = input pixel index (b * H + sy) * W + sx of output
  // pixel c_begin * 32 + j, or -1 outside the image.  Built once (a pixel's row / column from its index inside the image --
  // the host requires OH * OW % 32 == 0 and < 2^16 -- by one multiply-shift and a correction); the loop then spends one LDS
  // read per row instead of ~28 VALU instructions, which at four rows per thread and chunk were a fifth of the MFMA time.
  const int ohw = g.OH * g.OW;
  int* tab = reinterpret_cast<int*>(smem + 2 * A_STAGE + 2 * B_STAGE);
  {
    const unsigned magic = 65536u / (unsigned)g.OW;        // floor: the quotient is never too large, at most one too small
    const long long p_first = c_begin * KC;
    const int npix = (int)((c_end > c_begin ? c_end - c_begin : 0) * KC);
    for (int j = tid; j < npix; j += NTHR) {
      const long long p = p_first + j;
      const int b = (int)(p / ohw);
      const int q = (int)(p - (long long)b * ohw);
      int oy = (int)(((unsigned)q * magic) >> 16);
      int ox = q - oy * g.OW;
      if (ox >= g.OW) { ox -= g.OW; ++oy; }
      const int sy = oy * STRIDE + ky - g.pad, sx = ox * STRIDE + kx - g.pad;
      tab[j] = (sy >= 0 && sy < g.H && sx >= 0 && sx < g.W) ? (b * g.H + sy) * g.W + sx : -1;
    }
  }
  __syncthreads();

  // issue() is called for consecutive chunks (c_begin, c_begin + 1, ...)
  long long ic_next = c_begin;
  auto issue = [&](Stage& s, long long c) {
    (void)c;
    const bool live = ic_next < c_end;                      // past the end: the last chunk again (committed as zeros, never used)
    const long long cc = live ? ic_next : c_end - 1;
    ++ic_next;
    const int j0 = (int)(cc - c_begin) * KC;
    const size_t orow = (size_t)cc * KC * g.Cout;
    s.ok = live ? 0xffffffffu : 0u;
    static_for<AR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int e = tid + NTHR * i;
      const int k = e / (MT / 4), c4 = e % (MT / 4);
      st_a<i>(s) = *reinterpret_cast<const float4*>(dout + orow + (size_t)k * g.Cout + co0 + 4 * c4);
    });
    static_for<BR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int e = tid + NTHR * i;
      const int k = e / (NT / 4), c4 = e % (NT / 4);
      const int t = tab[j0 + k];
      st_b<i>(s) = *reinterpret_cast<const float4*>(in + (size_t)(t < 0 ? 0 : t) * g.Cin + ci0 + 4 * c4);
      if (t < 0) s.ok &= ~(1u << (8 + i));
    });
  };
  auto commit = [&](Stage& s, int buf) {
    float* as = As + buf * A_STAGE;
    float* bs = Bs + buf * B_STAGE;
    static_for<AR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int e = tid + NTHR * i;
      *reinterpret_cast<float4*>(as + (e / (MT / 4)) * LDA + 4 * (e % (MT / 4))) = keep_if((s.ok >> i) & 1u, st_a<i>(s));   // (storing the loaded value unmasked measured 15 % SLOWER: 892 vs 772 us at 512 -> 512)
    });
    static_for<BR>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int e = tid + NTHR * i;
      *reinterpret_cast<float4*>(bs + (e / (NT / 4)) * LDB + 4 * (e % (NT / 4))) = keep_if((s.ok >> (8 + i)) & 1u, st_b<i>(s));
    });
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  auto mma = [&](int buf) {
    const float* as = As + buf * A_STAGE + (4 * half) * LDA + (wm * TM) * 32 + l31;
    const float* bs = Bs + buf * B_STAGE + (4 * half) * LDB + (wn * TN) * 32 + l31;
#pragma unroll
    for (int g8 = 0; g8 < KC / 8; ++g8) {
      float af[TM][4], bf[TN][4];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int t = 0; t < 4; ++t) af[mi][t] = as[(8 * g8 + t) * LDA + mi * 32];
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int t = 0; t < 4; ++t) bf[ni][t] = bs[(8 * g8 + t) * LDB + ni * 32];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][t], bf[ni][t], acc[mi][ni], 0, 0, 0);
    }
  };
  Stage sa, sb;                        // two chunks of loads in flight, as in conv_igemm_kernel
  if (c_begin < c_end) {
    issue(sa, c_begin);
    commit(sa, 0);
    if constexpr (WM * WN == 4) {
      issue(sa, c_begin + 1);          // past the end: clamped rows, masked off when committed (and never used)
      __syncthreads();
      for (long long c = c_begin; c < c_end; c += 2) {
        issue(sb, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(0);
        __builtin_amdgcn_sched_barrier(0);
        commit(sa, 1);
        __syncthreads();
        if (c + 1 >= c_end) break;
        issue(sa, c + 3);
        __builtin_amdgcn_sched_barrier(0);
        mma(1);
        __builtin_amdgcn_sched_barrier(0);
        commit(sb, 0);
        __syncthreads();
      }
    } else {
      __syncthreads();
      int buf = 0;
      for (long long c = c_begin; c < c_end; ++c) {
        issue(sa, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(buf);
        __builtin_amdgcn_sched_barrier(0);
        commit(sa, buf ^ 1);
        __syncthreads();
        buf ^= 1;
      }
    }
  }
  float* dst = part + (size_t)split * g.Cout * taps * g.Cin;
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + (wm * TM + mi) * 32 + mfma_row(r, lane);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
        dst[((size_t)co * taps + tap) * g.Cin + ci0 + (wn * TN + ni) * 32 + l31] = acc[mi][ni][r];
    }
}

// dw[i] = sum over the splits of part[s][i], four elements per thread, four chains in flight, fixed order
__global__ __launch_bounds__(kThreads) void conv_wrw_reduce_kernel(long long n4, int splits, const float4* __restrict__ part,
                                                                   float4* __restrict__ dw) {
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n4) return;
  float4 s[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) s[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (; k + 4 <= splits; k += 4) {
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = part[(size_t)(k + j) * n4 + i];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s[j].x += v[j].x; s[j].y += v[j].y; s[j].z += v[j].z; s[j].w += v[j].w; }
  }
  for (; k < splits; ++k) {
    const float4 v = part[(size_t)k * n4 + i];
    s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
  }
  dw[i] = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                      (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
}

inline bool geom_ok(int b, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
  return b > 0 && h > 0 && w > 0 && (long long)b * h * w < (1ll << 24) && istnet_conv_supported(cin, cout, kh, kw, stride, pad) && h + 2 * pad >= kh && w + 2 * pad >= kw;
}
inline ConvGeom make_geom(int b, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
  ConvGeom g;
  g.B = b; g.H = h; g.W = w; g.Cin = cin; g.Cout = cout; g.KH = kh; g.KW = kw; g.stride = stride; g.pad = pad;
  g.OH = (h + 2 * pad - kh) / stride + 1;
  g.OW = (w + 2 * pad - kw) / stride + 1;
  return g;
}
int g_conv_split = 0;      // istnet_conv_set_tuning(1, v): forward kernel arithmetic, 0 fp32 MFMA, 1..3 split-precision variants
template <int MT, int NT, int MODE>
constexpr size_t igemm_lds() {
  return (size_t)(2 * MT * LDK + 2 * (MODE == 0 ? NT * LDK : KC * (NT + 4))) * sizeof(float);
}
constexpr int kWrwMaxChunks = 96;      // chunks per split at most: the source-pixel table of a split is 4 * 32 * 96 = 12 KB of LDS
template <int MT, int NT>
constexpr size_t wrw_lds() { return (size_t)(2 * KC * (MT + 4) + 2 * KC * (NT + 4)) * sizeof(float); }

// more than 64 KB of dynamic LDS has to be granted once per kernel (the static flag is per expansion site)
// (the attribute is per DEVICE: a process that drives several GPUs needs it on each, so the flag is a per-device bit)
#define ISTNET_ALLOW_LDS(KERNEL, BYTES)                                                                              \
  do {                                                                                                               \
    static std::atomic<unsigned long long> done_{0};                                                                 \
    if ((BYTES) > 64 * 1024) {                                                                                       \
      int dev_ = 0;                                                                                                  \
      (void)hipGetDevice(&dev_);                                                                                     \
      const unsigned long long bit_ = 1ull << (dev_ & 63);                                                           \
      if (!(done_.load(std::memory_order_relaxed) & bit_)) {                                                         \
        auto k_ = KERNEL;                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_), hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                  (int)(BYTES));                                                                     \
        done_.fetch_or(bit_, std::memory_order_relaxed);                                                             \
      }                                                                                                              \
    }                                                                                                                \
  } while (0)

}  // namespace

extern "C" {

int istnet_conv_set_tuning(int key, int value) {
  // key 1: forward kernel arithmetic -- 0 exact fp32 matrix cores (default); 1 split precision (three bf16 terms per operand,
  // six bf16 MFMA products, fp32 accumulation: fp32-class accuracy, opt-in experiment); 2, 3: its A/B variants
  if (key == 1) { if (value < 0 || value > 7) return ISTNET_PN2_EINVAL; g_conv_split = value; return 0; }
  return ISTNET_PN2_EINVAL;
}
int istnet_conv_get_tuning(int key) { return key == 1 ? g_conv_split : -1; }

int istnet_conv_supported(int cin, int cout, int kh, int kw, int stride, int pad) {
  return cin > 0 && cout > 0 && cin % 64 == 0 && cout % 64 == 0 && kh >= 1 && kh <= 7 && kh == kw &&
         (stride == 1 || stride == 2) && pad >= 0 && pad <= kh / 2;
}

// How an implicit-GEMM product is launched.  The chip has 512 workgroup slots (two per CU); `tiles` output tiles take
// ceil(tiles / 512) rounds and the last round may be nearly empty (576 tiles: the second round is an eighth full).  So: the
// row tiles that fill whole rounds run unsplit (part A); the remaining row tiles run with K split over `splits` workgroups
// each (part B: 64 tiles x 8 = one full round of an eighth of the length), every split accumulating into its own slab of
// the work space, summed afterwards in a fixed order.  A product with less than one full round is all part B.
struct IgemmPlan {
  int nt;                 // column tile
  int rows_a;             // row tiles of part A (unsplit), a multiple of 8
  int rows_b, splits, per;  // part B: row tiles, K splits, chunks per split
};
static IgemmPlan igemm_plan(const ConvGeom& g, int mode) {
  IgemmPlan p;
  const int ncols = mode == 0 ? g.Cout : g.Cin, kch = mode == 0 ? g.Cin : g.Cout;
  const long long m = mode == 0 ? (long long)g.B * g.OH * g.OW : (long long)g.B * g.H * g.W;
  const int nchunks = g.KH * g.KW * (kch / KC);
  const int row_tiles = (int)((m + 127) / 128);
  p.nt = ncols % 128 == 0 ? 128 : 64;
  if (p.nt == 128 && (long long)row_tiles * (ncols / 128) < 512 && ncols % 64 == 0) p.nt = 64;   // more, smaller tiles
  const int tiles_n = ncols / p.nt;
  const long long tiles = (long long)row_tiles * tiles_n;
  p.rows_a = (int)((tiles / 512) * 512 / tiles_n) / 8 * 8;
  p.rows_b = row_tiles - p.rows_a;
  p.splits = 1;
  if (p.rows_b > 0) {
    const long long tb = (long long)p.rows_b * tiles_n;
    int sp = (int)(512 / tb);                               // fill one round
    if (sp > 16) sp = 16;
    while (sp > 1 && nchunks / sp < 6) --sp;
    if (sp < 1) sp = 1;
    // a nearly full round is better left unsplit (the slabs are not free)
    if (tb * 10 >= 512 * 8) sp = 1;
    p.splits = sp;
  }
  p.per = (nchunks + p.splits - 1) / p.splits;
  p.splits = (nchunks + p.per - 1) / p.per;
  return p;
}

// variants: 1 = K chunks of 16 (73 KB of LDS, 128 VGPRs: two workgroups per CU, whose phases interleave) -- the default of
// "split precision"; 2 = K chunks of 32, one workgroup per CU (A/B).  128 x 64 tiles always take the K-32 form.
#define ISTNET_IGEMM_SPLIT_ONE(NT, STRIDE, KCS, MINW, PERMUL)                                                          \
  do {                                                                                                                 \
    constexpr size_t lds_ = (size_t)2 * 3 * (128 + NT) * (KCS + 8) * sizeof(unsigned short);                           \
    ISTNET_ALLOW_LDS((conv_igemm_split_kernel<NT, STRIDE, 4, KCS, MINW>), lds_);                                       \
    hipLaunchKernelGGL((conv_igemm_split_kernel<NT, STRIDE, 4, KCS, MINW>), dim3(grid), dim3(512), lds_,               \
                       (hipStream_t)stream, g, a, wgt, c, ws, (PERMUL) * per, splits, first, count);                   \
  } while (0)
#define ISTNET_IGEMM_SPLIT(NT, STRIDE)                                                                                 \
  do {                                                                                                                 \
    if (g_conv_split == 1 && NT == 128) ISTNET_IGEMM_SPLIT_ONE(128, STRIDE, 16, 4, 2);                                 \
    else ISTNET_IGEMM_SPLIT_ONE(NT, STRIDE, KC, 1, 1);                                                                 \
  } while (0)

#define ISTNET_IGEMM(MT, NT, WM, WN, MODE, STRIDE)                                                                     \
  do {                                                                                                                 \
    ISTNET_ALLOW_LDS((conv_igemm_kernel<MT, NT, WM, WN, MODE, STRIDE>), (igemm_lds<MT, NT, MODE>()));                 \
    hipLaunchKernelGGL((conv_igemm_kernel<MT, NT, WM, WN, MODE, STRIDE>), dim3(grid), dim3(64 * WM * WN),             \
                       (igemm_lds<MT, NT, MODE>()), (hipStream_t)stream, g, a, wgt, c, ws, per, splits, first, count); \
  } while (0)

static void igemm_part(const ConvGeom& g, int mode, int nt, int first, int count, int splits, int per, const float* a,
                       const float* wgt, float* c, float* ws, void* stream) {
  const int ncols = mode == 0 ? g.Cout : g.Cin;
  const unsigned grid = (unsigned)((count + 7) / 8 * 8 * (ncols / nt) * splits);
  if (mode == 0 && g_conv_split) {
    if (nt == 128) { if (g.stride == 1) ISTNET_IGEMM_SPLIT(128, 1); else ISTNET_IGEMM_SPLIT(128, 2); }
    else { if (g.stride == 1) ISTNET_IGEMM_SPLIT(64, 1); else ISTNET_IGEMM_SPLIT(64, 2); }
  } else if (mode == 0) {
    if (nt == 128) { if (g.stride == 1) ISTNET_IGEMM(128, 128, 2, 2, 0, 1); else ISTNET_IGEMM(128, 128, 2, 2, 0, 2); }
    else { if (g.stride == 1) ISTNET_IGEMM(128, 64, 4, 1, 0, 1); else ISTNET_IGEMM(128, 64, 4, 1, 0, 2); }
  } else {
    if (nt == 128) { if (g.stride == 1) ISTNET_IGEMM(128, 128, 2, 2, 1, 1); else ISTNET_IGEMM(128, 128, 2, 2, 1, 2); }
    else { if (g.stride == 1) ISTNET_IGEMM(128, 64, 4, 1, 1, 1); else ISTNET_IGEMM(128, 64, 4, 1, 1, 2); }
  }
}
#undef ISTNET_IGEMM
#undef ISTNET_IGEMM_SPLIT
#undef ISTNET_IGEMM_SPLIT_ONE

// floats of the K-split slabs of a plan (0: none)
static long long plan_slab_floats(const ConvGeom& g, int mode, const IgemmPlan& plan) {
  if (plan.splits == 1) return 0;
  const long long m = mode == 0 ? (long long)g.B * g.OH * g.OW : (long long)g.B * g.H * g.W;
  return (long long)plan.splits * (m - (long long)plan.rows_a * 128) * (mode == 0 ? g.Cout : g.Cin);
}
static int igemm_launch(const ConvGeom& g, int mode, const float* a, const float* wgt, float* c, float* ws, void* stream) {
  const IgemmPlan plan = igemm_plan(g, mode);
  if (plan.splits > 1 && ws == nullptr) return ISTNET_PN2_EINVAL;
  const int ncols = mode == 0 ? g.Cout : g.Cin, kch = mode == 0 ? g.Cin : g.Cout;
  const long long m = mode == 0 ? (long long)g.B * g.OH * g.OW : (long long)g.B * g.H * g.W;
  const int nchunks = g.KH * g.KW * (kch / KC);
  if (plan.rows_a > 0) igemm_part(g, mode, plan.nt, 0, plan.rows_a, 1, nchunks, a, wgt, c, ws, stream);
  if (plan.rows_b > 0) {
    igemm_part(g, mode, plan.nt, plan.rows_a, plan.rows_b, plan.splits, plan.per, a, wgt, c, ws, stream);
    if (plan.splits > 1) {
      const long long m_first = (long long)plan.rows_a * 128;
      const long long n4 = (m - m_first) * ncols / 4;
      hipLaunchKernelGGL(conv_wrw_reduce_kernel, dim3((unsigned)((n4 + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                         (hipStream_t)stream, n4, plan.splits, reinterpret_cast<const float4*>(ws),
                         reinterpret_cast<float4*>(c + (size_t)m_first * ncols));
    }
  }
  return (int)hipGetLastError();
}

int istnet_conv_workspace_floats(int backward_data, int b, int h, int w, int cin, int cout, int kh, int kw, int stride,
                                       int pad) {
  if (!geom_ok(b, h, w, cin, cout, kh, kw, stride, pad)) return -1;
  const ConvGeom g = make_geom(b, h, w, cin, cout, kh, kw, stride, pad);
  const IgemmPlan plan = igemm_plan(g, backward_data ? 1 : 0);
  long long n = plan_slab_floats(g, backward_data ? 1 : 0, plan);
  if (backward_data && g_conv_split && stride == 1) {
    // split precision: backward-data runs as a forward product of dout with the rotated weights (rotated copy + that plan's slabs)
    const ConvGeom gt = make_geom(b, g.OH, g.OW, cout, cin, kh, kw, 1, kh - 1 - pad);
    n = plan_slab_floats(gt, 0, igemm_plan(gt, 0)) + (long long)cout * cin * kh * kw;
  }
  return n < (1ll << 31) ? (int)n : -1;
}

int istnet_conv_forward(int b, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, const float* in,
                        const float* wgt, float* out, float* ws, void* stream) {
  if (!geom_ok(b, h, w, cin, cout, kh, kw, stride, pad) || !in || !wgt || !out ||
      (((uintptr_t)in | (uintptr_t)wgt | (uintptr_t)out | (uintptr_t)ws) & 15))
    return ISTNET_PN2_EINVAL;
  return igemm_launch(make_geom(b, h, w, cin, cout, kh, kw, stride, pad), 0, in, wgt, out, ws, stream);
}

int istnet_conv_backward_data(int b, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, const float* dout,
                              const float* wgt, float* din, float* ws, void* stream) {
  if (!geom_ok(b, h, w, cin, cout, kh, kw, stride, pad) || !dout || !wgt || !din ||
      (((uintptr_t)dout | (uintptr_t)wgt | (uintptr_t)din | (uintptr_t)ws) & 15))
    return ISTNET_PN2_EINVAL;
  const ConvGeom g = make_geom(b, h, w, cin, cout, kh, kw, stride, pad);
  if (g_conv_split && stride == 1 && kh - 1 - pad >= 0) {
    if (ws == nullptr) return ISTNET_PN2_EINVAL;
    const ConvGeom gt = make_geom(b, g.OH, g.OW, cout, cin, kh, kw, 1, kh - 1 - pad);     // its output is (b, h, w, cin)
    if (gt.OH != h || gt.OW != w) return ISTNET_PN2_EINVAL;
    const long long nw = (long long)cout * cin * kh * kw;
    const long long slabs = plan_slab_floats(gt, 0, igemm_plan(gt, 0));
    float* wt = ws + slabs;                                                              // behind the forward launch's own work space
    hipLaunchKernelGGL(rotate_weights_kernel, dim3((unsigned)((nw + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, cout, cin, kh, kw, wgt, wt);
    return igemm_launch(gt, 0, dout, wt, din, ws, stream);
  }
  return igemm_launch(g, 1, dout, wgt, din, ws, stream);
}

static int wrw_plan(const ConvGeom& g, int& mt, int& nt, int& per) {
  mt = g.Cout % 128 == 0 ? 128 : 64;
  nt = g.Cin % 128 == 0 ? 128 : 64;
  const long long tiles = (long long)(g.Cout / mt) * (g.Cin / nt) * g.KH * g.KW;
  const long long chunks = ((long long)g.B * g.OH * g.OW + KC - 1) / KC;
  // the split count that fills whole rounds of the 512 workgroup slots best (a split costs one slab of dW: small)
  const double wbytes = (double)g.Cout * g.KH * g.KW * g.Cin * 4.0;
  const double t_full = 2.0 * (double)g.B * g.OH * g.OW * g.Cout * g.Cin * g.KH * g.KW / 130e12;
  long long best = 1;
  double best_t = 1e30;
  for (long long sp = (chunks + kWrwMaxChunks - 1) / kWrwMaxChunks; sp <= 1024 && (chunks / sp >= 6 || sp == (chunks + kWrwMaxChunks - 1) / kWrwMaxChunks); ++sp) {
    const long long units = tiles * sp, rounds = (units + 511) / 512;
    const double t = t_full * (double)(rounds * 512) / (double)units + (sp + 1) * wbytes / 3.0e12 + 2e-6 * rounds / sp;
    if (t < best_t) { best_t = t; best = sp; }
  }
  per = (int)((chunks + best - 1) / best);
  if (per > kWrwMaxChunks) return 0;      // unreachable behind wrw_ok(); the launch's LDS table is sized by `per`
  return (int)((chunks + per - 1) / per);
}

// the split search stops at 1024 splits; beyond 96 * 1024 chunks (3.1 M output pixels) no plan keeps a split's
// source-pixel table within kWrwMaxChunks, so such a layer is not taken (the caller keeps the framework's product)
static bool wrw_ok(const ConvGeom& g) {
  return (g.OH * g.OW) % KC == 0 && g.OH * g.OW < 65536 && g.OW >= 2 &&
         ((long long)g.B * g.OH * g.OW + KC - 1) / KC <= (long long)kWrwMaxChunks * 1024;
}

int istnet_conv_wrw_splits(int b, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
  if (!geom_ok(b, h, w, cin, cout, kh, kw, stride, pad) || !wrw_ok(make_geom(b, h, w, cin, cout, kh, kw, stride, pad))) return 0;
  int mt, nt, per;
  return wrw_plan(make_geom(b, h, w, cin, cout, kh, kw, stride, pad), mt, nt, per);
}

int istnet_conv_backward_weights(int b, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, const float* in,
                                 const float* dout, float* part, float* dwgt, void* stream) {
  if (!geom_ok(b, h, w, cin, cout, kh, kw, stride, pad) || !in || !dout || !part || !dwgt ||
      (((uintptr_t)in | (uintptr_t)dout | (uintptr_t)part | (uintptr_t)dwgt) & 15))
    return ISTNET_PN2_EINVAL;
  const ConvGeom g = make_geom(b, h, w, cin, cout, kh, kw, stride, pad);
  if (!wrw_ok(g)) return ISTNET_PN2_EINVAL;
  int mt, nt, per;
  const int splits = wrw_plan(g, mt, nt, per);
  if (splits <= 0) return ISTNET_PN2_EINVAL;
  const size_t tab_bytes = (size_t)per * KC * 4;
  const dim3 grid((unsigned)((splits * (cout / mt) * (cin / nt) + 7) / 8 * 8 * kh * kw));
#define ISTNET_WRW(MT, NT, WM, WN)                                                                                       \
  do {                                                                                                                   \
    if (stride == 1) {                                                                                                   \
      ISTNET_ALLOW_LDS((conv_wrw_kernel<MT, NT, WM, WN, 1>), (wrw_lds<MT, NT>() + (size_t)kWrwMaxChunks * KC * 4));        \
      hipLaunchKernelGGL((conv_wrw_kernel<MT, NT, WM, WN, 1>), grid, dim3(64 * WM * WN), (wrw_lds<MT, NT>() + tab_bytes), \
                         (hipStream_t)stream, g, in, dout, part, per, splits);                                          \
    } else {                                                                                                             \
      ISTNET_ALLOW_LDS((conv_wrw_kernel<MT, NT, WM, WN, 2>), (wrw_lds<MT, NT>() + (size_t)kWrwMaxChunks * KC * 4));        \
      hipLaunchKernelGGL((conv_wrw_kernel<MT, NT, WM, WN, 2>), grid, dim3(64 * WM * WN), (wrw_lds<MT, NT>() + tab_bytes), \
                         (hipStream_t)stream, g, in, dout, part, per, splits);                                          \
    }                                                                                                                    \
  } while (0)
  if (mt == 128 && nt == 128) ISTNET_WRW(128, 128, 2, 2);
  else if (mt == 128) ISTNET_WRW(128, 64, 4, 1);
  else if (nt == 128) ISTNET_WRW(64, 128, 1, 4);
  else ISTNET_WRW(64, 64, 2, 2);
#undef ISTNET_WRW
  const long long n4 = (long long)cout * kh * kw * cin / 4;
  hipLaunchKernelGGL(conv_wrw_reduce_kernel, dim3((unsigned)((n4 + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, n4, splits, reinterpret_cast<const float4*>(part), reinterpret_cast<float4*>(dwgt));
  return (int)hipGetLastError();
}

}  // extern "C"
