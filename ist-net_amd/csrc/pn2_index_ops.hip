// pn2_index_ops.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the nine
// PointNet++ index / gather ops behind the reference's `pointnet2._ext`
// (reference: model/pointnet2/_ext_src/src/{sampling,ball_query,group_points,interpolate}_gpu.cu).
//
// Design (MI355X-first, not a translation of the reference launch shapes):
//   * the reference launches grid = B (one block per cloud, <=512 threads, long
//     serial loops).  Here every op is decomposed so a launch has >>256
//     workgroups (point tiles x channel chunks x clouds), except FPS which is
//     inherently sequential per cloud and is built for minimum per-round latency:
//     one wave per cloud (n <= 1024) keeps xyz + running min-distances in VGPRs,
//     does the arg-max with DPP cross-lane ops, and never touches a barrier.
//   * ball_query: one wave per centroid, xyz staged SoA in LDS, 64 points tested
//     per step, __ballot + mbcnt prefix count emits the first `nsample` hits in
//     index order (bit-identical to the serial scan) with early exit.
//   * three_nn: thread per unknown point, known set broadcast from LDS.
//   * group / interpolate: output-coalesced (float4 where possible), indices
//     loaded once per thread and reused across a channel chunk.
//   * group_grad / interpolate_grad / gather_grad: scatter-add into an LDS
//     accumulator per (cloud, channel) row, written back once -- no global
//     atomics, no pre-zeroed output needed.
//
// Arithmetic convention for index-deciding distances (shared with oracle/pn2_oracle.c):
// IEEE f32 in the source order of the reference expression.  Convention 0 (default): NO FMA contraction
// (file-wide pragma below + the -ffp-contract=off build flag).  Conventions 1 and 2 are the two ways a compiler
// with contraction on (nvcc's default -fmad=true) can fuse  dx*dx + dy*dy + dz*dz :
//     1:  fma(dz, dz, fma(dx, dx, dy*dy))        2:  fma(dz, dz, fma(dy, dy, dx*dx))
// selected at run time with istnet_pn2_set_tuning(1, c) -- explicit fmaf calls, so the build flags stay as they are.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/istnet_pn2.h"

#pragma clang fp contract(off)

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// d2 in the reference's source order ((dx*dx + dy*dy) + dz*dz): un-contracted (CONV 0) or with the products
// fused the way a contracting compiler would (CONV 1 / 2, see the header).
template <int CONV>
__device__ __forceinline__ float sqdist_d(float dx, float dy, float dz) {
  if (CONV == 1) return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
  if (CONV == 2) return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
  return (dx * dx + dy * dy) + dz * dz;
}
template <int CONV>
__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
  return sqdist_d<CONV>(ax - bx, ay - by, az - bz);
}

// Two points per instruction: v_pk_add_f32 / v_pk_mul_f32 round each half exactly like the scalar ops
// (IEEE RN, no fusion), so the packed distance is bit-identical to sqdist() and halves the VALU work.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int CONV>
__device__ __forceinline__ f32x2 sqdist2(f32x2 ax, f32x2 ay, f32x2 az, float bx, float by, float bz) {
  const f32x2 dx = ax - bx, dy = ay - by, dz = az - bz;
  if (CONV != 0) {
    f32x2 r;
    r[0] = sqdist_d<CONV>(dx[0], dy[0], dz[0]);
    r[1] = sqdist_d<CONV>(dx[1], dy[1], dz[1]);
    return r;
  }
  return (dx * dx + dy * dy) + dz * dz;
}

// ---- wave64 DPP reductions (result uniform, taken from lane 63) -------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ unsigned dpp_mov(unsigned identity, unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, dpp_mov<0x111>(0u, v));        // row_shr:1
  v = max(v, dpp_mov<0x112>(0u, v));        // row_shr:2
  v = max(v, dpp_mov<0x114>(0u, v));        // row_shr:4
  v = max(v, dpp_mov<0x118>(0u, v));        // row_shr:8  -> lane 15 of each row = row max
  v = max(v, dpp_mov<0x142, 0xa>(0u, v));   // row_bcast:15 into rows 1,3
  v = max(v, dpp_mov<0x143, 0xc>(0u, v));   // row_bcast:31 into rows 2,3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = min(v, dpp_mov<0x111>(0xffffffffu, v));
  v = min(v, dpp_mov<0x112>(0xffffffffu, v));
  v = min(v, dpp_mov<0x114>(0xffffffffu, v));
  v = min(v, dpp_mov<0x118>(0xffffffffu, v));
  v = min(v, dpp_mov<0x142, 0xa>(0xffffffffu, v));
  v = min(v, dpp_mov<0x143, 0xc>(0xffffffffu, v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() also waits for
// vmcnt(0), i.e. for the acknowledgement of the global stores of the previous FPS round (idxs[j], picked[j]) -- a full
// memory round trip per round in the multi-wave kernel.
__device__ __forceinline__ void lds_only_barrier() {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

// ============================================================================
// Furthest point sampling
// ============================================================================
// Tie order of the reference (sampling_gpu.cu:64-70,113-173): each of
// bs = opt_n_threads(n) threads keeps the LOWEST k of its maximum (strict '>'),
// and the shared-memory tree keeps the lower slot on ties, level offsets
// bs/2 ... 1.  The last level compares slot bit 0, the one before bit 1, ... so
// among equal distances the winner is the point with the smallest
//     tiekey(k) = bitrev_{log2 bs}(k mod bs) * nper + (k / bs),  nper = ceil(n / bs).
// We hand point slots to lanes in tiekey order (slot i of thread t <-> tiekey
// i*THREADS + t), so a strict '>' scan per lane plus a (value max, tiekey min)
// cross-lane reduction reproduces the reference winner exactly.
__device__ __forceinline__ int tiekey_to_k(unsigned tk, int bs_log2, int nper) {
  // nper is a power of two for every n <= 512 and for the usual 1024/2048/4096 clouds: shift, not divide
  unsigned slot_rev;
  if ((nper & (nper - 1)) == 0) slot_rev = tk >> (31 - __clz(nper));
  else slot_rev = tk / (unsigned)nper;
  const unsigned row = tk - slot_rev * (unsigned)nper;
  const unsigned slot = bs_log2 ? (__brev(slot_rev) >> (32 - bs_log2)) : 0u;
  return (int)(slot + (row << bs_log2));
}

// Chained sampling (the encoder samples level l+1 from level l's picks IN PICK ORDER, pointnet2_modules.py:49-58 applied
// level after level): round j of the child run maximises the running minimum distance over S = {p_0 .. p_{m_parent-1}},
// the parent's picks.  The parent's pick p_j maximised the same quantity -- same picked set by induction, bit-identical
// distances (same expression, same operands) -- over the full cloud X, a superset of S, and p_j is in S; so whenever the
// parent's maximum of round j was UNIQUE, the child's pick of round j is p_j, i.e. index j of its input.  TRACK makes a
// run report the first round whose maximum was attained by more than one point (`tie_out`, conservative for its
// children: a tie inside S is a tie inside X), and a run whose parent had no tie before round m (`tie_in[cloud] >= m`)
// writes the prefix 0 .. m-1 and returns.  Any tie (duplicate points, equal distances) falls back to the full scan.
constexpr int kFpsNoTie = 0x7fffffff;

template <bool V> struct fps_tag { static constexpr bool value = V; };
__device__ __forceinline__ int med3_i32(int a, int b, int c) {     // v_med3_i32
  return max(min(a, b), min(max(a, b), c));
}

template <int NW, int PPT, int CONV, bool TRACK>
__global__ __launch_bounds__(NW * 64) void fps_regs_kernel(int n, int m, int bs_log2, int nper,
                                                           const float* __restrict__ dataset_all,
                                                           int* __restrict__ idxs_all,
                                                           float* __restrict__ picked_all,
                                                           const int* __restrict__ tie_in,
                                                           int* __restrict__ tie_out, int track_rounds) {
  constexpr int THREADS = NW * 64;
  extern __shared__ __attribute__((aligned(16))) float fps_lds[];  // xyz AoS copy [3n] (+ exchange)
  // a dependent chain of one wave per cloud: when GEMM / convolution waves of other streams share the SIMD the chain is
  // served in turn with them (the level-1 pass measured 290 us alone, 890 us beside the RGB trunk); highest issue priority
  // costs the others nothing measurable (32 waves on 1 024 SIMDs)
  __builtin_amdgcn_s_setprio(3);
  const int cloud = blockIdx.x;
  const float* dataset = dataset_all + (size_t)cloud * n * 3;
  int* idxs = idxs_all + (size_t)cloud * m;
  float* picked = picked_all ? picked_all + (size_t)cloud * m * 3 : nullptr;  // optional: coordinates of the picks
  const int tid = threadIdx.x;

  if (TRACK && tie_in != nullptr) {
    const int parent_tie = tie_in[cloud];        // uniform: one cloud per workgroup
    if (parent_tie >= m && m <= n) {             // the parent's rounds 1 .. m-1 had unique maxima: picks = prefix
      for (int j = tid; j < m; j += THREADS) idxs[j] = j;
      if (picked != nullptr)
        for (int e = tid; e < 3 * m; e += THREADS) picked[e] = dataset[e];
      if (tid == 0 && tie_out != nullptr) tie_out[cloud] = parent_tie;
      return;
    }
  }

  for (int e = tid; e < 3 * n; e += THREADS) fps_lds[e] = dataset[e];
  unsigned* xchg = reinterpret_cast<unsigned*>(fps_lds + 3 * n);  // [2][NW][4]
  __syncthreads();

  constexpr int PP = (PPT + 1) / 2;  // point slots are held in pairs (packed f32 math)
  f32x2 px[PP], py[PP], pz[PP];
  // Running minimum distances as the BITS of the f32 value: squared distances are >= +0, and for non-negative floats
  // the signed-integer order of the bits is the float order, so min / max are v_min_i32 / v_max3_i32 -- exact, and
  // without the canonicalising v_max_f32 x,x the compiler must put in front of every loop-carried fminf operand.
  // An invalid slot holds -1: min(d, -1) = -1 stays invalid and never beats best = -1.  (A NaN distance has positive
  // bits above every finite value: min keeps the old value, as fminf and the reference's min() do.)
  int tmp[2 * PP];
#pragma unroll
  for (int i = 0; i < 2 * PP; ++i) {
    const unsigned tk = (unsigned)(i * THREADS + tid);
    const int k = tiekey_to_k(tk, bs_log2, nper);
    const bool valid = i < PPT && tk < ((unsigned)nper << bs_log2) && k < n;
    px[i / 2][i % 2] = valid ? fps_lds[3 * k + 0] : 0.f;
    py[i / 2][i % 2] = valid ? fps_lds[3 * k + 1] : 0.f;
    pz[i / 2][i % 2] = valid ? fps_lds[3 * k + 2] : 0.f;
    tmp[i] = valid ? __float_as_int(1e10f) : -1;
  }

  int old = 0;
  int first_tie = kFpsNoTie;
  if (tid == 0) idxs[0] = 0;
  if (picked != nullptr && tid < 3) picked[tid] = fps_lds[tid];
  // one round; `track` (a compile-time tag) adds the uniqueness test of the maximum: per pair of slots the running
  // SECOND largest value of the lane (v_med3 + v_max: the second largest of {lane's values so far} is
  // max(old second, median(old best, a, c))), and after the reduction "more than one lane holds the maximum, or the
  // lane that holds it holds it twice"
  auto round = [&](auto track, int j) {
    constexpr bool TR = decltype(track)::value;
    const float x1 = fps_lds[3 * old + 0];
    const float y1 = fps_lds[3 * old + 1];
    const float z1 = fps_lds[3 * old + 2];
    // Per PAIR of slots: two integer minima, one v_max3 for the running maximum, and the index bookkeeping once per
    // pair (which of the two is larger -- the lower slot on a tie -- and whether the pair beat the maximum strictly,
    // so a lane keeps its LOWEST slot among equals): 7 instructions per pair besides the distances.
    int best = -1;
    int besti = 0;
    int second = -1;
#pragma unroll
    for (int i = 0; i < PP; ++i) {
      const f32x2 d = sqdist2<CONV>(px[i], py[i], pz[i], x1, y1, z1);
      const int a = min(__float_as_int(d[0]), tmp[2 * i]);
      tmp[2 * i] = a;
      if (2 * i + 1 < PPT) {
        const int c = min(__float_as_int(d[1]), tmp[2 * i + 1]);
        tmp[2 * i + 1] = c;
        const int code = c > a ? 2 * i + 1 : 2 * i;
        if (TR) second = max(second, med3_i32(best, a, c));
        const int nb = max(best, max(a, c));
        besti = nb > best ? code : besti;
        best = nb;
      } else {
        if (TR) second = max(second, min(best, a));
        besti = a > best ? 2 * i : besti;
        best = max(best, a);
      }
    }
    // value key: 0 for "no valid slot", otherwise float bits + 1 (d2 >= +0 => monotone)
    const unsigned vkey = (unsigned)(best + 1);
    unsigned vmax = wave_max_u32(vkey);
    const bool holds = vkey == vmax;
    const unsigned tk = holds ? (unsigned)(besti * THREADS + tid) : 0xffffffffu;
    unsigned tkmin = wave_min_u32(tk);
    bool tied = false;
    if (TR) {
      const unsigned long long hm = __ballot(holds);
      const unsigned long long twice = __ballot(holds && second == best);
      tied = vmax != 0u && ((hm & (hm - 1)) != 0ull || twice != 0ull);
    }
    if (NW > 1) {
      // The NW waves' (maximum, tie key, tie flag) records meet in LDS.  Round 6: combined in VECTOR form -- every lane takes the
      // record of wave (lane mod NW) and log2(NW) DPP exchanges inside each group of NW lanes leave the workgroup's result in
      // all of them (the combination is commutative and associative: largest value, smallest tie key among equals, number of
      // waves holding the maximum, OR of their flags).  The scalar loop over the records it replaces was ~100 of the round's
      // 240 instructions (v_readfirstlane + s_cmp / s_cselect chains per record).
      unsigned* slot = xchg + (j & 1) * (NW * 4);
      if (lane_id() == 0) {
        slot[(tid >> 6) * 4 + 0] = vmax;
        slot[(tid >> 6) * 4 + 1] = tkmin;
        if (TR) slot[(tid >> 6) * 4 + 2] = tied ? 1u : 0u;
      }
      lds_only_barrier();
      const unsigned* rec = slot + (lane_id() & (NW - 1)) * 4;
      unsigned bv = rec[0], bt = rec[1], bf = TR ? rec[2] : 0u, nbest = 1u;
#pragma unroll
      for (int step = 1; step < NW; step <<= 1) {
        // partner: lane ^ 1, lane ^ 2 (quad permutes), then the other quad of the lane's group of eight (half-row mirror:
        // after two steps the four lanes of a quad agree, so any lane of the other quad will do)
        unsigned pv, pt, pf = 0u, pn = 0u;
        if (step == 1) {
          pv = dpp_mov<0xB1>(0u, bv); pt = dpp_mov<0xB1>(0u, bt);
          if (TR) { pf = dpp_mov<0xB1>(0u, bf); pn = dpp_mov<0xB1>(0u, nbest); }
        } else if (step == 2) {
          pv = dpp_mov<0x4E>(0u, bv); pt = dpp_mov<0x4E>(0u, bt);
          if (TR) { pf = dpp_mov<0x4E>(0u, bf); pn = dpp_mov<0x4E>(0u, nbest); }
        } else {
          pv = dpp_mov<0x141>(0u, bv); pt = dpp_mov<0x141>(0u, bt);
          if (TR) { pf = dpp_mov<0x141>(0u, bf); pn = dpp_mov<0x141>(0u, nbest); }
        }
        const bool gt = pv > bv, eq = pv == bv;
        const bool better = gt || (eq && pt < bt);
        if (TR) {
          nbest = eq ? nbest + pn : (gt ? pn : nbest);
          bf = eq ? (bf | pf) : (gt ? pf : bf);
        }
        bv = better ? pv : bv;
        bt = better ? pt : bt;
      }
      tkmin = (unsigned)__builtin_amdgcn_readfirstlane((int)bt);
      if (TR) {
        const unsigned rv = (unsigned)__builtin_amdgcn_readfirstlane((int)bv);
        const unsigned rn = (unsigned)__builtin_amdgcn_readfirstlane((int)nbest);
        const unsigned rf = (unsigned)__builtin_amdgcn_readfirstlane((int)bf);
        tied = rv != 0u && (rn > 1u || rf != 0u);
      }
    }
    if (TR && tied && first_tie == kFpsNoTie) first_tie = j;
    old = tiekey_to_k(tkmin, bs_log2, nper);
    if (tid == 0) idxs[j] = old;
    if (picked != nullptr && tid < 3) picked[3 * j + tid] = fps_lds[3 * old + tid];
  };
  int j = 1;
  if (TRACK) {
    const int upto = min(m, track_rounds);      // children sample at most this many points
    for (; j < upto; ++j) round(fps_tag<true>{}, j);
  }
  for (; j < m; ++j) round(fps_tag<false>{}, j);
  // "no tie before round tie_out": rounds >= track_rounds were not examined
  if (TRACK && tid == 0 && tie_out != nullptr) tie_out[cloud] = min(first_tie, track_rounds >= m ? kFpsNoTie : track_rounds);
}

// Generic fallback for very large clouds (n > 4096): running distances live in
// `temp` (global), xyz read from global/L2.  Same winner rule.
template <int CONV>
__global__ __launch_bounds__(1024) void fps_generic_kernel(int n, int m, int bs_log2, int nper,
                                                           const float* __restrict__ dataset_all,
                                                           float* __restrict__ temp_all,
                                                           int* __restrict__ idxs_all) {
  constexpr int THREADS = 1024, NW = 16;
  __shared__ unsigned xchg[2][NW][2];
  const int cloud = blockIdx.x;
  const float* dataset = dataset_all + (size_t)cloud * n * 3;
  float* temp = temp_all + (size_t)cloud * n;
  int* idxs = idxs_all + (size_t)cloud * m;
  const int tid = threadIdx.x;
  const unsigned ntk = (unsigned)nper << bs_log2;  // number of tiekeys incl. holes
  for (int k = tid; k < n; k += THREADS) temp[k] = 1e10f;
  int old = 0;
  if (tid == 0) idxs[0] = 0;
  __syncthreads();
  for (int j = 1; j < m; ++j) {
    const float x1 = dataset[3 * old + 0], y1 = dataset[3 * old + 1], z1 = dataset[3 * old + 2];
    float best = -1.0f;
    unsigned besttk = 0;
    for (unsigned tk = tid; tk < ntk; tk += THREADS) {  // ascending tiekey per thread
      const int k = tiekey_to_k(tk, bs_log2, nper);
      if (k < n) {
        const float d = sqdist<CONV>(dataset[3 * k + 0], dataset[3 * k + 1], dataset[3 * k + 2], x1, y1, z1);
        const float d2 = fminf(d, temp[k]);
        temp[k] = d2;
        if (d2 > best) { best = d2; besttk = tk; }
      }
    }
    const unsigned vkey = best < 0.0f ? 0u : (__float_as_uint(best) + 1u);
    const unsigned vmax = wave_max_u32(vkey);
    const unsigned tkmin = wave_min_u32((vkey == vmax) ? besttk : 0xffffffffu);
    if (lane_id() == 0) { xchg[j & 1][tid >> 6][0] = vmax; xchg[j & 1][tid >> 6][1] = tkmin; }
    __syncthreads();
    unsigned bv = 0u, bt = 0xffffffffu;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const unsigned v = xchg[j & 1][w][0], t = xchg[j & 1][w][1];
      const bool better = (v > bv) || (v == bv && t < bt);
      bv = better ? v : bv;
      bt = better ? t : bt;
    }
    old = tiekey_to_k(bt, bs_log2, nper);
    if (tid == 0) idxs[j] = old;
  }
}

// ============================================================================
// gather_points (+grad)   (sampling_gpu.cu:13-52)
// ============================================================================
__global__ void gather_points_kernel(int c, int n, int m, const float* __restrict__ points,
                                     const int* __restrict__ idx, float* __restrict__ out) {
  const int b = blockIdx.z, l = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const int a = idx[(size_t)b * m + j];
  out[((size_t)b * c + l) * m + j] = points[((size_t)b * c + l) * n + a];
}

// ============================================================================
// ball query   (ball_query_gpu.cu:14-49)
// ============================================================================
constexpr int kBqWaves = 4;            // waves per workgroup
constexpr int kBqCentroidsPerWave = 4;  // centroids handled sequentially by one wave

template <bool LDS_XYZ, int CONV>
__global__ __launch_bounds__(kBqWaves * 64) void ball_query_kernel(
    int n, int m, float radius2, int nsample, const float* __restrict__ new_xyz_all,
    const float* __restrict__ xyz_all, int* __restrict__ idx_all) {
  extern __shared__ __attribute__((aligned(16))) float bq_lds[];  // SoA: x[n] y[n] z[n]
  const int cloud = blockIdx.y;
  const float* xyz = xyz_all + (size_t)cloud * n * 3;
  const float* new_xyz = new_xyz_all + (size_t)cloud * m * 3;
  int* idx = idx_all + (size_t)cloud * m * nsample;
  const int lane = lane_id(), wave = threadIdx.x >> 6;

  if (LDS_XYZ) {
    for (int e = threadIdx.x; e < 3 * n; e += kBqWaves * 64) {
      const int k = e / 3, comp = e - 3 * k;
      bq_lds[comp * n + k] = xyz[e];
    }
    __syncthreads();
  }
  const int j0 = (blockIdx.x * kBqWaves + wave) * kBqCentroidsPerWave;
  for (int jj = 0; jj < kBqCentroidsPerWave; ++jj) {
    const int j = j0 + jj;
    if (j >= m) break;  // wave-uniform
    const float cx = new_xyz[3 * j + 0], cy = new_xyz[3 * j + 1], cz = new_xyz[3 * j + 2];
    int* row = idx + (size_t)j * nsample;
    int cnt = 0, first = 0;
    for (int base = 0; base < n && cnt < nsample; base += 64) {
      const int k = base + lane;
      bool hit = false;
      if (k < n) {
        float x, y, z;
        if (LDS_XYZ) { x = bq_lds[k]; y = bq_lds[n + k]; z = bq_lds[2 * n + k]; }
        else { x = xyz[3 * k + 0]; y = xyz[3 * k + 1]; z = xyz[3 * k + 2]; }
        hit = sqdist<CONV>(cx, cy, cz, x, y, z) < radius2;
      }
      const unsigned long long mask = __ballot(hit);
      if (mask) {
        if (cnt == 0) first = base + __ffsll((long long)mask) - 1;
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        const int pos = cnt + before;
        if (hit && pos < nsample) row[pos] = k;
        cnt += __popcll(mask);
      }
    }
    cnt = cnt < nsample ? cnt : nsample;
    for (int pos = cnt + lane; pos < nsample; pos += 64) row[pos] = first;  // pad / zeros
  }
}

// Both radii of an MSG level in ONE pass over the cloud (model/modules.py:249-297: every level queries the same centroids
// with two radii): the squared distance of a (centroid, point) pair is evaluated once and tested against both radii; each
// list is emitted exactly as ball_query_kernel emits it (ordered ballot emit, first-hit padding), so both index tensors are
// bit-identical to two single launches.  Optional: glen[cloud * m + j] = compact-column count of the row (csrc/sa_compact.hip:
// hits capped at nsample, plus one representative of the padded repeats), which the stand-alone count launch re-derives.
struct BqList {
  float radius2;
  int nsample;
  int* idx;
  int* glen;   // or null
};
template <bool LDS_XYZ, int CONV>
__global__ __launch_bounds__(kBqWaves * 64) void ball_query_pair_kernel(
    int n, int m, BqList la, BqList lb, const float* __restrict__ new_xyz_all, const float* __restrict__ xyz_all) {
  extern __shared__ __attribute__((aligned(16))) float bq_lds[];  // SoA: x[n] y[n] z[n]
  const int cloud = blockIdx.y;
  const float* xyz = xyz_all + (size_t)cloud * n * 3;
  const float* new_xyz = new_xyz_all + (size_t)cloud * m * 3;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  if (LDS_XYZ) {
    for (int e = threadIdx.x; e < 3 * n; e += kBqWaves * 64) {
      const int k = e / 3, comp = e - 3 * k;
      bq_lds[comp * n + k] = xyz[e];
    }
    __syncthreads();
  }
  const int j0 = (blockIdx.x * kBqWaves + wave) * kBqCentroidsPerWave;
  for (int jj = 0; jj < kBqCentroidsPerWave; ++jj) {
    const int j = j0 + jj;
    if (j >= m) break;  // wave-uniform
    const float cx = new_xyz[3 * j + 0], cy = new_xyz[3 * j + 1], cz = new_xyz[3 * j + 2];
    int* rowa = la.idx + ((size_t)cloud * m + j) * la.nsample;
    int* rowb = lb.idx + ((size_t)cloud * m + j) * lb.nsample;
    int cnta = 0, firsta = 0, cntb = 0, firstb = 0;
    for (int base = 0; base < n && (cnta < la.nsample || cntb < lb.nsample); base += 64) {
      const int k = base + lane;
      bool hita = false, hitb = false;
      if (k < n) {
        float x, y, z;
        if (LDS_XYZ) { x = bq_lds[k]; y = bq_lds[n + k]; z = bq_lds[2 * n + k]; }
        else { x = xyz[3 * k + 0]; y = xyz[3 * k + 1]; z = xyz[3 * k + 2]; }
        const float d2 = sqdist<CONV>(cx, cy, cz, x, y, z);
        hita = d2 < la.radius2;
        hitb = d2 < lb.radius2;
      }
      if (cnta < la.nsample) {
        const unsigned long long mask = __ballot(hita);
        if (mask) {
          if (cnta == 0) firsta = base + __ffsll((long long)mask) - 1;
          const int pos = cnta + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          if (hita && pos < la.nsample) rowa[pos] = k;
          cnta += __popcll(mask);
        }
      }
      if (cntb < lb.nsample) {
        const unsigned long long mask = __ballot(hitb);
        if (mask) {
          if (cntb == 0) firstb = base + __ffsll((long long)mask) - 1;
          const int pos = cntb + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          if (hitb && pos < lb.nsample) rowb[pos] = k;
          cntb += __popcll(mask);
        }
      }
    }
    cnta = cnta < la.nsample ? cnta : la.nsample;
    cntb = cntb < lb.nsample ? cntb : lb.nsample;
    for (int pos = cnta + lane; pos < la.nsample; pos += 64) rowa[pos] = firsta;  // pad / zeros
    for (int pos = cntb + lane; pos < lb.nsample; pos += 64) rowb[pos] = firstb;
    if (lane == 0) {
      // a row without a hit is all zeros: one distinct entry (index 0), like a row with one hit
      if (la.glen != nullptr) { const int c = cnta > 0 ? cnta : 1; la.glen[(size_t)cloud * m + j] = c + (c < la.nsample ? 1 : 0); }
      if (lb.glen != nullptr) { const int c = cntb > 0 ? cntb : 1; lb.glen[(size_t)cloud * m + j] = c + (c < lb.nsample ? 1 : 0); }
    }
  }
}

// ============================================================================
// group_points   (group_points_gpu.cu:13-33)
// ============================================================================
constexpr int kGroupChunk = 8;  // channels per thread (index reuse)

__global__ __launch_bounds__(256) void group_points_vec4_kernel(int c, int n, int P4,
                                                                const float* __restrict__ points,
                                                                const int* __restrict__ idx,
                                                                float* __restrict__ out) {
  const int b = blockIdx.z;
  const int p4 = blockIdx.x * 256 + threadIdx.x;
  if (p4 >= P4) return;
  const int4 ii = reinterpret_cast<const int4*>(idx + (size_t)b * P4 * 4)[p4];
  const int c0 = blockIdx.y * kGroupChunk;
  const int c1 = min(c0 + kGroupChunk, c);
  for (int l = c0; l < c1; ++l) {
    const float* row = points + ((size_t)b * c + l) * n;
    float4 v;
    v.x = row[ii.x]; v.y = row[ii.y]; v.z = row[ii.z]; v.w = row[ii.w];
    reinterpret_cast<float4*>(out + ((size_t)b * c + l) * P4 * 4)[p4] = v;
  }
}
__global__ __launch_bounds__(256) void group_points_scalar_kernel(int c, int n, int P,
                                                                  const float* __restrict__ points,
                                                                  const int* __restrict__ idx,
                                                                  float* __restrict__ out) {
  const int b = blockIdx.z;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int ii = idx[(size_t)b * P + p];
  const int c0 = blockIdx.y * kGroupChunk;
  const int c1 = min(c0 + kGroupChunk, c);
  for (int l = c0; l < c1; ++l)
    out[((size_t)b * c + l) * P + p] = points[((size_t)b * c + l) * n + ii];
}

// ============================================================================
// scatter-add family (group_points_grad, group_points_gpu.cu:48-69; three_interpolate_grad,
// interpolate_gpu.cu:121-148; gather_points_grad, sampling_gpu.cu:39-52).
// A workgroup owns CH channel rows of one cloud, accumulates them in LDS and writes them back
// once: no global atomics, no pre-zeroed output.  Padded ball-query slots repeat the first hit,
// so a group row is mostly runs of one index: each thread walks one group and flushes a run
// with a single LDS atomic instead of nsample conflicting ones.
// ============================================================================
constexpr int kScatterThreads = 256;
#ifndef ISTNET_GROUP_GRAD_CH
#define ISTNET_GROUP_GRAD_CH 2
#endif
#ifndef ISTNET_GROUP_GRAD_UNROLL
#define ISTNET_GROUP_GRAD_UNROLL 4
#endif
constexpr int kGroupGradCH = ISTNET_GROUP_GRAD_CH;
constexpr int kInterpGradCH = 8;

// group_points_grad: lanes walk consecutive grouped slots p (coalesced), runs of equal indices inside a
// 16-lane DPP row are summed with a segmented scan (row_shr 1,2,4,8) and only the last lane of a run
// issues the LDS atomic.
template <int CTRL>
__device__ __forceinline__ int dpp_row_i(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_row_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
constexpr int kGroupGradUnroll = ISTNET_GROUP_GRAD_UNROLL;   // slots per thread in flight: the loop is latency-bound (few waves per CU)
__global__ __launch_bounds__(kScatterThreads) void group_grad_kernel(
    int c, int n, int P, const float* __restrict__ grad_out, const int* __restrict__ idx_all,
    float* __restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [CH][n]
  const int b = blockIdx.y, c0 = blockIdx.x * kGroupGradCH;
  const int nch = min(kGroupGradCH, c - c0);
  for (int i = threadIdx.x; i < nch * n; i += kScatterThreads) acc[i] = 0.f;
  __syncthreads();
  const int* idx = idx_all + (size_t)b * P;
  const float* g = grad_out + ((size_t)b * c + c0) * P;
  constexpr int STEP = kScatterThreads * kGroupGradUnroll;
  const int Pr = (P + STEP - 1) / STEP * STEP;  // whole waves stay converged
  for (int p0 = threadIdx.x; p0 < Pr; p0 += STEP) {
    int ii[kGroupGradUnroll];
    float v[kGroupGradUnroll][kGroupGradCH];
#pragma unroll
    for (int u = 0; u < kGroupGradUnroll; ++u) {          // all loads of the step are issued before the first use
      const int p = p0 + u * kScatterThreads;
      ii[u] = p < P ? idx[p] : -1;
#pragma unroll
      for (int ch = 0; ch < kGroupGradCH; ++ch)
        v[u][ch] = (p < P && ch < nch) ? g[(size_t)ch * P + p] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kGroupGradUnroll; ++u) {
      // head of a run: first lane of the 16-lane row or index differs from the previous lane
      const int prev = dpp_row_i<0x111>(-2, ii[u]);
      const int head0 = (prev != ii[u]) ? 1 : 0;
      const int nxt = dpp_row_i<0x101>(-3, ii[u]);   // row_shl:1 -> index of the next lane (or -3 at the row end)
      const bool tail = nxt != ii[u] && ii[u] >= 0;
#pragma unroll
      for (int ch = 0; ch < kGroupGradCH; ++ch) {
        float x = v[u][ch];
        int f = head0;
        float pv; int pf;
        pv = dpp_row_f<0x111>(x); pf = dpp_row_i<0x111>(1, f); x = f ? x : x + pv; f |= pf;
        pv = dpp_row_f<0x112>(x); pf = dpp_row_i<0x112>(1, f); x = f ? x : x + pv; f |= pf;
        pv = dpp_row_f<0x114>(x); pf = dpp_row_i<0x114>(1, f); x = f ? x : x + pv; f |= pf;
        pv = dpp_row_f<0x118>(x); pf = dpp_row_i<0x118>(1, f); x = f ? x : x + pv; f |= pf;
        if (tail && ch < nch) atomicAdd(&acc[ch * n + ii[u]], x);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nch * n; i += kScatterThreads)
    grad_points[((size_t)b * c + c0) * n + i] = acc[i];
}

// group_points_grad over inverse lists (istnet_pn2_csr_build): the LDS float-atomic unit bounds the kernel above at
// ~1 TB/s whatever the number of loads in flight (round-2 sweep of channels per workgroup x unroll: no effect), so
// the default route stages CH gradient rows point-major in LDS ([slot][CH], coalesced float4 reads of grad_out) and
// lets every source point sum its own list: no atomics, summation in a fixed order (four lanes take contiguous
// quarters of a list, combined as (q0 + q1) + (q2 + q3)), each grad_out element read from HBM exactly once.
template <int CH>
__global__ __launch_bounds__(256) void group_grad_csr_kernel(int c, int n, int P, const float* __restrict__ grad_out,
                                                             const int* __restrict__ off_all,
                                                             const int* __restrict__ ent_all,
                                                             float* __restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float gg_rows[];   // [P][CH]
  const int b = blockIdx.y, c0 = blockIdx.x * CH;
  const int nch = min(CH, c - c0);
  const float* g = grad_out + ((size_t)b * c + c0) * P;
  if ((P & 3) == 0) {
    for (int p = threadIdx.x * 4; p < P; p += 1024) {
      float4 v[CH];
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) v[ch] = *reinterpret_cast<const float4*>(g + (size_t)min(ch, nch - 1) * P + p);
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        gg_rows[(size_t)(p + 0) * CH + ch] = v[ch].x; gg_rows[(size_t)(p + 1) * CH + ch] = v[ch].y;
        gg_rows[(size_t)(p + 2) * CH + ch] = v[ch].z; gg_rows[(size_t)(p + 3) * CH + ch] = v[ch].w;
      }
    }
  } else {
    for (int p = threadIdx.x; p < P; p += 256)
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) gg_rows[(size_t)p * CH + ch] = g[(size_t)min(ch, nch - 1) * P + p];
  }
  __syncthreads();
  const int* off = off_all + (size_t)b * (n + 1);
  const int* ent = ent_all + (size_t)b * P;
  const int part = threadIdx.x & 3;
  const int n_round = (n + 63) / 64 * 64;            // whole quads stay converged for the DPP combine
  for (int i = threadIdx.x >> 2; i < n_round; i += 64) {
    const bool valid = i < n;
    const int a0 = valid ? off[i] : 0, z0 = valid ? off[i + 1] : 0;
    const int q = (z0 - a0 + 3) >> 2;
    const int a = min(a0 + part * q, z0), z = min(a + q, z0);
    float sum[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) sum[ch] = 0.f;
    for (int u = a; u < z; ++u) {
      const int e = ent[u];
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) sum[ch] += gg_rows[(size_t)e * CH + ch];
    }
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      sum[ch] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sum[ch]), 0xB1, 0xf, 0xf, false));   // [1,0,3,2]
      sum[ch] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sum[ch]), 0x4E, 0xf, 0xf, false));   // [2,3,0,1]
    }
    if (valid && part == 0) {
#pragma unroll
      for (int ch = 0; ch < CH; ++ch)
        if (ch < nch) grad_points[((size_t)b * c + c0 + ch) * n + i] = sum[ch];
    }
  }
}

// ---- three_interpolate_grad without atomics: per-cloud inverse lists (CSR) of the 3n taps, then a gather.
// build: offsets off[b][m+1], entries ent[b][3n] = tap number e = 3*j + t, grouped by known point and
// sorted ascending inside each group (=> the summation order of the serial reference loop).
__global__ __launch_bounds__(256) void interp_csr_build_kernel(int E, int m, const int* __restrict__ idx_all,
                                                               int* __restrict__ off_all,
                                                               int* __restrict__ ent_all) {
  extern __shared__ __attribute__((aligned(16))) int lds_i[];  // cnt[m] | cur[m] | off[m+1] | part[256]
  int* cnt = lds_i;
  int* cur = lds_i + m;
  int* off = lds_i + 2 * m;
  int* part = lds_i + 3 * m + 1;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int* idx = idx_all + (size_t)b * E;   // E index slots per cloud (3n taps of three_nn, or npoint*nsample ball slots)
  int* ent = ent_all + (size_t)b * E;
  for (int i = tid; i < m; i += 256) cnt[i] = 0;
  __syncthreads();
  for (int e = tid; e < E; e += 256) atomicAdd(&cnt[idx[e]], 1);
  __syncthreads();
  // exclusive scan of cnt -> off: each thread owns a contiguous slice
  const int per = (m + 255) / 256;
  const int lo = min(tid * per, m), hi = min(lo + per, m);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += cnt[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < 256; ++t) { const int v = part[t]; part[t] = run; run += v; }
    off[m] = run;
  }
  __syncthreads();
  int run = part[tid];
  for (int i = lo; i < hi; ++i) { off[i] = run; cur[i] = run; run += cnt[i]; }
  __syncthreads();
  for (int e = tid; e < E; e += 256) {
    const int slot = atomicAdd(&cur[idx[e]], 1);
    ent[slot] = e;
  }
  __syncthreads();  // ent was written with global stores by this workgroup: make them visible to it
  __threadfence_block();
  for (int i = tid; i < m; i += 256) {  // sort each (short) list ascending
    const int a = off[i], z = off[i + 1];
    for (int u = a + 1; u < z; ++u) {
      const int key = ent[u];
      int v = u - 1;
      while (v >= a && ent[v] > key) { ent[v + 1] = ent[v]; --v; }
      ent[v + 1] = key;
    }
  }
  for (int i = tid; i <= m; i += 256) off_all[(size_t)b * (m + 1) + i] = off[i];
}

// The same lists with the entry array resident in LDS (E + 3m + 257 ints must fit) and a stable placement instead
// of fill-then-sort: a ball-query source point that pads many groups is referenced by hundreds of slots, and the
// per-list insertion sort of the kernel above is O(L^2) dependent accesses (milliseconds against global memory).
__global__ __launch_bounds__(256) void csr_build_lds_kernel(int E, int m, const int* __restrict__ idx_all,
                                                            int* __restrict__ off_all, int* __restrict__ ent_all) {
  extern __shared__ __attribute__((aligned(16))) int lds_i[];  // keys[256] | cnt[m] | cur[m] | off[m+1] | part[256] | ent[E]
  int* keys = lds_i;                                            // 16-byte aligned: read as int4
  int* cnt = lds_i + 256;
  int* cur = cnt + m;
  int* off = cnt + 2 * m;
  int* part = cnt + 3 * m + 1;
  int* ent = cnt + 3 * m + 1 + 256;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int* idx = idx_all + (size_t)b * E;
  for (int i = tid; i < m; i += 256) cnt[i] = 0;
  __syncthreads();
  for (int e = tid; e < E; e += 256) atomicAdd(&cnt[idx[e]], 1);
  __syncthreads();
  const int per = (m + 255) / 256;
  const int lo = min(tid * per, m), hi = min(lo + per, m);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += cnt[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < 256; ++t) { const int v = part[t]; part[t] = run; run += v; }
    off[m] = run;
  }
  __syncthreads();
  int run = part[tid];
  for (int i = lo; i < hi; ++i) { off[i] = run; cur[i] = run; run += cnt[i]; }
  __syncthreads();
  // Stable placement, no sort: slots are taken in ascending rounds of 256; a slot's position in its list is the
  // list's fill level before the round plus the number of lower threads of the round with the same key (counted
  // against an LDS copy of the round's keys; every thread reads the same address at a time -> broadcast reads).
  for (int e0 = 0; e0 < E; e0 += 256) {
    const int e = e0 + tid;
    const int key = e < E ? idx[e] : -1;
    keys[tid] = key;
    __syncthreads();
    int rank = 0;
    for (int j = 0; j < tid; ++j) rank += (keys[j] == key) ? 1 : 0;   // (an int4-per-read variant measured slower)
    const int base = key >= 0 ? cur[key] : 0;
    __syncthreads();                                   // everyone has read the fill levels of this round
    if (key >= 0) {
      ent[base + rank] = e;
      atomicAdd(&cur[key], 1);
    }
    __syncthreads();
  }
  for (int e = tid; e < E; e += 256) ent_all[(size_t)b * E + e] = ent[e];
  for (int i = tid; i <= m; i += 256) off_all[(size_t)b * (m + 1) + i] = off[i];
}

// ---- inverse lists, key-range decomposition: the default build ------------------------------------------------
// csr_build_lds_kernel above runs ONE workgroup per cloud with a 256-iteration rank loop per thread per round
// (grid = B = 32 on a 256-CU chip: 8 % of all kernel time of a training step in round 1).  Here a workgroup owns a
// RANGE of 16..64 keys (source points) of one cloud, so a launch has B * ceil(m / range) workgroups and needs no atomics,
// no sort and no communication between workgroups:
//   phase 1  every workgroup scans all E slots of its cloud (L2-resident, coalesced); each of its four waves takes a
//            contiguous quarter of the slots and appends the ones whose key is in the range to a wave-private LDS
//            queue with __ballot / mbcnt ordered compaction -- the queue is in ascending slot order -- and counts the
//            slots with a smaller key (their total is where the range's first list starts);
//   phase 2  each wave counts the keys of its own queue (lane k holds key k's counter), the four waves' counts are
//            exchanged through LDS (one barrier), a 64-lane scan turns the list lengths into offsets, and a second
//            pass over the queue writes the entries.  Lists come out in ascending slot order by construction: the
//            summation order of the serial reference loops, bit-identical to the kernels above.
// Several independent problems (the ball-query and three_nn index tensors of all levels of an encoder pass) share one
// launch through a descriptor table.
constexpr int kCsrKeys = 64;
constexpr int kCsrMaxProblems = 12;
struct CsrProblem {
  const int* idx;   // (B, E) keys in [0, m)
  int* off;         // (B, m + 1)
  int* ent;         // (B, E)
  int E, m, ranges, block_begin, qcap, kr;   // kr = keys per workgroup (16 / 32 / 64)
  const int* seg;   // optional (B + 1) device offsets: cloud b owns slots idx[seg[b] .. seg[b+1]) (at most E of them) and
                    // writes its entries (slot numbers relative to seg[b]) to ent + seg[b]; NULL = the (B, E) layout
  int key_sub;      // seg != NULL: keys are idx[..] - b * key_sub (global -> per-cloud point numbers)
};
struct CsrBatch {
  CsrProblem p[kCsrMaxProblems];
  int n;
};
__global__ __launch_bounds__(256) void csr_range_kernel(CsrBatch cb) {
  extern __shared__ __attribute__((aligned(16))) int csr_lds[];   // queue[4][qcap] | cnt[4][64] | less[4]
  int l = 0;
  while (l + 1 < cb.n && (int)blockIdx.x >= cb.p[l + 1].block_begin) ++l;
  const CsrProblem& P = cb.p[l];
  const int local = (int)blockIdx.x - P.block_begin;
  const int b = local / P.ranges, r = local - b * P.ranges;
  const int KR = P.kr;
  const int k0 = r * KR, m = P.m, qcap = P.qcap;
  const int E = P.seg != nullptr ? P.seg[b + 1] - P.seg[b] : P.E;
  const int* idx = P.seg != nullptr ? P.idx + P.seg[b] : P.idx + (size_t)b * P.E;
  const int ksub = P.seg != nullptr ? b * P.key_sub : 0;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  int* q = csr_lds + wave * qcap;
  int* cnts = csr_lds + 4 * qcap;
  int* lessw = cnts + 4 * kCsrKeys;
  // ---- phase 1: ordered compaction of this wave's quarter of the slots ----
  const int per = ((E + 3) / 4 + 63) / 64 * 64;
  const int beg = min(wave * per, E), end = min(beg + per, E);
  int qlen = 0, nless = 0;
  for (int e0 = beg; e0 < end; e0 += 4 * 64) {
    int key[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                       // four loads in flight
      const int e = e0 + u * 64 + lane;
      key[u] = e < end ? idx[e] - ksub : 0x7fffffff;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * 64 + lane;
      const bool in = (unsigned)(key[u] - k0) < (unsigned)KR;
      const unsigned long long bi = __ballot(in);
      nless += __popcll(__ballot(key[u] < k0));
      if (in) {
        const int pos = qlen + __builtin_amdgcn_mbcnt_hi((unsigned)(bi >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bi, 0u));
        q[pos] = (e << 7) | (key[u] - k0);
      }
      qlen += __popcll(bi);
    }
  }
  // ---- phase 2: lane = key holds that key's counters.  Two ways to consume the (ordered) queue, same result:
  //   match   blocks of 64 items (lane = item), one iteration per DISTINCT key of a block: all items with the leader's
  //           key are handled at once (ballot) and leave as consecutive stores.  Cost ~ distinct keys: right for the
  //           queues dominated by runs of one key -- padded ball rows repeat their first hit up to nsample times,
  //           and the small indices that are "first hit" of most balls concentrate in the first key ranges;
  //   walk    every lane reads every item (LDS broadcast) and keeps what carries its key.  Cost ~ items: right for
  //           queues of mostly distinct keys, where a match iteration is a serial chain (ballot -> readlane -> compare).
  // A wave picks by the number of distinct keys in its first block.
  if (lane < 4) q[qlen + lane] = 127;          // pad to a multiple of 4 with a key no lane owns (walk reads int4)
  const int qlen4 = (qlen + 3) & ~3;
  bool use_match = false;
  {
    const int item = lane < qlen ? q[lane] : -1;
    const int kl = item < 0 ? 127 : (item & 127);
    unsigned long long todo = __ballot(item >= 0);
    int distinct = 0;
    while (todo && distinct <= 16) {
      const int k = __builtin_amdgcn_readlane(kl, __ffsll((long long)todo) - 1);
      todo &= ~__ballot(kl == k);
      ++distinct;
    }
    use_match = distinct <= 16 && qlen > 64;
  }
  int cnt = 0;
  if (use_match) {
    for (int i = 0; i < qlen; i += 64) {
      const int item = i + lane < qlen ? q[i + lane] : -1;
      const int kl = item < 0 ? 127 : (item & 127);
      unsigned long long todo = __ballot(item >= 0);
      while (todo) {
        const int k = __builtin_amdgcn_readlane(kl, __ffsll((long long)todo) - 1);
        const unsigned long long mask = __ballot(kl == k);
        cnt += lane == k ? __popcll(mask) : 0;
        todo &= ~mask;
      }
    }
  } else {
    for (int i = 0; i < qlen4; i += 4) {
      const int4 it = *reinterpret_cast<const int4*>(q + i);   // same address in every lane: LDS broadcast
      cnt += ((it.x & 127) == lane) + ((it.y & 127) == lane) + ((it.z & 127) == lane) + ((it.w & 127) == lane);
    }
  }
  cnts[wave * kCsrKeys + lane] = cnt;
  if (lane == 0) lessw[wave] = nless;
  __syncthreads();
  const int c0 = cnts[lane], c1 = cnts[kCsrKeys + lane], c2 = cnts[2 * kCsrKeys + lane], c3 = cnts[3 * kCsrKeys + lane];
  const int total = (c0 + c1) + (c2 + c3);
  int incl = total;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  const int start = (lessw[0] + lessw[1]) + (lessw[2] + lessw[3]) + (incl - total);   // off[k0 + lane]
  int* off = P.off + (size_t)b * (m + 1);
  if (wave == 0) {
    if (lane < KR && k0 + lane < m) off[k0 + lane] = start;
    if (r == P.ranges - 1 && lane == 0) off[m] = E;
  }
  int pos = start + (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);   // next free entry of key `lane`
  int* ent = P.seg != nullptr ? P.ent + P.seg[b] : P.ent + (size_t)b * E;
  if (use_match) {
    for (int i = 0; i < qlen; i += 64) {
      const int item = i + lane < qlen ? q[i + lane] : -1;
      const int kl = item < 0 ? 127 : (item & 127);
      unsigned long long todo = __ballot(item >= 0);
      while (todo) {
        const int k = __builtin_amdgcn_readlane(kl, __ffsll((long long)todo) - 1);
        const unsigned long long mask = __ballot(kl == k);
        const int base = __builtin_amdgcn_readlane(pos, k);
        if (kl == k)
          ent[base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))] = item >> 7;
        pos += lane == k ? __popcll(mask) : 0;
        todo &= ~mask;
      }
    }
  } else {
    for (int i = 0; i < qlen4; i += 4) {
      const int4 it = *reinterpret_cast<const int4*>(q + i);
      if ((it.x & 127) == lane) ent[pos++] = it.x >> 7;
      if ((it.y & 127) == lane) ent[pos++] = it.y >> 7;
      if ((it.z & 127) == lane) ent[pos++] = it.z >> 7;
      if ((it.w & 127) == lane) ent[pos++] = it.w >> 7;
    }
  }
}

constexpr int kInterpCsrCH = 8;
__global__ __launch_bounds__(256) void interp_grad_csr_kernel(int c, int n, int m,
                                                              const float* __restrict__ grad_out,
                                                              const float* __restrict__ w_all,
                                                              const int* __restrict__ off_all,
                                                              const int* __restrict__ ent_all,
                                                              float* __restrict__ grad_points) {
  const int b = blockIdx.z, c0 = blockIdx.y * kInterpCsrCH;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const int* off = off_all + (size_t)b * (m + 1);
  const int* ent = ent_all + (size_t)b * n * 3;
  const float* w = w_all + (size_t)b * n * 3;
  const int a = off[i], z = off[i + 1];
  const int nch = min(kInterpCsrCH, c - c0);
  float sum[kInterpCsrCH];
#pragma unroll
  for (int ch = 0; ch < kInterpCsrCH; ++ch) sum[ch] = 0.f;
  // four taps per step: the dependent chain entry -> (weight, gradient) is two HBM/L2 round trips, so issue the
  // loads of four taps together (summation order stays ascending)
  for (int u = a; u < z; u += 4) {
    int e[4];
    float we[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = ent[min(u + q, z - 1)];
#pragma unroll
    for (int q = 0; q < 4; ++q) we[q] = (u + q < z) ? w[e[q]] : 0.f;
    float gq[4][kInterpCsrCH];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = e[q] / 3;
#pragma unroll
      for (int ch = 0; ch < kInterpCsrCH; ++ch)
        gq[q][ch] = grad_out[((size_t)b * c + c0 + min(ch, nch - 1)) * n + j];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int ch = 0; ch < kInterpCsrCH; ++ch) sum[ch] += gq[q][ch] * we[q];
  }
#pragma unroll
  for (int ch = 0; ch < kInterpCsrCH; ++ch)
    if (ch < nch) grad_points[((size_t)b * c + c0 + ch) * m + i] = sum[ch];
}

__global__ __launch_bounds__(kScatterThreads) void interp_grad_kernel(
    int c, int m, int n, const float* __restrict__ grad_out, const int* __restrict__ idx_all,
    const float* __restrict__ w_all, float* __restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [CH][m]
  const int b = blockIdx.y, c0 = blockIdx.x * kInterpGradCH;
  const int nch = min(kInterpGradCH, c - c0);
  for (int i = threadIdx.x; i < nch * m; i += kScatterThreads) acc[i] = 0.f;
  __syncthreads();
  for (int wi = threadIdx.x; wi < nch * n; wi += kScatterThreads) {  // one (channel, point) per thread
    const int ch = wi / n, j = wi - ch * n;
    const int* ix = idx_all + ((size_t)b * n + j) * 3;
    const float* w = w_all + ((size_t)b * n + j) * 3;
    const int i1 = ix[0], i2 = ix[1], i3 = ix[2];
    const float w1 = w[0], w2 = w[1], w3 = w[2];
    {
      const float g = grad_out[((size_t)b * c + c0 + ch) * n + j];
      float* a = acc + ch * m;
      atomicAdd(&a[i1], g * w1);
      atomicAdd(&a[i2], g * w2);
      atomicAdd(&a[i3], g * w3);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nch * m; i += kScatterThreads)
    grad_points[((size_t)b * c + c0) * m + i] = acc[i];
}

// Fallback when a destination row does not fit in LDS: zero + global atomics.
__global__ void fill_zero_kernel(size_t count, float* __restrict__ p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) p[i] = 0.f;
}
template <int MODE>
__global__ __launch_bounds__(256) void scatter_add_global_kernel(
    int c, int dst_len, int src_len, const float* __restrict__ src_all,
    const int* __restrict__ idx_all, const float* __restrict__ w_all, float* __restrict__ dst_all) {
  const int b = blockIdx.z, l = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= src_len) return;
  const float g = src_all[((size_t)b * c + l) * src_len + p];
  float* dst = dst_all + ((size_t)b * c + l) * dst_len;
  if (MODE == 0) {
    atomicAdd(&dst[idx_all[(size_t)b * src_len + p]], g);
  } else {
    const int* idx = idx_all + ((size_t)b * src_len + p) * 3;
    const float* w = w_all + ((size_t)b * src_len + p) * 3;
    atomicAdd(&dst[idx[0]], g * w[0]);
    atomicAdd(&dst[idx[1]], g * w[1]);
    atomicAdd(&dst[idx[2]], g * w[2]);
  }
}

// ============================================================================
// three_nn   (interpolate_gpu.cu:14-64)
// ============================================================================
constexpr int kNnTile = 2048;  // known points staged per LDS tile (24 KB)

// FOUR lanes per unknown point (round 4): lane q of a quad scans the known points q, q + 4, ... in ascending order with
// the reference's strict-less insertion, then the quad merges its four sorted triples (two quad_perm exchanges) under the
// order (distance, index) -- which is exactly what the reference's sequential scan produces: a candidate displaces an entry
// only when strictly closer, so among equal distances the smaller index stays ahead.  One thread per point gave 128
// workgroups of 512-step loops on the encoder's largest level (32 us); this form gives 512 workgroups of 128-step loops.
struct Nn3 { float d1, d2, d3; int i1, i2, i3; };
__device__ __forceinline__ bool nn_before(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }
__device__ __forceinline__ void nn_insert(Nn3& t, float d, int k) {          // (d, k) into the sorted triple, (distance, index) order
  if (nn_before(d, k, t.d1, t.i1)) { t.d3 = t.d2; t.i3 = t.i2; t.d2 = t.d1; t.i2 = t.i1; t.d1 = d; t.i1 = k; }
  else if (nn_before(d, k, t.d2, t.i2)) { t.d3 = t.d2; t.i3 = t.i2; t.d2 = d; t.i2 = k; }
  else if (nn_before(d, k, t.d3, t.i3)) { t.d3 = d; t.i3 = k; }
}
template <int CTRL>
__device__ __forceinline__ void nn_merge_quad(Nn3& t) {
  const float e1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t.d1), CTRL, 0xf, 0xf, false));
  const float e2 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t.d2), CTRL, 0xf, 0xf, false));
  const float e3 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t.d3), CTRL, 0xf, 0xf, false));
  const int j1 = __builtin_amdgcn_mov_dpp(t.i1, CTRL, 0xf, 0xf, false);
  const int j2 = __builtin_amdgcn_mov_dpp(t.i2, CTRL, 0xf, 0xf, false);
  const int j3 = __builtin_amdgcn_mov_dpp(t.i3, CTRL, 0xf, 0xf, false);
  // an unfilled slot is (+inf, index 0, as the reference initialises it): it must never displace a real candidate, and two
  // unfilled slots are interchangeable -- insert the partner's entries only while they are real
  if (e1 < __builtin_inff()) nn_insert(t, e1, j1);
  if (e2 < __builtin_inff()) nn_insert(t, e2, j2);
  if (e3 < __builtin_inff()) nn_insert(t, e3, j3);
}

// one workgroup: 64 unknown points (tile) of one cloud, four lanes per point
template <int CONV>
__device__ __forceinline__ void three_nn_tile(float* kn, int n, int m, int tile, int cloud,
                                              const float* __restrict__ unknown_all,
                                              const float* __restrict__ known_all,
                                              float* __restrict__ dist2_all,
                                              int* __restrict__ idx_all,
                                              float* __restrict__ weight_all) {
  const float* unknown = unknown_all + (size_t)cloud * n * 3;
  const float* known = known_all + (size_t)cloud * m * 3;
  const int j = tile * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
  const bool active = j < n;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (active) { ux = unknown[3 * j + 0]; uy = unknown[3 * j + 1]; uz = unknown[3 * j + 2]; }
  // The reference keeps doubles initialised to 1e40 and stores them back as f32
  // (=> +inf when unfilled); f32 +inf gives the same comparisons and result.
  Nn3 t{__builtin_inff(), __builtin_inff(), __builtin_inff(), 0, 0, 0};
  for (int base = 0; base < m; base += kNnTile) {
    const int cntk = min(kNnTile, m - base);
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * cntk; e += 256) kn[e] = known[(size_t)3 * base + e];
    __syncthreads();
    for (int kk = sub; kk < cntk; kk += 4) {
      const float d = sqdist<CONV>(ux, uy, uz, kn[3 * kk + 0], kn[3 * kk + 1], kn[3 * kk + 2]);
      const int k = base + kk;
      // ascending k inside a lane: the reference's strict-less chain (a NaN distance is never taken, as there)
      if (d < t.d1) { t.d3 = t.d2; t.i3 = t.i2; t.d2 = t.d1; t.i2 = t.i1; t.d1 = d; t.i1 = k; }
      else if (d < t.d2) { t.d3 = t.d2; t.i3 = t.i2; t.d2 = d; t.i2 = k; }
      else if (d < t.d3) { t.d3 = d; t.i3 = k; }
    }
  }
  nn_merge_quad<0xB1>(t);          // quad_perm [1,0,3,2]: lanes 0<->1, 2<->3
  nn_merge_quad<0x4E>(t);          // quad_perm [2,3,0,1]: pairs 0,1 <-> 2,3
  if (active && sub == 0) {
    int* ix = idx_all + ((size_t)cloud * n + j) * 3;
    ix[0] = t.i1; ix[1] = t.i2; ix[2] = t.i3;
    if (dist2_all != nullptr) {
      float* d2 = dist2_all + ((size_t)cloud * n + j) * 3;
      d2[0] = t.d1; d2[1] = t.d2; d2[2] = t.d3;
    }
    if (weight_all != nullptr) {
      // inverse-distance weights of PointnetFPModule (pointnet2_modules.py:185-188 with ThreeNN's sqrt,
      // pointnet2_utils.py:140-149): r = 1 / (sqrt(d2) + 1e-8), w = r / (r0 + r1 + r2); IEEE sqrt and division
      const float r1 = 1.0f / (sqrtf(t.d1) + 1e-8f), r2 = 1.0f / (sqrtf(t.d2) + 1e-8f), r3 = 1.0f / (sqrtf(t.d3) + 1e-8f);
      const float norm = (r1 + r2) + r3;
      float* w = weight_all + ((size_t)cloud * n + j) * 3;
      w[0] = r1 / norm; w[1] = r2 / norm; w[2] = r3 / norm;
    }
  }
}

template <int CONV>
__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m,
                                                       const float* __restrict__ unknown_all,
                                                       const float* __restrict__ known_all,
                                                       float* __restrict__ dist2_all,
                                                       int* __restrict__ idx_all,
                                                       float* __restrict__ weight_all) {
  __shared__ __attribute__((aligned(16))) float kn[kNnTile * 3];
  three_nn_tile<CONV>(kn, n, m, blockIdx.x, blockIdx.y, unknown_all, known_all, dist2_all, idx_all, weight_all);
}

// The neighbour searches of all feature-propagation levels of an encoder pass (model/modules.py:322-325: four levels, each
// its own (n, m)) in ONE launch: blockIdx.x walks the tiles of problem 0, then problem 1, ...; every problem is evaluated by
// the same code as its stand-alone launch.
constexpr int kNnMaxProblems = 8;
struct NnBatch {
  int count;
  int n[kNnMaxProblems], m[kNnMaxProblems];
  int first[kNnMaxProblems + 1];          // first tile of every problem
  const float* unknown[kNnMaxProblems];
  const float* known[kNnMaxProblems];
  int* idx[kNnMaxProblems];
  float* weight[kNnMaxProblems];
};
template <int CONV>
__global__ __launch_bounds__(256) void three_nn_multi_kernel(NnBatch nb) {
  __shared__ __attribute__((aligned(16))) float kn[kNnTile * 3];
  int p = 0;
  while (p + 1 < nb.count && (int)blockIdx.x >= nb.first[p + 1]) ++p;      // uniform
  three_nn_tile<CONV>(kn, nb.n[p], nb.m[p], (int)blockIdx.x - nb.first[p], blockIdx.y, nb.unknown[p], nb.known[p],
                      (float*)nullptr, nb.idx[p], nb.weight[p]);
}

// ============================================================================
// three_interpolate   (interpolate_gpu.cu:77-106)
// ============================================================================
constexpr int kInterpChunk = 8;

__global__ __launch_bounds__(256) void three_interpolate_kernel(int c, int m, int n,
                                                                const float* __restrict__ points,
                                                                const int* __restrict__ idx,
                                                                const float* __restrict__ weight,
                                                                float* __restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int* ix = idx + ((size_t)b * n + j) * 3;
  const float* w = weight + ((size_t)b * n + j) * 3;
  const int i1 = ix[0], i2 = ix[1], i3 = ix[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int c0 = blockIdx.y * kInterpChunk;
  const int c1 = min(c0 + kInterpChunk, c);
  for (int l = c0; l < c1; ++l) {
    const float* row = points + ((size_t)b * c + l) * m;
    out[((size_t)b * c + l) * n + j] = (row[i1] * w1 + row[i2] * w2) + row[i3] * w3;
  }
}

int g_csr_legacy = 0;  // istnet_pn2_set_tuning key 2: 1 = one-workgroup-per-cloud list builds (A/B, tests)
int g_dist_conv = 0;  // distance convention (file header); istnet_pn2_set_tuning key 1
// expands LAUNCH three times with CONV_ = 0, 1, 2 and runs the one g_dist_conv selects
#define ISTNET_CONV_DISPATCH(LAUNCH)                                           \
  do {                                                                         \
    if (g_dist_conv == 1) { constexpr int CONV_ = 1; LAUNCH; }                 \
    else if (g_dist_conv == 2) { constexpr int CONV_ = 2; LAUNCH; }            \
    else { constexpr int CONV_ = 0; LAUNCH; }                                  \
  } while (0)
int g_fps_multiwave_min = 1024;  // key 0: tiekey-slot count from which a cloud gets several waves.  Round 6: with the records of the
                                 // waves combined by a DPP butterfly (fps_regs_kernel) four waves beat one at 1 024 points -- 512 of
                                 // 1 024, B = 32 / 4 / 1: 247 / 244 / 243 us against 288 / 283 / 280 (279 vs 295 with the tie
                                 // tracking of the chained form), un-pipelined step -20 us (profiles/r06_fps_waves.txt); below
                                 // 1 024 slots one wave stays ahead
int g_fps_waves = 4;             // key 3: waves per cloud of the multi-wave form (2, 4 or 8)
inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int ilog2_floor(int v) { int r = 0; while ((1 << (r + 1)) <= v) ++r; return r; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
constexpr int kMaxLdsRowBytes = 64 * 1024;

template <int MODE>
int launch_scatter_fallback(int b, int c, int dst_len, int src_len, const float* src, const int* idx,
                            const float* w, float* dst, hipStream_t st) {
  const size_t count = (size_t)b * c * dst_len;
  hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, count, dst);
  hipLaunchKernelGGL(scatter_add_global_kernel<MODE>, dim3(ceil_div(src_len, 256), c, b), dim3(256), 0,
                     st, c, dst_len, src_len, src, idx, w, dst);
  return (int)hipGetLastError();
}

int launch_group_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out,
                      const int* idx, float* grad_points, hipStream_t st) {
  const size_t lds = (size_t)kGroupGradCH * n * 4;
  if (lds > (size_t)kMaxLdsRowBytes)
    return launch_scatter_fallback<0>(b, c, n, npoints * nsample, grad_out, idx, nullptr, grad_points, st);
  hipLaunchKernelGGL(group_grad_kernel, dim3(ceil_div(c, kGroupGradCH), b), dim3(kScatterThreads), lds, st, c,
                     n, npoints * nsample, grad_out, idx, grad_points);
  return (int)hipGetLastError();
}

int launch_interp_grad(int b, int c, int n, int m, const float* grad_out, const int* idx, const float* w,
                       float* grad_points, hipStream_t st) {
  const size_t lds = (size_t)kInterpGradCH * m * 4;
  if (lds > (size_t)kMaxLdsRowBytes)
    return launch_scatter_fallback<1>(b, c, m, n, grad_out, idx, w, grad_points, st);
  hipLaunchKernelGGL(interp_grad_kernel, dim3(ceil_div(c, kInterpGradCH), b), dim3(kScatterThreads), lds, st,
                     c, m, n, grad_out, idx, w, grad_points);
  return (int)hipGetLastError();
}

template <int NW>
int launch_fps_regs(int b, int n, int m, int bs_log2, int nper, const float* dataset, int* idxs,
                    float* picked, const int* tie_in, int* tie_out, int track_rounds, hipStream_t st) {
  const int ppt = ceil_div(nper << bs_log2, NW * 64);  // tiekey slots per thread (holes included)
  const size_t lds = (size_t)3 * n * 4 + (NW > 1 ? 2 * NW * 4 * 4 : 0);
  const bool track = tie_in != nullptr || tie_out != nullptr;
#define ISTNET_FPS_CASE(P)                                                                                          \
  do {                                                                                                              \
    if (track)                                                                                                      \
      ISTNET_CONV_DISPATCH(hipLaunchKernelGGL((fps_regs_kernel<NW, P, CONV_, true>), dim3(b), dim3(NW * 64), lds, st, n, m, \
                                              bs_log2, nper, dataset, idxs, picked, tie_in, tie_out, track_rounds)); \
    else                                                                                                            \
      ISTNET_CONV_DISPATCH(hipLaunchKernelGGL((fps_regs_kernel<NW, P, CONV_, false>), dim3(b), dim3(NW * 64), lds, st, n, m, \
                                              bs_log2, nper, dataset, idxs, picked, tie_in, tie_out, track_rounds)); \
  } while (0)
  if (ppt <= 1) ISTNET_FPS_CASE(1);
  else if (ppt <= 2) ISTNET_FPS_CASE(2);
  else if (ppt <= 4) ISTNET_FPS_CASE(4);
  else if (ppt <= 8) ISTNET_FPS_CASE(8);
  else ISTNET_FPS_CASE(16);
#undef ISTNET_FPS_CASE
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int istnet_pn2_abi_version(void) { return ISTNET_PN2_ABI_VERSION; }

// Debug aid (tools/step_timeline.py): one thread stores the 100 MHz wall clock when the stream reaches this point.
__global__ void debug_marker_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
int istnet_debug_marker(unsigned long long* slot, void* stream) {
  if (slot == nullptr) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(debug_marker_kernel, dim3(1), dim3(1), 0, as_stream(stream), slot);
  return (int)hipGetLastError();
}
int istnet_pn2_set_tuning(int key, int value) {
  // key 0: tiekey-slot count from which FPS uses four waves per cloud; one wave holds at most 16 slots per lane
  if (key == 0) { if (value < 1 || value > 1025) return ISTNET_PN2_EINVAL; g_fps_multiwave_min = value; return 0; }   // 1025: one wave up to 1 024 slots
  // key 1: distance convention of FPS / ball query / three_nn (0 un-contracted, 1 / 2 FMA-contracted; file header)
  if (key == 1) { if (value < 0 || value > 2) return ISTNET_PN2_EINVAL; g_dist_conv = value; return 0; }
  // key 2: 1 = build inverse lists with the one-workgroup-per-cloud kernels (the fallback for very large slot counts)
  if (key == 2) { g_csr_legacy = value ? 1 : 0; return 0; }
  // key 3: waves per cloud of the multi-wave FPS (2, 4, 8)
  if (key == 3) { if (value != 2 && value != 4 && value != 8) return ISTNET_PN2_EINVAL; g_fps_waves = value; return 0; }
  return ISTNET_PN2_EINVAL;
}
const char* istnet_pn2_target(void) { return "gfx950"; }

static int fps_impl(int b, int n, int m, const float* dataset, float* temp, int* idxs, float* picked,
                    const int* tie_in, int* tie_out, int track_rounds, void* stream) {
  if (b < 0 || n <= 0 || m < 0) return ISTNET_PN2_EINVAL;
  if (b == 0 || m == 0) return 0;
  // reference block size: opt_n_threads(n) = clamp(2^floor(log2 n), 1, 512)  (cuda_utils.h:18-22)
  int bs_log2 = ilog2_floor(n);
  if (bs_log2 > 9) bs_log2 = 9;
  const int nper = ceil_div(n, 1 << bs_log2);
  const int slots = nper << bs_log2;
  hipStream_t st = as_stream(stream);
  // one wave per cloud is barrier-free and ahead of several waves below 1 024 slots (an LDS exchange + barrier per round cost
  // ~0.5 us per round in rounds 1-2: profiles/r01_index_microbench.txt); from 1 024 slots on four waves win since the records are
  // combined by a DPP butterfly (g_fps_multiwave_min above)
  if (slots < g_fps_multiwave_min)
    return launch_fps_regs<1>(b, n, m, bs_log2, nper, dataset, idxs, picked, tie_in, tie_out, track_rounds, st);
  if (slots <= 128 * 16 && g_fps_waves == 2)
    return launch_fps_regs<2>(b, n, m, bs_log2, nper, dataset, idxs, picked, tie_in, tie_out, track_rounds, st);
  if (slots <= 512 * 16 && g_fps_waves == 8)
    return launch_fps_regs<8>(b, n, m, bs_log2, nper, dataset, idxs, picked, tie_in, tie_out, track_rounds, st);
  if (slots <= 256 * 16)
    return launch_fps_regs<4>(b, n, m, bs_log2, nper, dataset, idxs, picked, tie_in, tie_out, track_rounds, st);
  if (temp == nullptr || picked != nullptr || tie_in != nullptr || tie_out != nullptr)
    return ISTNET_PN2_EINVAL;  // large clouds: scratch buffer, no fused gather, no chained sampling
  ISTNET_CONV_DISPATCH(hipLaunchKernelGGL(fps_generic_kernel<CONV_>, dim3(b), dim3(1024), 0, st, n, m, bs_log2, nper,
                                          dataset, temp, idxs));
  return (int)hipGetLastError();
}

int istnet_pn2_furthest_point_sampling(int b, int n, int m, const float* dataset, float* temp,
                                       int* idxs, void* stream) {
  return fps_impl(b, n, m, dataset, temp, idxs, nullptr, nullptr, nullptr, 0, stream);
}

int istnet_pn2_fps_gather(int b, int n, int m, const float* dataset, int* idxs, float* picked, void* stream) {
  if (picked == nullptr || n > 4096) return ISTNET_PN2_EINVAL;
  return fps_impl(b, n, m, dataset, nullptr, idxs, picked, nullptr, nullptr, 0, stream);
}

int istnet_pn2_fps_gather_chain(int b, int n, int m, const float* dataset, int* idxs, float* picked,
                                const int* tie_in, int* tie_out, int track_rounds, void* stream) {
  if (picked == nullptr || n > 4096 || tie_out == nullptr || track_rounds < 0) return ISTNET_PN2_EINVAL;
  return fps_impl(b, n, m, dataset, nullptr, idxs, picked, tie_in, tie_out, track_rounds, stream);
}

int istnet_pn2_gather_points(int b, int c, int n, int npoints, const float* points, const int* idx,
                             float* out, void* stream) {
  if (b < 0 || c < 0 || n <= 0 || npoints < 0) return ISTNET_PN2_EINVAL;
  if (b == 0 || c == 0 || npoints == 0) return 0;
  hipLaunchKernelGGL(gather_points_kernel, dim3(ceil_div(npoints, 256), c, b), dim3(256), 0,
                     as_stream(stream), c, n, npoints, points, idx, out);
  return (int)hipGetLastError();
}

int istnet_pn2_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out,
                                  const int* idx, float* grad_points, void* stream) {
  if (b < 0 || c < 0 || n <= 0 || npoints < 0) return ISTNET_PN2_EINVAL;
  if (b == 0 || c == 0) return 0;
  return launch_group_grad(b, c, n, npoints, 1, grad_out, idx, grad_points, as_stream(stream));
}

int istnet_pn2_query_ball_point(int b, int n, int m, float radius, int nsample,
                                const float* new_xyz, const float* xyz, int* idx, void* stream) {
  if (b < 0 || n <= 0 || m < 0 || nsample < 0) return ISTNET_PN2_EINVAL;
  if (b == 0 || m == 0 || nsample == 0) return 0;
  const float radius2 = radius * radius;  // ball_query_gpu.cu:27, f32 product
  const dim3 grid(ceil_div(m, kBqWaves * kBqCentroidsPerWave), b);
  const size_t lds = (size_t)3 * n * 4;
  if (lds <= 64 * 1024) {
    ISTNET_CONV_DISPATCH(hipLaunchKernelGGL((ball_query_kernel<true, CONV_>), grid, dim3(kBqWaves * 64), lds,
                                            as_stream(stream), n, m, radius2, nsample, new_xyz, xyz, idx));
  } else {
    ISTNET_CONV_DISPATCH(hipLaunchKernelGGL((ball_query_kernel<false, CONV_>), grid, dim3(kBqWaves * 64), 0,
                                            as_stream(stream), n, m, radius2, nsample, new_xyz, xyz, idx));
  }
  return (int)hipGetLastError();
}

int istnet_pn2_query_ball_point_pair(int b, int n, int m, float radius_a, int nsample_a, float radius_b, int nsample_b,
                                     const float* new_xyz, const float* xyz, int* idx_a, int* idx_b, int* glen_a,
                                     int* glen_b, void* stream) {
  if (b < 0 || n <= 0 || m < 0 || nsample_a <= 0 || nsample_b <= 0 || !idx_a || !idx_b) return ISTNET_PN2_EINVAL;
  if (b == 0 || m == 0) return 0;
  const BqList la{radius_a * radius_a, nsample_a, idx_a, glen_a};   // ball_query_gpu.cu:27, f32 products
  const BqList lb{radius_b * radius_b, nsample_b, idx_b, glen_b};
  const dim3 grid(ceil_div(m, kBqWaves * kBqCentroidsPerWave), b);
  const size_t lds = (size_t)3 * n * 4;
  if (lds <= 64 * 1024) {
    ISTNET_CONV_DISPATCH(hipLaunchKernelGGL((ball_query_pair_kernel<true, CONV_>), grid, dim3(kBqWaves * 64), lds,
                                            as_stream(stream), n, m, la, lb, new_xyz, xyz));
  } else {
    ISTNET_CONV_DISPATCH(hipLaunchKernelGGL((ball_query_pair_kernel<false, CONV_>), grid, dim3(kBqWaves * 64), 0,
                                            as_stream(stream), n, m, la, lb, new_xyz, xyz));
  }
  return (int)hipGetLastError();
}

int istnet_pn2_group_points(int b, int c, int n, int npoints, int nsample, const float* points,
                            const int* idx, float* out, void* stream) {
  if (b < 0 || c < 0 || n <= 0 || npoints < 0 || nsample < 0) return ISTNET_PN2_EINVAL;
  const long long P = (long long)npoints * nsample;
  if (b == 0 || c == 0 || P == 0) return 0;
  if (P % 4 == 0) {
    const int P4 = (int)(P / 4);
    hipLaunchKernelGGL(group_points_vec4_kernel, dim3(ceil_div(P4, 256), ceil_div(c, kGroupChunk), b),
                       dim3(256), 0, as_stream(stream), c, n, P4, points, idx, out);
  } else {
    hipLaunchKernelGGL(group_points_scalar_kernel,
                       dim3(ceil_div((int)P, 256), ceil_div(c, kGroupChunk), b), dim3(256), 0,
                       as_stream(stream), c, n, (int)P, points, idx, out);
  }
  return (int)hipGetLastError();
}

int istnet_pn2_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                 const float* grad_out, const int* idx, float* grad_points,
                                 void* stream) {
  if (b < 0 || c < 0 || n <= 0 || npoints < 0 || nsample < 0) return ISTNET_PN2_EINVAL;
  if (b == 0 || c == 0) return 0;
  if (npoints == 0 || nsample == 0) return launch_group_grad(b, c, n, 0, 1, grad_out, idx, grad_points, as_stream(stream));
  return launch_group_grad(b, c, n, npoints, nsample, grad_out, idx, grad_points, as_stream(stream));
}

// channels per workgroup of group_grad_csr_kernel: the rows must fit 64 KB of LDS and the grid should fill the chip
static int group_grad_csr_ch(int b, int c, long long P) {
  int ch = 16;
  while (ch > 1 && ((size_t)ch * P * 4 > (size_t)kMaxLdsRowBytes || (long long)b * ceil_div(c, ch) < 1024)) ch >>= 1;
  return ch;
}

int istnet_pn2_group_points_grad_csr(int b, int c, int n, int npoints, int nsample, const float* grad_out,
                                     const int* offsets, const int* entries, float* grad_points, void* stream) {
  if (b < 0 || c < 0 || n <= 0 || npoints <= 0 || nsample <= 0 || !offsets || !entries) return ISTNET_PN2_EINVAL;
  if (b == 0 || c == 0) return 0;
  const long long P = (long long)npoints * nsample;
  const int ch = group_grad_csr_ch(b, c, P);
  if ((size_t)ch * P * 4 > (size_t)kMaxLdsRowBytes) return ISTNET_PN2_EINVAL;   // one row does not fit: atomic kernel
  const dim3 grid(ceil_div(c, ch), b);
  const size_t lds = (size_t)ch * P * 4;
#define ISTNET_GGC(CH)                                                                                           \
  hipLaunchKernelGGL(group_grad_csr_kernel<CH>, grid, dim3(256), lds, as_stream(stream), c, n, (int)P, grad_out, \
                     offsets, entries, grad_points)
  switch (ch) {
    case 16: ISTNET_GGC(16); break;
    case 8: ISTNET_GGC(8); break;
    case 4: ISTNET_GGC(4); break;
    case 2: ISTNET_GGC(2); break;
    default: ISTNET_GGC(1); break;
  }
#undef ISTNET_GGC
  return (int)hipGetLastError();
}

int istnet_pn2_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2,
                        int* idx, void* stream) {
  if (b < 0 || n < 0 || m < 0) return ISTNET_PN2_EINVAL;
  if (b == 0 || n == 0) return 0;
  ISTNET_CONV_DISPATCH(hipLaunchKernelGGL(three_nn_kernel<CONV_>, dim3(ceil_div(n, 64), b), dim3(256), 0,
                                          as_stream(stream), n, m, unknown, known, dist2, idx, (float*)nullptr));
  return (int)hipGetLastError();
}

int istnet_pn2_three_nn_weights(int b, int n, int m, const float* unknown, const float* known, int* idx, float* weight,
                                void* stream) {
  if (b < 0 || n < 0 || m < 0 || !idx || !weight) return ISTNET_PN2_EINVAL;
  if (b == 0 || n == 0) return 0;
  ISTNET_CONV_DISPATCH(hipLaunchKernelGGL(three_nn_kernel<CONV_>, dim3(ceil_div(n, 64), b), dim3(256), 0,
                                          as_stream(stream), n, m, unknown, known, (float*)nullptr, idx, weight));
  return (int)hipGetLastError();
}

int istnet_pn2_three_nn_weights_multi(int nprob, int b, const int* n, const int* m, const float* const* unknown,
                                      const float* const* known, int* const* idx, float* const* weight, void* stream) {
  if (nprob <= 0 || nprob > kNnMaxProblems || b < 0 || !n || !m || !unknown || !known || !idx || !weight) return ISTNET_PN2_EINVAL;
  if (b == 0) return 0;
  NnBatch nb;
  nb.count = 0;
  int tiles = 0;
  for (int p = 0; p < nprob; ++p) {
    if (n[p] < 0 || m[p] < 0 || !idx[p] || !weight[p]) return ISTNET_PN2_EINVAL;
    if (n[p] == 0) continue;
    const int q = nb.count++;
    nb.n[q] = n[p]; nb.m[q] = m[p]; nb.first[q] = tiles;
    nb.unknown[q] = unknown[p]; nb.known[q] = known[p]; nb.idx[q] = idx[p]; nb.weight[q] = weight[p];
    tiles += ceil_div(n[p], 64);
  }
  if (nb.count == 0) return 0;
  nb.first[nb.count] = tiles;
  ISTNET_CONV_DISPATCH(hipLaunchKernelGGL(three_nn_multi_kernel<CONV_>, dim3(tiles, b), dim3(256), 0, as_stream(stream), nb));
  return (int)hipGetLastError();
}

int istnet_pn2_three_interpolate(int b, int c, int m, int n, const float* points, const int* idx,
                                 const float* weight, float* out, void* stream) {
  if (b < 0 || c < 0 || m <= 0 || n < 0) return ISTNET_PN2_EINVAL;
  if (b == 0 || c == 0 || n == 0) return 0;
  hipLaunchKernelGGL(three_interpolate_kernel, dim3(ceil_div(n, 256), ceil_div(c, kInterpChunk), b),
                     dim3(256), 0, as_stream(stream), c, m, n, points, idx, weight, out);
  return (int)hipGetLastError();
}

// queue capacity per wave of csr_range_kernel (ints): its quarter of the slots, rounded to whole 64-slot steps, + pad
static int csr_qcap(int e) { return ((e + 3) / 4 + 63) / 64 * 64 + 8; }
static size_t csr_range_lds(int e) { return ((size_t)4 * csr_qcap(e) + 4 * kCsrKeys + 4) * 4; }

static int csr_build_multi_impl(int nprob, int b, const int* e, const int* m, const int* const* idx,
                                int* const* offsets, int* const* entries, const int* const* seg, const int* key_sub,
                                void* stream);

int istnet_pn2_csr_build_multi(int nprob, int b, const int* e, const int* m, const int* const* idx,
                               int* const* offsets, int* const* entries, void* stream) {
  return csr_build_multi_impl(nprob, b, e, m, idx, offsets, entries, nullptr, nullptr, stream);
}

int istnet_pn2_csr_build_segmented(int nprob, int b, const int* e, const int* m, const int* const* idx,
                                   int* const* offsets, int* const* entries, const int* const* seg,
                                   const int* key_sub, void* stream) {
  if (!seg || !key_sub) return ISTNET_PN2_EINVAL;
  return csr_build_multi_impl(nprob, b, e, m, idx, offsets, entries, seg, key_sub, stream);
}

static int csr_build_multi_impl(int nprob, int b, const int* e, const int* m, const int* const* idx,
                                int* const* offsets, int* const* entries, const int* const* seg, const int* key_sub,
                                void* stream) {
  if (nprob <= 0 || nprob > kCsrMaxProblems || b < 0 || !e || !m || !idx || !offsets || !entries) return ISTNET_PN2_EINVAL;
  if (b == 0) return 0;
  CsrBatch cb;
  cb.n = nprob;
  int blocks = 0;
  size_t lds = 0;
  for (int l = 0; l < nprob; ++l) {
    if (e[l] < 0 || m[l] <= 0 || e[l] >= (1 << 24) || !idx[l] || !offsets[l] || !entries[l]) return ISTNET_PN2_EINVAL;
    const size_t need = csr_range_lds(e[l]);
    if (need > (size_t)kMaxLdsRowBytes) return ISTNET_PN2_EINVAL;     // caller builds this one with istnet_pn2_csr_build
    lds = need > lds ? need : lds;
    CsrProblem& P = cb.p[l];
    P.idx = idx[l]; P.off = offsets[l]; P.ent = entries[l];
    P.seg = seg != nullptr ? seg[l] : nullptr;
    P.key_sub = (seg != nullptr && key_sub != nullptr) ? key_sub[l] : 0;
    if (seg != nullptr && seg[l] == nullptr) return ISTNET_PN2_EINVAL;
    // keys per workgroup: 64 when that already gives >= 512 workgroups, else fewer keys -> more, shorter workgroups
    int kr = kCsrKeys;
    while (kr > 16 && (long long)b * ceil_div(m[l], kr) < 512) kr >>= 1;
    P.E = e[l]; P.m = m[l]; P.kr = kr; P.ranges = ceil_div(m[l], kr); P.block_begin = blocks; P.qcap = csr_qcap(e[l]);
    blocks += b * P.ranges;
  }
  hipLaunchKernelGGL(csr_range_kernel, dim3(blocks), dim3(256), lds, as_stream(stream), cb);
  return (int)hipGetLastError();
}

int istnet_pn2_csr_build(int b, int e, int m, const int* idx, int* offsets, int* entries, void* stream);

int istnet_pn2_interp_csr_build(int b, int n, int m, const int* idx, int* offsets, int* entries, void* stream) {
  if (b < 0 || n < 0 || m <= 0) return ISTNET_PN2_EINVAL;
  if (b == 0) return 0;
  return istnet_pn2_csr_build(b, 3 * n, m, idx, offsets, entries, stream);   // same lists: e = 3n taps per cloud
}

int istnet_pn2_csr_build(int b, int e, int m, const int* idx, int* offsets, int* entries, void* stream) {
  if (b < 0 || e < 0 || m <= 0) return ISTNET_PN2_EINVAL;
  if (b == 0) return 0;
  if (csr_range_lds(e) <= (size_t)kMaxLdsRowBytes && e < (1 << 24) && g_csr_legacy == 0)
    return istnet_pn2_csr_build_multi(1, b, &e, &m, &idx, &offsets, &entries, stream);
  // slot counts beyond the LDS queue of the range kernel: one workgroup per cloud
  const size_t lds = ((size_t)3 * m + 1 + 256) * 4;
  if (lds > (size_t)kMaxLdsRowBytes) return ISTNET_PN2_EINVAL;  // caller falls back to the atomic kernels
  if (lds + ((size_t)e + 256) * 4 <= (size_t)kMaxLdsRowBytes)
    hipLaunchKernelGGL(csr_build_lds_kernel, dim3(b), dim3(256), lds + ((size_t)e + 256) * 4, as_stream(stream), e, m,
                       idx, offsets, entries);
  else
    hipLaunchKernelGGL(interp_csr_build_kernel, dim3(b), dim3(256), lds, as_stream(stream), e, m, idx, offsets, entries);
  return (int)hipGetLastError();
}

int istnet_pn2_three_interpolate_grad_csr(int b, int c, int n, int m, const float* grad_out,
                                          const float* weight, const int* offsets, const int* entries,
                                          float* grad_points, void* stream) {
  if (b < 0 || c < 0 || m <= 0 || n < 0) return ISTNET_PN2_EINVAL;
  if (b == 0 || c == 0) return 0;
  hipLaunchKernelGGL(interp_grad_csr_kernel, dim3(ceil_div(m, 256), ceil_div(c, kInterpCsrCH), b), dim3(256), 0,
                     as_stream(stream), c, n, m, grad_out, weight, offsets, entries, grad_points);
  return (int)hipGetLastError();
}

int istnet_pn2_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out,
                                      const int* idx, const float* weight, float* grad_points,
                                      void* stream) {
  if (b < 0 || c < 0 || m <= 0 || n < 0) return ISTNET_PN2_EINVAL;
  if (b == 0 || c == 0) return 0;
  return launch_interp_grad(b, c, n, m, grad_out, idx, weight, grad_points, as_stream(stream));
}

}  // extern "C"
