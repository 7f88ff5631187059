// pw_mlp.hip -- fused per-point MLP ("SharedMLP") kernels for gfx950, fp32 MFMA.
//
// A SharedMLP layer of the reference (model/pointnet2/pytorch_utils.py:25-50,80-134) is
//     Conv2d 1x1 (no bias) -> BatchNorm2d -> ReLU          over (B, C, npoint, nsample)
// executed there as >=6 full passes over the activation per layer (cuDNN conv, BN statistics,
// BN apply, ReLU, plus layout transposes) and a separate max_pool2d.  Here a layer is ONE GEMM
// kernel on the raw (pre-BN) activations, with the previous layer's BN+ReLU applied while the
// operand is staged into LDS and this layer's BN statistics reduced in the epilogue:
//
//   forward   Y_l[b] = W_l . act(Y_{l-1}[b])            act(v) = relu(v*scale + shift)
//   backward  dY_l formed on the fly from (dA_l | pooled grad, Y_l, BN constants), then
//             dA_{l-1}[b] = W_l^T . dY_l[b]   (dgrad)   and   dW_l = sum_b dY_l[b] . act(Y_{l-1}[b])^T (wgrad)
//
// Every activation is written once and read once per direction; normalised / ReLU'd copies and
// the dense gradient of the max-pool never exist in memory.
//
// GEMM core: v_mfma_f32_32x32x2_f32 (exact f32, 157 TF peak -- the only way to keep the 1e-4 fp32
// parity bar; there is no TF32 on gfx950).  Layout is the reference's (B, C, P) with P contiguous,
// so the point dimension is the MFMA N dimension: B fragments (lane -> 32 consecutive points of one
// channel) and D stores are 128-byte coalesced runs, A fragments are the weights.  Both LDS tiles
// are k-major ([k][m], [k][n]) so every fragment read is a conflict-free ds_read_b32.
// 256 threads = 4 waves per workgroup, LDS double-buffered, global loads for chunk t+1 in flight
// during the MFMAs of chunk t (register staging, one barrier per chunk).
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>

#include "../../include/istnet_pw.h"

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: one bit per device ordinal, so that a process
// driving several GPUs (nn.DataParallel-style callers) opts in on each of them.
struct PerDeviceOnce {
  std::atomic<unsigned long long> bits{0};
  bool pending(unsigned long long& bit) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    bit = 1ull << (dev & 63);
    return !(bits.load(std::memory_order_relaxed) & bit);
  }
  void done(unsigned long long bit) { bits.fetch_or(bit, std::memory_order_relaxed); }
};

namespace {

// Build-time experiment (-DISTNET_PHASE_TIMING, tools/fwd_sk_phases.py): cycles per phase of pw_fwd_sk_kernel (thread 0 of
// every workgroup, clock64; accumulated in registers, one set of atomics at the end), read back with
// istnet_debug_phase_read.  Not in the default build.
#ifdef ISTNET_PHASE_TIMING
__device__ unsigned long long g_phase_sum[16];     // [kind][phase]
__device__ unsigned long long g_phase_wgs[2];
#define PHASE_INIT(kind) long long ph_last = clock64(); const int ph_kind = (kind); long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PHASE_T(i)                                                         \
  if (threadIdx.x == 0) {                                                  \
    const long long ph_now = clock64();                                    \
    ph_acc[i] = ph_now - ph_last;                                          \
    ph_last = ph_now;                                                      \
  }
#define PHASE_END                                                                                     \
  if (threadIdx.x == 0) {                                                                             \
    for (int ph_i = 0; ph_i < 8; ++ph_i)                                                              \
      atomicAdd(&g_phase_sum[8 * ph_kind + ph_i], (unsigned long long)ph_acc[ph_i]);                  \
    atomicAdd(&g_phase_wgs[ph_kind], 1ull);                                                           \
  }
// the same for pw_bwd_mid_kernel (tools/bwd_mid_phases.py): [template instance][role: 0 MFMA wave 0, 1 loader wave 4][phase]
__device__ unsigned long long g_mid_phase[6][2][8];
__device__ unsigned long long g_mid_wgs[6];
#define MID_T0() long long mt_last = clock64(); long long mt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MID_T(i) { const long long mt_now = clock64(); mt_acc[i] += mt_now - mt_last; mt_last = mt_now; }
#define MID_END(kind, role)                                                                                   \
  if ((threadIdx.x & 255) == 0) {                                                                             \
    for (int mt_i = 0; mt_i < 8; ++mt_i) atomicAdd(&g_mid_phase[kind][role][mt_i], (unsigned long long)mt_acc[mt_i]); \
    if (role == 0) atomicAdd(&g_mid_wgs[kind], 1ull);                                                         \
  }
#else
#define PHASE_INIT(kind)
#define PHASE_T(i)
#define PHASE_END
#define MID_T0()
#define MID_T(i)
#define MID_END(kind, role)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kKT = 16;    // K chunk of the fwd / dgrad GEMMs (channels); 32 measured slower (profiles/r01_tile_sweep.txt era, re-checked with 64x64 tiles)
constexpr int kKTW = 32;   // K chunk of the wgrad GEMM (points)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// sum over the 32 lanes of each half-wave; valid in lane 31 (lower half) and lane 63 (upper half)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// Each step is ONE v_add_f32 with the shifted value as its DPP operand (bound_ctrl: a lane without a source adds 0).  The
// builtin form compiles to v_mov 0 / v_mov_dpp / v_add per step -- 15 instructions per sum, two sums per accumulator row
// in every GEMM epilogue.  Same additions in the same order: bit-identical.  s_nop: a DPP read of a VGPR the previous
// VALU instruction wrote needs two wait states, and the hazard recogniser does not look inside inline assembly.
__device__ __forceinline__ float half_wave_sum(float v) {
  asm volatile(
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"   // lanes 31 / 63 hold the 32-lane sums
      : "+v"(v));
  return v;
}
// two sums at once: the steps of one fill the wait states of the other
__device__ __forceinline__ void half_wave_sum2(float& a, float& b) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf"
      : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = half_wave_sum(v);
  v += dpp_f<0x143, 0xc>(v);    // row_bcast:31 -> lane 63 holds the wave sum
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// four wave sums at once: with four independent chains no DPP step reads a register the previous instruction wrote, so the
// wait states disappear altogether (the same additions in the same order as four wave_sum calls: bit-identical)
__device__ __forceinline__ void wave_sum4(float& a, float& b, float& c, float& d) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
  c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
  d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));
}
// two wave sums at once (the same additions in the same order as two wave_sum calls; the DPP steps of one fill the wait states
// of the other)
__device__ __forceinline__ void wave_sum2(float& a, float& b) {
  half_wave_sum2(a, b);
  a += dpp_f<0x143, 0xc>(a);
  b += dpp_f<0x143, 0xc>(b);
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
}

// (cloud, in-cloud point) of a flattened point index q = b*P + p.  The launchers reject b*P >= 2^31, so this is a
// 32-bit division; the 64-bit one it replaces cost ~100 scalar instructions per K chunk.
__device__ __forceinline__ void split_point(long long q, int P, int& b, int& p) {
  const unsigned uq = (unsigned)q, up = (unsigned)P;
  const unsigned ub = uq / up;
  b = (int)ub;
  p = (int)(uq - ub * up);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a release / acquire fence pair around s_barrier,
// and on gfx9 the release waits for vmcnt(0): every global load in flight -- in the role-split kernels the loaders' prefetch of
// the chunk after next and the compute waves' stores -- would be drained at each chunk.
__device__ __forceinline__ void lds_barrier() {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes have landed; vmcnt / expcnt untouched
  __builtin_amdgcn_s_barrier();
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}
constexpr int kMidThreads = 512;   // role-split kernels: waves 0..3 issue MFMAs, waves 4..7 load


// MFMA C/D layout of v_mfma_f32_32x32x2_f32: reg r of lane l holds row (r&3)+8*(r>>2)+4*(l>>5), col l&31.
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Layer-0 input of a set-abstraction scale, gathered on the fly instead of materialising the grouped
// tensor (reference QueryAndGroup.forward, pointnet2_utils.py:348-358): channel k < 3 is
// xyz[idx[p]][k] - new_xyz[p / S][k], channel k >= 3 is feat[k - 3][idx[p]].
struct GatherSrc {
  const float* xyz;      // (B, n, 3)
  const float* new_xyz;  // (B, P / S, 3)
  const float* feat;     // (B, cfeat, n) or null
  const int* idx;        // (B, P) int32
  int n, S, cfeat;
};
__device__ __forceinline__ int4 gather_idx4(const GatherSrc& gs, int b, int P, int p) {
  return *reinterpret_cast<const int4*>(gs.idx + (size_t)b * P + p);
}
__device__ __forceinline__ float4 gather4(const GatherSrc& gs, int b, int k, int P, int p, int4 ii) {
  float4 v;
  if (k < 3) {
    const float* xb = gs.xyz + (size_t)b * gs.n * 3 + k;
    const float c = gs.new_xyz[((size_t)b * (P / gs.S) + p / gs.S) * 3 + k];
    v.x = xb[ii.x * 3] - c; v.y = xb[ii.y * 3] - c; v.z = xb[ii.z * 3] - c; v.w = xb[ii.w * 3] - c;
  } else {
    const float* row = gs.feat + ((size_t)b * gs.cfeat + (k - 3)) * gs.n;
    v.x = row[ii.x]; v.y = row[ii.y]; v.z = row[ii.z]; v.w = row[ii.w];
  }
  return v;
}

#ifdef ISTNET_TRACE
// Debug build only (tools/trace_fwd.py): per-workgroup timestamps (100 MHz wall clock) of the forward GEMM phases.
__device__ unsigned long long g_trace[8][8192];  // [0..3] wall clock (100 MHz), [4..7] shader clock
#define ISTNET_TRACE_MARK(i)                                                                             \
  do {                                                                                                   \
    if (threadIdx.x == 0) {                                                                              \
      const unsigned wg = blockIdx.x + gridDim.x * blockIdx.y;                                           \
      if (wg < 8192) { g_trace[i][wg] = wall_clock64(); g_trace[4 + i][wg] = clock64(); }                \
    }                                                                                                    \
  } while (0)
#else
#define ISTNET_TRACE_MARK(i)
#endif

// Layer-0 input given as several tensors instead of their channel concatenation (the IST / pose heads concatenate 3-5
// feature tensors in front of every stack, model/ist_net.py:167-171,253,322): source s holds channels
// cbeg[s] .. cbeg[s+1]-1 of the virtual input, each (B, C_s, P).  Channel counts are multiples of the K chunk, so a chunk
// lies in ONE source and the loader only swaps its base pointer per chunk.
constexpr int kMaxSrc = 6;
struct MultiSrc {
  const float* ptr[kMaxSrc];
  int cbeg[kMaxSrc + 1];
  int n;          // 0: single tensor x
};

template <int M_T, int N_T, int WM, int WN>
struct Tile {
  static constexpr int TM = M_T / (32 * WM);
  static constexpr int TN = N_T / (32 * WN);
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves per workgroup");
};

// One K chunk of MFMAs from k-major LDS tiles.  a_lds: [KT][LDA], b_lds: [KT][LDB].
// Fragments of k-step kk+1 are read while the MFMAs of k-step kk execute (register double buffer).
template <int KT, int TM, int TN, int LDA, int LDB>
__device__ __forceinline__ void mma_chunk(const float* a_lds, const float* b_lds, int a_col0, int b_col0,
                                          f32x16 (&acc)[TM][TN]) {
  const int lane = lane_id();
  const float* ap = a_lds + (lane >> 5) * LDA + a_col0 + (lane & 31);
  const float* bp = b_lds + (lane >> 5) * LDB + b_col0 + (lane & 31);
  float a[2][TM], b[2][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) a[0][tm] = ap[tm * 32];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) b[0][tn] = bp[tn * 32];
#pragma unroll
  for (int kk = 0; kk < KT / 2; ++kk) {
    const int cur = kk & 1, nxt = cur ^ 1;
    if (kk + 1 < KT / 2) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[nxt][tm] = ap[(2 * kk + 2) * LDA + tm * 32];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[nxt][tn] = bp[(2 * kk + 2) * LDB + tn * 32];
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the next step's ds_reads ahead of this step's MFMAs
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tm], b[cur][tn], acc[tm][tn], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// relu(v * s + h) on 4 lanes
__device__ __forceinline__ float4 bn_relu4(float4 v, float s, float h) {
  v.x = fmaxf(v.x * s + h, 0.f); v.y = fmaxf(v.y * s + h, 0.f);
  v.z = fmaxf(v.z * s + h, 0.f); v.w = fmaxf(v.w * s + h, 0.f);
  return v;
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ============================================================================================
// forward:  y[b][co][p] = sum_ci wt[ci][co] * act(x[b][ci][p]),  optional BN-statistics partials
// ============================================================================================
template <int M_T, int N_T, int WM, int WN>
__global__ __launch_bounds__(kThreads) void pw_fwd_kernel(
    int cin, int cout, int P, int tiles_per_cloud, const float* __restrict__ x,
    const float* __restrict__ w, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    float* __restrict__ y, float* __restrict__ part_sum, float* __restrict__ part_sq, int nt_total, int ldw,
    const float* __restrict__ c_init, const int* __restrict__ ncols, const float* __restrict__ colw, MultiSrc msrc,
    const float* __restrict__ row_init) {
  using T = Tile<M_T, N_T, WM, WN>;
  constexpr int TM = T::TM, TN = T::TN;
  // compact-column mode (csrc/sa_compact.hip): ONE point axis of static capacity P whose first *ncols columns are
  // valid, colw = per-column multiplicity in the statistics.  A tile past the valid columns only zeroes its partials.
  if (ncols != nullptr && (int)((blockIdx.x % tiles_per_cloud) * N_T) >= *ncols) {
    if (part_sum != nullptr)
      for (int rl = threadIdx.x; rl < M_T; rl += kThreads) {
        const int row = blockIdx.y * M_T + rl;
        if (row < cout) {
          part_sum[(size_t)row * nt_total + blockIdx.x] = 0.f;
          part_sq[(size_t)row * nt_total + blockIdx.x] = 0.f;
        }
      }
    return;
  }
  constexpr int NA = kKT * M_T / kThreads;        // scalar weight loads per thread per chunk
  constexpr int NB = kKT * N_T / 4 / kThreads;    // float4 activation loads per thread per chunk
  constexpr int LDA = M_T + 1;                    // w is (cout, cin): lanes walk k, odd stride spreads the banks
  static_assert(NA >= 1 && NB >= 1, "tile too small for 256 threads");
  __shared__ __attribute__((aligned(16))) float As[2][kKT][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][kKT][N_T];

  const int tid = threadIdx.x;
  const int b = blockIdx.x / tiles_per_cloud;
  const int p0 = (blockIdx.x - b * tiles_per_cloud) * N_T;
  const int m0 = blockIdx.y * M_T;
  const float* xb = x + (size_t)b * cin * P;
  const bool has_bn = in_scale != nullptr;

  // Staging is split in two so the global loads of chunk t+1 stay in flight during the MFMAs of
  // chunk t: load_chunk only issues loads (clamped addresses, no branches, no use of the data),
  // store_chunk applies zero-fill / BN+ReLU and writes LDS.
  float areg[NA];
  float4 braw[NB];
  float bsc[NB], bsh[NB];
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + kThreads * i;
      const int k = min(k0 + e % kKT, cin - 1);
      const int m = min(m0 + e / kKT, cout - 1);
      areg[i] = w[(size_t)m * ldw + k];   // ldw > cin: w is a column slice of a wider matrix
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + kThreads * i;
      {
        const int k = min(k0 + e / (N_T / 4), cin - 1), p = min(p0 + (e % (N_T / 4)) * 4, P - 4);
        if (msrc.n > 0) {          // the chunk's source tensor (uniform: chunks do not straddle sources)
          int sidx = 0;
#pragma unroll
          for (int q = 1; q < kMaxSrc; ++q) sidx += (q < msrc.n && k0 >= msrc.cbeg[q]) ? 1 : 0;
          const int cs = msrc.cbeg[sidx + 1] - msrc.cbeg[sidx];
          braw[i] = *reinterpret_cast<const float4*>(msrc.ptr[sidx] + ((size_t)b * cs + (k - msrc.cbeg[sidx])) * P + p);
        } else {
          braw[i] = *reinterpret_cast<const float4*>(xb + (size_t)k * P + p);
          if (has_bn) { bsc[i] = in_scale[k]; bsh[i] = in_shift[k]; }
        }
      }
    }
  };
  auto store_chunk = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + kThreads * i;
      const int kk = k0 + e % kKT;
      As[buf][e % kKT][e / kKT] = (kk < cin && m0 + e / kKT < cout) ? areg[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + kThreads * i;
      float4 v = braw[i];
      {
        const int row = e / (N_T / 4);
        const bool kok = k0 + row < cin;
        if (has_bn) v = bn_relu4(v, bsc[i], bsh[i]);
        if (!(kok && p0 + (e % (N_T / 4)) * 4 < P)) v = zero4();
        *reinterpret_cast<float4*>(&Bs[buf][row][(e % (N_T / 4)) * 4]) = v;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const int wv = wave_id();
  const int a_col0 = (wv / WN) * TM * 32, b_col0 = (wv % WN) * TN * 32;
  if (c_init != nullptr) {   // y = c_init + w . act(x): the accumulators start from a (b, cout, P) tensor
    const float* cb = c_init + (size_t)b * cout * P;
    const int ln = lane_id();
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + a_col0 + tm * 32 + mfma_row(r, ln);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int col = p0 + b_col0 + tn * 32 + (ln & 31);
          if (row < cout && col < P) acc[tm][tn][r] = cb[(size_t)row * P + col];
        }
      }
  }
  if (row_init != nullptr) {   // y = row_init[b][row] + w . x: a per-cloud bias (the global-mean term of the heads)
    const int ln = lane_id();
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + a_col0 + tm * 32 + mfma_row(r, ln);
        const float v = row < cout ? row_init[(size_t)b * cout + row] : 0.f;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn][r] = v;
      }
  }
  const int nchunks = (cin + kKT - 1) / kKT;
  ISTNET_TRACE_MARK(0);
  load_chunk(0);
  store_chunk(0, 0);
  __syncthreads();
  ISTNET_TRACE_MARK(1);
  for (int t = 0; t < nchunks; ++t) {
    const int buf = t & 1;
    if (t + 1 < nchunks) load_chunk((t + 1) * kKT);
    mma_chunk<kKT, TM, TN, LDA, N_T>(&As[buf][0][0], &Bs[buf][0][0], a_col0, b_col0, acc);
    if (t + 1 < nchunks) store_chunk(buf ^ 1, (t + 1) * kKT);
    __syncthreads();
  }
  ISTNET_TRACE_MARK(2);

  // ---- epilogue: store raw y, reduce per-channel sum / sum of squares of this tile ------------
  const int lane = lane_id();
  float* yb = y + (size_t)b * cout * P;
  float* red = &As[0][0][0];  // reuse LDS: [WN][M_T][2]
  const bool full_tile = (m0 + M_T <= cout) && (p0 + N_T <= P);  // workgroup-uniform
  float wcol[TN];                                                  // column multiplicities (compact-column mode)
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
    wcol[tn] = colw != nullptr ? colw[min(p0 + b_col0 + tn * 32 + (lane & 31), P - 1)] : 1.f;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row_l = a_col0 + tm * 32 + mfma_row(r, lane);
      const int row = m0 + row_l;
      float s = 0.f, q = 0.f;
      if (full_tile) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const float v = acc[tm][tn][r];
          yb[(size_t)row * P + p0 + b_col0 + tn * 32 + (lane & 31)] = v;
          s += wcol[tn] * v;
          q += wcol[tn] * v * v;
        }
      } else {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int col = p0 + b_col0 + tn * 32 + (lane & 31);
          const float v = acc[tm][tn][r];
          if (row < cout && col < P) {
            yb[(size_t)row * P + col] = v;
            s += wcol[tn] * v;
            q += wcol[tn] * v * v;
          }
        }
      }
      if (part_sum != nullptr) {
        half_wave_sum2(s, q);
        if ((lane & 31) == 31) {
          red[((wv % WN) * M_T + row_l) * 2 + 0] = s;
          red[((wv % WN) * M_T + row_l) * 2 + 1] = q;
        }
      }
    }
  }
  if (part_sum != nullptr) {
    __syncthreads();
    for (int rl = tid; rl < M_T; rl += kThreads) {
      const int row = m0 + rl;
      if (row < cout) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int wn = 0; wn < WN; ++wn) { s += red[(wn * M_T + rl) * 2 + 0]; q += red[(wn * M_T + rl) * 2 + 1]; }
        part_sum[(size_t)row * nt_total + blockIdx.x] = s;
        part_sq[(size_t)row * nt_total + blockIdx.x] = q;
      }
    }
  }
  ISTNET_TRACE_MARK(3);
}

// ============================================================================================
// forward, dense input, cin % 16 == 0, P % 128 == 0 (every layer after the first of a stack): the B operand never
// touches LDS.  In  y[co][p] = sum_ci w[co][ci] act(x[ci][p])  the points are the N index and contiguous in memory, and
// N is only a LABEL of the MFMA's columns: a lane loads a float4 of FOUR consecutive points of row ci = 2 kk + half
// (one global_load_dwordx4; the 32 lanes of a half cover 512 contiguous bytes of the row) and uses element q as the
// B operand of MFMA q, whose 32 columns are then the points {4 l + q}.  So one load feeds 4 x TMW MFMAs, the accumulators
// of q = 0..3 hold four consecutive points of an output row, and y leaves as float4 stores of 512 contiguous bytes per
// row.  Only the weights go through LDS (k-major, shared by the four waves, double-buffered in K chunks).  Per k-step
// and wave: 1 global load, TMW ds_read_b32, 8 VALU (BatchNorm + ReLU of the previous layer on the loaded values) and
// 4 TMW MFMAs -- the LDS-tiled kernel above needs 2 ds_read_b32 per MFMA at its 64 x 64 tiles.
// Wave tile: 32 TMW rows x 128 points; WM x WN waves per workgroup; grid (tiles_per_cloud * B, ceil(cout / M_WG)).
// ============================================================================================
template <int TMW, int WM, int WN, int KC>
__global__ __launch_bounds__(kThreads, 2) void pw_fwd2_kernel(
    int cin, int cout, int P, int tiles_per_cloud, const float* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, float* __restrict__ y,
    float* __restrict__ part_sum, float* __restrict__ part_sq, int nt_total) {
  static_assert(WM * WN == 4 && (KC == 16 || KC == 32), "4 waves; K chunk of 16 or 32 channels");
  constexpr int M_WG = 32 * TMW * WM, N_WG = 128 * WN;
  constexpr int LDA = M_WG + 1;                 // odd: the transposed scalar stores of a weight chunk spread over the banks
  constexpr int NA = KC * M_WG / kThreads;      // weight elements per thread per chunk
  constexpr int DEPTH = 8;                      // k-steps of B loads in flight per wave
  constexpr int kMaxCin = 1024;
  static_assert(NA >= 1 && (KC / 2) % DEPTH == 0, "chunk too small");
  __shared__ float As[2][KC][LDA];
  __shared__ float s_in[2][kMaxCin];            // scale / shift of the input layer's BatchNorm (host: cin <= kMaxCin)
  const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
  const int l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / tiles_per_cloud;
  const int p0 = (blockIdx.x - b * tiles_per_cloud) * N_WG + (wv % WN) * 128;   // this wave's 128 points
  const int m0 = blockIdx.y * M_WG, a_col0 = (wv / WN) * 32 * TMW;
  const bool has_bn = in_scale != nullptr;
  const bool live = p0 < P;                     // P % 128 == 0: a wave's tile is all inside or all outside the cloud
  const float* xb = x + (size_t)b * cin * P + (live ? p0 : 0) + 4 * l31;
  if (has_bn)
    for (int c = tid; c < cin; c += kThreads) { s_in[0][c] = in_scale[c]; s_in[1][c] = in_shift[c]; }

  float areg[NA];
  auto load_a = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + kThreads * i;
      areg[i] = w[(size_t)min(m0 + e / KC, cout - 1) * cin + k0 + e % KC];
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + kThreads * i;
      As[buf][e % KC][e / KC] = (m0 + e / KC < cout) ? areg[i] : 0.f;
    }
  };
  const int ksteps = cin / 2;
  float4 ring[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH; ++i)
    if (i < ksteps) ring[i] = *reinterpret_cast<const float4*>(xb + (size_t)(2 * i + half) * P);

  f32x16 acc[TMW][4];
#pragma unroll
  for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][q][r] = 0.f;

  const int nchunks = cin / KC;
  load_a(0);
  store_a(0);
  __syncthreads();
  // Every prefetch below is issued UNCONDITIONALLY, clamped to the last valid chunk / k-step (the tail re-reads it: a few
  // redundant L2 hits per workgroup).  With the prefetches under `if (t + 1 < nchunks)` / `if (g + DEPTH < ksteps)` the
  // compiler's wait counts at the end of every chunk merged over both branch outcomes and ended in `s_waitcnt vmcnt(0)`:
  // the whole ring of B loads was drained once per chunk (ISA of round 3), now the waits in front of the weight staging
  // stores leave the DEPTH newest loads in flight.
  for (int t = 0; t < nchunks; ++t) {
    const int buf = t & 1;
    load_a(min(t + 1, nchunks - 1) * KC);
    const float* ap = &As[buf][half][a_col0 + l31];
    float a[2][TMW];
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm) a[0][tm] = ap[tm * 32];
#pragma unroll
    for (int kk = 0; kk < KC / 2; ++kk) {
      const int g = t * (KC / 2) + kk, cur = kk & 1, nxt = cur ^ 1;
      float4 bv = ring[kk % DEPTH];
      ring[kk % DEPTH] = *reinterpret_cast<const float4*>(xb + (size_t)(2 * min(g + DEPTH, ksteps - 1) + half) * P);
      if (has_bn) bv = bn_relu4(bv, s_in[0][2 * g + half], s_in[1][2 * g + half]);
      if (kk + 1 < KC / 2) {
#pragma unroll
        for (int tm = 0; tm < TMW; ++tm) a[nxt][tm] = ap[(2 * kk + 2) * LDA + tm * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm) {
        acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tm], bv.x, acc[tm][0], 0, 0, 0);
        acc[tm][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tm], bv.y, acc[tm][1], 0, 0, 0);
        acc[tm][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tm], bv.z, acc[tm][2], 0, 0, 0);
        acc[tm][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tm], bv.w, acc[tm][3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    store_a(buf ^ 1);
    lds_barrier();
  }
  // ---- epilogue: register r of accumulator (tm, q) is output row 32 tm + mfma_row(r, lane), point 4 l31 + q ----
  float* red = &As[0][0][0];      // [WN][M_WG][2]; the loop ended on a barrier
  float* yb = y + (size_t)b * cout * P + (live ? p0 : 0) + 4 * l31;
#pragma unroll
  for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row_l = a_col0 + 32 * tm + mfma_row(r, lane), row = m0 + row_l;
      const float4 v = make_float4(acc[tm][0][r], acc[tm][1][r], acc[tm][2][r], acc[tm][3][r]);
      if (live && row < cout) *reinterpret_cast<float4*>(yb + (size_t)row * P) = v;
      if (part_sum != nullptr) {
        float s = live ? (v.x + v.y) + (v.z + v.w) : 0.f;
        float q = live ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f;
        half_wave_sum2(s, q);
        if (l31 == 31) {
          red[((wv % WN) * M_WG + row_l) * 2 + 0] = s;
          red[((wv % WN) * M_WG + row_l) * 2 + 1] = q;
        }
      }
    }
  if (part_sum != nullptr) {
    __syncthreads();
    for (int rl = tid; rl < M_WG; rl += kThreads) {
      const int row = m0 + rl;
      if (row < cout) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int wn = 0; wn < WN; ++wn) { s += red[(wn * M_WG + rl) * 2 + 0]; q += red[(wn * M_WG + rl) * 2 + 1]; }
        part_sum[(size_t)row * nt_total + blockIdx.x] = s;
        part_sq[(size_t)row * nt_total + blockIdx.x] = q;
      }
    }
  }
}

// ============================================================================================
// forward for SMALL launches (the feature-propagation levels: 4 096 - 32 768 points, K = 128 - 768): neither operand
// goes through LDS and the four waves of a workgroup split K.  With few points there are only a few hundred 64 x 64
// tiles; one tile per wave walks the whole K with one MFMA per two ds_read_b32 and a barrier every 16 channels, and the
// launch is latency, not throughput.  Here a workgroup owns ONE 32 x 128 output tile and its four waves each take every
// fourth group of 8 input channels:
//   B (activations): float4 of four consecutive points of row k -- the column relabelling of pw_fwd2_kernel;
//   A (weights):     float4 of four consecutive k of row co -- K is a dummy index too: a lane's four values serve four
//                    k-steps as long as both operands agree that (lane half, step t) of group j means k = 8j + 4 half + t;
// per group a wave issues 5 global loads for 16 MFMAs: no LDS traffic, no barrier, a quarter of the chain length.  The
// four partial accumulators meet in LDS at the end and each wave finishes 8 of the 32 rows (float4 stores of 512
// contiguous bytes, statistics partials straight from the half-wave sums).  Optional: accumulators start from a tensor
// (c_init: the interpolated term of a feature-propagation layer 0); weights may be a column slice of a wider matrix (ldw).
// grid: (B * P / 128, ceil(cout / 32)); cin % 8 == 0, cin <= 2048, P % 128 == 0.
// ============================================================================================
template <int TM>   // row blocks of 32 output channels per workgroup (1: 32 x 128 tile, 2: 64 x 128)
__global__ __launch_bounds__(kThreads, 2) void pw_fwd_sk_kernel(
    int cin, int cout, int P, int tiles_per_cloud, const float* __restrict__ x, const float* __restrict__ w, int ldw,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, const float* __restrict__ c_init,
    float* __restrict__ y, float* __restrict__ part_sum, float* __restrict__ part_sq, int nt_total,
    const float* __restrict__ zk, int m_known, const int* __restrict__ iidx, const float* __restrict__ iw) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 4 * 16 * 64];   // 64 KB: the four waves' partial sums
  const int lane = lane_id(), wv = wave_id();
  const int l31 = lane & 31, half = lane >> 5;
  // A tile is 128 consecutive points of the FLATTENED (cloud, point) axis; a lane owns four of them.  With P % 128 == 0 the
  // tile lies in one cloud; with fewer points per cloud (the coarsest propagation level: P = 64) it spans several and the
  // cloud is a per-lane quantity (P % 4 == 0: a lane's four points never straddle two clouds).  (round 5: such launches took
  // the LDS-tiled 64 x 64 kernel, 31-35 us for 1.07 GFLOP.)
  (void)tiles_per_cloud;
  const long long qpt = (long long)blockIdx.x * 128 + 4 * l31;
  const int b = (int)(qpt / P);
  const int pl = (int)(qpt - (long long)b * P);              // this lane's first point inside its cloud
  const int m0 = blockIdx.y * 32 * TM;
  const bool has_bn = in_scale != nullptr;
  const bool w_vec = (ldw & 3) == 0 && ((uintptr_t)w & 15) == 0;
  PHASE_INIT(zk != nullptr ? 1 : 0)
  // (round 4) the BatchNorm constants of a group's channels travel with the group's operands: two 16-byte loads per group and
  // lane half (the same address for 32 lanes) into registers, instead of a staging pass through LDS and a barrier in front
  // of the first operand load
  PHASE_T(0)                    // BatchNorm constants staged
  const float* xb = x + (size_t)b * cin * P + pl;
  const float* wrow[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) wrow[tm] = w + (size_t)min(m0 + 32 * tm + l31, cout - 1) * ldw + 4 * half;

  f32x16 acc[TM][4];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][q][r] = 0.f;

  const int ngroups = cin / 8;
  float4 a4[2][TM], b4[2][4], c4[2][2];
  auto load_group = [&](float4 (&a)[TM], float4 (&bq)[4], float4 (&cq)[2], int j) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float* wp = wrow[tm] + 8 * j;
      a[tm] = w_vec ? *reinterpret_cast<const float4*>(wp) : make_float4(wp[0], wp[1], wp[2], wp[3]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) bq[t] = *reinterpret_cast<const float4*>(xb + (size_t)(8 * j + 4 * half + t) * P);
    if (has_bn) {
      cq[0] = *reinterpret_cast<const float4*>(in_scale + 8 * j + 4 * half);
      cq[1] = *reinterpret_cast<const float4*>(in_shift + 8 * j + 4 * half);
    }
  };
  auto mma_group = [&](const float4 (&a)[TM], float4 (&bq)[4], const float4 (&cq)[2], int j) {
    (void)j;
    if (has_bn) {
      const float4 sc = cq[0];
      const float4 sh = cq[1];
      bq[0] = bn_relu4(bq[0], sc.x, sh.x);
      bq[1] = bn_relu4(bq[1], sc.y, sh.y);
      bq[2] = bn_relu4(bq[2], sc.z, sh.z);
      bq[3] = bn_relu4(bq[3], sc.w, sh.w);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const float av = t == 0 ? a[tm].x : (t == 1 ? a[tm].y : (t == 2 ? a[tm].z : a[tm].w));
        acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[t].x, acc[tm][0], 0, 0, 0);
        acc[tm][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[t].y, acc[tm][1], 0, 0, 0);
        acc[tm][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[t].z, acc[tm][2], 0, 0, 0);
        acc[tm][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[t].w, acc[tm][3], 0, 0, 0);
      }
  };
  // this wave's groups: wv, wv + 4, ...; two register sets, the next group's loads in flight during the MFMAs
  int j = wv;
  if (j < ngroups) load_group(a4[0], b4[0], c4[0], j);
#ifdef ISTNET_PHASE_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  PHASE_T(1)                    // first operand group arrived
#endif
  // main loop without a branch (both groups of the pair and the next pair's first exist), the last one or two groups
  // peeled: with the prefetches under `if (j + 4 < ngroups)` the compiler's wait counts merged over the branch outcomes
  // and drained the prefetch (`s_waitcnt vmcnt(0)` inside the loop, ISA of round 3); same accumulation order
  for (; j + 8 < ngroups; j += 8) {
    load_group(a4[1], b4[1], c4[1], j + 4);
    mma_group(a4[0], b4[0], c4[0], j);
    load_group(a4[0], b4[0], c4[0], j + 8);
    mma_group(a4[1], b4[1], c4[1], j + 4);
  }
  if (j < ngroups) {
    if (j + 4 < ngroups) load_group(a4[1], b4[1], c4[1], j + 4);
    mma_group(a4[0], b4[0], c4[0], j);
    if (j + 4 < ngroups) mma_group(a4[1], b4[1], c4[1], j + 4);
  }
#ifdef ISTNET_PHASE_TIMING
  asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[0][1][0]), "v"(acc[0][2][0]), "v"(acc[0][3][0]));   // MFMA results landed
  PHASE_T(2)                    // K loop of wave 0
#endif
  // ---- the four partial sums meet in LDS; wave w finishes registers 4w .. 4w + 3, i.e. rows 8w .. 8w + 7 ----
  PHASE_T(3)                    // (nothing to wait for: LDS is untouched until the partial sums are written)
  float* yb = y + (size_t)b * cout * P + pl;
  const float* cb = c_init != nullptr ? c_init + (size_t)b * cout * P + pl : nullptr;
  // feature propagation, layer 0: the accumulators start from three_interpolate(zk) (reference
  // pointnet2_utils.py:249-273), evaluated here instead of by a launch of its own: the three neighbours and weights of
  // this lane's four points, then 3 gathers per output from the (B, cout, m) product over the known points
  int nb[12];
  float nw[12];
  if (zk != nullptr) {
    const int4* ip = reinterpret_cast<const int4*>(iidx + (size_t)qpt * 3);
    const float4* wp = reinterpret_cast<const float4*>(iw + (size_t)qpt * 3);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int4 iv = ip[u];
      const float4 wv4 = wp[u];
      nb[4 * u + 0] = iv.x; nb[4 * u + 1] = iv.y; nb[4 * u + 2] = iv.z; nb[4 * u + 3] = iv.w;
      nw[4 * u + 0] = wv4.x; nw[4 * u + 1] = wv4.y; nw[4 * u + 2] = wv4.z; nw[4 * u + 3] = wv4.w;
    }
  }
  // one row block at a time through the 64 KB exchange buffer (TM = 2: two rounds)
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if (tm > 0) __syncthreads();      // every wave has read the previous block's partial sums
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) lds[((wv * 4 + q) * 16 + r) * 64 + lane] = acc[tm][q][r];
    __syncthreads();
    PHASE_T(4)                    // partial sums exchanged through LDS
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = 4 * wv + rr;
      const int row = m0 + 32 * tm + mfma_row(r, lane);
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* pr = lds + (q * 16 + r) * 64 + lane;
        v[q] = (pr[0] + pr[4 * 16 * 64]) + (pr[2 * 4 * 16 * 64] + pr[3 * 4 * 16 * 64]);
      }
      float4 o = make_float4(v[0], v[1], v[2], v[3]);
      const bool ok = row < cout;
      if (cb != nullptr && ok) {
        const float4 c0 = *reinterpret_cast<const float4*>(cb + (size_t)row * P);
        o.x += c0.x; o.y += c0.y; o.z += c0.z; o.w += c0.w;
      }
      if (zk != nullptr && ok) {
        const float* zr = zk + ((size_t)b * cout + row) * m_known;
        o.x += (zr[nb[0]] * nw[0] + zr[nb[1]] * nw[1]) + zr[nb[2]] * nw[2];
        o.y += (zr[nb[3]] * nw[3] + zr[nb[4]] * nw[4]) + zr[nb[5]] * nw[5];
        o.z += (zr[nb[6]] * nw[6] + zr[nb[7]] * nw[7]) + zr[nb[8]] * nw[8];
        o.w += (zr[nb[9]] * nw[9] + zr[nb[10]] * nw[10]) + zr[nb[11]] * nw[11];
      }
      if (ok) *reinterpret_cast<float4*>(yb + (size_t)row * P) = o;
      if (part_sum != nullptr) {
        float s1 = (o.x + o.y) + (o.z + o.w);
        float s2 = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
        half_wave_sum2(s1, s2);
        if (l31 == 31 && ok) {
          part_sum[(size_t)row * nt_total + blockIdx.x] = s1;
          part_sq[(size_t)row * nt_total + blockIdx.x] = s2;
        }
      }
    }
  }
#ifdef ISTNET_PHASE_TIMING
  PHASE_T(5)                    // epilogue: rows formed and stored (s_memtime drains this wave's older stores first)
  __builtin_amdgcn_s_waitcnt(0);
  PHASE_T(6)                    // the stores acknowledged -- s_endpgm waits for this in the plain build as well
  PHASE_END
#endif
}

// ============================================================================================
// Layer 0 of a set-abstraction scale, split by linearity.  The grouped input of point p is
// [xyz[idx[p]] - centre(p) ; feat[:, idx[p]]], so
//     y0[:, p] = W0x . (xyz[idx[p]] - centre(p)) + (W0f . feat)[:, idx[p]].
// Z = W0f . feat is a GEMM over the n source points of the cloud -- nsample * npoint / n (16..32) times fewer
// MACs than the same product over the P grouped points -- and this kernel gathers it: one thread per grouped
// point, loop over the output channels, per-channel sum / sum-of-squares partials for the BatchNorm.
// grid (ceil(P / 256), B); partials [cout][B * tiles].
// ============================================================================================
// (Round 4: staging the Z rows of 8 channels + the cloud's xyz in LDS, as interp_grad_csr_dy_lds_kernel does for its rows, was
// 10-35 % SLOWER here -- 14.1 / 23.6 / 13.4 / 21.8 / 13.1 / 21.9 us against 12.4 / 17.2 / 12.2 / 16.8 / 11.7 / 20.0 on the six
// SA2-SA4 scales: Z is 16-64 KB per cloud, L2-resident, and each gathered word is used once.)
constexpr int kGatherAddCO = 32;   // output channels per workgroup (grid.z): keeps enough waves in flight for the
                                   // deep levels, where a cloud has only a few 256-point tiles
__global__ __launch_bounds__(256) void pw_gather_add_kernel(int n, int P, int S, int cout, int ldw,
                                                            const float* __restrict__ xyz,
                                                            const float* __restrict__ new_xyz,
                                                            const int* __restrict__ idx,
                                                            const float* __restrict__ z,
                                                            const float* __restrict__ w0,
                                                            float* __restrict__ y, float* __restrict__ part_sum,
                                                            float* __restrict__ part_sq, int nt_total) {
  __shared__ float wx[kGatherAddCO * 3];          // xyz weights of this channel chunk
  __shared__ float red[4][kGatherAddCO][2];       // per-wave statistics
  const int b = blockIdx.y, tid = threadIdx.x;
  const int c0 = blockIdx.z * kGatherAddCO;
  const int nco = min(kGatherAddCO, cout - c0);
  if (tid < nco * 3) wx[tid] = w0[(size_t)(c0 + tid / 3) * ldw + (tid % 3)];
  const int p = blockIdx.x * 256 + tid;
  const bool valid = p < P;
  const int pc = valid ? p : P - 1;
  const int src = idx[(size_t)b * P + pc];
  const float* xs = xyz + ((size_t)b * n + src) * 3;
  const float* xc = new_xyz + ((size_t)b * (P / S) + pc / S) * 3;
  const float dx = xs[0] - xc[0], dy = xs[1] - xc[1], dz = xs[2] - xc[2];
  const float* zb = z + ((size_t)b * cout + c0) * n + src;
  float* yb = y + ((size_t)b * cout + c0) * P + pc;
  __syncthreads();
  const int wv = tid >> 6;
  const bool stats = part_sum != nullptr;
  // the gathers of the NEXT four channels are issued before this group's arithmetic and stores (unconditionally, clamped to
  // the last channel): eight gathers in flight instead of four, and the wait in front of a group's first use leaves the
  // stores behind it outstanding (`vmcnt(4)`) instead of draining them with the loads (`vmcnt(0)`, round 3)
  float zn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) zn[j] = z != nullptr ? zb[(size_t)min(j, nco - 1) * n] : 0.f;
  for (int co = 0; co < nco; co += 4) {
    float zv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) zv[j] = zn[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) zn[j] = z != nullptr ? zb[(size_t)min(co + 4 + j, nco - 1) * n] : 0.f;
    float vv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = min(co + j, nco - 1);
      float v = zv[j] + ((wx[3 * c] * dx + wx[3 * c + 1] * dy) + wx[3 * c + 2] * dz);
      if (valid && co + j < nco) yb[(size_t)c * P] = v;
      vv[j] = valid ? v : 0.f;
    }
    if (stats) {
      // (round 6) the eight wave sums of four channels as two interleaved groups of four: the same additions in the same order
      // (bit-identical partials) without the wait states a single DPP chain needs -- the sums were ~80 % of this kernel's issue slots
      float s0 = vv[0], q0 = vv[0] * vv[0], s1 = vv[1], q1 = vv[1] * vv[1];
      float s2 = vv[2], q2 = vv[2] * vv[2], s3 = vv[3], q3 = vv[3] * vv[3];
      wave_sum4(s0, q0, s1, q1);
      wave_sum4(s2, q2, s3, q3);
      if ((tid & 63) == 0) {
        red[wv][co][0] = s0; red[wv][co][1] = q0;
        if (co + 1 < nco) { red[wv][co + 1][0] = s1; red[wv][co + 1][1] = q1; }
        if (co + 2 < nco) { red[wv][co + 2][0] = s2; red[wv][co + 2][1] = q2; }
        if (co + 3 < nco) { red[wv][co + 3][0] = s3; red[wv][co + 3][1] = q3; }
      }
    }
  }
  if (stats) {
    __syncthreads();
    const int tile = b * gridDim.x + blockIdx.x;
    if (tid < nco) {
      const float s = (red[0][tid][0] + red[1][tid][0]) + (red[2][tid][0] + red[3][tid][0]);
      const float q = (red[0][tid][1] + red[1][tid][1]) + (red[2][tid][1] + red[3][tid][1]);
      part_sum[(size_t)(c0 + tid) * nt_total + tile] = s;
      part_sq[(size_t)(c0 + tid) * nt_total + tile] = q;
    }
  }
}

// ============================================================================================
// BN finalize (forward): partials -> mean / invstd / scale / shift, running statistics update
// ============================================================================================
// Sum two rows of nt float partials in f64 with 256 threads (fixed order: thread t takes i = t, t+256, ...;
// four independent chains keep several loads in flight), result valid on thread 0.
constexpr int kFinThreads = 256;
__device__ __forceinline__ void reduce_partials2(const float* __restrict__ pa, const float* __restrict__ pb, int nt,
                                                 double& ra, double& rb) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * kFinThreads < nt; i += 4 * kFinThreads) {
    const float x0 = pa[i], x1 = pa[i + kFinThreads], x2 = pa[i + 2 * kFinThreads], x3 = pa[i + 3 * kFinThreads];
    const float y0 = pb[i], y1 = pb[i + kFinThreads], y2 = pb[i + 2 * kFinThreads], y3 = pb[i + 3 * kFinThreads];
    a0 += (double)x0; a1 += (double)x1; a2 += (double)x2; a3 += (double)x3;
    b0 += (double)y0; b1 += (double)y1; b2 += (double)y2; b3 += (double)y3;
  }
  for (; i < nt; i += kFinThreads) { a0 += (double)pa[i]; b0 += (double)pb[i]; }
  double a = (a0 + a1) + (a2 + a3), b = (b0 + b1) + (b2 + b3);
  for (int off = 32; off >= 1; off >>= 1) {
    a += __shfl_xor(a, off);
    b += __shfl_xor(b, off);
  }
  __shared__ double sh[2][kFinThreads / 64];
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  ra = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
  rb = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
}

// out: bn[0]=scale, bn[1]=shift, bn[2]=mean, bn[3]=invstd   (each [C])
__global__ __launch_bounds__(kFinThreads) void bn_finalize_fwd_kernel(
    int C, int nt, double count, const float* __restrict__ part_sum, const float* __restrict__ part_sq,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, const float* __restrict__ momentum_p,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ bn,
    long long* __restrict__ nbt) {
  const int c = blockIdx.x;
  double s, q;
  reduce_partials2(part_sum + (size_t)c * nt, part_sq + (size_t)c * nt, nt, s, q);
  // the module's num_batches_tracked, counted by the launch that updates its running statistics (round 5: the framework's
  // multi-tensor add for all counters sat between the forward and the backward pass, behind a hop across hardware queues)
  if (threadIdx.x == 0 && c == 0 && nbt != nullptr) *nbt += 1;
  if (threadIdx.x == 0) {
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * istd;
    bn[0 * C + c] = sc;
    bn[1 * C + c] = beta[c] - (float)mean * sc;
    bn[2 * C + c] = (float)mean;
    bn[3 * C + c] = istd;
    if (running_mean != nullptr) {
      // momentum lives in device memory: a captured step replays with whatever BNMomentumScheduler.step wrote last
      // (reference utils/solver.py:91-92 steps it every iteration)
      const float momentum = *momentum_p;
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

// ============================================================================================
// BN + ReLU + max over nsample  (forward tail of an SA scale), nsample == 1 -> plain BN+ReLU
// ============================================================================================
// y (B*C, G, S) raw, out (B*C, G), arg (B*C, G) uint8 index of the first maximum
template <int S4>  // S = 4*S4 samples per group, one thread per group
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(int C, int G, const float* __restrict__ y,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           float* __restrict__ out, long long out_bstride,
                                                           uint8_t* __restrict__ arg, float* __restrict__ ymax,
                                                           int rows) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= G) return;
  for (int bc = blockIdx.y; bc < rows; bc += gridDim.y) {   // rows = B * C may exceed the 65 535 limit of grid.y
  const float s = scale[bc % C], h = shift[bc % C];
  const float4* src = reinterpret_cast<const float4*>(y + ((size_t)bc * G + g) * (S4 * 4));
  float best = -1.f;  // relu output is >= 0, so the first element always replaces this
  float raw = 0.f;    // raw y at the arg-max: lets the backward statistics skip the full activation tensor
  int besti = 0;
#pragma unroll
  for (int i = 0; i < S4; ++i) {
    const float4 v = src[i];
    const float a0 = fmaxf(v.x * s + h, 0.f), a1 = fmaxf(v.y * s + h, 0.f);
    const float a2 = fmaxf(v.z * s + h, 0.f), a3 = fmaxf(v.w * s + h, 0.f);
    if (a0 > best) { best = a0; besti = 4 * i + 0; raw = v.x; }
    if (a1 > best) { best = a1; besti = 4 * i + 1; raw = v.y; }
    if (a2 > best) { best = a2; besti = 4 * i + 2; raw = v.z; }
    if (a3 > best) { best = a3; besti = 4 * i + 3; raw = v.w; }
  }
  out[(size_t)(bc / C) * out_bstride + (size_t)(bc % C) * G + g] = best;
  arg[(size_t)bc * G + g] = (uint8_t)besti;
  if (ymax != nullptr) ymax[(size_t)bc * G + g] = raw;
  }
}
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(int C, int P4, const float* __restrict__ y,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            float* __restrict__ out, int rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P4) return;
  for (int bc = blockIdx.y; bc < rows; bc += gridDim.y) {
    const float s = scale[bc % C], h = shift[bc % C];
    float4 v = reinterpret_cast<const float4*>(y + (size_t)bc * P4 * 4)[i];
    v.x = fmaxf(v.x * s + h, 0.f); v.y = fmaxf(v.y * s + h, 0.f);
    v.z = fmaxf(v.z * s + h, 0.f); v.w = fmaxf(v.w * s + h, 0.f);
    reinterpret_cast<float4*>(out + (size_t)bc * P4 * 4)[i] = v;
  }
}

// ============================================================================================
// The LAST BatchNorm finalize of a stack folded into its consumer.  The tail kernels above are per-channel consumers: a
// workgroup of the fused forms below serves ONE channel, so it can reduce that channel's statistics partials itself
// (2 nt floats, the same fixed-order float64 reduction as bn_finalize_fwd_kernel -> bit-identical constants in every
// workgroup of the channel) and go straight on to its rows.  One launch and one kernel boundary less on the forward
// chain of every set-abstraction scale and of the encoder's last feature-propagation level.  The chunk-0 workgroup of a
// channel writes bn[4][C] and the running statistics.  grid (C, chunks): chunk j serves clouds [B j / chunks, B (j+1) / chunks).
// ============================================================================================
__device__ __forceinline__ void finalize_channel(int C, int c, int nt, double count, const float* __restrict__ part_sum,
                                                 const float* __restrict__ part_sq, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps,
                                                 const float* __restrict__ momentum_p, float* __restrict__ running_mean,
                                                 float* __restrict__ running_var, float* __restrict__ bn, bool writer,
                                                 float& sc_out, float& sh_out, long long* __restrict__ nbt = nullptr) {
  double s, q;
  reduce_partials2(part_sum + (size_t)c * nt, part_sq + (size_t)c * nt, nt, s, q);
  __shared__ float s_aff[2];
  if (threadIdx.x == 0 && writer && c == 0 && nbt != nullptr) *nbt += 1;     // see bn_finalize_fwd_kernel
  if (threadIdx.x == 0) {
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * istd;
    const float shf = beta[c] - (float)mean * sc;
    if (writer) {
      bn[0 * C + c] = sc;
      bn[1 * C + c] = shf;
      bn[2 * C + c] = (float)mean;
      bn[3 * C + c] = istd;
      if (running_mean != nullptr) {
        const float momentum = *momentum_p;
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
      }
    }
    s_aff[0] = sc;
    s_aff[1] = shf;
  }
  __syncthreads();
  sc_out = s_aff[0];
  sh_out = s_aff[1];
}

template <int S4>  // finalize + bn_relu_pool_kernel<S4>
__global__ __launch_bounds__(kFinThreads) void bn_fin_relu_pool_kernel(
    int C, int B, int G, int nt, double count, const float* __restrict__ part_sum, const float* __restrict__ part_sq,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, const float* __restrict__ momentum_p,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ bn,
    const float* __restrict__ y, float* __restrict__ out, long long out_bstride, uint8_t* __restrict__ arg,
    float* __restrict__ ymax, long long* __restrict__ nbt) {
  const int c = blockIdx.x;
  const int b0 = (int)((long long)B * blockIdx.y / gridDim.y), b1 = (int)((long long)B * (blockIdx.y + 1) / gridDim.y);
  const int total = (b1 - b0) * G;
  // Round 6: a group's samples do not depend on the constants, so the thread's FIRST group is loaded before the partials are
  // reduced (two dependent memory round trips + a barrier that the tail of every SA2-SA4 forward chain waited for before it
  // issued its own first loads).  (Also tried: the next group's loads in flight while a group is compared -- most threads own one
  // group, the clamped extra loads doubled the traffic: 17.9 -> 29.5 us; profiles/r06_finalize_loads.txt.)
  auto load_group = [&](int e, float4 (&v)[S4]) {
    const int bl = e / G, g = e - bl * G;
    const float4* src = reinterpret_cast<const float4*>(y + (((size_t)(b0 + bl) * C + c) * G + g) * (S4 * 4));
#pragma unroll
    for (int i = 0; i < S4; ++i) v[i] = src[i];
  };
  float4 cur[S4];
  if ((int)threadIdx.x < total) load_group(threadIdx.x, cur);
  float s, h;
  finalize_channel(C, c, nt, count, part_sum, part_sq, gamma, beta, eps, momentum_p, running_mean, running_var, bn,
                   blockIdx.y == 0, s, h, nbt);
  for (int e = threadIdx.x; e < total; e += kFinThreads) {
    const int bl = e / G, g = e - bl * G;
    const size_t bc = (size_t)(b0 + bl) * C + c;
    if (e != (int)threadIdx.x) load_group(e, cur);
    float best = -1.f, raw = 0.f;
    int besti = 0;
#pragma unroll
    for (int i = 0; i < S4; ++i) {
      const float4 v = cur[i];
      const float a0 = fmaxf(v.x * s + h, 0.f), a1 = fmaxf(v.y * s + h, 0.f);
      const float a2 = fmaxf(v.z * s + h, 0.f), a3 = fmaxf(v.w * s + h, 0.f);
      if (a0 > best) { best = a0; besti = 4 * i + 0; raw = v.x; }
      if (a1 > best) { best = a1; besti = 4 * i + 1; raw = v.y; }
      if (a2 > best) { best = a2; besti = 4 * i + 2; raw = v.z; }
      if (a3 > best) { best = a3; besti = 4 * i + 3; raw = v.w; }
    }
    out[(size_t)(b0 + bl) * out_bstride + (size_t)c * G + g] = best;
    arg[bc * G + g] = (uint8_t)besti;
    if (ymax != nullptr) ymax[bc * G + g] = raw;
  }
}

// finalize + bn_relu_apply_kernel (nsample == 1: the tail of a feature-propagation stack)
__global__ __launch_bounds__(kFinThreads) void bn_fin_relu_apply_kernel(
    int C, int B, int P4, int nt, double count, const float* __restrict__ part_sum, const float* __restrict__ part_sq,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, const float* __restrict__ momentum_p,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ bn,
    const float* __restrict__ y, float* __restrict__ out, long long* __restrict__ nbt) {
  const int c = blockIdx.x;
  const int b0 = (int)((long long)B * blockIdx.y / gridDim.y), b1 = (int)((long long)B * (blockIdx.y + 1) / gridDim.y);
  float s, h;
  finalize_channel(C, c, nt, count, part_sum, part_sq, gamma, beta, eps, momentum_p, running_mean, running_var, bn,
                   blockIdx.y == 0, s, h, nbt);
  for (int b = b0; b < b1; ++b) {
    const size_t row = ((size_t)b * C + c) * P4;
    for (int i = threadIdx.x; i < P4; i += kFinThreads) {
      float4 v = reinterpret_cast<const float4*>(y)[row + i];
      v.x = fmaxf(v.x * s + h, 0.f); v.y = fmaxf(v.y * s + h, 0.f);
      v.z = fmaxf(v.z * s + h, 0.f); v.w = fmaxf(v.w * s + h, 0.f);
      reinterpret_cast<float4*>(out)[row + i] = v;
    }
  }
}

// bn[4][C] of a layer whose normalisation is a fixed affine map: eval-mode BatchNorm (running statistics) or a
// plain conv bias (gamma / mean / var absent).  One launch instead of the 4-9 tiny tensor ops it replaces.
// mean over the points of relu(scale y + shift): the AdaptiveAvgPool1d(1) that ends pose_mlp2 of both estimators
// (model/ist_net.py:246,314) taken straight from the raw output of the stack's last layer -- the (B, C, N) activation is
// neither written nor read back.  One wave per (cloud, channel) row; rows (B*C), P % 4 == 0.
__global__ __launch_bounds__(256) void bn_relu_mean_kernel(int C, int P, int rows, const float* __restrict__ y,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ out) {
  const int row = blockIdx.x * 4 + wave_id(), lane = lane_id();
  if (row >= rows) return;
  const float s = scale[row % C], h = shift[row % C];
  const float4* src = reinterpret_cast<const float4*>(y + (size_t)row * P);
  float acc = 0.f;
  for (int i = lane; i < P / 4; i += 64) {
    const float4 v = src[i];
    acc += (fmaxf(v.x * s + h, 0.f) + fmaxf(v.y * s + h, 0.f)) + (fmaxf(v.z * s + h, 0.f) + fmaxf(v.w * s + h, 0.f));
  }
  acc = wave_sum(acc);
  if (lane == 0) out[row] = acc / (float)P;
}
// its adjoint as a dense gradient for the stack's backward: out[row][p] = g[row] * inv_p
__global__ __launch_bounds__(256) void expand_rows_kernel(int P4, int rows, float inv_p, const float* __restrict__ g,
                                                          float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P4) return;
  for (int row = blockIdx.y; row < rows; row += gridDim.y) {
    const float v = g[row] * inv_p;
    reinterpret_cast<float4*>(out + (size_t)row * P4 * 4)[i] = make_float4(v, v, v, v);
  }
}

__global__ __launch_bounds__(256) void affine_consts_kernel(int C, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ var, float eps,
                                                            float* __restrict__ bn) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float istd = var != nullptr ? (float)(1.0 / sqrt((double)var[c] + (double)eps)) : 1.f;
  const float m = mean != nullptr ? mean[c] : 0.f;
  const float sc = (gamma != nullptr ? gamma[c] : 1.f) * istd;
  bn[0 * C + c] = sc;
  bn[1 * C + c] = beta[c] - m * sc;
  bn[2 * C + c] = m;
  bn[3 * C + c] = istd;
}

// the same for up to 8 layers in ONE launch: the constants depend on parameters only, so a stack computes all of
// them before its first GEMM instead of one tiny launch per layer inside the dependent chain
struct AffineBatch {
  const float* gamma[8];
  const float* beta[8];
  const float* mean[8];
  const float* var[8];
  float* bn[8];
  int c[8];
  float eps[8];
  int block_begin[9];  // prefix sum of ceil(c / 256)
  int n;
};
__global__ __launch_bounds__(256) void affine_consts_multi_kernel(AffineBatch ab) {
  int l = 0;
  while (l + 1 < ab.n && (int)blockIdx.x >= ab.block_begin[l + 1]) ++l;
  const int C = ab.c[l];
  const int c = ((int)blockIdx.x - ab.block_begin[l]) * 256 + threadIdx.x;
  if (c >= C) return;
  const float* gamma = ab.gamma[l];
  const float* mean = ab.mean[l];
  const float* var = ab.var[l];
  float* bn = ab.bn[l];
  const float istd = var != nullptr ? (float)(1.0 / sqrt((double)var[c] + (double)ab.eps[l])) : 1.f;
  const float m = mean != nullptr ? mean[c] : 0.f;
  const float sc = (gamma != nullptr ? gamma[c] : 1.f) * istd;
  bn[0 * C + c] = sc;
  bn[1 * C + c] = ab.beta[l][c] - m * sc;
  bn[2 * C + c] = m;
  bn[3 * C + c] = istd;
}

__global__ __launch_bounds__(256) void affine_apply_kernel(int C, int P4, int relu, const float* __restrict__ y,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           float* __restrict__ out, int rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P4) return;
  for (int bc = blockIdx.y; bc < rows; bc += gridDim.y) {
    const float s = scale[bc % C], h = shift[bc % C];
    float4 v = reinterpret_cast<const float4*>(y + (size_t)bc * P4 * 4)[i];
    v.x = v.x * s + h; v.y = v.y * s + h; v.z = v.z * s + h; v.w = v.w * s + h;
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    reinterpret_cast<float4*>(out + (size_t)bc * P4 * 4)[i] = v;
  }
}

// ============================================================================================
// backward helpers: gradient w.r.t. the BN output,  g = dA * [relu active]
//   dense : dA (B, C, P)
//   pooled: dO (B, C, G) + arg (B, C, G): dA[p] = dO[p / S] if p % S == arg[p / S] else 0
// ============================================================================================
struct GradSrc {
  const float* dense;    // (B*C, P) or null
  const float* pooled;   // (B, C, G) or null; cloud b starts at pooled + b * pooled_bstride (a channel slice of a
                         // wider (B, Ctot, G) tensor is addressed without a copy)
  const uint8_t* arg;    // (B*C, G)
  int S;                 // nsample (pooled mode)
  long long pooled_bstride;  // elements between clouds of `pooled`; C*G when it is a plain (B, C, G) tensor
  int C;                 // channels of this layer (row = b*C + c)
};
// (b, c) are passed separately: deriving them from row = b*C + c cost a 64-bit division per loaded element
__device__ __forceinline__ float pooled_at(const GradSrc& gs, int b, int c, int G, int g) {
  return gs.pooled[(size_t)b * (size_t)gs.pooled_bstride + (size_t)c * (size_t)G + g];
}
__device__ __forceinline__ float4 load_grad4(const GradSrc& gs, int b, int c, int P, int p) {
  const size_t row = (size_t)b * gs.C + c;
  if (gs.dense != nullptr) return *reinterpret_cast<const float4*>(gs.dense + row * (size_t)P + p);
  // pooled: the 4 consecutive points p..p+3 lie in one group when S % 4 == 0
  const int G = P / gs.S;
  const int g = p / gs.S, k = p - g * gs.S;
  const float d = pooled_at(gs, b, c, G, g);
  const int a = gs.arg[row * (size_t)G + g];
  return make_float4(a == k ? d : 0.f, a == k + 1 ? d : 0.f, a == k + 2 ? d : 0.f, a == k + 3 ? d : 0.f);
}

// per-channel partial sums of y and y*y of a (B, C, P) tensor (BatchNorm statistics of a tensor that was not
// produced by one of the GEMM kernels).  grid: (chunks_per_row, C, B); partials [C][B*chunks]
__global__ __launch_bounds__(256) void pw_channel_stats_kernel(int C, int P, const float* __restrict__ y,
                                                               float* __restrict__ part_sum,
                                                               float* __restrict__ part_sq, int nt_total) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float* row = y + ((size_t)b * C + c) * P;
  const int pbeg = blockIdx.x * 4096, pend = min(pbeg + 4096, P);
  float s = 0.f, q = 0.f;
  for (int p = pbeg + threadIdx.x * 4; p < pend; p += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(row + p);
    s += (v.x + v.y) + (v.z + v.w);
    q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  __shared__ float red[4][2];
  s = wave_sum(s);
  q = wave_sum(q);
  if (lane_id() == 0) { red[wave_id()][0] = s; red[wave_id()][1] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = b * gridDim.x + blockIdx.x;
    part_sum[(size_t)c * nt_total + t] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    part_sq[(size_t)c * nt_total + t] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
  }
}

// three_interpolate (interpolate_gpu.cu:77-106) of a (B, C, m) tensor to n points TOGETHER with the BatchNorm statistics
// partials of the result: layer 0 of a feature-propagation level without skip features (the finest level: its raw output IS
// the interpolated product over the known points), where three_interpolate_kernel + pw_channel_stats_kernel were two
// launches on the forward chain.  A thread owns one point and 8 channels (the same expression and operand order as the
// stand-alone kernel: bit-identical values); the workgroup's 256 points are summed per channel in wave order.
// grid: (ceil(n / 256), ceil(C / 8), B); partials [C][B * gridDim.x]
__global__ __launch_bounds__(256) void interp_stats_kernel(int c, int m, int n, const float* __restrict__ points,
                                                           const int* __restrict__ idx, const float* __restrict__ weight,
                                                           float* __restrict__ out, float* __restrict__ part_sum,
                                                           float* __restrict__ part_sq, int nt_total) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool live = j < n;
  const int jj = live ? j : n - 1;
  const int* ix = idx + ((size_t)b * n + jj) * 3;
  const float* w = weight + ((size_t)b * n + jj) * 3;
  const int i1 = ix[0], i2 = ix[1], i3 = ix[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int c0 = blockIdx.y * 8;
  __shared__ float red[4][8][2];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int l = min(c0 + u, c - 1);
    const float* row = points + ((size_t)b * c + l) * m;
    const float v = (row[i1] * w1 + row[i2] * w2) + row[i3] * w3;
    if (live && c0 + u < c) out[((size_t)b * c + l) * n + j] = v;
    const float vv = live ? v : 0.f;
    const float s = wave_sum(vv), q = wave_sum(vv * vv);
    if (lane_id() == 0) { red[wave_id()][u][0] = s; red[wave_id()][u][1] = q; }
  }
  __syncthreads();
  if (threadIdx.x < 8 && c0 + threadIdx.x < c) {
    const int u = threadIdx.x;
    const size_t t = (size_t)(c0 + u) * nt_total + (size_t)b * gridDim.x + blockIdx.x;
    part_sum[t] = (red[0][u][0] + red[1][u][0]) + (red[2][u][0] + red[3][u][0]);
    part_sq[t] = (red[0][u][1] + red[1][u][1]) + (red[2][u][1] + red[3][u][1]);
  }
}

// dY = ca * (dA * [relu(bn(y)) > 0]) + cb + cc * y, written out (dense gradient source).  Used where the same dY
// feeds several small products (feature-propagation layer 0).  grid (ceil(P/4 / 256), B*C)
__global__ __launch_bounds__(256) void pw_dy_kernel(int C, int P4, const float* __restrict__ y,
                                                    const float* __restrict__ d, const float* __restrict__ bn,
                                                    const float* __restrict__ bwdc, float* __restrict__ out,
                                                    int rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P4) return;
  for (int bc = blockIdx.y; bc < rows; bc += gridDim.y) {
    const int c = bc % C;
    const float rs = bn[c], rh = bn[C + c], ca = bwdc[c], cb = bwdc[C + c], cc = bwdc[2 * C + c];
    const float4 v = reinterpret_cast<const float4*>(y + (size_t)bc * P4 * 4)[i];
    const float4 g = reinterpret_cast<const float4*>(d + (size_t)bc * P4 * 4)[i];
    float4 o;
    o.x = ca * ((v.x * rs + rh > 0.f) ? g.x : 0.f) + cb + cc * v.x;
    o.y = ca * ((v.y * rs + rh > 0.f) ? g.y : 0.f) + cb + cc * v.y;
    o.z = ca * ((v.z * rs + rh > 0.f) ? g.z : 0.f) + cb + cc * v.z;
    o.w = ca * ((v.w * rs + rh > 0.f) ? g.w : 0.f) + cb + cc * v.w;
    reinterpret_cast<float4*>(out + (size_t)bc * P4 * 4)[i] = o;
  }
}

// Gradient of three_interpolate (reference interpolate_gpu.cu:115-148) taken over the inverse lists of the taps, with the
// layer's dY formed on the fly from (y, dA, BatchNorm constants) -- pw_dy_kernel's expression, evaluated per gathered
// element instead of materialising dY for one consumer (the feature-propagation backward chain loses a launch; the other
// consumers of dY, the skip dgrad and the weight gradients, form it in their loaders anyway).  Same terms in the same
// order as interp_grad_csr_kernel (csrc/pn2_index_ops.hip) on the materialised tensor: bit-identical.
// grid (ceil(m / 256), ceil(C / 8), B).
constexpr int kInterpDyCH = 8;
__global__ __launch_bounds__(256) void interp_grad_csr_dy_kernel(int C, int n, int m, const float* __restrict__ y,
                                                                 const float* __restrict__ d, const float* __restrict__ bn,
                                                                 const float* __restrict__ bwdc,
                                                                 const float* __restrict__ w_all,
                                                                 const int* __restrict__ off_all,
                                                                 const int* __restrict__ ent_all,
                                                                 float* __restrict__ grad_points) {
  const int b = blockIdx.z, c0 = blockIdx.y * kInterpDyCH;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const int* off = off_all + (size_t)b * (m + 1);
  const int* ent = ent_all + (size_t)b * n * 3;
  const float* w = w_all + (size_t)b * n * 3;
  const int a = off[i], z = off[i + 1];
  const int nch = min(kInterpDyCH, C - c0);
  float rs[kInterpDyCH], rh[kInterpDyCH], ca[kInterpDyCH], cb[kInterpDyCH], cc[kInterpDyCH], sum[kInterpDyCH];
#pragma unroll
  for (int ch = 0; ch < kInterpDyCH; ++ch) {
    const int c = c0 + min(ch, nch - 1);
    rs[ch] = bn[c]; rh[ch] = bn[C + c];
    ca[ch] = bwdc[c]; cb[ch] = bwdc[C + c]; cc[ch] = bwdc[2 * C + c];
    sum[ch] = 0.f;
  }
  for (int u = a; u < z; u += 4) {
    int e[4];
    float we[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = ent[min(u + q, z - 1)];
#pragma unroll
    for (int q = 0; q < 4; ++q) we[q] = (u + q < z) ? w[e[q]] : 0.f;
    float vq[4][kInterpDyCH], gq[4][kInterpDyCH];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = e[q] / 3;
#pragma unroll
      for (int ch = 0; ch < kInterpDyCH; ++ch) {
        const size_t o = ((size_t)b * C + c0 + min(ch, nch - 1)) * n + j;
        vq[q][ch] = y[o];
        gq[q][ch] = d[o];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int ch = 0; ch < kInterpDyCH; ++ch) {
        const float v = vq[q][ch];
        const float dy = ca[ch] * ((v * rs[ch] + rh[ch] > 0.f) ? gq[q][ch] : 0.f) + cb[ch] + cc[ch] * v;
        sum[ch] += dy * we[q];
      }
  }
#pragma unroll
  for (int ch = 0; ch < kInterpDyCH; ++ch)
    if (ch < nch) grad_points[((size_t)b * C + c0 + ch) * m + i] = sum[ch];
}

// The same sums with the dY rows staged in LDS: a workgroup owns kInterpDyCH channels of one cloud, streams their (y, d)
// rows coalesced, forms dY once per element and keeps it point-major ([n][CH], 32 KB at n = 1024); the list walk then
// gathers from LDS instead of moving a 64-byte sector per 4 useful bytes (31.8 -> see profiles/r04_interp_grad_microbench.txt).
// Same terms in the same order per output: bit-identical with the kernel above.  grid (ceil(C / 8), B).
template <int NT>
__global__ __launch_bounds__(NT) void interp_grad_csr_dy_lds_kernel(int C, int n, int m, const float* __restrict__ y,
                                                                    const float* __restrict__ d,
                                                                    const float* __restrict__ bn,
                                                                    const float* __restrict__ bwdc,
                                                                    const float* __restrict__ w_all,
                                                                    const int* __restrict__ off_all,
                                                                    const int* __restrict__ ent_all,
                                                                    float* __restrict__ grad_points) {
  constexpr int CH = kInterpDyCH;
  extern __shared__ __attribute__((aligned(16))) float dyl[];     // [n][CH]
  const int b = blockIdx.y, c0 = blockIdx.x * CH;
  const int nch = min(CH, C - c0);
  float rs[CH], rh[CH], ca[CH], cb[CH], cc[CH];
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    const int c = c0 + min(ch, nch - 1);
    rs[ch] = bn[c]; rh[ch] = bn[C + c];
    ca[ch] = bwdc[c]; cb[ch] = bwdc[C + c]; cc[ch] = bwdc[2 * C + c];
  }
  for (int p = threadIdx.x * 4; p < n; p += NT * 4) {             // n % 4 == 0 (host-checked)
    float4 v[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      const size_t row = ((size_t)b * C + c0 + min(ch, nch - 1)) * n + p;
      const float4 yv = *reinterpret_cast<const float4*>(y + row);
      const float4 dv = *reinterpret_cast<const float4*>(d + row);
      v[ch].x = ca[ch] * ((yv.x * rs[ch] + rh[ch] > 0.f) ? dv.x : 0.f) + cb[ch] + cc[ch] * yv.x;
      v[ch].y = ca[ch] * ((yv.y * rs[ch] + rh[ch] > 0.f) ? dv.y : 0.f) + cb[ch] + cc[ch] * yv.y;
      v[ch].z = ca[ch] * ((yv.z * rs[ch] + rh[ch] > 0.f) ? dv.z : 0.f) + cb[ch] + cc[ch] * yv.z;
      v[ch].w = ca[ch] * ((yv.w * rs[ch] + rh[ch] > 0.f) ? dv.w : 0.f) + cb[ch] + cc[ch] * yv.w;
    }
    // LDS row of point j: (j & 3) * n / 4 + (j >> 2) -- consecutive lanes write consecutive rows (see pw_scatter_csr_kernel)
    const size_t r0 = (size_t)(p >> 2), q4 = (size_t)(n >> 2);
#pragma unroll
    for (int h = 0; h < CH / 4; ++h) {
      *reinterpret_cast<float4*>(&dyl[(r0) * CH + 4 * h]) = make_float4(v[4 * h].x, v[4 * h + 1].x, v[4 * h + 2].x, v[4 * h + 3].x);
      *reinterpret_cast<float4*>(&dyl[(q4 + r0) * CH + 4 * h]) = make_float4(v[4 * h].y, v[4 * h + 1].y, v[4 * h + 2].y, v[4 * h + 3].y);
      *reinterpret_cast<float4*>(&dyl[(2 * q4 + r0) * CH + 4 * h]) = make_float4(v[4 * h].z, v[4 * h + 1].z, v[4 * h + 2].z, v[4 * h + 3].z);
      *reinterpret_cast<float4*>(&dyl[(3 * q4 + r0) * CH + 4 * h]) = make_float4(v[4 * h].w, v[4 * h + 1].w, v[4 * h + 2].w, v[4 * h + 3].w);
    }
  }
  __syncthreads();
  const int* off = off_all + (size_t)b * (m + 1);
  const int* ent = ent_all + (size_t)b * n * 3;
  const float* w = w_all + (size_t)b * n * 3;
  for (int i = threadIdx.x; i < m; i += NT) {
    const int a = off[i], z = off[i + 1];
    float sum[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) sum[ch] = 0.f;
    for (int u = a; u < z; u += 4) {
      int e[4];
      float we[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = ent[min(u + q, z - 1)];
#pragma unroll
      for (int q = 0; q < 4; ++q) we[q] = (u + q < z) ? w[e[q]] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = e[q] / 3;
        const float4* src = reinterpret_cast<const float4*>(&dyl[((size_t)(j & 3) * (n >> 2) + (j >> 2)) * CH]);
#pragma unroll
        for (int h = 0; h < CH / 4; ++h) {
          const float4 t = src[h];
          sum[4 * h + 0] += t.x * we[q]; sum[4 * h + 1] += t.y * we[q];
          sum[4 * h + 2] += t.z * we[q]; sum[4 * h + 3] += t.w * we[q];
        }
      }
    }
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
      if (ch < nch) grad_points[((size_t)b * C + c0 + ch) * m + i] = sum[ch];
  }
}

// per-channel partial sums of g and g * y  (-> dbeta, dgamma after finalize)
// grid: (chunks_per_row, C, B); partials [C][B*chunks]
constexpr int kStatChunk = 4096;
__global__ __launch_bounds__(256) void pw_bwd_stats_kernel(int C, int P, GradSrc gs,
                                                           const float* __restrict__ y,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           float* __restrict__ part_g,
                                                           float* __restrict__ part_gy, int nt_total) {
  const int c = blockIdx.y, b = blockIdx.z;
  const size_t row = (size_t)b * C + c;
  const float s = scale[c], h = shift[c];
  const int pbeg = blockIdx.x * kStatChunk;
  const int pend = min(pbeg + kStatChunk, P);
  float sg = 0.f, sgy = 0.f;
  for (int p = pbeg + threadIdx.x * 4; p < pend; p += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(y + row * (size_t)P + p);
    const float4 d = load_grad4(gs, b, c, P, p);
    const float g0 = (v.x * s + h > 0.f) ? d.x : 0.f, g1 = (v.y * s + h > 0.f) ? d.y : 0.f;
    const float g2 = (v.z * s + h > 0.f) ? d.z : 0.f, g3 = (v.w * s + h > 0.f) ? d.w : 0.f;
    sg += (g0 + g1) + (g2 + g3);
    sgy += (g0 * v.x + g1 * v.y) + (g2 * v.z + g3 * v.w);
  }
  __shared__ float red[4][2];
  sg = wave_sum(sg);
  sgy = wave_sum(sgy);
  if (lane_id() == 0) { red[wave_id()][0] = sg; red[wave_id()][1] = sgy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = b * gridDim.x + blockIdx.x;
    part_g[(size_t)c * nt_total + t] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    part_gy[(size_t)c * nt_total + t] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
  }
}

// pw_bwd_stats_kernel (dense gradient source) + bn_finalize_bwd_kernel in ONE launch: the first layer met by the backward
// pass of a feature-propagation stack.  One workgroup of 16 waves per channel walks the B rows of that channel (float4
// along the points, float partial per wave and row chunk, accumulated over the chunks in double).
__global__ __launch_bounds__(1024) void bn_bwd_dense_finalize_kernel(
    int C, int B, int P, double count, int training, const float* __restrict__ y, const float* __restrict__ dA,
    const float* __restrict__ gamma, const float* __restrict__ bn, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ bwdc) {
  const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float s = bn[c], h = bn[C + c];
  const int P4 = P >> 2;                       // float4 per row
  const int chunks = (P4 + 63) / 64;           // 64 float4 (256 points) per wave visit
  double ag = 0.0, agy = 0.0;
  // (round 6, measured and not kept: the loads of a block of 2 / 4 / 8 visits issued before the first is consumed -- identical bits,
  //  7.0-10.2 us against 7.4-10.2 alone: the kernel is one memory round trip + the reductions + the launch, not a chain of round
  //  trips; profiles/r06_finalize_loads.txt)
  for (int t = wv; t < B * chunks; t += 16) {
    const int b = t / chunks, i = (t - b * chunks) * 64 + lane;
    float sg = 0.f, sgy = 0.f;
    if (i < P4) {
      const size_t off = ((size_t)b * C + c) * (size_t)P + 4 * (size_t)i;
      const float4 v = *reinterpret_cast<const float4*>(y + off);
      const float4 d = *reinterpret_cast<const float4*>(dA + off);
      const float g0 = (v.x * s + h > 0.f) ? d.x : 0.f, g1 = (v.y * s + h > 0.f) ? d.y : 0.f;
      const float g2 = (v.z * s + h > 0.f) ? d.z : 0.f, g3 = (v.w * s + h > 0.f) ? d.w : 0.f;
      sg = (g0 + g1) + (g2 + g3);
      sgy = (g0 * v.x + g1 * v.y) + (g2 * v.z + g3 * v.w);
    }
    ag += (double)wave_sum(sg);
    agy += (double)wave_sum(sgy);
  }
  __shared__ double sh[2][16];
  if (lane == 0) { sh[0][wv] = ag; sh[1][wv] = agy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sg = 0.0, sgy = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { sg += sh[0][k]; sgy += sh[1][k]; }
    const double mean = bn[2 * C + c], istd = bn[3 * C + c];
    const double dg = (sgy - mean * sg) * istd;
    dgamma[c] = (float)dg;
    dbeta[c] = (float)sg;
    const double gsc = (double)gamma[c] * istd;
    if (training) {
      const double c1 = sg / count, c2 = dg / count;
      bwdc[0 * C + c] = (float)gsc;
      bwdc[1 * C + c] = (float)(-gsc * c1 + gsc * mean * istd * c2);
      bwdc[2 * C + c] = (float)(-gsc * istd * c2);
    } else {
      bwdc[0 * C + c] = (float)gsc;
      bwdc[1 * C + c] = 0.f;
      bwdc[2 * C + c] = 0.f;
    }
  }
}

// Same sums when the gradient arrives through the max-pool: g is non-zero only at the arg-max of each
// group, where y is the raw maximum `ymax` saved by bn_relu_pool -- (B, C, G) reads instead of (B, C, G*S).
// grid: (C, B), one wave per row; partials [C][B]
__global__ __launch_bounds__(64) void pw_bwd_stats_pooled_kernel(int C, int G, const float* __restrict__ pooled,
                                                                 long long pooled_bstride,
                                                                 const float* __restrict__ ymax,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 float* __restrict__ part_g,
                                                                 float* __restrict__ part_gy) {
  const int c = blockIdx.x, b = blockIdx.y, B = gridDim.y;
  const float s = scale[c], h = shift[c];
  const float* d = pooled + (size_t)b * pooled_bstride + (size_t)c * G;
  const float* v = ymax + ((size_t)b * C + c) * G;
  float sg = 0.f, sgy = 0.f;
  for (int g = threadIdx.x; g < G; g += 64) {
    const float y = v[g];
    const float gr = (y * s + h > 0.f) ? d[g] : 0.f;
    sg += gr;
    sgy += gr * y;
  }
  sg = wave_sum(sg);
  sgy = wave_sum(sgy);
  if (threadIdx.x == 0) {
    part_g[(size_t)c * B + b] = sg;
    part_gy[(size_t)c * B + b] = sgy;
  }
}

// finalize: dbeta = sum g, dgamma = sum g*yhat; constants for dY = ca*g + cb + cc*y
// bn: [4][C] from the forward (scale, shift, mean, invstd); bwdc: [3][C] = ca, cb, cc
__global__ __launch_bounds__(kFinThreads) void bn_finalize_bwd_kernel(int C, int nt, double count, int training,
                                                             const float* __restrict__ part_g,
                                                             const float* __restrict__ part_gy,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ bn,
                                                             float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta,
                                                             float* __restrict__ bwdc) {
  const int c = blockIdx.x;
  double sg, sgy;
  reduce_partials2(part_g + (size_t)c * nt, part_gy + (size_t)c * nt, nt, sg, sgy);
  if (threadIdx.x == 0) {
    const double mean = bn[2 * C + c], istd = bn[3 * C + c];
    const double dg = (sgy - mean * sg) * istd;  // sum g * (y - mean) * istd
    dgamma[c] = (float)dg;
    dbeta[c] = (float)sg;
    const double gsc = (double)gamma[c] * istd;
    if (training) {
      const double c1 = sg / count, c2 = dg / count;
      bwdc[0 * C + c] = (float)gsc;
      bwdc[1 * C + c] = (float)(-gsc * c1 + gsc * mean * istd * c2);
      bwdc[2 * C + c] = (float)(-gsc * istd * c2);
    } else {  // eval-mode BN is a fixed affine map
      bwdc[0 * C + c] = (float)gsc;
      bwdc[1 * C + c] = 0.f;
      bwdc[2 * C + c] = 0.f;
    }
  }
}

// pw_bwd_stats_pooled_kernel + bn_finalize_bwd_kernel in ONE launch (the last layer of a set-abstraction scale: both are
// tiny and sit back to back on the critical chain of the backward pass).  One workgroup of 16 waves per channel; wave w
// sums the clouds w, w + 16, ...: per cloud the same float wave sum over the G groups as the stand-alone kernel, accumulated over
// clouds in double.
constexpr int kPoolFinWaves = 16;
__global__ __launch_bounds__(64 * kPoolFinWaves) void bn_bwd_pooled_finalize_kernel(
    int C, int B, int G, double count, int training, const float* __restrict__ pooled, long long pooled_bstride,
    const float* __restrict__ ymax, const float* __restrict__ gamma, const float* __restrict__ bn,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ bwdc) {
  const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float s = bn[c], h = bn[C + c];
  double ag = 0.0, agy = 0.0;
  // Round 6: two clouds per trip and up to eight 64-group chunks of each loaded before anything is consumed (32 loads in flight per
  // lane instead of 2): at G = 512 a wave walked 16 dependent memory round trips, 13-24 us at the head of every set-abstraction
  // backward chain.  Same sums in the same order: per cloud the lane's groups ascending, the wave sum, clouds ascending.
  constexpr int U = 8;
  for (int b0 = wv; b0 < B; b0 += 2 * kPoolFinWaves) {
    const int b1 = b0 + kPoolFinWaves;
    const bool has1 = b1 < B;                                   // wave-uniform
    const float* d0 = pooled + (size_t)b0 * pooled_bstride + (size_t)c * G;
    const float* v0 = ymax + ((size_t)b0 * C + c) * G;
    const float* d1 = pooled + (size_t)(has1 ? b1 : b0) * pooled_bstride + (size_t)c * G;
    const float* v1 = ymax + ((size_t)(has1 ? b1 : b0) * C + c) * G;
    float sg0 = 0.f, sgy0 = 0.f, sg1 = 0.f, sgy1 = 0.f;
    for (int gb = 0; gb < G; gb += 64 * U) {
      float y0[U], e0[U], y1[U], e1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int g = min(gb + 64 * u + lane, G - 1);           // clamped: issued unconditionally
        y0[u] = v0[g]; e0[u] = d0[g];
        y1[u] = v1[g]; e1[u] = d1[g];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (gb + 64 * u + lane < G) {
          const float gr0 = (y0[u] * s + h > 0.f) ? e0[u] : 0.f;
          sg0 += gr0;
          sgy0 += gr0 * y0[u];
          const float gr1 = (y1[u] * s + h > 0.f) ? e1[u] : 0.f;
          sg1 += gr1;
          sgy1 += gr1 * y1[u];
        }
      }
    }
    wave_sum2(sg0, sgy0);
    ag += (double)sg0;
    agy += (double)sgy0;
    if (has1) {
      wave_sum2(sg1, sgy1);
      ag += (double)sg1;
      agy += (double)sgy1;
    }
  }
  __shared__ double sh[2][kPoolFinWaves];
  if (lane == 0) { sh[0][wv] = ag; sh[1][wv] = agy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sg = 0.0, sgy = 0.0;
#pragma unroll
    for (int k = 0; k < kPoolFinWaves; ++k) { sg += sh[0][k]; sgy += sh[1][k]; }
    const double mean = bn[2 * C + c], istd = bn[3 * C + c];
    const double dg = (sgy - mean * sg) * istd;
    dgamma[c] = (float)dg;
    dbeta[c] = (float)sg;
    const double gsc = (double)gamma[c] * istd;
    if (training) {
      const double c1 = sg / count, c2 = dg / count;
      bwdc[0 * C + c] = (float)gsc;
      bwdc[1 * C + c] = (float)(-gsc * c1 + gsc * mean * istd * c2);
      bwdc[2 * C + c] = (float)(-gsc * istd * c2);
    } else {
      bwdc[0 * C + c] = (float)gsc;
      bwdc[1 * C + c] = 0.f;
      bwdc[2 * C + c] = 0.f;
    }
  }
}

// dY for 4 consecutive points of one channel, split into a pure load (issued early, nothing consumed)
// and the arithmetic (run after the MFMAs of the previous chunk).
struct DyRaw {
  float4 y, d;   // raw activation; dense gradient, or pooled gradient in d.x
  int a;         // arg-max slot (pooled mode)
};
__device__ __forceinline__ void load_dy_raw(DyRaw& r, const GradSrc& gs, const float* __restrict__ y,
                                            int b, int ch, int P, int p) {
  const size_t row = (size_t)b * gs.C + ch;
  r.y = *reinterpret_cast<const float4*>(y + row * (size_t)P + p);
  if (gs.dense != nullptr) {
    r.d = *reinterpret_cast<const float4*>(gs.dense + row * (size_t)P + p);
    r.a = 0;
  } else {
    const int G = P / gs.S, g = p / gs.S;
    r.d = make_float4(pooled_at(gs, b, ch, G, g), 0.f, 0.f, 0.f);
    r.a = gs.arg[row * (size_t)G + g];
  }
}
// ch is clamped by the caller; the five per-channel constants are tiny and cache-resident, so they are
// read here (after the MFMAs) rather than carried in registers across the chunk.
__device__ __forceinline__ float4 finish_dy(const DyRaw& r, const GradSrc& gs, int p, int ch,
                                            const float* __restrict__ bn, const float* __restrict__ bwdc,
                                            int C) {
  const float rs = bn[ch], rh = bn[C + ch];
  const float rca = bwdc[ch], rcb = bwdc[C + ch], rcc = bwdc[2 * C + ch];
  float4 d = r.d;
  if (gs.dense == nullptr) {
    const int k = p % gs.S;
    const float v = r.d.x;
    d = make_float4(r.a == k ? v : 0.f, r.a == k + 1 ? v : 0.f, r.a == k + 2 ? v : 0.f, r.a == k + 3 ? v : 0.f);
  }
  float4 o;
  o.x = rca * ((r.y.x * rs + rh > 0.f) ? d.x : 0.f) + rcb + rcc * r.y.x;
  o.y = rca * ((r.y.y * rs + rh > 0.f) ? d.y : 0.f) + rcb + rcc * r.y.y;
  o.z = rca * ((r.y.z * rs + rh > 0.f) ? d.z : 0.f) + rcb + rcc * r.y.z;
  o.w = rca * ((r.y.w * rs + rh > 0.f) ? d.w : 0.f) + rcb + rcc * r.y.w;
  return o;
}

// ============================================================================================
// Layer-0 feature gradient of a set-abstraction scale without the big dgrad GEMM.  The scatter-add of
// group_points_grad is linear and acts on the point index only, so it commutes with the channel mixing:
//     dfeat[b] = scatter(W0f^T . dY0[b]) = W0f^T . scatter(dY0[b])
// This kernel forms G[b][co][i] = sum_{p: idx[b][p] == i} dY0[b][co][p]  (Cout0 x n per cloud) directly from
// (y0, gradient source, BN constants); the remaining (C x Cout0).(Cout0 x n) product is a small GEMM.
// Runs of equal indices inside a 16-lane DPP row are pre-summed (padded ball slots repeat the first hit)
// and only the last lane of a run issues the LDS atomic.
// ============================================================================================
constexpr int kScatterCH = 4;
template <int CTRL>
__device__ __forceinline__ int dpp_row_i(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_row_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__global__ __launch_bounds__(256) void pw_scatter_dy_kernel(int cout, int n, int P, const float* __restrict__ y,
                                                            GradSrc gs, const float* __restrict__ bn,
                                                            const float* __restrict__ bwdc,
                                                            const int* __restrict__ idx_all,
                                                            float* __restrict__ out, long long out_bstride,
                                                            const float* __restrict__ xyz,
                                                            const float* __restrict__ new_xyz, int group_s,
                                                            float* __restrict__ dwx) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [CH][n]
  const int b = blockIdx.y, c0 = blockIdx.x * kScatterCH;
  const int nch = min(kScatterCH, cout - c0);
  const bool scatter = out != nullptr;   // false: only the xyz-weight partials are wanted (no LDS image, no atomics)
  if (scatter) {
    for (int i = threadIdx.x; i < nch * n; i += 256) acc[i] = 0.f;
    __syncthreads();
  }
  const int* idx = idx_all + (size_t)b * P;
  // gridDim.z > 1 (dwx-only mode) splits the points of a cloud over workgroups
  const int pchunk = ((P + (int)gridDim.z - 1) / (int)gridDim.z + 255) / 256 * 256;
  const int pbeg = blockIdx.z * pchunk;
  const int Pr = min(pbeg + pchunk, (P + 255) / 256 * 256);  // whole waves stay converged for the DPP scan
  const int G = gs.dense == nullptr ? P / gs.S : 0;
  // One 256-slot step per iteration; the loads of step t+1 (index + y + gradient of every channel) are issued
  // before step t is reduced, otherwise each step pays a full HBM round trip (the kernel was latency-bound:
  // 32 dependent steps x ~2 us at P = 8192).
  int ii_n = -1;
  float yv_n[kScatterCH], d_n[kScatterCH];
  float xr_n[3] = {0.f, 0.f, 0.f};   // dwx != nullptr: offset of the neighbour to its centroid (layer-0 xyz input)
  float wxa[kScatterCH][3];          // per-thread partial of dW0[:, 0:3] = sum_p dY0[:, p] * xrel[p]
#pragma unroll
  for (int ch = 0; ch < kScatterCH; ++ch) wxa[ch][0] = wxa[ch][1] = wxa[ch][2] = 0.f;
  auto prefetch = [&](int p) {
    const bool valid = p < P;
    const int pc = valid ? p : P - 1;
    ii_n = valid ? idx[p] : -1;
    if (dwx != nullptr) {
      const float* xs = xyz + ((size_t)b * n + idx[pc]) * 3;
      const float* xc = new_xyz + ((size_t)b * (P / group_s) + pc / group_s) * 3;
      xr_n[0] = xs[0] - xc[0]; xr_n[1] = xs[1] - xc[1]; xr_n[2] = xs[2] - xc[2];
    }
#pragma unroll
    for (int ch = 0; ch < kScatterCH; ++ch) {
      const int co = min(c0 + ch, cout - 1);
      const size_t row = (size_t)b * cout + co;
      yv_n[ch] = y[row * (size_t)P + pc];
      if (gs.dense != nullptr) {
        d_n[ch] = gs.dense[row * (size_t)P + pc];
      } else {
        const int g = pc / gs.S;
        d_n[ch] = (gs.arg[row * (size_t)G + g] == pc - g * gs.S) ? pooled_at(gs, b, co, G, g) : 0.f;
      }
    }
  };
  prefetch(pbeg + threadIdx.x);
  for (int p = pbeg + threadIdx.x; p < Pr; p += 256) {
    const bool valid = p < P;
    const int ii = ii_n;
    float yv_c[kScatterCH], d_c[kScatterCH];
#pragma unroll
    for (int ch = 0; ch < kScatterCH; ++ch) { yv_c[ch] = yv_n[ch]; d_c[ch] = d_n[ch]; }
    const float xr0 = xr_n[0], xr1 = xr_n[1], xr2 = xr_n[2];
    if (p + 256 < Pr) prefetch(p + 256);
    const int prev = dpp_row_i<0x111>(-2, ii);   // row_shr:1
    const int head0 = (prev != ii) ? 1 : 0;
    const int nxt = dpp_row_i<0x101>(-3, ii);    // row_shl:1
    const bool tail = nxt != ii;
#pragma unroll
    for (int ch = 0; ch < kScatterCH; ++ch) {
      if (ch < nch) {
        const int co = c0 + ch;
        const float yv = yv_c[ch];
        const float act = yv * bn[co] + bn[cout + co];
        float v = bwdc[co] * (act > 0.f ? d_c[ch] : 0.f) + bwdc[cout + co] + bwdc[2 * cout + co] * yv;
        if (!valid) v = 0.f;
        if (dwx != nullptr) { wxa[ch][0] += v * xr0; wxa[ch][1] += v * xr1; wxa[ch][2] += v * xr2; }
        int f = head0;
        float pv; int pf;
        pv = dpp_row_f<0x111>(v); pf = dpp_row_i<0x111>(1, f); v = f ? v : v + pv; f |= pf;
        pv = dpp_row_f<0x112>(v); pf = dpp_row_i<0x112>(1, f); v = f ? v : v + pv; f |= pf;
        pv = dpp_row_f<0x114>(v); pf = dpp_row_i<0x114>(1, f); v = f ? v : v + pv; f |= pf;
        pv = dpp_row_f<0x118>(v); pf = dpp_row_i<0x118>(1, f); v = f ? v : v + pv; f |= pf;
        if (scatter && valid && tail) atomicAdd(&acc[ch * n + ii], v);
      }
    }
  }
  __syncthreads();
  if (scatter)
    for (int i = threadIdx.x; i < nch * n; i += 256) out[(size_t)b * out_bstride + (size_t)c0 * n + i] = acc[i];
  if (dwx != nullptr) {   // workgroup sum of the xyz-weight partials -> dwx[b][co][0:3]
    __shared__ float wred[4][kScatterCH * 3];
#pragma unroll
    for (int ch = 0; ch < kScatterCH; ++ch)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float t = wave_sum(wxa[ch][k]);
        if (lane_id() == 0) wred[wave_id()][ch * 3 + k] = t;
      }
    __syncthreads();
    if (threadIdx.x < nch * 3)
      dwx[(((size_t)b * gridDim.z + blockIdx.z) * cout + c0) * 3 + threadIdx.x] = (wred[0][threadIdx.x] + wred[1][threadIdx.x]) +
                                                        (wred[2][threadIdx.x] + wred[3][threadIdx.x]);
  }
}

// The same scatter without atomics, over inverse lists of the ball-query indices (built from coordinates only,
// istnet_pn2_csr_build): source point i sums dY0 over the slots that picked it, in ascending slot order, so the
// result is deterministic, and the LDS atomic unit that bounds pw_scatter_dy_kernel (one ds_add_f32
// wave-instruction per channel per 256 slots) is out of the picture.  A workgroup owns CH channels of one cloud:
// phase 1 streams (y0, dA0) coalesced, forms dY0 and keeps it in LDS point-major ([slot][CH]); phase 2 walks the
// lists, one source point per thread at a time, reading dY0 from LDS (a gather straight from global memory moves a
// cache line per 4 useful bytes and measured 3x slower than the atomics).
// dwx[b][co][0:3] = sum_i xyz[i] * G[co][i] - sum_slots dY0[co][e] * centre[e / S].
// Optional: bn_finalize_bwd of this layer done here (the statistics partials of layer 0 come from the fused backward
// kernel of layer 1; each workgroup needs the constants of its CH channels only): one launch less on the chain.
__device__ __forceinline__ float f4_get(const float4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }
struct BwdFinArgs {
  const float* part_g;     // [cout][nt]; null: constants come from bwdc
  const float* part_gy;
  const float* gamma;
  float* dgamma;
  float* dbeta;
  float* bwdc_out;         // [3][cout], written by the workgroups of cloud 0 (later consumers: deferred weight gradients)
  double count;
  int nt;
  int training;
};
template <int CH, int NT>
__global__ __launch_bounds__(NT) void pw_scatter_csr_kernel(int cout, int n, int P, const float* __restrict__ y,
                                                             const float* __restrict__ d,
                                                             const float* __restrict__ bn,
                                                             const float* __restrict__ bwdc,
                                                             const int* __restrict__ off_all,
                                                             const int* __restrict__ ent_all,
                                                             float* __restrict__ out, long long out_bstride,
                                                             const float* __restrict__ xyz,
                                                             const float* __restrict__ new_xyz, int group_s,
                                                             float* __restrict__ dwx, BwdFinArgs fin) {
  extern __shared__ __attribute__((aligned(16))) float dy[];   // [P][CH]
  const int b = blockIdx.y, c0 = blockIdx.x * CH;
  const int nch = min(CH, cout - c0);
  float rs[CH], rh[CH], ca[CH], cb[CH], cc[CH];
  if (fin.part_g != nullptr) {
    // wave w reduces the partials of channels w, w + 4, ... of this workgroup (double, fixed order), then the constants
    // go through LDS to every thread
    float* cst = dy;                     // [CH][3] (dy is written after the barrier below)
    for (int ch = wave_id(); ch < CH; ch += NT / 64) {
      const int co = min(c0 + ch, cout - 1);
      const float* pg = fin.part_g + (size_t)co * fin.nt;
      const float* pgy = fin.part_gy + (size_t)co * fin.nt;
      double a = 0.0, c = 0.0;
      for (int i = lane_id(); i < fin.nt; i += 64) { a += (double)pg[i]; c += (double)pgy[i]; }
      for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); c += __shfl_xor(c, off); }
      if (lane_id() == 0) {
        const double mean = bn[2 * cout + co], istd = bn[3 * cout + co];
        const double dg = (c - mean * a) * istd;
        const double gsc = (double)fin.gamma[co] * istd;
        float fa, fb, fc;
        if (fin.training) {
          const double c1 = a / fin.count, c2 = dg / fin.count;
          fa = (float)gsc; fb = (float)(-gsc * c1 + gsc * mean * istd * c2); fc = (float)(-gsc * istd * c2);
        } else {
          fa = (float)gsc; fb = 0.f; fc = 0.f;
        }
        cst[ch * 3 + 0] = fa; cst[ch * 3 + 1] = fb; cst[ch * 3 + 2] = fc;
        if (b == 0 && c0 + ch < cout) {
          fin.dgamma[co] = (float)dg;
          fin.dbeta[co] = (float)a;
          fin.bwdc_out[co] = fa; fin.bwdc_out[cout + co] = fb; fin.bwdc_out[2 * cout + co] = fc;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      const int co = min(c0 + ch, cout - 1);
      rs[ch] = bn[co]; rh[ch] = bn[cout + co];
      ca[ch] = cst[ch * 3 + 0]; cb[ch] = cst[ch * 3 + 1]; cc[ch] = cst[ch * 3 + 2];
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      const int co = min(c0 + ch, cout - 1);
      rs[ch] = bn[co]; rh[ch] = bn[cout + co];
      ca[ch] = bwdc[co]; cb[ch] = bwdc[cout + co]; cc[ch] = bwdc[2 * cout + co];
    }
  }
  // ---- phase 1: dY0 of CH channel rows -> LDS (P % 4 == 0) ----
  for (int p = threadIdx.x * 4; p < P; p += NT * 4) {
    float4 v[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      const size_t row = ((size_t)b * cout + min(c0 + ch, cout - 1)) * P + p;
      const float4 yv = *reinterpret_cast<const float4*>(y + row);
      const float4 dv = *reinterpret_cast<const float4*>(d + row);
      v[ch].x = ca[ch] * ((yv.x * rs[ch] + rh[ch] > 0.f) ? dv.x : 0.f) + cb[ch] + cc[ch] * yv.x;
      v[ch].y = ca[ch] * ((yv.y * rs[ch] + rh[ch] > 0.f) ? dv.y : 0.f) + cb[ch] + cc[ch] * yv.y;
      v[ch].z = ca[ch] * ((yv.z * rs[ch] + rh[ch] > 0.f) ? dv.z : 0.f) + cb[ch] + cc[ch] * yv.z;
      v[ch].w = ca[ch] * ((yv.w * rs[ch] + rh[ch] > 0.f) ? dv.w : 0.f) + cb[ch] + cc[ch] * yv.w;
    }
    // LDS row of slot e: (e & 3) * P / 4 + (e >> 2).  A thread holds slots p .. p + 3 of CH channels; with slot-major rows
    // ([e][CH]) consecutive lanes would write 4 * CH words apart -- the same bank for every lane at CH = 8 (64-way conflict on
    // 32 scalar writes per thread and round).  With the rows of one (e & 3) class contiguous, lanes write consecutive rows:
    // whole rows as 16-byte pieces, at most 2-way conflicts (4-way at CH = 16).
    {
      const int r0 = p >> 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float* row = dy + ((size_t)k * (P >> 2) + r0) * CH;
        if constexpr (CH % 4 == 0) {
#pragma unroll
          for (int h = 0; h < CH / 4; ++h)
            *reinterpret_cast<float4*>(row + 4 * h) =
                make_float4(f4_get(v[4 * h], k), f4_get(v[4 * h + 1], k), f4_get(v[4 * h + 2], k), f4_get(v[4 * h + 3], k));
        } else if constexpr (CH == 2) {
          *reinterpret_cast<float2*>(row) = make_float2(f4_get(v[0], k), f4_get(v[1], k));
        } else {
#pragma unroll
          for (int ch = 0; ch < CH; ++ch) row[ch] = f4_get(v[ch], k);
        }
      }
    }
  }
  __syncthreads();
  // ---- phase 2: each source point sums its list; four lanes split a list into contiguous quarters (padded ball
  // rows make a few lists hundreds of slots long) and combine as (q0 + q1) + (q2 + q3): a fixed order ----
  const int* off = off_all + (size_t)b * (n + 1);
  const int* ent = ent_all + (size_t)b * P;
  const float* ctr = new_xyz + (size_t)b * (P / group_s) * 3;
  float wx[CH][3];
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) wx[ch][0] = wx[ch][1] = wx[ch][2] = 0.f;
  const int part = threadIdx.x & 3;
  const int n_round = (n + 63) / 64 * 64;            // whole quads stay converged for the DPP combine
  for (int i = threadIdx.x >> 2; i < n_round; i += NT / 4) {
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
      if (dwx != nullptr) { const float* cp = ctr + (size_t)(e / group_s) * 3; c3[0] = cp[0]; c3[1] = cp[1]; c3[2] = cp[2]; }
      const float* row = dy + ((size_t)(e & 3) * (P >> 2) + (e >> 2)) * CH;
      float rv[CH];
      if constexpr (CH % 4 == 0) {
#pragma unroll
        for (int h = 0; h < CH / 4; ++h) {
          const float4 t = *reinterpret_cast<const float4*>(row + 4 * h);
          rv[4 * h] = t.x; rv[4 * h + 1] = t.y; rv[4 * h + 2] = t.z; rv[4 * h + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) rv[ch] = row[ch];
      }
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        const float v = rv[ch];
        sum[ch] += v;
        wc[ch][0] += v * c3[0]; wc[ch][1] += v * c3[1]; wc[ch][2] += v * c3[2];
      }
    }
    // quad combine: lane ^ 1, then lane ^ 2 (quad_perm DPP); every lane of the quad ends with the same total
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      sum[ch] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sum[ch]), 0xB1, 0xf, 0xf, false));   // [1,0,3,2]
      sum[ch] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sum[ch]), 0x4E, 0xf, 0xf, false));   // [2,3,0,1]
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
  if (dwx != nullptr) {   // fixed-order workgroup sum -> dwx[b][co][0:3]
    __syncthreads();
    float* wred = dy;     // [NT / 64 waves][CH*3]
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float t = wave_sum(wx[ch][k]);
        if (lane_id() == 0) wred[wave_id() * CH * 3 + ch * 3 + k] = t;
      }
    __syncthreads();
    if (threadIdx.x < nch * 3) {
      float t = 0.f;                      // waves in groups of four, ((w0 + w1) + (w2 + w3)) + ...: the 256-thread order first
#pragma unroll
      for (int w = 0; w < NT / 64; w += 4)
        t += (wred[w * CH * 3 + threadIdx.x] + wred[(w + 1) * CH * 3 + threadIdx.x]) +
             (wred[(w + 2) * CH * 3 + threadIdx.x] + wred[(w + 3) * CH * 3 + threadIdx.x]);
      dwx[((size_t)b * cout + c0) * 3 + threadIdx.x] = t;
    }
  }
}

// ============================================================================================
// dgrad:  dx[b][m][p] = sum_co w[co][ci_off + m] * dY[b][co][p]
// ============================================================================================
template <int M_T, int N_T, int WM, int WN, bool FAST>   // FAST: every tile interior, cout % kKT == 0 (host-checked)
__global__ __launch_bounds__(kThreads) void pw_dgrad_kernel(
    int cin_total, int ci_off, int m_rows, int cout, int P, int tiles_per_cloud,
    const float* __restrict__ w, const float* __restrict__ y, GradSrc gs, const float* __restrict__ bn,
    const float* __restrict__ bwdc, float* __restrict__ dx, const float* __restrict__ y_in,
    const float* __restrict__ bn_in, float* __restrict__ part_g, float* __restrict__ part_gy, int nt_total,
    const int* __restrict__ ncols, const float* __restrict__ colw) {
  using T = Tile<M_T, N_T, WM, WN>;
  constexpr int TM = T::TM, TN = T::TN;
  constexpr int NA = kKT * M_T / kThreads;
  constexpr int NB = kKT * N_T / 4 / kThreads;
  // compact-column mode (csrc/sa_compact.hip): a tile past the valid columns only zeroes its statistics partials
  if (ncols != nullptr && (int)((blockIdx.x % tiles_per_cloud) * N_T) >= *ncols) {
    if (part_g != nullptr)
      for (int rl = threadIdx.x; rl < M_T; rl += kThreads) {
        const int row = blockIdx.y * M_T + rl;
        if (row < m_rows) {
          part_g[(size_t)row * nt_total + blockIdx.x] = 0.f;
          part_gy[(size_t)row * nt_total + blockIdx.x] = 0.f;
        }
      }
    return;
  }
  __shared__ __attribute__((aligned(16))) float As[2][kKT][M_T];
  __shared__ __attribute__((aligned(16))) float Bs[2][kKT][N_T];

  const int tid = threadIdx.x;
  const int b = blockIdx.x / tiles_per_cloud;
  const int p0 = (blockIdx.x - b * tiles_per_cloud) * N_T;
  const int m0 = blockIdx.y * M_T;
  float areg[NA];
  DyRaw braw[NB];
  // ---- staging -------------------------------------------------------------------------------------------
  // These kernels are VALU-issue-bound (PMC: ~12 VALU instructions per MFMA), and most of them were index
  // arithmetic: clamps, 64-bit address products, zero-fill selects.  An interior tile (whole M / N tile inside the
  // tensor, cout a multiple of the K chunk -- every tile of the encoder's layers) therefore takes a FAST path: every
  // address is  uniform base (scalar registers, advanced once per chunk) + a per-thread offset computed ONCE
  // before the loop, and nothing is clamped or masked.  Edge tiles keep the general code.
  constexpr bool fast = FAST;
  // per-thread invariant offsets (FAST)
  int aoff[NA], boff[NB], coff[NB], goff[NB], kslot[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int e = tid + kThreads * i;
    aoff[i] = (e / M_T) * cin_total + (e % M_T);
  }
  const int G = gs.dense == nullptr ? P / gs.S : 0;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int e = tid + kThreads * i;
    const int kl = e / (N_T / 4), pl = (e % (N_T / 4)) * 4;
    boff[i] = kl * P + pl;           // into y / dense, relative to (row k0, point p0)
    coff[i] = kl;                    // into the per-channel constant tables, relative to k0
    goff[i] = gs.dense == nullptr ? kl * G + (p0 + pl) / gs.S : 0;   // into pooled / arg, relative to row k0
    kslot[i] = gs.dense == nullptr ? (p0 + pl) % gs.S : 0;           // slot of the first of the 4 points in its group
  }
  // uniform bases (FAST), advanced by one chunk per iteration
  const float* wk = w + ci_off + m0;
  const float* yk = y + ((size_t)b * cout) * P + p0;
  const float* dk = gs.dense != nullptr ? gs.dense + ((size_t)b * cout) * P + p0 : nullptr;
  const float* pk_ = gs.dense == nullptr ? gs.pooled + (size_t)b * gs.pooled_bstride : nullptr;
  const uint8_t* ak = gs.dense == nullptr ? gs.arg + ((size_t)b * cout) * G : nullptr;

  auto load_chunk = [&](int k0) {
    if (fast) {
      const float* wc = wk + (size_t)k0 * cin_total;
      const float* yc = yk + (size_t)k0 * P;
#pragma unroll
      for (int i = 0; i < NA; ++i) areg[i] = wc[aoff[i]];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        braw[i].y = *reinterpret_cast<const float4*>(yc + boff[i]);
        if (gs.dense != nullptr) {
          braw[i].d = *reinterpret_cast<const float4*>(dk + (size_t)k0 * P + boff[i]);
          braw[i].a = 0;
        } else {
          braw[i].d = make_float4((pk_ + (size_t)k0 * G)[goff[i]], 0.f, 0.f, 0.f);
          braw[i].a = (ak + (size_t)k0 * G)[goff[i]];
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + kThreads * i;
      const int k = min(k0 + e / M_T, cout - 1), m = min(m0 + e % M_T, m_rows - 1);
      areg[i] = w[(size_t)k * cin_total + ci_off + m];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + kThreads * i;
      const int k = min(k0 + e / (N_T / 4), cout - 1), p = min(p0 + (e % (N_T / 4)) * 4, P - 4);
      load_dy_raw(braw[i], gs, y, b, k, P, p);
    }
  };
  auto store_chunk = [&](int buf, int k0) {
    if (fast) {
      const float* bnk = bn + k0;
      const float* bwk = bwdc + k0;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int e = tid + kThreads * i;
        As[buf][e / M_T][e % M_T] = areg[i];
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int e = tid + kThreads * i;
        const float rs = bnk[coff[i]], rh = bnk[cout + coff[i]];
        const float rca = bwk[coff[i]], rcb = bwk[cout + coff[i]], rcc = bwk[2 * cout + coff[i]];
        const float4 yv = braw[i].y;
        float4 d = braw[i].d;
        if (gs.dense == nullptr) {
          const int k = kslot[i], a = braw[i].a;
          const float v = d.x;
          d = make_float4(a == k ? v : 0.f, a == k + 1 ? v : 0.f, a == k + 2 ? v : 0.f, a == k + 3 ? v : 0.f);
        }
        float4 o;
        o.x = rca * ((yv.x * rs + rh > 0.f) ? d.x : 0.f) + rcb + rcc * yv.x;
        o.y = rca * ((yv.y * rs + rh > 0.f) ? d.y : 0.f) + rcb + rcc * yv.y;
        o.z = rca * ((yv.z * rs + rh > 0.f) ? d.z : 0.f) + rcb + rcc * yv.z;
        o.w = rca * ((yv.w * rs + rh > 0.f) ? d.w : 0.f) + rcb + rcc * yv.w;
        *reinterpret_cast<float4*>(&Bs[buf][e / (N_T / 4)][(e % (N_T / 4)) * 4]) = o;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + kThreads * i;
      const bool ok = (k0 + e / M_T < cout) && (m0 + e % M_T < m_rows);
      As[buf][e / M_T][e % M_T] = ok ? areg[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + kThreads * i;
      const int p = p0 + (e % (N_T / 4)) * 4;
      const bool ok = (k0 + e / (N_T / 4) < cout) && (p < P);
      float4 v = finish_dy(braw[i], gs, min(p, P - 4), min(k0 + e / (N_T / 4), cout - 1), bn, bwdc, cout);
      if (!ok) v = zero4();
      *reinterpret_cast<float4*>(&Bs[buf][e / (N_T / 4)][(e % (N_T / 4)) * 4]) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const int wv = wave_id();
  const int a_col0 = (wv / WN) * TM * 32, b_col0 = (wv % WN) * TN * 32;
  const int nchunks = (cout + kKT - 1) / kKT;
  load_chunk(0);
  store_chunk(0, 0);
  __syncthreads();
  for (int t = 0; t < nchunks; ++t) {
    const int buf = t & 1;
    if (t + 1 < nchunks) load_chunk((t + 1) * kKT);
    mma_chunk<kKT, TM, TN, M_T, N_T>(&As[buf][0][0], &Bs[buf][0][0], a_col0, b_col0, acc);
    if (t + 1 < nchunks) store_chunk(buf ^ 1, (t + 1) * kKT);
    __syncthreads();
  }
  // ---- epilogue: store dA of the producing layer; optionally reduce that layer's BN-backward statistics
  // (sum g, sum g*y with g = dA * [relu active]) so it needs no separate pass over (dA, y) ----------------
  const int lane = lane_id();
  float* dxb = dx + (size_t)b * m_rows * P;
  const bool full_tile = (m0 + M_T <= m_rows) && (p0 + N_T <= P);  // workgroup-uniform
  const bool stats = part_g != nullptr;
  const float* yin_b = stats ? y_in + (size_t)b * m_rows * P : nullptr;
  float* red = &As[0][0][0];  // reuse LDS: [WN][M_T][2]
  float wcol[TN];             // column multiplicities in the statistics (compact-column mode)
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
    wcol[tn] = colw != nullptr ? colw[min(p0 + b_col0 + tn * 32 + (lane & 31), P - 1)] : 1.f;
  if (full_tile && stats) {
    // common case: issue every y_in load of a 16-row slab before consuming any (one latency, not 16*TN)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      float yv[16][TN];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + a_col0 + tm * 32 + mfma_row(r, lane);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          yv[r][tn] = yin_b[(size_t)row * P + p0 + b_col0 + tn * 32 + (lane & 31)];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row_l = a_col0 + tm * 32 + mfma_row(r, lane);
        const int row = m0 + row_l;
        const float sc = bn_in[row], sh = bn_in[m_rows + row];
        float sg = 0.f, sgy = 0.f;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const float v = acc[tm][tn][r];
          dxb[(size_t)row * P + p0 + b_col0 + tn * 32 + (lane & 31)] = v;
          const float gq = (yv[r][tn] * sc + sh > 0.f) ? v * wcol[tn] : 0.f;
          sg += gq;
          sgy += gq * yv[r][tn];
        }
        half_wave_sum2(sg, sgy);
        if ((lane & 31) == 31) {
          red[((wv % WN) * M_T + row_l) * 2 + 0] = sg;
          red[((wv % WN) * M_T + row_l) * 2 + 1] = sgy;
        }
      }
    }
  } else {
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row_l = a_col0 + tm * 32 + mfma_row(r, lane);
      const int row = m0 + row_l;
      float sg = 0.f, sgy = 0.f;
      float sc = 0.f, sh = 0.f;
      if (stats) { const int rc = min(row, m_rows - 1); sc = bn_in[rc]; sh = bn_in[m_rows + rc]; }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int col = p0 + b_col0 + tn * 32 + (lane & 31);
        const float v = acc[tm][tn][r];
        if (full_tile || (row < m_rows && col < P)) {
          dxb[(size_t)row * P + col] = v;
          if (stats) {
            const float yv = yin_b[(size_t)row * P + col];
            const float gq = (yv * sc + sh > 0.f) ? v * wcol[tn] : 0.f;
            sg += gq;
            sgy += gq * yv;
          }
        }
      }
      if (stats) {
        half_wave_sum2(sg, sgy);
        if ((lane & 31) == 31) {
          red[((wv % WN) * M_T + row_l) * 2 + 0] = sg;
          red[((wv % WN) * M_T + row_l) * 2 + 1] = sgy;
        }
      }
    }
  }
  if (stats) {
    __syncthreads();
    for (int rl = tid; rl < M_T; rl += kThreads) {
      const int row = m0 + rl;
      if (row < m_rows) {
        float a = 0.f, c = 0.f;
#pragma unroll
        for (int wn = 0; wn < WN; ++wn) { a += red[(wn * M_T + rl) * 2 + 0]; c += red[(wn * M_T + rl) * 2 + 1]; }
        part_g[(size_t)row * nt_total + blockIdx.x] = a;
        part_gy[(size_t)row * nt_total + blockIdx.x] = c;
      }
    }
  }
}

// ============================================================================================
// dgrad for SMALL launches with a dense gradient source (the feature-propagation levels, the level-wide feature
// gradients of the set-abstraction levels): pw_fwd_sk_kernel's structure for  dA[ci][p] = sum_co w[co][ci] dY[co][p].
// One 32 x 128 output tile per workgroup, the four waves take every fourth group of 8 output channels (K), no LDS
// operands: dY is formed in registers from float4 loads of y and dA_l along the points (the column relabelling: MFMA q
// owns the points {4 l + q}) with the five per-channel constants of row k read from an LDS table; the weights are rows
// of w (a lane's input channel is contiguous in memory, so one coalesced dword per k and half).  The partial
// accumulators meet in LDS; each wave finishes 8 rows: float4 stores of dA and, when the layer below has a BatchNorm,
// the partial sums of g = dA [relu active] and g y_in over the tile (y_in read as float4 in the same layout).
// grid: (B * P / 128, ceil(m_rows / 32)); cout % 8 == 0, cout <= 2048, P % 128 == 0.
// ============================================================================================
// (TM = 2 holds 128 accumulator registers + two operand sets of 40: over the 256 a two-waves-per-SIMD bound allows -- 392 bytes of
//  scratch per lane when forced --, so that instance is built for one wave per SIMD: its launches have one workgroup per CU)
template <int TM>   // row blocks of 32 input channels per workgroup (1: 32 x 128 tile, 2: 64 x 128: dY is formed once for both)
__global__ __launch_bounds__(kThreads, TM == 1 ? 2 : 1) void pw_dgrad_sk_kernel(
    int cin_total, int ci_off, int m_rows, int cout, int P, int tiles_per_cloud, const float* __restrict__ w,
    const float* __restrict__ y, const float* __restrict__ dA, const float* __restrict__ bn,
    const float* __restrict__ bwdc, float* __restrict__ dx, const float* __restrict__ y_in,
    const float* __restrict__ bn_in, float* __restrict__ part_g, float* __restrict__ part_gy, int nt_total) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 4 * 16 * 64];   // 64 KB: constants during the loop, then the partials
  const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
  const int l31 = lane & 31, half = lane >> 5;
  (void)tiles_per_cloud;                     // tiles of the flattened (cloud, point) axis, cloud per lane: see pw_fwd_sk_kernel
  const long long qpt = (long long)blockIdx.x * 128 + 4 * l31;
  const int b = (int)(qpt / P);
  const int pl = (int)(qpt - (long long)b * P);
  const int m0 = blockIdx.y * 32 * TM;
  // [5][cout]: scale, shift of this layer's BatchNorm; ca, cb, cc of dY = ca * g + cb + cc * y
  for (int c = tid; c < cout; c += kThreads) {
    lds[c] = bn[c]; lds[2048 + c] = bn[cout + c];
    lds[4096 + c] = bwdc[c]; lds[6144 + c] = bwdc[cout + c]; lds[8192 + c] = bwdc[2 * cout + c];
  }
  __syncthreads();
  const float* yb = y + (size_t)b * cout * P + pl;
  const float* gb = dA + (size_t)b * cout * P + pl;
  const float* wcol[TM];                     // + k * cin_total: w[k][ci]
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) wcol[tm] = w + ci_off + min(m0 + 32 * tm + l31, m_rows - 1);

  f32x16 acc[TM][4];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][q][r] = 0.f;

  const int ngroups = cout / 8;
  float a4[2][TM][4];
  float4 y4[2][4], g4[2][4];
  auto load_group = [&](float (&a)[TM][4], float4 (&yq)[4], float4 (&gq)[4], int j) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = 8 * j + 4 * half + t;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm][t] = wcol[tm][(size_t)k * cin_total];
      yq[t] = *reinterpret_cast<const float4*>(yb + (size_t)k * P);
      gq[t] = *reinterpret_cast<const float4*>(gb + (size_t)k * P);
    }
  };
  auto mma_group = [&](const float (&a)[TM][4], const float4 (&yq)[4], const float4 (&gq)[4], int j) {
    const int k0 = 8 * j + 4 * half;
    const float4 rs = *reinterpret_cast<const float4*>(&lds[k0]), rh = *reinterpret_cast<const float4*>(&lds[2048 + k0]);
    const float4 ca = *reinterpret_cast<const float4*>(&lds[4096 + k0]), cb = *reinterpret_cast<const float4*>(&lds[6144 + k0]);
    const float4 cc = *reinterpret_cast<const float4*>(&lds[8192 + k0]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float s = t == 0 ? rs.x : (t == 1 ? rs.y : (t == 2 ? rs.z : rs.w));
      const float h = t == 0 ? rh.x : (t == 1 ? rh.y : (t == 2 ? rh.z : rh.w));
      const float fa = t == 0 ? ca.x : (t == 1 ? ca.y : (t == 2 ? ca.z : ca.w));
      const float fb = t == 0 ? cb.x : (t == 1 ? cb.y : (t == 2 ? cb.z : cb.w));
      const float fc = t == 0 ? cc.x : (t == 1 ? cc.y : (t == 2 ? cc.z : cc.w));
      const float4 yv = yq[t], gv = gq[t];
      float4 d;
      d.x = fa * ((yv.x * s + h > 0.f) ? gv.x : 0.f) + fb + fc * yv.x;
      d.y = fa * ((yv.y * s + h > 0.f) ? gv.y : 0.f) + fb + fc * yv.y;
      d.z = fa * ((yv.z * s + h > 0.f) ? gv.z : 0.f) + fb + fc * yv.z;
      d.w = fa * ((yv.w * s + h > 0.f) ? gv.w : 0.f) + fb + fc * yv.w;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][t], d.x, acc[tm][0], 0, 0, 0);
        acc[tm][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][t], d.y, acc[tm][1], 0, 0, 0);
        acc[tm][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][t], d.z, acc[tm][2], 0, 0, 0);
        acc[tm][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][t], d.w, acc[tm][3], 0, 0, 0);
      }
    }
  };
  int j = wv;
  if (j < ngroups) load_group(a4[0], y4[0], g4[0], j);
  // (branch-free main loop, tail peeled: see pw_fwd_sk_kernel.  The scheduling fences pin "all twelve loads of the next
  // group, THEN this group's arithmetic and MFMAs": left alone the scheduler sank the loads between the MFMAs, each a few
  // instructions in front of its first use -- four `s_waitcnt vmcnt(0)` per group in the ISA of round 3, i.e. no prefetch)
  for (; j + 8 < ngroups; j += 8) {
    load_group(a4[1], y4[1], g4[1], j + 4);
    __builtin_amdgcn_sched_barrier(0);
    mma_group(a4[0], y4[0], g4[0], j);
    __builtin_amdgcn_sched_barrier(0);
    load_group(a4[0], y4[0], g4[0], j + 8);
    __builtin_amdgcn_sched_barrier(0);
    mma_group(a4[1], y4[1], g4[1], j + 4);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (j < ngroups) {
    if (j + 4 < ngroups) load_group(a4[1], y4[1], g4[1], j + 4);
    mma_group(a4[0], y4[0], g4[0], j);
    if (j + 4 < ngroups) mma_group(a4[1], y4[1], g4[1], j + 4);
  }
  float* dxb = dx + (size_t)b * m_rows * P + pl;
  const bool stats = part_g != nullptr;
  const float* xin = stats ? y_in + (size_t)b * m_rows * P + pl : nullptr;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    __syncthreads();            // every wave is done with the constants (tm = 0) / has read the previous block's partial sums
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) lds[((wv * 4 + q) * 16 + r) * 64 + lane] = acc[tm][q][r];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = 4 * wv + rr;
      const int row = m0 + 32 * tm + mfma_row(r, lane);
      const bool ok = row < m_rows;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* pr = lds + (q * 16 + r) * 64 + lane;
        v[q] = (pr[0] + pr[4 * 16 * 64]) + (pr[2 * 4 * 16 * 64] + pr[3 * 4 * 16 * 64]);
      }
      const float4 o = make_float4(v[0], v[1], v[2], v[3]);
      if (ok) *reinterpret_cast<float4*>(dxb + (size_t)row * P) = o;
      if (stats) {
        float sg = 0.f, sgy = 0.f;
        if (ok) {
          const float4 yi = *reinterpret_cast<const float4*>(xin + (size_t)row * P);
          const float is = bn_in[row], ih = bn_in[m_rows + row];
          const float g0 = (yi.x * is + ih > 0.f) ? o.x : 0.f, g1 = (yi.y * is + ih > 0.f) ? o.y : 0.f;
          const float g2 = (yi.z * is + ih > 0.f) ? o.z : 0.f, g3 = (yi.w * is + ih > 0.f) ? o.w : 0.f;
          sg = (g0 + g1) + (g2 + g3);
          sgy = (g0 * yi.x + g1 * yi.y) + (g2 * yi.z + g3 * yi.w);
        }
        half_wave_sum2(sg, sgy);
        if (l31 == 31 && ok) {
          part_g[(size_t)row * nt_total + blockIdx.x] = sg;
          part_gy[(size_t)row * nt_total + blockIdx.x] = sgy;
        }
      }
    }
  }
}

// ============================================================================================
// wgrad:  dWpart[split][co][ci] = sum_{q in split} dY[q][co] * act(x[q][ci]),  q = b*P + p flattened
// ============================================================================================
// grid: (splits, ceil(cout / M_T), ceil(cin / N_T)); K = the points of one split.  A split is a range of
// the flattened (cloud, point) index, so few-point layers (FP levels) are not forced to one split per
// cloud; a 32-point K chunk never straddles two clouds because P % 32 == 0 is required by the launcher.
template <int M_T, int N_T, int WM, int WN, bool GATHER>  // false: tensor input, true: gathered layer-0 input (channel-major)
__global__ __launch_bounds__(kThreads) void pw_wgrad_kernel(
    int cin, int cout, int P, long long total, int split_len, const float* __restrict__ x, GatherSrc gsrc,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, const float* __restrict__ y,
    GradSrc gs, const float* __restrict__ bn, const float* __restrict__ bwdc, float* __restrict__ dw_part,
    const int* __restrict__ ncols, const float* __restrict__ colw) {
  using T = Tile<M_T, N_T, WM, WN>;
  constexpr int TM = T::TM, TN = T::TN;
  if (ncols != nullptr) {   // compact columns (csrc/sa_compact.hip): valid range and an even split of it from the device
    total = ((long long)*ncols + kKTW - 1) / kKTW * kKTW;
    const long long per = (total + gridDim.x - 1) / gridDim.x;
    split_len = (int)((per + kKTW - 1) / kKTW * kKTW);
  }
  constexpr int LDA = M_T + 1;                          // odd leading dim: transposed scalar writes spread over banks
  constexpr int LDB = N_T + 1;
  constexpr int NA = M_T * kKTW / 4 / kThreads;  // float4 (along p) per thread per chunk
  constexpr int NB = N_T * kKTW / 4 / kThreads;
  static_assert(NA >= 1 && NB >= 1, "tile too small");
  __shared__ float As[2][kKTW][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][kKTW][LDB];

  const int tid = threadIdx.x;
  const long long qbeg = (long long)blockIdx.x * split_len;
  const long long qend = max(qbeg, min(qbeg + (long long)split_len, total));
  const int m0 = blockIdx.y * M_T, n0 = blockIdx.z * N_T;
  const bool has_bn = in_scale != nullptr;

  DyRaw araw[NA];
  float4 braw[NB];
  float bsc[NB], bsh[NB];
  int4 gidx[NB];  // GATHER: neighbour indices, loaded one chunk ahead of their use
  // chunk start qk (multiple of 32, inside one cloud) -> cloud b and in-cloud point of this thread's float4
  auto load_gidx = [&](long long qk) {
    const long long qc = min(qk, total - kKTW);
    int b, pk;
    split_point(qc, P, b, pk);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + kThreads * i;
      gidx[i] = gather_idx4(gsrc, b, P, pk + (e % (kKTW / 4)) * 4);
    }
  };
  auto load_chunk = [&](long long qk) {
    const long long qc = min(qk, total - kKTW);
    int b, pk;
    split_point(qc, P, b, pk);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + kThreads * i;
      const int m = min(m0 + e / (kKTW / 4), cout - 1), p = pk + (e % (kKTW / 4)) * 4;
      load_dy_raw(araw[i], gs, y, b, m, P, p);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + kThreads * i;
      const int n = min(n0 + e / (kKTW / 4), cin - 1), p = pk + (e % (kKTW / 4)) * 4;
      if (GATHER) {
        braw[i] = gather4(gsrc, b, n, P, p, gidx[i]);
      } else {
        braw[i] = *reinterpret_cast<const float4*>(x + ((size_t)b * cin + n) * P + p);
        if (has_bn) { bsc[i] = in_scale[n]; bsh[i] = in_shift[n]; }
      }
    }
    if (GATHER) load_gidx(qk + kKTW);
  };
  auto store_chunk = [&](int buf, long long qk) {
    int b_unused, pk;
    split_point(min(qk, total - kKTW), P, b_unused, pk);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + kThreads * i;
      const int m = e / (kKTW / 4), k = (e % (kKTW / 4)) * 4;
      const bool ok = (m0 + m < cout) && (qk + k < qend);
      float4 v = finish_dy(araw[i], gs, pk + k, min(m0 + m, cout - 1), bn, bwdc, cout);
      if (colw != nullptr) {      // column multiplicities of the compact evaluation
        const float4 cw = *reinterpret_cast<const float4*>(colw + min(qk, total - kKTW) + k);
        v.x *= cw.x; v.y *= cw.y; v.z *= cw.z; v.w *= cw.w;
      }
      if (!ok) v = zero4();
      As[buf][k + 0][m] = v.x; As[buf][k + 1][m] = v.y; As[buf][k + 2][m] = v.z; As[buf][k + 3][m] = v.w;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + kThreads * i;
      float4 v = braw[i];
      const int n = e / (kKTW / 4), k = (e % (kKTW / 4)) * 4;
      const bool ok = (n0 + n < cin) && (qk + k < qend);
      if (!GATHER && has_bn) v = bn_relu4(v, bsc[i], bsh[i]);
      if (!ok) v = zero4();
      Bs[buf][k + 0][n] = v.x; Bs[buf][k + 1][n] = v.y; Bs[buf][k + 2][n] = v.z; Bs[buf][k + 3][n] = v.w;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const int wv = wave_id();
  const int a_col0 = (wv / WN) * TM * 32, b_col0 = (wv % WN) * TN * 32;
  const int nchunks = (int)((qend - qbeg + kKTW - 1) / kKTW);
  if (nchunks > 0) {
    if (GATHER) load_gidx(qbeg);
    load_chunk(qbeg);
    store_chunk(0, qbeg);
  }
  __syncthreads();
  for (int t = 0; t < nchunks; ++t) {
    const int buf = t & 1;
    if (t + 1 < nchunks) load_chunk(qbeg + (long long)(t + 1) * kKTW);
    mma_chunk<kKTW, TM, TN, LDA, LDB>(&As[buf][0][0], &Bs[buf][0][0], a_col0, b_col0, acc);
    if (t + 1 < nchunks) store_chunk(buf ^ 1, qbeg + (long long)(t + 1) * kKTW);
    __syncthreads();
  }
  const int lane = lane_id();
  float* out = dw_part + (size_t)blockIdx.x * cout * cin;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + a_col0 + tm * 32 + mfma_row(r, lane);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + b_col0 + tn * 32 + (lane & 31);
        if (row < cout && col < cin) out[(size_t)row * cin + col] = acc[tm][tn][r];
      }
    }
}

// ============================================================================================
// wgrad, dense input, cout and cin >= 64 (the FP levels, SA4, the layers pw_bwd_mid_kernel does not take): the same
// product as pw_wgrad_kernel with the machinery of pw_bwd_mid_kernel -- eight waves in two roles, one workgroup per
// CU.  Waves 4..7 load (one chunk of 32 points ahead, every slot re-issued as soon as it has been consumed), finish dY
// and write both operand tiles into the other half of a double-buffered LDS IN THE TENSORS' OWN LAYOUT ([channel][point], float4 stores; the
// k-major tiles of pw_wgrad_kernel need four scalar ds_write_b32 per float4).  Waves 0..3 own a 2 x 2 grid of
// (M_T/2) x (N_T/2) sub-tiles and read both operands as float4 ALONG THE POINTS: K is a dummy index, so a lane's four
// values serve four consecutive k-steps as long as A and B agree on the point a (lane half, step) pair means -- a
// quarter of the LDS reads and no per-step address arithmetic.  Barriers order LDS only (lds_barrier).
// grid: (splits, ceil(cout / M_T), ceil(cin / N_T)); P % 32 == 0, total = B * P, split_len % 32 == 0.
// ============================================================================================
template <int M_T, int N_T, bool POOLED>
__global__ __launch_bounds__(kMidThreads) void pw_wgrad2_kernel(
    int cin, int cout, int P, long long total, int split_len, const float* __restrict__ x,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, const float* __restrict__ y, GradSrc gs,
    const float* __restrict__ bn, const float* __restrict__ bwdc, float* __restrict__ dw_part) {
  constexpr int PT = 32, LD = PT + 4, F4 = PT / 4;
  constexpr int NA = M_T * F4 / 256, NB = N_T * F4 / 256;   // float4 per loader thread
  constexpr int TM = M_T / 64, TN = N_T / 64;               // 32x32 MFMA tiles per compute wave (2 x 2 waves)
  constexpr int TILE = (M_T + N_T) * LD;
  static_assert(NA >= 1 && NB >= 1 && TM >= 1 && TN >= 1, "tile too small");
  extern __shared__ __attribute__((aligned(16))) float wg2_lds[];
  float* const s_c = wg2_lds + 2 * TILE;        // [5][M_T]: BN scale / shift of this layer, the three BN-backward constants
  const int lane = lane_id();
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = wv >= 4;
  const int cw = wv & 3, tid = threadIdx.x & 255;
  const int l31 = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.y * M_T, n0 = blockIdx.z * N_T;
  const long long qbeg = (long long)blockIdx.x * split_len;
  const long long qend = max(qbeg, min(qbeg + (long long)split_len, total));
  const int nchunks = (int)((qend - qbeg) / PT);
  const bool has_bn = in_scale != nullptr;
  for (int c = threadIdx.x; c < M_T; c += kMidThreads) {
    const int ch = min(m0 + c, cout - 1);
    s_c[c] = bn[ch]; s_c[M_T + c] = bn[cout + ch];
    s_c[2 * M_T + c] = bwdc[ch]; s_c[3 * M_T + c] = bwdc[cout + ch]; s_c[4 * M_T + c] = bwdc[2 * cout + ch];
  }
  __syncthreads();
  if (loader) {
    // One register set, re-issued slot by slot (see pw_bwd_mid_kernel): consumption always wants the oldest outstanding
    // loads -> exact `s_waitcnt vmcnt(n - k)`.  Round 3's two whole-chunk sets read the input layer's BatchNorm scale /
    // shift from GLOBAL memory inside store_chunk (four dword loads per chunk, each followed by a wait that drained the
    // other set's prefetch: `vmcnt(3) .. vmcnt(0)` once per chunk in the ISA); a thread's rows are the same in every
    // chunk, so they are registers now.
    static_assert(NB == 2 || NB == 4, "64- or 128-wide x tile: two or four float4 per loader thread");
    float4 ry[NA];
    float4 rd[POOLED ? 1 : NA];
    float rpv[POOLED ? NA : 1];
    int rarg[POOLED ? NA : 1];
    float4 rx0, rx1, rx2, rx3;      // named, not an array (an array of them ends up in scratch memory)
    float xsc[NB], xsh[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int ch = min(n0 + (tid + 256 * i) / F4, cin - 1);
      xsc[i] = has_bn ? in_scale[ch] : 1.f;
      xsh[i] = has_bn ? in_shift[ch] : 0.f;
    }
    auto issue_y = [&](int i, int b, int pk) {
      const int e = tid + 256 * i, row = min(m0 + e / F4, cout - 1), p = pk + (e % F4) * 4;
      const size_t rowo = (size_t)b * cout + row;
      ry[i] = *reinterpret_cast<const float4*>(y + rowo * (size_t)P + p);
      if (POOLED) {
        const int G = P / gs.S, g = p / gs.S;
        rpv[i] = pooled_at(gs, b, row, G, g);
        rarg[i] = gs.arg[rowo * (size_t)G + g];
      } else {
        rd[i] = *reinterpret_cast<const float4*>(gs.dense + rowo * (size_t)P + p);
      }
    };
    auto load_x = [&](int i, int b, int pk) {
      const int e = tid + 256 * i, row = min(n0 + e / F4, cin - 1), p = pk + (e % F4) * 4;
      return *reinterpret_cast<const float4*>(x + ((size_t)b * cin + row) * P + p);
    };
    auto finish_y = [&](int i, float* As, int pk) {
      const int e = tid + 256 * i, row = e / F4, k = (e % F4) * 4;
      const float rs = s_c[row], rh = s_c[M_T + row], rca = s_c[2 * M_T + row], rcb = s_c[3 * M_T + row],
                  rcc = s_c[4 * M_T + row];
      float4 d;
      if (POOLED) {
        const int ks = (pk + k) % gs.S, a = rarg[i];
        const float pv = rpv[i];
        d = make_float4(a == ks ? pv : 0.f, a == ks + 1 ? pv : 0.f, a == ks + 2 ? pv : 0.f, a == ks + 3 ? pv : 0.f);
      } else {
        d = rd[i];
      }
      const float4 yv = ry[i];
      float4 v;
      v.x = rca * ((yv.x * rs + rh > 0.f) ? d.x : 0.f) + rcb + rcc * yv.x;
      v.y = rca * ((yv.y * rs + rh > 0.f) ? d.y : 0.f) + rcb + rcc * yv.y;
      v.z = rca * ((yv.z * rs + rh > 0.f) ? d.z : 0.f) + rcb + rcc * yv.z;
      v.w = rca * ((yv.w * rs + rh > 0.f) ? d.w : 0.f) + rcb + rcc * yv.w;
      if (m0 + row >= cout) v = zero4();
      *reinterpret_cast<float4*>(&As[row * LD + k]) = v;
    };
    auto put_x = [&](int i, float* Bs, float4 v) {
      const int e = tid + 256 * i, row = e / F4, k = (e % F4) * 4;
      if (has_bn) v = bn_relu4(v, xsc[i], xsh[i]);
      if (n0 + row >= cin) v = zero4();
      *reinterpret_cast<float4*>(&Bs[row * LD + k]) = v;
    };
    if (nchunks > 0) {
      int b, pk;
      split_point(qbeg, P, b, pk);
#pragma unroll
      for (int i = 0; i < NA; ++i) issue_y(i, b, pk);
      rx0 = load_x(0, b, pk); rx1 = load_x(1, b, pk);
      if (NB > 2) { rx2 = load_x(2, b, pk); rx3 = load_x(3, b, pk); }
    }
    for (int t = 0; t < nchunks; ++t) {
      float* As = wg2_lds + (t & 1) * TILE;
      float* Bs = As + M_T * LD;
      int b_cur, pk_cur, b_nxt, pk_nxt;
      split_point(qbeg + (long long)t * PT, P, b_cur, pk_cur);
      // the next chunk unconditionally (the last iteration re-reads its own chunk): a branch around the issue makes the
      // compiler's wait counts merge to vmcnt(0); scheduling fences keep "finish slot i, re-issue slot i" in order
      split_point(qbeg + (long long)min(t + 1, nchunks - 1) * PT, P, b_nxt, pk_nxt);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        finish_y(i, As, pk_cur);
        __builtin_amdgcn_sched_barrier(0);
        issue_y(i, b_nxt, pk_nxt);
        __builtin_amdgcn_sched_barrier(0);
      }
      put_x(0, Bs, rx0); __builtin_amdgcn_sched_barrier(0); rx0 = load_x(0, b_nxt, pk_nxt); __builtin_amdgcn_sched_barrier(0);
      put_x(1, Bs, rx1); __builtin_amdgcn_sched_barrier(0); rx1 = load_x(1, b_nxt, pk_nxt); __builtin_amdgcn_sched_barrier(0);
      if (NB > 2) {
        put_x(2, Bs, rx2); __builtin_amdgcn_sched_barrier(0); rx2 = load_x(2, b_nxt, pk_nxt); __builtin_amdgcn_sched_barrier(0);
        put_x(3, Bs, rx3); __builtin_amdgcn_sched_barrier(0); rx3 = load_x(3, b_nxt, pk_nxt); __builtin_amdgcn_sched_barrier(0);
      }
      lds_barrier();
    }
    return;
  }
  // ---------------- compute waves ----------------
  const int wm = cw >> 1, wn = cw & 1;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
  for (int t = 0; t < nchunks; ++t) {
    lds_barrier();
    const float* buf = wg2_lds + (t & 1) * TILE;
    const float* ap = buf + (32 * wm * TM + l31) * LD + 4 * half;
    const float* bp = buf + M_T * LD + (32 * wn * TN + l31) * LD + 4 * half;
    float4 a4[2][TM], b4[2][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a4[0][tm] = *reinterpret_cast<const float4*>(ap + tm * 32 * LD);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b4[0][tn] = *reinterpret_cast<const float4*>(bp + tn * 32 * LD);
#pragma unroll
    for (int j = 0; j < PT / 8; ++j) {
      const int cur = j & 1, nxt = cur ^ 1;
      if (j + 1 < PT / 8) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a4[nxt][tm] = *reinterpret_cast<const float4*>(ap + tm * 32 * LD + 8 * (j + 1));
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b4[nxt][tn] = *reinterpret_cast<const float4*>(bp + tn * 32 * LD + 8 * (j + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const float av = q == 0 ? a4[cur][tm].x : (q == 1 ? a4[cur][tm].y : (q == 2 ? a4[cur][tm].z : a4[cur][tm].w));
            const float bv = q == 0 ? b4[cur][tn].x : (q == 1 ? b4[cur][tn].y : (q == 2 ? b4[cur][tn].z : b4[cur][tn].w));
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tm][tn], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float* out = dw_part + (size_t)blockIdx.x * cout * cin;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 32 * (wm * TM + tm) + mfma_row(r, lane);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + 32 * (wn * TN + tn) + l31;
        if (row < cout && col < cin) out[(size_t)row * cin + col] = acc[tm][tn][r];
      }
    }
}

// ============================================================================================
// wgrad for small layers (cout <= 32 and cin <= 32: SA1, the 32->32 layers of SA2).  The generic kernel
// would pad the 16x16 .. 32x32 output to a 64x64 tile (16x wasted MFMA, 4x redundant loads).  Here the
// output is ONE 32x32 MFMA tile and the four waves of a workgroup split K instead of the tile: each wave
// walks its own 32-point chunks through wave-private LDS (no workgroup barrier in the loop), and the four
// accumulators are summed through LDS at the end.
// ============================================================================================
template <bool GATHER>
__global__ __launch_bounds__(kThreads) void pw_wgrad_small_kernel(
    int cin, int cout, int P, long long total, int split_len, const float* __restrict__ x, GatherSrc gsrc,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, const float* __restrict__ y,
    GradSrc gs, const float* __restrict__ bn, const float* __restrict__ bwdc, float* __restrict__ dw_part) {
  constexpr int LD = 33;
  __shared__ float lds[4][2][kKTW][LD];  // [wave][A|B][k = point][row = channel]
  const int lane = lane_id(), wv = wave_id();
  const long long qbeg = (long long)blockIdx.x * split_len;
  const long long qend = min(qbeg + (long long)split_len, total);
  const bool has_bn = in_scale != nullptr;
  float* As = &lds[wv][0][0][0];
  float* Bs = &lds[wv][1][0][0];

  // a wave's 32x32 tile of 4-point groups: 32 rows x 8 float4 = 256 items, 4 per lane
  DyRaw araw[4];
  float4 braw[4];
  float bsc[4], bsh[4];
  auto load_chunk = [&](long long qk) {
    const long long qc = min(qk, total - kKTW);
    int b, pk;
    split_point(qc, P, b, pk);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + 64 * i;
      const int row = e >> 3, p = pk + (e & 7) * 4;
      load_dy_raw(araw[i], gs, y, b, min(row, cout - 1), P, p);
      const int n = min(row, cin - 1);
      if (GATHER) {
        braw[i] = gather4(gsrc, b, n, P, p, gather_idx4(gsrc, b, P, p));
      } else {
        braw[i] = *reinterpret_cast<const float4*>(x + ((size_t)b * cin + n) * P + p);
        if (has_bn) { bsc[i] = in_scale[n]; bsh[i] = in_shift[n]; }
      }
    }
  };
  auto store_chunk = [&](long long qk) {
    int b_unused, pk;
    split_point(min(qk, total - kKTW), P, b_unused, pk);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + 64 * i;
      const int row = e >> 3, k = (e & 7) * 4;
      const bool okq = qk + k < qend;
      float4 v = finish_dy(araw[i], gs, pk + k, min(row, cout - 1), bn, bwdc, cout);
      if (!(okq && row < cout)) v = zero4();
      As[(k + 0) * LD + row] = v.x; As[(k + 1) * LD + row] = v.y;
      As[(k + 2) * LD + row] = v.z; As[(k + 3) * LD + row] = v.w;
      float4 u = braw[i];
      if (!GATHER && has_bn) u = bn_relu4(u, bsc[i], bsh[i]);
      if (!(okq && row < cin)) u = zero4();
      Bs[(k + 0) * LD + row] = u.x; Bs[(k + 1) * LD + row] = u.y;
      Bs[(k + 2) * LD + row] = u.z; Bs[(k + 3) * LD + row] = u.w;
    }
  };

  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  // chunk c of this split belongs to wave (c & 3)
  const int nchunks = (int)((qend - qbeg + kKTW - 1) / kKTW);
  int t = wv;
  if (t < nchunks) load_chunk(qbeg + (long long)t * kKTW);
  for (; t < nchunks; t += 4) {
    store_chunk(qbeg + (long long)t * kKTW);                       // wave-private LDS: no workgroup barrier
    if (t + 4 < nchunks) load_chunk(qbeg + (long long)(t + 4) * kKTW);  // in flight during the MFMAs
    __builtin_amdgcn_s_waitcnt(0xc07f);                             // lgkmcnt(0): this wave's ds_writes landed
    mma_chunk<kKTW, 1, 1, LD, LD>(As, Bs, 0, 0, acc);
  }
  // cross-wave sum through LDS (reuse: [4][32*32] floats fit in the staging area)
  __syncthreads();
  float* red = &lds[0][0][0][0];
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wv * 1024 + mfma_row(r, lane) * 32 + (lane & 31)] = acc[0][0][r];
  __syncthreads();
  float* out = dw_part + (size_t)blockIdx.x * cout * cin;
  for (int i = threadIdx.x; i < 1024; i += kThreads) {
    const int row = i >> 5, col = i & 31;
    if (row < cout && col < cin)
      out[(size_t)row * cin + col] = (red[i] + red[1024 + i]) + (red[2048 + i] + red[3072 + i]);
  }
}

// ============================================================================================
// Fused backward of a small layer (cin <= 32, cout <= 32, not the first of its stack): ONE pass over
// (y_l, gradient source, y_{l-1}) produces
//     dA_{l-1} = W^T . dY_l                      (what pw_dgrad_kernel would write),
//     partials of sum g, sum g*y_{l-1}           (its fused BN-backward statistics, g = dA * [relu active]),
//     a split-K partial of dW = dY_l . act(y_{l-1})^T   (what pw_wgrad_small_kernel would write).
// These layers (SA1, SA2 layer 1) are HBM-bound: dgrad and wgrad each read y_l and the gradient source, so one
// pass saves ~45 % of the bytes.  Same structure as pw_wgrad_small_kernel: the four waves of a workgroup walk
// their own 32-point chunks through wave-private LDS.  Per chunk and wave: dY [pt][co] and RAW y_{l-1} [pt][ci]
// tiles (odd stride, so both the k-major and the transposed fragment reads are conflict-free), 16 MFMAs for dW
// (K = points) and 16 for dA (K = cout, A operand = W held in registers).
// ============================================================================================
template <bool POOLED>   // gradient source: dense dA (B, C, P), or the max-pool's (dO, arg) pair (the last layer of a scale)
#ifndef ISTNET_BWD_SMALL_WAVES
#define ISTNET_BWD_SMALL_WAVES 2   // waves per SIMD the register allocation targets (A/B: 3 spills 26 registers)
#endif
__global__ __launch_bounds__(kThreads, ISTNET_BWD_SMALL_WAVES) void pw_bwd_small_kernel(
    int cin, int cout, int P, long long total, int split_len, const float* __restrict__ w,
    const float* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const float* __restrict__ y, GradSrc gs, const float* __restrict__ bn, const float* __restrict__ bwdc,
    float* __restrict__ dx, float* __restrict__ part_g, float* __restrict__ part_gy, int nt_total,
    float* __restrict__ dw_part, const int* __restrict__ ncols, const float* __restrict__ colw) {
  constexpr int LD = 33;
  __shared__ float lds[4][2][kKTW][LD];  // [wave][dY | raw x][k = point][row = channel]
  __shared__ float s_in[2][32];          // BN constants of the input layer
  __shared__ float s_k[5][32];           // this layer's BN scale / shift and the three BN-backward constants per output channel:
                                         // read from LDS in store_chunk -- as global loads inside the chunk loop (round 3) every
                                         // one of them was followed by a wait that drained the next chunk's prefetch
  __shared__ float s_w[4][kKTW];         // column multiplicities of a wave's chunk (compact-column mode)
  const int lane = lane_id(), wv = wave_id();
  if (ncols != nullptr) {                // compact columns: the valid range and an even split of it come from the device
    total = ((long long)*ncols + kKTW - 1) / kKTW * kKTW;
    const long long per = (total + gridDim.x - 1) / gridDim.x;
    split_len = (int)((per + 4 * kKTW - 1) / (4 * kKTW) * (4 * kKTW));
  }
  const bool weighted = colw != nullptr;
  const long long qbeg = (long long)blockIdx.x * split_len;
  const long long qend = max(qbeg, min(qbeg + (long long)split_len, total));
  float* As = &lds[wv][0][0][0];
  float* Bs = &lds[wv][1][0][0];
  if (threadIdx.x < 32) {
    const int c = min((int)threadIdx.x, cin - 1);
    s_in[0][threadIdx.x] = in_scale[c];
    s_in[1][threadIdx.x] = in_shift[c];
    const int co = min((int)threadIdx.x, cout - 1);
    s_k[0][threadIdx.x] = bn[co]; s_k[1][threadIdx.x] = bn[cout + co];
    s_k[2][threadIdx.x] = bwdc[co]; s_k[3][threadIdx.x] = bwdc[cout + co]; s_k[4][threadIdx.x] = bwdc[2 * cout + co];
  }
  // B operand of the dgrad MFMAs: B[k = co][j = ci] = w[co][ci], co = 2*kk + (lane >> 5), ci = lane & 31
  float wfrag[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int co = 2 * kk + (lane >> 5), ci = lane & 31;
    wfrag[kk] = (co < cout && ci < cin) ? w[(size_t)co * cin + ci] : 0.f;
  }
  __syncthreads();
  const float bsc = s_in[0][lane & 31], bsh = s_in[1][lane & 31];  // BN constants of input channel ci = lane & 31

  // Register budget: the kernel is HBM-bound and lives on waves in flight, so everything that is carried across a
  // chunk is kept small -- the loads of the next chunk hold only the fields their gradient mode uses (dense: y + dA;
  // pooled: y + one pooled value + its arg-max slot), and dA^T is what the dgrad MFMAs produce (rows = points,
  // column = input channel of the lane): its BN-backward statistics are then two scalars per lane instead of two
  // per accumulator register, and dA leaves as four 16-byte stores per lane (4 consecutive points of one channel).
  float4 ay[4], bx4[4];
  float4 ad[POOLED ? 1 : 4];
  float apv[POOLED ? 4 : 1];
  int aarg[POOLED ? 4 : 1];
  auto load_chunk = [&](long long qk) {
    const long long qc = min(qk, total - kKTW);
    int b, pk;
    split_point(qc, P, b, pk);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + 64 * i;
      const int row = e >> 3, p = pk + (e & 7) * 4;
      const int ch = min(row, cout - 1);
      const size_t rowo = (size_t)b * cout + ch;
      ay[i] = *reinterpret_cast<const float4*>(y + rowo * (size_t)P + p);
      if (POOLED) {
        const int G = P / gs.S, g = p / gs.S;
        apv[i] = pooled_at(gs, b, ch, G, g);
        aarg[i] = gs.arg[rowo * (size_t)G + g];
      } else {
        ad[i] = *reinterpret_cast<const float4*>(gs.dense + rowo * (size_t)P + p);
      }
      bx4[i] = *reinterpret_cast<const float4*>(x + ((size_t)b * cin + min(row, cin - 1)) * P + p);
    }
  };
  auto store_chunk = [&](long long qk) {
    int b_unused, pk;
    split_point(min(qk, total - kKTW), P, b_unused, pk);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + 64 * i;
      const int row = e >> 3, k = (e & 7) * 4;
      const bool okq = qk + k < qend;
      const float rs = s_k[0][row], rh = s_k[1][row];
      const float rca = s_k[2][row], rcb = s_k[3][row], rcc = s_k[4][row];
      float4 d;
      if (POOLED) {
        const int ks = (pk + k) % gs.S, a = aarg[i];
        const float pv = apv[i];
        d = make_float4(a == ks ? pv : 0.f, a == ks + 1 ? pv : 0.f, a == ks + 2 ? pv : 0.f, a == ks + 3 ? pv : 0.f);
      } else {
        d = ad[i];
      }
      const float4 yv = ay[i];
      float4 v;
      v.x = rca * ((yv.x * rs + rh > 0.f) ? d.x : 0.f) + rcb + rcc * yv.x;
      v.y = rca * ((yv.y * rs + rh > 0.f) ? d.y : 0.f) + rcb + rcc * yv.y;
      v.z = rca * ((yv.z * rs + rh > 0.f) ? d.z : 0.f) + rcb + rcc * yv.z;
      v.w = rca * ((yv.w * rs + rh > 0.f) ? d.w : 0.f) + rcb + rcc * yv.w;
      if (!(okq && row < cout)) v = zero4();
      As[(k + 0) * LD + row] = v.x; As[(k + 1) * LD + row] = v.y;
      As[(k + 2) * LD + row] = v.z; As[(k + 3) * LD + row] = v.w;
      const float4 u = bx4[i];   // raw: the statistics need y_{l-1} itself, the activation is applied on read
      Bs[(k + 0) * LD + row] = u.x; Bs[(k + 1) * LD + row] = u.y;
      Bs[(k + 2) * LD + row] = u.z; Bs[(k + 3) * LD + row] = u.w;
    }
  };

  f32x16 accw;
#pragma unroll
  for (int r = 0; r < 16; ++r) accw[r] = 0.f;
  float sg = 0.f, sgy = 0.f;   // statistics of input channel ci = lane & 31 over the point rows this lane holds
  const int nchunks = (int)((qend - qbeg + kKTW - 1) / kKTW);
  int t = wv;
  if (t < nchunks) load_chunk(qbeg + (long long)t * kKTW);
  for (; t < nchunks; t += 4) {
    const long long qk = qbeg + (long long)t * kKTW;
    store_chunk(qk);                                                // wave-private LDS: no workgroup barrier
    if (weighted && lane < kKTW) s_w[wv][lane] = colw[min(qk, total - kKTW) + lane];
    __builtin_amdgcn_sched_barrier(0);                              // the next loads reuse the registers just consumed
    if (t + 4 < nchunks) load_chunk(qbeg + (long long)(t + 4) * kKTW);  // in flight during the MFMAs
    __builtin_amdgcn_s_waitcnt(0xc07f);                             // lgkmcnt(0): this wave's ds_writes landed
    f32x16 accd;
#pragma unroll
    for (int r = 0; r < 16; ++r) accd[r] = 0.f;
    const float* ap = As + (lane >> 5) * LD + (lane & 31);   // dY[pt = 2kk + half][co = lane & 31]
    const float* bp = Bs + (lane >> 5) * LD + (lane & 31);   // x [pt = 2kk + half][ci = lane & 31]
    const float* tp = As + (lane & 31) * LD + (lane >> 5);   // dY[pt = lane & 31][co = 2kk + half]
    // fragments of step kk+1 are read while the MFMAs of step kk run; the scheduling barriers keep the compiler from
    // hoisting all 48 LDS reads of the unrolled loop (registers)
    float fa[2], fb[2], ft[2];
    fa[0] = ap[0]; fb[0] = bp[0]; ft[0] = tp[0];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < 16) { fa[nxt] = ap[2 * (kk + 1) * LD]; fb[nxt] = bp[2 * (kk + 1) * LD]; ft[nxt] = tp[2 * (kk + 1)]; }
      __builtin_amdgcn_sched_barrier(0);
      float bx = fmaxf(fb[cur] * bsc + bsh, 0.f);
      if (weighted) bx *= s_w[wv][2 * kk + (lane >> 5)];                                // multiplicity of point 2kk + half
      accw = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur], bx, accw, 0, 0, 0);         // dW[co][ci] += dY . act(x)^T
      accd = __builtin_amdgcn_mfma_f32_32x32x2f32(ft[cur], wfrag[kk], accd, 0, 0, 0);  // dA^T[pt][ci] = dY^T . W
      __builtin_amdgcn_sched_barrier(0);
    }
    // dA^T of this chunk: register r of a lane is (pt = mfma_row(r, lane), ci = lane & 31); registers 4j .. 4j+3 are
    // four consecutive points
    const long long qc = min(qk, total - kKTW);
    int b, pk;
    split_point(qc, P, b, pk);
    const int ci = lane & 31;
    float* dxb = dx + ((size_t)b * cin + ci) * P + pk + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pt0 = 8 * j + 4 * (lane >> 5);
      if (ci < cin && qk + pt0 < qend) {            // splits end on multiples of 4 points (P % 32 == 0)
        float4 o;
        o.x = accd[4 * j + 0]; o.y = accd[4 * j + 1]; o.z = accd[4 * j + 2]; o.w = accd[4 * j + 3];
        *reinterpret_cast<float4*>(dxb + 8 * j) = o;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float yin = Bs[(pt0 + u) * LD + ci];
          float gq = (yin * bsc + bsh > 0.f) ? accd[4 * j + u] : 0.f;
          if (weighted) gq *= s_w[wv][pt0 + u];
          sg += gq;
          sgy += gq * yin;
        }
      }
    }
  }
  // ---- per-workgroup results: dW partial (sum of the four waves) and the statistics partials ----
  __syncthreads();
  float* red = &lds[0][0][0][0];   // [4][32*32] floats fit in the staging area
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wv * 1024 + mfma_row(r, lane) * 32 + (lane & 31)] = accw[r];
  __syncthreads();
  float* out = dw_part + (size_t)blockIdx.x * cout * cin;
  for (int i = threadIdx.x; i < 1024; i += kThreads) {
    const int row = i >> 5, col = i & 31;
    if (row < cout && col < cin)
      out[(size_t)row * cin + col] = (red[i] + red[1024 + i]) + (red[2048 + i] + red[3072 + i]);
  }
  __syncthreads();
  float* sred = red;               // [4 waves][2 halves][32 channels][2]
  sred[((wv * 2 + (lane >> 5)) * 32 + (lane & 31)) * 2 + 0] = sg;
  sred[((wv * 2 + (lane >> 5)) * 32 + (lane & 31)) * 2 + 1] = sgy;
  __syncthreads();
  if (threadIdx.x < cin) {
    const int ci = threadIdx.x;
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a += sred[(k * 32 + ci) * 2 + 0]; c += sred[(k * 32 + ci) * 2 + 1]; }
    part_g[(size_t)ci * nt_total + blockIdx.x] = a;
    part_gy[(size_t)ci * nt_total + blockIdx.x] = c;
  }
}

// ============================================================================================
// Fused backward of a mid-size layer (cout in {64, 128}, cin in {32, 64, 128}, not the first of its stack): the
// pw_bwd_small_kernel contract -- dA_{l-1}, its BN-backward statistics partials and a split-K partial of dW from ONE
// pass over (y_l, gradient source, y_{l-1}) -- for the layers whose whole weight matrix still fits one workgroup.
// The separate dgrad / wgrad pair reads y_l and the gradient source twice, y_{l-1} twice, and forms dY (relu mask,
// three BN-backward constants) in both operand loaders; here a chunk of PT points is loaded and finished ONCE into
// two LDS tiles kept in the tensors' own layout ([channel][point], float4 stores straight from the load registers):
//   dgrad  dA^T[pt][ci] = sum_co dY^T[pt][co] W[co][ci]  A = dY^T: ds_read_b32 over 32 consecutive points of a row,
//                                                        B = W fragments held in registers for the whole kernel;
//   wgrad  dW[co][ci]  += sum_pt dY[co][pt] act(x)[ci][pt]  K = points: both operands are rows of the tiles, read as
//          float4 along the points -- a lane's four values feed four consecutive MFMA k-steps (the k index is a dummy
//          index, so A and B only have to agree on which point a (lane half, step) pair means: point 8j + 4*half + t).
// Eight waves with two roles (one workgroup per CU, one wave of each role per SIMD): waves 4..7 are LOADERS -- global
// loads two chunks ahead in two register sets, dY finished and both tiles written into the other half of a
// double-buffered LDS -- and waves 0..3 only issue MFMAs (one 32x32 dA^T tile each, PT = 128 / (cin/32) points per
// chunk; the dW tiles as a (WGM x WGN) grid of TM x TN tiles, over point halves (WGK = 2) when the matrix has only two
// tiles) and store dA with its statistics.  One barrier per chunk.  Measured before the split (four waves doing both,
// two workgroups per CU): MFMA phase, loads and the dY arithmetic ran back to back (24 + 10 + 10 us on the 128 -> 128
// layer of SA4), because every workgroup of the launch is in the same phase at the same time.
// ============================================================================================
template <int COT, int CIT>
struct MidCfg {
  static constexpr int COUT = 32 * COT, CIN = 32 * CIT;
  static constexpr int PT = 128 / CIT;                  // points per chunk: CIT * PT / 32 == 4 dA^T tiles, one per compute wave
  static constexpr int LD = PT + 4;                     // 16-byte aligned rows; +4 floats: float4 reads of 8 rows cover 32 banks
  static constexpr int F4 = PT / 4;                     // float4 per tile row
  static constexpr int NY = COUT * F4 / 256;            // float4 per loader thread, y / gradient tile
  static constexpr int NX = CIN * F4 / 256;
  static constexpr int WGN = CIT < 2 ? CIT : 2;
  static constexpr int WGM = COT < 4 / WGN ? COT : 4 / WGN;
  static constexpr int WGK = 4 / (WGM * WGN);
  static constexpr int TM = COT / WGM, TN = CIT / WGN;
  static constexpr int TILE = (COUT + CIN) * LD;        // floats of one LDS buffer: dY tile, then the raw x tile
  static constexpr size_t LDS_BYTES = (2 * TILE + 5 * COUT) * sizeof(float);
  static_assert(NY >= 1 && NX >= 1 && WGM * WGN * WGK == 4 && TM * WGM == COT && TN * WGN == CIT, "unsupported shape");
};

template <int COT, int CIT, bool POOLED, bool WGRAD = true>   // WGRAD false: dA and its statistics only (cout = 256)
__global__ __launch_bounds__(kMidThreads) void pw_bwd_mid_kernel(
    int P, long long total, int split_len, const float* __restrict__ w, const float* __restrict__ x,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, const float* __restrict__ y, GradSrc gs,
    const float* __restrict__ bn, const float* __restrict__ bwdc, float* __restrict__ dx,
    float* __restrict__ part_g, float* __restrict__ part_gy, int nt_total, float* __restrict__ dw_part) {
  using C = MidCfg<COT, CIT>;
  constexpr int COUT = C::COUT, CIN = C::CIN, PT = C::PT, LD = C::LD, F4 = C::F4, NY = C::NY, NX = C::NX;
  constexpr int TM = C::TM, TN = C::TN;
  extern __shared__ __attribute__((aligned(16))) float mid_lds[];
  float* const s_c = mid_lds + 2 * C::TILE;     // [5][COUT]: scale, shift of this layer's BN; the three BN-backward constants
  const int lane = lane_id();
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the role branches are wave-uniform
  const bool loader = wv >= 4;
  const int cw = wv & 3;                        // index within the role
  const int tid = threadIdx.x & 255;            // thread index within the role
  const int l31 = lane & 31, half = lane >> 5;
  const long long qbeg = (long long)blockIdx.x * split_len;
  const long long qend = max(qbeg, min(qbeg + (long long)split_len, total));
  const int nchunks = (int)((qend - qbeg) / PT);   // P % PT == 0 (host-checked): whole chunks, each inside one cloud
  for (int c = threadIdx.x; c < COUT; c += kMidThreads) {
    s_c[0 * COUT + c] = bn[c]; s_c[1 * COUT + c] = bn[COUT + c];
    s_c[2 * COUT + c] = bwdc[c]; s_c[3 * COUT + c] = bwdc[COUT + c]; s_c[4 * COUT + c] = bwdc[2 * COUT + c];
  }

  __syncthreads();                                 // s_c
  // ---------------- loader role ----------------
  // ONE register set, re-issued SLOT BY SLOT: as soon as a float4 (pair) of chunk t has been finished into the LDS tile,
  // the load of the same slot of chunk t + 1 goes out into the same registers.  Every load is then in flight for a whole
  // chunk period (the rest of this iteration, the barrier, the start of the next) -- several times the HBM latency --
  // and consumption always wants the OLDEST outstanding loads, so the compiler's waits are exact `s_waitcnt vmcnt(11)` /
  // `(10)` and nothing is ever drained.  Rounds 2-3 kept two whole-chunk register sets (a chunk's loads issued after
  // the previous chunk's stores as one block); in the 252-register kernel the allocator re-used pieces of an in-flight
  // set for temporaries and copied them away right after the load, which put a `vmcnt(7)` behind every second chunk's
  // issue and a full drain in front of every second chunk's stores (ISA of round 3): the MFMA waves stood 17-34 % of
  // their time at the chunk barrier (tools/bwd_mid_phases.py, profiles/r04_bwd_mid_phases.txt: now 12-31 %; what is left
  // is the loaders' own throughput -- 6.5 B / cycle / CU on the cout-256 layer, near what a CU streams -- and a second
  // set in flight, tried with the same exact waits (vmcnt(23)), changes nothing).  Named x registers, not `float4 x[NX]`:
  // as an array the four rows went to scratch memory, and a kernel with a scratch segment does not share the chip with
  // the kernels of other streams (tools/exp/corun.py).
  if (loader) {
    static_assert(NX == 4, "CIN * PT == 4096: four float4 of the x tile per loader thread");
    struct Set {
      float4 y[NY];
      float4 d[POOLED ? 1 : NY];
      float pv[POOLED ? NY : 1];
      int arg[POOLED ? NY : 1];
      float4 x0, x1, x2, x3;
    };
    Set s0;
    auto issue_y = [&](Set& st, int i, int b, int pk) {
      const int e = tid + 256 * i, row = e / F4, p = pk + (e % F4) * 4;
      const size_t rowo = (size_t)b * COUT + row;
      st.y[i] = *reinterpret_cast<const float4*>(y + rowo * (size_t)P + p);
      if (POOLED) {
        const int G = P / gs.S, g = p / gs.S;
        st.pv[i] = pooled_at(gs, b, row, G, g);
        st.arg[i] = gs.arg[rowo * (size_t)G + g];
      } else {
        st.d[i] = *reinterpret_cast<const float4*>(gs.dense + rowo * (size_t)P + p);
      }
    };
    auto load_x = [&](int i, int b, int pk) {
      const int e = tid + 256 * i, row = e / F4, p = pk + (e % F4) * 4;
      return *reinterpret_cast<const float4*>(x + ((size_t)b * CIN + row) * P + p);
    };
    auto put_x = [&](int i, float* Xs, const float4& v) {      // RAW y_{l-1}: the statistics need it, act() is applied on read
      const int e = tid + 256 * i, row = e / F4, k = (e % F4) * 4;
      *reinterpret_cast<float4*>(&Xs[row * LD + k]) = v;
    };
    auto finish_y = [&](const Set& st, int i, float* dYs, int pk) {
      const int e = tid + 256 * i, row = e / F4, k = (e % F4) * 4;
      const float rs = s_c[row], rh = s_c[COUT + row], rca = s_c[2 * COUT + row], rcb = s_c[3 * COUT + row],
                  rcc = s_c[4 * COUT + row];
      float4 d;
      if (POOLED) {
        const int ks = (pk + k) % gs.S, a = st.arg[i];
        const float pv = st.pv[i];
        d = make_float4(a == ks ? pv : 0.f, a == ks + 1 ? pv : 0.f, a == ks + 2 ? pv : 0.f, a == ks + 3 ? pv : 0.f);
      } else {
        d = st.d[i];
      }
      const float4 yv = st.y[i];
      float4 v;
      v.x = rca * ((yv.x * rs + rh > 0.f) ? d.x : 0.f) + rcb + rcc * yv.x;
      v.y = rca * ((yv.y * rs + rh > 0.f) ? d.y : 0.f) + rcb + rcc * yv.y;
      v.z = rca * ((yv.z * rs + rh > 0.f) ? d.z : 0.f) + rcb + rcc * yv.z;
      v.w = rca * ((yv.w * rs + rh > 0.f) ? d.w : 0.f) + rcb + rcc * yv.w;
      *reinterpret_cast<float4*>(&dYs[row * LD + k]) = v;
    };
    auto issue_all = [&](Set& st, long long qk) {
      int b, pk;
      split_point(qk, P, b, pk);
#pragma unroll
      for (int i = 0; i < NY; ++i) issue_y(st, i, b, pk);
      st.x0 = load_x(0, b, pk); st.x1 = load_x(1, b, pk); st.x2 = load_x(2, b, pk); st.x3 = load_x(3, b, pk);
    };
    // chunk t out of the set into LDS buffer `buf`; every slot is re-issued for chunk t + 1 (clamped to the last chunk:
    // the tail re-reads it -- one chunk of redundant L2 hits per workgroup -- because a branch around the issue makes
    // the compiler's wait counts merge to vmcnt(0)).  Scheduling fences keep "finish slot i, re-issue slot i" in this
    // order: left alone, the scheduler hoists the re-issues and double-buffers the old values.
    auto step_set = [&](Set& st, float* buf, int t) {
      float* Xs = buf + COUT * LD;
      int b_cur, pk_cur, b_nxt, pk_nxt;
      split_point(qbeg + (long long)t * PT, P, b_cur, pk_cur);
      split_point(qbeg + (long long)min(t + 1, nchunks - 1) * PT, P, b_nxt, pk_nxt);
#pragma unroll
      for (int i = 0; i < NY; ++i) {
        finish_y(st, i, buf, pk_cur);
        __builtin_amdgcn_sched_barrier(0);
        issue_y(st, i, b_nxt, pk_nxt);
        __builtin_amdgcn_sched_barrier(0);
      }
      put_x(0, Xs, st.x0); __builtin_amdgcn_sched_barrier(0); st.x0 = load_x(0, b_nxt, pk_nxt); __builtin_amdgcn_sched_barrier(0);
      put_x(1, Xs, st.x1); __builtin_amdgcn_sched_barrier(0); st.x1 = load_x(1, b_nxt, pk_nxt); __builtin_amdgcn_sched_barrier(0);
      put_x(2, Xs, st.x2); __builtin_amdgcn_sched_barrier(0); st.x2 = load_x(2, b_nxt, pk_nxt); __builtin_amdgcn_sched_barrier(0);
      put_x(3, Xs, st.x3); __builtin_amdgcn_sched_barrier(0); st.x3 = load_x(3, b_nxt, pk_nxt); __builtin_amdgcn_sched_barrier(0);
    };
    // The loader's VALU work shares the SIMD's issue port with the MFMA wave and loses to it (measured: the store phase
    // takes 1.7x longer beside the MFMAs).  Where the loaders are the longer side (cin <= 64) they get priority; on
    // the 128 -> 128 layers the MFMA wave is the critical path and priority costs 8 %.
    if (CIT <= 2) __builtin_amdgcn_s_setprio(3);
    [[maybe_unused]] constexpr int kMidKind = COT == 8 ? 0 : (COT == 4 && CIT == 4 ? 1 : (COT == 4 && CIT == 2 ? 2 : (COT == 2 && CIT == 2 ? 3 : (COT == 2 && CIT == 1 ? 4 : 5))));
    MID_T0()
    if (nchunks > 0) issue_all(s0, qbeg);
    MID_T(0)                                                       // the first chunk's loads issued
    // Chunk t: the loaders fill buffer t & 1 BEFORE barrier t, the compute waves read it AFTER barrier t.  A loader is
    // at most one chunk ahead: it reaches barrier t + 1 (buffer (t + 1) & 1 written) only after the compute waves
    // passed barrier t, i.e. finished chunk t - 1, the last reader of that buffer.
    for (int t = 0; t < nchunks; ++t) {
      step_set(s0, mid_lds + (t & 1) * C::TILE, t);
      MID_T(1)                                                     // wait for loads + dY arithmetic + LDS writes + re-issue
      lds_barrier();
      MID_T(3)                                                     // barrier: waiting for the compute waves
    }
    MID_END(kMidKind, 1)
    __syncthreads();                               // the four barriers of the compute waves' epilogue
    if (WGRAD && C::WGK == 2) { __syncthreads(); __syncthreads(); }
    __syncthreads();
    return;
  }
  // ---------------- compute waves ----------------
  // dgrad: wave -> (point block pb, input-channel block cb); B[k = co][j = ci] = w[co][32 cb + j]
  const int pb = cw / CIT, cb = cw % CIT;
  // wgrad: wave -> (row group wm, column group wn, point half wk)
  const int wk = cw / (C::WGM * C::WGN), wm = (cw / C::WGN) % C::WGM, wn = cw % C::WGN;
  float wfrag[COUT / 2];
  float wsc[TN], wsh[TN];
  f32x16 accw[TM][TN];
  float sg = 0.f, sgy = 0.f;   // statistics of input channel 32 cb + l31 over the point rows this lane holds
#pragma unroll
  for (int kk = 0; kk < COUT / 2; ++kk) wfrag[kk] = w[(size_t)(2 * kk + half) * CIN + 32 * cb + l31];
  const float dsc = in_scale[32 * cb + l31], dsh = in_shift[32 * cb + l31];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    wsc[tn] = in_scale[32 * (wn * TN + tn) + l31];
    wsh[tn] = in_shift[32 * (wn * TN + tn) + l31];
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) accw[tm][tn][r] = 0.f;

  auto compute_chunk = [&](const float* buf, long long qk) {
    const float* dYs = buf;
    const float* Xs = buf + COUT * LD;
    // ---- dgrad: dA^T tile (points 32 pb .., input channels 32 cb ..) ----
    {
      f32x16 accd;
#pragma unroll
      for (int r = 0; r < 16; ++r) accd[r] = 0.f;
      const float* ap = dYs + half * LD + 32 * pb + l31;   // dY[co = 2kk + half][pt = 32 pb + l31]
      // fragments of the NEXT group of kDG k-steps are read while the MFMAs of this group run
      constexpr int kDG = 8;
      float fa[2][kDG];
#pragma unroll
      for (int u = 0; u < kDG; ++u) fa[0][u] = ap[2 * u * LD];
#pragma unroll
      for (int gk = 0; gk < COUT / 2 / kDG; ++gk) {
        const int cur = gk & 1, nxt = cur ^ 1;
        if (gk + 1 < COUT / 2 / kDG) {
#pragma unroll
          for (int u = 0; u < kDG; ++u) fa[nxt][u] = ap[2 * ((gk + 1) * kDG + u) * LD];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kDG; ++u)
          accd = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][u], wfrag[gk * kDG + u], accd, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // register r of a lane: point mfma_row(r, lane) of the block, input channel l31; 4j .. 4j+3 = 4 consecutive points
      int b, pk;
      split_point(qk, P, b, pk);
      const int ci = 32 * cb + l31;
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
    // ---- wgrad: this wave's TM x TN tiles over its share of the chunk's points ----
    if (WGRAD) {
      constexpr int KP = PT / C::WGK;          // points of this wave's K range
      const float* ap = dYs + (32 * wm * TM + l31) * LD + wk * KP + 4 * half;
      const float* bp = Xs + (32 * wn * TN + l31) * LD + wk * KP + 4 * half;
      float4 a4[2][TM], b4[2][TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a4[0][tm] = *reinterpret_cast<const float4*>(ap + tm * 32 * LD);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b4[0][tn] = *reinterpret_cast<const float4*>(bp + tn * 32 * LD);
#pragma unroll
      for (int j = 0; j < KP / 8; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        if (j + 1 < KP / 8) {      // next group's rows in flight during this group's MFMAs
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) a4[nxt][tm] = *reinterpret_cast<const float4*>(ap + tm * 32 * LD + 8 * (j + 1));
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) b4[nxt][tn] = *reinterpret_cast<const float4*>(bp + tn * 32 * LD + 8 * (j + 1));
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b4[cur][tn] = bn_relu4(b4[cur][tn], wsc[tn], wsh[tn]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
              const float av = t == 0 ? a4[cur][tm].x : (t == 1 ? a4[cur][tm].y : (t == 2 ? a4[cur][tm].z : a4[cur][tm].w));
              const float bv = t == 0 ? b4[cur][tn].x : (t == 1 ? b4[cur][tn].y : (t == 2 ? b4[cur][tn].z : b4[cur][tn].w));
              accw[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accw[tm][tn], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  [[maybe_unused]] constexpr int kMidKindC = COT == 8 ? 0 : (COT == 4 && CIT == 4 ? 1 : (COT == 4 && CIT == 2 ? 2 : (COT == 2 && CIT == 2 ? 3 : (COT == 2 && CIT == 1 ? 4 : 5))));
  MID_T0()
  for (int t = 0; t < nchunks; ++t) {
    lds_barrier();
    MID_T(0)                                                       // barrier: waiting for the loaders
    compute_chunk(mid_lds + (t & 1) * C::TILE, qbeg + (long long)t * PT);
#ifdef ISTNET_PHASE_TIMING
    asm volatile("s_nop 0" ::"v"(accw[0][0][0]), "v"(sg));
#endif
    MID_T(1)                                                       // the chunk's MFMAs, epilogue stores and statistics
  }
  if (cw == 0) { MID_END(kMidKindC, 0) }
  // ---- per-workgroup results ----
  __syncthreads();      // the compute waves are done with the tiles
  float* red = mid_lds;
  if (WGRAD && C::WGK == 2) {    // two waves hold halves of the same dW tile
    if (wk == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(cw & 1) * 1024 + r * 64 + lane] = accw[0][0][r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) accw[0][0][r] += red[(cw & 1) * 1024 + r * 64 + lane];
    }
    __syncthreads();
  }
  if (WGRAD && wk == 0) {
    float* out = dw_part + (size_t)blockIdx.x * COUT * CIN;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * (wm * TM + tm) + mfma_row(r, lane), col = 32 * (wn * TN + tn) + l31;
          out[(size_t)row * CIN + col] = accw[tm][tn][r];
        }
  }
  float* sred = red;    // [4 waves][2 halves][32][2]
  sred[((cw * 2 + half) * 32 + l31) * 2 + 0] = sg;
  sred[((cw * 2 + half) * 32 + l31) * 2 + 1] = sgy;
  __syncthreads();
  if (threadIdx.x < CIN) {
    const int cbk = threadIdx.x >> 5, l = threadIdx.x & 31;
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 4 / CIT; ++k) {        // the waves with this channel block, in wave order; both halves
      const int wsrc = k * CIT + cbk;
      a += sred[((wsrc * 2 + 0) * 32 + l) * 2 + 0] + sred[((wsrc * 2 + 1) * 32 + l) * 2 + 0];
      c += sred[((wsrc * 2 + 0) * 32 + l) * 2 + 1] + sred[((wsrc * 2 + 1) * 32 + l) * 2 + 1];
    }
    part_g[(size_t)threadIdx.x * nt_total + blockIdx.x] = a;
    part_gy[(size_t)threadIdx.x * nt_total + blockIdx.x] = c;
  }
}

// dw[i] = sum_s dw_part[s][i]: 16 elements x 16 split groups per workgroup, fixed reduction order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(int count, int splits, const float* __restrict__ part,
                                                           float* __restrict__ dw) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < count) {
    int k = sg;
    for (; k + 48 < splits; k += 64) {
      s0 += part[(size_t)k * count + i];
      s1 += part[(size_t)(k + 16) * count + i];
      s2 += part[(size_t)(k + 32) * count + i];
      s3 += part[(size_t)(k + 48) * count + i];
    }
    for (; k < splits; k += 16) s0 += part[(size_t)k * count + i];
  }
  red[sg][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.x < 16 && i < count) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[g][el];
    dw[i] = s;
  }
}

// Same reduction for up to 8 layers in one launch (descriptors by value): one launch per SharedMLP instead of
// one per layer -- the reduce is pure launch latency (~9 us for ~1 us of work).
// An item may be a COLUMN BLOCK of a wider destination matrix (cols / ld: element i goes to row i / cols, column
// i % cols of a matrix with leading dimension ld) and its partials may be a row block of wider partial matrices
// (pstride: distance between consecutive splits): the pieces of a weight gradient that come from different launches
// (xyz columns + feature columns of a set-abstraction layer 0; interpolated + skip columns of a feature-propagation
// layer 0) land in the parameter's gradient directly, with no concatenation afterwards.
struct ReduceBatch {
  const float* part[8];
  float* dw[8];
  int count[8];
  int splits[8];
  int cols[8];
  int ld[8];
  long long pstride[8];
  int groups[8];       // split groups of the item's workgroups (a power of two <= kRedGroups); 1024 / groups elements each
  int block_begin[9];  // prefix sum of ceil(count / (1024 / groups))
  int n;
};
// Workgroup = 1024 threads = (1024 / G consecutive elements) x (G split groups): a wave reads 256 contiguous bytes per
// split, four independent chains per thread keep loads in flight, the G group sums meet in LDS in a fixed order.  G follows
// the item's split count (reduce_groups(): about four or more splits per thread, at most 16 groups): round 3 used four
// groups for everything (32 - 128 dependent-latency steps per thread at 128 - 512 splits, 10 us per launch); the first form of
// round 4 used sixteen for everything, which at 8 - 32 splits -- the 256- and 512-channel layers -- left most of a
// workgroup's threads without a load and launched 4 096 workgroups where 512 do.  Deterministic; the summation order is a
// function of the split count only.  (Also tried: 64 groups of 16 lanes x float4, +1 % on the step in an in-box A/B.)
constexpr int kRedGroups = 16;
static int reduce_groups(int splits) {
  int g = 1;
  while (g < kRedGroups && g * 4 < splits) g <<= 1;
  return g;
}
__global__ __launch_bounds__(1024) void wgrad_reduce_multi_kernel(ReduceBatch rb) {
  __shared__ float red[1024];
  int l = 0;
  while (l + 1 < rb.n && (int)blockIdx.x >= rb.block_begin[l + 1]) ++l;
  const int count = rb.count[l], splits = rb.splits[l], groups = rb.groups[l];
  const int epw = 1024 / groups;                       // elements per workgroup
  const size_t ps = (size_t)rb.pstride[l];
  const float* __restrict__ part = rb.part[l];
  const int el = threadIdx.x & (epw - 1), sg = threadIdx.x / epw;
  const int i = ((int)blockIdx.x - rb.block_begin[l]) * epw + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < count) {
    int k = sg;
    for (; k + 3 * groups < splits; k += 4 * groups) {
      s0 += part[(size_t)k * ps + i];
      s1 += part[(size_t)(k + groups) * ps + i];
      s2 += part[(size_t)(k + 2 * groups) * ps + i];
      s3 += part[(size_t)(k + 3 * groups) * ps + i];
    }
    for (; k < splits; k += groups) s0 += part[(size_t)k * ps + i];
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sg == 0 && i < count) {
    float s = red[el];
    for (int g = 1; g < groups; ++g) s += red[g * epw + el];
    const int cols = rb.cols[l];
    rb.dw[l][(size_t)(i / cols) * rb.ld[l] + i % cols] = s;
  }
}

// Row blocks of up to 8 small matrices copied into one buffer in ONE launch (the layer-0 weights of a level's scales
// stacked for the level-wide feature-gradient product; the tensors of a geometry slot packed into its flat storage).
struct PackBatch {
  const unsigned* src[64];
  long long begin[65];   // prefix sum of the sources' sizes in 4-byte words
  int n;
};
__global__ __launch_bounds__(256) void pack_words_kernel(PackBatch pb, unsigned* __restrict__ dst) {
  const long long total = pb.begin[pb.n];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    int lo = 0, hi = pb.n - 1;          // source holding word i: last l with begin[l] <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (pb.begin[mid] <= i) lo = mid; else hi = mid - 1;
    }
    dst[i] = pb.src[lo][i - pb.begin[lo]];
  }
}

// ---------------------------------------------------------------------------------------------
inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int grid_rows(long long rows) { return (int)(rows < 65535 ? rows : 65535); }   // grid.y limit; the kernels loop over rows

// tile selection shared by the forward launch and istnet_pw_stat_tiles()
enum TileCfg { kCfg128x128, kCfg64x128, kCfg64x64, kCfg32x256, kCfg128x64, kCfg256x64 };   // 128x64 / 256x64: dgrad only
int g_force_fwd_cfg = -1, g_force_dgrad_cfg = -1;  // experiments (istnet_pw_set_tuning keys 3, 4): TileCfg or -1
inline TileCfg pick_cfg(int b, int m, int P, int force = -1) {
  if (force >= 0) return (TileCfg)force;
  if (m <= 32) return kCfg32x256;
  const long long n128 = (long long)b * ceil_div(P, 128);
  if (m > 64 && n128 * ceil_div(m, 128) >= 512) return kCfg128x128;
  if (n128 * ceil_div(m, 64) >= 512) return kCfg64x128;
  return kCfg64x64;
}
// dgrad: the B operand is dY, formed from (y, gradient source, 5 per-channel constants) by the loader -- the expensive
// operand.  Every row tile of a point tile re-reads and re-forms the same dY, so the tile is made as TALL as the
// launch allows (the whole output height when that still gives enough workgroups): 64-point tiles, 64 / 128 / 256
// rows.  -2 = the forward rule, 0..5 = a fixed TileCfg (experiments).
int g_dgrad_min_wgs = 384;   // istnet_pw_set_tuning key 7
inline TileCfg pick_dgrad_cfg(int b, int m, int P, int force) {
  if (force >= 0) return (TileCfg)force;
  if (force == -2) return pick_cfg(b, m, P);
  if (m <= 32) return kCfg32x256;
  if (m <= 64) return kCfg64x64;
  const long long n64 = (long long)b * ceil_div(P, 64);
  if (m > 128 && n64 * ceil_div(m, 256) >= g_dgrad_min_wgs) return kCfg256x64;
  if (n64 * ceil_div(m, 128) >= g_dgrad_min_wgs) return kCfg128x64;
  return kCfg64x64;
}
inline int cfg_nt(TileCfg c) { return c == kCfg32x256 ? 256 : ((c == kCfg64x64 || c == kCfg128x64 || c == kCfg256x64) ? 64 : 128); }
inline int cfg_mt(TileCfg c) {
  return c == kCfg256x64 ? 256 : ((c == kCfg128x128 || c == kCfg128x64) ? 128 : (c == kCfg32x256 ? 32 : 64));
}

inline bool wgrad_small(int cin, int cout) { return cin <= 32 && cout <= 32; }
// tuning knobs (istnet_pw_set_tuning): experiments only, defaults are the measured best
int g_wg_small_pts = 0x7fffffff;  // layers with b*P <= this use 64x64 wgrad tiles: measured best for every encoder layer (smaller split-K partials)
int g_wg_target_big = 512;   // target workgroup count, outputs >= 128x128 (re-tuned end to end once the wgrads ran beside the dgrad chain: 768/1024 -> 512/512 is 1.5 % faster)
int g_wg_target_small = 512;
int g_bwd_small_target = 512; // workgroups of the fused small-layer backward (key 5; inside the step 512 beats 256 by 0.4 %, alone 256 wins)
int g_bwd_mid_target = 128;   // workgroups of the fused mid-size-layer backward (key 8).  Alone, one per CU (256) is 10 % faster;
                              // inside the step 128 wins by 0.9 % (2.848 -> 2.823 ms): the other half of the chip stays open to
                              // the other scale's chain and the weight-gradient stream
int g_bwd_mid_enable = 1;     // key 9: 0 = those layers run the dgrad / wgrad pair
int g_exp_no_fast = 0;        // experiment (key 6): 1 = never take the interior-tile fast kernels
inline int wgrad_mt(int cout, long long pts) { return (cout >= 128 && pts > g_wg_small_pts) ? 128 : 64; }
inline int wgrad_nt(int cin, long long pts) { return (cin >= 96 && pts > g_wg_small_pts) ? 128 : 64; }
int g_fwd_sk_enable = 1;       // key 15: 0 = no pw_fwd_sk_kernel
int g_dgrad_rs_enable = 1;     // key 19: 0 = no role-split dgrad for cout = 256 / cin = 128
int g_dgrad_sk_enable = 1;     // key 17: 0 = no pw_dgrad_sk_kernel
int g_dgrad_sk_min_k = 256;    // key 18
int g_fwd_sk_max_tiles = 1024; // key 16: launches with more 32 x 128 tiles than this keep the LDS-tiled kernel (measured:
                               // +15-35 % at <= 1024 tiles -- the FP levels --, -8 % at 2048)
int g_sk_tm2_min_wgs = 256;     // key 24: the split-K FORWARD kernel takes 64 x 128 tiles (TM = 2: the activation operand is loaded once
                               // for two row blocks of weights -- 6 instead of 10 operand loads per 32 MFMAs) when the launch still
                               // has this many workgroups; 0 = always 32 x 128
int g_fwd2_enable = 1;         // key 13: 0 = pw_fwd_kernel for every forward launch
int g_fwd2_min_waves = 1024;   // key 14 (step time at 2048 / 1024 / 768 / 512 / 256: 2.912 / 2.870 / 2.879 / 2.933 / 2.997 ms)
int g_wgrad2_enable = 1;       // key 11: 0 = pw_wgrad_kernel for every dense layer
int g_wgrad2_target = 512;     // key 12: workgroups of a pw_wgrad2_kernel launch (two per CU with the 64 x 64 tiles, see key 20)
inline bool wgrad2_ok(int cin, int cout) {
  return g_wgrad2_enable && cin >= 64 && cout >= 64 && cin % 32 == 0 && cout % 32 == 0;
}
int g_interp_dy_lds = 1;           // key 22: 0 = interp_grad_csr_dy gathers dY from global memory (no LDS staging)
int g_scatter_csr_threads = 0;    // key 21: threads per workgroup of pw_scatter_csr_kernel (256 / 512 / 1024); 0 = by the cloud's source
                                  // points: 1024 from 512 points, 512 from 256, else 256.  Alone on layer 0 of SA2 / SA3 / SA4 (two
                                  // scales each, profiles/r04_scatter_microbench.txt): 256 threads 36.4 + 51.5 / 22.6 + 28.6 / 15.0 +
                                  // 22.6 us, 512: 26.1 + 40.9 / 17.5 + 23.8 / 17.8 + 27.7, 1024: 20.9 + 35.8 / 21.2 + 25.2 / 18.9 + 28.7
int g_wgrad2_tile_max = 64;    // key 20.  Round 3: 64 x 64 output tiles for every role-split wgrad.  A launch's split-K partials
                               // are (workgroups x tile bytes): 256 x 64 KB = 16 MB at 128 x 128, 512 x 16 KB = 8 MB now -- half the
                               // partial traffic of the 13 launches per step -- and inside the step the smaller tiles with 512
                               // workgroups measure 2.74-2.75 ms against 2.77-2.78 (A/B on one box; 384 / 768 / 1024 workgroups:
                               // 2.79).  Alone the 128 x 128 tiles are 25-45 % faster (round 2), but these launches run beside the
                               // dgrad chain and their partials compete with it for HBM.
inline int wgrad2_mt(int cout) { return (cout >= 128 && g_wgrad2_tile_max >= 128) ? 128 : 64; }
int g_wgrad2_nt_max = 0;       // key 23: 0 = follow key 20; 128 = 64 x 128 tiles (x is the single-tensor operand: (2 M + N) loads per M N products)
inline int wgrad2_nt(int cin) { return (cin >= 128 && (g_wgrad2_nt_max ? g_wgrad2_nt_max : g_wgrad2_tile_max) >= 128) ? 128 : 64; }
inline int wgrad_split_len(int b, int cin, int cout, int P) {
  if (wgrad2_ok(cin, cout)) {
    // one workgroup of eight waves per CU: as many splits as keep tiles * splits at or under the target
    const long long tiles = (long long)ceil_div(cout, wgrad2_mt(cout)) * ceil_div(cin, wgrad2_nt(cin));
    long long want = g_wgrad2_target / tiles;
    if (want < 1) want = 1;
    const long long total = (long long)b * P;
    long long len = (total + want - 1) / want;
    len = (len + kKTW - 1) / kKTW * kKTW;
    if (len < 4 * kKTW) len = 4 * kKTW;
    return (int)len;
  }
  // Split-K partials cost cout*cin*4 bytes per split (written here, read back by the reduce): aim for ~1024
  // workgroups when the output is small, ~768 when it is large (PMC: at 1024 the partials of a 128x256
  // layer were as much HBM traffic as its activations; at 256-512 the launch no longer fills the chip).
  const long long tiles = wgrad_small(cin, cout)
                              ? 1 : (long long)ceil_div(cout, wgrad_mt(cout, (long long)b * P)) * ceil_div(cin, wgrad_nt(cin, (long long)b * P));
  const long long out_elems = (long long)cout * cin;
  const long long target = out_elems >= 128 * 128 ? g_wg_target_big : g_wg_target_small;  // measured: fewer workgroups lose more than the partials cost
  long long want = (target + tiles - 1) / tiles;  // splits over the flattened (cloud, point) range
  if (want < 1) want = 1;
  const long long total = (long long)b * P;
  long long len = (total + want - 1) / want;
  len = (len + kKTW - 1) / kKTW * kKTW;
  if (len < 4 * kKTW) len = 4 * kKTW;
  return (int)len;
}
inline int wgrad_splits(int b, int cin, int cout, int P) {
  const long long total = (long long)b * P;
  return (int)((total + wgrad_split_len(b, cin, cout, P) - 1) / wgrad_split_len(b, cin, cout, P));
}

}  // namespace

extern "C" {

#ifdef ISTNET_TRACE
ISTNET_PN2_API int istnet_pw_trace_read(unsigned long long* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 8 * 8192);
}
#endif

int istnet_pw_tile_cfg(int b, int m, int p) {
  const TileCfg c = pick_cfg(b, m, p);
  return cfg_mt(c) * 1000 + cfg_nt(c);  // e.g. 128128, 64128, 64064, 32256
}

int istnet_pw_dgrad_tile_cfg(int b, int m, int p) {
  const TileCfg c = pick_dgrad_cfg(b, m, p, g_force_dgrad_cfg);
  return cfg_mt(c) * 1000 + cfg_nt(c);
}

int istnet_pw_wgrad_tile_cfg(int b, int cin, int cout, int p) {
  const long long pts = (long long)b * p;
  if (wgrad2_ok(cin, cout)) return 1000000 + wgrad2_mt(cout) * 1000 + wgrad2_nt(cin);   // pw_wgrad2_kernel (dense input)
  return wgrad_small(cin, cout) ? 32032 : wgrad_mt(cout, pts) * 1000 + wgrad_nt(cin, pts);
}

#ifdef ISTNET_PHASE_TIMING
__attribute__((visibility("default"))) int istnet_debug_mid_phase_read(unsigned long long* out102, int reset) {
  unsigned long long zero[102] = {0};
  if (hipMemcpyFromSymbol(out102, HIP_SYMBOL(g_mid_phase), 96 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out102 + 96, HIP_SYMBOL(g_mid_wgs), 6 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mid_phase), zero, 96 * sizeof(unsigned long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mid_wgs), zero, 6 * sizeof(unsigned long long));
  }
  return 0;
}
__attribute__((visibility("default"))) int istnet_debug_phase_read(unsigned long long* out18, int reset) {
  unsigned long long zero[18] = {0};
  if (hipMemcpyFromSymbol(out18, HIP_SYMBOL(g_phase_sum), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out18 + 16, HIP_SYMBOL(g_phase_wgs), 2 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_sum), zero, 16 * sizeof(unsigned long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_wgs), zero, 2 * sizeof(unsigned long long));
  }
  return 0;
}
#endif
int istnet_pw_set_tuning(int key, int value) {
  switch (key) {
    case 0: g_wg_small_pts = value; return 0;
    case 1: g_wg_target_big = value; return 0;
    case 2: g_wg_target_small = value; return 0;
    case 3: g_force_fwd_cfg = value; return 0;
    case 4: g_force_dgrad_cfg = value; return 0;
    case 5: g_bwd_small_target = value > 0 ? value : 512; return 0;
    case 6: g_exp_no_fast = value; return 0;
    case 7: g_dgrad_min_wgs = value > 0 ? value : 384; return 0;
    case 8: g_bwd_mid_target = value > 0 ? value : 128; return 0;
    case 9: g_bwd_mid_enable = value != 0; return 0;
    case 11: g_wgrad2_enable = value != 0; return 0;
    case 13: g_fwd2_enable = value != 0; return 0;
    case 15: g_fwd_sk_enable = value != 0; return 0;
    case 17: g_dgrad_sk_enable = value != 0; return 0;
    case 19: g_dgrad_rs_enable = value != 0; return 0;
    case 18: g_dgrad_sk_min_k = value > 0 ? value : 256; return 0;
    case 16: g_fwd_sk_max_tiles = value > 0 ? value : 1024; return 0;
    case 14: g_fwd2_min_waves = value > 0 ? value : 1024; return 0;
    case 12: g_wgrad2_target = value > 0 ? value : 512; return 0;
    case 20: g_wgrad2_tile_max = value >= 128 ? 128 : 64; return 0;
    case 21: g_scatter_csr_threads = value; return 0;
    case 22: g_interp_dy_lds = value != 0; return 0;
    case 23: g_wgrad2_nt_max = value >= 128 ? 128 : (value > 0 ? 64 : 0); return 0;
    case 24: g_sk_tm2_min_wgs = value > 0 ? value : 0; return 0;
    default: return ISTNET_PN2_EINVAL;
  }
}

int istnet_pw_get_tuning(int key) {
  switch (key) {
    case 5: return g_bwd_small_target;
    case 7: return g_dgrad_min_wgs;
    case 8: return g_bwd_mid_target;
    case 12: return g_wgrad2_target;
    case 14: return g_fwd2_min_waves;
    case 16: return g_fwd_sk_max_tiles;
    case 18: return g_dgrad_sk_min_k;
    case 20: return g_wgrad2_tile_max;
    case 21: return g_scatter_csr_threads;
    default: return -1;
  }
}

int istnet_pw_stat_tiles(int b, int cout, int p) {
  return b * ceil_div(p, cfg_nt(pick_cfg(b, cout, p, g_force_fwd_cfg)));
}

// ---- pw_fwd_sk_kernel (small launches: no LDS operands, K split over the waves of a workgroup) ----
static bool fwd_sk_ok(int b, int cin, int cout, int p) {
  // 128-point tiles of the flattened (cloud, point) axis: whole clouds (p % 128 == 0) or several small ones (128 % p == 0)
  if (!g_fwd_sk_enable || cin % 8 || cin > 2048 || cin < 64 || (p % 128 && (128 % p || p % 4)) || ((long long)b * p) % 128 ||
      cout < 32)
    return false;
  const long long tiles = (long long)b * p / 128 * ceil_div(cout, 32);
  return tiles <= g_fwd_sk_max_tiles;
}
// row blocks per workgroup of a split-K launch over `tiles` 128-point tiles and `rows` output rows
static int sk_tm(int tiles, int rows) {
  return (g_sk_tm2_min_wgs > 0 && rows >= 64 && (long long)tiles * ceil_div(rows, 64) >= g_sk_tm2_min_wgs) ? 2 : 1;
}
// ---- pw_fwd2_kernel (dense input, B operand straight from global memory) ----
// 0: pw_fwd_kernel; else TMW * 1000 + WM * 100 + WN * 10 + (KC == 32)
static int fwd2_cfg(int b, int cin, int cout, int p) {
  if (!g_fwd2_enable || cin % 16 || cin > 1024 || p % 128 || cout < 16) return 0;
  const long long wave_tiles = (long long)b * (p / 128);
  int tmw, wm, wn;
  if (cout <= 32) { tmw = 1; wm = 1; wn = 4; }
  else if (cout <= 64) { tmw = 2; wm = 1; wn = 4; }
  else if (cout <= 128) { tmw = 2; wm = 2; wn = 2; }
  else { tmw = 2; wm = 4; wn = 1; }
  // A wave owns 32 TMW rows x 128 points: the launch needs about one wave per SIMD of the chip (1024) to hide its
  // loads; measured at 256 - 512 waves (the FP levels) the 64 x 64 LDS tiles are 1.5 - 3x faster.
  if (wave_tiles * ceil_div(cout, 32 * tmw) < g_fwd2_min_waves) return 0;
  while (wn > 1 && (wave_tiles / wn) * ceil_div(cout, 32 * tmw * wm) < 256) { wn /= 2; wm *= 2; }
  if (p % (128 * wn)) return 0;
  const int kc = (wm > 1 || cin % 32) ? 16 : 32;       // more than one row block of weights per workgroup: 16-channel chunks
                                                        // (registers of the staging loads, As under 64 KB)
  return tmw * 1000 + wm * 100 + wn * 10 + (kc == 32);
}
int istnet_pw_forward_cfg(int b, int cin, int cout, int p) {
  const int c2 = fwd2_cfg(b, cin, cout, p);
  return c2 ? c2 : (fwd_sk_ok(b, cin, cout, p) ? 1 : 0);      // 1: pw_fwd_sk_kernel
}

static int launch_pw_fwd2(int cfg, int b, int cin, int cout, int p, const float* x, const float* w,
                          const float* in_scale, const float* in_shift, float* y, float* part_sum, float* part_sq,
                          void* stream) {
  const int tmw = cfg / 1000, wm = (cfg / 100) % 10, wn = (cfg / 10) % 10, kc32 = cfg % 10;
  const int tpc = p / (128 * wn);
  const dim3 grid(tpc * b, ceil_div(cout, 32 * tmw * wm));
#define ISTNET_FWD2(TMW, WM, WN, KC)                                                                               \
  hipLaunchKernelGGL((pw_fwd2_kernel<TMW, WM, WN, KC>), grid, dim3(kThreads), 0, as_stream(stream), cin, cout, p, tpc, x, \
                     w, in_scale, in_shift, y, part_sum, part_sq, tpc * b)
#define ISTNET_FWD2_K(TMW, WM, WN)                                                                                 \
  do {                                                                                                             \
    if (kc32) ISTNET_FWD2(TMW, WM, WN, 32); else ISTNET_FWD2(TMW, WM, WN, 16);                                     \
  } while (0)
  if (tmw == 1) {
    if (wn == 4) ISTNET_FWD2_K(1, 1, 4); else if (wn == 2) ISTNET_FWD2(1, 2, 2, 16); else ISTNET_FWD2(1, 4, 1, 16);
  } else {
    if (wn == 4) ISTNET_FWD2_K(2, 1, 4); else if (wn == 2) ISTNET_FWD2(2, 2, 2, 16); else ISTNET_FWD2(2, 4, 1, 16);
  }
#undef ISTNET_FWD2_K
#undef ISTNET_FWD2
  return (int)hipGetLastError();
}

static int launch_pw_forward(int b, int cin, int cout, int p, const float* x,
                             const float* w, int ldw, const float* in_scale, const float* in_shift, float* y,
                             float* part_sum, float* part_sq, void* stream, const float* c_init = nullptr,
                             const int* ncols = nullptr, const float* colw = nullptr, const MultiSrc* msrc_p = nullptr,
                             const float* row_init = nullptr) {
  MultiSrc msrc;
  msrc.n = 0;
  if (msrc_p != nullptr) msrc = *msrc_p;
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0 || (p & 3)) return ISTNET_PN2_EINVAL;
  if (msrc_p == nullptr && ncols == nullptr && row_init == nullptr && fwd_sk_ok(b, cin, cout, p)) {
    const int tiles = (int)((long long)b * p / 128);          // of the flattened (cloud, point) axis
    if (sk_tm(tiles, cout) == 2)
      hipLaunchKernelGGL(pw_fwd_sk_kernel<2>, dim3(tiles, ceil_div(cout, 64)), dim3(kThreads), 0, as_stream(stream), cin,
                         cout, p, 0, x, w, ldw, in_scale, in_shift, c_init, y, part_sum, part_sq, tiles, nullptr, 0, nullptr,
                         nullptr);
    else
      hipLaunchKernelGGL(pw_fwd_sk_kernel<1>, dim3(tiles, ceil_div(cout, 32)), dim3(kThreads), 0, as_stream(stream), cin,
                         cout, p, 0, x, w, ldw, in_scale, in_shift, c_init, y, part_sum, part_sq, tiles, nullptr, 0, nullptr,
                         nullptr);
    return (int)hipGetLastError();
  }
  const TileCfg cfg = pick_cfg(b, cout, p, g_force_fwd_cfg);
  const int tpc = ceil_div(p, cfg_nt(cfg));
  const dim3 grid(tpc * b, ceil_div(cout, cfg_mt(cfg)));
  const int nt = tpc * b;
#define ISTNET_FWD(MT, NT, WM, WN)                                                                          \
  hipLaunchKernelGGL((pw_fwd_kernel<MT, NT, WM, WN>), grid, dim3(kThreads), 0, as_stream(stream), cin, cout, p, tpc, x, w, \
                     in_scale, in_shift, y, part_sum, part_sq, nt, ldw, c_init, ncols, colw, msrc, row_init)
  switch (cfg) {
    case kCfg128x128: ISTNET_FWD(128, 128, 2, 2); break;
    case kCfg64x128: ISTNET_FWD(64, 128, 2, 2); break;
    case kCfg64x64: ISTNET_FWD(64, 64, 2, 2); break;
    case kCfg32x256: ISTNET_FWD(32, 256, 1, 4); break;
    default: return ISTNET_PN2_EINVAL;     // the tall 64-point tiles exist for dgrad only
  }
#undef ISTNET_FWD
  return (int)hipGetLastError();
}

int istnet_pw_forward(int b, int cin, int cout, int p, const float* x, const float* w,
                      const float* in_scale, const float* in_shift, float* y, float* part_sum,
                      float* part_sq, void* stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0 || (p & 3)) return ISTNET_PN2_EINVAL;
  const int cfg2 = fwd2_cfg(b, cin, cout, p);
  if (cfg2) return launch_pw_fwd2(cfg2, b, cin, cout, p, x, w, in_scale, in_shift, y, part_sum, part_sq, stream);
  return launch_pw_forward(b, cin, cout, p, x, w, cin, in_scale, in_shift, y, part_sum, part_sq,
                           stream);
}

int istnet_pw_forward_tiles(int b, int cin, int cout, int p) {
  const int cfg2 = fwd2_cfg(b, cin, cout, p);
  if (cfg2) return b * (p / (128 * ((cfg2 / 10) % 10)));
  if (fwd_sk_ok(b, cin, cout, p)) return (int)((long long)b * p / 128);
  return istnet_pw_stat_tiles(b, cout, p);
}

int istnet_pw_forward_ld_tiles(int b, int cin, int cout, int p) {
  return fwd_sk_ok(b, cin, cout, p) ? (int)((long long)b * p / 128) : istnet_pw_stat_tiles(b, cout, p);
}

int istnet_pw_forward_ld(int b, int cin, int cout, int p, const float* x, const float* w, int ldw,
                         const float* in_scale, const float* in_shift, float* y, float* part_sum,
                         float* part_sq, void* stream) {
  if (ldw < cin) return ISTNET_PN2_EINVAL;
  return launch_pw_forward(b, cin, cout, p, x, w, ldw, in_scale, in_shift, y, part_sum, part_sq,
                           stream);
}

int istnet_pw_forward_acc(int b, int cin, int cout, int p, const float* x, const float* w, int ldw,
                          const float* c_init, float* y, float* part_sum, float* part_sq, void* stream) {
  if (ldw < cin || c_init == nullptr) return ISTNET_PN2_EINVAL;
  return launch_pw_forward(b, cin, cout, p, x, w, ldw, nullptr, nullptr, y, part_sum, part_sq,
                           stream, c_init);
}

// y = three_interpolate(zk, idx, weight) + w . x for a launch the split-K kernel takes (istnet_pw_forward_cfg == 1):
// istnet_pw_forward_acc without the interpolated tensor
int istnet_pw_forward_acc_interp(int b, int cin, int cout, int p, const float* x, const float* w, int ldw, const float* zk,
                                 int m, const int* idx, const float* weight, float* y, float* part_sum, float* part_sq,
                                 void* stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0 || m <= 0 || ldw < cin || !x || !w || !zk || !idx || !weight || !y)
    return ISTNET_PN2_EINVAL;
  if (!fwd_sk_ok(b, cin, cout, p)) return ISTNET_PN2_EINVAL;
  const int tiles = (int)((long long)b * p / 128);
  if (sk_tm(tiles, cout) == 2)
    hipLaunchKernelGGL(pw_fwd_sk_kernel<2>, dim3(tiles, ceil_div(cout, 64)), dim3(kThreads), 0, as_stream(stream), cin, cout,
                       p, 0, x, w, ldw, nullptr, nullptr, nullptr, y, part_sum, part_sq, tiles, zk, m, idx, weight);
  else
    hipLaunchKernelGGL(pw_fwd_sk_kernel<1>, dim3(tiles, ceil_div(cout, 32)), dim3(kThreads), 0, as_stream(stream), cin, cout,
                       p, 0, x, w, ldw, nullptr, nullptr, nullptr, y, part_sum, part_sq, tiles, zk, m, idx, weight);
  return (int)hipGetLastError();
}

int istnet_pw_forward_multi(int b, int nsrc, const float* const* srcs, const int* chans, int cout, int p,
                            const float* w, int ldw, const float* row_init, float* y, void* stream) {
  if (nsrc <= 0 || nsrc > kMaxSrc || srcs == nullptr || chans == nullptr) return ISTNET_PN2_EINVAL;
  MultiSrc ms;
  ms.n = nsrc;
  ms.cbeg[0] = 0;
  for (int s = 0; s < nsrc; ++s) {
    if (srcs[s] == nullptr || chans[s] <= 0 || (chans[s] % kKT)) return ISTNET_PN2_EINVAL;   // chunks must not straddle sources
    ms.ptr[s] = srcs[s];
    ms.cbeg[s + 1] = ms.cbeg[s] + chans[s];
  }
  for (int s = nsrc; s < kMaxSrc; ++s) { ms.ptr[s] = nullptr; ms.cbeg[s + 1] = ms.cbeg[nsrc]; }
  const int cin = ms.cbeg[nsrc];
  if (ldw < cin) return ISTNET_PN2_EINVAL;
  return launch_pw_forward(b, cin, cout, p, nullptr, w, ldw, nullptr, nullptr, y, nullptr, nullptr,
                           stream, nullptr, nullptr, nullptr, &ms, row_init);
}

int istnet_pw_forward_cols(int cin, int cout, long long cap, const float* x, const float* w, const float* in_scale,
                           const float* in_shift, float* y, float* part_sum, float* part_sq, const int* ncols,
                           const float* colw, void* stream) {
  if (cap <= 0 || (cap & 255) || cap >= (1LL << 31) || ncols == nullptr || colw == nullptr) return ISTNET_PN2_EINVAL;
  return launch_pw_forward(1, cin, cout, (int)cap, x, w, cin, in_scale, in_shift, y, part_sum,
                           part_sq, stream, nullptr, ncols, colw);
}

int istnet_pw_gather_add_tiles(int b, int p) { return b * ceil_div(p, 256); }

int istnet_pw_gather_add(int b, int n, int npoint, int nsample, int cout, const float* xyz, const float* new_xyz,
                         const int* idx, const float* z, const float* w0, int ldw, float* y, float* part_sum,
                         float* part_sq, void* stream) {
  if (b <= 0 || n <= 0 || npoint <= 0 || nsample <= 0 || cout <= 0 || ldw < 3) return ISTNET_PN2_EINVAL;
  const int p = npoint * nsample;
  const dim3 grid(ceil_div(p, 256), b, ceil_div(cout, kGatherAddCO));
  hipLaunchKernelGGL(pw_gather_add_kernel, grid, dim3(256), 0, as_stream(stream), n, p, nsample, cout, ldw, xyz,
                     new_xyz, idx, z, w0, y, part_sum, part_sq, (int)(grid.x * b));
  return (int)hipGetLastError();
}

int istnet_bn_finalize_fwd_nbt(int c, int nt, double count, const float* part_sum, const float* part_sq,
                               const float* gamma, const float* beta, float eps, const float* momentum,
                               float* running_mean, float* running_var, float* bn, long long* num_batches_tracked,
                               void* stream) {
  if (c <= 0 || nt <= 0 || (running_mean != nullptr && momentum == nullptr)) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(c), dim3(kFinThreads), 0, as_stream(stream), c, nt, count,
                     part_sum, part_sq, gamma, beta, eps, momentum, running_mean, running_var, bn, num_batches_tracked);
  return (int)hipGetLastError();
}
int istnet_bn_finalize_fwd(int c, int nt, double count, const float* part_sum, const float* part_sq,
                           const float* gamma, const float* beta, float eps, const float* momentum,
                           float* running_mean, float* running_var, float* bn, void* stream) {
  return istnet_bn_finalize_fwd_nbt(c, nt, count, part_sum, part_sq, gamma, beta, eps, momentum, running_mean, running_var, bn,
                                    nullptr, stream);
}

int istnet_bn_relu_pool(int b, int c, int g, int s, const float* y, const float* bn, float* out,
                        long long out_bstride, unsigned char* arg, float* ymax, void* stream) {
  if (out_bstride <= 0) out_bstride = (long long)c * g;
  if (b <= 0 || c <= 0 || g <= 0 || s <= 0) return ISTNET_PN2_EINVAL;
  const float* scale = bn;
  const float* shift = bn + c;
  if (s == 1) {
    const long long P = g;
    if ((P & 3) || out_bstride != (long long)c * g) return ISTNET_PN2_EINVAL;
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(ceil_div((int)(P / 4), 256), grid_rows(b * c)), dim3(256), 0,
                       as_stream(stream), c, (int)(P / 4), y, scale, shift, out, b * c);
    return (int)hipGetLastError();
  }
  const dim3 grid(ceil_div(g, 256), grid_rows(b * c));
#define ISTNET_POOL(S4)                                                                                  \
  hipLaunchKernelGGL((bn_relu_pool_kernel<S4>), grid, dim3(256), 0, as_stream(stream), c, g, y, scale,  \
                     shift, out, out_bstride, arg, ymax, b * c)
  switch (s) {
    case 4: ISTNET_POOL(1); break;
    case 8: ISTNET_POOL(2); break;
    case 16: ISTNET_POOL(4); break;
    case 32: ISTNET_POOL(8); break;
    case 64: ISTNET_POOL(16); break;
    default: return ISTNET_PN2_EINVAL;
  }
#undef ISTNET_POOL
  return (int)hipGetLastError();
}

// chunks of clouds per channel: about 32 KB of y per workgroup, so that the redundant reduction of the channel's partials
// (8 nt bytes) stays small beside the rows it serves
static int fin_chunks(int b, long long row_bytes) {
  long long ch = (long long)b * row_bytes / 32768;
  if (ch < 1) ch = 1;
  if (ch > b) ch = b;
  return (int)ch;
}

int istnet_bn_fin_relu_pool_nbt(int b, int c, int g, int s, int nt, double count, const float* part_sum, const float* part_sq,
                                const float* gamma, const float* beta, float eps, const float* momentum, float* running_mean,
                                float* running_var, float* bn, const float* y, float* out, long long out_bstride,
                                unsigned char* arg, float* ymax, long long* num_batches_tracked, void* stream) {
  if (out_bstride <= 0) out_bstride = (long long)c * g;
  if (b <= 0 || c <= 0 || g <= 0 || nt <= 0 || count <= 0.0 || !part_sum || !part_sq || !gamma || !beta || !bn || !y ||
      !out || (running_mean != nullptr && momentum == nullptr))
    return ISTNET_PN2_EINVAL;
  if (s == 1) {
    if ((g & 3) || out_bstride != (long long)c * g) return ISTNET_PN2_EINVAL;
    hipLaunchKernelGGL(bn_fin_relu_apply_kernel, dim3(c, fin_chunks(b, 4LL * g)), dim3(kFinThreads), 0, as_stream(stream),
                       c, b, g / 4, nt, count, part_sum, part_sq, gamma, beta, eps, momentum, running_mean, running_var, bn,
                       y, out, num_batches_tracked);
    return (int)hipGetLastError();
  }
  if (arg == nullptr) return ISTNET_PN2_EINVAL;
  const dim3 grid(c, fin_chunks(b, 4LL * g * s));
#define ISTNET_FPOOL(S4)                                                                                             \
  hipLaunchKernelGGL((bn_fin_relu_pool_kernel<S4>), grid, dim3(kFinThreads), 0, as_stream(stream), c, b, g, nt, count, \
                     part_sum, part_sq, gamma, beta, eps, momentum, running_mean, running_var, bn, y, out, out_bstride, \
                     arg, ymax, num_batches_tracked)
  switch (s) {
    case 4: ISTNET_FPOOL(1); break;
    case 8: ISTNET_FPOOL(2); break;
    case 16: ISTNET_FPOOL(4); break;
    case 32: ISTNET_FPOOL(8); break;
    case 64: ISTNET_FPOOL(16); break;
    default: return ISTNET_PN2_EINVAL;
  }
#undef ISTNET_FPOOL
  return (int)hipGetLastError();
}

int istnet_bn_fin_relu_pool(int b, int c, int g, int s, int nt, double count, const float* part_sum, const float* part_sq,
                            const float* gamma, const float* beta, float eps, const float* momentum, float* running_mean,
                            float* running_var, float* bn, const float* y, float* out, long long out_bstride,
                            unsigned char* arg, float* ymax, void* stream) {
  return istnet_bn_fin_relu_pool_nbt(b, c, g, s, nt, count, part_sum, part_sq, gamma, beta, eps, momentum, running_mean,
                                     running_var, bn, y, out, out_bstride, arg, ymax, nullptr, stream);
}

int istnet_interp_grad_csr_dy(int b, int c, int n, int m, const float* y, const float* d_dense, const float* bn,
                              const float* bwdc, const float* weight, const int* offsets, const int* entries,
                              float* grad_points, void* stream) {
  if (b <= 0 || c <= 0 || m <= 0 || n <= 0 || !y || !d_dense || !bn || !bwdc || !weight || !offsets || !entries ||
      !grad_points)
    return ISTNET_PN2_EINVAL;
  const size_t lds = (size_t)kInterpDyCH * n * 4;
  if (g_interp_dy_lds && (n & 3) == 0 && lds <= 64 * 1024) {
    if (m >= 512)
      hipLaunchKernelGGL(interp_grad_csr_dy_lds_kernel<512>, dim3(ceil_div(c, kInterpDyCH), b), dim3(512), lds,
                         as_stream(stream), c, n, m, y, d_dense, bn, bwdc, weight, offsets, entries, grad_points);
    else
      hipLaunchKernelGGL(interp_grad_csr_dy_lds_kernel<256>, dim3(ceil_div(c, kInterpDyCH), b), dim3(256), lds,
                         as_stream(stream), c, n, m, y, d_dense, bn, bwdc, weight, offsets, entries, grad_points);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(interp_grad_csr_dy_kernel, dim3(ceil_div(m, 256), ceil_div(c, kInterpDyCH), b), dim3(256), 0,
                     as_stream(stream), c, n, m, y, d_dense, bn, bwdc, weight, offsets, entries, grad_points);
  return (int)hipGetLastError();
}

int istnet_bn_relu_mean(int b, int c, int p, const float* y, const float* bn, float* out, void* stream) {
  if (b <= 0 || c <= 0 || p <= 0 || (p & 3) || !y || !bn || !out) return ISTNET_PN2_EINVAL;
  const int rows = b * c;
  hipLaunchKernelGGL(bn_relu_mean_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, as_stream(stream), c, p, rows, y, bn,
                     bn + c, out);
  return (int)hipGetLastError();
}
int istnet_expand_rows(int rows, int p, const float* g, float* out, void* stream) {
  if (rows <= 0 || p <= 0 || (p & 3) || !g || !out) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(expand_rows_kernel, dim3(ceil_div(p / 4, 256), grid_rows(rows)), dim3(256), 0, as_stream(stream),
                     p / 4, rows, 1.f / (float)p, g, out);
  return (int)hipGetLastError();
}

int istnet_affine_consts(int c, const float* gamma, const float* beta, const float* mean, const float* var,
                         float eps, float* bn, void* stream) {
  if (c <= 0 || beta == nullptr || bn == nullptr) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(affine_consts_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, as_stream(stream), c, gamma, beta,
                     mean, var, eps, bn);
  return (int)hipGetLastError();
}

int istnet_affine_consts_multi(int n, const int* c, const float* const* gamma, const float* const* beta,
                               const float* const* mean, const float* const* var, const float* eps,
                               float* const* bn, void* stream) {
  if (n <= 0 || n > 8 || c == nullptr || gamma == nullptr || beta == nullptr || mean == nullptr || var == nullptr ||
      eps == nullptr || bn == nullptr)
    return ISTNET_PN2_EINVAL;
  AffineBatch ab;
  ab.n = n;
  ab.block_begin[0] = 0;
  for (int l = 0; l < n; ++l) {
    if (c[l] <= 0 || beta[l] == nullptr || bn[l] == nullptr) return ISTNET_PN2_EINVAL;
    ab.gamma[l] = gamma[l]; ab.beta[l] = beta[l]; ab.mean[l] = mean[l]; ab.var[l] = var[l]; ab.bn[l] = bn[l];
    ab.c[l] = c[l]; ab.eps[l] = eps[l];
    ab.block_begin[l + 1] = ab.block_begin[l] + ceil_div(c[l], 256);
  }
  hipLaunchKernelGGL(affine_consts_multi_kernel, dim3(ab.block_begin[n]), dim3(256), 0, as_stream(stream), ab);
  return (int)hipGetLastError();
}

int istnet_affine_apply(int b, int c, int p, int relu, const float* y, const float* bn, float* out, void* stream) {
  if (b <= 0 || c <= 0 || p <= 0 || (p & 3)) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(affine_apply_kernel, dim3(ceil_div(p / 4, 256), grid_rows(b * c)), dim3(256), 0, as_stream(stream), c,
                     p / 4, relu, y, bn, bn + c, out, b * c);
  return (int)hipGetLastError();
}

int istnet_pw_bwd_stat_tiles(int b, int p) { return b * ceil_div(p, kStatChunk); }

int istnet_pw_channel_stats(int b, int c, int p, const float* y, float* part_sum, float* part_sq, void* stream) {
  if (b <= 0 || c <= 0 || p <= 0 || (p & 3)) return ISTNET_PN2_EINVAL;
  const dim3 grid(ceil_div(p, 4096), c, b);   // partials [c][istnet_pw_bwd_stat_tiles(b, p)]
  hipLaunchKernelGGL(pw_channel_stats_kernel, grid, dim3(256), 0, as_stream(stream), c, p, y, part_sum, part_sq,
                     (int)(grid.x * b));
  return (int)hipGetLastError();
}

int istnet_pw_interp_stats_tiles(int b, int n) { return b * ceil_div(n, 256); }

int istnet_pw_interp_stats(int b, int c, int m, int n, const float* points, const int* idx, const float* weight, float* out,
                           float* part_sum, float* part_sq, void* stream) {
  if (b <= 0 || c <= 0 || m <= 0 || n <= 0 || !points || !idx || !weight || !out || !part_sum || !part_sq) return ISTNET_PN2_EINVAL;
  const dim3 grid(ceil_div(n, 256), ceil_div(c, 8), b);   // partials [c][istnet_pw_interp_stats_tiles(b, n)]
  hipLaunchKernelGGL(interp_stats_kernel, grid, dim3(256), 0, as_stream(stream), c, m, n, points, idx, weight, out, part_sum,
                     part_sq, (int)(grid.x * b));
  return (int)hipGetLastError();
}

int istnet_pw_dy(int b, int c, int p, const float* y, const float* d_dense, const float* bn, const float* bwdc,
                 float* out, void* stream) {
  if (b <= 0 || c <= 0 || p <= 0 || (p & 3)) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(pw_dy_kernel, dim3(ceil_div(p / 4, 256), grid_rows(b * c)), dim3(256), 0, as_stream(stream), c, p / 4, y,
                     d_dense, bn, bwdc, out, b * c);
  return (int)hipGetLastError();
}

int istnet_pw_bwd_stats_pooled(int b, int c, int g, const float* d_pooled, long long pooled_bstride,
                               const float* ymax, const float* bn, float* part_g, float* part_gy, void* stream) {
  if (b <= 0 || c <= 0 || g <= 0 || d_pooled == nullptr || ymax == nullptr) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(pw_bwd_stats_pooled_kernel, dim3(c, b), dim3(64), 0, as_stream(stream), c, g, d_pooled,
                     pooled_bstride > 0 ? pooled_bstride : (long long)c * g, ymax, bn, bn + c, part_g, part_gy);
  return (int)hipGetLastError();
}

int istnet_bn_bwd_dense_finalize(int b, int c, int p, double count, int training, const float* y, const float* d_dense,
                                 const float* gamma, const float* bn, float* dgamma, float* dbeta, float* bwdc,
                                 void* stream) {
  if (b <= 0 || c <= 0 || p <= 0 || (p & 3) || !y || !d_dense || !gamma || !bn || !dgamma || !dbeta || !bwdc) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(bn_bwd_dense_finalize_kernel, dim3(c), dim3(1024), 0, as_stream(stream), c, b, p, count, training, y,
                     d_dense, gamma, bn, dgamma, dbeta, bwdc);
  return (int)hipGetLastError();
}

int istnet_bn_bwd_pooled_finalize(int b, int c, int g, double count, int training, const float* d_pooled,
                                  long long pooled_bstride, const float* ymax, const float* gamma, const float* bn,
                                  float* dgamma, float* dbeta, float* bwdc, void* stream) {
  if (b <= 0 || c <= 0 || g <= 0 || !d_pooled || !ymax || !gamma || !bn || !dgamma || !dbeta || !bwdc) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(bn_bwd_pooled_finalize_kernel, dim3(c), dim3(64 * kPoolFinWaves), 0, as_stream(stream), c, b, g, count,
                     training, d_pooled, pooled_bstride > 0 ? pooled_bstride : (long long)c * g, ymax, gamma, bn, dgamma,
                     dbeta, bwdc);
  return (int)hipGetLastError();
}

int istnet_pw_bwd_stats(int b, int c, int p, int nsample, const float* y, const float* d_dense,
                        const float* d_pooled, long long pooled_bstride, const unsigned char* arg,
                        const float* bn, float* part_g, float* part_gy, void* stream) {
  const int GS_C = c;
  if (b <= 0 || c <= 0 || p <= 0 || (p & 3)) return ISTNET_PN2_EINVAL;
  if (d_dense == nullptr && (d_pooled == nullptr || arg == nullptr || nsample <= 0 || (nsample & 3)))
    return ISTNET_PN2_EINVAL;
  GradSrc gs{d_dense, d_pooled, arg, nsample, pooled_bstride > 0 ? pooled_bstride : (long long)GS_C * (nsample > 0 ? p / nsample : 0), GS_C};
  const int chunks = ceil_div(p, kStatChunk);
  hipLaunchKernelGGL(pw_bwd_stats_kernel, dim3(chunks, c, b), dim3(256), 0, as_stream(stream), c, p, gs, y,
                     bn, bn + c, part_g, part_gy, chunks * b);
  return (int)hipGetLastError();
}

int istnet_bn_finalize_bwd(int c, int nt, double count, int training, const float* part_g,
                           const float* part_gy, const float* gamma, const float* bn, float* dgamma,
                           float* dbeta, float* bwdc, void* stream) {
  if (c <= 0 || nt <= 0) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(c), dim3(kFinThreads), 0, as_stream(stream), c, nt, count,
                     training, part_g, part_gy, gamma, bn, dgamma, dbeta, bwdc);
  return (int)hipGetLastError();
}

// ---- pw_dgrad_sk_kernel (small launches, dense gradient source) ----
static bool dgrad_rs_ok(int m_rows, int cout, int p);
static int bwd_mid_len(int b, int cin, int p);
static bool dgrad_sk_ok(int b, int m_rows, int cout, int p) {
  // (measured, tools/bench_pw.py: +6-17 % at K = cout >= 256, 10-50 % SLOWER at K = 64-128 -- forming dY costs ~14 VALU
  //  per element and short K leaves nothing to hide it behind)
  if (!g_dgrad_sk_enable || cout % 8 || cout > 2048 || cout < g_dgrad_sk_min_k || (p % 128 && (128 % p || p % 4)) ||
      ((long long)b * p) % 128 || m_rows < 32)
    return false;
  return (long long)b * p / 128 * ceil_div(m_rows, 32) <= g_fwd_sk_max_tiles;
}
/* row blocks (1 or 2) per workgroup of a split-K forward launch with `rows` output rows (the template argument of
 * pw_fwd_sk_kernel as a trace names it) */
int istnet_pw_sk_tm(int b, int rows, int p) { return sk_tm((int)((long long)b * p / 128), rows); }
/* 1 when istnet_pw_dgrad with a dense gradient source runs the split-K kernel for this shape */
int istnet_pw_dgrad_sk(int b, int m_rows, int cout, int p) { return dgrad_sk_ok(b, m_rows, cout, p) ? 1 : 0; }
int istnet_pw_dgrad_rs(int b, int m_rows, int cout, int p, int dense) {
  return (dgrad_rs_ok(m_rows, cout, p) && !(dense && dgrad_sk_ok(b, m_rows, cout, p))) ? 1 : 0;
}
int istnet_pw_dgrad_tiles(int b, int m_rows, int cout, int p, int dense) {
  if (dense && dgrad_sk_ok(b, m_rows, cout, p)) return (int)((long long)b * p / 128);
  if (dgrad_rs_ok(m_rows, cout, p)) {       // (a launch with ci_off = 0, cin_total = m_rows and statistics)
    const int len = bwd_mid_len(b, m_rows, p);
    return (int)(((long long)b * p + len - 1) / len);
  }
  return b * ceil_div(p, cfg_nt(pick_dgrad_cfg(b, m_rows, p, g_force_dgrad_cfg)));
}

// ---- dgrad through the loader / MFMA-wave kernel (pw_bwd_mid_kernel<8, 4, POOLED, false>): cout = 256, all 128 input
// channels, statistics requested -- the last layer of an SA4 scale, whose weight matrix is too large for the fused form ----
static bool dgrad_rs_ok(int m_rows, int cout, int p) {
  return g_dgrad_rs_enable && cout == 256 && m_rows == 128 && p % 128 == 0;
}

static int launch_pw_dgrad(int b, int cin_total, int ci_off, int m_rows, int cout, int p, int nsample,
                           const float* w, const float* y, const float* d_dense, const float* d_pooled,
                           long long pooled_bstride, const unsigned char* arg, const float* bn, const float* bwdc,
                           float* dx, const float* y_in, const float* bn_in, float* part_g, float* part_gy,
                           void* stream, const int* ncols, const float* colw) {
  const int GS_C = cout;
  if (b <= 0 || m_rows <= 0 || cout <= 0 || p <= 0 || (p & 3)) return ISTNET_PN2_EINVAL;
  if (part_g != nullptr && (y_in == nullptr || bn_in == nullptr || part_gy == nullptr)) return ISTNET_PN2_EINVAL;
  if (d_dense == nullptr && (d_pooled == nullptr || arg == nullptr || nsample <= 0 || (nsample & 3)))
    return ISTNET_PN2_EINVAL;
  if (ncols == nullptr && part_g != nullptr && ci_off == 0 && cin_total == m_rows && dgrad_rs_ok(m_rows, cout, p) &&
      !(d_dense != nullptr && dgrad_sk_ok(b, m_rows, cout, p))) {
    GradSrc gsr{d_dense, d_pooled, arg, nsample, pooled_bstride > 0 ? pooled_bstride : (long long)GS_C * (nsample > 0 ? p / nsample : 0), GS_C};
    const int len = bwd_mid_len(b, m_rows, p);
    const int splits = (int)(((long long)b * p + len - 1) / len);
    constexpr size_t lds = MidCfg<8, 4>::LDS_BYTES;
    static PerDeviceOnce attr_once; unsigned long long attr_bit;
    if (attr_once.pending(attr_bit)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_bwd_mid_kernel<8, 4, false, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_bwd_mid_kernel<8, 4, true, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ISTNET_PN2_EINVAL;
      attr_once.done(attr_bit);
    }
    if (d_dense != nullptr)
      hipLaunchKernelGGL((pw_bwd_mid_kernel<8, 4, false, false>), dim3(splits), dim3(kMidThreads), lds, as_stream(stream), p,
                         (long long)b * p, len, w, y_in, bn_in, bn_in + m_rows, y, gsr, bn, bwdc, dx, part_g, part_gy, splits,
                         nullptr);
    else
      hipLaunchKernelGGL((pw_bwd_mid_kernel<8, 4, true, false>), dim3(splits), dim3(kMidThreads), lds, as_stream(stream), p,
                         (long long)b * p, len, w, y_in, bn_in, bn_in + m_rows, y, gsr, bn, bwdc, dx, part_g, part_gy, splits,
                         nullptr);
    return (int)hipGetLastError();
  }
  if (d_dense != nullptr && ncols == nullptr && dgrad_sk_ok(b, m_rows, cout, p)) {
    const int tiles = (int)((long long)b * p / 128);
    // (round 6, measured and not taken: the 64 x 128 form of this kernel.  Its 128 accumulators + two operand sets of 40 registers
    //  need the one-wave-per-SIMD register budget, and with ONE wave per SIMD the ~11 VALU per element that form dY no longer
    //  overlap a partner wave's MFMAs: 33.6 -> 71.7 us at 4096 x 512 x 512, 20.1 -> 39.6 at 8192 x 256 x 256,
    //  profiles/r06_splitk_tiles.txt.  The forward kernel has no such VALU work and takes the larger tile.)
    hipLaunchKernelGGL(pw_dgrad_sk_kernel<1>, dim3(tiles, ceil_div(m_rows, 32)), dim3(kThreads), 0, as_stream(stream),
                       cin_total, ci_off, m_rows, cout, p, 0, w, y, d_dense, bn, bwdc, dx, y_in, bn_in, part_g, part_gy,
                       tiles);
    return (int)hipGetLastError();
  }
  GradSrc gs{d_dense, d_pooled, arg, nsample, pooled_bstride > 0 ? pooled_bstride : (long long)GS_C * (nsample > 0 ? p / nsample : 0), GS_C};
  const TileCfg cfg = pick_dgrad_cfg(b, m_rows, p, g_force_dgrad_cfg);
  const int tpc = ceil_div(p, cfg_nt(cfg));
  const dim3 grid(tpc * b, ceil_div(m_rows, cfg_mt(cfg)));
  const bool interior = (m_rows % cfg_mt(cfg) == 0) && (p % cfg_nt(cfg) == 0) && (cout % kKT == 0) &&
                        (g_exp_no_fast == 0);
#define ISTNET_DGRAD(MT, NT, WM, WN)                                                                       \
  do {                                                                                                     \
    if (interior)                                                                                          \
      hipLaunchKernelGGL((pw_dgrad_kernel<MT, NT, WM, WN, true>), grid, dim3(kThreads), 0, as_stream(stream), \
                         cin_total, ci_off, m_rows, cout, p, tpc, w, y, gs, bn, bwdc, dx, y_in, bn_in, part_g,  \
                         part_gy, tpc * b, ncols, colw);                                                    \
    else                                                                                                   \
      hipLaunchKernelGGL((pw_dgrad_kernel<MT, NT, WM, WN, false>), grid, dim3(kThreads), 0, as_stream(stream), \
                         cin_total, ci_off, m_rows, cout, p, tpc, w, y, gs, bn, bwdc, dx, y_in, bn_in, part_g,  \
                         part_gy, tpc * b, ncols, colw);                                                    \
  } while (0)
  switch (cfg) {
    case kCfg128x128: ISTNET_DGRAD(128, 128, 2, 2); break;
    case kCfg64x128: ISTNET_DGRAD(64, 128, 2, 2); break;
    case kCfg64x64: ISTNET_DGRAD(64, 64, 2, 2); break;
    case kCfg32x256: ISTNET_DGRAD(32, 256, 1, 4); break;
    case kCfg128x64: ISTNET_DGRAD(128, 64, 2, 2); break;
    case kCfg256x64: ISTNET_DGRAD(256, 64, 4, 1); break;
  }
#undef ISTNET_DGRAD
  return (int)hipGetLastError();
}

int istnet_pw_dgrad(int b, int cin_total, int ci_off, int m_rows, int cout, int p, int nsample,
                    const float* w, const float* y, const float* d_dense, const float* d_pooled,
                    long long pooled_bstride, const unsigned char* arg, const float* bn, const float* bwdc,
                    float* dx, const float* y_in, const float* bn_in, float* part_g, float* part_gy,
                    void* stream) {
  return launch_pw_dgrad(b, cin_total, ci_off, m_rows, cout, p, nsample, w, y, d_dense, d_pooled, pooled_bstride, arg,
                         bn, bwdc, dx, y_in, bn_in, part_g, part_gy, stream, nullptr, nullptr);
}

int istnet_pw_dgrad_cols(int cin_total, int ci_off, int m_rows, int cout, long long cap, const float* w,
                         const float* y, const float* d_dense, const float* bn, const float* bwdc, float* dx,
                         const float* y_in, const float* bn_in, float* part_g, float* part_gy, const int* ncols,
                         const float* colw, void* stream) {
  if (cap <= 0 || (cap & 255) || cap >= (1LL << 31) || !d_dense || !ncols || !colw) return ISTNET_PN2_EINVAL;
  return launch_pw_dgrad(1, cin_total, ci_off, m_rows, cout, (int)cap, 0, w, y, d_dense, nullptr, 0, nullptr, bn, bwdc,
                         dx, y_in, bn_in, part_g, part_gy, stream, ncols, colw);
}

// dwx-only mode of istnet_pw_scatter_dy (out == NULL): point chunks per cloud so that ~1024 workgroups run
int istnet_pw_dwx_chunks(int b, int cout, int p) {
  const int wgs = b * ceil_div(cout, kScatterCH);
  int chunks = ceil_div(1024, wgs > 0 ? wgs : 1);
  const int maxc = ceil_div(p, 1024);   // at least four 256-point steps per workgroup
  if (chunks > maxc) chunks = maxc;
  return chunks < 1 ? 1 : chunks;
}

int istnet_pw_scatter_dy(int b, int cout, int n, int p, int nsample, const float* y, const float* d_dense,
                         const float* d_pooled, long long pooled_bstride, const unsigned char* arg,
                         const float* bn, const float* bwdc, const int* idx, float* out, long long out_bstride,
                         const float* xyz, const float* new_xyz, int group_nsample, float* dwx, void* stream) {
  const int GS_C = cout;
  if (b <= 0 || cout <= 0 || n <= 0 || p <= 0) return ISTNET_PN2_EINVAL;
  if (d_dense == nullptr && (d_pooled == nullptr || arg == nullptr || nsample <= 0)) return ISTNET_PN2_EINVAL;
  if (dwx != nullptr && (xyz == nullptr || new_xyz == nullptr || group_nsample <= 0 || p % group_nsample))
    return ISTNET_PN2_EINVAL;
  if (out == nullptr && dwx == nullptr) return ISTNET_PN2_EINVAL;
  const size_t lds = out != nullptr ? (size_t)kScatterCH * n * 4 : 0;
  if (lds > 64 * 1024) return ISTNET_PN2_EINVAL;
  const int chunks = out != nullptr ? 1 : istnet_pw_dwx_chunks(b, cout, p);
  GradSrc gs{d_dense, d_pooled, arg, nsample, pooled_bstride > 0 ? pooled_bstride : (long long)GS_C * (nsample > 0 ? p / nsample : 0), GS_C};
  hipLaunchKernelGGL(pw_scatter_dy_kernel, dim3(ceil_div(cout, kScatterCH), b, chunks), dim3(256), lds,
                     as_stream(stream), cout, n, p, y, gs, bn, bwdc, idx, out,
                     out_bstride > 0 ? out_bstride : (long long)cout * n, xyz, new_xyz,
                     group_nsample > 0 ? group_nsample : 1, dwx);
  return (int)hipGetLastError();
}

int istnet_pw_scatter_csr_chunks(int n) { (void)n; return 1; }

// channels per workgroup of the list scatter: the dY0 rows must fit 64 KB of LDS, and the grid should fill the chip
static int scatter_csr_ch(int b, int cout, int p) {
  int ch = 16;
  while (ch > 1 && ((size_t)ch * p * 4 > 64 * 1024 || (long long)b * ceil_div(cout, ch) < 512)) ch >>= 1;
  return ch;
}

static int scatter_dy_csr_impl(int b, int cout, int n, int p, const float* y, const float* d_dense, const float* bn,
                               const float* bwdc, const int* offsets, const int* entries, float* out,
                               long long out_bstride, const float* xyz, const float* new_xyz, int group_nsample,
                               float* dwx, void* stream, const BwdFinArgs& fin) {
  if (b <= 0 || cout <= 0 || n <= 0 || p <= 0 || (p & 3) || !y || !d_dense || !bn || !bwdc || !offsets || !entries || !out)
    return ISTNET_PN2_EINVAL;
  if (dwx != nullptr && (xyz == nullptr || new_xyz == nullptr || group_nsample <= 0 || p % group_nsample))
    return ISTNET_PN2_EINVAL;
  const int ch = scatter_csr_ch(b, cout, p);
  if ((size_t)ch * p * 4 > 64 * 1024) return ISTNET_PN2_EINVAL;    // a single row does not fit: caller uses the atomic kernel
  const size_t lds = (size_t)ch * p * 4 < 16 * 16 * 3 * 4 ? 16 * 16 * 3 * 4 : (size_t)ch * p * 4;
  const dim3 grid(ceil_div(cout, ch), b);
  const long long obs = out_bstride > 0 ? out_bstride : (long long)cout * n;
  const int gsz = group_nsample > 0 ? group_nsample : 1;
#define ISTNET_SCSR_NT(CH, NT)                                                                                       \
  hipLaunchKernelGGL((pw_scatter_csr_kernel<CH, NT>), grid, dim3(NT), lds, as_stream(stream), cout, n, p, y, d_dense, \
                     bn, bwdc, offsets, entries, out, obs, xyz, new_xyz, gsz, dwx, fin)
#define ISTNET_SCSR(CH)                                                                                            \
  do {                                                                                                             \
    const int nt_ = g_scatter_csr_threads > 0 ? g_scatter_csr_threads : (n >= 512 ? 1024 : (n >= 256 ? 512 : 256)); \
    /* register budget: 1024 threads leave 128 VGPRs (CH <= 4 fits), 512 threads 256 (CH <= 8) */                   \
    if constexpr (CH <= 4) { if (nt_ >= 1024) { ISTNET_SCSR_NT(CH, 1024); break; } }                               \
    if constexpr (CH <= 8) { if (nt_ >= 512) { ISTNET_SCSR_NT(CH, 512); break; } }                                 \
    ISTNET_SCSR_NT(CH, 256);                                                                                       \
  } while (0)
  switch (ch) {
    case 16: ISTNET_SCSR(16); break;
    case 8: ISTNET_SCSR(8); break;
    case 4: ISTNET_SCSR(4); break;
    case 2: ISTNET_SCSR(2); break;
    default: ISTNET_SCSR(1); break;
  }
#undef ISTNET_SCSR
#undef ISTNET_SCSR_NT
  return (int)hipGetLastError();
}

int istnet_pw_scatter_dy_csr(int b, int cout, int n, int p, const float* y, const float* d_dense, const float* bn,
                             const float* bwdc, const int* offsets, const int* entries, float* out,
                             long long out_bstride, const float* xyz, const float* new_xyz, int group_nsample,
                             float* dwx, void* stream) {
  BwdFinArgs fin{};
  return scatter_dy_csr_impl(b, cout, n, p, y, d_dense, bn, bwdc, offsets, entries, out, out_bstride, xyz, new_xyz,
                             group_nsample, dwx, stream, fin);
}

int istnet_pw_scatter_dy_csr_fin(int b, int cout, int n, int p, const float* y, const float* d_dense, const float* bn,
                                 int nt, double count, int training, const float* part_g, const float* part_gy,
                                 const float* gamma, float* dgamma, float* dbeta, float* bwdc, const int* offsets,
                                 const int* entries, float* out, long long out_bstride, const float* xyz,
                                 const float* new_xyz, int group_nsample, float* dwx, void* stream) {
  if (nt <= 0 || !part_g || !part_gy || !gamma || !dgamma || !dbeta || !bwdc) return ISTNET_PN2_EINVAL;
  BwdFinArgs fin{part_g, part_gy, gamma, dgamma, dbeta, bwdc, count, nt, training};
  return scatter_dy_csr_impl(b, cout, n, p, y, d_dense, bn, bwdc, offsets, entries, out, out_bstride, xyz, new_xyz,
                             group_nsample, dwx, stream, fin);
}

int istnet_pw_dgrad_stat_tiles(int b, int m_rows, int p) {
  return b * ceil_div(p, cfg_nt(pick_dgrad_cfg(b, m_rows, p, g_force_dgrad_cfg)));
}

int istnet_pw_wgrad_splits(int b, int cin, int cout, int p) { return wgrad_splits(b, cin, cout, p); }

static int launch_pw_wgrad(bool gather, int b, int cin, int cout, int p, int nsample, const float* x,
                           const GatherSrc& g, const float* in_scale, const float* in_shift, const float* y,
                           const float* d_dense, const float* d_pooled, long long pooled_bstride,
                           const unsigned char* arg, const float* bn, const float* bwdc, float* dw_part,
                           void* stream, const int* ncols = nullptr, const float* colw = nullptr, int fixed_splits = 0) {
  const int GS_C = cout;
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0 || (p % kKTW)) return ISTNET_PN2_EINVAL;
  if ((long long)b * p >= (1LL << 31)) return ISTNET_PN2_EINVAL;   // 32-bit point indexing in the kernels
  if (d_dense == nullptr && (d_pooled == nullptr || arg == nullptr || nsample <= 0 || (nsample & 3)))
    return ISTNET_PN2_EINVAL;
  GradSrc gs{d_dense, d_pooled, arg, nsample, pooled_bstride > 0 ? pooled_bstride : (long long)GS_C * (nsample > 0 ? p / nsample : 0), GS_C};
  const int len = wgrad_split_len(b, cin, cout, p);
  const long long total = (long long)b * p;
  if (ncols != nullptr && (wgrad_small(cin, cout) || fixed_splits <= 0)) return ISTNET_PN2_EINVAL;
  if (wgrad_small(cin, cout)) {
    const dim3 sgrid(wgrad_splits(b, cin, cout, p));
    if (gather)
      hipLaunchKernelGGL(pw_wgrad_small_kernel<true>, sgrid, dim3(kThreads), 0, as_stream(stream), cin, cout, p,
                         total, len, x, g, in_scale, in_shift, y, gs, bn, bwdc, dw_part);
    else
      hipLaunchKernelGGL(pw_wgrad_small_kernel<false>, sgrid, dim3(kThreads), 0, as_stream(stream), cin, cout, p,
                         total, len, x, g, in_scale, in_shift, y, gs, bn, bwdc, dw_part);
    return (int)hipGetLastError();
  }
  if (!gather && ncols == nullptr && wgrad2_ok(cin, cout)) {
    const int mt2 = wgrad2_mt(cout), nt2 = wgrad2_nt(cin);
    const dim3 grid2(wgrad_splits(b, cin, cout, p), ceil_div(cout, mt2), ceil_div(cin, nt2));
#define ISTNET_WGRAD2(MT, NT)                                                                                      \
  do {                                                                                                             \
    constexpr size_t lds = (2 * (MT + NT) * 36 + 5 * MT) * sizeof(float);                                          \
    static PerDeviceOnce attr_once; unsigned long long attr_bit;   /* > 64 KB of LDS per workgroup: opt in once per kernel */                     \
    if (attr_once.pending(attr_bit)) {                                                                                               \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_wgrad2_kernel<MT, NT, false>),                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||              \
          hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_wgrad2_kernel<MT, NT, true>),                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                \
        return ISTNET_PN2_EINVAL;                                                                                  \
      attr_once.done(attr_bit);                                                                                             \
    }                                                                                                              \
    if (d_dense != nullptr)                                                                                        \
      hipLaunchKernelGGL((pw_wgrad2_kernel<MT, NT, false>), grid2, dim3(kMidThreads), lds, as_stream(stream), cin, \
                         cout, p, total, len, x, in_scale, in_shift, y, gs, bn, bwdc, dw_part);                    \
    else                                                                                                           \
      hipLaunchKernelGGL((pw_wgrad2_kernel<MT, NT, true>), grid2, dim3(kMidThreads), lds, as_stream(stream), cin,  \
                         cout, p, total, len, x, in_scale, in_shift, y, gs, bn, bwdc, dw_part);                    \
  } while (0)
    if (mt2 == 128 && nt2 == 128) ISTNET_WGRAD2(128, 128);
    else if (mt2 == 128) ISTNET_WGRAD2(128, 64);
    else if (nt2 == 128) ISTNET_WGRAD2(64, 128);
    else ISTNET_WGRAD2(64, 64);
#undef ISTNET_WGRAD2
    return (int)hipGetLastError();
  }
  const int mt = wgrad_mt(cout, total), nt = wgrad_nt(cin, total);
  const dim3 grid(fixed_splits > 0 ? fixed_splits : wgrad_splits(b, cin, cout, p), ceil_div(cout, mt), ceil_div(cin, nt));
#define ISTNET_WGRAD(MT, NT)                                                                                  \
  do {                                                                                                        \
    if (gather)                                                                                               \
      hipLaunchKernelGGL((pw_wgrad_kernel<MT, NT, 2, 2, true>), grid, dim3(kThreads), 0, as_stream(stream),   \
                         cin, cout, p, total, len, x, g, in_scale, in_shift, y, gs, bn, bwdc, dw_part, ncols, colw); \
    else                                                                                                      \
      hipLaunchKernelGGL((pw_wgrad_kernel<MT, NT, 2, 2, false>), grid, dim3(kThreads), 0, as_stream(stream),  \
                         cin, cout, p, total, len, x, g, in_scale, in_shift, y, gs, bn, bwdc, dw_part, ncols, colw); \
  } while (0)
  if (mt == 128 && nt == 128) ISTNET_WGRAD(128, 128);
  else if (mt == 128) ISTNET_WGRAD(128, 64);
  else if (nt == 128) ISTNET_WGRAD(64, 128);
  else ISTNET_WGRAD(64, 64);
#undef ISTNET_WGRAD
  return (int)hipGetLastError();
}

int istnet_pw_bwd_small_ok(int cin, int cout, int p) {
  return cin > 0 && cout > 0 && cin <= 32 && cout <= 32 && p % kKTW == 0;
}

// the fused kernel has more per-workgroup set-up (weight fragments, two cross-wave reductions) than the plain
// wgrad: ~256 workgroups (16+ chunks per wave) measured 15-25 % faster than 1024 (tools/bench_bwd_small.py)
static int bwd_small_len(int b, int p) {
  const long long total = (long long)b * p;
  long long len = (total + g_bwd_small_target - 1) / g_bwd_small_target;
  len = (len + 4 * kKTW - 1) / (4 * kKTW) * (4 * kKTW);   // whole rounds of the four waves
  return (int)len;
}
int istnet_pw_bwd_small_splits(int b, int p) {
  const long long total = (long long)b * p;
  return (int)((total + bwd_small_len(b, p) - 1) / bwd_small_len(b, p));
}

int istnet_pw_bwd_small(int b, int cin, int cout, int p, int nsample, const float* w, const float* x,
                        const float* bn_in, const float* y, const float* d_dense, const float* d_pooled,
                        long long pooled_bstride, const unsigned char* arg, const float* bn, const float* bwdc,
                        float* dx, float* part_g, float* part_gy, float* dw_part, void* stream) {
  const int GS_C = cout;
  if (b <= 0 || !istnet_pw_bwd_small_ok(cin, cout, p) || (long long)b * p >= (1LL << 31)) return ISTNET_PN2_EINVAL;
  if (w == nullptr || x == nullptr || bn_in == nullptr || dx == nullptr || part_g == nullptr || part_gy == nullptr ||
      dw_part == nullptr)
    return ISTNET_PN2_EINVAL;
  if (d_dense == nullptr && (d_pooled == nullptr || arg == nullptr || nsample <= 0 || (nsample & 3)))
    return ISTNET_PN2_EINVAL;
  GradSrc gs{d_dense, d_pooled, arg, nsample, pooled_bstride > 0 ? pooled_bstride : (long long)GS_C * (nsample > 0 ? p / nsample : 0), GS_C};
  const int len = bwd_small_len(b, p);
  const int splits = istnet_pw_bwd_small_splits(b, p);   // also the number of statistics partials per channel
  if (d_dense != nullptr)
    hipLaunchKernelGGL(pw_bwd_small_kernel<false>, dim3(splits), dim3(kThreads), 0, as_stream(stream), cin, cout, p,
                       (long long)b * p, len, w, x, bn_in, bn_in + cin, y, gs, bn, bwdc, dx, part_g, part_gy, splits,
                       dw_part, nullptr, nullptr);
  else
    hipLaunchKernelGGL(pw_bwd_small_kernel<true>, dim3(splits), dim3(kThreads), 0, as_stream(stream), cin, cout, p,
                       (long long)b * p, len, w, x, bn_in, bn_in + cin, y, gs, bn, bwdc, dx, part_g, part_gy, splits,
                       dw_part, nullptr, nullptr);
  return (int)hipGetLastError();
}

// ---- fused backward of a mid-size layer (pw_bwd_mid_kernel) ----
int istnet_pw_bwd_mid_ok(int cin, int cout, int p) {
  if (!g_bwd_mid_enable || p <= 0 || (p % 128)) return 0;
  return (cout == 64 && (cin == 32 || cin == 64)) || (cout == 128 && (cin == 64 || cin == 128));
}
static int bwd_mid_pt(int cin) { return 128 / (cin / 32); }
static int bwd_mid_len(int b, int cin, int p) {
  const long long total = (long long)b * p;
  const int pt = bwd_mid_pt(cin);
  long long len = (total + g_bwd_mid_target - 1) / g_bwd_mid_target;
  len = (len + pt - 1) / pt * pt;
  if (len < 2 * pt) len = 2 * pt;
  return (int)len;
}
int istnet_pw_bwd_mid_splits(int b, int cin, int cout, int p) {
  if (!istnet_pw_bwd_mid_ok(cin, cout, p) || b <= 0) return 0;
  const long long total = (long long)b * p;
  const int len = bwd_mid_len(b, cin, p);
  return (int)((total + len - 1) / len);
}

int istnet_pw_bwd_mid(int b, int cin, int cout, int p, int nsample, const float* w, const float* x,
                      const float* bn_in, const float* y, const float* d_dense, const float* d_pooled,
                      long long pooled_bstride, const unsigned char* arg, const float* bn, const float* bwdc,
                      float* dx, float* part_g, float* part_gy, float* dw_part, void* stream) {
  if (b <= 0 || !istnet_pw_bwd_mid_ok(cin, cout, p) || (long long)b * p >= (1LL << 31)) return ISTNET_PN2_EINVAL;
  if (!w || !x || !bn_in || !y || !bn || !bwdc || !dx || !part_g || !part_gy || !dw_part) return ISTNET_PN2_EINVAL;
  if (d_dense == nullptr && (d_pooled == nullptr || arg == nullptr || nsample <= 0 || (nsample & 3)))
    return ISTNET_PN2_EINVAL;
  GradSrc gs{d_dense, d_pooled, arg, nsample, pooled_bstride > 0 ? pooled_bstride : (long long)cout * (nsample > 0 ? p / nsample : 0), cout};
  const int len = bwd_mid_len(b, cin, p);
  const int splits = istnet_pw_bwd_mid_splits(b, cin, cout, p);
#define ISTNET_BWD_MID(COT, CIT)                                                                                   \
  do {                                                                                                             \
    constexpr size_t lds = MidCfg<COT, CIT>::LDS_BYTES;                                                            \
    static PerDeviceOnce attr_once; unsigned long long attr_bit;   /* more than 64 KB of LDS per workgroup needs the opt-in, once per kernel */   \
    if (attr_once.pending(attr_bit)) {                                                                                               \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_bwd_mid_kernel<COT, CIT, false>),                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||              \
          hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_bwd_mid_kernel<COT, CIT, true>),                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                \
        return ISTNET_PN2_EINVAL;                                                                                  \
      attr_once.done(attr_bit);                                                                                             \
    }                                                                                                              \
    if (d_dense != nullptr)                                                                                        \
      hipLaunchKernelGGL((pw_bwd_mid_kernel<COT, CIT, false>), dim3(splits), dim3(kMidThreads), lds,               \
                         as_stream(stream), p, (long long)b * p, len, w, x, bn_in, bn_in + cin, y, gs, bn, bwdc,   \
                         dx, part_g, part_gy, splits, dw_part);                                                    \
    else                                                                                                           \
      hipLaunchKernelGGL((pw_bwd_mid_kernel<COT, CIT, true>), dim3(splits), dim3(kMidThreads), lds,                \
                         as_stream(stream), p, (long long)b * p, len, w, x, bn_in, bn_in + cin, y, gs, bn, bwdc,   \
                         dx, part_g, part_gy, splits, dw_part);                                                    \
  } while (0)
  if (cout == 64 && cin == 32) ISTNET_BWD_MID(2, 1);
  else if (cout == 64) ISTNET_BWD_MID(2, 2);
  else if (cin == 64) ISTNET_BWD_MID(4, 2);
  else ISTNET_BWD_MID(4, 4);
#undef ISTNET_BWD_MID
  return (int)hipGetLastError();
}

// compact-column form (csrc/sa_compact.hip): one point axis of capacity `cap`, *ncols valid columns, column
// multiplicities colw in the weight gradient and the statistics; dense gradient source only
int istnet_pw_bwd_small_cols_splits(void) { return g_bwd_small_target; }

int istnet_pw_bwd_small_cols(int cin, int cout, long long cap, const float* w, const float* x, const float* bn_in,
                             const float* y, const float* d_dense, const float* bn, const float* bwdc, float* dx,
                             float* part_g, float* part_gy, float* dw_part, const int* ncols, const float* colw,
                             void* stream) {
  if (cap <= 0 || (cap & 255) || cap >= (1LL << 31) || !istnet_pw_bwd_small_ok(cin, cout, (int)cap)) return ISTNET_PN2_EINVAL;
  if (!w || !x || !bn_in || !y || !d_dense || !bn || !bwdc || !dx || !part_g || !part_gy || !dw_part || !ncols || !colw)
    return ISTNET_PN2_EINVAL;
  GradSrc gs{d_dense, nullptr, nullptr, 0, 0, cout};
  const int splits = g_bwd_small_target;
  hipLaunchKernelGGL(pw_bwd_small_kernel<false>, dim3(splits), dim3(kThreads), 0, as_stream(stream), cin, cout,
                     (int)cap, cap, 0, w, x, bn_in, bn_in + cin, y, gs, bn, bwdc, dx, part_g, part_gy, splits, dw_part,
                     ncols, colw);
  return (int)hipGetLastError();
}

int istnet_pw_wgrad(int b, int cin, int cout, int p, int nsample, const float* x, const float* in_scale,
                    const float* in_shift, const float* y, const float* d_dense, const float* d_pooled,
                    long long pooled_bstride, const unsigned char* arg, const float* bn, const float* bwdc,
                    float* dw_part, void* stream) {
  return launch_pw_wgrad(false, b, cin, cout, p, nsample, x, GatherSrc{}, in_scale, in_shift, y, d_dense,
                         d_pooled, pooled_bstride, arg, bn, bwdc, dw_part, stream);
}

// compact-column form: splits = istnet_pw_wgrad_cols_splits(cin, cout) workgroup columns share the valid range evenly
int istnet_pw_wgrad_cols_splits(int cin, int cout) {
  if (wgrad_small(cin, cout)) return 0;     // those layers run the fused small-layer backward
  const long long tiles = (long long)ceil_div(cout, 64) * ceil_div(cin, 64);
  const long long want = (g_wg_target_small + tiles - 1) / tiles;
  return (int)(want < 1 ? 1 : want);
}

int istnet_pw_wgrad_cols(int cin, int cout, long long cap, const float* x, const float* in_scale,
                         const float* in_shift, const float* y, const float* d_dense, const float* bn,
                         const float* bwdc, float* dw_part, const int* ncols, const float* colw, void* stream) {
  if (cap <= 0 || (cap & 255) || cap >= (1LL << 31) || !d_dense || !ncols || !colw) return ISTNET_PN2_EINVAL;
  return launch_pw_wgrad(false, 1, cin, cout, (int)cap, 0, x, GatherSrc{}, in_scale, in_shift, y, d_dense, nullptr, 0,
                         nullptr, bn, bwdc, dw_part, stream, ncols, colw, istnet_pw_wgrad_cols_splits(cin, cout));
}

int istnet_pw_wgrad_gather(int b, int n, int npoint, int nsample, int cfeat, int cout, int grad_nsample,
                           const float* xyz, const float* new_xyz, const float* feat,
                           const int* idx, const float* y, const float* d_dense, const float* d_pooled,
                           long long pooled_bstride, const unsigned char* arg, const float* bn,
                           const float* bwdc, float* dw_part, void* stream) {
  if (n <= 0 || npoint <= 0 || nsample <= 0 || (nsample & 3) || cfeat < 0) return ISTNET_PN2_EINVAL;
  const GatherSrc g{xyz, new_xyz, feat, idx, n, nsample, cfeat};
  return launch_pw_wgrad(true, b, 3 + cfeat, cout, npoint * nsample, grad_nsample, nullptr, g, nullptr, nullptr, y,
                         d_dense, d_pooled, pooled_bstride, arg, bn, bwdc, dw_part, stream);
}

static int reduce_multi_impl(int n, const int* counts, const int* splits, const float* const* parts, float* const* dws,
                             const int* cols, const int* lds, const long long* pstrides, void* stream) {
  if (n <= 0 || n > 8) return ISTNET_PN2_EINVAL;
  ReduceBatch rb;
  rb.n = n;
  rb.block_begin[0] = 0;
  for (int l = 0; l < n; ++l) {
    if (counts[l] <= 0 || splits[l] <= 0) return ISTNET_PN2_EINVAL;
    rb.part[l] = parts[l];
    rb.dw[l] = dws[l];
    rb.count[l] = counts[l];
    rb.splits[l] = splits[l];
    rb.cols[l] = cols != nullptr ? cols[l] : counts[l];
    rb.ld[l] = lds != nullptr ? lds[l] : counts[l];
    rb.pstride[l] = pstrides != nullptr ? pstrides[l] : (long long)counts[l];
    if (rb.cols[l] <= 0 || counts[l] % rb.cols[l] || rb.ld[l] < rb.cols[l] || rb.pstride[l] < counts[l]) return ISTNET_PN2_EINVAL;
    rb.groups[l] = reduce_groups(splits[l]);
    rb.block_begin[l + 1] = rb.block_begin[l] + ceil_div(counts[l], 1024 / rb.groups[l]);
  }
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(rb.block_begin[n]), dim3(1024), 0, as_stream(stream), rb);
  return (int)hipGetLastError();
}

int istnet_pw_wgrad_reduce_multi(int n, const int* counts, const int* splits, const float* const* parts,
                                 float* const* dws, void* stream) {
  return reduce_multi_impl(n, counts, splits, parts, dws, nullptr, nullptr, nullptr, stream);
}

int istnet_pw_wgrad_reduce_multi_ld(int n, const int* counts, const int* splits, const float* const* parts,
                                    float* const* dws, const int* cols, const int* lds, const long long* pstrides,
                                    void* stream) {
  if (cols == nullptr || lds == nullptr || pstrides == nullptr) return ISTNET_PN2_EINVAL;
  return reduce_multi_impl(n, counts, splits, parts, dws, cols, lds, pstrides, stream);
}

int istnet_pack_words(int n, const void* const* srcs, const long long* words, void* dst, void* stream) {
  if (n <= 0 || n > 64 || dst == nullptr) return ISTNET_PN2_EINVAL;
  PackBatch pb;
  pb.n = n;
  pb.begin[0] = 0;
  for (int l = 0; l < n; ++l) {
    if (words[l] < 0 || (words[l] > 0 && srcs[l] == nullptr)) return ISTNET_PN2_EINVAL;
    pb.src[l] = static_cast<const unsigned*>(srcs[l]);
    pb.begin[l + 1] = pb.begin[l] + words[l];
  }
  for (int l = n; l < 64; ++l) { pb.src[l] = nullptr; pb.begin[l + 1] = pb.begin[n]; }
  const long long total = pb.begin[n];
  if (total == 0) return 0;
  long long blocks = (total + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_words_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), pb,
                     static_cast<unsigned*>(dst));
  return (int)hipGetLastError();
}

int istnet_pw_wgrad_reduce(int count, int splits, const float* dw_part, float* dw, void* stream) {
  if (count <= 0 || splits <= 0) return ISTNET_PN2_EINVAL;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(count, 16)), dim3(256), 0, as_stream(stream), count,
                     splits, dw_part, dw);
  return (int)hipGetLastError();
}

}  // extern "C"
