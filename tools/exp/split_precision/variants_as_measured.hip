// NOT BUILT -- the split-precision section of csrc/conv_nhwc.hip as it stood when profiles/r05_split_precision_conv.txt and
// r05_split_precision_diagnostics.txt were measured: the kernel with its INTERLEAVE (sched_group_barrier) and DIAG (timing
// experiments) template parameters, and the "W" variant (weights pre-split in global memory in fragment order,
// conv_igemm_splitw_kernel + split_weights_kernel).  Measured negatives: INTERLEAVE +-0; four waves per workgroup -15 %;
// four register stages +-0; W variant -12 % (605 vs 521 us on layer4 3x3, B = 32) although it has a third of the LDS traffic
// and half of the VALU work.  The library keeps the two variants that won (K chunks of 16 / of 32, eight waves).
// ============================================================================================
// Split-precision forward (opt-in experiment, istnet_conv_set_tuning(1, 1); DESIGN.md "split precision").
// Every fp32 operand is split EXACTLY into three bf16 terms x = hi + mid + lo (each takes the top 8 significand bits of what
// is left: truncation, so the remainders are exact fp32 subtractions), and a.b is evaluated as the six products
//   a_lo b_hi + a_hi b_lo + a_mid b_mid + a_mid b_hi + a_hi b_mid + a_hi b_hi
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Every bf16 x bf16 product is exact in fp32; the three dropped terms
// (mid lo, lo mid, lo lo) are <= 2^-24 |a b| each -- the size of ONE fp32 rounding of the product.  The bf16 matrix pipe runs
// 16x the fp32 one, so six products cost 6/16 of v_mfma_f32_32x32x2_f32: the roof moves from 157 to ~417 TFLOP/s.
// Operands are split once, when a chunk goes from registers to LDS (three bf16 planes per operand tile, rows K-contiguous,
// 80-byte pitch); a lane's MFMA fragment is one ds_read_b128 per plane.  128 x NT tiles, one workgroup per CU (120 KB of LDS).
// ============================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDKH = KC + 8;          // bf16 elements per LDS row (80 bytes: 16-byte aligned, rows spread over the banks)

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

template <int NT, int STRIDE, int WM, bool INTERLEAVE, int DIAG = 0, int KCS = KC, int MINW = 1>      // WM = 4: eight waves of 32 x NT/2;  WM = 2: four waves of 64 x NT/2
// DIAG (timing experiments, wrong results): 1 no MFMAs, 2 no split / LDS writes, 3 no LDS reads, 4 no global loads
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
    if (DIAG == 4) {
      const float f = (float)(tap + cb);
      st.a0 = st.a1 = st.a2 = st.a3 = st.b0 = st.b1 = st.b2 = st.b3 = make_float4(f, f + 1.f, f + 2.f, f + 3.f);
      return;
    }
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
    if (DIAG == 2) {        // keep the loads alive without the split and the LDS writes
      if (st.a0.x == 123.456f && st.b0.x == 654.321f) as[tid] = 1;
      return;
    }
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
        if (DIAG == 3) { ah[mi] = am[mi] = al[mi] = bf16x8{}; ah[mi][0] = (__bf16)(float)lane; continue; }
        ah[mi] = *reinterpret_cast<const bf16x8*>(ap);
        am[mi] = *reinterpret_cast<const bf16x8*>(ap + A_PLANE);
        al[mi] = *reinterpret_cast<const bf16x8*>(ap + 2 * A_PLANE);
      }
      bf16x8 bh[TN], bm[TN], bl[TN];
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const unsigned short* bp = bs + ni * 32 * LDKH_ + 16 * g16;
        if (DIAG == 3) { bh[ni] = bm[ni] = bl[ni] = bf16x8{}; bh[ni][0] = (__bf16)(float)lane; continue; }
        bh[ni] = *reinterpret_cast<const bf16x8*>(bp);
        bm[ni] = *reinterpret_cast<const bf16x8*>(bp + B_PLANE);
        bl[ni] = *reinterpret_cast<const bf16x8*>(bp + 2 * B_PLANE);
      }
      // products outer, accumulators inner: consecutive MFMAs never share an accumulator (a dependent bf16 MFMA waits for
      // its predecessor's result; with TM * TN >= 2 tiles per wave the chain of one tile hides behind the other's)
#define ISTNET_P(A, B)                                                                         \
  _Pragma("unroll") for (int mi = 0; mi < TM; ++mi)                                             \
    _Pragma("unroll") for (int ni = 0; ni < TN; ++ni)                                           \
      if (DIAG != 1) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mi], B[ni], acc[mi][ni], 0, 0, 0); \
      else acc[mi][ni][0] += (float)A[mi][0] + (float)B[ni][0];
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
  // mma(buf) and commit(next chunk -> buf ^ 1) are INDEPENDENT (the other buffer was last read before the previous barrier),
  // and the waves of a workgroup are all in the same phase: as two blocks, the matrix pipe idles while every wave splits and
  // the VALU idles while every wave multiplies.  INTERLEAVE puts them in one scheduling region and asks (group barriers) for
  // one MFMA, then a few VALU / LDS instructions, and so on.
  Stage sa, sb;
  issue(sa, ky, kx, cb); advance();
  commit(sa, 0);
  issue(sa, ky, kx, cb); advance();       // chunk 1
  __syncthreads();
  auto body = [&](Stage& snew, Stage& scommit, int buf) {
    issue(snew, ky, kx, cb); advance();
    __builtin_amdgcn_sched_barrier(0);
    mma(buf);
    if (!INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
    commit(scommit, buf ^ 1);
    if (INTERLEAVE) {
#pragma unroll
      for (int q = 0; q < 12 * TM * TN; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // VALU
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
      }
      __builtin_amdgcn_sched_barrier(0);
    }
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

// ---- split precision, weights pre-split in global memory ("W" variant) ----
// The first split kernel is bound by LDS traffic: three planes for BOTH operands are 208 KB of LDS reads + writes per 32-k
// chunk against 1 536 matrix-pipe cycles.  The weights are the same for every pixel tile, so they are split ONCE per call by
// split_weights_kernel into three bf16 planes in global memory (layout of wgt: [cout][tap][cin], rows K-contiguous), and
// every wave loads its B fragments straight from there into registers in MFMA layout (one 16-byte load per plane, column and
// 16-k group; the tile's weights are L2 / L1 resident and shared by the four row-waves of the workgroup), one chunk ahead.
// Only the A operand (gathered pixels) goes through registers -> split -> LDS: a third of the LDS traffic, half of the VALU.
// Output layout = MFMA fragment order, so that a wave's fragment load is ONE contiguous kilobyte (in the weights' own
// [cout][tap][cin] order a lane's 16 bytes sit 2 * taps * cin bytes from its neighbour's: every load touched 32 cache lines
// for 32 useful bytes each and the kernel ran at half the speed of the LDS variant):
//   [column tile of 32][tap][16-k group][plane hi / mid / lo][lane = 32 (k % 16 / 8) + column % 32][8 bf16]
__global__ __launch_bounds__(kThreads) void split_weights_kernel(long long n4, int taps, int cin, const float4* __restrict__ w,
                                                                 unsigned short* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n4) return;
  const Split3 s = split3(w[i]);
  const long long e = 4 * i;                                  // element index in [cout][tap][cin]
  const int k = (int)(e % cin);
  const long long nt_ = e / cin;
  const int tap = (int)(nt_ % taps), n = (int)(nt_ / taps);
  const size_t frag = ((size_t)(n >> 5) * taps + tap) * (cin >> 4) + (k >> 4);
  unsigned short* p = dst + frag * (3 * 512) + (size_t)(((k & 15) >> 3) * 32 + (n & 31)) * 8 + (k & 7);
  *reinterpret_cast<uint2*>(p) = s.hi;
  *reinterpret_cast<uint2*>(p + 512) = s.mid;
  *reinterpret_cast<uint2*>(p + 1024) = s.lo;
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

struct BFrag { bf16x8 h, m, l; };

template <int NT, int STRIDE>
__global__ __launch_bounds__(512, 1) void conv_igemm_splitw_kernel(ConvGeom g, const float* __restrict__ a_src,
                                                                   const unsigned short* __restrict__ wsplit, long long,
                                                                   float* __restrict__ c_dst, float* __restrict__ ws,
                                                                   int chunks_per_split, int nsplits, int tile_m_first,
                                                                   int tile_m_count) {
  constexpr int MT = 128, WM = 4, WN = 2, NTHR = 512;
  constexpr int TN = NT / (32 * WN);
  constexpr int RPP = NTHR / LPR, AR = MT / RPP;        // 64 rows per pass, 2 float4 of A per thread
  constexpr int A_PLANE = MT * LDKH, STAGE = 3 * A_PLANE;
  constexpr int G16 = KC / 16;
  static_assert(AR == 2 && TN >= 1 && TN <= 2, "128 x 128 or 128 x 64 tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];       // [2][STAGE]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wv / WN, wn = wv % WN;
  (void)WM;
  const int taps = g.KH * g.KW;
  const int Ka = g.Cin, Ncols = g.Cout;
  const int MH = g.OH, MW = g.OW, SH = g.H, SW = g.W;
  const long long M = (long long)g.B * MH * MW;
  const int tiles_n = Ncols / NT;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int split = jj % nsplits, jt = jj / nsplits;
  const int tile_local = (jt / tiles_n) * 8 + xcd;
  if (tile_local >= tile_m_count) return;
  const long long m0 = (long long)(tile_m_first + tile_local) * MT;
  const int n0 = (jt % tiles_n) * NT;
  const int acol = (tid % LPR) * 4;
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
  // this wave's column tiles in the fragment-ordered planes (split_weights_kernel): fragment index of (tile, tap 0, k 0)
  size_t wrow[TN];
  const int kg_all = g.Cin >> 4;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) wrow[ni] = (size_t)((n0 >> 5) + wn * TN + ni) * taps * kg_all;
  struct AStage { float4 a0, a1; unsigned ok; };
  auto issue_a = [&](AStage& st, int cb) {
    st.ok = tap_ok;
    st.a0 = *reinterpret_cast<const float4*>(a_src + aoff[0] + (size_t)cb * KC);
    st.a1 = *reinterpret_cast<const float4*>(a_src + aoff[1] + (size_t)cb * KC);
  };
  auto issue_b = [&](BFrag (&bf)[TN][G16], int tap, int cb) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int q = 0; q < G16; ++q) {
        const unsigned short* p = wsplit + (wrow[ni] + (size_t)tap * kg_all + (size_t)cb * G16 + q) * (3 * 512) + lane * 8;
        bf[ni][q].h = *reinterpret_cast<const bf16x8*>(p);
        bf[ni][q].m = *reinterpret_cast<const bf16x8*>(p + 512);
        bf[ni][q].l = *reinterpret_cast<const bf16x8*>(p + 1024);
      }
  };
  auto put = [&](unsigned short* plane0, int row, const float4& v) {
    const Split3 s = split3(v);
    unsigned short* p = plane0 + row * LDKH + acol;
    *reinterpret_cast<uint2*>(p) = s.hi;
    *reinterpret_cast<uint2*>(p + A_PLANE) = s.mid;
    *reinterpret_cast<uint2*>(p + 2 * A_PLANE) = s.lo;
  };
  auto commit = [&](AStage& st, int buf) {
    unsigned short* as = smem_h + buf * STAGE;
    put(as, (tid / LPR), keep_if(st.ok & 1u, st.a0));
    put(as, (tid / LPR) + RPP, keep_if((st.ok >> 1) & 1u, st.a1));
  };
  f32x16 acc[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
  auto mma = [&](int buf, BFrag (&bf)[TN][G16]) {
    const unsigned short* as = smem_h + buf * STAGE + (wm * 32 + l31) * LDKH + 8 * half;
#pragma unroll
    for (int q = 0; q < G16; ++q) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(as + 16 * q);
      const bf16x8 am = *reinterpret_cast<const bf16x8*>(as + A_PLANE + 16 * q);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(as + 2 * A_PLANE + 16 * q);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bf[ni][q].h, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[ni][q].l, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bf[ni][q].m, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bf[ni][q].h, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[ni][q].m, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[ni][q].h, acc[ni], 0, 0, 0);
      }
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
  // A: two register stages (loads issued two MFMA phases before the split), LDS double buffered.  B: the fragments of the
  // chunk being multiplied and of the next one (loaded during this chunk's MFMAs).
  AStage sa, sb;
  BFrag b0[TN][G16], b1[TN][G16];
  issue_a(sa, cb); issue_b(b0, ky * g.KW + kx, cb); advance();
  commit(sa, 0);
  issue_a(sa, cb); issue_b(b1, ky * g.KW + kx, cb); advance();       // chunk 1
  __syncthreads();
  for (int c = 0; c < nchunks; c += 2) {
    issue_a(sb, cb);                      // A of chunk c + 2
    const int tap2 = ky * g.KW + kx, cb2 = cb;
    advance();
    __builtin_amdgcn_sched_barrier(0);
    mma(0, b0);                           // chunk c
    __builtin_amdgcn_sched_barrier(0);
    issue_b(b0, tap2, cb2);               // B of chunk c + 2 (b0 is free now)
    commit(sa, 1);                        // chunk c + 1
    __syncthreads();
    if (c + 1 >= nchunks) break;
    issue_a(sa, cb);                      // A of chunk c + 3
    const int tap3 = ky * g.KW + kx, cb3 = cb;
    advance();
    __builtin_amdgcn_sched_barrier(0);
    mma(1, b1);                           // chunk c + 1
    __builtin_amdgcn_sched_barrier(0);
    issue_b(b1, tap3, cb3);
    commit(sb, 0);                        // chunk c + 2
    __syncthreads();
  }
  const long long m_first = (long long)tile_m_first * MT;
  float* dst = nsplits == 1 ? c_dst : ws + ((size_t)split * (M - m_first) - m_first) * Ncols;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long long row = m0 + wm * 32 + mfma_row(r, lane);
    if (row < M) {
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) dst[(size_t)row * Ncols + n0 + (wn * TN + ni) * 32 + l31] = acc[ni][r];
    }
  }
}

