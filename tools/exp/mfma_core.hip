// Experiment: ceiling of the 4-wave / 2x2-tile f32 MFMA main loop used by pw_mlp.hip, without global traffic.
//   variant 0: MFMA loop only (fragments re-read from LDS every k-step, no barrier)
//   variant 1: + __syncthreads per chunk
//   variant 2: + 8 ds_write_b32 + 2 ds_write_b128 per thread per chunk (the staging writes) + barrier
//   variant 3: + ~60 VALU of fake transform per chunk
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_core.hip -o /tmp/mfma_core && /tmp/mfma_core
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KT = 16, M_T = 128, N_T = 128;

template <int VARIANT, int TM, int TN>
__global__ __launch_bounds__(256) void core(int nchunks, float* out, const float* in) {
  __shared__ __attribute__((aligned(16))) float As[2][KT][M_T + 1];
  __shared__ __attribute__((aligned(16))) float Bs[2][KT][N_T];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 2 * KT * (M_T + 1); i += 256) (&As[0][0][0])[i] = in[i % 1024];
  for (int i = tid; i < 2 * KT * N_T; i += 256) (&Bs[0][0][0])[i] = in[(i * 7) % 1024];
  __syncthreads();
  f32x16 acc[TM][TN];
  for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int a_col0 = (w / 2) * TM * 32, b_col0 = (w % 2) * TN * 32;
  float regs[8]; float4 r4[2];
  for (int i = 0; i < 8; ++i) regs[i] = in[tid + i];
  r4[0] = make_float4(regs[0], regs[1], regs[2], regs[3]); r4[1] = r4[0];
  for (int t = 0; t < nchunks; ++t) {
    const int buf = t & 1;
    const float* ap = &As[buf][0][0] + (lane >> 5) * (M_T + 1) + a_col0 + (lane & 31);
    const float* bp = &Bs[buf][0][0] + (lane >> 5) * N_T + b_col0 + (lane & 31);
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
        for (int tm = 0; tm < TM; ++tm) a[nxt][tm] = ap[(2 * kk + 2) * (M_T + 1) + tm * 32];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[nxt][tn] = bp[(2 * kk + 2) * N_T + tn * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tm], b[cur][tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (VARIANT >= 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { float v = regs[i]; for (int q = 0; q < 8; ++q) v = fmaxf(v * 1.0001f + 0.5f, 0.f); regs[i] = v; }
    }
    if (VARIANT >= 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int e = tid + 256 * i; As[buf ^ 1][e % KT][e / KT] = regs[i]; }
#pragma unroll
      for (int i = 0; i < 2; ++i) { const int e = tid + 256 * i; *reinterpret_cast<float4*>(&Bs[buf ^ 1][e / 32][(e % 32) * 4]) = r4[i]; }
    }
    if (VARIANT >= 1) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int V, int TM, int TN>
void run(const char* name, int wgs, int nchunks, float* out, float* in) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((core<V, TM, TN>), dim3(wgs), dim3(256), 0, 0, nchunks, out, in);
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((core<V, TM, TN>), dim3(wgs), dim3(256), 0, 0, nchunks, out, in);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double flops = 2.0 * 32 * 32 * 2 * TM * TN * 4 /*waves*/ * (KT / 2) * (double)nchunks * wgs;
  printf("%-40s wgs %5d chunks %4d: %8.1f us  %6.1f TFLOP/s\n", name, wgs, nchunks, ms * 1e3, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float *out, *in; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 8192 * 4); hipMemset(in, 0, 8192 * 4);
  for (int wgs : {256, 512, 1024}) {
    run<0, 2, 2>("core only 2x2", wgs, 64, out, in);
    run<1, 2, 2>("core + barrier 2x2", wgs, 64, out, in);
    run<2, 2, 2>("core + barrier + ds_write 2x2", wgs, 64, out, in);
    run<3, 2, 2>("core + barrier + ds_write + valu 2x2", wgs, 64, out, in);
    run<2, 2, 2>("short K (8 chunks) + ds_write 2x2", wgs, 8, out, in);
    run<2, 1, 2>("core + barrier + ds_write 1x2", wgs, 64, out, in);
  }
  return 0;
}
