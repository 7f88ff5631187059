// Store acknowledgement wait of a GEMM-like output tile on MI355X: a workgroup writes 32 rows x 128 floats (512 B per row) of
// a (rows x pitch) float matrix, rows spaced by `pitch` floats -- pitch a power of two (the activations' point count) vs
// padded.  grid = (P / 128) x (rows / 32) workgroups, as pw_fwd_sk_kernel launches.
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/store_tile.hip -o tools/exp/store_tile
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NT>
__global__ __launch_bounds__(256) void k(int P, int pitch, float* __restrict__ dst, unsigned long long* __restrict__ cyc) {
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, l31 = lane & 31, half = lane >> 5;
  const int p0 = blockIdx.x * 128, m0 = blockIdx.y * 32;
  __syncthreads();
  const long long t0 = clock64();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int row = m0 + 8 * wv + 2 * rr + half;               // each wave: 8 rows, two per store instruction
    float* q = dst + (size_t)row * pitch + p0 + 4 * l31;
    if (NT) {
      __builtin_nontemporal_store(1.f, q); __builtin_nontemporal_store(2.f, q + 1);
      __builtin_nontemporal_store(3.f, q + 2); __builtin_nontemporal_store((float)row, q + 3);
    } else {
      *reinterpret_cast<float4*>(q) = make_float4(1.f, 2.f, 3.f, (float)row);
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  const long long t1 = clock64();
  if (t == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
}

int main() {
  float* dst; unsigned long long* cyc;
  const size_t total = (size_t)3 << 30;                      // 3 GB: every launch below writes a region not touched for > 1 GB
  hipMalloc(&dst, total); hipMalloc(&cyc, 8);
  hipMemset(dst, 0, total);
  printf("  points/row   rows   workgroups   store kind     cold destination: cycles waited per workgroup (median of 7 regions)\n");
  const int cfg[][2] = {{4096, 256}, {16384, 128}, {32768, 128}, {16384, 512}};
  size_t off = 0;
  for (auto& c : cfg)
    for (int nt : {0, 1}) {
      const int P = c[0], rows = c[1], pitch = P;
      double v[7];
      for (int rep = 0; rep < 7; ++rep) {
        off = (off + ((size_t)64 << 20)) % (total / 4 - (size_t)rows * pitch - 1024);
        off &= ~(size_t)1023;
        hipMemset(cyc, 0, 8);
        if (nt) hipLaunchKernelGGL(k<1>, dim3(P / 128, rows / 32), dim3(256), 0, 0, P, pitch, dst + off, cyc);
        else hipLaunchKernelGGL(k<0>, dim3(P / 128, rows / 32), dim3(256), 0, 0, P, pitch, dst + off, cyc);
        hipDeviceSynchronize();
        unsigned long long x; hipMemcpy(&x, cyc, 8, hipMemcpyDeviceToHost);
        v[rep] = (double)x / ((P / 128) * (rows / 32));
      }
      for (int a = 0; a < 7; ++a) for (int b = a + 1; b < 7; ++b) if (v[b] < v[a]) { double t = v[a]; v[a] = v[b]; v[b] = t; }
      printf("%10d %7d %10d   %-12s %16.0f   (min %.0f max %.0f)\n", P, rows, (P / 128) * (rows / 32), nt ? "nontemporal" : "plain", v[3], v[0], v[6]);
    }
  return 0;
}
