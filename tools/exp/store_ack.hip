// How long a workgroup waits for its stores to be acknowledged (what s_endpgm implicitly waits for) on MI355X, by pattern:
//   full:    each thread writes PER float4 (contiguous per workgroup: 16 KB at PER = 4)
//   scatter: 64 lanes of the workgroup write one float each with a large stride (the per-tile statistics partials)
//   both
// after an optional read phase of RD KB per workgroup (streams in flight when the stores start).
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/store_ack.hip -o tools/exp/store_ack
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k(int mode, int per, int rd4, const float4* __restrict__ src, float4* __restrict__ dst,
                                         float* __restrict__ part, int stride, unsigned long long* __restrict__ cyc) {
  const int g = blockIdx.x, t = threadIdx.x;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 0; i < rd4; ++i) {
    const float4 v = src[((size_t)g * rd4 + i) * 256 + t];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  const long long t0 = clock64();
  if (mode & 1)
    for (int i = 0; i < per; ++i) dst[((size_t)g * per + i) * 256 + t] = acc;
  if ((mode & 2) && t < 64) part[(size_t)t * stride + g] = acc.x;
  __builtin_amdgcn_s_waitcnt(0);
  const long long t1 = clock64();
  if (t == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
}

int main() {
  const size_t cap = (size_t)2048 * 64 * 256;      // float4 elements
  float4 *src, *dst; float* part; unsigned long long* cyc;
  hipMalloc(&src, cap * sizeof(float4)); hipMalloc(&dst, cap * sizeof(float4));
  hipMalloc(&part, (size_t)64 * 4096 * sizeof(float)); hipMalloc(&cyc, 8);
  hipMemset(src, 0, cap * sizeof(float4));
  printf("   G  mode     read KB/wg  store KB/wg   cycles waited per workgroup for the stores' acknowledgement\n");
  for (int rd : {0, 16, 64})
    for (int mode : {1, 2, 3})
      for (int G : {64, 256, 512, 1024, 2048}) {
        const int per = 4, rd4 = rd * 1024 / (256 * 16);
        unsigned long long best = ~0ull;
        for (int rep = 0; rep < 5; ++rep) {
          hipMemset(cyc, 0, 8);
          hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, mode, per, rd4, src, dst, part, 4096, cyc);
          hipDeviceSynchronize();
          unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
          if (c < best) best = c;
        }
        printf("%4d  %-8s %6d      %6d       %8.0f\n", G, mode == 1 ? "full" : (mode == 2 ? "scatter" : "both"), rd,
               (mode & 1) ? 16 : 0, (double)best / G);
      }
  return 0;
}
