// Cost of an in-kernel grid barrier on MI355X (8 XCDs, L2s not mutually coherent): G workgroups, R rounds of
//   write a slice -> release fence -> arrive (agent-scope atomic add) -> spin (agent-scope atomic load) -> acquire -> read
//   the slice of ANOTHER workgroup (other XCD) and check it.
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/grid_barrier.hip -o /tmp/grid_barrier ; run: /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void barrier_rounds(int rounds, int slice, float* data, unsigned* counter, int* errors) {
  const int g = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
  for (int r = 0; r < rounds; ++r) {
    float* mine = data + (size_t)g * slice;
    for (int i = tid; i < slice; i += 256) mine[i] = (float)(r * 1000 + g);
    __threadfence();                       // release: write back to memory visible to the other XCDs
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(r + 1) * G;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __threadfence();                       // acquire side for the whole workgroup
    const int other = (g + G / 2 + 1) % G;
    const float* theirs = data + (size_t)other * slice;
    int bad = 0;
    for (int i = tid; i < slice; i += 256) bad += theirs[i] != (float)(r * 1000 + other);
    if (bad) atomicAdd(errors, bad);
    // second barrier so nobody overwrites a slice that is still being read
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(counter + 32, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(r + 1) * G;
      while (__hip_atomic_load(counter + 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
}

__global__ void empty_kernel() {}

int main() {
  const int rounds = 50;
  for (int G : {64, 128, 256, 512}) {
    for (int slice : {256, 4096, 16384}) {
      float* data; unsigned* counter; int* errors;
      hipMalloc(&data, (size_t)G * slice * sizeof(float));
      hipMalloc(&counter, 64 * sizeof(unsigned));
      hipMalloc(&errors, sizeof(int));
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      float best = 1e9f;
      int err = 0;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemset(counter, 0, 64 * sizeof(unsigned)); hipMemset(errors, 0, sizeof(int));
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(barrier_rounds, dim3(G), dim3(256), 0, 0, rounds, slice, data, counter, errors);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        hipMemcpy(&err, errors, sizeof(int), hipMemcpyDeviceToHost);
      }
      printf("G=%3d slice=%6d floats: %.2f us per round (2 barriers + write + fence + read), errors %d\n", G, slice,
             best * 1e3f / rounds, err);
      hipFree(data); hipFree(counter); hipFree(errors);
    }
  }
  // for scale: a chain of dependent empty kernels
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("dependent empty kernels (eager, one stream): %.2f us each\n", ms * 1e3f / 200);
  return 0;
}
