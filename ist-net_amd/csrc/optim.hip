// Flat-buffer Adam (include/istnet_optim.h).  HBM-bound: 4 reads + 3 writes of 4 B per parameter.
#include <hip/hip_runtime.h>

#include "../../include/istnet_optim.h"

namespace {

constexpr int kThreads = 256;

struct AdamConsts {
  float b1, one_minus_b1, b2, one_minus_b2, eps, wd, gscale;
  double lr, beta1, beta2;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamConsts& c, float step_size,
                                         float inv_bc2_sqrt) {
  g = g * c.gscale;
  if (c.wd != 0.0f) g = g + c.wd * p;
  m = c.b1 * m + c.one_minus_b1 * g;
  v = c.b2 * v + c.one_minus_b2 * (g * g);
  const float denom = sqrtf(v) * inv_bc2_sqrt + c.eps;
  p = p - step_size * (m / denom);
}

// COUNTING (istnet_adam_step_counting): the kernel itself advances the step count -- every workgroup reads *step, uses
// t = *step + 1, and the LAST workgroup to finish (a ticket counter) stores t and resets the ticket: by then every workgroup has
// read the old value.  The framework's `step += 1` launch in front of the update -- one small kernel on the tail of every
// training step -- is gone.  The grid is capped (grid-stride loop) so that the tickets of one launch do not queue up on the
// counter (one word serves ~88 atomics / us).
template <bool COUNTING>
__global__ void adam_step_kernel(long long n, float* __restrict__ param, const float* __restrict__ grad,
                                 float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                 float* __restrict__ step, const float* __restrict__ lr_dev,
                                 AdamConsts c, unsigned* __restrict__ ticket) {
  // bias corrections in double, once per thread (the step count is the same for every element)
  const float step_next = *step + (COUNTING ? 1.0f : 0.0f);
  const double t = (double)step_next;
  const double lr = lr_dev != nullptr ? (double)*lr_dev : c.lr;
  const float step_size = (float)(lr / (1.0 - pow(c.beta1, t)));
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow(c.beta2, t)));
  if (COUNTING) {
    const long long quads = n / 4;
    for (long long q = (long long)blockIdx.x * kThreads + threadIdx.x; q < quads; q += (long long)gridDim.x * kThreads) {
      const long long i = q * 4;
      float4 p = *reinterpret_cast<float4*>(param + i);
      const float4 g = *reinterpret_cast<const float4*>(grad + i);
      float4 m = *reinterpret_cast<float4*>(exp_avg + i);
      float4 v = *reinterpret_cast<float4*>(exp_avg_sq + i);
      adam_one(p.x, g.x, m.x, v.x, c, step_size, inv_bc2_sqrt);
      adam_one(p.y, g.y, m.y, v.y, c, step_size, inv_bc2_sqrt);
      adam_one(p.z, g.z, m.z, v.z, c, step_size, inv_bc2_sqrt);
      adam_one(p.w, g.w, m.w, v.w, c, step_size, inv_bc2_sqrt);
      *reinterpret_cast<float4*>(param + i) = p;
      *reinterpret_cast<float4*>(exp_avg + i) = m;
      *reinterpret_cast<float4*>(exp_avg_sq + i) = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
      for (long long j = quads * 4; j < n; ++j) {
        float p = param[j], m = exp_avg[j], v = exp_avg_sq[j];
        adam_one(p, grad[j], m, v, c, step_size, inv_bc2_sqrt);
        param[j] = p; exp_avg[j] = m; exp_avg_sq[j] = v;
      }
    __syncthreads();                         // every thread of this workgroup has read *step
    if (threadIdx.x == 0) {
      const unsigned done = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done == gridDim.x - 1) {           // all workgroups have taken a ticket, hence read the old count
        __hip_atomic_store(step, step_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  const long long i = ((long long)blockIdx.x * kThreads + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 p = *reinterpret_cast<float4*>(param + i);
    const float4 g = *reinterpret_cast<const float4*>(grad + i);
    float4 m = *reinterpret_cast<float4*>(exp_avg + i);
    float4 v = *reinterpret_cast<float4*>(exp_avg_sq + i);
    adam_one(p.x, g.x, m.x, v.x, c, step_size, inv_bc2_sqrt);
    adam_one(p.y, g.y, m.y, v.y, c, step_size, inv_bc2_sqrt);
    adam_one(p.z, g.z, m.z, v.z, c, step_size, inv_bc2_sqrt);
    adam_one(p.w, g.w, m.w, v.w, c, step_size, inv_bc2_sqrt);
    *reinterpret_cast<float4*>(param + i) = p;
    *reinterpret_cast<float4*>(exp_avg + i) = m;
    *reinterpret_cast<float4*>(exp_avg_sq + i) = v;
  } else {
    for (long long j = i; j < n; ++j) {
      float p = param[j], m = exp_avg[j], v = exp_avg_sq[j];
      adam_one(p, grad[j], m, v, c, step_size, inv_bc2_sqrt);
      param[j] = p;
      exp_avg[j] = m;
      exp_avg_sq[j] = v;
    }
  }
}

}  // namespace

extern "C" int istnet_adam_step(long long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                const float* step, const float* lr_dev, double lr, double beta1, double beta2,
                                double eps, double weight_decay, double grad_scale, void* stream) {
  if (n < 0 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq || !step))) return ISTNET_PN2_EINVAL;
  // float4 path needs 16-byte aligned bases; torch allocations are, sub-views at odd offsets are not
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return ISTNET_PN2_EINVAL;
  if (n == 0) return 0;
  AdamConsts c{(float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
               (float)weight_decay, (float)grad_scale, lr, beta1, beta2};
  const long long quads = (n + 3) / 4;
  const unsigned blocks = (unsigned)((quads + kThreads - 1) / kThreads);
  adam_step_kernel<false><<<blocks, kThreads, 0, (hipStream_t)stream>>>(n, param, grad, exp_avg, exp_avg_sq,
                                                                         const_cast<float*>(step), lr_dev, c, nullptr);
  return (int)hipGetLastError();
}

extern "C" int istnet_adam_step_counting(long long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                         float* step, unsigned* ticket, const float* lr_dev, double lr, double beta1,
                                         double beta2, double eps, double weight_decay, double grad_scale, void* stream) {
  if (n < 0 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq)) || !step || !ticket) return ISTNET_PN2_EINVAL;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return ISTNET_PN2_EINVAL;
  AdamConsts c{(float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
               (float)weight_decay, (float)grad_scale, lr, beta1, beta2};
  const long long quads = (n + 3) / 4;
  long long blocks = (quads + kThreads - 1) / kThreads;
  if (blocks > 512) blocks = 512;            // grid-stride beyond: few tickets per launch
  if (blocks < 1) blocks = 1;                // n == 0 still counts the step
  adam_step_kernel<true><<<(unsigned)blocks, kThreads, 0, (hipStream_t)stream>>>(n, param, grad, exp_avg, exp_avg_sq, step,
                                                                                 lr_dev, c, ticket);
  return (int)hipGetLastError();
}
