/*
 * pn2_oracle.c -- CPU restatement of the nine PointNet++ device kernels of the
 * reference (CVMI-Lab/IST-Net, model/pointnet2/_ext_src/src/, the four .cu files).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The
 * product path (ist-net_amd/) never links, imports or calls anything in here.
 *
 * PARITY STATUS: "parity unpinned" for kernel semantics.  The reference has no
 * CPU implementation of these ops (every host entry ends in
 * TORCH_CHECK(false, "CPU not supported"), e.g. ball_query.cpp:32-34) and nvcc is
 * not available, so the reference kernels cannot be executed here.  The only
 * reference-held known answer is pointnet2_test.py:25-30 (three_interpolate);
 * tests/test_oracle.py checks it.  Everything else follows the .cu sources
 * statement by statement, each function citing file:line.
 *
 * Arithmetic convention (shared with the HIP kernels): IEEE-754 binary32,
 * round-to-nearest-even, operations in SOURCE ORDER, NO fused multiply-add
 * contraction by default.  Build with -ffp-contract=off (see oracle/Makefile); the
 * two FMA-contracted variants a CUDA build may use instead are explicit fmaf calls
 * selected with oracle_set_convention() (see g_conv below).
 *
 * All functions are batch-parallel with OpenMP when built with -fopenmp; the
 * per-cloud arithmetic is unchanged by that.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* Distance convention of the index-deciding squared distances (FPS, ball query, three_nn).
 *   0 (default)  ((dx*dx + dy*dy) + dz*dz), every operation rounded: the convention the product and the
 *                golden vectors use.
 *   1            fmaf(dz, dz, fmaf(dx, dx, dy*dy))   } the two ways a contracting compiler (nvcc's default
 *   2            fmaf(dz, dz, fmaf(dy, dy, dx*dx))   } -fmad=true) can fuse the reference expression
 * 1 and 2 exist to MEASURE how many index decisions depend on the choice (tools/fma_flip_table.py,
 * profiles/r02_fma_convention_flips.txt) and to keep the HIP kernels checked under each of them. */
static int g_conv = 0;
ORACLE_API int oracle_set_convention(int c) {
  if (c < 0 || c > 2) return -1;
  g_conv = c;
  return 0;
}
ORACLE_API int oracle_get_convention(void) { return g_conv; }
static inline float sqdist_d(float dx, float dy, float dz) {
  if (g_conv == 1) return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
  if (g_conv == 2) return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  return (dx * dx + dy * dy) + dz * dz;
}

/* OpenMP team size of the batch-parallel loops (cpu_baseline thread sweep); 0 = runtime default */
ORACLE_API void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* include/cuda_utils.h:18-22  opt_n_threads: clamp(2^floor(log2 w), 1, 512) */
ORACLE_API int oracle_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

/* ------------------------------------------------------------------------- */
/* sampling_gpu.cu:13-25  gather_points_kernel: out[b,c,j] = points[b,c,idx[b,j]] */
ORACLE_API void oracle_gather_points(int b, int c, int n, int m,
                                     const float *points, const int *idx,
                                     float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        int a = idx[(size_t)i * m + j];
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
      }
}

/* sampling_gpu.cu:39-52  gather_points_grad_kernel: scatter-add into zeros.
 * The reference uses fp32 atomicAdd (order undefined); the oracle adds in
 * ascending j.  grad_points must be zero-filled by the caller (sampling.cpp:56-58). */
ORACLE_API void oracle_gather_points_grad(int b, int c, int n, int m,
                                          const float *grad_out, const int *idx,
                                          float *grad_points) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        int a = idx[(size_t)i * m + j];
        grad_points[((size_t)i * c + l) * n + a] +=
            grad_out[((size_t)i * c + l) * m + j];
      }
}

/* ------------------------------------------------------------------------- */
/* sampling_gpu.cu:75-178  furthest_point_sampling_kernel<block_size>.
 * Simulates the thread block literally: block_size = opt_n_threads(n) "threads",
 * each scanning k = tid, tid+bs, ... with strict '>' (:113-114), then the
 * shared-memory tree of __update steps (:64-70, :120-173) where a tie keeps
 * the LOWER slot.  temp must be pre-filled with 1e10 (sampling.cpp:78-80). */
ORACLE_API void oracle_furthest_point_sampling(int b, int n, int m,
                                               const float *dataset_all,
                                               float *temp_all, int *idxs_all) {
  if (m <= 0) return;
  const int bs = oracle_opt_n_threads(n);
#pragma omp parallel for schedule(dynamic, 1)
  for (int bi = 0; bi < b; ++bi) {
    const float *dataset = dataset_all + (size_t)bi * n * 3;
    float *temp = temp_all + (size_t)bi * n;
    int *idxs = idxs_all + (size_t)bi * m;
    float *dists = (float *)malloc(sizeof(float) * bs);
    int *dists_i = (int *)malloc(sizeof(int) * bs);
    int old = 0;
    idxs[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = dataset[old * 3 + 0];
      const float y1 = dataset[old * 3 + 1];
      const float z1 = dataset[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) {
        int besti = 0;
        float best = -1.0f;
        for (int k = tid; k < n; k += bs) {
          const float x2 = dataset[k * 3 + 0];
          const float y2 = dataset[k * 3 + 1];
          const float z2 = dataset[k * 3 + 2];
          /* :108-109, source order; contraction per g_conv (default none) */
          const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
          const float d = sqdist_d(dx, dy, dz);
          const float d2 = d < temp[k] ? d : temp[k]; /* min(d, temp[k]) :111 */
          temp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int off = bs / 2; off >= 1; off /= 2) { /* :120-173 */
        for (int tid = 0; tid < off; ++tid) {
          const float v1 = dists[tid], v2 = dists[tid + off];
          const int i1 = dists_i[tid], i2 = dists_i[tid + off];
          dists[tid] = v1 > v2 ? v1 : v2; /* max(v1, v2) */
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      }
      old = dists_i[0];
      idxs[j] = old;
    }
    free(dists);
    free(dists_i);
  }
}

/* ------------------------------------------------------------------------- */
/* ball_query_gpu.cu:14-49  query_ball_point_kernel.  idx must be zero-filled by
 * the caller (ball_query.cpp:24-26): a centroid with no hit keeps its row of 0. */
ORACLE_API void oracle_query_ball_point(int b, int n, int m, float radius,
                                        int nsample, const float *new_xyz_all,
                                        const float *xyz_all, int *idx_all) {
  const float radius2 = radius * radius; /* :27 */
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < m; ++j) {
      const float *xyz = xyz_all + (size_t)bi * n * 3;
      const float *new_xyz = new_xyz_all + (size_t)bi * m * 3;
      int *idx = idx_all + (size_t)bi * m * nsample;
      const float new_x = new_xyz[j * 3 + 0];
      const float new_y = new_xyz[j * 3 + 1];
      const float new_z = new_xyz[j * 3 + 2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float x = xyz[k * 3 + 0];
        const float y = xyz[k * 3 + 1];
        const float z = xyz[k * 3 + 2];
        const float dx = new_x - x, dy = new_y - y, dz = new_z - z;
        const float d2 = sqdist_d(dx, dy, dz);          /* :36-37 */
        if (d2 < radius2) {                             /* :38 strict */
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) idx[j * nsample + l] = k;
          idx[j * nsample + cnt] = k;
          ++cnt;
        }
      }
    }
}

/* ------------------------------------------------------------------------- */
/* group_points_gpu.cu:13-33  out[b,c,j,k] = points[b,c,idx[b,j,k]] */
ORACLE_API void oracle_group_points(int b, int c, int n, int npoints,
                                    int nsample, const float *points,
                                    const int *idx, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *p = points + ((size_t)bi * c + l) * n;
      const int *ix = idx + (size_t)bi * npoints * nsample;
      float *o = out + ((size_t)bi * c + l) * npoints * nsample;
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          o[j * nsample + k] = p[ix[j * nsample + k]];
    }
}

/* group_points_gpu.cu:48-69  atomicAdd scatter into zeros (group_points.cpp:52-54);
 * oracle adds in ascending (j,k). */
ORACLE_API void oracle_group_points_grad(int b, int c, int n, int npoints,
                                         int nsample, const float *grad_out,
                                         const int *idx, float *grad_points) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      float *gp = grad_points + ((size_t)bi * c + l) * n;
      const int *ix = idx + (size_t)bi * npoints * nsample;
      const float *go = grad_out + ((size_t)bi * c + l) * npoints * nsample;
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          gp[ix[j * nsample + k]] += go[j * nsample + k];
    }
}

/* ------------------------------------------------------------------------- */
/* interpolate_gpu.cu:14-64  three_nn_kernel.  bests are doubles initialised to
 * 1e40 (:32), compared against the f32 distance with strict '<' (:39-53), stored
 * back as f32 (:55-57): an unfilled slot becomes +inf with index 0. */
ORACLE_API void oracle_three_nn(int b, int n, int m, const float *unknown_all,
                                const float *known_all, float *dist2_all,
                                int *idx_all) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < n; ++j) {
      const float *unknown = unknown_all + (size_t)bi * n * 3;
      const float *known = known_all + (size_t)bi * m * 3;
      float *dist2 = dist2_all + (size_t)bi * n * 3;
      int *idx = idx_all + (size_t)bi * n * 3;
      const float ux = unknown[j * 3 + 0];
      const float uy = unknown[j * 3 + 1];
      const float uz = unknown[j * 3 + 2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float x = known[k * 3 + 0];
        const float y = known[k * 3 + 1];
        const float z = known[k * 3 + 2];
        const float dx = ux - x, dy = uy - y, dz = uz - z;
        const float d = sqdist_d(dx, dy, dz); /* :38 */
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d;     besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d;     besti2 = k;
        } else if (d < best3) {
          best3 = d;     besti3 = k;
        }
      }
      dist2[j * 3 + 0] = (float)best1;
      dist2[j * 3 + 1] = (float)best2;
      dist2[j * 3 + 2] = (float)best3;
      idx[j * 3 + 0] = besti1;
      idx[j * 3 + 1] = besti2;
      idx[j * 3 + 2] = besti3;
    }
}

/* interpolate_gpu.cu:77-106  out[b,c,j] = p[i1]*w1 + p[i2]*w2 + p[i3]*w3 (:103-104) */
ORACLE_API void oracle_three_interpolate(int b, int c, int m, int n,
                                         const float *points, const int *idx,
                                         const float *weight, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *p = points + ((size_t)bi * c + l) * m;
      const int *ix = idx + (size_t)bi * n * 3;
      const float *w = weight + (size_t)bi * n * 3;
      float *o = out + ((size_t)bi * c + l) * n;
      for (int j = 0; j < n; ++j) {
        const float t1 = p[ix[j * 3 + 0]] * w[j * 3 + 0];
        const float t2 = p[ix[j * 3 + 1]] * w[j * 3 + 1];
        const float t3 = p[ix[j * 3 + 2]] * w[j * 3 + 2];
        o[j] = (t1 + t2) + t3;
      }
    }
}

/* interpolate_gpu.cu:121-148  three atomicAdds of grad_out*w_t into zeros
 * (interpolate.cpp:90-92); oracle adds in ascending j, taps 1,2,3. */
ORACLE_API void oracle_three_interpolate_grad(int b, int c, int n, int m,
                                              const float *grad_out,
                                              const int *idx,
                                              const float *weight,
                                              float *grad_points) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      float *gp = grad_points + ((size_t)bi * c + l) * m;
      const int *ix = idx + (size_t)bi * n * 3;
      const float *w = weight + (size_t)bi * n * 3;
      const float *go = grad_out + ((size_t)bi * c + l) * n;
      for (int j = 0; j < n; ++j) {
        gp[ix[j * 3 + 0]] += go[j] * w[j * 3 + 0];
        gp[ix[j * 3 + 1]] += go[j] * w[j * 3 + 1];
        gp[ix[j * 3 + 2]] += go[j] * w[j * 3 + 2];
      }
    }
}
