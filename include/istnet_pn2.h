/*
 * istnet_pn2.h -- C ABI of libistnet_pn2.so, the MI355X (gfx950) replacement for
 * the nine launch wrappers behind the reference's pybind module `pointnet2._ext`
 * (reference: model/pointnet2/_ext_src/src/bindings.cpp:11-24).
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; no torch types; the library owns no memory
 *     and keeps no state; re-entrant; every launch is asynchronous on `stream`
 *     (a hipStream_t passed as void*, NULL = the default stream), matching the
 *     reference's "launch on the caller's current stream, no sync"
 *     (e.g. ball_query_gpu.cu:54).
 *   - tensors are contiguous; floats are IEEE f32; indices are int32
 *     (reference include/utils.h:15-30).
 *   - return value: 0 on success, otherwise a hipError_t (launch failure) or
 *     ISTNET_PN2_EINVAL for an invalid argument.  Never exits the process
 *     (the reference fprintf+exit(-1)s, cuda_utils.h:35-44).
 *   - OUTPUT OWNERSHIP: unlike the reference host code, which allocates outputs
 *     with torch::zeros, every kernel here writes EVERY element of its output,
 *     so the caller may pass uninitialised memory.  The only exception is
 *     `temp` of furthest_point_sampling, which is pure scratch (never read
 *     before written) and may be NULL.
 *   - arithmetic that decides an index (squared distances) is IEEE f32 in the
 *     reference's source order without FMA contraction; see DESIGN.md.
 */
#ifndef ISTNET_PN2_H_
#define ISTNET_PN2_H_

#ifdef __cplusplus
extern "C" {
#endif

#define ISTNET_PN2_ABI_VERSION 2
#define ISTNET_PN2_API __attribute__((visibility("default")))
#define ISTNET_PN2_EINVAL 100001

/* ABI version of the loaded library (== ISTNET_PN2_ABI_VERSION). */
ISTNET_PN2_API int istnet_pn2_abi_version(void);
/* Process-wide knobs.  key 0 = minimum FPS slot count that uses the 4-wave kernel (1..1025, default 1025;
 * benchmarking only).  key 1 = convention of the index-deciding squared distances of FPS / ball query /
 * three_nn: 0 (default) un-contracted ((dx*dx + dy*dy) + dz*dz); 1 fma(dz,dz,fma(dx,dx,dy*dy)); 2
 * fma(dz,dz,fma(dy,dy,dx*dx)) -- the forms a reference build with nvcc's default -fmad=true may use
 * (profiles/r02_fma_convention_flips.txt).  key 2 = 1: inverse lists by the one-workgroup-per-cloud
 * kernels (A/B and tests).  Out-of-range values: ISTNET_PN2_EINVAL. */
ISTNET_PN2_API int istnet_pn2_set_tuning(int key, int value);
/* Debug aid: enqueue a one-thread kernel that stores the GPU's 100 MHz wall clock into *slot (device memory)
 * when `stream` reaches this point -- used to draw the timeline of a captured step (tools/step_timeline.py). */
ISTNET_PN2_API int istnet_debug_marker(unsigned long long *slot, void *stream);

/* Name of the code object's target, "gfx950". */
ISTNET_PN2_API const char *istnet_pn2_target(void);

/* replaces furthest_point_sampling_kernel_wrapper (sampling.cpp:16-18, sampling_gpu.cu:180-234)
 * dataset (b,n,3) f32 -> idxs (b,m) i32, idxs[:,0] = 0.  temp (b,n) f32 scratch or NULL.
 * Tie-break identical to the reference block tree with block = opt_n_threads(n). */
ISTNET_PN2_API int istnet_pn2_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                       float *temp, int *idxs, void *stream);
/* the same sampling, additionally writing the coordinates of the picks, picked (b,m,3) -- what the reference
 * composes as gather_operation(xyz^T, idxs)^T right after sampling (pointnet2_modules.py:49-58).  n <= 4096. */
ISTNET_PN2_API int istnet_pn2_fps_gather(int b, int n, int m, const float *dataset, int *idxs,
                                         float *picked, void *stream);

/* Chained sampling for a stack of set-abstraction levels (model/modules.py:311-320 applies
 * pointnet2_modules.py:49-58 level after level: level l+1 samples from level l's picks, in pick order, starting at
 * index 0).  Same results as istnet_pn2_fps_gather, bit for bit.  tie_out (b) i32: the first round of THIS run whose
 * maximum was attained by more than one point among rounds 1 .. min(m, track_rounds)-1, else INT_MAX (or track_rounds
 * when track_rounds < m: later rounds were not examined).  tie_in (b) i32 or NULL: the tie_out of the run whose picks
 * are this run's `dataset` rows; a cloud with tie_in >= m needs no scan -- its picks are provably 0 .. m-1 (the parent's
 * pick of round j maximised the same running minimum distance over a superset, uniquely) -- and gets tie_out = tie_in;
 * every other cloud runs the full scan.  n <= 4096. */
ISTNET_PN2_API int istnet_pn2_fps_gather_chain(int b, int n, int m, const float *dataset, int *idxs, float *picked,
                                               const int *tie_in, int *tie_out, int track_rounds, void *stream);

/* replaces gather_points_kernel_wrapper (sampling.cpp:9-11): points (b,c,n), idx (b,npoints) -> out (b,c,npoints) */
ISTNET_PN2_API int istnet_pn2_gather_points(int b, int c, int n, int npoints, const float *points,
                             const int *idx, float *out, void *stream);

/* replaces gather_points_grad_kernel_wrapper (sampling.cpp:12-14): grad_out (b,c,npoints) -> grad_points (b,c,n)
 * (fully written: zero where no index lands). */
ISTNET_PN2_API int istnet_pn2_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                  const int *idx, float *grad_points, void *stream);

/* replaces query_ball_point_kernel_wrapper (ball_query.cpp:9-11): new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample) */
ISTNET_PN2_API int istnet_pn2_query_ball_point(int b, int n, int m, float radius, int nsample,
                                const float *new_xyz, const float *xyz, int *idx, void *stream);

/* query_ball_point_kernel_wrapper (ball_query.cpp:9-11) for the TWO radii an MSG level asks for on the same centroids
 * (reference model/modules.py:249-297, pointnet2_modules.py:48-58: one QueryAndGroup per radius): one pass over the cloud,
 * idx_a (b,m,nsample_a) / idx_b (b,m,nsample_b) bit-identical to two istnet_pn2_query_ball_point calls.  glen_a / glen_b
 * (b*m each, or NULL): compact-column count of every row for istnet_sa_compact_pair (include/istnet_pw.h). */
ISTNET_PN2_API int istnet_pn2_query_ball_point_pair(int b, int n, int m, float radius_a, int nsample_a, float radius_b,
                                                    int nsample_b, const float *new_xyz, const float *xyz, int *idx_a,
                                                    int *idx_b, int *glen_a, int *glen_b, void *stream);

/* replaces group_points_kernel_wrapper (group_points.cpp:9-11): points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample) */
ISTNET_PN2_API int istnet_pn2_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                            const int *idx, float *out, void *stream);

/* replaces group_points_grad_kernel_wrapper (group_points.cpp:13-15): grad_out (b,c,npoints,nsample) -> grad_points (b,c,n) */
ISTNET_PN2_API int istnet_pn2_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                 const float *grad_out, const int *idx, float *grad_points,
                                 void *stream);

/* replaces three_nn_kernel_wrapper (interpolate.cpp:9-10): unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) f32, idx (b,n,3) i32 */
ISTNET_PN2_API int istnet_pn2_three_nn(int b, int n, int m, const float *unknown, const float *known,
                        float *dist2, int *idx, void *stream);

/* replaces three_interpolate_kernel_wrapper (interpolate.cpp:11-13): points (b,c,m), idx/weight (b,n,3) -> out (b,c,n) */
ISTNET_PN2_API int istnet_pn2_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                 const float *weight, float *out, void *stream);

/* replaces three_interpolate_grad_kernel_wrapper (interpolate.cpp:14-17): grad_out (b,c,n) -> grad_points (b,c,m) */
ISTNET_PN2_API int istnet_pn2_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                      const int *idx, const float *weight, float *grad_points,
                                      void *stream);

/* Deterministic, atomic-free form of three_interpolate_grad (no reference counterpart; the reference uses
 * three fp32 atomicAdds per element, interpolate_gpu.cu:144-146).
 * istnet_pn2_interp_csr_build: per cloud, groups the 3n taps e = 3*j + t by the known point idx[e] they read:
 *   offsets (b, m+1) i32, entries (b, 3n) i32 sorted ascending inside each group.  Needs 3m+257 ints of LDS;
 *   returns ISTNET_PN2_EINVAL when m is too large (use istnet_pn2_three_interpolate_grad then).
 * istnet_pn2_three_interpolate_grad_csr: grad_points[b][c][i] = sum_{e in group i} grad_out[b][c][e/3] * weight[b][e]
 *   in ascending e, i.e. the summation order of the serial loop. */
ISTNET_PN2_API int istnet_pn2_interp_csr_build(int b, int n, int m, const int *idx, int *offsets, int *entries,
                                               void *stream);
/* the same inverse lists for any index tensor: idx (b, e) with values in [0, m) -> offsets (b, m+1), entries (b, e)
 * (entries of a list in ascending slot order).  Used with the ball-query indices (e = npoint*nsample, m = n) by the
 * atomic-free layer-0 gradient scatter, istnet_pw_scatter_dy_csr. */
ISTNET_PN2_API int istnet_pn2_csr_build(int b, int e, int m, const int *idx, int *offsets, int *entries,
                                        void *stream);
/* nprob (<= 12) independent list builds over the same batch size in ONE launch (e.g. the ball-query and three_nn
 * index tensors of every level of an encoder pass): problem l has e[l] slots per cloud with keys in [0, m[l]).
 * A workgroup owns 64 consecutive keys of one cloud (grid = b * sum ceil(m[l] / 64)); no atomics; the lists are
 * identical to istnet_pn2_csr_build's.  ISTNET_PN2_EINVAL when a problem's slots do not fit the LDS queue
 * (e[l] > ~15000): build that one with istnet_pn2_csr_build. */
ISTNET_PN2_API int istnet_pn2_csr_build_multi(int nprob, int b, const int *e, const int *m, const int *const *idx,
                                              int *const *offsets, int *const *entries, void *stream);
/* the same over SEGMENTED slot arrays (compact columns, csrc/sa_compact.hip): problem l keeps the slots of all clouds
 * on one axis, cloud c owning idx[l][seg[l][c] .. seg[l][c+1]) (seg on the device, at most e[l] slots per cloud); its
 * keys are idx - c * key_sub[l]; entries are slot numbers relative to seg[l][c], written to entries[l] + seg[l][c];
 * offsets stay (b, m[l] + 1) per cloud. */
ISTNET_PN2_API int istnet_pn2_csr_build_segmented(int nprob, int b, const int *e, const int *m, const int *const *idx,
                                                  int *const *offsets, int *const *entries, const int *const *seg,
                                                  const int *key_sub, void *stream);
ISTNET_PN2_API int istnet_pn2_three_interpolate_grad_csr(int b, int c, int n, int m, const float *grad_out,
                                                         const float *weight, const int *offsets,
                                                         const int *entries, float *grad_points, void *stream);

/* three_nn plus the inverse-distance weights PointnetFPModule forms from it (pointnet2_modules.py:185-188 over
 * ThreeNN's sqrt, pointnet2_utils.py:140-149) in the same launch: idx (b, n, 3) i32 as istnet_pn2_three_nn,
 * weight (b, n, 3) = r_k / (r_0 + r_1 + r_2), r_k = 1 / (sqrt(dist2_k) + 1e-8).  Replaces five framework
 * elementwise / reduce launches per propagation level. */
ISTNET_PN2_API int istnet_pn2_three_nn_weights(int b, int n, int m, const float *unknown, const float *known,
                                               int *idx, float *weight, void *stream);
/* the same for up to 8 (n, m) problems over the same b clouds in ONE launch -- the four propagation levels of an encoder pass
 * (reference model/modules.py:322-325); arrays of nprob entries, every problem bit-identical to its stand-alone launch */
ISTNET_PN2_API int istnet_pn2_three_nn_weights_multi(int nprob, int b, const int *n, const int *m,
                                                     const float *const *unknown, const float *const *known,
                                                     int *const *idx, float *const *weight, void *stream);

/* Deterministic, atomic-free form of group_points_grad over the inverse lists of idx (istnet_pn2_csr_build with
 * e = npoints*nsample, m = n): grad_points[b][c][i] = sum over the slots that picked point i.  The reference uses one
 * fp32 atomicAdd per slot (group_points_gpu.cu:62-66).  Needs one gradient row (npoints*nsample floats) to fit 64 KB
 * of LDS; ISTNET_PN2_EINVAL otherwise (use istnet_pn2_group_points_grad then). */
ISTNET_PN2_API int istnet_pn2_group_points_grad_csr(int b, int c, int n, int npoints, int nsample,
                                                    const float *grad_out, const int *offsets, const int *entries,
                                                    float *grad_points, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ISTNET_PN2_H_ */
