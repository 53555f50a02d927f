/*
 * s2c_ops.h -- C ABI of libs2c_hip.so, the MI355X (gfx950) implementation of the
 * Scan2Cap point-cloud hot path.
 *
 * Drop-in boundary: these entry points replace, one for one, the
 * `*_kernel_wrapper` functions that the reference's C++ op layer forward-declares
 * and calls with raw device pointers (the .cpp files of lib/pointnet2/_ext_src/src), i.e.
 * exactly what a binding for the `pointnet2._ext` module
 * (lib/pointnet2/_ext_src/src/bindings.cpp:6-19) links against.  Same argument
 * order and meaning as the reference wrappers, plus a trailing stream.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to dense row-major buffers
 *     (the reference requires contiguous tensors, include/utils.h:10-13);
 *   - floats are IEEE binary32, indices are int32 (include/utils.h:15-25);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); work is
 *     enqueued asynchronously, nothing synchronises (the reference launches on
 *     at::cuda::getCurrentCUDAStream(), e.g. ball_query_gpu.cu:49);
 *   - inputs are borrowed and never written; outputs are fully overwritten (the
 *     caller does NOT need to zero them, unlike the reference which relies on
 *     torch::zeros, e.g. ball_query.cpp:19-21) -- except that *_grad outputs are
 *     zeroed by the callee before accumulation;
 *   - return value: 0 on success; a negative S2C_E* code for invalid arguments;
 *     a positive hipError_t if the launch failed.  Nothing ever calls exit()
 *     (the reference does: include/cuda_utils.h:30-39).  s2c_last_error_string()
 *     describes the last failure on the calling thread.
 *
 * Index results are bit-exact with the reference algorithm under the canonical
 * arithmetic documented in DESIGN.md (IEEE binary32, source order, no FMA
 * contraction).
 */
#ifndef S2C_OPS_H
#define S2C_OPS_H

#ifdef __cplusplus
extern "C" {
#endif

#define S2C_ABI_VERSION 1

#define S2C_EINVAL (-1)  /* bad size / null pointer */
#define S2C_ENOSUP (-2)  /* configuration not supported by this build */

typedef void *s2c_stream_t; /* hipStream_t */

int s2c_abi_version(void);
const char *s2c_last_error_string(void);

/* replaces furthest_point_sampling_kernel_wrapper (sampling.cpp:11-13,
 * sampling_gpu.cu:175-229).  xyz (b,n,3) -> idx (b,m).
 * `temp` is the reference's (b,n) f32 scratch; this implementation keeps the
 * running min-distances in registers and ignores it (may be NULL) unless
 * n exceeds the register-resident limit reported by
 * s2c_fps_resident_limit(), in which case it must be a (b,n) f32 buffer. */
int s2c_furthest_point_sampling(int b, int n, int m, const float *xyz,
                                float *temp, int *idx, s2c_stream_t stream);
int s2c_fps_resident_limit(void);

/* Spatially-bucketed exact FPS (csrc/s2c_fps_bucket.hip): identical output to
 * s2c_furthest_point_sampling for every input, for large point sets.  Needs a
 * caller-owned, 16-byte aligned scratch of s2c_fps_workspace_bytes(b, n) bytes
 * (sorted points + tie-break ranks); nothing is allocated inside the library. */
long long s2c_fps_workspace_bytes(int b, int n);
int s2c_furthest_point_sampling_bucketed(int b, int n, int m, const float *xyz,
                                         void *workspace, int *idx,
                                         s2c_stream_t stream);

/* The same with every grid cell owned by one wave and a single workgroup barrier per round
 * (csrc/s2c_fps_cells.hip); identical output.  Scratch of s2c_fps_cells_workspace_bytes(b, n)
 * bytes; waves = workgroup size of the rounds kernel in waves (4, 8, 16; 0 = default). */
long long s2c_fps_cells_workspace_bytes(int b, int n);
int s2c_furthest_point_sampling_cells(int b, int n, int m, const float *xyz, void *workspace,
                                      int *idx, int waves, s2c_stream_t stream);

/* Latency-optimised register-resident FPS for n <= s2c_fps_small_limit() points
 * (csrc/s2c_fps_small.hip); identical output.  threads = 0 picks the geometry. */
int s2c_fps_small_limit(void);
int s2c_furthest_point_sampling_small(int b, int n, int m, const float *xyz, int *idx,
                                      int threads, s2c_stream_t stream);

/* FPS of a point set that is EXPECTED to be in FPS pick order already (SA2..SA4 sample the
 * previous stage's centres, backbone_module.py:106-115, so their picks are 0..m-1 unless a
 * tie or a skipped point breaks the property): the expectation is PROVEN per scene on the
 * device under the reference's exact rule (n*m parallel pair tests, no serial rounds); scenes
 * that fail it run the real rounds.  Identical output to s2c_furthest_point_sampling for every
 * input.  Scratch of s2c_fps_prefix_workspace_bytes(b, m) bytes; after the call its first b
 * ints hold 1 for the scenes that ran the rounds, 0 for the proven ones. */
long long s2c_fps_prefix_workspace_bytes(int b, int m);
int s2c_furthest_point_sampling_prefix(int b, int n, int m, const float *xyz, void *workspace,
                                       int *idx, int threads, s2c_stream_t stream);

/* replaces gather_points_kernel_wrapper (sampling.cpp:5-7, sampling_gpu.cu:22-30).
 * points (b,c,n), idx (b,npoints) -> out (b,c,npoints) */
int s2c_gather_points(int b, int c, int n, int npoints, const float *points,
                      const int *idx, float *out, s2c_stream_t stream);

/* replaces gather_points_grad_kernel_wrapper (sampling.cpp:8-10,
 * sampling_gpu.cu:49-57).  grad_out (b,c,npoints) -> grad_points (b,c,n) */
int s2c_gather_points_grad(int b, int c, int n, int npoints,
                           const float *grad_out, const int *idx,
                           float *grad_points, s2c_stream_t stream);

/* replaces query_ball_point_kernel_wrapper (ball_query.cpp:4-6,
 * ball_query_gpu.cu:46-54).  new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample) */
int s2c_ball_query(int b, int n, int m, float radius, int nsample,
                   const float *new_xyz, const float *xyz, int *idx,
                   s2c_stream_t stream);

/* The same operator for large point sets (csrc/s2c_bq_grid.hip): identical output; the
 * points are counting-sorted into a uniform grid of edge >= radius in a caller-owned,
 * 16-byte aligned scratch of s2c_ball_query_workspace_bytes(b, n) bytes and a centre
 * visits the <= 27 cells its ball can touch.  nsample <=
 * s2c_ball_query_grid_max_nsample(); radius > 0. */
long long s2c_ball_query_workspace_bytes(int b, int n);
int s2c_ball_query_grid_max_nsample(void);
int s2c_ball_query_grid(int b, int n, int m, float radius, int nsample,
                        const float *new_xyz, const float *xyz, void *workspace,
                        int *idx, s2c_stream_t stream);

/* replaces group_points_kernel_wrapper (group_points.cpp:4-6,
 * group_points_gpu.cu:30-39).  points (b,c,n), idx (b,npoints,nsample) ->
 * out (b,c,npoints,nsample) */
int s2c_group_points(int b, int c, int n, int npoints, int nsample,
                     const float *points, const int *idx, float *out,
                     s2c_stream_t stream);

/* replaces group_points_grad_kernel_wrapper (group_points.cpp:8-10,
 * group_points_gpu.cu:66-75).  grad_out (b,c,npoints,nsample) ->
 * grad_points (b,c,n) */
int s2c_group_points_grad(int b, int c, int n, int npoints, int nsample,
                          const float *grad_out, const int *idx,
                          float *grad_points, s2c_stream_t stream);

/* replaces three_nn_kernel_wrapper (interpolate.cpp:4-5,
 * interpolate_gpu.cu:61-68).  unknown (b,n,3), known (b,m,3) ->
 * dist2 (b,n,3) squared distances, idx (b,n,3) */
int s2c_three_nn(int b, int n, int m, const float *unknown, const float *known,
                 float *dist2, int *idx, s2c_stream_t stream);

/* replaces three_interpolate_kernel_wrapper (interpolate.cpp:6-8,
 * interpolate_gpu.cu:103-111).  points (b,c,m), idx (b,n,3), weight (b,n,3) ->
 * out (b,c,n) */
int s2c_three_interpolate(int b, int c, int m, int n, const float *points,
                          const int *idx, const float *weight, float *out,
                          s2c_stream_t stream);

/* replaces three_interpolate_grad_kernel_wrapper (interpolate.cpp:9-12,
 * interpolate_gpu.cu:145-154).  grad_out (b,c,n) -> grad_points (b,c,m) */
int s2c_three_interpolate_grad(int b, int c, int n, int m,
                               const float *grad_out, const int *idx,
                               const float *weight, float *grad_points,
                               s2c_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* S2C_OPS_H */
