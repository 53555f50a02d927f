/*
 * s2c_oracle.c -- CPU restatement of the nine native point-cloud ops of
 * daveredrum/Scan2Cap (lib/pointnet2/_ext_src).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP
 * kernels in scan2cap_amd/csrc/.  It may be imported / linked / executed only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
 * path (scan2cap_amd/) never calls it and fails loudly when the HIP library is
 * missing.
 *
 * Pinning status: the reference ships no CPU implementation and no golden
 * vectors for these ops (every op asserts "CPU not supported",
 * lib/pointnet2/_ext_src/src/ball_query.cpp:27-29 etc.); the only reference
 * test on this path is the three_interpolate KAT of
 * lib/pointnet2/pointnet2_test.py:18-30, which tests/test_oracle_kat.py checks.
 * The other ops are pinned by hand-derived known-answer tests plus an
 * independent numpy restatement (tests/test_oracle_kat.py) and by running the
 * reference's own Python layers on top of this oracle (tests/golden/).
 *
 * Arithmetic convention (DESIGN.md "canonical arithmetic"): IEEE-754 binary32,
 * operations in the order the reference source writes them, NO fused
 * multiply-add contraction.  Build with -ffp-contract=off (oracle/Makefile).
 * s2c_oracle_set_contract(1 | 2) switches the three-term sums of squares of the
 * index-producing ops to the two fused-multiply-add contractions an nvcc build of
 * the reference (--fmad=true, its default) may use -- the same switch as
 * -DS2C_NVCC_CONTRACT in scan2cap_amd/csrc/s2c_common.h; 0 restores the canonical form.
 *
 * Each function cites the reference file:line it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define S2C_TOTAL_THREADS 512 /* cuda_utils.h:11 */

/* a*a + b*b + c*c (sampling_gpu.cu:100-104, ball_query_gpu.cu:31-32,
 * interpolate_gpu.cu:36-37): canonical, or one of the two nvcc-style contractions. */
static int g_contract = 0;
void s2c_oracle_set_contract(int mode) { g_contract = (mode == 1 || mode == 2) ? mode : 0; }
int s2c_oracle_get_contract(void) { return g_contract; }
static inline float sq3(float a, float b, float c) {
  if (g_contract == 1) return fmaf(c, c, fmaf(a, a, b * b));
  if (g_contract == 2) return fmaf(c, c, fmaf(b, b, a * a));
  return (a * a + b * b) + c * c;
}

/* cuda_utils.h:13-19 -- block size chosen by the reference host code.
 * Restated with the same double-precision log expression so the
 * power-of-two rounding quirks (if any) are identical. */
int s2c_oracle_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > S2C_TOTAL_THREADS) t = S2C_TOTAL_THREADS;
  if (t < 1) t = 1;
  return t;
}

/* sampling_gpu.cu:59-65 (__update) and :69-173 (kernel), sampling.cpp:66-87
 * (temp initialised to 1e10, idx zero-initialised).
 * xyz (b,n,3) f32, temp (b,n) f32 scratch (overwritten), idx (b,m) i32. */
void s2c_oracle_furthest_point_sampling(int b, int n, int m, const float *xyz,
                                        float *temp, int *idx) {
  if (m <= 0) return; /* sampling_gpu.cu:73 */
  const int bs = s2c_oracle_opt_n_threads(n); /* sampling_gpu.cu:178 */
#pragma omp parallel for schedule(dynamic, 1)
  for (int bi = 0; bi < b; ++bi) {
    float *dists = (float *)malloc(sizeof(float) * (size_t)bs);
    int *dists_i = (int *)malloc(sizeof(int) * (size_t)bs);
    const float *ds = xyz + (size_t)bi * n * 3;
    float *tp = temp + (size_t)bi * n;
    int *out = idx + (size_t)bi * m;
    for (int k = 0; k < n; ++k) tp[k] = 1e10f; /* sampling.cpp:74-76 */
    for (int j = 0; j < m; ++j) out[j] = 0;     /* sampling.cpp:70-72 */
    int old = 0;
    out[0] = old; /* sampling_gpu.cu:85-86 */
    for (int j = 1; j < m; ++j) {
      /* per-virtual-thread running best, sampling_gpu.cu:90-91 */
      for (int t = 0; t < bs; ++t) {
        dists[t] = -1.0f;
        dists_i[t] = 0;
      }
      const float x1 = ds[old * 3 + 0];
      const float y1 = ds[old * 3 + 1];
      const float z1 = ds[old * 3 + 2];
      /* thread t visits k = t, t+bs, ... in ascending order; iterating k
       * ascending visits every thread's points in that same order. */
      for (int k = 0; k < n; ++k) {
        const int t = k % bs;
        const float x2 = ds[k * 3 + 0];
        const float y2 = ds[k * 3 + 1];
        const float z2 = ds[k * 3 + 2];
        const float mag = sq3(x2, y2, z2);
        /* sampling_gpu.cu:101 compares the float against the double literal
         * 1e-3: the comparison is done in double. */
        if ((double)mag <= 1e-3) continue;
        const float d = sq3(x2 - x1, y2 - y1, z2 - z1);
        const float d2 = d < tp[k] ? d : tp[k]; /* min(d, temp[k]) :106 */
        tp[k] = d2;
        if (d2 > dists[t]) { /* strict, :108-109 */
          dists_i[t] = k;
          dists[t] = d2;
        }
      }
      /* tree reduction, sampling_gpu.cu:115-168 with __update :59-65 */
      for (int h = bs / 2; h >= 1; h /= 2) {
        for (int t = 0; t < h; ++t) {
          const float v1 = dists[t], v2 = dists[t + h];
          const int i1 = dists_i[t], i2 = dists_i[t + h];
          dists[t] = v1 > v2 ? v1 : v2;
          dists_i[t] = v2 > v1 ? i2 : i1;
        }
      }
      old = dists_i[0];
      out[j] = old;
    }
    free(dists);
    free(dists_i);
  }
}

/* sampling_gpu.cu:8-20.  points (b,c,n), idx (b,m) -> out (b,c,m) */
void s2c_oracle_gather_points(int b, int c, int n, int m, const float *points,
                              const int *idx, float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[i * m + j];
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
      }
}

/* sampling_gpu.cu:34-47 (atomicAdd scatter; summation order is unspecified in
 * the reference, here ascending j).  grad_out (b,c,m) -> grad_points (b,c,n),
 * zero-initialised as in sampling.cpp:52-54. */
void s2c_oracle_gather_points_grad(int b, int c, int n, int m,
                                   const float *grad_out, const int *idx,
                                   float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[i * m + j];
        grad_points[((size_t)i * c + l) * n + a] +=
            grad_out[((size_t)i * c + l) * m + j];
      }
}

/* ball_query_gpu.cu:9-44; idx zero-initialised (ball_query.cpp:19-21).
 * new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample) */
void s2c_oracle_ball_query(int b, int n, int m, float radius, int nsample,
                           const float *new_xyz, const float *xyz, int *idx) {
  memset(idx, 0, sizeof(int) * (size_t)b * m * nsample);
  const float radius2 = radius * radius; /* :22 */
  for (int bi = 0; bi < b; ++bi) {
    const float *p = xyz + (size_t)bi * n * 3;
    const float *q = new_xyz + (size_t)bi * m * 3;
    int *o = idx + (size_t)bi * m * nsample;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < m; ++j) {
      const float new_x = q[j * 3 + 0];
      const float new_y = q[j * 3 + 1];
      const float new_z = q[j * 3 + 2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float x = p[k * 3 + 0];
        const float y = p[k * 3 + 1];
        const float z = p[k * 3 + 2];
        const float d2 = sq3(new_x - x, new_y - y, new_z - z);
        if (d2 < radius2) { /* strict, :33 */
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) o[j * nsample + l] = k; /* :34-38 */
          o[j * nsample + cnt] = k;
          ++cnt;
        }
      }
    }
  }
}

/* group_points_gpu.cu:8-28.  points (b,c,n), idx (b,npoints,nsample) ->
 * out (b,c,npoints,nsample) */
void s2c_oracle_group_points(int b, int c, int n, int npoints, int nsample,
                             const float *points, const int *idx, float *out) {
  for (int bi = 0; bi < b; ++bi) {
    const float *p = points + (size_t)bi * n * c;
    const int *ix = idx + (size_t)bi * npoints * nsample;
    float *o = out + (size_t)bi * npoints * nsample * c;
#pragma omp parallel for schedule(static)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) {
          const int ii = ix[j * nsample + k];
          o[((size_t)l * npoints + j) * nsample + k] = p[(size_t)l * n + ii];
        }
  }
}

/* group_points_gpu.cu:43-64 (atomicAdd; here ascending (j,k)).
 * grad_out (b,c,npoints,nsample) -> grad_points (b,c,n) zero-initialised
 * (group_points.cpp:50-52). */
void s2c_oracle_group_points_grad(int b, int c, int n, int npoints,
                                  int nsample, const float *grad_out,
                                  const int *idx, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int bi = 0; bi < b; ++bi) {
    const float *g = grad_out + (size_t)bi * npoints * nsample * c;
    const int *ix = idx + (size_t)bi * npoints * nsample;
    float *gp = grad_points + (size_t)bi * n * c;
#pragma omp parallel for schedule(static)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) {
          const int ii = ix[j * nsample + k];
          gp[(size_t)l * n + ii] += g[((size_t)l * npoints + j) * nsample + k];
        }
  }
}

/* interpolate_gpu.cu:9-59.  unknown (b,n,3), known (b,m,3) ->
 * dist2 (b,n,3) f32, idx (b,n,3) i32.  best* are doubles initialised to 1e40
 * and compared against the float d (:27-49). */
void s2c_oracle_three_nn(int b, int n, int m, const float *unknown,
                         const float *known, float *dist2, int *idx) {
  for (int bi = 0; bi < b; ++bi) {
    const float *u = unknown + (size_t)bi * n * 3;
    const float *kn = known + (size_t)bi * m * 3;
    float *d2o = dist2 + (size_t)bi * n * 3;
    int *io = idx + (size_t)bi * n * 3;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; ++j) {
      const float ux = u[j * 3 + 0];
      const float uy = u[j * 3 + 1];
      const float uz = u[j * 3 + 2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float x = kn[k * 3 + 0];
        const float y = kn[k * 3 + 1];
        const float z = kn[k * 3 + 2];
        const float d = sq3(ux - x, uy - y, uz - z);
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
      d2o[j * 3 + 0] = (float)best1; /* double -> float store, :52-54 */
      d2o[j * 3 + 1] = (float)best2;
      d2o[j * 3 + 2] = (float)best3;
      io[j * 3 + 0] = besti1;
      io[j * 3 + 1] = besti2;
      io[j * 3 + 2] = besti3;
    }
  }
}

/* interpolate_gpu.cu:72-101.  points (b,c,m), idx (b,n,3), weight (b,n,3) ->
 * out (b,c,n); sum order p1*w1 + p2*w2 + p3*w3 (:98-99). */
void s2c_oracle_three_interpolate(int b, int c, int m, int n,
                                  const float *points, const int *idx,
                                  const float *weight, float *out) {
  for (int bi = 0; bi < b; ++bi) {
    const float *p = points + (size_t)bi * m * c;
    const int *ix = idx + (size_t)bi * n * 3;
    const float *w = weight + (size_t)bi * n * 3;
    float *o = out + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float w1 = w[j * 3 + 0], w2 = w[j * 3 + 1], w3 = w[j * 3 + 2];
        const int i1 = ix[j * 3 + 0], i2 = ix[j * 3 + 1], i3 = ix[j * 3 + 2];
        o[(size_t)l * n + j] = p[(size_t)l * m + i1] * w1 +
                               p[(size_t)l * m + i2] * w2 +
                               p[(size_t)l * m + i3] * w3;
      }
  }
}

/* interpolate_gpu.cu:116-143 (three atomicAdds per element; here ascending
 * (l,j)).  grad_out (b,c,n) -> grad_points (b,c,m) zero-initialised
 * (interpolate.cpp:83-85). */
void s2c_oracle_three_interpolate_grad(int b, int c, int n, int m,
                                       const float *grad_out, const int *idx,
                                       const float *weight,
                                       float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * m);
  for (int bi = 0; bi < b; ++bi) {
    const float *g = grad_out + (size_t)bi * n * c;
    const int *ix = idx + (size_t)bi * n * 3;
    const float *w = weight + (size_t)bi * n * 3;
    float *gp = grad_points + (size_t)bi * m * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float w1 = w[j * 3 + 0], w2 = w[j * 3 + 1], w3 = w[j * 3 + 2];
        const int i1 = ix[j * 3 + 0], i2 = ix[j * 3 + 1], i3 = ix[j * 3 + 2];
        const float go = g[(size_t)l * n + j];
        gp[(size_t)l * m + i1] += go * w1;
        gp[(size_t)l * m + i2] += go * w2;
        gp[(size_t)l * m + i3] += go * w3;
      }
  }
}
