// s2c_boxes.hip -- the non-differentiable box bookkeeping of the proposal and caption
// stages as two launches instead of ~45 framework micro-kernels (each a 3-6 us node of the
// captured step):
//   proposal_decode   argmax size class, box size, AABB corners (float64), objectness /
//                     semantic arg-max     (models/proposal_module.py:80-144,
//                     data/scannet/model_util_scannet.py:165-172, utils/box_util.py:360-383)
//   select_target     best-IoU proposal per ground-truth box (models/caption_module.py:16-38,
//                     utils/box_util.py:183-209)
// Arithmetic is the reference's: float32 residual * float32 mean size, float64 mean size +
// residual, corners = +-size/2 + centre (ScanNet heading is 0, so the rotation is the
// identity), AABB IoU in float64 in the reference's operation order; arg-max = first maximum.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

namespace {

__device__ __forceinline__ int argmax_first(const float *p, int n) {
  int best = 0;
  float bv = p[0];
  for (int i = 1; i < n; ++i) {
    const float v = p[i];
    if (v > bv || (v != v && bv == bv)) { bv = v; best = i; }   // NaN counts as maximal
  }
  return best;
}

// thread = proposal
__global__ __launch_bounds__(256) void proposal_decode_kernel(
    int total, int nout, int NH, int NS, int num_class, const float *__restrict__ net,
    const float *__restrict__ center, const float *__restrict__ mean32,
    const double *__restrict__ mean64, double *__restrict__ corners,
    long long *__restrict__ bbox_mask, long long *__restrict__ sem_cls,
    long long *__restrict__ size_class_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float *row = net + (size_t)i * nout;
  const int o_size = 5 + 2 * NH;
  const int cls = argmax_first(row + o_size, NS);
  const float *resn = row + o_size + NS + 3 * cls;       // normalised residual of the class
  double size[3], c[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float res = resn[d] * mean32[cls * 3 + d];      // data_dict["size_residuals"]
    size[d] = mean64[cls * 3 + d] + (double)res;          // class2size_batch
    c[d] = (double)center[(size_t)i * 3 + d];
  }
  const double sx[8] = {1, 1, -1, -1, 1, 1, -1, -1};
  const double sy[8] = {1, -1, -1, 1, 1, -1, -1, 1};
  const double sz[8] = {1, 1, 1, 1, -1, -1, -1, -1};
  double *out = corners + (size_t)i * 24;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    out[k * 3 + 0] = (size[0] / 2) * sx[k] + c[0];
    out[k * 3 + 1] = (size[1] / 2) * sy[k] + c[1];
    out[k * 3 + 2] = (size[2] / 2) * sz[k] + c[2];
  }
  bbox_mask[i] = argmax_first(row, 2);
  sem_cls[i] = argmax_first(row + o_size + 4 * NS, num_class);
  if (size_class_out) size_class_out[i] = cls;
}

// workgroup = sample; thread = proposal (strided when K > 1024)
__global__ __launch_bounds__(1024) void select_target_kernel(
    int K, const double *__restrict__ corners, const double *__restrict__ gt,
    long long *__restrict__ target_ids, float *__restrict__ target_ious) {
  __shared__ double s_iou[1024];
  __shared__ int s_idx[1024];
  const int b = blockIdx.x, t = threadIdx.x;
  const double *g = gt + (size_t)b * 24;
  double gmin[3], gmax[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    gmin[d] = gmax[d] = g[d];
    for (int k = 1; k < 8; ++k) {
      gmin[d] = fmin(gmin[d], g[k * 3 + d]);
      gmax[d] = fmax(gmax[d], g[k * 3 + d]);
    }
  }
  double best = -1.0;
  int bidx = 0x7fffffff;
  for (int k = t; k < K; k += 1024) {
    const double *p = corners + ((size_t)b * K + k) * 24;
    double inter = 1.0, vol1 = 1.0, vol2 = 1.0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      double lo1 = p[d], hi1 = p[d];
      for (int q = 1; q < 8; ++q) {
        lo1 = fmin(lo1, p[q * 3 + d]);
        hi1 = fmax(hi1, p[q * 3 + d]);
      }
      const double lo = fmax(lo1, gmin[d]), hi = fmin(hi1, gmax[d]);
      const double ext = fmax(hi - lo, 0.0);
      inter = (d == 0) ? ext : inter * ext;
      vol1 = (d == 0) ? (hi1 - lo1) : vol1 * (hi1 - lo1);
      vol2 = (d == 0) ? (gmax[d] - gmin[d]) : vol2 * (gmax[d] - gmin[d]);
    }
    const double iou = inter / (vol1 + vol2 - inter + 1e-8);
    if (iou > best) { best = iou; bidx = k; }        // first maximum of this thread's stride
  }
  s_iou[t] = best;
  s_idx[t] = bidx;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (t < s) {
      const double o = s_iou[t + s];
      const int oi = s_idx[t + s];
      if (o > s_iou[t] || (o == s_iou[t] && oi < s_idx[t])) { s_iou[t] = o; s_idx[t] = oi; }
    }
    __syncthreads();
  }
  if (t == 0) {
    target_ids[b] = s_idx[0];
    target_ious[b] = (float)s_iou[0];
  }
}

// good[b] = ious[b] > min_iou; mean = mean IoU over the good samples, 0 when there is none
// (caption_module.py:30-36 / :494-498: `pred_ious`, `good_bbox_masks`).  One wave; the sum
// runs in ascending sample order like the framework's reduction of <= 64 values.
__global__ __launch_bounds__(64) void good_bbox_stats_kernel(int B, const float *__restrict__ ious,
                                                             float min_iou, bool *__restrict__ good,
                                                             float *__restrict__ mean) {
  float sum = 0.f;
  int n = 0;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int b = b0 + (int)threadIdx.x;
    const float v = b < B ? ious[b] : 0.f;
    const bool g = b < B && v > min_iou;
    if (b < B) good[b] = g;
    const unsigned long long m = __ballot(g);
    n += (int)__builtin_popcountll(m);
    for (int l = 0; l < 64; ++l) {                 // fixed (ascending) order
      const float x = __shfl(v, l, 64);
      if ((m >> l) & 1ull) sum += x;
    }
  }
  if (threadIdx.x == 0) *mean = n > 0 ? sum / (float)n : 0.f;
}

int chk7(const char *k) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c: %s launch failed: %s\n", k, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // namespace

extern "C" int s2c_proposal_decode(int B, int K, int nout, int num_heading_bin,
                                   int num_size_cluster, int num_class, const float *net,
                                   const float *center, const float *mean_size_f32,
                                   const double *mean_size_f64, double *bbox_corner,
                                   long long *bbox_mask, long long *sem_cls,
                                   long long *size_class, void *stream) {
  if (B <= 0 || K <= 0 || num_heading_bin <= 0 || num_size_cluster <= 0 || num_class <= 0 ||
      nout != 2 + 3 + 2 * num_heading_bin + 4 * num_size_cluster + num_class || !net ||
      !center || !mean_size_f32 || !mean_size_f64 || !bbox_corner || !bbox_mask || !sem_cls)
    return -1;
  const int total = B * K;
  hipLaunchKernelGGL(proposal_decode_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, total, nout, num_heading_bin, num_size_cluster,
                     num_class, net, center, mean_size_f32, mean_size_f64, bbox_corner,
                     bbox_mask, sem_cls, size_class);
  return chk7("proposal_decode");
}

extern "C" int s2c_select_target(int B, int K, const double *bbox_corner,
                                 const double *ref_box_corner, long long *target_ids,
                                 float *target_ious, void *stream) {
  if (B <= 0 || K <= 0 || !bbox_corner || !ref_box_corner || !target_ids || !target_ious)
    return -1;
  hipLaunchKernelGGL(select_target_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, K,
                     bbox_corner, ref_box_corner, target_ids, target_ious);
  return chk7("select_target");
}

extern "C" int s2c_good_bbox_stats(int B, const float *ious, float min_iou, unsigned char *good,
                                   float *mean, void *stream) {
  if (B <= 0 || !ious || !good || !mean) return -1;
  hipLaunchKernelGGL(good_bbox_stats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, B, ious,
                     min_iou, (bool *)good, mean);
  return chk7("good_bbox_stats");
}

// ---------------------------------------------------------------------------------------
// Vote head (models/voting_module.py:49-58 + models/capnet.py:97-98): from the rows of the
// vote MLP net (M, 3 + C) = [xyz offset | feature residual],
//     vote_xyz = seed_xyz + net[:, 0:3]
//     y        = f / ||f||_2,   f = seed_features + net[:, 3:]
// and its backward, one launch each (the framework path: 6 + ~15 kernels).  Wave per row.
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void vote_head_fwd_kernel(
    int M, int C, const float *__restrict__ net, const float *__restrict__ seed_xyz,
    const float *__restrict__ seed_feat, long long seed_ld, float *__restrict__ vote_xyz,
    float *__restrict__ y, float *__restrict__ norm_out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float *n = net + (size_t)row * (3 + C);
  const float *s = seed_feat + (size_t)row * seed_ld;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float f = s[c] + n[3 + c];
    ss += f * f;
  }
  ss = wave_sum(ss);
  const float nrm = sqrtf(ss);
  for (int c = lane; c < C; c += 64) y[(size_t)row * C + c] = (s[c] + n[3 + c]) / nrm;
  if (lane < 3) vote_xyz[(size_t)row * 3 + lane] = seed_xyz[(size_t)row * 3 + lane] + n[lane];
  if (lane == 0) norm_out[row] = nrm;
}

// d_net (M, 3+C) = [g_xyz | (g_y - y <g_y, y>) / norm]
__global__ __launch_bounds__(256) void vote_head_bwd_kernel(
    int M, int C, const float *__restrict__ g_xyz, const float *__restrict__ g_y,
    long long gy_ld, long long gy_cs, const float *__restrict__ y,
    const float *__restrict__ norm, float *__restrict__ d_net, float *__restrict__ d_seed) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float *yy = y + (size_t)row * C;
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot += g_y[(size_t)row * gy_ld + (size_t)c * gy_cs] * yy[c];
  dot = wave_sum(dot);
  const float inv = 1.0f / norm[row];
  float *d = d_net + (size_t)row * (3 + C);
  for (int c = lane; c < C; c += 64) {
    const float v = (g_y[(size_t)row * gy_ld + (size_t)c * gy_cs] - yy[c] * dot) * inv;
    d[3 + c] = v;
    if (d_seed) d_seed[(size_t)row * C + c] = v;
  }
  if (lane < 3) d[lane] = g_xyz ? g_xyz[(size_t)row * 3 + lane] : 0.0f;
}

}  // namespace

extern "C" int s2c_vote_head_fwd(int M, int C, const float *net, const float *seed_xyz,
                                 const float *seed_feat, long long seed_ld, float *vote_xyz,
                                 float *y, float *norm, void *stream) {
  if (M <= 0 || C <= 0 || !net || !seed_xyz || !seed_feat || seed_ld < C || !vote_xyz || !y ||
      !norm)
    return -1;
  hipLaunchKernelGGL(vote_head_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     M, C, net, seed_xyz, seed_feat, seed_ld, vote_xyz, y, norm);
  return chk7("vote_head_fwd");
}

extern "C" int s2c_vote_head_bwd(int M, int C, const float *g_xyz, const float *g_y,
                                 long long gy_row_stride, long long gy_col_stride,
                                 const float *y, const float *norm, float *d_net,
                                 float *d_seed, void *stream) {
  if (M <= 0 || C <= 0 || !g_y || !y || !norm || !d_net) return -1;
  hipLaunchKernelGGL(vote_head_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     M, C, g_xyz, g_y, gy_row_stride, gy_col_stride, y, norm, d_net, d_seed);
  return chk7("vote_head_bwd");
}

// ---------------------------------------------------------------------------------------
// Batched "prepare" launch: up to 8 matrix transposes (dst (cols x rows) = src^T, src row
// stride lds) and up to 8 buffers to zero, in ONE launch -- the set-up of the decoder's
// backward pass (7 transposed weight matrices + 3 accumulators) was 10 framework kernels.
namespace {

__global__ __launch_bounds__(256) void batch_prep_kernel(s2c_prep_args a) {
  __shared__ float tile[32][33];
  int blk = blockIdx.x;
  for (int j = 0; j < a.n_transpose; ++j) {
    const int tr = (a.rows[j] + 31) / 32, tc = (a.cols[j] + 31) / 32;
    const int nb = tr * tc;
    if (blk < nb) {
      const int r0 = (blk / tc) * 32, c0 = (blk % tc) * 32;
      const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
      const float *src = a.src[j];
      float *dst = a.dst[j];
      for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < a.rows[j] && c < a.cols[j]) ? src[(size_t)r * a.lds[j] + c] : 0.f;
      }
      __syncthreads();
      for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < a.cols[j] && r < a.rows[j]) dst[(size_t)c * a.rows[j] + r] = tile[tx][i];
      }
      return;
    }
    blk -= nb;
  }
  for (int j = 0; j < a.n_zero; ++j) {
    const long long nb = (a.zero_count[j] + 1023) / 1024;
    if (blk < nb) {
      const long long base = (long long)blk * 1024;
      for (int i = threadIdx.x; i < 1024; i += 256)
        if (base + i < a.zero_count[j]) a.zero[j][base + i] = 0.f;
      return;
    }
    blk -= (int)nb;
  }
}

}  // namespace

extern "C" int s2c_batch_prep(const s2c_prep_args *a, void *stream) {
  if (!a || a->n_transpose < 0 || a->n_transpose > 8 || a->n_zero < 0 || a->n_zero > 8) return -1;
  long long blocks = 0;
  for (int j = 0; j < a->n_transpose; ++j) {
    if (!a->src[j] || !a->dst[j] || a->rows[j] <= 0 || a->cols[j] <= 0 || a->lds[j] < a->cols[j])
      return -1;
    blocks += (long long)((a->rows[j] + 31) / 32) * ((a->cols[j] + 31) / 32);
  }
  for (int j = 0; j < a->n_zero; ++j) {
    if (!a->zero[j] || a->zero_count[j] <= 0) return -1;
    blocks += (a->zero_count[j] + 1023) / 1024;
  }
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(batch_prep_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, *a);
  return chk7("batch_prep");
}

// ---------------------------------------------------------------------------------------
// Batched partial-sum: up to S2C_COLSUM_MAX_JOBS jobs out[j][e] = sum_{s < S[j]} part[j][s * n[j] + e] in ONE
// launch (the split-K partial products of the weight gradients of one layer stack; each was
// its own framework reduction kernel).  Fixed summation order.
namespace {

// Workgroup = 64 outputs x 4 phases of the partial index (thread = (e, phase)): phase p adds
// the partials s = p, p + 4, ... on four independent chains, the four phase sums meet through
// LDS in a fixed order.  (One thread per output walked all S partials alone: 85 us for the
// 768 x 8256 table of the pooled-layer algebra.)
__global__ __launch_bounds__(256) void multi_colsum_kernel(s2c_colsum_args a) {
  __shared__ float s_ph[4][64];
  long long blk = blockIdx.x;
  const int el = threadIdx.x & 63, ph = threadIdx.x >> 6;
  for (int j = 0; j < a.n_jobs; ++j) {
    const long long nb = (a.n[j] + 63) / 64;
    if (blk < nb) {
      const long long e = blk * 64 + el;
      const long long n = a.n[j];
      const int S = a.S[j];
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      if (e < n) {
        const float *p = a.part[j] + e;
        int s = ph;
        for (; s + 28 < S; s += 32) {         // eight loads in flight, four chains, fixed order
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = p[(long long)(s + 4 * u) * n];
          s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
          s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
        }
        for (; s + 12 < S; s += 16) {
          s0 += p[(long long)s * n];
          s1 += p[(long long)(s + 4) * n];
          s2 += p[(long long)(s + 8) * n];
          s3 += p[(long long)(s + 12) * n];
        }
        for (; s < S; s += 4) s0 += p[(long long)s * n];
      }
      s_ph[ph][el] = (s0 + s1) + (s2 + s3);
      __syncthreads();
      if (ph == 0 && e < n) {
        float tot = (s_ph[0][el] + s_ph[1][el]) + (s_ph[2][el] + s_ph[3][el]);
        if (a.sub[j] != nullptr) {
          const int nc = a.ncol[j], sc = a.sub_cols[j];
          const long long row = e / nc;
          const int c = (int)(e - row * nc);
          if (c < sc) {
            const long long rows = n / nc;
            const float *q = a.sub[j] + row * sc + c;
            float t = 0.f;
            for (int s = 0; s < a.sub_S[j]; ++s) t += q[(long long)s * rows * sc];
            tot -= t;
            if (a.sub_div[j] != 0.f) tot /= a.sub_div[j];
          }
        }
        a.out[j][e] = tot;
      }
      return;
    }
    blk -= nb;
  }
}

}  // namespace

extern "C" int s2c_multi_colsum(const s2c_colsum_args *a, void *stream) {
  if (!a || a->n_jobs <= 0 || a->n_jobs > S2C_COLSUM_MAX_JOBS) return -1;
  long long blocks = 0;
  for (int j = 0; j < a->n_jobs; ++j) {
    if (!a->part[j] || !a->out[j] || a->S[j] <= 0 || a->n[j] <= 0) return -1;
    if (a->sub[j] && (a->sub_S[j] <= 0 || a->ncol[j] <= 0 || a->sub_cols[j] <= 0 ||
                      a->sub_cols[j] > a->ncol[j] || a->n[j] % a->ncol[j] != 0))
      return -1;
    blocks += (a->n[j] + 63) / 64;
  }
  hipLaunchKernelGGL(multi_colsum_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, *a);
  return chk7("multi_colsum");
}

// ---------------------------------------------------------------------------------------
// Batched row sums: up to 16 jobs out[j][c] = sum_{m < M[j]} X[j][m * ld[j] + c] (the bias
// gradients of a layer stack / of the decoder: each was its own framework reduction).
// Workgroup = (job, 64 columns): 64 columns x 16 row phases, fixed combination order.
namespace {

__global__ __launch_bounds__(1024) void multi_rowsum_kernel(s2c_rowsum_args a) {
  // chunk_rows > 0: a job is cut into slabs of chunk_rows rows, slab q of job j writes its
  // sums to out[j] + q * C[j] (the caller adds the slabs up, s2c_multi_colsum)
  __shared__ float s_part[16][64];
  int blk = blockIdx.x;
  for (int j = 0; j < a.n_jobs; ++j) {
    const int ncg = (a.C[j] + 63) / 64;
    const long long rows = a.chunk_rows > 0 ? a.chunk_rows : a.M[j];
    const int nslab = (int)((a.M[j] + rows - 1) / rows);
    const int nb = ncg * nslab;
    if (blk < nb) {
      const int slab = blk / ncg, cg = blk - slab * ncg;
      const int col = cg * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
      const long long m0 = (long long)slab * rows;
      const long long m1 = m0 + rows < a.M[j] ? m0 + rows : a.M[j];
      float s = 0.f;
      if (col < a.C[j]) {
        const float *x = a.X[j] + col;
        const long long ld = a.ld[j];
        // eight row loads in flight (one at a time, a 1024-row slab was 64 dependent L2 round
        // trips per thread: 21 us per launch); same order of the additions
        long long m = m0 + ph;
        for (; m + 16 * 7 < m1; m += 16 * 8) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = x[(m + 16 * u) * ld];
#pragma unroll
          for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; m < m1; m += 16) s += x[m * ld];
      }
      s_part[ph][threadIdx.x & 63] = s;
      __syncthreads();
      if (ph == 0 && col < a.C[j]) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += s_part[q][threadIdx.x];
        a.out[j][(size_t)slab * a.C[j] + col] = t;
      }
      return;
    }
    blk -= nb;
  }
}

}  // namespace

extern "C" int s2c_multi_rowsum(const s2c_rowsum_args *a, void *stream) {
  if (!a || a->n_jobs <= 0 || a->n_jobs > 16) return -1;
  long long blocks = 0;
  for (int j = 0; j < a->n_jobs; ++j) {
    if (!a->X[j] || !a->out[j] || a->M[j] <= 0 || a->C[j] <= 0 || a->ld[j] < a->C[j]) return -1;
    const long long rows = a->chunk_rows > 0 ? a->chunk_rows : a->M[j];
    blocks += (long long)((a->C[j] + 63) / 64) * ((a->M[j] + rows - 1) / rows);
  }
  hipLaunchKernelGGL(multi_rowsum_kernel, dim3((unsigned)blocks), dim3(1024), 0,
                     (hipStream_t)stream, *a);
  return chk7("multi_rowsum");
}
