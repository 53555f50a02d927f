// s2c_graph.hip -- relational-graph glue of GraphModule / TopDownSceneCaptionModule as
// HIP kernels.
//
// query_locals: `_query_locals` (models/graph_module.py:182-222, duplicated at
// models/caption_module.py:322-381): for a target box, the distance from its 8
// corners (query_mode "corner") or its centre to every box centre, with invalid
// boxes (objectness 0), boxes overlapping the target (IoU >= threshold) and the
// target itself pushed to 1e30 (self -> 0 with include_self), then the num_locals
// nearest.  The reference runs it in a Python loop over the K targets (~12 launches
// each); the batched torch restatement still needs ~35 launches and materialises
// (B,T,8,K,3) float64 temporaries.  Here: one launch, one wave per (scene, target).
// float64 like the reference (bbox_corner is float64).  Top-L ties (equal
// distances): smallest index first (torch.topk leaves the order unspecified).
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

using namespace s2c;

namespace {

constexpr int QL_MAXK = 1024;
constexpr int QL_WAVES = 8;      // targets per block

__device__ __forceinline__ u64 wave_min_u64(u64 v) { return ~wave_max_u64(~v); }

__global__ __launch_bounds__(64 * QL_WAVES) void query_locals_kernel(
    int K, int T, int L, const double *__restrict__ corners,
    const long long *__restrict__ object_masks, const long long *__restrict__ target_ids,
    int corner_mode, int include_self, double overlay_threshold,
    float *__restrict__ local_masks, long long *__restrict__ ids_out) {
  __shared__ double s_min[QL_MAXK * 3], s_max[QL_MAXK * 3];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double *cb = corners + (size_t)b * K * 24;
  for (int k = threadIdx.x; k < K; k += 64 * QL_WAVES) {
    double lo[3], hi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = cb[k * 24 + c]; hi[c] = lo[c]; }
    for (int j = 1; j < 8; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double v = cb[k * 24 + j * 3 + c];
        lo[c] = fmin(lo[c], v);
        hi[c] = fmax(hi[c], v);
      }
#pragma unroll
    for (int c = 0; c < 3; ++c) { s_min[k * 3 + c] = lo[c]; s_max[k * 3 + c] = hi[c]; }
  }
  __syncthreads();
  const int t = blockIdx.x * QL_WAVES + wave;
  if (t >= T) return;
  const int tid = (int)target_ids[(size_t)b * T + t];
  // query points: the 8 corners, or the centre
  double qx[8], qy[8], qz[8];
  const int nq = corner_mode ? 8 : 1;
  if (corner_mode) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      qx[j] = cb[tid * 24 + j * 3];
      qy[j] = cb[tid * 24 + j * 3 + 1];
      qz[j] = cb[tid * 24 + j * 3 + 2];
    }
  } else {
    qx[0] = (s_min[tid * 3] + s_max[tid * 3]) / 2;
    qy[0] = (s_min[tid * 3 + 1] + s_max[tid * 3 + 1]) / 2;
    qz[0] = (s_min[tid * 3 + 2] + s_max[tid * 3 + 2]) / 2;
  }
  const double tl[3] = {s_min[tid * 3], s_min[tid * 3 + 1], s_min[tid * 3 + 2]};
  const double th[3] = {s_max[tid * 3], s_max[tid * 3 + 1], s_max[tid * 3 + 2]};
  const double tvol = (th[0] - tl[0]) * (th[1] - tl[1]) * (th[2] - tl[2]);

  constexpr int PER = QL_MAXK / 64;
  u64 val[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int k = lane + 64 * j;
    u64 bits = ~0ull;                       // beyond K: never selected
    if (k < K) {
      const double cx = (s_min[k * 3] + s_max[k * 3]) / 2;
      const double cy = (s_min[k * 3 + 1] + s_max[k * 3 + 1]) / 2;
      const double cz = (s_min[k * 3 + 2] + s_max[k * 3 + 2]) / 2;
      double d2 = 0.0;
      for (int q = 0; q < nq; ++q) {
        const double dx = qx[q] - cx, dy = qy[q] - cy, dz = qz[q] - cz;
        const double s = (dx * dx + dy * dy) + dz * dz;
        d2 = q == 0 ? s : fmin(d2, s);
      }
      double d = sqrt(d2 + 1e-8);          // sqrt is monotone: min commutes with it
      if (object_masks[(size_t)b * K + k] == 0) d = 1e30;
      // axis-aligned IoU with the target box (utils/box_util.py:183-209)
      double inter = 1.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double lo = fmax(tl[c], s_min[k * 3 + c]), hi = fmin(th[c], s_max[k * 3 + c]);
        inter *= fmax(hi - lo, 0.0);
      }
      const double vol = (s_max[k * 3] - s_min[k * 3]) * (s_max[k * 3 + 1] - s_min[k * 3 + 1]) *
                         (s_max[k * 3 + 2] - s_min[k * 3 + 2]);
      const double iou = inter / (tvol + vol - inter + 1e-8);
      if (iou >= overlay_threshold) d = 1e30;
      if (k == tid) d = include_self ? 0.0 : 1e30;
      bits = (u64)__double_as_longlong(d);  // d >= 0: bit order == value order
    }
    val[j] = bits;
  }
  // L rounds of wave arg-min
  int my_pick = -1;                          // lane r keeps the r-th selected id
  for (int r = 0; r < L; ++r) {
    u64 best = ~0ull;
    int bj = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j)
      if (val[j] < best) { best = val[j]; bj = j; }
    const u64 wmin = wave_min_u64(best);
    // among the lanes holding the minimum, the smallest candidate index wins
    const u32 cand = best == wmin ? (u32)(lane + 64 * bj) : 0xFFFFFFFFu;
    const u32 widx = (u32)~wave_max_u64((u64)(u32)(~cand));      // min over the wave
    if (cand == widx) {
#pragma unroll
      for (int j = 0; j < PER; ++j)
        if (j == bj) val[j] = ~0ull;       // taken
    }
    if (lane == r) my_pick = (int)widx;
  }
  // publish: the 0/1 row of the mask, and the picks sorted ascending (rank by counting)
  float *lm = local_masks + ((size_t)b * T + t) * K;
  for (int k = lane; k < K; k += 64) {
    int hit = 0;
    for (int r = 0; r < L; ++r) hit |= (__builtin_amdgcn_readlane(my_pick, r) == k) ? 1 : 0;
    lm[k] = hit ? 1.0f : 0.0f;
  }
  int rank = 0;
  for (int r = 0; r < L; ++r) {
    const int other = __builtin_amdgcn_readlane(my_pick, r);
    rank += (other < my_pick) ? 1 : 0;
  }
  if (lane < L) ids_out[((size_t)b * T + t) * L + rank] = my_pick;
}

// ---------------------------------------------------------------------------
// EdgeConv message passing (models/graph_module.py:74-115 on torch_geometric's
// source_to_target flow): edge e = (b, i, l) runs from row i to column j = nbr[b,i,l];
//   rows[e] = [ x[b,j,:] | x[b,i,:] - x[b,j,:] ]              (message input)
//   out[b,j,:] += msg[e,:] * slot[e] ; msgm[e,:] = msg[e,:] * slot[e]   (aggregation "add")
// One wave per edge, lanes over the feature dimension; row-coalesced float atomics.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_rows_kernel(
    int K, int L, int F, const float *__restrict__ x, const long long *__restrict__ nbr,
    float *__restrict__ rows, long long E) {
  const int lane = threadIdx.x & 63;
  for (long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); e < E;
       e += (long long)gridDim.x * 4) {
    const long long bi = e / L, b = bi / K;
    const long long j = nbr[e];
    const float *xi = x + bi * F, *xj = x + (b * K + j) * F;
    float *r = rows + e * 2 * F;
    for (int f = lane; f < F; f += 64) {
      const float vj = xj[f];
      r[f] = vj;
      r[F + f] = xi[f] - vj;
    }
  }
}

// dx[b,j,:] += dR[e,:F] - dR[e,F:] ; dx[b,i,:] += dR[e,F:]      (dx zeroed by the caller)
// One wave per CENTRE i: its L edges' contributions to dx[b,i] are summed in registers and added
// once (L + 1 float atomics per centre and feature instead of 2 L, and the L same-address adds
// of consecutive waves -- which the memory side serialises -- are gone: 42 -> 25 us at
// 8 x 256 x 10 edges x 128 features).
__global__ __launch_bounds__(256) void edge_rows_grad_kernel(
    int K, int L, int F, const float *__restrict__ dR, const long long *__restrict__ nbr,
    float *__restrict__ dx, long long E) {
  const int lane = threadIdx.x & 63;
  const long long nodes = E / L;
  for (long long bi = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); bi < nodes;
       bi += (long long)gridDim.x * 4) {
    const long long b = bi / K;
    float *di = dx + bi * F;
    for (int f = lane; f < F; f += 64) {
      float acc = 0.0f;
      // five edges of loads in flight (one edge per iteration left two loads and an index
      // outstanding per wave between atomics: latency-bound)
      for (int l0 = 0; l0 < L; l0 += 5) {
        float a[5], d[5];
        long long to[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const long long e = bi * L + (l0 + u < L ? l0 + u : l0);
          const float *g = dR + e * 2 * F;
          a[u] = g[f]; d[u] = g[F + f];
          to[u] = nbr[e];
        }
#pragma unroll
        for (int u = 0; u < 5; ++u)
          if (l0 + u < L) {
            atomicAdd(dx + (b * K + to[u]) * F + f, a[u] - d[u]);
            acc += d[u];
          }
      }
      atomicAdd(di + f, acc);
    }
  }
}

__global__ __launch_bounds__(256) void edge_scatter_kernel(
    int K, int L, int F, const float *__restrict__ msg, const long long *__restrict__ nbr,
    const unsigned char *__restrict__ slot, float *__restrict__ out,
    float *__restrict__ msgm, long long E) {
  const int lane = threadIdx.x & 63;
  for (long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); e < E;
       e += (long long)gridDim.x * 4) {
    const long long b = e / ((long long)K * L);
    const bool on = slot[e] != 0;
    const float *m = msg + e * F;
    float *o = out + (b * K + nbr[e]) * F;
    for (int f = lane; f < F; f += 64) {
      const float v = on ? m[f] : 0.0f;
      msgm[e * F + f] = v;
      if (on) atomicAdd(o + f, v);
    }
  }
}

// d_msg[e,:] = (d_out[b,j,:] + d_msgm[e,:]) * slot[e]        (d_msgm may be NULL)
__global__ __launch_bounds__(256) void edge_scatter_grad_kernel(
    int K, int L, int F, const float *__restrict__ d_out, const float *__restrict__ d_msgm,
    const long long *__restrict__ nbr, const unsigned char *__restrict__ slot,
    float *__restrict__ d_msg, long long E) {
  const int lane = threadIdx.x & 63;
  for (long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); e < E;
       e += (long long)gridDim.x * 4) {
    const long long b = e / ((long long)K * L);
    const bool on = slot[e] != 0;
    const float *g = d_out + (b * K + nbr[e]) * F;
    for (int f = lane; f < F; f += 64) {
      float v = 0.0f;
      if (on) v = g[f] + (d_msgm ? d_msgm[e * F + f] : 0.0f);
      d_msg[e * F + f] = v;
    }
  }
}

static unsigned edge_grid(long long E) {
  long long b = (E + 3) / 4;
  return (unsigned)(b > 256 * 16 ? 256 * 16 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int s2c_query_locals(int B, int K, int T, int L, const double *corners,
                                const long long *object_masks,
                                const long long *target_ids, int corner_mode,
                                int include_self, double overlay_threshold,
                                float *local_masks, long long *ids_out, void *stream) {
  if (B <= 0 || K <= 0 || K > QL_MAXK || T <= 0 || L <= 0 || L > 64 || L > K || !corners ||
      !object_masks || !target_ids || !local_masks || !ids_out)
    return -1;
  hipLaunchKernelGGL(query_locals_kernel, dim3((T + QL_WAVES - 1) / QL_WAVES, B),
                     dim3(64 * QL_WAVES), 0, (hipStream_t)stream, K, T, L, corners,
                     object_masks, target_ids, corner_mode, include_self, overlay_threshold,
                     local_masks, ids_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c: query_locals launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

static int chk_graph(const char *k) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c: %s launch failed: %s\n", k, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int s2c_edge_rows(int B, int K, int L, int F, const float *x, const long long *nbr,
                             float *rows, void *stream) {
  if (B <= 0 || K <= 0 || L <= 0 || F <= 0 || !x || !nbr || !rows) return -1;
  const long long E = (long long)B * K * L;
  hipLaunchKernelGGL(edge_rows_kernel, dim3(edge_grid(E)), dim3(256), 0, (hipStream_t)stream,
                     K, L, F, x, nbr, rows, E);
  return chk_graph("edge_rows");
}

extern "C" int s2c_edge_rows_grad(int B, int K, int L, int F, const float *d_rows,
                                  const long long *nbr, float *dx, void *stream) {
  if (B <= 0 || K <= 0 || L <= 0 || F <= 0 || !d_rows || !nbr || !dx) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (zero_async(dx, sizeof(float) * (size_t)B * K * F, st) != hipSuccess) return -1;
  const long long E = (long long)B * K * L;
  hipLaunchKernelGGL(edge_rows_grad_kernel, dim3(edge_grid(E / L)), dim3(256), 0, st, K, L, F,
                     d_rows, nbr, dx, E);
  return chk_graph("edge_rows_grad");
}

extern "C" int s2c_edge_scatter(int B, int K, int L, int F, const float *msg,
                                const long long *nbr, const unsigned char *slot, float *out,
                                float *msgm, void *stream) {
  if (B <= 0 || K <= 0 || L <= 0 || F <= 0 || !msg || !nbr || !slot || !out || !msgm) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (zero_async(out, sizeof(float) * (size_t)B * K * F, st) != hipSuccess) return -1;
  const long long E = (long long)B * K * L;
  hipLaunchKernelGGL(edge_scatter_kernel, dim3(edge_grid(E)), dim3(256), 0, st, K, L, F, msg,
                     nbr, slot, out, msgm, E);
  return chk_graph("edge_scatter");
}

extern "C" int s2c_edge_scatter_grad(int B, int K, int L, int F, const float *d_out,
                                     const float *d_msgm, const long long *nbr,
                                     const unsigned char *slot, float *d_msg, void *stream) {
  if (B <= 0 || K <= 0 || L <= 0 || F <= 0 || !d_out || !nbr || !slot || !d_msg) return -1;
  const long long E = (long long)B * K * L;
  hipLaunchKernelGGL(edge_scatter_grad_kernel, dim3(edge_grid(E)), dim3(256), 0,
                     (hipStream_t)stream, K, L, F, d_out, d_msgm, nbr, slot, d_msg, E);
  return chk_graph("edge_scatter_grad");
}

// ---------------------------------------------------------------------------------------
// The teacher-forced decoder's inputs in one launch (models/caption_module.py:250-292 with
// _add_relation_feat :394-414 restricted to the rows the decoder reads): target_feats[b] =
// obj[b, tgt[b]] and local[b, l] = obj[b, id] + sum_t [nbr[b, tgt[b], t] == id] rel[b, tgt[b], t],
// id = local_ids[b, l] -- what gather / clone / scatter_add_ / gather over the whole (B, K, F) tensor
// produced in seven framework launches.  rel == nullptr: no relation rows (use_relation off).
__global__ __launch_bounds__(256) void local_feats_kernel(
    int K, int L, int LR, int F, const float *__restrict__ obj, const float *__restrict__ rel,
    const long long *__restrict__ nbr, const long long *__restrict__ tgt,
    const long long *__restrict__ local_ids, float *__restrict__ target_feats,
    float *__restrict__ local) {
  const int b = blockIdx.x / (L + 1), l = blockIdx.x % (L + 1);
  long long t0 = tgt[b];
  t0 = t0 < 0 ? 0 : (t0 >= K ? K - 1 : t0);
  if (l == L) {
    for (int f = threadIdx.x; f < F; f += blockDim.x)
      target_feats[(size_t)b * F + f] = obj[((size_t)b * K + t0) * F + f];
    return;
  }
  long long id = local_ids[(size_t)b * L + l];
  id = id < 0 ? 0 : (id >= K ? K - 1 : id);
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float v = obj[((size_t)b * K + id) * F + f];
    if (rel != nullptr)
      for (int t = 0; t < LR; ++t)
        if (nbr[((size_t)b * K + t0) * LR + t] == id) v += rel[(((size_t)b * K + t0) * LR + t) * F + f];
    local[((size_t)b * L + l) * F + f] = v;
  }
}

// gradients of the above, every element of d_obj (B, K, F) and d_rel (B, K, LR, F) written (no zero
// fill, no atomics): block = one proposal row (b, k)
__global__ __launch_bounds__(256) void local_feats_grad_kernel(
    int K, int L, int LR, int F, const float *__restrict__ d_target,
    const float *__restrict__ d_local, const long long *__restrict__ nbr,
    const long long *__restrict__ tgt, const long long *__restrict__ local_ids,
    float *__restrict__ d_obj, float *__restrict__ d_rel) {
  const int b = blockIdx.x / K, k = blockIdx.x % K;
  long long t0 = tgt[b];
  t0 = t0 < 0 ? 0 : (t0 >= K ? K - 1 : t0);
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float v = 0.f;
    for (int l = 0; l < L; ++l) {
      long long id = local_ids[(size_t)b * L + l];
      id = id < 0 ? 0 : (id >= K ? K - 1 : id);
      if (id == k) v += d_local[((size_t)b * L + l) * F + f];
    }
    if (k == t0 && d_target != nullptr) v += d_target[(size_t)b * F + f];
    d_obj[((size_t)b * K + k) * F + f] = v;
    if (d_rel != nullptr)
      for (int t = 0; t < LR; ++t) {
        float r = 0.f;
        if (k == t0) {
          const long long idt = nbr[((size_t)b * K + t0) * LR + t];
          for (int l = 0; l < L; ++l) {
            long long id = local_ids[(size_t)b * L + l];
            id = id < 0 ? 0 : (id >= K ? K - 1 : id);
            if (id == idt) r += d_local[((size_t)b * L + l) * F + f];
          }
        }
        d_rel[(((size_t)b * K + k) * LR + t) * F + f] = r;
      }
  }
}

extern "C" int s2c_local_feats(int B, int K, int L, int LR, int F, const float *obj, const float *rel,
                               const long long *nbr, const long long *tgt,
                               const long long *local_ids, float *target_feats, float *local,
                               void *stream) {
  if (B <= 0 || K <= 0 || L <= 0 || F <= 0 || !obj || !tgt || !local_ids || !target_feats || !local ||
      (rel && (!nbr || LR <= 0)))
    return -1;
  hipLaunchKernelGGL(local_feats_kernel, dim3(B * (L + 1)), dim3(F >= 256 ? 256 : (F + 63) / 64 * 64), 0,
                     (hipStream_t)stream, K, L, LR, F, obj, rel, nbr, tgt, local_ids, target_feats, local);
  return chk_graph("local_feats");
}

extern "C" int s2c_local_feats_grad(int B, int K, int L, int LR, int F, const float *d_target,
                                    const float *d_local, const long long *nbr, const long long *tgt,
                                    const long long *local_ids, float *d_obj, float *d_rel,
                                    void *stream) {
  if (B <= 0 || K <= 0 || L <= 0 || F <= 0 || !d_local || !tgt || !local_ids || !d_obj ||
      (d_rel && (!nbr || LR <= 0)))
    return -1;
  hipLaunchKernelGGL(local_feats_grad_kernel, dim3(B * K), dim3(F >= 256 ? 256 : (F + 63) / 64 * 64), 0,
                     (hipStream_t)stream, K, L, LR, F, d_target, d_local, nbr, tgt, local_ids, d_obj, d_rel);
  return chk_graph("local_feats_grad");
}
