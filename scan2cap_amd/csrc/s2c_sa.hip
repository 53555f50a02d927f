// s2c_sa.hip -- point-major set-abstraction kernels (MI355X-first path).
//
// The reference materialises channel-major grouped tensors
// (B, 3+C, npoint, nsample) (pointnet2_utils.py:347-359) and runs cuDNN-style
// 1x1 Conv2d + BatchNorm2d + ReLU + max_pool2d over them
// (pointnet2_modules.py:251-257, pytorch_utils.py:11-120).  Here every grouped
// tensor is POINT-MAJOR: one row per (scene, centre, sample) with the channels
// contiguous, so
//   * a neighbour is ONE contiguous row read of the (B,N,3+C) input instead of
//     3+C scattered 4-byte reads (group_points_gpu.cu:20-26),
//   * the shared MLP is a plain row-major GEMM  Y = X W^T,
//   * BatchNorm statistics are column reductions, BN+ReLU(+max over the nsample
//     rows of a centre) one fused elementwise pass,
//   * the gradient scatter is a row-coalesced atomic add.
// The kernels below are the non-GEMM parts (HBM-bound): gather rows, column
// statistics, BN+ReLU apply, BN+ReLU+max-pool, their backward passes and the
// row scatter.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>
#include <stdlib.h>

using namespace s2c;

static thread_local char g_err2[256] = "";
static int fail2(const char *what) {
  snprintf(g_err2, sizeof(g_err2), "s2c: invalid argument: %s", what);
  return -1;
}
static int check2(const char *kernel) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err2, sizeof(g_err2), "s2c: %s launch failed: %s", kernel,
             hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
extern "C" const char *s2c_fused_last_error_string(void) { return g_err2; }

static unsigned grid1d(long long items, int per_block, long long cap = 256 * 16) {
  long long g = (items + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// ---------------------------------------------------------------------------
// 1. gather rows:  X[(b,j,k), :] = [ (xyz[b,idx] - new_xyz[b,j]) (/ radius) ,
//                                    feats[b, idx, 0:C] ]
// (QueryAndGroup.forward, pointnet2_utils.py:347-359, in point-major form; the
// subtraction and the division are two separate roundings as in the reference.)
// One wave per output row; lanes run over the channels => both the source row
// (a contiguous 4*(C) B segment of the point-major input) and the destination
// row are coalesced.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sa_gather_rows_kernel(
    int n, int m, int ns, int C, long long feat_row_stride,
    long long feat_batch_stride, float radius, int normalize,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ feats, const int *__restrict__ idx,
    float *__restrict__ X, long long rows) {
  const int lane = threadIdx.x & 63;
  const int ld = 3 + C;
  for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows;
       r += (long long)gridDim.x * 4) {
    const long long bj = r / ns;           // b*m + j
    const long long b = bj / m;
    const int p = idx[r];                   // wave-uniform
    float *dst = X + r * ld;
    if (lane < 3) {
      float v = xyz[(b * n + p) * 3 + lane] - new_xyz[bj * 3 + lane];
      if (normalize) v = v / radius;
      dst[lane] = v;
    }
    if (C > 0) {
      const float *src = feats + b * feat_batch_stride + (long long)p * feat_row_stride;
      for (int c = lane; c < C; c += 64) dst[3 + c] = src[c];
    }
  }
}

extern "C" int s2c_sa_gather_rows(int b, int n, int m, int ns, int C,
                                  long long feat_row_stride,
                                  long long feat_batch_stride, float radius,
                                  int normalize, const float *xyz,
                                  const float *new_xyz, const float *feats,
                                  const int *idx, float *X, void *stream) {
  if (b < 0 || n <= 0 || m < 0 || ns < 0 || C < 0) return fail2("sa_gather_rows sizes");
  const long long rows = (long long)b * m * ns;
  if (rows == 0) return 0;
  if (!xyz || !new_xyz || !idx || !X || (C > 0 && !feats))
    return fail2("sa_gather_rows: null pointer");
  hipLaunchKernelGGL(sa_gather_rows_kernel, dim3(grid1d(rows, 4, 256 * 32)),
                     dim3(256), 0, (hipStream_t)stream, n, m, ns, C,
                     feat_row_stride, feat_batch_stride, radius, normalize, xyz,
                     new_xyz, feats, idx, X, rows);
  return check2("sa_gather_rows");
}

// backward of (1): scatter-add dX rows into d_feats (B,N,C) [row-coalesced
// hardware float atomics], and optionally into d_xyz (B,N,3) / d_new_xyz (B,m,3).
__global__ __launch_bounds__(256) void sa_scatter_rows_kernel(
    int n, int m, int ns, int C, float radius, int normalize,
    const float *__restrict__ dX, const int *__restrict__ idx,
    float *__restrict__ d_feats, float *__restrict__ d_xyz,
    float *__restrict__ d_new_xyz, long long rows) {
  const int lane = threadIdx.x & 63;
  const int ld = 3 + C;
  for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows;
       r += (long long)gridDim.x * 4) {
    const long long bj = r / ns;
    const long long b = bj / m;
    const int p = idx[r];
    const float *src = dX + r * ld;
    if (d_xyz != nullptr && lane < 3) {
      float g = src[lane];
      if (normalize) g = g / radius;
      atomicAdd(d_xyz + (b * n + p) * 3 + lane, g);
      atomicAdd(d_new_xyz + bj * 3 + lane, -g);
    }
    if (d_feats != nullptr) {
      float *dst = d_feats + (b * n + p) * (long long)C;
      for (int c = lane; c < C; c += 64) atomicAdd(dst + c, src[3 + c]);
    }
  }
}

extern "C" int s2c_sa_scatter_rows(int b, int n, int m, int ns, int C,
                                   float radius, int normalize, const float *dX,
                                   const int *idx, float *d_feats, float *d_xyz,
                                   float *d_new_xyz, void *stream) {
  if (b < 0 || n <= 0 || m < 0 || ns < 0 || C < 0) return fail2("sa_scatter_rows sizes");
  hipStream_t st = (hipStream_t)stream;
  if (d_feats && (long long)b * n * C > 0)
    if (zero_async(d_feats, sizeof(float) * (size_t)b * n * C, st) != hipSuccess)
      return fail2("memset");
  if (d_xyz) {
    if (!d_new_xyz) return fail2("sa_scatter_rows: d_xyz needs d_new_xyz");
    if (zero_async(d_xyz, sizeof(float) * (size_t)b * n * 3, st) != hipSuccess ||
        zero_async(d_new_xyz, sizeof(float) * (size_t)b * m * 3, st) != hipSuccess)
      return fail2("memset");
  }
  const long long rows = (long long)b * m * ns;
  if (rows == 0) return 0;
  if (!dX || !idx) return fail2("sa_scatter_rows: null pointer");
  hipLaunchKernelGGL(sa_scatter_rows_kernel, dim3(grid1d(rows, 4, 256 * 32)),
                     dim3(256), 0, st, n, m, ns, C, radius, normalize, dX, idx,
                     d_feats, d_xyz, d_new_xyz, rows);
  return check2("sa_scatter_rows");
}

// Weight gradient of a gather-fused first layer WITHOUT re-materialising the gathered
// operand (used when xyz / features need no gradient, e.g. SA1 on the input cloud):
//   dW = dY^T G,  G[(b,j,s),:] = [ (xyz[b,p]-new_xyz[b,j])(/r) | feats[b,p,:] ], p = idx[row]
//      = [ (Z^T xyz - S^T new_xyz)(/r) | Z^T feats ]
// with Z[b,p,:] = sum of the dY rows that gathered point p (row-coalesced hardware
// atomics) and S[b,j,:] = sum over the ns rows of centre j (in-block, deterministic).
// The two products run over B*N point rows instead of B*m*ns gathered rows, and the
// (rows x (3+C)) operand (566 MB at SA1) is neither written nor re-read.
// Block = one centre; wave = every 4th sample row; lane = channel (strided).
// BNBWD: dY is not read but formed on the fly from the upstream gradient dA and the layer's
// pre-activation Y (the arithmetic of bn_bwd_apply_kernel, bit for bit): the first layer of a
// stack has no input gradient to produce, so its dY (268 MB at SA1) is neither written nor
// re-read -- it only ever feeds these sums.
struct BnBwdCoefs {
  const float *Y, *scale, *shift, *mean, *invstd, *coef;   // coef = 3 C (s2c_bn_relu_bwd_stats)
  int relu;
};

template <bool BNBWD, int U>
__global__ __launch_bounds__(256) void sa_scatter_sum_kernel(
    int n, int m, int ns, int C, const float *__restrict__ dY,
    const int *__restrict__ idx, float *__restrict__ Z, float *__restrict__ S, BnBwdCoefs bw) {
  __shared__ float s_sum[4][1024];
  __shared__ float s_pad[4][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long bj = blockIdx.x;
  const long long b = bj / m;
  // A ball with fewer than ns points is padded with its FIRST index (ball_query_gpu.cu:34-41): the
  // rows of a centre that gathered that point -- row 0 and every pad -- are summed here and leave
  // as ONE atomic row per channel.  (Synthetic cfg3 scenes: 79 % of SA1's rows, 87 % of SA2's; the
  // float atomics are what bounds this kernel.)
  const int p0 = idx[bj * ns];
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    float acc = 0.f, pad = 0.f;
    float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f, k0 = 0.f, k1 = 0.f, k2 = 0.f;
    if (BNBWD && c < C) {
      sc = bw.scale[c]; sh = bw.shift[c]; mu = bw.mean[c]; is = bw.invstd[c];
      k0 = bw.coef[c]; k1 = bw.coef[C + c]; k2 = bw.coef[2 * C + c];
    }
    // U rows of the wave in flight (a row per iteration left ONE index load and one row of loads
    // outstanding per wave: latency-bound at 1.9 TB/s); addresses clamped, every load unconditional
    const int cl = c < C ? c : C - 1;
    for (int s0 = wave; s0 < ns; s0 += 4 * U) {
      int p[U];
      float g[U], y[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int s = s0 + 4 * u;
        const long long r = bj * ns + (s < ns ? s : s0);
        p[u] = idx[r];
        g[u] = dY[r * C + cl];
        y[u] = BNBWD ? bw.Y[r * C + cl] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v = g[u];
        if (BNBWD) {
          if (bw.relu && !(y[u] * sc + sh > 0.f)) v = 0.f;
          v = k0 * (v - k1 - ((y[u] - mu) * is) * k2);
        }
        if (c < C && s0 + 4 * u < ns) {
          acc += v;
          if (p[u] == p0) pad += v;
          else atomicAdd(Z + (b * n + p[u]) * (long long)C + c, v);
        }
      }
    }
    if (c < C) { s_sum[wave][c] = acc; s_pad[wave][c] = pad; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    S[bj * C + c] = (s_sum[0][c] + s_sum[1][c]) + (s_sum[2][c] + s_sum[3][c]);
    atomicAdd(Z + (b * n + p0) * (long long)C + c, (s_pad[0][c] + s_pad[1][c]) + (s_pad[2][c] + s_pad[3][c]));
  }
}

extern "C" int s2c_sa_scatter_sum(int b, int n, int m, int ns, int C, const float *dY,
                                  const int *idx, float *Z, float *S, void *stream) {
  if (b <= 0 || n <= 0 || m <= 0 || ns <= 0 || C <= 0 || C > 1024 || !dY || !idx || !Z || !S)
    return fail2("sa_scatter_sum: sizes / null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (zero_async(Z, sizeof(float) * (size_t)b * n * C, st) != hipSuccess)
    return fail2("memset");
  if (ns > 32)
    hipLaunchKernelGGL((sa_scatter_sum_kernel<false, 8>), dim3((unsigned)(b * m)), dim3(256), 0, st, n,
                       m, ns, C, dY, idx, Z, S, BnBwdCoefs());
  else
    hipLaunchKernelGGL((sa_scatter_sum_kernel<false, 4>), dim3((unsigned)(b * m)), dim3(256), 0, st, n,
                       m, ns, C, dY, idx, Z, S, BnBwdCoefs());
  return check2("sa_scatter_sum");
}

// The same sums of dY = bn_relu_backward(dA, Y) without materialising dY (see BnBwdCoefs).
extern "C" int s2c_sa_scatter_sum_bn_bwd(int b, int n, int m, int ns, int C, const float *dA,
                                         const float *Y, const float *scale, const float *shift,
                                         const float *mean, const float *invstd,
                                         const float *coef, int relu, const int *idx, float *Z,
                                         float *S, void *stream) {
  if (b <= 0 || n <= 0 || m <= 0 || ns <= 0 || C <= 0 || C > 1024 || !dA || !Y || !scale ||
      !shift || !mean || !invstd || !coef || !idx || !Z || !S)
    return fail2("sa_scatter_sum_bn_bwd: sizes / null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (zero_async(Z, sizeof(float) * (size_t)b * n * C, st) != hipSuccess)
    return fail2("memset");
  BnBwdCoefs bw = {Y, scale, shift, mean, invstd, coef, relu};
  if (ns > 32)
    hipLaunchKernelGGL((sa_scatter_sum_kernel<true, 8>), dim3((unsigned)(b * m)), dim3(256), 0, st, n,
                       m, ns, C, dA, idx, Z, S, bw);
  else
    hipLaunchKernelGGL((sa_scatter_sum_kernel<true, 4>), dim3((unsigned)(b * m)), dim3(256), 0, st, n,
                       m, ns, C, dA, idx, Z, S, bw);
  return check2("sa_scatter_sum_bn_bwd");
}

// ---------------------------------------------------------------------------------------
// First layer of a set-abstraction stack in POINT SPACE.  The layer is linear and the grouping is
// a gather, so they commute:
//   Y[(b,j,s), :] = W [ (xyz[b,p] - new_xyz[b,j]) (/r) | feats[b,p,:] ],   p = idx[b,j,s]
//                 = P[b,p,:] + W_x rel(b,j,s),      P = feats W_f^T   (one row per POINT)
// (QueryAndGroup + the first Conv2d of the shared MLP: pointnet2_utils.py:347-359,
// pointnet2_modules.py:251-253, pytorch_utils.py:67-120.)  The feature product runs over the
// B n points, not over the B m ns gathered rows -- 3.3x fewer rows at SA1 (40000 points, 2048 x
// 64 samples), 16x / 8x / 8x / 4x at SA2 / SA3 / SA4 / vote aggregation -- on the plain rows GEMM;
// what is left per gathered row is this kernel: one contiguous N-float row of P (L2-resident:
// 10 MB per scene at SA1), three FMAs per output for the relative position -- computed from the
// coordinates exactly as the reference does, no cancellation of absolute positions -- a 16-byte
// store, and the BatchNorm column sums.  HBM-bound on the Y write.
//
// Wave = 64 consecutive rows: lane l first OWNS row l (index, point row, relative position:
// coalesced loads), then the wave walks its rows G = 64 / LPR at a time, LPR lanes x float4 per
// row, the owner's values fetched with ds_bpermute; eight steps of P loads in flight per wave.
struct GatherAddArgs {
  int n, m, ns, N, ldw, normalize, rpb, has_p;
  long long rows;
  float radius;
  const float *xyz, *new_xyz, *P, *W;
  const int *idx;
  float *Y, *partial;
  // inference epilogue (em != nullptr): Y = relu?(y * sc[c] + sh[c]), sc = gamma / sqrt(var + eps),
  // sh = beta - mean * sc -- the frozen BatchNorm (+ ReLU) behind the first layer
  // (affine_epilogue of s2c_gemm.hip, the arithmetic of bn_eval_coeffs + bn_relu)
  const float *eg, *eb, *em, *ev;
  float eeps;
  int erelu;
};

template <int LPR>
__global__ __launch_bounds__(256) void sa_gather_add_kernel(GatherAddArgs a) {
  constexpr int G = 64 / LPR, U = 8;
  __shared__ float s_stat[2][4][4 * LPR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane % LPR, lg = lane / LPR;
  const int c0 = 4 * lc;
  const bool cok = c0 < a.N;
  const int c0l = cok ? c0 : 0;
  float wx[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int d = 0; d < 3; ++d) wx[q][d] = a.W[(long long)(c0l + q) * a.ldw + d];
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const bool epi = a.em != nullptr;
  float esc[4] = {1.f, 1.f, 1.f, 1.f}, esh[4] = {0.f, 0.f, 0.f, 0.f};
  if (epi) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float invstd = 1.0f / sqrtf(a.ev[c0l + q] + a.eeps);
      esc[q] = (a.eg ? a.eg[c0l + q] : 1.0f) * invstd;
      esh[q] = (a.eb ? a.eb[c0l + q] : 0.0f) - a.em[c0l + q] * esc[q];
    }
  }
  const int rpw = a.rpb >> 2;                                   // rows per wave
  const long long w0 = (long long)blockIdx.x * a.rpb + (long long)wave * rpw;
  for (int ch = 0; ch < rpw; ch += 64) {
    const long long r0 = w0 + ch;
    long long left = a.rows - r0;
    if (left > rpw - ch) left = rpw - ch;
    const int nrow = left > 64 ? 64 : (int)left;                // wave-uniform
    if (nrow <= 0) break;
    // ---- owner phase: lane l <-> row r0 + l
    int pr = 0;
    float rx = 0.f, ry = 0.f, rz = 0.f;
    {
      const long long r = r0 + (lane < nrow ? lane : nrow - 1);
      const long long bj = r / a.ns;
      const long long b = bj / a.m;
      pr = (int)(b * a.n + a.idx[r]);
      const float *x = a.xyz + 3LL * pr, *c = a.new_xyz + 3 * bj;
      rx = x[0] - c[0]; ry = x[1] - c[1]; rz = x[2] - c[2];
      if (a.normalize) { rx = rx / a.radius; ry = ry / a.radius; rz = rz / a.radius; }
    }
    // ---- the wave's rows, G per step, U steps of loads in flight
    for (int st0 = 0; st0 < nrow; st0 += G * U) {
      float4 pv[U];
      int rr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int want = st0 + u * G + lg;
        rr[u] = want < nrow ? want : nrow - 1;                  // clamped: every load is valid
        pv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.has_p) {
          const int prr = __shfl(pr, rr[u], 64);
          pv[u] = *reinterpret_cast<const float4 *>(a.P + (long long)prr * a.N + c0l);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float x = __shfl(rx, rr[u], 64), y = __shfl(ry, rr[u], 64), z = __shfl(rz, rr[u], 64);
        float o[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          o[q] = __builtin_fmaf(wx[q][2], z, __builtin_fmaf(wx[q][1], y, __builtin_fmaf(wx[q][0], x, o[q])));
        if (epi) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[q] = o[q] * esc[q] + esh[q];
            if (a.erelu) o[q] = fmaxf(o[q], 0.f);
          }
        }
        if (st0 + u * G + lg < nrow && cok) {
          *reinterpret_cast<float4 *>(a.Y + (r0 + rr[u]) * a.N + c0) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
          for (int q = 0; q < 4; ++q) { s1[q] += o[q]; s2[q] += o[q] * o[q]; }
        }
      }
    }
  }
  if (a.partial == nullptr) return;
  // ---- column sums: row groups of the wave, the four waves, one partial per workgroup
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
      s1[q] += __shfl_xor(s1[q], off, 64);
      s2[q] += __shfl_xor(s2[q], off, 64);
    }
    if (lg == 0) { s_stat[0][wave][c0 + q] = s1[q]; s_stat[1][wave][c0 + q] = s2[q]; }
  }
  __syncthreads();
  for (int c = tid; c < a.N; c += 256) {
    float *p = a.partial + (long long)blockIdx.x * 2 * a.N;
    p[c] = (s_stat[0][0][c] + s_stat[0][1][c]) + (s_stat[0][2][c] + s_stat[0][3][c]);
    p[a.N + c] = (s_stat[1][0][c] + s_stat[1][1][c]) + (s_stat[1][2][c] + s_stat[1][3][c]);
  }
}

static int gather_add_rpb(long long rows) { return rows >= 131072 ? 256 : 64; }

extern "C" int s2c_sa_gather_add_blocks(long long rows) {
  const int rpb = gather_add_rpb(rows);
  return (int)((rows + rpb - 1) / rpb);
}

static int gather_add_launch(int b, int n, int m, int ns, int N, float radius, int normalize,
                             const float *xyz, const float *new_xyz, const float *P,
                             const int *idx, const float *W, int ldw, float *Y, float *partial,
                             const float *eg, const float *eb, const float *em, const float *ev,
                             float eeps, int erelu, void *stream) {
  if (b <= 0 || n <= 0 || m <= 0 || ns <= 0 || N <= 0 || N > 256 || (N & 3) || ldw < 3 || !xyz ||
      !new_xyz || !idx || !W || !Y || ((uintptr_t)Y & 15) || (P && ((uintptr_t)P & 15)))
    return fail2("sa_gather_add: sizes / alignment / null pointer");
  GatherAddArgs a;
  a.n = n; a.m = m; a.ns = ns; a.N = N; a.ldw = ldw; a.normalize = normalize;
  a.rows = (long long)b * m * ns;
  a.rpb = gather_add_rpb(a.rows); a.has_p = P != nullptr;
  a.radius = radius; a.xyz = xyz; a.new_xyz = new_xyz; a.P = P; a.W = W; a.idx = idx;
  a.Y = Y; a.partial = partial;
  a.eg = eg; a.eb = eb; a.em = em; a.ev = ev; a.eeps = eeps; a.erelu = erelu;
  const dim3 grid((unsigned)s2c_sa_gather_add_blocks(a.rows));
  hipStream_t st = (hipStream_t)stream;
  if (N <= 64) hipLaunchKernelGGL(sa_gather_add_kernel<16>, grid, dim3(256), 0, st, a);
  else if (N <= 128) hipLaunchKernelGGL(sa_gather_add_kernel<32>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(sa_gather_add_kernel<64>, grid, dim3(256), 0, st, a);
  return check2("sa_gather_add");
}

extern "C" int s2c_sa_gather_add(int b, int n, int m, int ns, int N, float radius, int normalize,
                                 const float *xyz, const float *new_xyz, const float *P,
                                 const int *idx, const float *W, int ldw, float *Y,
                                 float *partial, void *stream) {
  return gather_add_launch(b, n, m, ns, N, radius, normalize, xyz, new_xyz, P, idx, W, ldw, Y,
                           partial, nullptr, nullptr, nullptr, nullptr, 0.f, 0, stream);
}

// Inference: the first layer in point space with the frozen BatchNorm (+ ReLU) behind it in the
// same pass: Y = relu?((P[idx] + W_x rel) * sc + sh) (mean / var: the running statistics).
extern "C" int s2c_sa_gather_add_eval(int b, int n, int m, int ns, int N, float radius,
                                      int normalize, const float *xyz, const float *new_xyz,
                                      const float *P, const int *idx, const float *W, int ldw,
                                      const float *gamma, const float *beta, const float *mean,
                                      const float *var, float eps, int relu, float *Y,
                                      void *stream) {
  if (!mean || !var) return fail2("sa_gather_add_eval: null statistics");
  return gather_add_launch(b, n, m, ns, N, radius, normalize, xyz, new_xyz, P, idx, W, ldw, Y,
                           nullptr, gamma, beta, mean, var, eps, relu, stream);
}

// Feature propagation on point-major rows (PointnetFPModule.forward,
// pointnet2_modules.py:398-410): out[b,i,:] = [ sum_j w[b,i,j] * known[b,idx[b,i,j],:]
// | skip[b,i,:] ] written straight into the (B*n, C2+C1) operand of the FP MLP: no
// channel-major copy, no torch.cat, no transpose.  Same left-to-right arithmetic as
// three_interpolate (interpolate_gpu.cu:87-99).  Wave per output row, lane = channel.
__global__ __launch_bounds__(256) void fp_interp_rows_kernel(
    int n, int m, int C2, int C1, const float *__restrict__ known,
    const int *__restrict__ idx, const float *__restrict__ weight,
    const float *__restrict__ skip, long long skip_rs, long long skip_bs,
    float *__restrict__ out, long long rows) {
  const int lane = threadIdx.x & 63;
  const int ld = C2 + C1;
  for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows;
       r += (long long)gridDim.x * 4) {
    const long long b = r / n, i = r - b * n;
    const int i0 = idx[r * 3], i1 = idx[r * 3 + 1], i2 = idx[r * 3 + 2];
    const float w0 = weight[r * 3], w1 = weight[r * 3 + 1], w2 = weight[r * 3 + 2];
    const float *k0 = known + (b * m + i0) * (long long)C2;
    const float *k1 = known + (b * m + i1) * (long long)C2;
    const float *k2 = known + (b * m + i2) * (long long)C2;
    float *o = out + r * ld;
    for (int c = lane; c < C2; c += 64) o[c] = k0[c] * w0 + k1[c] * w1 + k2[c] * w2;
    if (C1 > 0) {
      const float *sk = skip + b * skip_bs + i * skip_rs;
      for (int c = lane; c < C1; c += 64) o[C2 + c] = sk[c];
    }
  }
}

// backward: d_known[b, idx, :] += w * dOut[b,i,:C2]  (row-coalesced float atomics);
// the skip part of dOut is returned as a view by the caller.
__global__ __launch_bounds__(256) void fp_interp_rows_grad_kernel(
    int n, int m, int C2, int ld, const float *__restrict__ dOut,
    const int *__restrict__ idx, const float *__restrict__ weight,
    float *__restrict__ d_known, long long rows) {
  const int lane = threadIdx.x & 63;
  for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows;
       r += (long long)gridDim.x * 4) {
    const long long b = r / n;
    const float *g = dOut + r * ld;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ij = idx[r * 3 + j];
      const float w = weight[r * 3 + j];
      float *dst = d_known + (b * m + ij) * (long long)C2;
      for (int c = lane; c < C2; c += 64) atomicAdd(dst + c, g[c] * w);
    }
  }
}

extern "C" int s2c_fp_interp_rows(int b, int n, int m, int C2, int C1, const float *known,
                                  const int *idx, const float *weight, const float *skip,
                                  long long skip_row_stride, long long skip_batch_stride,
                                  float *out, void *stream) {
  if (b <= 0 || n <= 0 || m <= 0 || C2 <= 0 || C1 < 0 || !known || !idx || !weight || !out ||
      (C1 > 0 && !skip))
    return fail2("fp_interp_rows: sizes / null pointer");
  const long long rows = (long long)b * n;
  hipLaunchKernelGGL(fp_interp_rows_kernel, dim3(grid1d(rows, 4, 256 * 32)), dim3(256), 0,
                     (hipStream_t)stream, n, m, C2, C1, known, idx, weight, skip,
                     skip_row_stride, skip_batch_stride, out, rows);
  return check2("fp_interp_rows");
}

extern "C" int s2c_fp_interp_rows_grad(int b, int n, int m, int C2, int ld,
                                       const float *dOut, const int *idx,
                                       const float *weight, float *d_known, void *stream) {
  if (b <= 0 || n <= 0 || m <= 0 || C2 <= 0 || ld < C2 || !dOut || !idx || !weight || !d_known)
    return fail2("fp_interp_rows_grad: sizes / null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (zero_async(d_known, sizeof(float) * (size_t)b * m * C2, st) != hipSuccess)
    return fail2("memset");
  const long long rows = (long long)b * n;
  hipLaunchKernelGGL(fp_interp_rows_grad_kernel, dim3(grid1d(rows, 4, 256 * 32)), dim3(256),
                     0, st, n, m, C2, ld, dOut, idx, weight, d_known, rows);
  return check2("fp_interp_rows_grad");
}

// ---------------------------------------------------------------------------
// 2. column statistics of a row-major (M x C) matrix, C % 4 == 0.
// Stage 1: each block reduces a slab of rows to partial [sum | sumsq] (float).
// Stage 2 (bn_finalize): fixed-order double reduction of the partials ->
// mean, biased var; writes scale = gamma*invstd, shift = beta - mean*scale,
// saves mean/invstd for backward and updates the running statistics exactly as
// torch.nn.BatchNorm does in training (momentum, UNBIASED var for running_var;
// pytorch_utils.py:100-120 instantiates nn.BatchNorm2d with defaults
// eps=1e-5, momentum=0.1).  Deterministic (no atomics).
// ---------------------------------------------------------------------------
constexpr int STAT_BLOCK = 256;

// generic 2-quantity column reduction; the functor gives (q1, q2) per element
template <class F>
__device__ __forceinline__ void col_reduce2(F f, long long M, int C,
                                            float *__restrict__ partial,
                                            long long rows_per_block) {
  // thread layout: tx in [0, C/4) handles 4 channels, ty = row lane
  __shared__ float4 s_a[STAT_BLOCK], s_b[STAT_BLOCK];
  const int c4n = C >> 2;
  const int ry = STAT_BLOCK / c4n;  // rows in flight (C/4 <= 256)
  const int tx = threadIdx.x % c4n, ty = threadIdx.x / c4n;
  float4 a = make_float4(0, 0, 0, 0), b2 = make_float4(0, 0, 0, 0);
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(M, r0 + rows_per_block);
  if (ty < ry) {
    long long r = r0 + ty;
    // 4 rows in flight per thread (8 independent 16-byte loads); the sums are
    // still accumulated in row order, so the result does not depend on the unroll
    for (; r + 3LL * ry < r1; r += 4LL * ry) {
      float4 p1[4], p2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) f(r + (long long)u * ry, tx, p1[u], p2[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a.x += p1[u].x; a.y += p1[u].y; a.z += p1[u].z; a.w += p1[u].w;
        b2.x += p2[u].x; b2.y += p2[u].y; b2.z += p2[u].z; b2.w += p2[u].w;
      }
    }
    for (; r < r1; r += ry) {
      float4 q1, q2;
      f(r, tx, q1, q2);
      a.x += q1.x; a.y += q1.y; a.z += q1.z; a.w += q1.w;
      b2.x += q2.x; b2.y += q2.y; b2.z += q2.z; b2.w += q2.w;
    }
  }
  s_a[threadIdx.x] = a;
  s_b[threadIdx.x] = b2;
  __syncthreads();
  if (ty == 0) {
    for (int y = 1; y < ry; ++y) {
      const float4 u = s_a[y * c4n + tx], v = s_b[y * c4n + tx];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b2.x += v.x; b2.y += v.y; b2.z += v.z; b2.w += v.w;
    }
    float *p = partial + (long long)blockIdx.x * 2 * C;
    reinterpret_cast<float4 *>(p)[tx] = a;
    reinterpret_cast<float4 *>(p + C)[tx] = b2;
  }
}

__global__ __launch_bounds__(STAT_BLOCK) void col_stats_kernel(
    const float *__restrict__ Y, long long M, int C, float *__restrict__ partial,
    long long rows_per_block) {
  col_reduce2(
      [&](long long r, int tx, float4 &q1, float4 &q2) {
        const float4 v = reinterpret_cast<const float4 *>(Y + r * C)[tx];
        q1 = v;
        q2 = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
      },
      M, C, partial, rows_per_block);
}

// one wave per channel: lanes stride over the partial blocks (double
// accumulation, fixed order => deterministic), DPP-free shuffle tree to lane 0
__device__ __forceinline__ void wave_sum2(double &a, double &b) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
}

// 256 threads per channel: wave trees, then the 4 wave results in fixed order
constexpr int FIN_BLOCK = 256;
__device__ __forceinline__ void block_sum2(double &a, double &b) {
  __shared__ double s_fin[2 * (FIN_BLOCK / 64)];
  wave_sum2(a, b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_fin[2 * w] = a;
    s_fin[2 * w + 1] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = s_fin[0];
    b = s_fin[1];
    for (int k = 1; k < FIN_BLOCK / 64; ++k) {
      a += s_fin[2 * k];
      b += s_fin[2 * k + 1];
    }
  }
}

// this thread's share of column c of a partial table (nblk rows of [s1 (C) | s2 (C)]): rows
// threadIdx.x, + FIN_BLOCK, ... in that order, eight rows (16 loads) in flight -- one row at a
// time the 4096-8192 rows of an SA1 layer were 16-32 dependent round trips (20 us per launch)
__device__ __forceinline__ void fin_column_sums(const float *__restrict__ partial, int nblk, int C,
                                                int c, double &s1, double &s2) {
  int k = threadIdx.x;
  for (; k + 7 * FIN_BLOCK < nblk; k += 8 * FIN_BLOCK) {
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float *row = partial + (long long)(k + u * FIN_BLOCK) * 2 * C;
      a[u] = row[c];
      b[u] = row[C + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s1 += (double)a[u]; s2 += (double)b[u]; }
  }
  for (; k < nblk; k += FIN_BLOCK) {
    s1 += (double)partial[(long long)k * 2 * C + c];
    s2 += (double)partial[(long long)k * 2 * C + C + c];
  }
}

__global__ __launch_bounds__(FIN_BLOCK) void bn_finalize_kernel(
    const float *__restrict__ partial, int nblk, int C, long long M, float eps,
    float momentum, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ running_mean,
    float *__restrict__ running_var, float *__restrict__ scale,
    float *__restrict__ shift, float *__restrict__ save_mean,
    float *__restrict__ save_invstd, long long *__restrict__ num_batches_tracked,
    const float *__restrict__ W, int ldw, int Cin, float *__restrict__ Wt) {
  const int c = blockIdx.x;
  // Wt != nullptr: the layer's weight (C x Cin) leaves transposed as well (column c of Wt = row c of
  // W): the backward's input-gradient GEMMs read W^T row-major -- a copy kernel per layer otherwise
  if (Wt != nullptr)
    for (int k = threadIdx.x; k < Cin; k += FIN_BLOCK) Wt[(size_t)k * C + c] = W[(size_t)c * ldw + k];
  double s1 = 0.0, s2 = 0.0;
  fin_column_sums(partial, nblk, C, c, s1, s2);
  block_sum2(s1, s2);
  if (threadIdx.x != 0) return;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;   // nn.BatchNorm bookkeeping
  const double mean = s1 / (double)M;
  double var = s2 / (double)M - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
  const float sc = g * invstd;
  scale[c] = sc;
  shift[c] = bt - (float)mean * sc;
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  if (running_mean) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// eval mode: scale/shift from the running statistics
__global__ void bn_eval_coeffs_kernel(int C, float eps,
                                      const float *__restrict__ gamma,
                                      const float *__restrict__ beta,
                                      const float *__restrict__ running_mean,
                                      const float *__restrict__ running_var,
                                      float *__restrict__ scale,
                                      float *__restrict__ shift,
                                      float *__restrict__ save_mean,
                                      float *__restrict__ save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(running_var[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.0f) * invstd;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.0f) - running_mean[c] * sc;
  save_mean[c] = running_mean[c];
  save_invstd[c] = invstd;
}

static int stat_blocks(long long M, long long *rows_per_block) {
  long long nb = (M + 63) / 64;  // >= 64 rows per block: small inputs still fill the chip
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  *rows_per_block = (M + nb - 1) / nb;
  return (int)((M + *rows_per_block - 1) / *rows_per_block);
}

extern "C" int s2c_bn_stat_blocks(long long M) {
  long long rpb;
  return stat_blocks(M, &rpb);
}

// training-mode BN statistics of Y (M x C): partial must hold
// s2c_bn_stat_blocks(M)*2*C floats.
extern "C" int s2c_bn_train_stats(long long M, int C, const float *Y,
                                  float *partial, float eps, float momentum,
                                  const float *gamma, const float *beta,
                                  float *running_mean, float *running_var,
                                  float *scale, float *shift, float *save_mean,
                                  float *save_invstd, long long *num_batches_tracked,
                                  void *stream) {
  if (M <= 0 || C <= 0 || (C & 3) || C > 1024) return fail2("bn_train_stats: C%4==0, C<=1024, M>0");
  if (!Y || !partial || !scale || !shift || !save_mean || !save_invstd)
    return fail2("bn_train_stats: null pointer");
  hipStream_t st = (hipStream_t)stream;
  long long rpb;
  const int nb = stat_blocks(M, &rpb);
  hipLaunchKernelGGL(col_stats_kernel, dim3(nb), dim3(STAT_BLOCK), 0, st, Y, M, C,
                     partial, rpb);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(FIN_BLOCK), 0, st,
                     partial, nb, C, M, eps, momentum, gamma, beta, running_mean,
                     running_var, scale, shift, save_mean, save_invstd, num_batches_tracked,
                     (const float *)nullptr, 0, 0, (float *)nullptr);
  return check2("bn_train_stats");
}

// finalize from partials produced elsewhere (the MFMA GEMM epilogue,
// csrc/s2c_gemm.hip): same layout [nblk][sum(C) | sumsq(C)]
extern "C" int s2c_bn_finalize_partials(int nblk, long long M, int C,
                                        const float *partial, float eps,
                                        float momentum, const float *gamma,
                                        const float *beta, float *running_mean,
                                        float *running_var, float *scale,
                                        float *shift, float *save_mean,
                                        float *save_invstd,
                                        long long *num_batches_tracked, void *stream) {
  if (nblk <= 0 || M <= 0 || C <= 0) return fail2("bn_finalize_partials sizes");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(FIN_BLOCK), 0, (hipStream_t)stream,
                     partial, nblk, C, M, eps, momentum, gamma, beta, running_mean,
                     running_var, scale, shift, save_mean, save_invstd, num_batches_tracked,
                     (const float *)nullptr, 0, 0, (float *)nullptr);
  return check2("bn_finalize_partials");
}

// ... and Wt (Cin x C, contiguous) = W^T of the layer's weight W (C x Cin, row stride ldw) out of the
// same launch: what the backward's input-gradient GEMMs of the tall layers read
extern "C" int s2c_bn_finalize_partials_wt(int nblk, long long M, int C, const float *partial,
                                           float eps, float momentum, const float *gamma,
                                           const float *beta, float *running_mean,
                                           float *running_var, float *scale, float *shift,
                                           float *save_mean, float *save_invstd,
                                           long long *num_batches_tracked, const float *W, int ldw,
                                           int Cin, float *Wt, void *stream) {
  if (nblk <= 0 || M <= 0 || C <= 0 || !W || !Wt || Cin <= 0 || ldw < Cin)
    return fail2("bn_finalize_partials_wt sizes");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(FIN_BLOCK), 0, (hipStream_t)stream,
                     partial, nblk, C, M, eps, momentum, gamma, beta, running_mean,
                     running_var, scale, shift, save_mean, save_invstd, num_batches_tracked, W, ldw,
                     Cin, Wt);
  return check2("bn_finalize_partials_wt");
}

extern "C" int s2c_bn_eval_coeffs(int C, float eps, const float *gamma,
                                  const float *beta, const float *running_mean,
                                  const float *running_var, float *scale,
                                  float *shift, float *save_mean,
                                  float *save_invstd, void *stream) {
  if (C <= 0) return fail2("bn_eval_coeffs: C");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0,
                     (hipStream_t)stream, C, eps, gamma, beta, running_mean,
                     running_var, scale, shift, save_mean, save_invstd);
  return check2("bn_eval_coeffs");
}

// ---------------------------------------------------------------------------
// 3. A = relu(Y*scale + shift)   (M x C, float4 over channels)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_relu_kernel(
    const float *__restrict__ Y, const float *__restrict__ scale,
    const float *__restrict__ shift, float *__restrict__ A, long long total4,
    int c4n, int relu) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4;
       e += (long long)gridDim.x * 256) {
    const int tx = (int)(e % c4n);
    const float4 y = reinterpret_cast<const float4 *>(Y)[e];
    const float4 sc = reinterpret_cast<const float4 *>(scale)[tx];
    const float4 sh = reinterpret_cast<const float4 *>(shift)[tx];
    float4 a;
    a.x = y.x * sc.x + sh.x; a.y = y.y * sc.y + sh.y;
    a.z = y.z * sc.z + sh.z; a.w = y.w * sc.w + sh.w;
    if (relu) {
      a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f);
      a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
    }
    reinterpret_cast<float4 *>(A)[e] = a;
  }
}

extern "C" int s2c_bn_relu(long long M, int C, const float *Y, const float *scale,
                           const float *shift, float *A, int relu, void *stream) {
  if (M < 0 || C <= 0 || (C & 3)) return fail2("bn_relu: C%4==0");
  if (M == 0) return 0;
  const long long total4 = M * (C >> 2);
  hipLaunchKernelGGL(bn_relu_kernel, dim3(grid1d(total4, 256)), dim3(256), 0,
                     (hipStream_t)stream, Y, scale, shift, A, total4, C >> 2, relu);
  return check2("bn_relu");
}

// ---------------------------------------------------------------------------
// 4. out[j, c] = max_k relu(Y[(j,k), c]*scale + shift), arg[j, c] = first k of
// the maximum (F.max_pool2d over nsample, pointnet2_modules.py:255-257).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_relu_max_kernel(
    const float *__restrict__ Y, const float *__restrict__ scale,
    const float *__restrict__ shift, float *__restrict__ out,
    int *__restrict__ arg, float *__restrict__ ymax, long long J, int ns, int c4n) {
  const long long total = J * c4n;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const int tx = (int)(e % c4n);
    const long long j = e / c4n;
    const float4 sc = reinterpret_cast<const float4 *>(scale)[tx];
    const float4 sh = reinterpret_cast<const float4 *>(shift)[tx];
    float4 best = make_float4(-1.f, -1.f, -1.f, -1.f);
    float4 ybest = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 bi = make_int4(0, 0, 0, 0);
    const float4 *src = reinterpret_cast<const float4 *>(Y) + (j * ns) * c4n + tx;
    for (int k = 0; k < ns; ++k) {
      const float4 y = src[(long long)k * c4n];
      const float ax = fmaxf(y.x * sc.x + sh.x, 0.f), ay = fmaxf(y.y * sc.y + sh.y, 0.f);
      const float az = fmaxf(y.z * sc.z + sh.z, 0.f), aw = fmaxf(y.w * sc.w + sh.w, 0.f);
      if (ax > best.x) { best.x = ax; bi.x = k; ybest.x = y.x; }
      if (ay > best.y) { best.y = ay; bi.y = k; ybest.y = y.y; }
      if (az > best.z) { best.z = az; bi.z = k; ybest.z = y.z; }
      if (aw > best.w) { best.w = aw; bi.w = k; ybest.w = y.w; }
    }
    reinterpret_cast<float4 *>(out)[e] = best;
    reinterpret_cast<int4 *>(arg)[e] = bi;
    // raw pre-BN value at the arg-max: lets the backward statistics run on
    // (J x C) data instead of gathering from the (J*ns x C) tensor
    if (ymax != nullptr) reinterpret_cast<float4 *>(ymax)[e] = ybest;
  }
}

extern "C" int s2c_bn_relu_max(long long J, int ns, int C, const float *Y,
                               const float *scale, const float *shift, float *out,
                               int *arg, float *ymax, void *stream) {
  if (J < 0 || ns <= 0 || C <= 0 || (C & 3)) return fail2("bn_relu_max: C%4==0");
  if (J == 0) return 0;
  hipLaunchKernelGGL(bn_relu_max_kernel, dim3(grid1d(J * (C >> 2), 256)), dim3(256),
                     0, (hipStream_t)stream, Y, scale, shift, out, arg, ymax, J, ns, C >> 2);
  return check2("bn_relu_max");
}

// 4b. The same outputs from the per-centre extremum of Y that s2c_rows_gemm_pool_raw leaves
// (relu(y * scale + shift) is monotone in y: max y for scale >= 0, min y otherwise; the GEMM
// already chose by the sign of gamma): out = relu(ext * scale + shift); `ext` IS bn_relu_max's
// ymax and `aext` its arg (the first row of the extremum; where "first maximum AFTER the ReLU"
// differs -- every row clamped to 0 -- the routed gradient is masked to zero anyway).
__global__ __launch_bounds__(256) void pool_select_kernel(
    const float *__restrict__ ext, const float *__restrict__ scale,
    const float *__restrict__ shift, float *__restrict__ out, long long total, int C) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const int c = (int)(e % C);
    out[e] = fmaxf(ext[e] * scale[c] + shift[c], 0.f);
  }
}

extern "C" int s2c_pool_select(long long J, int C, const float *ext, const float *scale,
                               const float *shift, float *out, void *stream) {
  if (J < 0 || C <= 0 || !ext || !scale || !shift || !out)
    return fail2("pool_select: sizes / null pointer");
  if (J == 0) return 0;
  hipLaunchKernelGGL(pool_select_kernel, dim3(grid1d(J * C, 256)), dim3(256), 0,
                     (hipStream_t)stream, ext, scale, shift, out, J * C, C);
  return check2("pool_select");
}

// ---------------------------------------------------------------------------
// 5. backward of BN(+ReLU):  dz = dA * [z > 0]   (z = Y*scale + shift)
//    s1 = sum dz, s2 = sum dz * xhat            (xhat = (Y - mean) * invstd)
//    dY = gamma*invstd * (dz - s1/M - xhat*s2/M)     (training statistics)
//    dY = gamma*invstd * dz                           (frozen / eval statistics)
//    dgamma = s2, dbeta = s1.
// Two passes: (a) column reduction -> partials -> finalize (double, fixed
// order); (b) apply.  The max-pool variant reads the sparse upstream gradient
// dOut[j,c] routed to row arg[j,c].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(STAT_BLOCK) void bn_bwd_stats_kernel(
    const float *__restrict__ dA, const float *__restrict__ Y,
    const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ mean, const float *__restrict__ invstd,
    long long M, int C, int relu, float *__restrict__ partial,
    long long rows_per_block) {
  col_reduce2(
      [&](long long r, int tx, float4 &q1, float4 &q2) {
        const float4 y = reinterpret_cast<const float4 *>(Y + r * C)[tx];
        float4 g = reinterpret_cast<const float4 *>(dA + r * C)[tx];
        const float4 sc = reinterpret_cast<const float4 *>(scale)[tx];
        const float4 sh = reinterpret_cast<const float4 *>(shift)[tx];
        const float4 mu = reinterpret_cast<const float4 *>(mean)[tx];
        const float4 is = reinterpret_cast<const float4 *>(invstd)[tx];
        if (relu) {
          if (!(y.x * sc.x + sh.x > 0.f)) g.x = 0.f;
          if (!(y.y * sc.y + sh.y > 0.f)) g.y = 0.f;
          if (!(y.z * sc.z + sh.z > 0.f)) g.z = 0.f;
          if (!(y.w * sc.w + sh.w > 0.f)) g.w = 0.f;
        }
        q1 = g;
        q2 = make_float4(g.x * ((y.x - mu.x) * is.x), g.y * ((y.y - mu.y) * is.y),
                         g.z * ((y.z - mu.z) * is.z), g.w * ((y.w - mu.w) * is.w));
      },
      M, C, partial, rows_per_block);
}

// sums -> dgamma (=s2), dbeta (=s1); also the per-channel coefficients used by
// the apply pass: k0 = gamma*invstd, k1 = s1/M, k2 = s2/M (0 when frozen)
__global__ __launch_bounds__(FIN_BLOCK) void bn_bwd_finalize_kernel(
    const float *__restrict__ partial, int nblk, int C, long long M, int frozen,
    const float *__restrict__ gamma, const float *__restrict__ invstd,
    float *__restrict__ dgamma, float *__restrict__ dbeta,
    float *__restrict__ coef) {
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  fin_column_sums(partial, nblk, C, c, s1, s2);
  block_sum2(s1, s2);
  if (threadIdx.x != 0) return;
  if (dbeta) dbeta[c] = (float)s1;
  if (dgamma) dgamma[c] = (float)s2;
  coef[c] = (gamma ? gamma[c] : 1.0f) * invstd[c];
  coef[C + c] = frozen ? 0.0f : (float)(s1 / (double)M);
  coef[2 * C + c] = frozen ? 0.0f : (float)(s2 / (double)M);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float *__restrict__ dA, const float *__restrict__ Y,
    const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ coef, float *__restrict__ dY, long long total4,
    int c4n, int C, int relu) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4;
       e += (long long)gridDim.x * 256) {
    const int tx = (int)(e % c4n);
    const float4 y = reinterpret_cast<const float4 *>(Y)[e];
    float4 g = reinterpret_cast<const float4 *>(dA)[e];
    const float4 sc = reinterpret_cast<const float4 *>(scale)[tx];
    const float4 sh = reinterpret_cast<const float4 *>(shift)[tx];
    const float4 mu = reinterpret_cast<const float4 *>(mean)[tx];
    const float4 is = reinterpret_cast<const float4 *>(invstd)[tx];
    const float4 k0 = reinterpret_cast<const float4 *>(coef)[tx];
    const float4 k1 = reinterpret_cast<const float4 *>(coef + C)[tx];
    const float4 k2 = reinterpret_cast<const float4 *>(coef + 2 * C)[tx];
    if (relu) {
      if (!(y.x * sc.x + sh.x > 0.f)) g.x = 0.f;
      if (!(y.y * sc.y + sh.y > 0.f)) g.y = 0.f;
      if (!(y.z * sc.z + sh.z > 0.f)) g.z = 0.f;
      if (!(y.w * sc.w + sh.w > 0.f)) g.w = 0.f;
    }
    float4 o;
    o.x = k0.x * (g.x - k1.x - ((y.x - mu.x) * is.x) * k2.x);
    o.y = k0.y * (g.y - k1.y - ((y.y - mu.y) * is.y) * k2.y);
    o.z = k0.z * (g.z - k1.z - ((y.z - mu.z) * is.z) * k2.z);
    o.w = k0.w * (g.w - k1.w - ((y.w - mu.w) * is.w) * k2.w);
    reinterpret_cast<float4 *>(dY)[e] = o;
  }
}

// dA / partial / coef semantics as documented above.  `coef` = 3*C floats.
extern "C" int s2c_bn_relu_bwd(long long M, int C, const float *dA, const float *Y,
                               const float *scale, const float *shift,
                               const float *mean, const float *invstd,
                               const float *gamma, int relu, int frozen,
                               float *partial, float *coef, float *dgamma,
                               float *dbeta, float *dY, void *stream) {
  if (M <= 0 || C <= 0 || (C & 3) || C > 1024) return fail2("bn_relu_bwd: C%4==0");
  hipStream_t st = (hipStream_t)stream;
  long long rpb;
  const int nb = stat_blocks(M, &rpb);
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(nb), dim3(STAT_BLOCK), 0, st, dA, Y,
                     scale, shift, mean, invstd, M, C, relu, partial, rpb);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(FIN_BLOCK), 0, st,
                     partial, nb, C, M, frozen, gamma, invstd, dgamma, dbeta, coef);
  const long long total4 = M * (C >> 2);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid1d(total4, 256)), dim3(256), 0,
                     st, dA, Y, scale, shift, mean, invstd, coef, dY, total4, C >> 2,
                     C, relu);
  return check2("bn_relu_bwd");
}

// The statistics half only (stats + finalize: dgamma, dbeta, coef); the apply half is then
// the operand prologue of s2c_bn_bwd_gemm (csrc/s2c_gemm.hip).
extern "C" int s2c_bn_relu_bwd_stats(long long M, int C, const float *dA, const float *Y,
                                     const float *scale, const float *shift,
                                     const float *mean, const float *invstd,
                                     const float *gamma, int relu, int frozen,
                                     float *partial, float *coef, float *dgamma,
                                     float *dbeta, void *stream) {
  if (M <= 0 || C <= 0 || (C & 3) || C > 1024) return fail2("bn_relu_bwd_stats: C%4==0");
  hipStream_t st = (hipStream_t)stream;
  long long rpb;
  const int nb = stat_blocks(M, &rpb);
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(nb), dim3(STAT_BLOCK), 0, st, dA, Y,
                     scale, shift, mean, invstd, M, C, relu, partial, rpb);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(FIN_BLOCK), 0, st,
                     partial, nb, C, M, frozen, gamma, invstd, dgamma, dbeta, coef);
  return check2("bn_relu_bwd_stats");
}

// The two halves on their own, for a caller whose column sums already exist (formed in the
// epilogue of the GEMM that produced dA: s2c_bn_bwd_gemm_next_stats): `partial` = nblk rows of
// [s1 (C) | s2 (C)].
extern "C" int s2c_bn_bwd_finalize_partials(int nblk, long long M, int C, const float *partial,
                                            int frozen, const float *gamma, const float *invstd,
                                            float *coef, float *dgamma, float *dbeta,
                                            void *stream) {
  if (nblk <= 0 || M <= 0 || C <= 0 || !partial || !invstd || !coef)
    return fail2("bn_bwd_finalize_partials: sizes / null pointer");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(FIN_BLOCK), 0, (hipStream_t)stream,
                     partial, nblk, C, M, frozen, gamma, invstd, dgamma, dbeta, coef);
  return check2("bn_bwd_finalize_partials");
}

extern "C" int s2c_bn_relu_bwd_apply(long long M, int C, const float *dA, const float *Y,
                                     const float *scale, const float *shift, const float *mean,
                                     const float *invstd, const float *coef, int relu, float *dY,
                                     void *stream) {
  if (M <= 0 || C <= 0 || (C & 3) || !dA || !Y || !coef || !dY)
    return fail2("bn_relu_bwd_apply: C%4==0 / null pointer");
  const long long total4 = M * (C >> 2);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid1d(total4, 256)), dim3(256), 0,
                     (hipStream_t)stream, dA, Y, scale, shift, mean, invstd, coef, dY, total4,
                     C >> 2, C, relu);
  return check2("bn_relu_bwd_apply");
}

// max-pool variant: upstream dOut (J x C), arg (J x C); rows M = J*ns.
__global__ __launch_bounds__(STAT_BLOCK) void pool_bwd_stats_kernel(
    const float *__restrict__ dOut, const float *__restrict__ ymax,
    const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ mean, const float *__restrict__ invstd, long long J,
    int C, float *__restrict__ partial, long long rows_per_block) {
  col_reduce2(
      [&](long long j, int tx, float4 &q1, float4 &q2) {
        float4 g = reinterpret_cast<const float4 *>(dOut + j * C)[tx];
        const float4 y = reinterpret_cast<const float4 *>(ymax + j * C)[tx];
        const float4 sc = reinterpret_cast<const float4 *>(scale)[tx];
        const float4 sh = reinterpret_cast<const float4 *>(shift)[tx];
        const float4 mu = reinterpret_cast<const float4 *>(mean)[tx];
        const float4 is = reinterpret_cast<const float4 *>(invstd)[tx];
        if (!(y.x * sc.x + sh.x > 0.f)) g.x = 0.f;
        if (!(y.y * sc.y + sh.y > 0.f)) g.y = 0.f;
        if (!(y.z * sc.z + sh.z > 0.f)) g.z = 0.f;
        if (!(y.w * sc.w + sh.w > 0.f)) g.w = 0.f;
        q1 = g;
        q2 = make_float4(g.x * ((y.x - mu.x) * is.x), g.y * ((y.y - mu.y) * is.y),
                         g.z * ((y.z - mu.z) * is.z), g.w * ((y.w - mu.w) * is.w));
      },
      J, C, partial, rows_per_block);
}

__global__ __launch_bounds__(256) void pool_bwd_apply_kernel(
    const float *__restrict__ dOut, const int *__restrict__ arg,
    const float *__restrict__ Y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ coef,
    float *__restrict__ dY, long long J, int ns, int c4n, int C) {
  // thread = (centre j, channel quad): per-centre operands are read once, the ns
  // rows of the centre are streamed (same access shape as the forward max kernel)
  const long long total = J * c4n;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const int tx = (int)(e % c4n);
    const long long j = e / c4n;
    float4 g = reinterpret_cast<const float4 *>(dOut)[e];
    const int4 a = reinterpret_cast<const int4 *>(arg)[e];
    const float4 sc = reinterpret_cast<const float4 *>(scale)[tx];
    const float4 sh = reinterpret_cast<const float4 *>(shift)[tx];
    const float4 mu = reinterpret_cast<const float4 *>(mean)[tx];
    const float4 is = reinterpret_cast<const float4 *>(invstd)[tx];
    const float4 k0 = reinterpret_cast<const float4 *>(coef)[tx];
    const float4 k1 = reinterpret_cast<const float4 *>(coef + C)[tx];
    const float4 k2 = reinterpret_cast<const float4 *>(coef + 2 * C)[tx];
    const float4 *src = reinterpret_cast<const float4 *>(Y) + (j * ns) * c4n + tx;
    float4 *dst = reinterpret_cast<float4 *>(dY) + (j * ns) * c4n + tx;
    for (int k = 0; k < ns; ++k) {
      const float4 y = src[(long long)k * c4n];
      const float gx = (a.x == k && (y.x * sc.x + sh.x > 0.f)) ? g.x : 0.f;
      const float gy = (a.y == k && (y.y * sc.y + sh.y > 0.f)) ? g.y : 0.f;
      const float gz = (a.z == k && (y.z * sc.z + sh.z > 0.f)) ? g.z : 0.f;
      const float gw = (a.w == k && (y.w * sc.w + sh.w > 0.f)) ? g.w : 0.f;
      float4 o;
      o.x = k0.x * (gx - k1.x - ((y.x - mu.x) * is.x) * k2.x);
      o.y = k0.y * (gy - k1.y - ((y.y - mu.y) * is.y) * k2.y);
      o.z = k0.z * (gz - k1.z - ((y.z - mu.z) * is.z) * k2.z);
      o.w = k0.w * (gw - k1.w - ((y.w - mu.w) * is.w) * k2.w);
      dst[(long long)k * c4n] = o;
    }
  }
}

extern "C" int s2c_bn_relu_max_bwd(long long J, int ns, int C, const float *dOut,
                                   const int *arg, const float *ymax, const float *Y,
                                   const float *scale, const float *shift,
                                   const float *mean, const float *invstd,
                                   const float *gamma, int frozen, float *partial,
                                   float *coef, float *dgamma, float *dbeta,
                                   float *dY, void *stream) {
  if (J <= 0 || ns <= 0 || C <= 0 || (C & 3) || C > 1024)
    return fail2("bn_relu_max_bwd: C%4==0");
  hipStream_t st = (hipStream_t)stream;
  long long rpb;
  const int nb = stat_blocks(J, &rpb);
  const long long M = J * ns;
  hipLaunchKernelGGL(pool_bwd_stats_kernel, dim3(nb), dim3(STAT_BLOCK), 0, st, dOut,
                     ymax, scale, shift, mean, invstd, J, C, partial, rpb);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(FIN_BLOCK), 0, st,
                     partial, nb, C, M, frozen, gamma, invstd, dgamma, dbeta, coef);
  hipLaunchKernelGGL(pool_bwd_apply_kernel, dim3(grid1d(J * (C >> 2), 256)), dim3(256),
                     0, st, dOut, arg, Y, scale, shift, mean, invstd, coef, dY, J, ns,
                     C >> 2, C);
  return check2("bn_relu_max_bwd");
}

// The statistics half of s2c_bn_relu_max_bwd only (pooled data: J x C): dgamma, dbeta, coef.
extern "C" int s2c_bn_relu_max_bwd_stats(long long J, int ns, int C, const float *dOut,
                                         const float *ymax, const float *scale,
                                         const float *shift, const float *mean,
                                         const float *invstd, const float *gamma, int frozen,
                                         float *partial, float *coef, float *dgamma,
                                         float *dbeta, void *stream) {
  if (J <= 0 || ns <= 0 || C <= 0 || (C & 3) || C > 1024)
    return fail2("bn_relu_max_bwd_stats: C%4==0");
  hipStream_t st = (hipStream_t)stream;
  long long rpb;
  const int nb = stat_blocks(J, &rpb);
  hipLaunchKernelGGL(pool_bwd_stats_kernel, dim3(nb), dim3(STAT_BLOCK), 0, st, dOut,
                     ymax, scale, shift, mean, invstd, J, C, partial, rpb);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(FIN_BLOCK), 0, st,
                     partial, nb, C, J * ns, frozen, gamma, invstd, dgamma, dbeta, coef);
  return check2("bn_relu_max_bwd_stats");
}

// dk[j,c] = k0[c] * (ymax[j,c] * scale[c] + shift[c] > 0 ? dOut[j,c] : 0): the routed pooled
// gradient times gamma * invstd (the one nonzero of k0 * dz per centre and channel).
__global__ __launch_bounds__(256) void pool_bwd_dk_kernel(
    const float *__restrict__ dOut, const float *__restrict__ ymax,
    const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ coef, const int *__restrict__ arg, float *__restrict__ dk,
    short *__restrict__ arg16, long long total, int C) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const float g = (ymax[e] * scale[c] + shift[c] > 0.f) ? dOut[e] : 0.f;
    dk[e] = coef[c] * g;
    if (arg16 != nullptr) arg16[e] = (short)arg[e];
  }
}

// SP[c, k] = sum_j dk[j,c] * A[j * ns + arg[j,c], k]   (C3 x K): the routed part of the pooled
// layer's weight gradient, and colsum[k] = sum_r A[r, k] -- both from ONE coalesced pass over
// the layer's INPUT activation A (M x K): a workgroup stages the ns x K rows of a centre in
// LDS, thread = (column k, group of C3 / G channels) adds dk * T[arg][k] for its channels into
// registers, group 0 also sums the column.  Partials per workgroup (summed by the caller:
// kernel-boundary reduction): partial[blk][C3 * K | K].
constexpr int SP_CG = 32;                    // channels per thread (registers)
constexpr int SP_RING = 3;                   // tiles in flight per workgroup (LDS-DMA ring)

__device__ __forceinline__ void sp_glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Tiles arrive by LDS-DMA into a ring of SP_RING slots (tile | dk row | arg row): with the
// register prefetch of the first version a workgroup had one 16 KB tile in flight and the
// pass ran at 1.2-1.7 TB/s; the ring keeps (SP_RING - 1) tiles per workgroup outstanding with
// no VGPR cost.  Every wave issues the same number of DMA instructions per tile (TI + 1), so
// one `s_waitcnt vmcnt` immediate serves all of them.
__global__ __launch_bounds__(256) void pool_bwd_sp_kernel(
    long long J, int ns, int C3, int K, const float *__restrict__ A, const int *__restrict__ arg,
    const float *__restrict__ dk, float *__restrict__ partial, const float *__restrict__ pscale,
    const float *__restrict__ pshift, int prelu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = tid % K, grp = tid / K;                  // 256 / K channel groups
  // pscale != nullptr: the tile holds the previous layer's pre-activation; this thread's column k of
  // the layer's input is relu?(x pscale[k] + pshift[k])
  const bool pro = pscale != nullptr;
  const float psc = pro ? pscale[k] : 1.f, psh = pro ? pshift[k] : 0.f;
  auto act = [&](float x) {
    if (!pro) return x;
    x = x * psc + psh;
    return prelu ? fmaxf(x, 0.f) : x;
  };
  const int tile_bytes = ns * K * 4;                     // multiple of 4096
  const int slot_bytes = tile_bytes + 1024;              // + dk[<=128] | arg[<=128]
  const int TI = tile_bytes / 4096;                      // DMA instructions per wave and tile
  const unsigned base = (unsigned)(size_t)sp_smem;
  float acc[SP_CG];
#pragma unroll
  for (int i = 0; i < SP_CG; ++i) acc[i] = 0.f;
  float csum = 0.f;
  auto issue = [&](long long j, int slot) {
    const char *src = reinterpret_cast<const char *>(A + j * ns * (long long)K);
    const unsigned dst = base + slot * slot_bytes;
    for (int i = 0; i < TI; ++i)
      sp_glds16(src + (wave * TI + i) * 1024 + lane * 16, dst + (wave * TI + i) * 1024);
    // lanes 0-31: dk row, lanes 32-63: arg row (every wave issues it: uniform counts; same data)
    const int l = lane & 31;
    const char *aux = lane < 32 ? reinterpret_cast<const char *>(dk + j * C3)
                                : reinterpret_cast<const char *>(arg + j * C3);
    const int off = l * 16 < C3 * 4 ? l * 16 : 0;        // C3 < 128: re-read the head (ignored)
    sp_glds16(aux + off, dst + tile_bytes);
  };
  const long long stride = gridDim.x;
  long long jn = blockIdx.x;                             // next tile to request
  int islot = 0;
  for (int d = 0; d < SP_RING - 1; ++d)
    if (jn < J) { issue(jn, islot); jn += stride; islot = islot + 1 == SP_RING ? 0 : islot + 1; }
  int slot = 0;
  for (long long j = blockIdx.x; j < J; j += stride) {
    if (jn < J) {
      issue(jn, islot); jn += stride; islot = islot + 1 == SP_RING ? 0 : islot + 1;
      // newer than tile j: SP_RING - 1 tiles of (TI + 1) instructions
      if (TI == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (TI == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (TI == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                        // every wave's pieces of tile j landed
    const float *s_t = reinterpret_cast<const float *>(sp_smem + slot * slot_bytes);
    const float *s_dk = s_t + ns * K;
    const int *s_arg = reinterpret_cast<const int *>(s_dk + 128);
    if (grp == 0)
      for (int rr = 0; rr < ns; ++rr) csum += act(s_t[rr * K + k]);
    const int c0 = grp * SP_CG;
#pragma unroll
    for (int i = 0; i < SP_CG; ++i) {
      const int c = c0 + i;
      if (c < C3) acc[i] += s_dk[c] * act(s_t[s_arg[c] * K + k]);
    }
    __builtin_amdgcn_s_barrier();                        // slot free for the request after next
    slot = slot + 1 == SP_RING ? 0 : slot + 1;
  }
  float *dst = partial + (size_t)blockIdx.x * ((size_t)C3 * K + K);
#pragma unroll
  for (int i = 0; i < SP_CG; ++i) {
    const int c = grp * SP_CG + i;
    if (c < C3) dst[(size_t)c * K + k] = acc[i];
  }
  if (grp == 0) dst[(size_t)C3 * K + k] = csum;
}

// The small matrices of the pooled-layer algebra in two launches (they were ~25 tiny torch
// kernels): with g = k0 k2 invstd, e = g mean - k0 k1 per channel (coef = k0 | k1 | k2)
//   prep:  Wcat (K x (K + C3)) = [ -G^T | W^T ],  G = W^T diag(g) W;  cvec (K) = e W;  ge = g | e
//   final: dW (C3 x K) = sum_blk SP_blk - diag(g) W Gram + e (x) sum_blk colsum_blk
// float64 accumulation, fixed order.
__global__ __launch_bounds__(256) void pool_bwd_prep_kernel(
    int C3, int K, const float *__restrict__ coef, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ W, float *__restrict__ Wcat,
    float *__restrict__ cvec, float *__restrict__ ge) {
  __shared__ double s_g[256], s_e[256];
  __shared__ double s_part[4][64], s_cv[4][64];
  const int k = blockIdx.x;                     // row of Wcat
  const int ld = K + C3;
  for (int c = threadIdx.x; c < C3; c += 256) {
    const double g = (double)coef[c] * (double)coef[2 * C3 + c] * (double)invstd[c];
    s_g[c] = g;
    s_e[c] = g * (double)mean[c] - (double)coef[c] * (double)coef[C3 + c];
  }
  __syncthreads();
  // thread = (column kp of a 64-wide pass, quarter of the channels); fixed-order combine
  const int part = threadIdx.x >> 6, kl = threadIdx.x & 63;
  const int cq = (C3 + 3) / 4;
  for (int k0 = 0; k0 < K; k0 += 64) {
    const int kp = k0 + kl;
    double s = 0.0, cv = 0.0;
    if (kp < K) {
      const int c1 = min(C3, (part + 1) * cq);
#pragma unroll 4
      for (int c = part * cq; c < c1; ++c) {
        const double w = (double)W[c * K + kp];
        s += s_g[c] * w * (double)W[c * K + k];
        cv += s_e[c] * w;
      }
    }
    s_part[part][kl] = s;
    s_cv[part][kl] = cv;
    __syncthreads();
    if (part == 0 && kp < K) {
      const double st = (s_part[0][kl] + s_part[1][kl]) + (s_part[2][kl] + s_part[3][kl]);
      Wcat[(size_t)k * ld + kp] = (float)(-st);           // -G^T[k][kp] = -G[kp][k] (symmetric)
      if (k == 0) cvec[kp] = (float)((s_cv[0][kl] + s_cv[1][kl]) + (s_cv[2][kl] + s_cv[3][kl]));
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < C3; c += 256) {
    Wcat[(size_t)k * ld + K + c] = W[c * K + k];
    if (k == 0) { ge[c] = (float)s_g[c]; ge[C3 + c] = (float)s_e[c]; }
  }
}

// partial_sum = the SP / colsum partials already summed over the workgroups (C3 K + K floats)
__global__ __launch_bounds__(256) void pool_bwd_final_kernel(
    int C3, int K, const float *__restrict__ partial_sum, const float *__restrict__ gram,
    const float *__restrict__ W, const float *__restrict__ coef, const float *__restrict__ mean,
    const float *__restrict__ invstd, float *__restrict__ dW) {
  const int c = blockIdx.x;
  const double g = (double)coef[c] * (double)coef[2 * C3 + c] * (double)invstd[c];
  const double e = g * (double)mean[c] - (double)coef[c] * (double)coef[C3 + c];
  for (int k = threadIdx.x; k < K; k += 256) {
    double wg = 0.0;
#pragma unroll 8
    for (int kk = 0; kk < K; ++kk) wg += (double)W[c * K + kk] * (double)gram[kk * K + k];
    dW[(size_t)c * K + k] = (float)((double)partial_sum[(size_t)c * K + k] - g * wg +
                                    e * (double)partial_sum[(size_t)C3 * K + k]);
  }
}

extern "C" int s2c_pool_bwd_prep(int C3, int K, const float *coef, const float *mean,
                                 const float *invstd, const float *W, float *Wcat, float *cvec,
                                 float *ge, void *stream) {
  if (C3 <= 0 || C3 > 256 || K <= 0 || !coef || !mean || !invstd || !W || !Wcat || !cvec || !ge)
    return fail2("pool_bwd_prep: sizes / null pointer");
  hipLaunchKernelGGL(pool_bwd_prep_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, C3, K, coef,
                     mean, invstd, W, Wcat, cvec, ge);
  return check2("pool_bwd_prep");
}

extern "C" int s2c_pool_bwd_final(int C3, int K, const float *partial_sum, const float *gram,
                                  const float *W, const float *coef, const float *mean,
                                  const float *invstd, float *dW, void *stream) {
  if (C3 <= 0 || K <= 0 || !partial_sum || !gram || !W || !coef || !mean || !invstd || !dW)
    return fail2("pool_bwd_final: sizes / null pointer");
  hipLaunchKernelGGL(pool_bwd_final_kernel, dim3(C3), dim3(256), 0, (hipStream_t)stream, C3, K,
                     partial_sum, gram, W, coef, mean, invstd, dW);
  return check2("pool_bwd_final");
}

extern "C" int s2c_pool_bwd_dk(long long J, int C, const float *dOut, const float *ymax,
                               const float *scale, const float *shift, const float *coef,
                               const int *arg, float *dk, short *arg16, void *stream) {
  if (J <= 0 || C <= 0 || !dOut || !ymax || !scale || !shift || !coef || !dk || (arg16 && !arg))
    return fail2("pool_bwd_dk: sizes / null pointer");
  hipLaunchKernelGGL(pool_bwd_dk_kernel, dim3(grid1d(J * C, 256)), dim3(256), 0,
                     (hipStream_t)stream, dOut, ymax, scale, shift, coef, arg, dk, arg16, J * C, C);
  return check2("pool_bwd_dk");
}

extern "C" int s2c_pool_bwd_sp_blocks(long long J) { return J < 768 ? (int)J : 768; }

// partial: s2c_pool_bwd_sp_blocks(J) x (C3 * K + K) floats (SP | column sums of A).
// K in {32, 64, 128, 256}, C3 <= (256 / K) * 32, ns * K * 4 bytes of LDS (<= 64 KB).
extern "C" int s2c_pool_bwd_sp(long long J, int ns, int C3, int K, const float *A, const int *arg,
                               const float *dk, float *partial, const float *pscale,
                               const float *pshift, int prelu, void *stream) {
  if ((pscale == nullptr) != (pshift == nullptr)) return fail2("pool_bwd_sp: pscale / pshift");
  if (J <= 0 || ns <= 0 || C3 <= 0 || !(K == 32 || K == 64 || K == 128 || K == 256) ||
      C3 > (256 / K) * SP_CG || C3 > 128 || (C3 & 3) || (ns * K * 4) % 4096 || ns * K * 4 > 16384 ||
      !A || !arg || !dk || !partial || ((uintptr_t)A & 15) || ((uintptr_t)dk & 15) || ((uintptr_t)arg & 15))
    return fail2("pool_bwd_sp: sizes / null pointer");
  const size_t lds = (size_t)SP_RING * ((size_t)ns * K * 4 + 1024);
  hipLaunchKernelGGL(pool_bwd_sp_kernel, dim3(s2c_pool_bwd_sp_blocks(J)), dim3(256), lds,
                     (hipStream_t)stream, J, ns, C3, K, A, arg, dk, partial, pscale, pshift, prelu);
  return check2("pool_bwd_sp");
}

