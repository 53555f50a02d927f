// s2c_gemm.hip -- hand-written fp32 MFMA GEMM for the shared-MLP layers of the
// point-major set-abstraction path:
//
//     Y[M x N] = pro(A)[M x K] * W^T            W is (N x K) row-major (conv weight)
//
// Prologue (applied while the A tile is staged into LDS, so the operand is never
// materialised in HBM):
//   PRO_NONE    A as is;
//   PRO_BNRELU  relu(A * scale[k] + shift[k]) -- the previous layer's BatchNorm +
//               ReLU (pytorch_utils.py:100-120) fused into this layer's load;
//   PRO_GATHER  A row r = (scene b, centre j, sample s) is GATHERED on the fly:
//               channels 0..2 = (xyz[b, idx[r]] - new_xyz[b, j]) (/ radius),
//               channels 3..  = feats[b, idx[r], :]  -- ball-query grouping
//               (pointnet2_utils.py:347-359) fused into the first layer, the
//               (B, 3+C, npoint, nsample) tensor of the reference never exists.
// Epilogue: per-column sum / sum-of-squares of the Y tile as partials (one
// [sum | sumsq] row per row-block) = this layer's BatchNorm batch statistics
// without another pass over Y.
//
// fp32-in / fp32-accumulate `v_mfma_f32_32x32x2_f32` (exact f32 FMA chain, 157 TF
// peak on gfx950) keeps the 1e-4 parity budget; CDNA4 has no TF32-like fast path
// and bf16 is not parity-safe.
//
// Geometry: 256 threads = 4 waves, each wave owns a 64x64 tile (2x2 MFMA tiles,
// 64 accumulator registers).  N <= 64: waves stacked 4x1 (block = 256 rows x 64
// cols); otherwise 2x2 (block = 128 x 128).  K is walked in slices of 32 through
// LDS.  A slice is stored [row][32] with the k order permuted to
// pos(k) = (k & 1) * 16 + (k >> 1): the 16 values lane (i, k&1) feeds to the 16
// MFMAs of a slice are then contiguous, i.e. 4 x ds_read_b128 per operand tile
// instead of 16 x ds_read_b32, and a 36-float row stride makes those reads
// conflict-free across 16 consecutive rows.  The next slice is prefetched into
// registers while the current one feeds the matrix cores.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>
#include <stdlib.h>

using namespace s2c;

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 36;  // floats per LDS row (32 + 4 pad, 16-byte aligned)

enum { PRO_NONE = 0, PRO_BNRELU = 1, PRO_GATHER = 2, PRO_BNBWD = 3 };

// 16 bytes, 4-byte aligned (rows of the (B,N,3+C) cloud are not 16-byte aligned)
struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };

struct GatherArgs {
  const float *xyz;       // (b, n, 3)
  const float *new_xyz;   // (b, m, 3)
  const float *feats;     // point-major, row stride frs, batch stride fbs (floats)
  const int *idx;         // (b, m, ns) flattened = one entry per A row
  long long frs, fbs;
  int n, m, ns;
  float radius;
  int normalize;
};

// Inference epilogue (frozen BatchNorm): out = relu(acc * scale[c] + shift[c]) with
// scale = gamma / sqrt(var + eps), shift = beta - mean * scale (the arithmetic of
// bn_eval_coeffs + bn_relu in s2c_sa.hip), optionally max-pooled over groups of
// pool_ns consecutive rows (the nsample rows of a centre) -- the layer's BN, ReLU and
// the set-abstraction max-pool leave with the GEMM: no Y tensor, no extra pass.
struct EpiArgs {
  const float *gamma, *beta, *mean, *var;   // mean == nullptr: epilogue off
  float eps;
  int relu, pool_ns;                         // pool_ns in {0, 16, 32, 64}
  float *out;
  int ldo;
};

// PRO_BNBWD: the operand is the BatchNorm(+ReLU) BACKWARD of the upstream gradient, formed
// while the tile is staged (the arithmetic of bn_bwd_apply_kernel in s2c_sa.hip, bit for
// bit):   dz = dA * [Y*scale + shift > 0];  dY = k0 * (dz - k1 - ((Y - mean) * invstd) * k2)
// with k0 = gamma*invstd, k1 = sum(dz)/M, k2 = sum(dz*xhat)/M from the statistics pass.
// The GEMM is then  dX = dY W  (the layer's input gradient), and dY itself leaves as a side
// output of the column-block-0 workgroups (the weight / bias gradients read it): one pass
// over (dA, Y) instead of an apply pass plus a GEMM that re-reads dY.
struct BwdArgs {
  const float *Y;                 // (M x K) pre-BN activations of the layer; nullptr: off
  const float *scale, *shift, *mean, *invstd, *coef;   // K each; coef = 3 K
  float *dY;                      // (M x K) side output, may be nullptr
  int relu;
  // The GEMM's output dX IS the upstream gradient of the PREVIOUS layer (N channels), whose own
  // BatchNorm backward starts with two column sums over (dX, that layer's pre-activation nY):
  //   s1 = sum dz, s2 = sum dz * (nY - nmean) * ninvstd,  dz = dX * [nY * nscale + nshift > 0]
  // (the arithmetic of bn_bwd_stats_kernel, s2c_sa.hip).  nY != nullptr: the epilogue forms
  // them from the accumulators -- the statistics pass of the previous layer (a read of dX and
  // nY, 512 MB at SA1) shrinks to a read of nY here -- and writes them where the forward
  // statistics go (`partial`: one [s1 | s2] row per row block).
  const float *nY, *nscale, *nshift, *nmean, *ninvstd;
  int nrelu;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TM = 2, int TN = 2>
__device__ __forceinline__ void affine_epilogue(const f32x16 (&acc)[TM][TN], const EpiArgs &ep,
                                                long long M, int N, long long m0w, int n0w,
                                                int li, int lk) {
  // m0w / n0w: first row / column of this wave's (32 TM) x (32 TN) tile (pooling: TM == 2)
  // C/D layout of 32x32: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0w + j * 32 + li;
    const bool ok = col < N;
    float sc = 0.f, sh = 0.f;
    if (ok) {
      const float invstd = 1.0f / sqrtf(ep.var[col] + ep.eps);
      sc = (ep.gamma ? ep.gamma[col] : 1.0f) * invstd;
      sh = (ep.beta ? ep.beta[col] : 0.0f) - ep.mean[col] * sc;
    }
    if (ep.pool_ns == 0 || TM != 2) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const long long row = m0w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
          float v = acc[i][j][e] * sc + sh;
          if (ep.relu) v = fmaxf(v, 0.f);
          if (ok && row < M) ep.out[row * ep.ldo + col] = v;
        }
    } else {
      const int ns = ep.pool_ns;
      float g[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[i][j][e] * sc + sh;
          if (ep.relu) v = fmaxf(v, 0.f);
          // 16-row quarter of the wave tile this element belongs to: 2 i + (e >> 3)
          const int q = 2 * i + (e >> 3);
          const int grp = ns == 64 ? 0 : (ns == 32 ? (q >> 1) : q);
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (t == grp) g[t] = fmaxf(g[t], v);
        }
      const int groups = 64 / ns;
      const long long centre0 = m0w / ns, centres = M / ns;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < groups) {
          const float m = fmaxf(g[t], __shfl_xor(g[t], 32, 64));
          if (lk == 0 && ok && centre0 + t < centres) ep.out[(centre0 + t) * ep.ldo + col] = m;
        }
      }
    }
  }
}

__device__ __forceinline__ void lds_store_quad(float *row, int sq, float4 v) {
  // k = 4sq .. 4sq+3 -> even ks at [2sq, 2sq+1], odd ks at 16 + [2sq, 2sq+1]
  *reinterpret_cast<float2 *>(row + 2 * sq) = make_float2(v.x, v.z);
  *reinterpret_cast<float2 *>(row + 16 + 2 * sq) = make_float2(v.y, v.w);
}

template <int WM, int WN, int PRO>
__global__ __launch_bounds__(256) void rows_gemm_kernel(
    long long M, int N, int K, const float *__restrict__ A, int lda,
    const float *__restrict__ W, int ldw, const float *__restrict__ pscale,
    const float *__restrict__ pshift, GatherArgs ga, float *__restrict__ Y, int ldy,
    float *__restrict__ partial, int avec, int wvec, EpiArgs ep) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Ws[BN * LDS_LD];
  __shared__ float s_stat[2][WM][BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // staging map: thread -> (row = tid / 8 + 32*i, k-quad = tid % 8)
  const int sq = tid & 7, sr = tid >> 3;
  constexpr int AI = BM / 32, WI = BN / 32;
  float4 ra[AI], rw[WI];

  // PRO_GATHER: per staged row, the source-row offsets (computed once)
  long long g_src[PRO == PRO_GATHER ? AI : 1];
  int g_pt[PRO == PRO_GATHER ? AI : 1], g_ctr[PRO == PRO_GATHER ? AI : 1];
  if (PRO == PRO_GATHER) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const long long row = m0 + sr + 32 * i;
      const long long rc = row < M ? row : M - 1;
      const long long bj = rc / ga.ns;
      const long long b = bj / ga.m;
      const int p = ga.idx[rc];
      g_src[i] = b * ga.fbs + (long long)p * ga.frs;
      g_pt[i] = (int)((b * ga.n + p) * 3);
      g_ctr[i] = (int)(bj * 3);
    }
  }

  auto load_slice = [&](int k0) {
    const int k = k0 + sq * 4;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PRO == PRO_BNRELU) {
      if (k + 3 < K) {
        sc = *reinterpret_cast<const float4 *>(pscale + k);
        sh = *reinterpret_cast<const float4 *>(pshift + k);
      } else {
        if (k < K) { sc.x = pscale[k]; sh.x = pshift[k]; }
        if (k + 1 < K) { sc.y = pscale[k + 1]; sh.y = pshift[k + 1]; }
        if (k + 2 < K) { sc.z = pscale[k + 2]; sh.z = pshift[k + 2]; }
      }
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const long long row = m0 + sr + 32 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < M) {
        if (PRO == PRO_GATHER) {
          if (k >= 3 && k + 3 < K) {
            // whole quad inside the feature part: ONE 16-byte load.  A gathered row
            // starts at a multiple of (3+C)*4 bytes, so the address is only 4-byte
            // aligned (global memory takes dword-aligned dwordx4 loads).
            const F4U q = *reinterpret_cast<const F4U *>(ga.feats + g_src[i] + (k - 3));
            v = make_float4(q.x, q.y, q.z, q.w);
          } else {
            float e[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int kc = k + c;
              float x = 0.f;
              if (kc < K) {
                if (kc < 3) {
                  x = ga.xyz[g_pt[i] + kc] - ga.new_xyz[g_ctr[i] + kc];
                  if (ga.normalize) x = x / ga.radius;
                } else {
                  x = ga.feats[g_src[i] + (kc - 3)];
                }
              }
              e[c] = x;
            }
            v = make_float4(e[0], e[1], e[2], e[3]);
          }
        } else {
          const float *p = A + row * lda + k;
          if (k + 3 < K) {
            if (avec) v = *reinterpret_cast<const float4 *>(p);
            else {     // rows of 3 + C floats (the cloud's feature columns): one dword-aligned 16-byte load
              const F4U q = *reinterpret_cast<const F4U *>(p);
              v = make_float4(q.x, q.y, q.z, q.w);
            }
          } else {
            if (k < K) v.x = p[0];
            if (k + 1 < K) v.y = p[1];
            if (k + 2 < K) v.z = p[2];
          }
          if (PRO == PRO_BNRELU) {
            // columns >= K keep scale 1 / shift 0 on a zero operand
            v.x = fmaxf(v.x * sc.x + sh.x, 0.f); v.y = fmaxf(v.y * sc.y + sh.y, 0.f);
            v.z = fmaxf(v.z * sc.z + sh.z, 0.f); v.w = fmaxf(v.w * sc.w + sh.w, 0.f);
            if (k + 3 >= K) {
              if (k >= K) v.x = 0.f;
              if (k + 1 >= K) v.y = 0.f;
              if (k + 2 >= K) v.z = 0.f;
              v.w = 0.f;
            }
          }
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int n = n0 + sr + 32 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N) {
        const float *p = W + (long long)n * ldw + k;
        if (k + 3 < K) {
          if (wvec) v = *reinterpret_cast<const float4 *>(p);
          else { v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3]; }
        } else {
          if (k < K) v.x = p[0];
          if (k + 1 < K) v.y = p[1];
          if (k + 2 < K) v.z = p[2];
        }
      }
      rw[i] = v;
    }
  };
  auto store_slice = [&]() {
#pragma unroll
    for (int i = 0; i < AI; ++i) lds_store_quad(As + (sr + 32 * i) * LDS_LD, sq, ra[i]);
#pragma unroll
    for (int i = 0; i < WI; ++i) lds_store_quad(Ws + (sr + 32 * i) * LDS_LD, sq, rw[i]);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int li = lane & 31, lk = lane >> 5;
  const float *a_base0 = As + (wm * 64 + li) * LDS_LD + lk * 16;
  const float *a_base1 = a_base0 + 32 * LDS_LD;
  const float *w_base0 = Ws + (wn * 64 + li) * LDS_LD + lk * 16;
  const float *w_base1 = w_base0 + 32 * LDS_LD;

  load_slice(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();           // previous slice fully consumed
    store_slice();
    __syncthreads();
    if (k0 + BK < K) load_slice(k0 + BK);   // prefetch under the MFMAs
    // operand reads of group q+1 are issued BEFORE the 16 MFMAs of group q (the
    // scheduling barriers pin that order): LDS latency hides under the matrix pipe
    float4 a0 = *reinterpret_cast<const float4 *>(a_base0);
    float4 a1 = *reinterpret_cast<const float4 *>(a_base1);
    float4 b0 = *reinterpret_cast<const float4 *>(w_base0);
    float4 b1 = *reinterpret_cast<const float4 *>(w_base1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
      if (q < 3) {
        na0 = *reinterpret_cast<const float4 *>(a_base0 + 4 * (q + 1));
        na1 = *reinterpret_cast<const float4 *>(a_base1 + 4 * (q + 1));
        nb0 = *reinterpret_cast<const float4 *>(w_base0 + 4 * (q + 1));
        nb1 = *reinterpret_cast<const float4 *>(w_base1 + 4 * (q + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      const float a0v[4] = {a0.x, a0.y, a0.z, a0.w}, a1v[4] = {a1.x, a1.y, a1.z, a1.w};
      const float b0v[4] = {b0.x, b0.y, b0.z, b0.w}, b1v[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[s], b0v[s], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[s], b1v[s], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[s], b0v[s], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[s], b1v[s], acc[1][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
  }

  if (ep.mean != nullptr) {    // inference: BN + ReLU (+ max-pool) leave with the GEMM
    affine_epilogue(acc, ep, M, N, m0 + wm * 64, n0 + wn * 64, li, lk);
    return;
  }
  // ---- epilogue: store Y, column statistics -------------------------------
  // C/D layout of 32x32: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + li;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long long row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        const float v = acc[i][j][e];
        if (row < M && col < N) {
          Y[row * ldy + col] = v;
          s1 += v;
          s2 += v * v;
        }
      }
    }
    if (partial != nullptr) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lk == 0) {
        s_stat[0][wm][wn * 64 + j * 32 + li] = s1;
        s_stat[1][wm][wn * 64 + j * 32 + li] = s2;
      }
    }
  }
  if (partial != nullptr) {
    __syncthreads();
    for (int c = tid; c < BN; c += 256) {
      if (n0 + c < N) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s1 += s_stat[0][w][c]; s2 += s_stat[1][w][c]; }
        float *p = partial + (long long)blockIdx.x * 2 * N;
        p[n0 + c] = s1;
        p[N + n0 + c] = s2;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// bf16x3 variant: fp32-accurate products on the bf16 matrix pipe.
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
// (the two residuals are exact in fp32; 3 x 8 mantissa bits cover the 24 of fp32), and
// x*y ~= the 6 plane products with i + j <= 2 (dropped: mid*lo, lo*mid, lo*lo <= 2^-24
// relative), each exact in the fp32 accumulator: the result differs from an fp32 FMA
// chain by ~1e-7 relative -- the size of fp32's own rounding -- while the bf16
// `v_mfma_f32_32x32x16_bf16` runs at 16x the fp32 MFMA rate (6 MFMAs = 2.7x faster).
// The split happens once per staged element on the VALU (v_cvt_pk_bf16_f32).
// ---------------------------------------------------------------------------
// Diagnostics (tools/prof_gemm.py): when set, wave 0 of workgroup `g_prof_block` writes
// s_memtime stamps of its phases to g_prof (device memory, 64 slots).
__device__ long long *g_prof = nullptr;
__device__ int g_prof_block = 0;
#define X3_STAMP() do { if (prof_on && nstamp < 64) { __builtin_amdgcn_sched_barrier(0); \
    pr[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int X3_LD = 40;   // bf16 per LDS row: 32 + 8 pad (80 bytes, conflict-free b128)
__host__ __device__ constexpr size_t x3_tile_bytes(int WM, int WN) {
  return (size_t)3 * (64 * WM + 64 * WN) * X3_LD * sizeof(unsigned short);
}

__device__ __forceinline__ void x3_split2(f32x2 v, unsigned &hi, unsigned &mid, unsigned &lo) {
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
  const bf16x2 m = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
  const bf16x2 l = __builtin_convertvector(r2, bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  mid = __builtin_bit_cast(unsigned, m);
  lo = __builtin_bit_cast(unsigned, l);
}

// quad sq of a row: k = 4 sq .. 4 sq + 3 -> 8 bytes at bf16 offset 4 sq of each plane
__device__ __forceinline__ void x3_store_quad(unsigned short *base, int rows, int row, int sq,
                                              float4 v) {
  unsigned h0, m0, l0, h1, m1, l1;
  f32x2 p0 = {v.x, v.y}, p1 = {v.z, v.w};
  x3_split2(p0, h0, m0, l0);
  x3_split2(p1, h1, m1, l1);
  unsigned short *d = base + row * X3_LD + 4 * sq;
  *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
  *reinterpret_cast<uint2 *>(d + rows * X3_LD) = make_uint2(m0, m1);
  *reinterpret_cast<uint2 *>(d + 2 * rows * X3_LD) = make_uint2(l0, l1);
}

template <int WM, int WN, int PRO>
__global__ __launch_bounds__(256, 2) void rows_gemm_x3_kernel(
    long long M, int N, int K, const float *__restrict__ A, int lda,
    const float *__restrict__ W, int ldw, const float *__restrict__ pscale,
    const float *__restrict__ pshift, GatherArgs ga, float *__restrict__ Y, int ldy,
    float *__restrict__ partial, int avec, int wvec, EpiArgs ep, BwdArgs bw) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  // three bf16 planes (hi, mid, lo) per operand tile, rows of X3_LD bf16 (32 + 8 pad)
  extern __shared__ __attribute__((aligned(16))) unsigned char x3_smem[];
  unsigned short *As = reinterpret_cast<unsigned short *>(x3_smem);
  unsigned short *Ws = As + 3 * BM * X3_LD;
  float (*s_stat)[WM][BN] = reinterpret_cast<float (*)[WM][BN]>(x3_smem + x3_tile_bytes(WM, WN));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  long long *pr = g_prof;
  const bool prof_on = pr != nullptr && (int)blockIdx.x == g_prof_block && blockIdx.y == 0 && tid == 0;
  int nstamp = 0;
  X3_STAMP();

  // staging map: thread -> (row = tid / 8 + 32*i, k-quad = tid % 8)
  const int sq = tid & 7, sr = tid >> 3;
  constexpr int AI = BM / 32, WI = BN / 32;
  float4 ra[AI], rw[WI], rb[AI], rwb[WI];   // two slices of register prefetch

  // PRO_GATHER: per staged row, the source-row offsets (computed once)
  long long g_src[PRO == PRO_GATHER ? AI : 1];
  int g_pt[PRO == PRO_GATHER ? AI : 1], g_ctr[PRO == PRO_GATHER ? AI : 1];
  if (PRO == PRO_GATHER) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const long long row = m0 + sr + 32 * i;
      const long long rc = row < M ? row : M - 1;
      const long long bj = rc / ga.ns;
      const long long b = bj / ga.m;
      const int p = ga.idx[rc];
      g_src[i] = b * ga.fbs + (long long)p * ga.frs;
      g_pt[i] = (int)((b * ga.n + p) * 3);
      g_ctr[i] = (int)(bj * 3);
    }
  }

  auto load_slice = [&](float4 (&ra)[AI], float4 (&rw)[WI], int k0) {
    const int k = k0 + sq * 4;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PRO == PRO_BNRELU) {
      if (k + 3 < K) {
        sc = *reinterpret_cast<const float4 *>(pscale + k);
        sh = *reinterpret_cast<const float4 *>(pshift + k);
      } else {
        if (k < K) { sc.x = pscale[k]; sh.x = pshift[k]; }
        if (k + 1 < K) { sc.y = pscale[k + 1]; sh.y = pshift[k + 1]; }
        if (k + 2 < K) { sc.z = pscale[k + 2]; sh.z = pshift[k + 2]; }
      }
    }
    // PRO_BNBWD (K % 4 == 0): the seven per-channel quads of this k-quad
    float4 b_mu = sh, b_is = sh, b_k0 = sh, b_k1 = sh, b_k2 = sh;
    if (PRO == PRO_BNBWD && k < K) {
      sc = *reinterpret_cast<const float4 *>(bw.scale + k);
      sh = *reinterpret_cast<const float4 *>(bw.shift + k);
      b_mu = *reinterpret_cast<const float4 *>(bw.mean + k);
      b_is = *reinterpret_cast<const float4 *>(bw.invstd + k);
      b_k0 = *reinterpret_cast<const float4 *>(bw.coef + k);
      b_k1 = *reinterpret_cast<const float4 *>(bw.coef + K + k);
      b_k2 = *reinterpret_cast<const float4 *>(bw.coef + 2 * K + k);
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const long long row = m0 + sr + 32 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < M) {
        if (PRO == PRO_GATHER) {
          if (k >= 3 && k + 3 < K) {
            // whole quad inside the feature part: ONE 16-byte load.  A gathered row
            // starts at a multiple of (3+C)*4 bytes, so the address is only 4-byte
            // aligned (global memory takes dword-aligned dwordx4 loads).
            const F4U q = *reinterpret_cast<const F4U *>(ga.feats + g_src[i] + (k - 3));
            v = make_float4(q.x, q.y, q.z, q.w);
          } else {
            float e[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int kc = k + c;
              float x = 0.f;
              if (kc < K) {
                if (kc < 3) {
                  x = ga.xyz[g_pt[i] + kc] - ga.new_xyz[g_ctr[i] + kc];
                  if (ga.normalize) x = x / ga.radius;
                } else {
                  x = ga.feats[g_src[i] + (kc - 3)];
                }
              }
              e[c] = x;
            }
            v = make_float4(e[0], e[1], e[2], e[3]);
          }
        } else if (PRO == PRO_BNBWD) {
          if (k < K) {
            float4 g = *reinterpret_cast<const float4 *>(A + row * lda + k);
            const float4 y = *reinterpret_cast<const float4 *>(bw.Y + row * (long long)K + k);
            if (bw.relu) {
              if (!(y.x * sc.x + sh.x > 0.f)) g.x = 0.f;
              if (!(y.y * sc.y + sh.y > 0.f)) g.y = 0.f;
              if (!(y.z * sc.z + sh.z > 0.f)) g.z = 0.f;
              if (!(y.w * sc.w + sh.w > 0.f)) g.w = 0.f;
            }
            v.x = b_k0.x * (g.x - b_k1.x - ((y.x - b_mu.x) * b_is.x) * b_k2.x);
            v.y = b_k0.y * (g.y - b_k1.y - ((y.y - b_mu.y) * b_is.y) * b_k2.y);
            v.z = b_k0.z * (g.z - b_k1.z - ((y.z - b_mu.z) * b_is.z) * b_k2.z);
            v.w = b_k0.w * (g.w - b_k1.w - ((y.w - b_mu.w) * b_is.w) * b_k2.w);
            if (bw.dY != nullptr && blockIdx.y == 0)
              *reinterpret_cast<float4 *>(bw.dY + row * (long long)K + k) = v;
          }
        } else {
          const float *p = A + row * lda + k;
          if (k + 3 < K) {
            if (avec) v = *reinterpret_cast<const float4 *>(p);
            else {     // rows of 3 + C floats (the cloud's feature columns): one dword-aligned 16-byte load
              const F4U q = *reinterpret_cast<const F4U *>(p);
              v = make_float4(q.x, q.y, q.z, q.w);
            }
          } else {
            if (k < K) v.x = p[0];
            if (k + 1 < K) v.y = p[1];
            if (k + 2 < K) v.z = p[2];
          }
          if (PRO == PRO_BNRELU) {
            // columns >= K keep scale 1 / shift 0 on a zero operand
            v.x = fmaxf(v.x * sc.x + sh.x, 0.f); v.y = fmaxf(v.y * sc.y + sh.y, 0.f);
            v.z = fmaxf(v.z * sc.z + sh.z, 0.f); v.w = fmaxf(v.w * sc.w + sh.w, 0.f);
            if (k + 3 >= K) {
              if (k >= K) v.x = 0.f;
              if (k + 1 >= K) v.y = 0.f;
              if (k + 2 >= K) v.z = 0.f;
              v.w = 0.f;
            }
          }
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int n = n0 + sr + 32 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N) {
        const float *p = W + (long long)n * ldw + k;
        if (k + 3 < K) {
          if (wvec) v = *reinterpret_cast<const float4 *>(p);
          else { v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3]; }
        } else {
          if (k < K) v.x = p[0];
          if (k + 1 < K) v.y = p[1];
          if (k + 2 < K) v.z = p[2];
        }
      }
      rw[i] = v;
    }
  };
  auto store_slice = [&](float4 (&ra)[AI], float4 (&rw)[WI]) {
#pragma unroll
    for (int i = 0; i < AI; ++i) x3_store_quad(As, BM, sr + 32 * i, sq, ra[i]);
#pragma unroll
    for (int i = 0; i < WI; ++i) x3_store_quad(Ws, BN, sr + 32 * i, sq, rw[i]);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int li = lane & 31, lk = lane >> 5;
  // operand address of (plane, row, k-half s, lane group lk): 8 bf16 = 16 bytes
  const unsigned short *a_row0 = As + (wm * 64 + li) * X3_LD + lk * 8;
  const unsigned short *w_row0 = Ws + (wn * 64 + li) * X3_LD + lk * 8;

  auto mfma_slice = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[t][p] = *reinterpret_cast<const bf16x8 *>(a_row0 + (p * BM + t * 32) * X3_LD + s * 16);
          b[t][p] = *reinterpret_cast<const bf16x8 *>(w_row0 + (p * BN + t * 32) * X3_LD + s * 16);
        }
      // x*y ~= sum of the 6 plane products with i + j <= 2 (hi=0, mid=1, lo=2):
      // small terms first
      // term-major order: the four accumulators take turns, so consecutive MFMAs are
      // independent (a chain of six on one accumulator stalls on every RAW: the SQ
      // counters showed 49 % issue-stall cycles)
      constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[t]], b[j][TB[t]],
                                                                acc[i][j], 0, 0, 0);
    }
  };

  // Two slices of loads are in flight at any time (the MFMA phase of a slice is now
  // shorter than one memory round trip): slice s+2 is requested as soon as the
  // registers of slice s have been drained into LDS.
  X3_STAMP();                  // [1] set-up done (gather: idx fetched)
  load_slice(ra, rw, 0);
  if (K > BK) load_slice(rb, rwb, BK);
  X3_STAMP();                  // [2] two slices of loads issued
  for (int k0 = 0; k0 < K; k0 += 2 * BK) {
    __syncthreads();           // previous slice fully consumed
    X3_STAMP();                // per slice: barrier
    store_slice(ra, rw);
    X3_STAMP();                //   loads landed + split + LDS stores
    __syncthreads();
    X3_STAMP();                //   barrier
    if (k0 + 2 * BK < K) load_slice(ra, rw, k0 + 2 * BK);
    X3_STAMP();                //   next loads issued
    mfma_slice();
    X3_STAMP();                //   MFMAs
    if (k0 + BK >= K) break;
    __syncthreads();
    X3_STAMP();
    store_slice(rb, rwb);
    X3_STAMP();
    __syncthreads();
    X3_STAMP();
    if (k0 + 3 * BK < K) load_slice(rb, rwb, k0 + 3 * BK);
    X3_STAMP();
    mfma_slice();
    X3_STAMP();
  }

  if (ep.mean != nullptr) {    // inference: BN + ReLU (+ max-pool) leave with the GEMM
    affine_epilogue(acc, ep, M, N, m0 + wm * 64, n0 + wn * 64, li, lk);
    return;
  }
  // ---- epilogue: store Y, column statistics -------------------------------
  // C/D layout of 32x32: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
  // (Wide stores were tried here -- the tile transposed through LDS or with 4x4 DPP
  // exchanges into dwordx4 row pieces -- and LOSE in this skeleton: (1M,64,64) 136 -> 154 us,
  // (262144,128,128) 94 -> 107 us.  They pay in the streaming kernel of s2c_gemm2.hip.)
  const bool next_stats = PRO == PRO_BNBWD && bw.nY != nullptr;
  // the previous layer's pre-activations of this lane's 64 elements: every load in flight before
  // the first store (one round trip, not one per 32 x 32 sub-tile)
  float ny[PRO == PRO_BNBWD ? 2 : 1][PRO == PRO_BNBWD ? 2 : 1][PRO == PRO_BNBWD ? 16 : 1];
  if (next_stats) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const long long row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
          ny[PRO == PRO_BNBWD ? j : 0][PRO == PRO_BNBWD ? i : 0][PRO == PRO_BNBWD ? e : 0] =
              (row < M && col < N) ? bw.nY[row * (long long)N + col] : 0.f;
        }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + li;
    float s1 = 0.f, s2 = 0.f;
    float nsc = 0.f, nsh = 0.f, nmu = 0.f, nis = 0.f;
    if (next_stats && col < N) {
      nsc = bw.nscale[col]; nsh = bw.nshift[col]; nmu = bw.nmean[col]; nis = bw.ninvstd[col];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long long row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        const float v = acc[i][j][e];
        if (row < M && col < N) {
          Y[row * ldy + col] = v;
          if (next_stats) {
            const float y = ny[PRO == PRO_BNBWD ? j : 0][PRO == PRO_BNBWD ? i : 0][PRO == PRO_BNBWD ? e : 0];
            float dz = v;
            if (bw.nrelu && !(y * nsc + nsh > 0.f)) dz = 0.f;
            s1 += dz;
            s2 += dz * ((y - nmu) * nis);
          } else {
            s1 += v;
            s2 += v * v;
          }
        }
      }
    }
    if (partial != nullptr) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lk == 0) {
        s_stat[0][wm][wn * 64 + j * 32 + li] = s1;
        s_stat[1][wm][wn * 64 + j * 32 + li] = s2;
      }
    }
  }
  X3_STAMP();                  // epilogue stores issued
  if (prof_on) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    X3_STAMP();                // ... and acknowledged
    pr[63] = nstamp;
  }
  if (partial != nullptr) {
    __syncthreads();
    for (int c = tid; c < BN; c += 256) {
      if (n0 + c < N) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s1 += s_stat[0][w][c]; s2 += s_stat[1][w][c]; }
        float *p = partial + (long long)blockIdx.x * 2 * N;
        p[n0 + c] = s1;
        p[N + n0 + c] = s2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// helpers of rows_gemm_c64_kernel below
struct MidPlanes { bf16x8 p[3]; };

__device__ __forceinline__ MidPlanes mid_split8(float4 lo4, float4 hi4) {
  unsigned h[4], m[4], l[4];
  const f32x2 x0 = {lo4.x, lo4.y}, x1 = {lo4.z, lo4.w}, x2 = {hi4.x, hi4.y}, x3 = {hi4.z, hi4.w};
  x3_split2(x0, h[0], m[0], l[0]);
  x3_split2(x1, h[1], m[1], l[1]);
  x3_split2(x2, h[2], m[2], l[2]);
  x3_split2(x3, h[3], m[3], l[3]);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 hv = {h[0], h[1], h[2], h[3]}, mv = {m[0], m[1], m[2], m[3]},
              lv = {l[0], l[1], l[2], l[3]};
  MidPlanes o;
  o.p[0] = __builtin_bit_cast(bf16x8, hv);
  o.p[1] = __builtin_bit_cast(bf16x8, mv);
  o.p[2] = __builtin_bit_cast(bf16x8, lv);
  return o;
}

// 4 consecutive floats of a row from column k on, zero from column K on (rows need not be
// 16-byte aligned: lda = 259)
__device__ __forceinline__ float4 mid_load4(const float *__restrict__ row, int k, int K) {
  if (k + 4 <= K) {
    const F4U a = *reinterpret_cast<const F4U *>(row + k);
    return make_float4(a.x, a.y, a.z, a.w);
  }
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K) v.x = row[k];
  if (k + 1 < K) v.y = row[k + 1];
  if (k + 2 < K) v.z = row[k + 2];
  return v;
}

// ---------------------------------------------------------------------------------------
// Tall problems with N > 64 that the streaming kernel (s2c_gemm2.hip) does not take (K or
// N = 256: its resident W planes do not fit): the 2x2 tiling of rows_gemm_x3_kernel with the
// whole 64-k chunks of both operands staged as fp32 --
//  * workgroup = 4 waves on a 128 x 128 tile (wave = 64 x 64, the accumulator layout and the
//    epilogues of the kernel above), K in chunks of 64 kept in LDS as fp32 (2 x 34 KB, rows
//    padded to 68 floats: conflict-free 16-byte fragment reads), split into bf16 planes in
//    registers after the fragment read;
//  * (tried first: feeding the MFMA lanes straight from global memory -- lane (i, half) = 32
//    bytes of row i, no LDS -- is bound by the vector L1's 64 tag look-ups per instruction;
//    split-K over 8 waves with wave-private LDS transposers needs no barrier but sums four
//    partial tiles through LDS; 128-k chunks with one workgroup per CU serialise load, MFMA and
//    store phases: all measured slower than this, tools/bench_mid_gemm.py)
//  * one chunk = 16 coalesced 16-byte loads per thread (256 contiguous bytes per 16 lanes), the
//    NEXT chunk in flight while this one is multiplied: half the barriers of the 32-k slices
//    and twice the bytes in flight per workgroup, two workgroups per CU (70 KB of LDS, <= 256
//    VGPRs) so that one's loads and stores run under the other's MFMAs.
// (Also tried: a persistent grid of two workgroups per CU walking the tiles, the first chunk of
// the NEXT tile requested before the current tile's stores -- slower everywhere, (65536,128,128)
// 25 -> 34 us, (262144,256,128) 161 -> 180 us: the hardware's dispatch of one workgroup per tile
// already overlaps one tile's stores with its neighbour's loads, and balances the tail.)
// The timeline of the 32-k-slice kernel at (262144, 256, 128) (tools/prof_gemm.py) showed what
// this replaces: 19-27 us per 128 x 128 x 128 tile, of which 3-8 us to issue the first two
// slices, 1.6 us to issue each later one and 4 us of epilogue -- against 2.6 us of MFMA time.
// Same products in the same k order as the kernel above: bit-identical results.
// the column sums of the next BatchNorm backward out of the epilogue (see BwdArgs::nY): the
// GEMM's output is the upstream gradient of a layer whose pre-activations are Y
struct NextStats {
  const float *Y, *scale, *shift, *mean, *invstd;   // Y == nullptr: off
  int relu;
};

// pooled training layer on the 64-k-chunk kernel: besides Y and the statistics, per centre (ns =
// 16 / 32 / 64 consecutive rows) and column the extremum of Y that BatchNorm + ReLU + max-pool will
// select -- the maximum of sign * Y, sign = the sign of the layer's gamma (StreamArgs::ext of
// s2c_gemm2.hip; here the sign is applied to the accumulators, not folded into the weights: Y is
// written as well) -- and its first row: the pooled pass over Y (s2c_bn_relu_max) shrinks to
// s2c_pool_select on J x N values.
struct PoolExt {
  float *ext; int *aext;          // (J x N); ext == nullptr: off
  const float *sign;              // gamma (N) or nullptr = all positive
  int ns;
};

constexpr int C64_LD = 68;
constexpr size_t C64_LDS_BYTES = 2 * 128 * C64_LD * sizeof(float);     // 69632

// Two shapes of the same kernel.  (WM, WN, TM, TN) = (2, 2, 2, 2): workgroup tile 128 x 128, wave
// 64 x 64 (all epilogues).  (4, 1, 1, 1): workgroup tile 128 x 32, wave 32 x 32 -- for launches whose
// 128 x 128 tiles would cover the chip once or less: such a launch lasts as long as ONE tile (two
// 64-k chunks of loads, staging, barriers, 64 four-byte stores per lane), and a quarter of the
// columns is a quarter of the MFMAs and stores per wave, 10 instead of 16 loads per thread and
// four times the workgroups.  The 128-row tile -- the statistics partials -- is the same; the
// pooled epilogues need the 64-row wave tile and stay on the wide shape.  Every output sums the
// same products in the same order: bit-identical values.
template <int PRO, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256, 2) void rows_gemm_c64_kernel(
    long long M, int N, int K, const float *__restrict__ A, int lda,
    const float *__restrict__ W, int ldw, const float *__restrict__ pscale,
    const float *__restrict__ pshift, GatherArgs ga, float *__restrict__ Y, int ldy,
    float *__restrict__ partial, EpiArgs ep, float *__restrict__ side, int ld_side,
    NextStats nx, PoolExt px) {
  static_assert(WM * WN == 4 && 32 * TM * WM == 128, "four waves on 128 rows");
  constexpr int WTM = 32 * TM, WTN = 32 * TN, BN = WTN * WN, NPW = BN / 16;
  extern __shared__ __attribute__((aligned(16))) float c64_smem[];
  __shared__ float s_stat[2][WM][BN];
  float *As = c64_smem, *Ws = c64_smem + 128 * C64_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lk = lane >> 5;
  // 1-D grid, XCD-aware: workgroup ids that differ by 8 share an XCD (its L2) and start one
  // after the other -- the column blocks of ONE row tile, so that the A tile comes from HBM once
  const int nby = (N + BN - 1) / BN;
  const int by = (int)((blockIdx.x >> 3) % nby);
  const long long bx = 8ll * ((blockIdx.x >> 3) / nby) + (blockIdx.x & 7);
  if (bx * 128 >= M) return;
  const long long m0 = bx * 128;
  const int n0 = by * BN;
  const int nchunks = (K + 63) >> 6;
  const bool live = n0 + WTN * wn < N;              // a wave beyond N multiplies nothing
  long long *pr = g_prof;
  const bool prof_on = pr != nullptr && (int)bx == g_prof_block && by == 0 && tid == 0;
  int nstamp = 0;
  X3_STAMP();                  // [0] start

  // load map: pass i = rows 16 i .. 16 i + 15 of the tile, thread -> (row 16 i + tid / 16,
  // k-quad tid % 16): 256 contiguous bytes per 16 lanes
  const int lr = tid >> 4, kq = tid & 15;
  // gather: per staged row the source offsets (32-bit element offsets: the cloud is below
  // 2^31 floats); the other prologues recompute their row offsets at every issue (registers)
  unsigned goff[PRO == PRO_GATHER ? 8 : 1];
  int g_pt[PRO == PRO_GATHER ? 8 : 1], g_ctr[PRO == PRO_GATHER ? 8 : 1];
  if (PRO == PRO_GATHER) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long row = m0 + 16 * i + lr;
      const long long rc = row < M ? row : M - 1;
      const long long bj = rc / ga.ns;
      const long long b = bj / ga.m;
      const int p = ga.idx[rc];
      goff[i] = (unsigned)(b * ga.fbs + (long long)p * ga.frs);
      g_pt[i] = (int)((b * ga.n + p) * 3);
      g_ctr[i] = (int)(bj * 3);
    }
  }

  float4 ra[8], rw[NPW];
  auto issue = [&](int c) {
    const int k = 64 * c + 4 * kq;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (PRO == PRO_GATHER) {
        const float *src = ga.feats + goff[PRO == PRO_GATHER ? i : 0];
        if (k >= 3) {                                 // column k = feature k - 3
          ra[i] = mid_load4(src - 3, k, K);
        } else {                                      // k == 0: dx, dy, dz, first feature
          float e[4];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            float x = ga.xyz[g_pt[PRO == PRO_GATHER ? i : 0] + q] -
                      ga.new_xyz[g_ctr[PRO == PRO_GATHER ? i : 0] + q];
            if (ga.normalize) x = x / ga.radius;
            e[q] = x;
          }
          e[3] = K > 3 ? src[0] : 0.f;
          ra[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
      } else {
        const long long row = m0 + 16 * i + lr;
        const long long rc = row < M ? row : M - 1;   // clamped: loads stay inside the matrix
        ra[i] = mid_load4(A + rc * lda, k, K);
      }
      if (i < NPW) {
        const int n = n0 + 16 * i + lr;
        rw[i < NPW ? i : 0] = mid_load4(W + (long long)(n < N ? n : N - 1) * ldw, k, K);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  issue(0);
  X3_STAMP();                  // [1] loads issued
  for (int c = 0; c < nchunks; ++c) {
    const int k = 64 * c + 4 * kq;
    if (PRO == PRO_BNRELU) {
      const float4 sc = mid_load4(pscale, k, K), sh = mid_load4(pshift, k, K);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float4 v = ra[i];
        v.x = k < K ? fmaxf(v.x * sc.x + sh.x, 0.f) : 0.f;
        v.y = k + 1 < K ? fmaxf(v.y * sc.y + sh.y, 0.f) : 0.f;
        v.z = k + 2 < K ? fmaxf(v.z * sc.z + sh.z, 0.f) : 0.f;
        v.w = k + 3 < K ? fmaxf(v.w * sc.w + sh.w, 0.f) : 0.f;
        ra[i] = v;
        // the activated operand leaves as a side output (the weight gradient needs it): the
        // separate BN+ReLU pass over the pre-activation tensor is gone
        const long long row = m0 + 16 * i + lr;
        if (side != nullptr && by == 0 && row < M && k < K) {
          float *dst = side + row * ld_side + k;
          if (k + 4 <= K && (ld_side & 3) == 0) {
            *reinterpret_cast<float4 *>(dst) = v;
          } else {
            dst[0] = v.x;
            if (k + 1 < K) dst[1] = v.y;
            if (k + 2 < K) dst[2] = v.z;
            if (k + 3 < K) dst[3] = v.w;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      *reinterpret_cast<float4 *>(As + (16 * i + lr) * C64_LD + 4 * kq) = ra[i];
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      *reinterpret_cast<float4 *>(Ws + (16 * i + lr) * C64_LD + 4 * kq) = rw[i];
    if (c + 1 < nchunks) issue(c + 1);               // flies while this chunk is multiplied
    __syncthreads();
    X3_STAMP();                // per chunk: operands landed, staged, barrier
    if (live) {
      const int kleft = K - 64 * c;
      const int nsteps = kleft >= 64 ? 4 : (kleft + 15) >> 4;
      const float *fa0 = As + (WTM * wm + li) * C64_LD + 8 * lk;
      const float *fw0 = Ws + (WTN * wn + li) * C64_LD + 8 * lk;
      for (int s = 0; s < nsteps; ++s) {
        // (one W fragment at a time: 20 registers less than holding both; every accumulator
        // still receives its six products in the same order)
        MidPlanes pa[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const float *fa = fa0 + 32 * t * C64_LD + 16 * s;
          pa[t] = mid_split8(*reinterpret_cast<const float4 *>(fa),
                             *reinterpret_cast<const float4 *>(fa + 4));
        }
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float *fw = fw0 + 32 * j * C64_LD + 16 * s;
          const MidPlanes pb = mid_split8(*reinterpret_cast<const float4 *>(fw),
                                          *reinterpret_cast<const float4 *>(fw + 4));
#pragma unroll
          for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i].p[TA[q]], pb.p[TB[q]],
                                                                  acc[i][j], 0, 0, 0);
        }
      }
    }
    X3_STAMP();                //            MFMAs issued
    if (c + 1 < nchunks) __syncthreads();            // the tiles are overwritten next
  }

  if (ep.mean != nullptr) {    // inference: BN + ReLU (+ max-pool) leave with the GEMM
    affine_epilogue<TM, TN>(acc, ep, M, N, m0 + wm * WTM, n0 + wn * WTN, li, lk);
    return;
  }
  if constexpr (PRO != PRO_GATHER && TM == 2 && TN == 2) if (px.ext != nullptr) {
    // ---- pooled layer: per centre and column the maximum of sign * y and its first row -----
    const int ns = px.ns;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
      const float sgn = (px.sign != nullptr && col < N && px.sign[col] < 0.f) ? -1.f : 1.f;
      float gv[2][2];
      int ga2[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float bv = -INFINITY;
          int ba = 0;
#pragma unroll
          for (int e8 = 0; e8 < 8; ++e8) {        // ascending rows, strict compare: first maximum
            const int ro = (e8 & 3) + 8 * (e8 >> 2) + 4 * lk;
            const long long row = m0 + wm * 64 + i * 32 + 16 * h + ro;
            const float t = acc[i][j][8 * h + e8] * sgn;
            if (row < M && t > bv) { bv = t; ba = ro; }
          }
          const float ov = __shfl_xor(bv, 32, 64);
          const int oa = __shfl_xor(ba, 32, 64);
          if (ov > bv || (ov == bv && oa < ba)) { bv = ov; ba = oa; }
          gv[i][h] = bv; ga2[i][h] = ba;
        }
      const long long J = M / ns;
      auto put = [&](long long centre, float v, int a) {
        if (lk == 0 && col < N && centre < J) {
          px.ext[centre * N + col] = v * sgn;
          px.aext[centre * N + col] = a;
        }
      };
      const long long rbase = m0 + wm * 64;
      if (ns == 16) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) put((rbase + 32 * i + 16 * h) / 16, gv[i][h], ga2[i][h]);
      } else {
        float v32[2];
        int a32[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          v32[i] = gv[i][0]; a32[i] = ga2[i][0];
          if (gv[i][1] > v32[i]) { v32[i] = gv[i][1]; a32[i] = 16 + ga2[i][1]; }
        }
        if (ns == 32) {
          put((rbase) / 32, v32[0], a32[0]);
          put((rbase + 32) / 32, v32[1], a32[1]);
        } else {                                   // 64: the wave's 64 rows are one centre
          float v = v32[0];
          int a = a32[0];
          if (v32[1] > v) { v = v32[1]; a = 32 + a32[1]; }
          put(rbase / 64, v, a);
        }
      }
    }
  }
  // C/D layout of 32x32: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
  const bool next_stats = PRO == PRO_NONE && nx.Y != nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WTN + j * 32 + li;
    float s1 = 0.f, s2 = 0.f;
    float nsc = 0.f, nsh = 0.f, nmu = 0.f, nis = 0.f;
    if (next_stats && col < N) {
      nsc = nx.scale[col]; nsh = nx.shift[col]; nmu = nx.mean[col]; nis = nx.invstd[col];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float ny[PRO == PRO_NONE ? 16 : 1];
      if (next_stats) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const long long row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
          ny[PRO == PRO_NONE ? e : 0] = (row < M && col < N) ? nx.Y[row * (long long)N + col] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long long row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        const float v = acc[i][j][e];
        if (row < M && col < N) {
          Y[row * ldy + col] = v;
          if (next_stats) {
            const float y = ny[PRO == PRO_NONE ? e : 0];
            float dz = v;
            if (nx.relu && !(y * nsc + nsh > 0.f)) dz = 0.f;
            s1 += dz;
            s2 += dz * ((y - nmu) * nis);
          } else {
            s1 += v;
            s2 += v * v;
          }
        }
      }
    }
    if (partial != nullptr) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lk == 0) {
        s_stat[0][wm][wn * WTN + j * 32 + li] = s1;
        s_stat[1][wm][wn * WTN + j * 32 + li] = s2;
      }
    }
  }
  X3_STAMP();                  // epilogue stores issued
  if (prof_on) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    X3_STAMP();                // ... and acknowledged
    pr[63] = nstamp;
  }
  if (partial != nullptr) {
    __syncthreads();
    for (int c = tid; c < BN; c += 256) {
      if (n0 + c < N) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s1 += s_stat[0][w][c]; s2 += s_stat[1][w][c]; }
        float *p = partial + bx * 2 * N;
        p[n0 + c] = s1;
        p[N + n0 + c] = s2;
      }
    }
  }
}

// s2c_gemm_set_c64(0): the N > 64 problems stay on the 32-k-slice kernel (A/B in the tests)
static int g_c64 = 1;
static bool c64_on() { return g_c64 != 0; }

// s2c_gemm_set_c64_narrow(0): always 128 x 128 tiles
static int g_c64_narrow = 1;
static bool c64_narrow_on() { return g_c64_narrow != 0; }

template <int PRO>
int launch_c64(long long M, int N, int K, const float *A, int lda, const float *W, int ldw,
               const float *pscale, const float *pshift, const GatherArgs &ga, float *Y, int ldy,
               float *partial, hipStream_t st, const EpiArgs &ep, float *side = nullptr,
               int ld_side = 0, const NextStats &nx = NextStats(), const PoolExt &px = PoolExt()) {
  static int attr_state[64];                   // per device: 0 unknown, 1 ok, -1 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (attr_state[dev] == 0)
    attr_state[dev] = (hipFuncSetAttribute((const void *)rows_gemm_c64_kernel<PRO, 2, 2, 2, 2>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)C64_LDS_BYTES) == hipSuccess &&
                       hipFuncSetAttribute((const void *)rows_gemm_c64_kernel<PRO, 4, 1, 1, 1>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)C64_LDS_BYTES) == hipSuccess) ? 1 : -1;
  if (attr_state[dev] < 0) {
    (void)hipGetLastError();
    return -2;                                 // not taken: the 32-k-slice kernel runs instead
  }
  const long long nbx = (M + 127) / 128;
  // 128 x 32 tiles while 128 x 128 ones cover the chip once or less (not for the pooled epilogues)
  const bool narrow = c64_narrow_on() && nbx * ((N + 127) / 128) <= 192 && px.ext == nullptr &&
                      !(ep.mean != nullptr && ep.pool_ns != 0);
  const long long nby = narrow ? (N + 31) / 32 : (N + 127) / 128;
  dim3 grid((unsigned)(8 * ((nbx + 7) / 8) * nby));
  if (narrow)
    hipLaunchKernelGGL((rows_gemm_c64_kernel<PRO, 4, 1, 1, 1>), grid, dim3(256),
                       (128 + 32) * C64_LD * sizeof(float), st, M, N, K, A, lda, W, ldw, pscale,
                       pshift, ga, Y, ldy, partial, ep, side, ld_side, nx, px);
  else
    hipLaunchKernelGGL((rows_gemm_c64_kernel<PRO, 2, 2, 2, 2>), grid, dim3(256), C64_LDS_BYTES, st,
                       M, N, K, A, lda, W, ldw, pscale, pshift, ga, Y, ldy, partial, ep, side,
                       ld_side, nx, px);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_rows_gemm(c64) launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

template <int PRO>
int launch(long long M, int N, int K, const float *A, int lda, const float *W, int ldw,
           const float *pscale, const float *pshift, const GatherArgs &ga, float *Y,
           int ldy, float *partial, hipStream_t st, const EpiArgs &ep = EpiArgs()) {
  const int avec = A && ((lda & 3) == 0) && (((uintptr_t)A & 15) == 0);
  const int wvec = ((ldw & 3) == 0) && (((uintptr_t)W & 15) == 0);
  if (N <= 64) {
    dim3 grid((unsigned)((M + 255) / 256), 1);
    hipLaunchKernelGGL((rows_gemm_kernel<4, 1, PRO>), grid, dim3(256), 0, st, M, N, K, A,
                       lda, W, ldw, pscale, pshift, ga, Y, ldy, partial, avec, wvec, ep);
  } else {
    dim3 grid((unsigned)((M + 127) / 128), (unsigned)((N + 127) / 128));
    hipLaunchKernelGGL((rows_gemm_kernel<2, 2, PRO>), grid, dim3(256), 0, st, M, N, K, A,
                       lda, W, ldw, pscale, pshift, ga, Y, ldy, partial, avec, wvec, ep);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_rows_gemm launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}


template <int WM, int WN>
constexpr size_t x3_lds_bytes() {
  return x3_tile_bytes(WM, WN) + sizeof(float) * 2 * WM * 64 * WN;
}

template <int PRO>
int launch_x3(long long M, int N, int K, const float *A, int lda, const float *W, int ldw,
              const float *pscale, const float *pshift, const GatherArgs &ga, float *Y,
              int ldy, float *partial, hipStream_t st, const EpiArgs &ep = EpiArgs(),
              const BwdArgs &bw = BwdArgs()) {
  // (the BatchNorm-backward prologue keeps a third operand in flight: 27 registers over the
  // budget of two workgroups per CU in the 64-k-chunk kernel, 168 vs 152 us at (262144,128,128))
  if constexpr (PRO != PRO_BNBWD) {
    if (N > 64 && c64_on()) {
      const int rc = launch_c64<PRO>(M, N, K, A, lda, W, ldw, pscale, pshift, ga, Y, ldy, partial,
                                     st, ep);
      if (rc != -2) return rc;
    }
  }
  const int avec = A && ((lda & 3) == 0) && (((uintptr_t)A & 15) == 0);
  const int wvec = ((ldw & 3) == 0) && (((uintptr_t)W & 15) == 0);
  static bool attr_dev[64] = {};               // per device (the attribute is the device's)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  if (!attr_dev[dev]) {
    if (hipFuncSetAttribute((const void *)rows_gemm_x3_kernel<4, 1, PRO>,
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)x3_lds_bytes<4, 1>()) != hipSuccess ||
        hipFuncSetAttribute((const void *)rows_gemm_x3_kernel<2, 2, PRO>,
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)x3_lds_bytes<2, 2>()) != hipSuccess)
      return -1;
    attr_dev[dev] = true;
  }
  const size_t lds41 = x3_lds_bytes<4, 1>(), lds22 = x3_lds_bytes<2, 2>();
  if (N <= 64) {
    dim3 grid((unsigned)((M + 255) / 256), 1);
    hipLaunchKernelGGL((rows_gemm_x3_kernel<4, 1, PRO>), grid, dim3(256), lds41,
                       st, M, N, K, A, lda, W, ldw, pscale, pshift, ga, Y, ldy, partial, avec,
                       wvec, ep, bw);
  } else {
    dim3 grid((unsigned)((M + 127) / 128), (unsigned)((N + 127) / 128));
    hipLaunchKernelGGL((rows_gemm_x3_kernel<2, 2, PRO>), grid, dim3(256), lds22,
                       st, M, N, K, A, lda, W, ldw, pscale, pshift, ga, Y, ldy, partial, avec,
                       wvec, ep, bw);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_rows_gemm(bf16x3) launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// 0: exact fp32 MFMA chain, 1: bf16x3 split (default; S2C_GEMM_SPLIT=0 turns it off)
static int g_gemm_split = -1;
static bool use_split() {
  if (g_gemm_split < 0) {
    const char *e = getenv("S2C_GEMM_SPLIT");
    g_gemm_split = e ? atoi(e) : 1;
  }
  return g_gemm_split != 0;
}

}  // namespace

// streaming variant for the tall SA1-type layers (s2c_gemm2.hip); -2 = shape not taken
extern "C" int s2c_rows_stream_gemm(long long M, int N, int K, const float *A, int lda,
                                    const float *W, int ldw, float *Y, int ldy, float *partial,
                                    int partial_rows, void *stream);
extern "C" int s2c_sa_gather_stream_gemm(int b, int n, int m, int ns, int C,
                                         long long feat_row_stride, long long feat_batch_stride,
                                         float radius, int normalize, const float *xyz,
                                         const float *new_xyz, const float *feats, const int *idx,
                                         int N, const float *W, int ldw, float *Y, int ldy,
                                         float *partial, int partial_rows, void *stream);

extern "C" int s2c_rows_stream_gemm_bn_eval(long long M, int N, int K, const float *A, int lda,
                                            const float *W, int ldw, const float *gamma,
                                            const float *beta, const float *mean, const float *var,
                                            float eps, int relu, int pool_ns, float *out, int ldo,
                                            void *stream);
extern "C" int s2c_sa_gather_stream_gemm_bn_eval(int b, int n, int m, int ns, int C,
                                                 long long feat_row_stride,
                                                 long long feat_batch_stride, float radius,
                                                 int normalize, const float *xyz,
                                                 const float *new_xyz, const float *feats,
                                                 const int *idx, int N, const float *W, int ldw,
                                                 const float *gamma, const float *beta,
                                                 const float *mean, const float *var, float eps,
                                                 int relu, int pool_ns, float *out, int ldo,
                                                 void *stream);

extern "C" int s2c_rows_gemm_blocks(long long M, int N) {
  const int BM = N <= 64 ? 256 : 128;
  return (int)((M + BM - 1) / BM);
}

// Y = pro(A) W^T (+ column-statistics partials).  pscale/pshift NULL: pro = id,
// else BN+ReLU of the operand.  partial NULL: no statistics; else
// s2c_rows_gemm_blocks(M,N) * 2N floats (reduce with s2c_bn_finalize_partials).
extern "C" int s2c_rows_gemm(long long M, int N, int K, const float *A, int lda,
                             const float *W, int ldw, const float *pscale,
                             const float *pshift, float *Y, int ldy,
                             float *partial, void *stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !Y || lda < K || ldw < K) {
    fprintf(stderr, "s2c_rows_gemm: bad arguments\n");
    return -1;
  }
  GatherArgs ga = {};
  if (pscale != nullptr) {
    if (((uintptr_t)pscale & 15) || ((uintptr_t)pshift & 15)) return -1;
    if (use_split())
      return launch_x3<PRO_BNRELU>(M, N, K, A, lda, W, ldw, pscale, pshift, ga, Y, ldy,
                                   partial, (hipStream_t)stream);
    return launch<PRO_BNRELU>(M, N, K, A, lda, W, ldw, pscale, pshift, ga, Y, ldy,
                              partial, (hipStream_t)stream);
  }
  if (use_split()) {
    const int rc = s2c_rows_stream_gemm(M, N, K, A, lda, W, ldw, Y, ldy, partial,
                                        s2c_rows_gemm_blocks(M, N), stream);
    if (rc != -2) return rc;
    return launch_x3<PRO_NONE>(M, N, K, A, lda, W, ldw, nullptr, nullptr, ga, Y, ldy, partial,
                               (hipStream_t)stream);
  }
  return launch<PRO_NONE>(M, N, K, A, lda, W, ldw, nullptr, nullptr, ga, Y, ldy, partial,
                          (hipStream_t)stream);
}

// First set-abstraction layer with the ball-query grouping fused into the operand
// load: Y[(b,j,s), :] = [ (xyz[b,idx]-new_xyz[b,j]) (/radius) | feats[b,idx,:] ] W^T.
// K = 3 + C.  feats may be NULL when C == 0.
extern "C" int s2c_sa_gather_gemm(int b, int n, int m, int ns, int C,
                                  long long feat_row_stride,
                                  long long feat_batch_stride, float radius,
                                  int normalize, const float *xyz,
                                  const float *new_xyz, const float *feats,
                                  const int *idx, int N, const float *W, int ldw,
                                  float *Y, int ldy, float *partial, void *stream) {
  const long long M = (long long)b * m * ns;
  const int K = 3 + C;
  if (M <= 0 || N <= 0 || !xyz || !new_xyz || !idx || !W || !Y || ldw < K ||
      (C > 0 && !feats)) {
    fprintf(stderr, "s2c_sa_gather_gemm: bad arguments\n");
    return -1;
  }
  GatherArgs ga;
  ga.xyz = xyz; ga.new_xyz = new_xyz; ga.feats = feats; ga.idx = idx;
  ga.frs = feat_row_stride; ga.fbs = feat_batch_stride;
  ga.n = n; ga.m = m; ga.ns = ns; ga.radius = radius; ga.normalize = normalize;
  // wide first layers (N > 64) up to 262144 rows (SA2) run on the
  // 64-k-chunk kernel, not the streaming one (its 4-wave N = 128 configuration keeps 32 KB in
  // flight per CU: 135 vs 122 us at SA2); 0 = the streaming kernel wherever it takes the shape
  const long long c64_rows = 262144;
  if (use_split() && !(N > 64 && M <= c64_rows && c64_on())) {
    const int rc = s2c_sa_gather_stream_gemm(b, n, m, ns, C, feat_row_stride, feat_batch_stride,
                                             radius, normalize, xyz, new_xyz, feats, idx, N, W,
                                             ldw, Y, ldy, partial, s2c_rows_gemm_blocks(M, N),
                                             stream);
    if (rc != -2) return rc;
  }
  if (use_split())
    return launch_x3<PRO_GATHER>(M, N, K, nullptr, K, W, ldw, nullptr, nullptr, ga, Y, ldy,
                                 partial, (hipStream_t)stream);
  return launch<PRO_GATHER>(M, N, K, nullptr, K, W, ldw, nullptr, nullptr, ga, Y, ldy,
                            partial, (hipStream_t)stream);
}

// Backward of one BatchNorm(+ReLU) + linear layer in one pass over (dA, Y):
//   dY = bn_relu_backward(dA, Y)           (side output, M x C; the arithmetic of s2c_bn_relu_bwd)
//   dX = dY Wt^T,  Wt = W^T (N x C row-major, N = the layer's input channels)
// coef (3 C floats) comes from s2c_bn_relu_bwd_stats.  C % 4 == 0, all pointers 16-byte
// aligned.  Needs the bf16x3 GEMM (returns -2 when S2C_GEMM_SPLIT=0: use the separate passes).
extern "C" int s2c_bn_bwd_gemm(long long M, int C, int N, const float *dA, const float *Y,
                               const float *scale, const float *shift, const float *mean,
                               const float *invstd, const float *coef, int relu,
                               const float *Wt, int ldw, float *dY, float *dX, int ldx,
                               void *stream) {
  if (!use_split()) return -2;
  if (M <= 0 || C <= 0 || (C & 3) || N <= 0 || !dA || !Y || !scale || !shift || !mean ||
      !invstd || !coef || !Wt || !dX || ldw < C || ldx < N ||
      (((uintptr_t)dA | (uintptr_t)Y | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean |
        (uintptr_t)invstd | (uintptr_t)coef | (uintptr_t)dY) & 15)) {
    fprintf(stderr, "s2c_bn_bwd_gemm: bad arguments\n");
    return -1;
  }
  GatherArgs ga = {};
  BwdArgs bw = {Y, scale, shift, mean, invstd, coef, dY, relu, nullptr, nullptr, nullptr, nullptr,
                nullptr, 0};
  return launch_x3<PRO_BNBWD>(M, N, C, dA, C, Wt, ldw, nullptr, nullptr, ga, dX, ldx, nullptr,
                              (hipStream_t)stream, EpiArgs(), bw);
}

// The same with the statistics half of the PREVIOUS layer's BatchNorm backward out of the
// epilogue (BwdArgs::nY): nY (M x N, contiguous) = that layer's pre-activations, nscale / nshift
// / nmean / ninvstd its N per-channel vectors, npartial = s2c_rows_gemm_blocks(M, N) x 2N floats
// for s2c_bn_bwd_finalize_partials.  dX must be contiguous (ldx == N).
extern "C" int s2c_bn_bwd_gemm_next_stats(long long M, int C, int N, const float *dA,
                                          const float *Y, const float *scale, const float *shift,
                                          const float *mean, const float *invstd,
                                          const float *coef, int relu, const float *Wt, int ldw,
                                          float *dY, float *dX, int ldx, const float *nY,
                                          const float *nscale, const float *nshift,
                                          const float *nmean, const float *ninvstd, int nrelu,
                                          float *npartial, void *stream) {
  if (!use_split()) return -2;
  if (M <= 0 || C <= 0 || (C & 3) || N <= 0 || !dA || !Y || !scale || !shift || !mean ||
      !invstd || !coef || !Wt || !dX || ldw < C || ldx != N || !nY || !nscale || !nshift ||
      !nmean || !ninvstd || !npartial ||
      (((uintptr_t)dA | (uintptr_t)Y | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean |
        (uintptr_t)invstd | (uintptr_t)coef | (uintptr_t)dY) & 15)) {
    fprintf(stderr, "s2c_bn_bwd_gemm_next_stats: bad arguments\n");
    return -1;
  }
  GatherArgs ga = {};
  BwdArgs bw = {Y, scale, shift, mean, invstd, coef, dY, relu, nY, nscale, nshift, nmean, ninvstd,
                nrelu};
  return launch_x3<PRO_BNBWD>(M, N, C, dA, C, Wt, ldw, nullptr, nullptr, ga, dX, ldx, npartial,
                              (hipStream_t)stream, EpiArgs(), bw);
}

// Diagnostics: phase stamps of workgroup `block` (wave 0) of every following bf16x3 GEMM
// launch go to prof[0..62], prof[63] = number of stamps.  prof == NULL switches it off.
extern "C" int s2c_gemm_set_profile(long long *prof, int block) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), &prof, sizeof(prof)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof_block), &block, sizeof(block)) != hipSuccess) return -1;
  return 0;
}

/* 1 (default): problems with N > 64 on rows_gemm_c64_kernel (K in
 * 64-chunks of fp32 in LDS), 0: on the 32-k-slice kernel.  Returns the previous setting. */
extern "C" int s2c_gemm_set_c64_narrow(int on) {
  const int old = c64_narrow_on() ? 1 : 0;
  g_c64_narrow = on ? 1 : 0;
  return old;
}

extern "C" int s2c_gemm_set_c64(int on) {
  const int old = c64_on() ? 1 : 0;
  g_c64 = on ? 1 : 0;
  return old;
}

// s2c_rows_gemm_bn_relu_side (s2c_gemm2.hip) for the shapes its streaming kernel leaves: N > 64
// on the 64-k-chunk kernel, the activated operand written by the column-block-0 workgroups
// while they stage it.  -2: not taken (N <= 64, split products or the kernel switched off).
extern "C" int s2c_rows_gemm_c64_bn_relu_side(long long M, int N, int K, const float *A, int lda,
                                              const float *scale, const float *shift,
                                              float *side, int ld_side, const float *W, int ldw,
                                              float *Y, int ldy, float *partial, void *stream) {
  if (!use_split() || !c64_on() || N <= 64) return -2;
  if (M <= 0 || K <= 0 || !A || !W || !Y || !scale || !shift || lda < K || ldw < K ||
      (side && ld_side < K))
    return -1;
  GatherArgs ga = {};
  return launch_c64<PRO_BNRELU>(M, N, K, A, lda, W, ldw, scale, shift, ga, Y, ldy, partial,
                                (hipStream_t)stream, EpiArgs(), side, ld_side);
}
// Y (M x N, contiguous) = A W^T on the 64-k-chunk kernel with the column sums of the BatchNorm
// backward whose upstream gradient Y is (nY etc.: that layer, N channels) in `npartial`
// (s2c_rows_gemm_blocks(M, N) rows, for s2c_bn_bwd_finalize_partials).  -2: not taken (N <= 64).
extern "C" int s2c_rows_gemm_next_stats(long long M, int N, int K, const float *A, int lda,
                                        const float *W, int ldw, float *Y, const float *nY,
                                        const float *nscale, const float *nshift,
                                        const float *nmean, const float *ninvstd, int nrelu,
                                        float *npartial, void *stream) {
  if (!use_split() || !c64_on() || N <= 64) return -2;
  if (M <= 0 || K <= 0 || !A || !W || !Y || lda < K || ldw < K || !nY || !nscale || !nshift ||
      !nmean || !ninvstd || !npartial)
    return -1;
  GatherArgs ga = {};
  const NextStats nx = {nY, nscale, nshift, nmean, ninvstd, nrelu};
  return launch_c64<PRO_NONE>(M, N, K, A, lda, W, ldw, nullptr, nullptr, ga, Y, N, npartial,
                              (hipStream_t)stream, EpiArgs(), nullptr, 0, nx);
}

// s2c_rows_gemm_pool_raw (s2c_gemm2.hip) for the shapes its streaming kernel leaves, with Y
// written (the materialised backward reads it): N > 64 on the 64-k-chunk kernel.  scale == NULL:
// plain operand.  -2: not taken.
extern "C" int s2c_rows_gemm_c64_pool_ext(long long M, int N, int K, const float *A, int lda,
                                          const float *scale, const float *shift, float *side,
                                          int ld_side, const float *W, int ldw, int pool_ns,
                                          const float *gamma, float *ext, int *aext, float *Y,
                                          int ldy, float *partial, void *stream) {
  if (!use_split() || !c64_on() || N <= 64 || !Y) return -2;
  if (M <= 0 || K <= 0 || !A || !W || !ext || !aext || lda < K || ldw < K ||
      !(pool_ns == 16 || pool_ns == 32 || pool_ns == 64) || M % pool_ns || (side && ld_side < K))
    return -1;
  GatherArgs ga = {};
  const PoolExt px = {ext, aext, gamma, pool_ns};
  if (scale != nullptr)
    return launch_c64<PRO_BNRELU>(M, N, K, A, lda, W, ldw, scale, shift, ga, Y, ldy, partial,
                                  (hipStream_t)stream, EpiArgs(), side, ld_side, NextStats(), px);
  return launch_c64<PRO_NONE>(M, N, K, A, lda, W, ldw, nullptr, nullptr, ga, Y, ldy, partial,
                              (hipStream_t)stream, EpiArgs(), nullptr, 0, NextStats(), px);
}

extern "C" int s2c_rows_gemm_c64_supported(long long M, int N, int K) {
  return use_split() && c64_on() && M > 0 && N > 64 && K > 0;
}

/* 1: bf16x3 split products (default), 0: exact fp32 MFMA chain.  Returns the previous
 * setting.  (Also: environment S2C_GEMM_SPLIT read at first use.) */
extern "C" int s2c_gemm_set_split(int on) {
  const int old = use_split() ? 1 : 0;
  g_gemm_split = on ? 1 : 0;
  return old;
}

static bool bad_epi(long long M, int pool_ns, const float *mean, const float *var, float *out) {
  return !mean || !var || !out ||
         !(pool_ns == 0 || ((pool_ns == 16 || pool_ns == 32 || pool_ns == 64) && M % pool_ns == 0));
}

// Inference layer in one launch: out = [max over groups of pool_ns rows of]
// relu?( (A W^T) * gamma/sqrt(var+eps) + beta - mean*gamma/sqrt(var+eps) ).
// out is (M x N) or (M/pool_ns x N), row stride ldo.
extern "C" int s2c_rows_gemm_bn_eval(long long M, int N, int K, const float *A, int lda,
                                     const float *W, int ldw, const float *gamma,
                                     const float *beta, const float *mean, const float *var,
                                     float eps, int relu, int pool_ns, float *out, int ldo,
                                     void *stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W || lda < K || ldw < K ||
      bad_epi(M, pool_ns, mean, var, out)) {
    fprintf(stderr, "s2c_rows_gemm_bn_eval: bad arguments\n");
    return -1;
  }
  GatherArgs ga = {};
  EpiArgs ep = {gamma, beta, mean, var, eps, relu, pool_ns, out, ldo};
  if (use_split()) {
    const int rc = s2c_rows_stream_gemm_bn_eval(M, N, K, A, lda, W, ldw, gamma, beta, mean, var,
                                                eps, relu, pool_ns, out, ldo, stream);
    if (rc != -2) return rc;
  }
  if (use_split())
    return launch_x3<PRO_NONE>(M, N, K, A, lda, W, ldw, nullptr, nullptr, ga, out, ldo, nullptr,
                               (hipStream_t)stream, ep);
  return launch<PRO_NONE>(M, N, K, A, lda, W, ldw, nullptr, nullptr, ga, out, ldo, nullptr,
                          (hipStream_t)stream, ep);
}

// the same with the ball-query grouping fused into the operand load
extern "C" int s2c_sa_gather_gemm_bn_eval(int b, int n, int m, int ns, int C,
                                          long long feat_row_stride,
                                          long long feat_batch_stride, float radius,
                                          int normalize, const float *xyz,
                                          const float *new_xyz, const float *feats,
                                          const int *idx, int N, const float *W, int ldw,
                                          const float *gamma, const float *beta,
                                          const float *mean, const float *var, float eps,
                                          int relu, int pool_ns, float *out, int ldo,
                                          void *stream) {
  const long long M = (long long)b * m * ns;
  const int K = 3 + C;
  if (M <= 0 || N <= 0 || !xyz || !new_xyz || !idx || !W || ldw < K || (C > 0 && !feats) ||
      bad_epi(M, pool_ns, mean, var, out)) {
    fprintf(stderr, "s2c_sa_gather_gemm_bn_eval: bad arguments\n");
    return -1;
  }
  GatherArgs ga;
  ga.xyz = xyz; ga.new_xyz = new_xyz; ga.feats = feats; ga.idx = idx;
  ga.frs = feat_row_stride; ga.fbs = feat_batch_stride;
  ga.n = n; ga.m = m; ga.ns = ns; ga.radius = radius; ga.normalize = normalize;
  EpiArgs ep = {gamma, beta, mean, var, eps, relu, pool_ns, out, ldo};
  if (use_split()) {
    const int rc = s2c_sa_gather_stream_gemm_bn_eval(b, n, m, ns, C, feat_row_stride,
                                                     feat_batch_stride, radius, normalize, xyz,
                                                     new_xyz, feats, idx, N, W, ldw, gamma, beta,
                                                     mean, var, eps, relu, pool_ns, out, ldo, stream);
    if (rc != -2) return rc;
  }
  if (use_split())
    return launch_x3<PRO_GATHER>(M, N, K, nullptr, K, W, ldw, nullptr, nullptr, ga, out, ldo,
                                 nullptr, (hipStream_t)stream, ep);
  return launch<PRO_GATHER>(M, N, K, nullptr, K, W, ldw, nullptr, nullptr, ga, out, ldo,
                            nullptr, (hipStream_t)stream, ep);
}
