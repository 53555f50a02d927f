// s2c_gemm.hip -- hand-written fp32 MFMA GEMM for the shared-MLP layers of the
// point-major set-abstraction path:
//
//     Y[M x N] = pro(A)[M x K] * W^T            W is (N x K) row-major (conv weight)
//
// * pro(A) = A, or relu(A * scale[k] + shift[k]) -- the previous layer's
//   BatchNorm + ReLU applied while the A tile is staged into LDS, so the
//   activation tensor between two layers is never materialised;
// * epilogue: per-column sum and sum-of-squares of the Y tile written as partials
//   (one row of [sum | sumsq] per row-block) -- the BatchNorm batch statistics of
//   THIS layer without another pass over Y.
// fp32-in / fp32-accumulate `v_mfma_f32_32x32x2_f32` (exact f32 FMA chain, 157 TF
// peak on gfx950) keeps the 1e-4 parity budget; there is no TF32-like fast path on
// CDNA4 and bf16 is not parity-safe.
//
// Geometry: 256 threads = 4 waves, each wave owns a 64x64 tile (2x2 MFMA tiles,
// 64 accumulator registers).  N <= 64: waves stacked 4x1 (block = 256 rows x 64
// cols); otherwise 2x2 (block = 128 x 128).  K is walked in slices of 32 through
// LDS; tiles are stored K-major ([k][row], row stride +1) so that the transposing
// store and the MFMA operand reads (lane -> consecutive rows) are both
// bank-conflict free; the next slice is prefetched into registers while the
// current one feeds the matrix cores.  M is huge (up to 1e6 rows), so the grid has
// thousands of workgroups.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

using namespace s2c;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;

template <int WM, int WN>  // waves along M / N; block tile = (64*WM) x (64*WN)
__global__ __launch_bounds__(256) void rows_gemm_kernel(
    int M, int N, int K, const float *__restrict__ A, int lda,
    const float *__restrict__ W, int ldw, const float *__restrict__ pscale,
    const float *__restrict__ pshift, float *__restrict__ Y, int ldy,
    float *__restrict__ partial, int avec, int wvec) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int LDA_S = BM + 1, LDW_S = BN + 1;
  __shared__ float As[BK][LDA_S];
  __shared__ float Ws[BK][LDW_S];
  __shared__ float s_stat[2][WM][BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // staging map: thread -> (row = tid / 8 + 32*i, k-quad = tid % 8)
  const int sq = tid & 7, sr = tid >> 3;
  constexpr int AI = BM / 32, WI = BN / 32;
  float4 ra[AI], rw[WI];

  auto load_slice = [&](int k0) {
    const int k = k0 + sq * 4;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const long long row = m0 + sr + 32 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < M) {
        const float *p = A + row * lda + k;
        if (k + 3 < K) {
          if (avec) v = *reinterpret_cast<const float4 *>(p);
          else { v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3]; }
        } else {
          if (k < K) v.x = p[0];
          if (k + 1 < K) v.y = p[1];
          if (k + 2 < K) v.z = p[2];
        }
        if (pscale != nullptr) {
          if (k < K) v.x = fmaxf(v.x * pscale[k] + pshift[k], 0.f);
          if (k + 1 < K) v.y = fmaxf(v.y * pscale[k + 1] + pshift[k + 1], 0.f);
          if (k + 2 < K) v.z = fmaxf(v.z * pscale[k + 2] + pshift[k + 2], 0.f);
          if (k + 3 < K) v.w = fmaxf(v.w * pscale[k + 3] + pshift[k + 3], 0.f);
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
    for (int i = 0; i < AI; ++i) {
      const int r = sr + 32 * i;
      As[sq * 4 + 0][r] = ra[i].x; As[sq * 4 + 1][r] = ra[i].y;
      As[sq * 4 + 2][r] = ra[i].z; As[sq * 4 + 3][r] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int r = sr + 32 * i;
      Ws[sq * 4 + 0][r] = rw[i].x; Ws[sq * 4 + 1][r] = rw[i].y;
      Ws[sq * 4 + 2][r] = rw[i].z; Ws[sq * 4 + 3][r] = rw[i].w;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int li = lane & 31, lk = lane >> 5;
  const int arow = wm * 64 + li, wcol = wn * 64 + li;

  load_slice(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();           // previous slice fully consumed
    store_slice();
    __syncthreads();
    if (k0 + BK < K) load_slice(k0 + BK);   // prefetch under the MFMAs
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = As[kk + lk][arow], a1 = As[kk + lk][arow + 32];
      const float b0 = Ws[kk + lk][wcol], b1 = Ws[kk + lk][wcol + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
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

}  // namespace

extern "C" int s2c_rows_gemm_blocks(long long M, int N) {
  const int BM = N <= 64 ? 256 : 128;
  return (int)((M + BM - 1) / BM);
}

// Y = pro(A) W^T (+ column-statistics partials).  pscale/pshift NULL: pro = id.
// partial NULL: no statistics; else s2c_rows_gemm_blocks(M,N) * 2N floats,
// laid out exactly like the partials of s2c_bn_train_stats (see
// s2c_bn_finalize_partials).
extern "C" int s2c_rows_gemm(long long M, int N, int K, const float *A, int lda,
                             const float *W, int ldw, const float *pscale,
                             const float *pshift, float *Y, int ldy,
                             float *partial, void *stream) {
  if (M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K) {
    fprintf(stderr, "s2c_rows_gemm: bad arguments\n");
    return -1;
  }
  // 16-byte row loads need aligned rows; otherwise four 4-byte loads per quad
  const int avec = ((lda & 3) == 0) && (((uintptr_t)A & 15) == 0);
  const int wvec = ((ldw & 3) == 0) && (((uintptr_t)W & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  if (N <= 64) {
    dim3 grid((unsigned)((M + 255) / 256), 1);
    hipLaunchKernelGGL((rows_gemm_kernel<4, 1>), grid, dim3(256), 0, st, (int)M, N, K, A,
                       lda, W, ldw, pscale, pshift, Y, ldy, partial, avec, wvec);
  } else {
    dim3 grid((unsigned)((M + 127) / 128), (unsigned)((N + 127) / 128));
    hipLaunchKernelGGL((rows_gemm_kernel<2, 2>), grid, dim3(256), 0, st, (int)M, N, K, A,
                       lda, W, ldw, pscale, pshift, Y, ldy, partial, avec, wvec);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_rows_gemm launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
