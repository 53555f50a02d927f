// s2c_mgemm.hip -- many SMALL fp32 GEMMs in one launch.
//
// The teacher-forced caption decoder (models/caption_module.py:428-500) runs R = batch rows (8) x
// T <= 31 steps: everything without a recurrence is hoisted out of the step loop into GEMMs whose
// one large dimension is R T = 240 -- the word / target / object projections in front of the loop,
// the classifier behind it (caption_module.py:252, 275, 472), and in the backward pass every weight
// gradient dW = G^T X (a (3500 x 512) .. (300 x 128) output from a 240-deep reduction), the input
// gradients of the classifier and of map_lang, and the attention's dO.  Thirty products of 0.01 to
// 0.9 GFLOP each: as library calls each is one launch of 6-48 us (the (240 x 300) x (300 x 128)
// product: 48 us) -- launch latency and a kernel selection tuned for other shapes, not work.
//
// Here a LIST of products is one launch: C_j = A_j B_j (+ bias_j) (+ C_j), every operand described
// by index maps instead of copies -- element (m, k) of A at A[ix(m, am) + ix(k, ak)] with
// ix(i, {div, hi, lo}) = (i / div) hi + (i % div) lo, so transposes, column blocks of a larger
// matrix and the (t, r) <-> (r, t) row orders of the decoder's (T, R, .) / (R, T, .) tensors are all
// just strides (no permute().contiguous(), no torch.cat of weights, no `out=` temporaries).
// 64 x 64 output tiles, 256 threads x (4 x 4) outputs, K in steps of 16 through LDS, plain fp32
// FMA chains in k order (these are exact-fp32 products; the bf16x3 MFMA kernels pay off from
// thousands of rows up).  The tiles of all jobs are numbered consecutively: a launch of ~1300
// workgroups fills the chip where a single 300 x 300 product has 25.  A product with few tiles and
// a long reduction (the classifier's input gradient: 240 x 512 from K = 3500) is cut into `ksplit`
// k ranges that add into a zeroed C with hardware float atomics.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

namespace {

constexpr int MG_T = 64, MG_K = 16, MG_LD = MG_T + 4;

__device__ __forceinline__ long long mg_ix(int i, const s2c_mgemm_axis &x) {
  return x.div > 0 ? (long long)(i / x.div) * x.hi + (long long)(i % x.div) * x.lo
                   : (long long)i * x.lo;
}

__global__ __launch_bounds__(256) void mgemm_kernel(s2c_mgemm_args a) {
  __shared__ __attribute__((aligned(16))) float As[MG_K][MG_LD], Bs[MG_K][MG_LD];
  // ---- which job, which tile ----
  int j = 0;
  while (j + 1 < a.n_jobs && (int)blockIdx.x >= a.job[j + 1].tile0) ++j;
  const s2c_mgemm_job &J = a.job[j];
  const int tiles_n = (J.N + MG_T - 1) / MG_T;
  const int S = J.ksplit > 1 ? J.ksplit : 1;        // k ranges of a tile run as S workgroups
  const int tile = ((int)blockIdx.x - J.tile0) / S, ks = ((int)blockIdx.x - J.tile0) % S;
  const int m0 = (tile / tiles_n) * MG_T, n0 = (tile % tiles_n) * MG_T;
  const int kper = ((J.K + S - 1) / S + MG_K - 1) / MG_K * MG_K;
  const int kbeg = ks * kper, kend = kbeg + kper < J.K ? kbeg + kper : J.K;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;

  // ---- load maps: along whichever index of an operand is contiguous in memory ----
  // k-fast (row-major A, or B given as W with k contiguous): thread -> (row tid / 4, 4 k's)
  // otherwise (m / n contiguous, or anything else):          thread -> (k tid / 16, 4 rows)
  const bool a_kfast = J.ak.div <= 0 && J.ak.lo == 1;
  const bool b_kfast = J.bk.div <= 0 && J.bk.lo == 1;
  long long a_row[4], b_row[4];       // offsets of the thread's rows (m resp. n part)
  if (a_kfast) {
    const int m = m0 + (tid >> 2);
    a_row[0] = mg_ix(m < J.M ? m : J.M - 1, J.am);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + (tid & 15) * 4 + q;
      a_row[q] = mg_ix(m < J.M ? m : J.M - 1, J.am);
    }
  }
  if (b_kfast) {
    const int n = n0 + (tid >> 2);
    b_row[0] = mg_ix(n < J.N ? n : J.N - 1, J.bn);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + (tid & 15) * 4 + q;
      b_row[q] = mg_ix(n < J.N ? n : J.N - 1, J.bn);
    }
  }

  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[i][q] = 0.f;

  for (int k0 = kbeg; k0 < kend; k0 += MG_K) {
    float ra[4], rb[4];
    if (a_kfast) {
      const int kq = k0 + (tid & 3) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) ra[q] = kq + q < kend ? J.A[a_row[0] + kq + q] : 0.f;
    } else {
      const int k = k0 + (tid >> 4);
      const long long ko = mg_ix(k < J.K ? k : J.K - 1, J.ak);
#pragma unroll
      for (int q = 0; q < 4; ++q) ra[q] = k < kend ? J.A[a_row[q] + ko] : 0.f;
    }
    if (b_kfast) {
      const int kq = k0 + (tid & 3) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) rb[q] = kq + q < kend ? J.B[b_row[0] + kq + q] : 0.f;
    } else {
      const int k = k0 + (tid >> 4);
      const long long ko = mg_ix(k < J.K ? k : J.K - 1, J.bk);
#pragma unroll
      for (int q = 0; q < 4; ++q) rb[q] = k < kend ? J.B[b_row[q] + ko] : 0.f;
    }
    __syncthreads();                               // the previous step's reads are done
    if (a_kfast) {
#pragma unroll
      for (int q = 0; q < 4; ++q) As[(tid & 3) * 4 + q][tid >> 2] = ra[q];
    } else {
      *reinterpret_cast<float4 *>(&As[tid >> 4][(tid & 15) * 4]) = make_float4(ra[0], ra[1], ra[2], ra[3]);
    }
    if (b_kfast) {
#pragma unroll
      for (int q = 0; q < 4; ++q) Bs[(tid & 3) * 4 + q][tid >> 2] = rb[q];
    } else {
      *reinterpret_cast<float4 *>(&Bs[tid >> 4][(tid & 15) * 4]) = make_float4(rb[0], rb[1], rb[2], rb[3]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MG_K; ++k) {
      const float4 av = *reinterpret_cast<const float4 *>(&As[k][ty * 4]);
      const float4 bv = *reinterpret_cast<const float4 *>(&Bs[k][tx * 4]);
      const float aa[4] = {av.x, av.y, av.z, av.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = __builtin_fmaf(aa[i], bb[q], acc[i][q]);
    }
  }

  // ---- epilogue: bias, accumulate, store (unit column stride) ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= J.M) continue;
    float *crow = J.C + mg_ix(m, J.cm);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + tx * 4 + q;
      if (n >= J.N) continue;
      float v = acc[i][q];
      if (J.bias != nullptr && ks == 0) v += J.bias[n];
      if (S > 1) {
        atomicAdd(crow + n, v);                      // C zeroed by the caller (or accumulate)
      } else {
        if (J.accumulate) v += crow[n];
        crow[n] = v;
      }
    }
  }
}

}  // namespace

extern "C" int s2c_mgemm(const s2c_mgemm_args *a, void *stream) {
  if (a == nullptr || a->n_jobs <= 0 || a->n_jobs > S2C_MGEMM_MAX_JOBS) return -1;
  s2c_mgemm_args loc = *a;
  int tiles = 0;
  for (int j = 0; j < loc.n_jobs; ++j) {
    s2c_mgemm_job &J = loc.job[j];
    if (J.M <= 0 || J.N <= 0 || J.K <= 0 || J.A == nullptr || J.B == nullptr || J.C == nullptr)
      return -1;
    J.tile0 = tiles;
    tiles += ((J.M + MG_T - 1) / MG_T) * ((J.N + MG_T - 1) / MG_T) * (J.ksplit > 1 ? J.ksplit : 1);
  }
  hipLaunchKernelGGL(mgemm_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, loc);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c: mgemm launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" long long s2c_mgemm_args_sizeof(void) { return (long long)sizeof(s2c_mgemm_args); }
