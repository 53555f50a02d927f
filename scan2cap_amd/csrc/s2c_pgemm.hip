// s2c_pgemm.hip -- the per-POINT product of a set-abstraction stage's first layer,
//     P (M x N) = X (M x K) W^T,     X = the stage's input features, one row per point,
// on the exact fp32 matrix instruction (pointnet2/fused.py POINT_SPACE; the gathered rows are then
// P[idx] + W_x rel, csrc/s2c_sa.hip: sa_gather_add).  Reference: the feature columns of the first
// Conv2d of a SharedMLP over QueryAndGroup's output (pointnet2_modules.py:251-253,
// pytorch_utils.py:67-120, pointnet2_utils.py:347-359).
//
// Shapes: M = B n points (320 000 at SA1 of the cfg3 step, 4096 .. 16 384 behind it), K = 132 /
// 128 / 256 input channels, N = 64 / 128 outputs.  X rows may start at any 4-byte address (SA1
// reads the cloud's feature columns in place: rows of 3 + C floats).  The tiled kernel of
// s2c_gemm.hip runs these at 112 us (SA1) and 30 us (each of the four small ones: 128 workgroups of
// 128 x 128 for 256 CUs, 32-k slices with two barriers each); here
//   * `v_mfma_f32_32x32x2_f32`, every product and sum in fp32 (an fp32 FMA chain over k: the
//     arithmetic of the reference's own GEMM up to the order of the chain);
//   * 64-k chunks in LDS, rows padded to 68 floats: a lane fetches FOUR consecutive k of its row
//     with one ds_read_b128 (17 x 16 bytes per row: conflict-free) and feeds four MFMAs -- the
//     reduction index is free to be placed: a 32-k slice lies in LDS as [16 even k | 16 odd k], so
//     MFMA j of the slice takes k = 2 j from lanes 0-31 and 2 j + 1 from lanes 32-63, in k order.  That is the chain of the tiled
//     kernel's exact path (s2c_gemm.hip: rows_gemm_kernel), so the two agree BIT FOR BIT
//     (tests/test_point_space_gpu.py) -- a speed-up, not another rounding of the same product;
//   * the next chunk's global loads are issued before the MFMAs of the current one;
//   * 64 x 64 tiles when 128 x 128 ones would not fill the chip.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) PF4U { float x, y, z, w; };

constexpr int PG_KC = 64, PG_LD = 68;

struct PgArgs {
  long long M;
  int N, K;
  const float *A; long long lda;
  const float *W; int ldw;
  float *P; int ldp;
};

__device__ __forceinline__ float4 pg_load4(const float *__restrict__ row, int k, int K, bool ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok && k < K) {
    if (k + 3 < K) {
      const PF4U q = *reinterpret_cast<const PF4U *>(row + k);
      v = make_float4(q.x, q.y, q.z, q.w);
    } else {
      v.x = row[k];
      if (k + 1 < K) v.y = row[k + 1];
      if (k + 2 < K) v.z = row[k + 2];
    }
  }
  return v;
}

// workgroup = WM x WN waves (= 4), wave = TM x TN accumulator tiles of 32 x 32
// ODD: K % 4 != 0 (guarded 4-byte loads in the last chunk); otherwise every load is one 16-byte load
template <int TM, int TN, int WM, int WN, bool ODD>
__global__ __launch_bounds__(256) void point_gemm_kernel(PgArgs a) {
  static_assert(WM * WN == 4, "four waves");
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int PA = BM / 16, PW = BN / 16;                 // staging passes (16 rows each)
  extern __shared__ __attribute__((aligned(16))) float pg_lds[];
  float *As = pg_lds, *Ws = pg_lds + BM * PG_LD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lk = lane >> 5;
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int sr = tid >> 4, sk = (tid & 15) * 4;             // staging: row of a pass, first k

  // rows beyond M / N read a valid row instead (their results are never stored): every load of a
  // full chunk is unconditional -- a guarded load is a branch of its own to the compiler
  const float *arow[PA];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const long long r = m0 + p * 16 + sr;
    arow[p] = a.A + (r < a.M ? r : a.M - 1) * a.lda + sk;
  }
  const float *wrow[PW];
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const int n = n0 + p * 16 + sr;
    wrow[p] = a.W + (long long)(n < a.N ? n : a.N - 1) * a.ldw + sk;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

  const float *ap = As + (wm * 32 * TM + li) * PG_LD + lk * 16;
  const float *wp = Ws + (wn * 32 * TN + li) * PG_LD + lk * 16;
  const int sp = (sk >> 5) * 32 + ((sk & 31) >> 1);          // slice, 2 * quad-of-slice
  float *const as_st = As + sr * PG_LD + sp, *const ws_st = Ws + sr * PG_LD + sp;

  // One chunk of both operands in registers.  Last chunk of a K % 4 == 0 problem: quads beyond K
  // load the row's last quad and are zeroed on their way into LDS -- still no guarded load; a K
  // that is no multiple of 4 (ODD) takes guarded 4-byte loads in its last chunk.
  float4 ra[PA], rw[PW];
  bool keep;
#define PG_LOAD_CHUNK(K0)                                                                     \
  {                                                                                           \
    const int k0_ = (K0);                                                                     \
    if (ODD && k0_ + PG_KC > a.K) {                                                           \
      keep = true;                                                                            \
      _Pragma("unroll") for (int p = 0; p < PA; ++p)                                          \
        ra[p] = pg_load4(arow[p] - sk, k0_ + sk, a.K, true);                                  \
      _Pragma("unroll") for (int p = 0; p < PW; ++p)                                          \
        rw[p] = pg_load4(wrow[p] - sk, k0_ + sk, a.K, true);                                  \
    } else {                                                                                  \
      keep = k0_ + sk < a.K;                                                                  \
      const int kk_ = keep ? k0_ : a.K - 4 - sk;                                              \
      _Pragma("unroll") for (int p = 0; p < PA; ++p) {                                        \
        const PF4U q_ = *reinterpret_cast<const PF4U *>(arow[p] + kk_);                       \
        ra[p] = make_float4(q_.x, q_.y, q_.z, q_.w);                                          \
      }                                                                                       \
      _Pragma("unroll") for (int p = 0; p < PW; ++p) {                                        \
        const PF4U q_ = *reinterpret_cast<const PF4U *>(wrow[p] + kk_);                       \
        rw[p] = make_float4(q_.x, q_.y, q_.z, q_.w);                                          \
      }                                                                                       \
    }                                                                                         \
  }

  const int kmain = a.K;
  PG_LOAD_CHUNK(0)
  for (int k0 = 0; k0 < kmain; k0 += PG_KC) {
    __syncthreads();                       // the previous chunk is consumed
    {
      // k = sk .. sk + 3 of the chunk: even k's to [2 sq, 2 sq + 1] of their 32-k slice, odd k's to
      // 16 + [2 sq, 2 sq + 1] (lanes 0-31 of an MFMA read the even half, lanes 32-63 the odd one)
#pragma unroll
      for (int p = 0; p < PA; ++p) {
        float *d = as_st + p * 16 * PG_LD;
        *reinterpret_cast<float2 *>(d) = make_float2(keep ? ra[p].x : 0.f, keep ? ra[p].z : 0.f);
        *reinterpret_cast<float2 *>(d + 16) = make_float2(keep ? ra[p].y : 0.f, keep ? ra[p].w : 0.f);
      }
#pragma unroll
      for (int p = 0; p < PW; ++p) {
        float *d = ws_st + p * 16 * PG_LD;
        *reinterpret_cast<float2 *>(d) = make_float2(keep ? rw[p].x : 0.f, keep ? rw[p].z : 0.f);
        *reinterpret_cast<float2 *>(d + 16) = make_float2(keep ? rw[p].y : 0.f, keep ? rw[p].w : 0.f);
      }
    }
    __syncthreads();
    if (k0 + PG_KC < kmain) PG_LOAD_CHUNK(k0 + PG_KC)
#define PG_GROUP(G)                                                                           \
    {                                                                                         \
      float4 av[TM], bv[TN];                                                                  \
      _Pragma("unroll") for (int t = 0; t < TM; ++t)                                          \
        av[t] = *reinterpret_cast<const float4 *>(ap + t * 32 * PG_LD + ((G) >> 2) * 32 + ((G) & 3) * 4);             \
      _Pragma("unroll") for (int u = 0; u < TN; ++u)                                          \
        bv[u] = *reinterpret_cast<const float4 *>(wp + u * 32 * PG_LD + ((G) >> 2) * 32 + ((G) & 3) * 4);             \
      PG_MMA(av, bv)                                                                          \
    }
#define PG_MMA(AV, BV)                                                                        \
      _Pragma("unroll") for (int q = 0; q < 4; ++q)                                           \
        _Pragma("unroll") for (int t = 0; t < TM; ++t)                                        \
          _Pragma("unroll") for (int u = 0; u < TN; ++u) {                                    \
            const float x = q == 0 ? AV[t].x : q == 1 ? AV[t].y : q == 2 ? AV[t].z : AV[t].w; \
            const float y = q == 0 ? BV[u].x : q == 1 ? BV[u].y : q == 2 ? BV[u].z : BV[u].w; \
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[t][u], 0, 0, 0);       \
          }
    const int kleft = kmain - k0;
    if (kleft >= PG_KC) {                  // a full chunk: unrolled, the LDS reads run ahead
#pragma unroll
      for (int g = 0; g < PG_KC / 8; ++g) PG_GROUP(g)
    } else {
      const int ngrp = 4 * ((kleft + 31) / 32);   // whole 32-k slices (zero padded)
      for (int g = 0; g < ngrp; ++g) PG_GROUP(g)
    }
  }
#undef PG_LOAD_CHUNK
#undef PG_GROUP
#undef PG_MMA

  // C/D layout of 32x32: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const int col = n0 + (wn * TN + u) * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long long row = m0 + (wm * TM + t) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        if (row < a.M && col < a.N) a.P[row * a.ldp + col] = acc[t][u][e];
      }
    }
}

template <int TM, int TN, int WM, int WN, bool ODD>
int pg_launch_v(const PgArgs &a, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int lds = (BM + BN) * PG_LD * (int)sizeof(float);
  static bool attr_done[64];                 // per device (the attribute is the device's)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_done[dev]) {
    if (hipFuncSetAttribute((const void *)point_gemm_kernel<TM, TN, WM, WN, ODD>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return -1;
    attr_done[dev] = true;
  }
  const dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)((a.N + BN - 1) / BN));
  hipLaunchKernelGGL((point_gemm_kernel<TM, TN, WM, WN, ODD>), grid, dim3(256), lds, st, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_point_gemm launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

template <int TM, int TN, int WM, int WN>
int pg_launch(const PgArgs &a, hipStream_t st) {
  return (a.K & 3) ? pg_launch_v<TM, TN, WM, WN, true>(a, st) : pg_launch_v<TM, TN, WM, WN, false>(a, st);
}

}  // namespace

extern "C" int s2c_point_gemm(long long M, int N, int K, const float *A, long long lda,
                              const float *W, int ldw, float *P, int ldp, void *stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !P || lda < K || ldw < K || ldp < N) {
    fprintf(stderr, "s2c_point_gemm: bad arguments\n");
    return -1;
  }
  PgArgs a = {M, N, K, A, lda, W, ldw, P, ldp};
  hipStream_t st = (hipStream_t)stream;
  const long long big = ((M + 127) / 128) * ((N + 127) / 128);
  if (N <= 64) {
    if ((M + 127) / 128 >= 512) return pg_launch<1, 2, 4, 1>(a, st);        // 128 x 64
    return pg_launch<1, 1, 2, 2>(a, st);                                    // 64 x 64
  }
  if (big >= 512) return pg_launch<2, 2, 2, 2>(a, st);                      // 128 x 128
  return pg_launch<1, 1, 2, 2>(a, st);
}
