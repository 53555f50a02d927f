// s2c_attn_scene.hip -- greedy decoding with num_locals = -1 (the reference's DEFAULT command line:
// scripts/train.py:322, benchmark/predict.py:249, scripts/eval.py:242): every proposal of a scene
// attends over ALL K proposals of that scene (models/caption_module.py:270-285 inside the loop of
// :502-592 with valid_prop_masks = object_masks, :536).
//
// The keys of a row are therefore its SCENE's objects: M = map_feat(bbox_feature) is (B, K, H), shared
// by the K rows of a scene, not (R, K, H).  The module's `_step` loop materialises the (R, K, H)
// broadcast sum, its tanh, the attend product, the masked softmax and the weighted sum as ATen ops --
// 268 M elements, ~8 passes over 1 GB per token at cfg3e (profiles/r05_cfg3e_locals_all_*: 24 ms of
// elementwise kernels + 14 ms of library GEMMs per batch).  Here, per token, ONE launch:
//   s[r, j] = wa . tanh(M[b(r), j, :] + q[r, :]) + ba      (invalid key: -1e30)
//   alpha[r, :] = softmax_j s[r, :];   att[r, :] = sum_j alpha[r, j] O[b(r), j, :]
// A workgroup = 4 waves x RW rows of one scene.  lane <-> key (64 keys per pass), the hidden index
// runs inside the lane: no cross-lane reduction per score.  M tiles (64 keys x 64 h) are staged in LDS
// and shared by the workgroup's rows (the 16 workgroups of a scene re-read them from L2); q and wa
// are LDS broadcasts.  The kernel is bound by the transcendental rate, not by memory: the tiles are
// staged as 2^{c m} and 2^{c q}, so an element costs one reciprocal (see the scores loop).
// att leaves as fp32 and / or as the bf16x3 planes s2c_planes_gemm consumes (models/greedy_fused.py).
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

namespace {

constexpr int AS_MAXK = 512;      // keys per scene
constexpr int AS_LD = 68;         // LDS row stride of the M tile (floats): 16-byte aligned, 4-bank skew

__device__ __forceinline__ float as_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

struct AsArgs {
  int R, rps, K, H, F;            // rows, rows per scene, keys per scene
  const float *M;                 // (B K, H)
  const float *valid;             // (B, K) 0/1 or NULL
  const float *O;                 // (B K, F)
  const float *q; int ldq;        // (R, H)
  const float *wa; float ba;
  float *alpha;                   // (R, K)
  float *att; int lda;            // (R, F) or NULL
  unsigned short *planes; long long pstride; int ldp, tiled;
};

// NWV waves x RW rows per workgroup.  (8 x 1 rather than 4 x 2 for the same 8 rows: two waves per SIMD
// cover each other's reciprocal / LDS latencies -- one wave per SIMD ran at a third of the VALU bound.)
template <int NWV, int RW>
__global__ __launch_bounds__(64 * NWV) void attn_scene_kernel(AsArgs a) {
  constexpr int RB = NWV * RW;                  // rows per workgroup
  constexpr int NT = 64 * NWV, MP = 1024 / NT;  // threads, M-tile float4 per thread
  __shared__ __attribute__((aligned(16))) float sM[64 * AS_LD];
  __shared__ __attribute__((aligned(16))) float sQ[RB][64];
  __shared__ __attribute__((aligned(16))) float sW[64];
  __shared__ float sA[RB][AS_MAXK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blocks_per_scene = (a.rps + RB - 1) / RB;
  const int b = blockIdx.x / blocks_per_scene;
  const int r0 = (blockIdx.x % blocks_per_scene) * RB;          // first row inside the scene
  const int K = a.K, H = a.H;
  const float *Mb = a.M + (size_t)b * K * H;
  const long long row_base = (long long)b * a.rps + r0;          // global row of local row 0
  const int nrow = a.rps - r0 < RB ? a.rps - r0 : RB;

  // ---- scores ---------------------------------------------------------------------------------
  // tanh(m + q) = 1 - 2 / (e^{2m} e^{2q} + 1): the tiles are staged as E_m = 2^{c m}, E_q = 2^{c q}
  // (c = 2 log2 e), so an element costs ONE transcendental (the reciprocal) instead of two -- the
  // kernel is bound by their quarter rate.  Exact as long as neither factor over- or underflows:
  // a tile with |m| or |q| > 20 (2^{+-58}: the product stays finite, and beyond |x| = 10 tanh is
  // +-1 in fp32 anyway) is re-staged raw and takes the two-transcendental formula instead.
  constexpr float C2 = 2.8853900817779268f;
  // tile t = (key block, h chunk); the NEXT tile's global loads are issued into registers before
  // the current tile is multiplied (the staging latency was exposed: one workgroup per tile loop,
  // two or three workgroups per CU)
  const int nhc = (H + 63) >> 6, nkb = (K + 63) >> 6, ntile = nhc * nkb;
  float4 pm[MP], pq;                             // this thread's pieces of a tile
  const int q_rr = tid >> 4, q_qd = tid & 15;    // its q piece (threads < RB * 16)
  auto fetch = [&](int t) {
    const int kb = (t / nhc) << 6, hc = (t % nhc) << 6;
#pragma unroll
    for (int p = 0; p < MP; ++p) {
      const int e = tid + NT * p;
      const int key = e >> 4, qd = e & 15;
      pm[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kb + key < K && hc + 4 * qd < H)
        pm[p] = *reinterpret_cast<const float4 *>(Mb + (size_t)(kb + key) * H + hc + 4 * qd);
    }
    pq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < RB * 16 && q_rr < nrow && hc + 4 * q_qd < H)
      pq = *reinterpret_cast<const float4 *>(a.q + (size_t)(row_base + q_rr) * a.ldq + hc + 4 * q_qd);
  };
  auto far = [](const float4 &v) {
    return !(fabsf(v.x) <= 20.f && fabsf(v.y) <= 20.f && fabsf(v.z) <= 20.f && fabsf(v.w) <= 20.f);
  };
  auto ex = [&](float4 v, bool raw) {
    if (!raw) {
      v.x = __builtin_amdgcn_exp2f(v.x * C2); v.y = __builtin_amdgcn_exp2f(v.y * C2);
      v.z = __builtin_amdgcn_exp2f(v.z * C2); v.w = __builtin_amdgcn_exp2f(v.w * C2);
    }
    return v;
  };
  auto stage = [&](bool raw) {
#pragma unroll
    for (int p = 0; p < MP; ++p) {
      const int e = tid + NT * p;
      *reinterpret_cast<float4 *>(sM + (e >> 4) * AS_LD + 4 * (e & 15)) = ex(pm[p], raw);
    }
    if (tid < RB * 16) *reinterpret_cast<float4 *>(&sQ[q_rr][4 * q_qd]) = ex(pq, raw);
  };
  float acc[RW];
  fetch(0);
  for (int t = 0; t < ntile; ++t) {
    const int kb = (t / nhc) << 6, hc = (t % nhc) << 6;
    if (hc == 0) {
#pragma unroll
      for (int i = 0; i < RW; ++i) acc[i] = 0.f;
    }
    __syncthreads();                             // the previous tile has been multiplied
    bool big = far(pq);
#pragma unroll
    for (int p = 0; p < MP; ++p) big = big || far(pm[p]);
    stage(false);
    if (tid < 16) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);       // wa = 0 beyond H: no contribution
      if (hc + 4 * tid < H) v = *reinterpret_cast<const float4 *>(a.wa + hc + 4 * tid);
      *reinterpret_cast<float4 *>(&sW[4 * tid]) = v;
    }
    big = __syncthreads_or(big ? 1 : 0) != 0;
    const float *mrow = sM + lane * AS_LD;
    if (!big) {
      if (t + 1 < ntile) fetch(t + 1);           // in flight under the products
#pragma unroll 4
      for (int h = 0; h < 64; h += 4) {
        const float4 m = *reinterpret_cast<const float4 *>(mrow + h);
        const float4 w = *reinterpret_cast<const float4 *>(&sW[h]);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
          const float4 qq = *reinterpret_cast<const float4 *>(&sQ[wave * RW + i][h]);
          const float tx = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_fmaf(m.x, qq.x, 1.0f)), 1.0f);
          const float ty = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_fmaf(m.y, qq.y, 1.0f)), 1.0f);
          const float tz = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_fmaf(m.z, qq.z, 1.0f)), 1.0f);
          const float tw = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_fmaf(m.w, qq.w, 1.0f)), 1.0f);
          acc[i] += w.x * tx + w.y * ty + w.z * tz + w.w * tw;
        }
      }
    } else {
      // the rare tile with huge pre-activations: raw values, the two-transcendental formula
      __syncthreads();
      stage(true);
      __syncthreads();
      if (t + 1 < ntile) fetch(t + 1);
      for (int h = 0; h < 64; h += 4) {
        const float4 m = *reinterpret_cast<const float4 *>(mrow + h);
        const float4 w = *reinterpret_cast<const float4 *>(&sW[h]);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
          const float4 qq = *reinterpret_cast<const float4 *>(&sQ[wave * RW + i][h]);
          acc[i] += w.x * as_tanh(m.x + qq.x) + w.y * as_tanh(m.y + qq.y) +
                    w.z * as_tanh(m.z + qq.z) + w.w * as_tanh(m.w + qq.w);
        }
      }
    }
    if (hc + 64 >= H) {                          // the key block's scores are complete
      const int key = kb + lane;
      const bool ok = key < K && (a.valid == nullptr || a.valid[(size_t)b * K + key] != 0.0f);
#pragma unroll
      for (int i = 0; i < RW; ++i)
        if (key < K) sA[wave * RW + i][key] = ok ? acc[i] + a.ba : -1e30f;
    }
  }
  __syncthreads();

  // ---- softmax over the K keys of each row (a wave owns its RW rows) ------------------------------
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int rr = wave * RW + i;
    float mx = -INFINITY;
    for (int j = lane; j < K; j += 64) mx = fmaxf(mx, sA[rr][j]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.f;
    for (int j = lane; j < K; j += 64) {
      const float e = expf(sA[rr][j] - mx);
      sA[rr][j] = e;
      sum += e;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float inv = 1.0f / sum;
    for (int j = lane; j < K; j += 64) {
      const float al = sA[rr][j] * inv;
      sA[rr][j] = al;
      if (rr < nrow) a.alpha[(size_t)(row_base + rr) * K + j] = al;
    }
  }
  // (each wave reads back only its own rows of sA: no barrier needed)

  // ---- att[r, :] = sum_j alpha[r, j] O[b, j, :]: lane <-> feature, the RW rows share the O loads ----
  const float *Ob = a.O + (size_t)b * K * a.F;
  for (int f0 = 0; f0 < a.F; f0 += 64) {
    const int f = f0 + lane;
    const bool fok = f < a.F;
    float out[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) out[i] = 0.f;
#pragma unroll 16
    for (int j = 0; j < K; ++j) {               // 16 L2 round trips in flight
      const float o = fok ? Ob[(size_t)j * a.F + f] : 0.f;
#pragma unroll
      for (int i = 0; i < RW; ++i) out[i] += sA[wave * RW + i][j] * o;
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int rr = wave * RW + i;
      if (rr >= nrow || !fok) continue;
      const long long r = row_base + rr;
      const float v = out[i];
      if (a.att != nullptr) a.att[(size_t)r * a.lda + f] = v;
      if (a.planes != nullptr) {      // the bf16x3 planes of the attended vector (s2c_planes.hip's operand)
        const __bf16 hi = (__bf16)v;
        const float r1 = v - (float)hi;
        const __bf16 mid = (__bf16)r1;
        const __bf16 lo = (__bf16)(r1 - (float)mid);
        const int r32 = (int)(r & 31);
        unsigned short *pp = a.planes + (a.tiled
            ? ((size_t)(r >> 5) * (a.ldp >> 4) + (f >> 4)) * 512 +
                  ((r32 * 2 + (((f >> 3) & 1) ^ ((r32 >> 3) & 1))) << 3) + (f & 7)
            : (size_t)r * a.ldp + f);
        pp[0] = __builtin_bit_cast(unsigned short, hi);
        pp[a.pstride] = __builtin_bit_cast(unsigned short, mid);
        pp[2 * a.pstride] = __builtin_bit_cast(unsigned short, lo);
      }
    }
  }
}

}  // namespace

// Scene-shared attention of the greedy decoder with num_locals = -1: R = B * rows_per_scene query
// rows, row r of scene b attends over that scene's K keys.  M (B K, H) = map_feat of the scene's
// objects, valid (B, K) 0/1 or NULL, O (B K, F) the attended features, q (R, H; row stride ldq),
// wa (H), ba; alpha (R, K) = softmax_j(wa . tanh(M[b, j] + q[r]) + ba), att (R, F; row stride lda; may
// be NULL) = sum_j alpha O, optionally also as bf16x3 planes (see s2c_attn_local_fwd_planes).
// K <= 512, H % 4 == 0, ldq % 4 == 0.  Forward only (evaluation).
extern "C" int s2c_attn_scene_fwd(int R, int rows_per_scene, int K, int H, int F, const float *M,
                                  const float *valid, const float *O, const float *q, int ldq,
                                  const float *wa, float ba, float *alpha, float *att, int lda,
                                  unsigned short *planes, long long pstride, int ldp, int tiled,
                                  void *stream) {
  if (R <= 0 || rows_per_scene <= 0 || R % rows_per_scene || K <= 0 || K > AS_MAXK || H <= 0 ||
      (H & 3) || (ldq & 3) || F <= 0 || !M || !O || !q || !wa || !alpha || (!att && !planes) ||
      (planes && (ldp < F || (tiled && (ldp & 15)))))
    return -1;
  AsArgs a;
  a.R = R; a.rps = rows_per_scene; a.K = K; a.H = H; a.F = F; a.M = M; a.valid = valid; a.O = O;
  a.q = q; a.ldq = ldq; a.wa = wa; a.ba = ba; a.alpha = alpha; a.att = att; a.lda = lda;
  a.planes = planes; a.pstride = pstride; a.ldp = ldp; a.tiled = tiled;
  const int B = R / rows_per_scene;
  // 8 rows per workgroup (8 waves x 1) while that gives at most ~2 workgroups per CU, else 16 (8 x 2)
  if (R <= 4096) {
    const int nb = B * ((rows_per_scene + 7) / 8);
    hipLaunchKernelGGL((attn_scene_kernel<8, 1>), dim3(nb), dim3(512), 0, (hipStream_t)stream, a);
  } else {
    const int nb = B * ((rows_per_scene + 15) / 16);
    hipLaunchKernelGGL((attn_scene_kernel<8, 2>), dim3(nb), dim3(512), 0, (hipStream_t)stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_attn_scene_fwd launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
