// s2c_sa_fused.hip -- a WHOLE set-abstraction stage of the inference path in one kernel:
//
//   ball-query rows -> gather (xyz centred / normalised | features) -> 3 x [1x1 conv, frozen
//   BatchNorm, ReLU] -> max over the nsample rows of a centre        (pointnet2_modules.py:226-257)
//
// Nothing between the (B,N,3+C) cloud and the (B*m, N3) pooled features touches HBM: the
// reference materialises the (B,3+C,m,ns) grouped tensor and three (B,C,m,ns) activations, the
// per-layer path of s2c_gemm2.hip still writes and re-reads two (M x 64) activations (1 GB at
// SA1).  Fused contract of SURVEY 8(d): compulsory input + idx + pooled output.
//
// When it applies: all three weight matrices must stay resident in LDS as bf16x3 planes
// (6 B per element) next to one activation patch per wave -- SA1 with few input channels
// (BASELINE configs[1]: 3 + 4 -> 64 -> 64 -> 128: 80 KB of planes).  With the 128 multiview
// channels (135 x 64 alone is 55 KB) or the 128..256-wide later stages they do not, and
// the per-layer kernels run (s2c_sa_fused_eval_supported).
//
// One wave = one pipeline over 32-row tiles (no workgroup barrier after the weight staging):
//   layer 1  the 16-wide operand [features | centred xyz | 0] is built in REGISTERS from the
//            gathered point (one k16 step, 12 MFMAs);
//   layers 2, 3  the accumulator tile gets its BatchNorm + ReLU in registers, leaves through 4x4
//            DPP transposes as 16-byte LDS stores into the wave's 8 KB patch (two 32x32 fp32
//            chunks, XOR-swizzled like the ring chunks of s2c_gemm2.hip) and is read back in
//            MFMA operand order (one row per lane, ds_read_b128) -- 48 + 96 MFMAs;
//   pool     BatchNorm + ReLU + max over the lane's 16 rows, the two half-waves, and the two
//            tiles of a 64-row centre; one 4-byte store per (centre, column).
// Products are the same bf16x3 splits (6 plane products with i + j <= 2, fp32 accumulate, same
// term order) as every other rows GEMM: results equal the per-layer path up to the order of
// the k-walk of layer 1 (features first, coordinates last -- as the streaming gather GEMM).
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

using namespace s2c;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int FW = 8;                 // waves per workgroup (one workgroup per CU)
constexpr int N1P = 64, N2P = 64, N3P = 128;     // padded layer widths
constexpr int KB1 = 2, KB2 = 8, KB3 = 8;         // k-blocks of 8 (K1 <= 16, K2 = K3 = 64)
constexpr unsigned W1B = 3u * KB1 * N1P * 16u, W2B = 3u * KB2 * N2P * 16u, W3B = 3u * KB3 * N3P * 16u;
constexpr unsigned CO_B = (N1P + N2P + N3P) * 2u * 4u;   // scale | shift of the three layers
constexpr unsigned PATCH_B = 8192;                      // 32 rows x 64 floats, two chunks

struct FusedArgs {
  int b, n, m, ns, C;
  long long frs, fbs;
  float radius; int normalize;
  const float *xyz, *new_xyz, *feats;
  const int *idx;
  int N1, N2, N3;
  const float *W[3]; int ldw[3];
  const float *gamma[3], *beta[3], *mean[3], *var[3]; float eps[3];
  float *out; int ldo;
};

__device__ __forceinline__ void split2(f32x2 v, unsigned &hi, unsigned &mid, unsigned &lo) {
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
  const bf16x2 m = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
  const bf16x2 l = __builtin_convertvector(r2, bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  mid = __builtin_bit_cast(unsigned, m);
  lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ void split8(const float4 &va, const float4 &vb, bf16x8 (&pl)[3]) {
  uint4 h, m, l;
  split2((f32x2){va.x, va.y}, h.x, m.x, l.x);
  split2((f32x2){va.z, va.w}, h.y, m.y, l.y);
  split2((f32x2){vb.x, vb.y}, h.z, m.z, l.z);
  split2((f32x2){vb.z, vb.w}, h.w, m.w, l.w);
  pl[0] = __builtin_bit_cast(bf16x8, h);
  pl[1] = __builtin_bit_cast(bf16x8, m);
  pl[2] = __builtin_bit_cast(bf16x8, l);
}

// W (N x K, row stride ldw) -> planes [plane][k-block of 8][column] x 16 B, zero padded.
// perm_c >= 0: layer 1 -- logical k < perm_c is feature column 3 + k, then the 3 coordinates.
__device__ void stage_planes(unsigned char *wp, const float *W, int ldw, int N, int K, int NP, int KB,
                             int perm_c, int tid, int nthreads) {
  const int quads = KB * 2;
  for (int e = tid; e < NP * quads; e += nthreads) {
    const int n = e / quads, k0 = (e - n * quads) * 4;
    float w[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = k0 + c;
      int src = -1;
      if (perm_c >= 0) {
        if (k < perm_c) src = 3 + k; else if (k < perm_c + 3) src = k - perm_c;
      } else if (k < K) {
        src = k;
      }
      w[c] = (n < N && src >= 0) ? W[(long long)n * ldw + src] : 0.f;
    }
    unsigned h0, m0, l0, h1, m1, l1;
    split2((f32x2){w[0], w[1]}, h0, m0, l0);
    split2((f32x2){w[2], w[3]}, h1, m1, l1);
    unsigned char *d = wp + ((unsigned)(k0 >> 3) * NP + n) * 16u + (k0 & 7) * 2;
    *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(d + (unsigned)KB * NP * 16u) = make_uint2(m0, m1);
    *reinterpret_cast<uint2 *>(d + 2u * KB * NP * 16u) = make_uint2(l0, l1);
  }
}

// one k16 step: acc[j] += a (x) W[ks] for the NT column tiles (6 plane products, small terms first)
template <int NT>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[NT], const float4 &va, const float4 &vb,
                                          const unsigned char *wp, int KB, int NP, int ks, int li,
                                          int lk) {
  bf16x8 a[3];
  split8(va, vb, a);
  bf16x8 b[NT][3];
  const unsigned char *wk = wp + ((unsigned)(2 * ks + lk) * NP + li) * 16u;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      b[j][pl] = *reinterpret_cast<const bf16x8 *>(wk + (unsigned)pl * KB * NP * 16u + j * 512);
  constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int tt = 0; tt < 6; ++tt)
#pragma unroll
    for (int j = 0; j < NT; ++j)
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[tt]], b[j][TB[tt]], acc[j], 0, 0, 0);
}

// BatchNorm + ReLU on the accumulator tile (C/D layout: col = lane & 31, row = (e & 3) +
// 8 (e >> 2) + 4 (lane >> 5)), then through 4x4 DPP transposes into the wave's patch as the
// [row][k] fp32 chunk layout the operand reads expect (quad q of row r at q ^ ((r >> 1) & 7)).
template <int NT>
__device__ __forceinline__ void to_patch(const f32x16 (&acc)[NT], const float *sc, const float *sh,
                                         unsigned char *patch, int lane) {
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const float s = sc[32 * j + li], t = sh[32 * j + li];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float a4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a4[i] = fmaxf(acc[j][4 * g + i] * s + t, 0.f);
      quad_transpose(a4, lane);
      const int row = 8 * g + 4 * lk + (lane & 3);
      const int q = li >> 2;                                 // quad of 4 columns inside chunk j
      *reinterpret_cast<float4 *>(patch + j * 4096 + row * 128 + ((q ^ ((row >> 1) & 7)) << 4)) =
          make_float4(a4[0], a4[1], a4[2], a4[3]);
    }
  }
}

__global__ __launch_bounds__(64 * FW) void sa_fused_eval_kernel(FusedArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *w1 = smem, *w2 = w1 + W1B, *w3 = w2 + W2B;
  float *co = reinterpret_cast<float *>(w3 + W3B);          // sc1 sh1 sc2 sh2 sc3 sh3
  float *sc1 = co, *sh1 = sc1 + N1P, *sc2 = sh1 + N1P, *sh2 = sc2 + N2P, *sc3 = sh2 + N2P,
        *sh3 = sc3 + N3P;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  unsigned char *patch = smem + W1B + W2B + W3B + CO_B + (unsigned)wave * PATCH_B;

  stage_planes(w1, p.W[0], p.ldw[0], p.N1, 3 + p.C, N1P, KB1, p.C, tid, 64 * FW);
  stage_planes(w2, p.W[1], p.ldw[1], p.N2, p.N1, N2P, KB2, -1, tid, 64 * FW);
  stage_planes(w3, p.W[2], p.ldw[2], p.N3, p.N2, N3P, KB3, -1, tid, 64 * FW);
  {
    float *scs[3] = {sc1, sc2, sc3}, *shs[3] = {sh1, sh2, sh3};
    const int NPs[3] = {N1P, N2P, N3P}, Ns[3] = {p.N1, p.N2, p.N3};
#pragma unroll
    for (int l = 0; l < 3; ++l)
      for (int c = tid; c < NPs[l]; c += 64 * FW) {
        float s = 0.f, t = 0.f;
        if (c < Ns[l]) {       // bn_eval_coeffs: scale = gamma / sqrt(var + eps), shift = beta - mean scale
          const float invstd = 1.0f / sqrtf(p.var[l][c] + p.eps[l]);
          s = (p.gamma[l] ? p.gamma[l][c] : 1.0f) * invstd;
          t = (p.beta[l] ? p.beta[l][c] : 0.0f) - p.mean[l][c] * s;
        }
        scs[l][c] = s; shs[l][c] = t;
      }
  }
  __syncthreads();

  const int ns = p.ns, C = p.C;
  const long long M = (long long)p.b * p.m * ns;
  const long long tiles = (M + 31) >> 5;
  const long long wid = (long long)blockIdx.x * FW + wave, nw = (long long)gridDim.x * FW;
  const int TG = ns == 64 ? 2 : 1;     // tiles of one 64-row centre stay with one wave
  auto tile_at = [&](long long k) -> long long {
    return TG == 1 ? wid + k * nw : (wid + (k >> 1) * nw) * 2 + (k & 1);
  };

  // gather of one tile into registers: lane (li = row, lk = k-half) holds logical k = 8 lk .. 8 lk + 7
  // of [features (C) | centred xyz (3) | 0].  The neighbour id is loaded one tile EARLIER than
  // the point it addresses (two dependent loads per tile, both in flight under MFMAs).
  auto load_id = [&](long long t) -> int {
    long long row = t * 32 + li;
    if (row >= M) row = M - 1;
    return p.idx[row];
  };
  auto gather = [&](long long t, int pt, float4 &va, float4 &vb) {
    long long row = t * 32 + li;
    if (row >= M) row = M - 1;
    const long long bidx = row / ((long long)p.m * ns);
    const float *f = p.feats + bidx * p.fbs + (long long)pt * p.frs;
    const float *x = p.xyz + (bidx * p.n + pt) * 3;
    const float *c = p.new_xyz + (row / ns) * 3;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 8 * lk + i;
      float val = 0.f;
      if (k < C) {
        val = f[k];
      } else if (k < C + 3) {
        const int a = k - C;
        val = x[a] - c[a];                                   // pointnet2_utils.py:350
        if (p.normalize) val = val / p.radius;               // :352 (a division, as the reference)
      }
      v[i] = val;
    }
    va = make_float4(v[0], v[1], v[2], v[3]);
    vb = make_float4(v[4], v[5], v[6], v[7]);
  };

  float gmax[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) gmax[j][0] = gmax[j][1] = -INFINITY;
  const int swz = (li >> 1) & 7;

  float4 na, nb;                                             // next tile's operand (prefetch)
  int pt_next = 0;                                           // neighbour id of the tile after it
  if (tile_at(0) < tiles) gather(tile_at(0), load_id(tile_at(0)), na, nb);
  if (tile_at(1) < tiles) pt_next = load_id(tile_at(1));
#pragma unroll 1
  for (long long k = 0; tile_at(k) < tiles; ++k) {
    const long long t = tile_at(k);
    const float4 va = na, vb = nb;
    if (tile_at(k + 1) < tiles) gather(tile_at(k + 1), pt_next, na, nb);   // under this tile's MFMAs
    if (tile_at(k + 2) < tiles) pt_next = load_id(tile_at(k + 2));

    // ---- layer 1: one k16 step from registers --------------------------------------------
    f32x16 a1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) a1[j][e] = 0.f;
    mfma_step<2>(a1, va, vb, w1, KB1, N1P, 0, li, lk);
    to_patch<2>(a1, sc1, sh1, patch, lane);

    // ---- layer 2 -------------------------------------------------------------------------
    f32x16 a2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) a2[j][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned char *sl = patch + (ks >> 1) * 4096;
      const int qa = 4 * (ks & 1) + 2 * lk;
      const float4 oa = *reinterpret_cast<const float4 *>(sl + li * 128 + ((qa ^ swz) << 4));
      const float4 ob = *reinterpret_cast<const float4 *>(sl + li * 128 + (((qa + 1) ^ swz) << 4));
      mfma_step<2>(a2, oa, ob, w2, KB2, N2P, ks, li, lk);
    }
    to_patch<2>(a2, sc2, sh2, patch, lane);     // (same wave: LDS operations stay in program order)

    // ---- layer 3 -------------------------------------------------------------------------
    f32x16 a3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) a3[j][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned char *sl = patch + (ks >> 1) * 4096;
      const int qa = 4 * (ks & 1) + 2 * lk;
      const float4 oa = *reinterpret_cast<const float4 *>(sl + li * 128 + ((qa ^ swz) << 4));
      const float4 ob = *reinterpret_cast<const float4 *>(sl + li * 128 + (((qa + 1) ^ swz) << 4));
      mfma_step<4>(a3, oa, ob, w3, KB3, N3P, ks, li, lk);
    }

    // ---- BatchNorm + ReLU + max over the rows of a centre ------------------------------------
    // lane (li, lk) holds rows 4 lk + {0..3, 8..11, 16..19, 24..27} of column li: e < 8 are rows
    // 0..15, e >= 8 rows 16..31 (two centres when ns = 16)
    const long long r0 = t * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s = sc3[32 * j + li], sh = sh3[32 * j + li];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = (e & 3) + 8 * (e >> 2) + 4 * lk;
        float v = fmaxf(a3[j][e] * s + sh, 0.f);
        if (r0 + rl >= M) v = -INFINITY;
        const int h = ns == 16 ? (e >> 3) : 0;
        if (h == 0) gmax[j][0] = fmaxf(gmax[j][0], v); else gmax[j][1] = fmaxf(gmax[j][1], v);
      }
    }
    if (ns != 64 || (k & 1)) {
      const long long centres = M / ns;
      const long long c0 = ns == 64 ? (t >> 1) : (ns == 32 ? t : 2 * t);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = 32 * j + li;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (h == 1 && ns != 16) continue;
          const float mx = fmaxf(gmax[j][h], __shfl_xor(gmax[j][h], 32, 64));
          if (lk == 0 && col < p.N3 && c0 + h < centres) p.out[(c0 + h) * p.ldo + col] = mx;
          gmax[j][h] = -INFINITY;
        }
      }
    }
  }
}

constexpr size_t FUSED_LDS = (size_t)W1B + W2B + W3B + CO_B + (size_t)FW * PATCH_B;

}  // namespace

// 1 when s2c_sa_fused_eval takes a stage of this shape: three layers (3 + C) -> N1 -> N2 -> N3
// with all weight planes resident in LDS, ns rows per centre.
extern "C" int s2c_sa_fused_eval_supported(int ns, int C, int N1, int N2, int N3) {
  return (ns == 16 || ns == 32 || ns == 64) && C >= 0 && C <= 13 && N1 > 0 && N1 <= N1P &&
         N2 > 0 && N2 <= N2P && N3 > 0 && N3 <= N3P && FUSED_LDS <= 160 * 1024;
}

extern "C" int s2c_sa_fused_eval(int b, int n, int m, int ns, int C, long long feat_row_stride,
                                 long long feat_batch_stride, float radius, int normalize,
                                 const float *xyz, const float *new_xyz, const float *feats,
                                 const int *idx, const s2c_eval_layer *layers, float *out, int ldo,
                                 void *stream) {
  if (b <= 0 || n <= 0 || m <= 0 || !xyz || !new_xyz || !idx || !layers || !out ||
      (C > 0 && !feats) || ((long long)m * ns) % 32 != 0)
    return -1;
  if (!s2c_sa_fused_eval_supported(ns, C, layers[0].N, layers[1].N, layers[2].N)) return -2;
  FusedArgs a;
  a.b = b; a.n = n; a.m = m; a.ns = ns; a.C = C;
  a.frs = feat_row_stride; a.fbs = feat_batch_stride;
  a.radius = radius; a.normalize = normalize;
  a.xyz = xyz; a.new_xyz = new_xyz; a.feats = feats; a.idx = idx;
  a.N1 = layers[0].N; a.N2 = layers[1].N; a.N3 = layers[2].N;
  for (int l = 0; l < 3; ++l) {
    if (!layers[l].W || !layers[l].mean || !layers[l].var) return -1;
    a.W[l] = layers[l].W; a.ldw[l] = layers[l].ldw;
    a.gamma[l] = layers[l].gamma; a.beta[l] = layers[l].beta;
    a.mean[l] = layers[l].mean; a.var[l] = layers[l].var; a.eps[l] = layers[l].eps;
  }
  a.out = out; a.ldo = ldo;
  // opt-in dynamic LDS size, per device; a refusal makes the stage "not taken" (-2)
  static int attr_state[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (attr_state[dev] == 0)
    attr_state[dev] = hipFuncSetAttribute((const void *)sa_fused_eval_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)FUSED_LDS) == hipSuccess ? 1 : -1;
  if (attr_state[dev] < 0) { (void)hipGetLastError(); return -2; }
  static int cu_count[64];
  if (cu_count[dev] == 0) {
    int v = 0;
    cu_count[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                     v > 0) ? v : 256;
  }
  const int cus = cu_count[dev];
  const long long tiles = ((long long)b * m * ns + 31) / 32;
  long long blocks = (tiles + FW - 1) / FW;
  if (blocks > cus) blocks = cus;
  hipLaunchKernelGGL(sa_fused_eval_kernel, dim3((unsigned)blocks), dim3(64 * FW), FUSED_LDS,
                     (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_sa_fused_eval launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
