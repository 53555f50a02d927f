// s2c_decoder.hip -- kernels for the teacher-forced top-down caption decoder
// (models/caption_module.py:250-292 `_step`, :428-500 `_forward_sample_batch`).
//
// In training the decoder advances R = batch-size rows (8 at the benchmark
// configuration) through T <= 31 strictly sequential steps.  Each step is ~10
// MFLOP per row: far too little for library GEMMs -- the reference (and a naive
// port) spends it in ~25 forward + ~50 backward micro-kernels per step whose cost
// is launch latency, not work.  Here a step is 7 forward / 6 backward launches of
// purpose-built kernels, the step-invariant pieces are hoisted into a few large
// GEMMs outside the loop (word / target projections, classifier, every weight
// gradient), and the whole sequence replays from a hipGraph.
//
// Lane layout of the small-batch matrix-vector kernels: a block of 256 threads owns a
// few outputs (4 for the plain products, 2 hidden units for the GRU); thread = (row r =
// tid & 7, chunk c = tid >> 3 of 32).  The chunk lanes stride over the float4s of the
// input row, which each lane keeps in registers for all outputs of the block; partial
// dot products are folded with 3 xor-shuffles per wave and one LDS exchange.  R <= 8
// rows per pass (blockIdx.y walks larger batches).  The greedy-decode kernels at the
// bottom (local attention, bf16x3 operand split) serve the evaluation path, where the
// batch is B*K rows and the GEMMs go to the library.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

using namespace s2c;

namespace {

constexpr int RB = 8;  // rows per pass

__device__ __forceinline__ float fold_chunks(float v) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(
      __float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum over aligned groups of G lanes (G = 8..64, power of two); every lane of the
// group receives the sum.  DPP inside a row of 16, bpermute above.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_f32<DPP_QUAD_1032>(v);
  v += dpp_f32<DPP_QUAD_2301>(v);
  v += dpp_f32<DPP_ROW_HALF_MIRROR>(v);
  if (G > 8) v += dpp_f32<DPP_ROW_MIRROR>(v);
  if (G > 16) v += __shfl_xor(v, 16, 64);
  if (G > 32) v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh(x) = 1 - 2 / (e^{2x} + 1) on the hardware exp2 / rcp units (|abs error| < 1e-6;
// saturates correctly for large |x|).  The attention kernels evaluate ~1M tanh per
// launch on a few hundred waves: libm's tanhf made them ALU-bound.
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// partial dot product of one weight row with this lane's input row; the 32 chunk
// lanes of a row (8 per wave x 4 waves of the block) stride over the float4s.
// Two independent accumulators / unroll 4 keep several 16-byte loads in flight
// (these kernels are pure latency: ~1 wave per SIMD, a few KB per wave).
constexpr int NCHUNK = 32;
__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
// The input slice of a lane kept in registers and reused for several weight rows:
// every block re-reads the (R x I) input from L2, so amortising it over OB outputs
// cuts the dominant L2->CU traffic of these kernels by ~OB.
template <int U>
struct XSlice {
  float4 v[U];
  __device__ __forceinline__ void load(const float *__restrict__ x, int n4, int chunk) {
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = chunk + u * NCHUNK;
      v[u] = j < n4 ? x4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ float dot(const float *__restrict__ w, int n4,
                                       int chunk) const {
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    float4 a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = chunk + u * NCHUNK;
      a[u] = j < n4 ? w4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u & 1) acc1 += dot4(a[u], v[u]);
      else acc0 += dot4(a[u], v[u]);
    }
    return acc0 + acc1;
  }
};

// block-wide fold of per-lane partials into s_part[k][wave][row]; after the
// barrier, fold_get(k, row) is the full sum (same association for every caller)
template <int NV>
__device__ __forceinline__ void block_fold(float (&v)[NV], float (*s_part)[4][RB]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    v[k] = fold_chunks(v[k]);
    if (lane < RB) s_part[k][wave][lane] = v[k];
  }
  __syncthreads();
}
__device__ __forceinline__ float fold_get(float (*s_part)[4][RB], int k, int r) {
  return (s_part[k][0][r] + s_part[k][1][r]) + (s_part[k][2][r] + s_part[k][3][r]);
}

// ---------------------------------------------------------------------------
// out[r, o] = epi( sum_i W[o, i] * x[r, i] + bias[o] + add1[r, o] + add2[r, o] )
// epi: 0 none, 1 relu, 2 multiply by (gate[r, o] > 0).
// One launch serves up to two independent problems (blockIdx.x < p1.O -> p1, else
// p2) and can finish problem 1 with the gate part of a GRUCell backward
// (value = dh' of unit o): the BPTT chain is 6 launches per step.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gru_gates_bwd_one(const s2c_gru_bwd_desc &g, int H,
                                                  int row, int u, float dh) {
  const size_t e = (size_t)row * H + u;
  const float r = g.sr[e], z = g.sz[e], n = g.sn[e];
  const float dz = dh * (g.hprev[e] - n);
  const float dn = dh * (1.0f - z);
  const float dpn = dn * (1.0f - n * n);
  const float dr = dpn * g.sghn[e];
  const float dpr = dr * r * (1.0f - r);
  const float dpz = dz * z * (1.0f - z);
  float *gi = g.dgi + (size_t)row * 3 * H;
  float *gh = g.dgh + (size_t)row * 3 * H;
  gi[u] = dpr; gi[H + u] = dpz; gi[2 * H + u] = dpn;
  gh[u] = dpr; gh[H + u] = dpz; gh[2 * H + u] = dpn * r;
  g.dh_direct[e] = dh * z;
}

constexpr int LIN_OB = 4;   // outputs per block

template <int U>
__device__ __forceinline__ void small_linear_body(int R, const s2c_lin_desc &p, int og,
                                                  const s2c_gru_bwd_desc &g1,
                                                  bool gates,
                                                  float (*s_part)[4][RB]) {
  const int r = threadIdx.x & 7, chunk = threadIdx.x >> 3;   // 32 chunks
  const int row = blockIdx.y * RB + r;
  const int rowc = row < R ? row : R - 1;
  const int n4 = p.I >> 2;
  const int o0 = og * LIN_OB;
  // epilogue thread = (row, output of the group); its operands are requested up
  // front so that they travel together with the weight rows (one round trip)
  const int eo = o0 + chunk;
  const bool eok = threadIdx.x < RB * LIN_OB && row < R && eo < p.O;
  float e_bias = 0.f, e_a1 = 0.f, e_a2 = 0.f, e_gate = 1.f;
  float e_r = 0.f, e_z = 0.f, e_n = 0.f, e_ghn = 0.f, e_hp = 0.f;
  if (eok) {
    if (p.bias) e_bias = p.bias[eo];
    if (p.add1) e_a1 = p.add1[(size_t)row * p.ld1 + eo];
    if (p.add2) e_a2 = p.add2[(size_t)row * p.ld2 + eo];
    if (p.epi == 2) e_gate = p.gate[(size_t)row * p.ldg + eo];
    if (gates) {
      const size_t e = (size_t)row * p.O + eo;
      e_r = g1.sr[e]; e_z = g1.sz[e]; e_n = g1.sn[e]; e_ghn = g1.sghn[e];
      e_hp = g1.hprev[e];
    }
  }
  XSlice<U> xs;
  xs.load(p.x + (size_t)rowc * p.ldx, n4, chunk);
  float v[LIN_OB];
#pragma unroll
  for (int k = 0; k < LIN_OB; ++k) {
    const int o = min(o0 + k, p.O - 1);
    v[k] = xs.dot(p.W + (size_t)o * p.ldw, n4, chunk);
    // long rows: keep two weight rows (not four) in flight -> no spills
    if (U > 4 && (k & 1)) __builtin_amdgcn_sched_barrier(0);
  }
  block_fold<LIN_OB>(v, s_part);
  if (eok) {
    float acc = fold_get(s_part, chunk, r);
    if (p.bias) acc += e_bias;
    if (p.add1) acc += e_a1;
    if (p.add2) acc += e_a2;
    if (p.epi == 1) acc = fmaxf(acc, 0.0f);
    else if (p.epi == 2) acc = e_gate > 0.0f ? acc : 0.0f;
    if (p.out) p.out[(size_t)row * p.ldo + eo] = acc;
    if (gates) {
      const int H = p.O;
      const size_t e = (size_t)row * H + eo;
      const float dh = acc;
      const float dz = dh * (e_hp - e_n);
      const float dn = dh * (1.0f - e_z);
      const float dpn = dn * (1.0f - e_n * e_n);
      const float dr = dpn * e_ghn;
      const float dpr = dr * e_r * (1.0f - e_r);
      const float dpz = dz * e_z * (1.0f - e_z);
      float *gi = g1.dgi + (size_t)row * 3 * H;
      float *gh = g1.dgh + (size_t)row * 3 * H;
      gi[eo] = dpr; gi[H + eo] = dpz; gi[2 * H + eo] = dpn;
      gh[eo] = dpr; gh[H + eo] = dpz; gh[2 * H + eo] = dpn * e_r;
      g1.dh_direct[e] = dh * e_z;
    }
  }
}

__global__ __launch_bounds__(256) void small_linear_kernel(
    int R, s2c_lin_desc p1, s2c_lin_desc p2, s2c_gru_bwd_desc g1, int has_g1,
    int groups1) {
  __shared__ float s_part[LIN_OB][4][RB];
  const bool second = (int)blockIdx.x >= groups1;
  const s2c_lin_desc &p = second ? p2 : p1;
  const int og = second ? (int)blockIdx.x - groups1 : (int)blockIdx.x;
  const bool gates = has_g1 && !second;
  const int n4 = p.I >> 2;
  if (n4 <= 4 * NCHUNK) small_linear_body<4>(R, p, og, g1, gates, s_part);
  else small_linear_body<12>(R, p, og, g1, gates, s_part);   // I <= 1536
}

// ---------------------------------------------------------------------------
// GRUCell forward (torch.nn.GRUCell semantics):
//   gi = W_ih x + b_ih ; gh = W_hh h + b_hh          (gate order r, z, n)
//   r = sig(gi_r + gh_r) ; z = sig(gi_z + gh_z) ; n = tanh(gi_n + r * gh_n)
//   h' = (1 - z) * n + z * h
// One wave per hidden unit; saves r, z, n and gh_n (bias included) for BPTT.
// ---------------------------------------------------------------------------
constexpr int GRU_UB = 2;   // hidden units per block

template <int UX, int UH>
__device__ __forceinline__ void gru_fwd_body(
    int R, int H, int I, const float *__restrict__ Wih,
    const float *__restrict__ Whh, const float *__restrict__ bih,
    const float *__restrict__ bhh, const float *__restrict__ x, int ldx,
    const float *__restrict__ h, float *__restrict__ hnew,
    float *__restrict__ sr, float *__restrict__ sz, float *__restrict__ sn,
    float *__restrict__ sghn, float (*s_part)[4][RB]) {
  const int r = threadIdx.x & 7, chunk = threadIdx.x >> 3;
  const int u0 = blockIdx.x * GRU_UB;
  const int row = blockIdx.y * RB + r;
  const int rowc = row < R ? row : R - 1;
  const float *hr = h + (size_t)rowc * H;
  // epilogue thread = (row, unit of the block): operands requested up front
  const int eu = u0 + chunk;
  const bool eok = threadIdx.x < RB * GRU_UB && row < R && eu < H;
  float b_i[3] = {0.f, 0.f, 0.f}, b_h[3] = {0.f, 0.f, 0.f}, hp = 0.f;
  if (eok) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      b_i[k] = bih[k * H + eu];
      b_h[k] = bhh[k * H + eu];
    }
    hp = hr[eu];
  }
  XSlice<UX> xs;
  XSlice<UH> hs;
  xs.load(x + (size_t)rowc * ldx, I >> 2, chunk);
  hs.load(hr, H >> 2, chunk);
  float g[6 * GRU_UB];
#pragma unroll
  for (int j = 0; j < GRU_UB; ++j) {
    const int u = min(u0 + j, H - 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      g[6 * j + k] = xs.dot(Wih + (size_t)(k * H + u) * I, I >> 2, chunk);
      g[6 * j + 3 + k] = hs.dot(Whh + (size_t)(k * H + u) * H, H >> 2, chunk);
    }
  }
  block_fold<6 * GRU_UB>(g, s_part);
  if (eok) {
    const int j = chunk;
    const float gir = fold_get(s_part, 6 * j, r) + b_i[0],
                giz = fold_get(s_part, 6 * j + 1, r) + b_i[1],
                gin = fold_get(s_part, 6 * j + 2, r) + b_i[2];
    const float ghr = fold_get(s_part, 6 * j + 3, r) + b_h[0],
                ghz = fold_get(s_part, 6 * j + 4, r) + b_h[1],
                ghn = fold_get(s_part, 6 * j + 5, r) + b_h[2];
    const float rr = sigmoidf_(gir + ghr);
    const float zz = sigmoidf_(giz + ghz);
    const float nn = tanhf(gin + rr * ghn);
    const size_t e = (size_t)row * H + eu;
    hnew[e] = (1.0f - zz) * nn + zz * hp;
    sr[e] = rr; sz[e] = zz; sn[e] = nn; sghn[e] = ghn;
  }
}

__global__ __launch_bounds__(256) void gru_fwd_kernel(
    int R, int H, int I, const float *__restrict__ Wih,
    const float *__restrict__ Whh, const float *__restrict__ bih,
    const float *__restrict__ bhh, const float *__restrict__ x, int ldx,
    const float *__restrict__ h, float *__restrict__ hnew,
    float *__restrict__ sr, float *__restrict__ sz, float *__restrict__ sn,
    float *__restrict__ sghn) {
  __shared__ float s_part[6 * GRU_UB][4][RB];
  gru_fwd_body<4, 4>(R, H, I, Wih, Whh, bih, bhh, x, ldx, h, hnew, sr, sz, sn, sghn, s_part);
  // (host side rejects I, H > 512)
}

// GRUCell backward, gate part (elementwise over R x H):
//   dgi = [dpre_r, dpre_z, dpre_n] ; dgh = [dpre_r, dpre_z, dpre_n * r]
//   dh_direct = dh' * z
__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(
    int R, int H, const float *__restrict__ dh1, const float *__restrict__ dh2,
    s2c_gru_bwd_desc g) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= R * H) return;
  const int row = e / H, u = e - row * H;
  float dh = dh1[e];
  if (dh2) dh += dh2[e];
  gru_gates_bwd_one(g, H, row, u, dh);
}

// ---------------------------------------------------------------------------
// Additive attention (caption_module.py:274-283).
//   s[r,k] = sum_h wa[h] * tanh(M[r,k,h] + q[r,h]) ; masked -> -1e30
// One wave per (row, k): 64 lanes over H.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_scores_kernel(
    int R, int K, int H, const float *__restrict__ M, const float *__restrict__ q,
    int ldq, const float *__restrict__ wa, const float *__restrict__ mask,
    float *__restrict__ scores) {
  const int lane = threadIdx.x & 63;
  const long long rk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rk >= (long long)R * K) return;
  const int row = (int)(rk / K);
  const float *m = M + rk * H;
  const float *qq = q + (size_t)row * ldq;
  float acc = 0.0f;
  for (int h4 = lane; h4 < (H >> 2); h4 += 64) {
    const float4 a = reinterpret_cast<const float4 *>(m)[h4];
    const float4 b = reinterpret_cast<const float4 *>(qq)[h4];
    const float4 w = reinterpret_cast<const float4 *>(wa)[h4];
    acc += w.x * fast_tanh(a.x + b.x) + w.y * fast_tanh(a.y + b.y) +
           w.z * fast_tanh(a.z + b.z) + w.w * fast_tanh(a.w + b.w);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) scores[rk] = mask[rk] == 0.0f ? -1e30f : acc;
}

// softmax over K and att[r,:] = sum_k alpha[r,k] * O[r,k,:].
// Block = (row, chunk of ATT_FC output features); every block of a row redoes the
// (tiny) softmax, block 0 of the row stores alpha.  Thread = (feature, k group).
constexpr int ATT_FC = 32;
__global__ __launch_bounds__(256) void attn_softmax_kernel(
    int K, int F, const float *__restrict__ scores, const float *__restrict__ O,
    float *__restrict__ alpha, float *__restrict__ att, int lda) {
  __shared__ float s_red[4];
  __shared__ float s_alpha[1024];
  __shared__ float s_acc[256];
  const int row = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *s = scores + (size_t)row * K;
  float mx = -INFINITY;
  for (int k = tid; k < K; k += 256) mx = fmaxf(mx, s[k]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) s_red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float sum = 0.0f;
  for (int k = tid; k < K; k += 256) {
    const float e = expf(s[k] - mx);
    s_alpha[k] = e;
    sum += e;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) s_red[wave] = sum;
  __syncthreads();
  sum = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  const float inv = 1.0f / sum;
  for (int k = tid; k < K; k += 256) {
    const float a = s_alpha[k] * inv;
    s_alpha[k] = a;
    if (blockIdx.x == 0) alpha[(size_t)row * K + k] = a;
  }
  __syncthreads();
  const int fl = tid & (ATT_FC - 1), kg = tid / ATT_FC;      // 8 k groups
  const int f = blockIdx.x * ATT_FC + fl;
  const float *o = O + (size_t)row * K * F;
  float acc = 0.0f;
  if (f < F) {
    float a0 = 0.0f, a1 = 0.0f;
    int k = kg;
    for (; k + 8 < K; k += 16) {
      a0 += s_alpha[k] * o[(size_t)k * F + f];
      a1 += s_alpha[k + 8] * o[(size_t)(k + 8) * F + f];
    }
    if (k < K) a0 += s_alpha[k] * o[(size_t)k * F + f];
    acc = a0 + a1;
  }
  s_acc[tid] = acc;
  __syncthreads();
  if (tid < ATT_FC && f < F) {
    float t = 0.0f;
#pragma unroll
    for (int g = 0; g < 256 / ATT_FC; ++g) t += s_acc[g * ATT_FC + tid];
    att[(size_t)row * lda + f] = t;
  }
}

// ---------------------------------------------------------------------------
// Few keys (the num_locals gather, K <= AX_MAXK): the whole attention AND the layer that
// consumes it in one launch -- scores, mask, softmax, weighted sum, then
//   x2[r, o] = relu( sum_f Wl[o, f] att[r, f] + bias[o] + add[r, o] )     (caption_module.py:284-287:
//   map_lang on [att | h1]; the h1 block arrives in `add`)
// Block = (row, 32 outputs): every block of a row redoes the row's scores (K x H tanh: 20 per
// thread at K = 10, H = 512) and its softmax / weighted sum (K x F multiply-adds) -- cheaper
// than the two launch boundaries it replaces (7 -> 5 dependent launches per decoder step).
// Blocks of output chunk 0 store alpha and att for the backward pass.
// ---------------------------------------------------------------------------
constexpr int AX_MAXK = 32, AX_OC = 32, AX_MAXF = 256;
__global__ __launch_bounds__(256) void attn_x2_kernel(
    int K, int H, int F, int E, const float *__restrict__ M, const float *__restrict__ q, int ldq,
    const float *__restrict__ wa, const float *__restrict__ mask, const float *__restrict__ O,
    const float *__restrict__ Wl, int ldw, const float *__restrict__ bias,
    const float *__restrict__ add, int ld_add, float *__restrict__ alpha, float *__restrict__ att,
    int lda, float *__restrict__ x2, int ldx2) {
  __shared__ float s_sc[AX_MAXK], s_e[AX_MAXK];
  __shared__ __attribute__((aligned(16))) float s_att[AX_MAXF];
  const int row = blockIdx.y, oc = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (1) scores: wave w takes keys w, w + 4, ...; 64 lanes over the H/4 float4s.  Every load of
  // a pass is issued before the first is used (these launches are pure latency: one L2 round
  // trip per pass instead of one per key)
  const float *qq = q + (size_t)row * ldq;
  constexpr int KW = AX_MAXK / 4;
  float sc[KW];
#pragma unroll
  for (int kk = 0; kk < KW; ++kk) sc[kk] = 0.0f;
  for (int h4 = lane; h4 < (H >> 2); h4 += 64) {
    const float4 b = reinterpret_cast<const float4 *>(qq)[h4];
    const float4 w = reinterpret_cast<const float4 *>(wa)[h4];
    float4 a[KW];
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      const int k = wave + 4 * kk;
      a[kk] = k < K ? reinterpret_cast<const float4 *>(M + ((size_t)row * K + k) * H)[h4]
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int kk = 0; kk < KW; ++kk)
      if (wave + 4 * kk < K)
        sc[kk] += w.x * fast_tanh(a[kk].x + b.x) + w.y * fast_tanh(a[kk].y + b.y) +
                  w.z * fast_tanh(a[kk].z + b.z) + w.w * fast_tanh(a[kk].w + b.w);
  }
#pragma unroll
  for (int kk = 0; kk < KW; ++kk) {
    const int k = wave + 4 * kk;
    if (k < K) {                                     // wave-uniform
      float acc = sc[kk];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (lane == 0) s_sc[k] = mask[(size_t)row * K + k] == 0.0f ? -1e30f : acc;
    }
  }
  __syncthreads();
  // (2) softmax over the K <= 32 scores: one exponential per key, shared through LDS
  if (tid < K) {
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, s_sc[k]);
    s_e[tid] = expf(s_sc[tid] - mx);
  }
  __syncthreads();
  float sum = 0.0f;
  for (int k = 0; k < K; ++k) sum += s_e[k];
  const float inv = 1.0f / sum;
  if (oc == 0 && tid < K) alpha[(size_t)row * K + tid] = s_e[tid] * inv;
  // (3) att[f] = sum_k alpha_k O[row, k, f]
  for (int f = tid; f < F; f += 256) {
    const float *o = O + (size_t)row * K * F + f;
    float a = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 8) {              // 8 loads in flight, summed in key order
      float ov[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) ov[u] = k0 + u < K ? o[(size_t)(k0 + u) * F] : 0.0f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u < K) a += (s_e[k0 + u] * inv) * ov[u];
    }
    s_att[f] = a;
    if (oc == 0) att[(size_t)row * lda + f] = a;
  }
  __syncthreads();
  // (4) 32 outputs of the consuming layer: thread = (output, 8 column slices)
  const int ol = tid >> 3, part = tid & 7;
  const int o = oc * AX_OC + ol;
  float acc = 0.0f;
  if (o < E) {
    const float4 *w4 = reinterpret_cast<const float4 *>(Wl + (size_t)o * ldw);
    const float4 *a4 = reinterpret_cast<const float4 *>(s_att);
    for (int j0 = part; j0 < (F >> 2); j0 += 32) {   // 4 weight loads in flight
      float4 wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        wv[u] = j0 + 8 * u < (F >> 2) ? w4[j0 + 8 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + 8 * u < (F >> 2)) acc += dot4(wv[u], a4[j0 + 8 * u]);
    }
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 4, 64);
  if (part == 0 && o < E) {
    float v = acc + (bias ? bias[o] : 0.0f) + (add ? add[(size_t)row * ld_add + o] : 0.0f);
    x2[(size_t)row * ldx2 + o] = fmaxf(v, 0.0f);
  }
}

// Attention backward for one step.  Block = (chunk of ATT_HC hidden units, row);
// it owns dq[row, chunk] and dwa_rows[row, chunk] outright: no atomics (device-scope
// float atomics on a few hundred shared addresses serialise at the memory side and
// made the first version of this kernel 5x slower), deterministic.
//  (a) softmax backward without a row-wide pass:  sum_k alpha_k dalpha_k
//      = <datt, sum_k alpha_k O_k> = <datt, att>  (att is the saved forward output);
//      dalpha_k = <datt, O[r,k,:]> ; ds_k = alpha_k (dalpha_k - <datt, att>)
//      (recomputed by every chunk block of the row: K*F MACs, L2-resident)
//  (b) dpre = ds * wa * (1 - c^2), c = tanh(M + q);
//      dM[r,k,:] += dpre ; dq[r,:] = sum_k dpre ; dwa_rows[r,:] += sum_k ds * c
// dO = sum_t alpha_t (outer) datt_t has no recurrence and is one batched GEMM after
// the time loop (decoder_fused.py); dwa = sum_r dwa_rows[r] likewise.
constexpr int ATT_HC = 32;                 // hidden units per block (8 float4)
constexpr int ATT_KG = 256 / (ATT_HC / 4); // key groups per block (32)
constexpr int ATT_KB = 8;                  // keys in flight per thread
constexpr int ATT_PB = 16;                 // stage (a) passes in flight
template <int F4>
__device__ __forceinline__ void attn_bwd_ds_stage(int K, int row, float dot,
                                                  const float *s_datt,
                                                  const float *__restrict__ O,
                                                  const float *__restrict__ alpha,
                                                  float *s_ds) {
  const int tid = threadIdx.x;
  const int sub = tid & (F4 - 1), kl = tid / F4;
  constexpr int kpp = 256 / F4;
  const float4 d4 = reinterpret_cast<const float4 *>(s_datt)[sub];
  const float4 *O4 = reinterpret_cast<const float4 *>(O) + (size_t)row * K * F4;
  for (int base = 0; base < K; base += kpp * ATT_PB) {
    float4 o[ATT_PB];
    float al[ATT_PB];
#pragma unroll
    for (int u = 0; u < ATT_PB; ++u) {
      const int k = base + u * kpp + kl;
      const bool in = k < K;
      o[u] = in ? O4[(size_t)k * F4 + sub] : make_float4(0.f, 0.f, 0.f, 0.f);
      al[u] = in ? alpha[(size_t)row * K + k] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < ATT_PB; ++u) {
      const int k = base + u * kpp + kl;
      const float da = group_sum<F4>(dot4(o[u], d4));
      if (sub == 0 && k < K) s_ds[k] = al[u] * (da - dot);
    }
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(
    int K, int H, int F, const float *__restrict__ datt, int ldd,
    const float *__restrict__ att, int lda, const float *__restrict__ alpha,
    const float *__restrict__ O, const float *__restrict__ M,
    const float *__restrict__ q, int ldq, const float *__restrict__ wa,
    float *__restrict__ dM, float *__restrict__ dq, float *__restrict__ dwa_rows) {
  __shared__ float s_datt[512];
  __shared__ float s_red[4];
  __shared__ float s_ds[1024];
  __shared__ float4 s_sq[ATT_KG][ATT_HC / 4], s_sw[ATT_KG][ATT_HC / 4];
  const int row = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float part = 0.0f;
  for (int f = tid; f < F; f += 256) {
    const float d = datt[(size_t)row * ldd + f];
    s_datt[f] = d;
    part += d * att[(size_t)row * lda + f];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
  if (lane == 0) s_red[wave] = part;
  __syncthreads();
  const float dot = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  // (a) O[row] is one contiguous (K x F) array: linear float4 streaming, F/4 lanes
  // per key, ATT_PB passes in flight, DPP reduction inside the key's lane group
  if (F == 128) attn_bwd_ds_stage<32>(K, row, dot, s_datt, O, alpha, s_ds);
  else if (F == 64) attn_bwd_ds_stage<16>(K, row, dot, s_datt, O, alpha, s_ds);
  else if (F == 32) attn_bwd_ds_stage<8>(K, row, dot, s_datt, O, alpha, s_ds);
  else attn_bwd_ds_stage<64>(K, row, dot, s_datt, O, alpha, s_ds);
  __syncthreads();
  // (b) thread = (float4 of hidden units, key group); ATT_KB keys in flight
  const int H4 = H >> 2;
  const int hl = tid & (ATT_HC / 4 - 1), kg = tid / (ATT_HC / 4);
  const int h4 = blockIdx.x * (ATT_HC / 4) + hl;
  const bool ok = h4 < H4;
  float4 sq = make_float4(0.f, 0.f, 0.f, 0.f), sw = sq;
  if (ok) {
    const float4 qh = reinterpret_cast<const float4 *>(q + (size_t)row * ldq)[h4];
    const float4 wh = reinterpret_cast<const float4 *>(wa)[h4];
    for (int kbase = kg; kbase < K; kbase += ATT_KG * ATT_KB) {
      float4 m[ATT_KB], dm[ATT_KB];
      float d[ATT_KB];
#pragma unroll
      for (int j = 0; j < ATT_KB; ++j) {
        const int k = kbase + j * ATT_KG;
        d[j] = k < K ? s_ds[k] : 0.0f;
        if (d[j] != 0.0f) {
          const size_t e4 = ((size_t)row * K + k) * H4 + h4;
          m[j] = reinterpret_cast<const float4 *>(M)[e4];
          dm[j] = reinterpret_cast<const float4 *>(dM)[e4];
        }
      }
#pragma unroll
      for (int j = 0; j < ATT_KB; ++j) {
        if (d[j] == 0.0f) continue;
        const int k = kbase + j * ATT_KG;
        const size_t e4 = ((size_t)row * K + k) * H4 + h4;
        const float cx = fast_tanh(m[j].x + qh.x), cy = fast_tanh(m[j].y + qh.y),
                    cz = fast_tanh(m[j].z + qh.z), cw = fast_tanh(m[j].w + qh.w);
        float4 dp;
        dp.x = d[j] * wh.x * (1.0f - cx * cx);
        dp.y = d[j] * wh.y * (1.0f - cy * cy);
        dp.z = d[j] * wh.z * (1.0f - cz * cz);
        dp.w = d[j] * wh.w * (1.0f - cw * cw);
        dm[j].x += dp.x; dm[j].y += dp.y; dm[j].z += dp.z; dm[j].w += dp.w;
        reinterpret_cast<float4 *>(dM)[e4] = dm[j];
        sq.x += dp.x; sq.y += dp.y; sq.z += dp.z; sq.w += dp.w;
        sw.x += d[j] * cx; sw.y += d[j] * cy; sw.z += d[j] * cz; sw.w += d[j] * cw;
      }
    }
  }
  s_sq[kg][hl] = sq;
  s_sw[kg][hl] = sw;
  __syncthreads();
  if (tid < ATT_HC / 4 && ok) {
    float4 a = s_sq[0][tid], b = s_sw[0][tid];
    for (int g = 1; g < ATT_KG; ++g) {             // fixed order: deterministic
      const float4 u = s_sq[g][tid], v = s_sw[g][tid];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
    reinterpret_cast<float4 *>(dq + (size_t)row * H)[h4] = a;
    float4 *dw = reinterpret_cast<float4 *>(dwa_rows + (size_t)row * H) + h4;
    float4 w0 = *dw;
    w0.x += b.x; w0.y += b.y; w0.z += b.z; w0.w += b.w;
    *dw = w0;
  }
}

// Few keys (K <= AX_MAXK), backward: the transposed map_lang product that yields datt and the
// whole attention backward in ONE launch (the mirror of attn_x2_kernel: both sides of that
// boundary only mix data of one batch row):
//   datt[f] = <WlT[f, :], da2[row, :]>      (WlT = W_lang^T, rows f < F: the att columns)
//   ds[k]   = alpha[k] * (<O[k, :], datt> - <datt, att>)
//   dM[k, h] += ds[k] wa[h] (1 - tanh^2(M[k, h] + q[h])),  dq[h] = sum_k (same),  dwa[h] += sum_k ds[k] tanh(.)
// ONE workgroup of 1024 threads per row: the F x E weights (150 KB) are read once per row, every
// load of a phase is in flight before the first use (a first version with 4 x 256-thread
// workgroups per row re-read the weights four times in two dependent batches: 8.6 us, no gain
// over the two launches it replaced).
constexpr int ABX_T = 1024;
__global__ __launch_bounds__(ABX_T) void attn_bwd_x2_kernel(
    int K, int H, int F, int E, const float *__restrict__ da2, int ldda,
    const float *__restrict__ WlT, int ldw, const float *__restrict__ att, int lda,
    const float *__restrict__ alpha, const float *__restrict__ O, const float *__restrict__ M,
    const float *__restrict__ q, int ldq, const float *__restrict__ wa, float *__restrict__ dM,
    float *__restrict__ dq, int lddq, float *__restrict__ dwa_rows) {
  __shared__ __attribute__((aligned(16))) float s_x[512];
  __shared__ __attribute__((aligned(16))) float s_datt[AX_MAXF];
  __shared__ float s_red[ABX_T / 64];
  __shared__ float s_ds[AX_MAXK];
  __shared__ float4 s_sq[8][128], s_sw[8][128];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < E; e += ABX_T) s_x[e] = da2[(size_t)row * ldda + e];
  __syncthreads();
  // (1) datt: 1024 / F threads per output (8 at F = 128: <= 10 weight loads each, all in flight)
  {
    const int tpf = ABX_T / F;                   // F a power of two in 32 .. 256
    const int f = tid / tpf, part = tid - f * tpf;
    const float4 *w4 = reinterpret_cast<const float4 *>(WlT + (size_t)f * ldw);
    const float4 *x4 = reinterpret_cast<const float4 *>(s_x);
    const int E4 = E >> 2;
    float acc = 0.0f;
    for (int j0 = part; j0 < E4; j0 += 12 * tpf) {
      float4 wv[12];
#pragma unroll
      for (int u = 0; u < 12; ++u)
        wv[u] = j0 + u * tpf < E4 ? w4[j0 + u * tpf] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 12; ++u)
        if (j0 + u * tpf < E4) acc += dot4(wv[u], x4[j0 + u * tpf]);
    }
    for (int off = 1; off < tpf; off <<= 1) acc += __shfl_xor(acc, off, 64);
    if (part == 0) s_datt[f] = acc;
  }
  __syncthreads();
  float part = 0.0f;
  for (int f = tid; f < F; f += ABX_T) part += s_datt[f] * att[(size_t)row * lda + f];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
  if (lane == 0) s_red[wave] = part;
  __syncthreads();
  float dot = 0.0f;
#pragma unroll
  for (int w = 0; w < ABX_T / 64; ++w) dot += s_red[w];
  // (2) ds: wave w takes keys w and w + 16; lanes over the F / 4 float4s of a key's features
  {
    const int F4 = F >> 2;
    const float4 d4 = lane < F4 ? reinterpret_cast<const float4 *>(s_datt)[lane]
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int k = wave + 16 * kk;
      o[kk] = (k < K && lane < F4)
                  ? reinterpret_cast<const float4 *>(O)[((size_t)row * K + k) * F4 + lane]
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int k = wave + 16 * kk;
      if (k < K) {                               // wave-uniform
        float da = dot4(o[kk], d4);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) da += __shfl_xor(da, off, 64);
        if (lane == 0) s_ds[k] = alpha[(size_t)row * K + k] * (da - dot);
      }
    }
  }
  __syncthreads();
  // (3) thread = (float4 of hidden units, key group of 8): keys kg, kg + 8, ...
  const int H4 = H >> 2;
  const int hl = tid & 127, kg = tid >> 7;
  for (int h0 = 0; h0 < H4; h0 += 128) {
    const int h4 = h0 + hl;
    const bool ok = h4 < H4;
    float4 sq = make_float4(0.f, 0.f, 0.f, 0.f), sw = sq;
    if (ok) {
      const float4 qh = reinterpret_cast<const float4 *>(q + (size_t)row * ldq)[h4];
      const float4 wh = reinterpret_cast<const float4 *>(wa)[h4];
      constexpr int KT = AX_MAXK / 8;
      float4 m[KT], dm[KT];
      float d[KT];
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        const int k = kg + 8 * j;
        d[j] = k < K ? s_ds[k] : 0.0f;
        if (d[j] != 0.0f) {
          const size_t e4 = ((size_t)row * K + k) * H4 + h4;
          m[j] = reinterpret_cast<const float4 *>(M)[e4];
          dm[j] = reinterpret_cast<const float4 *>(dM)[e4];
        }
      }
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        if (d[j] == 0.0f) continue;
        const int k = kg + 8 * j;
        const size_t e4 = ((size_t)row * K + k) * H4 + h4;
        const float cx = fast_tanh(m[j].x + qh.x), cy = fast_tanh(m[j].y + qh.y),
                    cz = fast_tanh(m[j].z + qh.z), cw = fast_tanh(m[j].w + qh.w);
        float4 dp;
        dp.x = d[j] * wh.x * (1.0f - cx * cx);
        dp.y = d[j] * wh.y * (1.0f - cy * cy);
        dp.z = d[j] * wh.z * (1.0f - cz * cz);
        dp.w = d[j] * wh.w * (1.0f - cw * cw);
        dm[j].x += dp.x; dm[j].y += dp.y; dm[j].z += dp.z; dm[j].w += dp.w;
        reinterpret_cast<float4 *>(dM)[e4] = dm[j];
        sq.x += dp.x; sq.y += dp.y; sq.z += dp.z; sq.w += dp.w;
        sw.x += d[j] * cx; sw.y += d[j] * cy; sw.z += d[j] * cz; sw.w += d[j] * cw;
      }
    }
    if (h0 > 0) __syncthreads();                 // the previous chunk's sums have been read
    s_sq[kg][hl] = sq;
    s_sw[kg][hl] = sw;
    __syncthreads();
    if (tid < 128 && ok) {
      float4 a = s_sq[0][tid], b = s_sw[0][tid];
      for (int g = 1; g < 8; ++g) {                // fixed order: deterministic
        const float4 u = s_sq[g][tid], v = s_sw[g][tid];
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
      }
      reinterpret_cast<float4 *>(dq + (size_t)row * lddq)[h4] = a;
      float4 *dw = reinterpret_cast<float4 *>(dwa_rows + (size_t)row * H) + h4;
      float4 w0 = *dw;
      w0.x += b.x; w0.y += b.y; w0.z += b.z; w0.w += b.w;
      *dw = w0;
    }
  }
}

// ---------------------------------------------------------------------------
// Greedy decoding of every proposal (caption_module.py:502-592) attends over the
// L = num_locals gathered objects of each row: R = B*K rows (2048..8192) per step.
// One wave per row, ONE pass over mapped (R,L,H) and feats (R,L,F) instead of the
// add / tanh / attend-GEMV / masked_fill / softmax / mul / sum chain (~8 passes over
// the 168 MB `mapped` tensor at cfg5):
//   s[l] = sum_h wa[h] * tanh(mapped[r,l,h] + q[r,h]) + ba ; invalid -> -1e30
//   alpha = softmax_l(s) ; att[r,:] = sum_l alpha[l] * feats[r,l,:]
// ---------------------------------------------------------------------------
constexpr int AL_MAXL = 32;
// LT > 0: L == LT known at compile time (straight-line code, every load of a row in
// flight at once); LT == 0: runtime L <= AL_MAXL.
template <int LT>
__global__ __launch_bounds__(256) void attn_local_kernel(
    int R, int Lrt, int H, int F, const float *__restrict__ mapped,
    const float *__restrict__ q, int ldq, const float *__restrict__ wa, float ba,
    const float *__restrict__ valid, const float *__restrict__ feats,
    float *__restrict__ alpha, float *__restrict__ att, int lda,
    unsigned short *__restrict__ planes, long long pstride, int ldp, int tiled) {
  constexpr int LM = LT > 0 ? LT : AL_MAXL;
  const int L = LT > 0 ? LT : Lrt;
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int H4 = H >> 2;
  float sc[LM];
#pragma unroll
  for (int l = 0; l < LM; ++l) sc[l] = 0.0f;
  const float4 *m4 = reinterpret_cast<const float4 *>(mapped + (size_t)r * L * H);
  for (int h4 = lane; h4 < H4; h4 += 64) {
    const float4 qq = reinterpret_cast<const float4 *>(q + (size_t)r * ldq)[h4];
    const float4 w = reinterpret_cast<const float4 *>(wa)[h4];
    float4 m[LM];
#pragma unroll
    for (int l = 0; l < LM; ++l)
      m[l] = (LT > 0 || l < L) ? m4[(size_t)l * H4 + h4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int l = 0; l < LM; ++l)
      sc[l] += w.x * fast_tanh(m[l].x + qq.x) + w.y * fast_tanh(m[l].y + qq.y) +
               w.z * fast_tanh(m[l].z + qq.z) + w.w * fast_tanh(m[l].w + qq.w);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int l = 0; l < LM; ++l) {
    if (LT > 0 || l < L) {
      float v = sc[l];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      v += ba;
      if (valid != nullptr && valid[(size_t)r * L + l] == 0.0f) v = -1e30f;
      sc[l] = v;
      mx = fmaxf(mx, v);
    }
  }
  float sum = 0.0f;
#pragma unroll
  for (int l = 0; l < LM; ++l)
    if (LT > 0 || l < L) { sc[l] = expf(sc[l] - mx); sum += sc[l]; }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int l = 0; l < LM; ++l)
    if (LT > 0 || l < L) {
      sc[l] *= inv;
      if (lane == 0) alpha[(size_t)r * L + l] = sc[l];
    }
  for (int f = lane; f < F; f += 64) {
    float v[LM];
#pragma unroll
    for (int l = 0; l < LM; ++l)
      v[l] = (LT > 0 || l < L) ? feats[((size_t)r * L + l) * F + f] : 0.0f;
    float a = 0.0f;
#pragma unroll
    for (int l = 0; l < LM; ++l)
      if (LT > 0 || l < L) a += sc[l] * v[l];
    if (att != nullptr) att[(size_t)r * lda + f] = a;
    if (planes != nullptr) {      // the bf16x3 planes of the attended vector (s2c_planes.hip's operand)
      const __bf16 h = (__bf16)a;
      const float r1 = a - (float)h;
      const __bf16 m = (__bf16)r1;
      const __bf16 lo = (__bf16)(r1 - (float)m);
      // (tiled: the 32-row x 16-column block layout of include/s2c_fused.h)
      const int r32 = r & 31;
      unsigned short *pp = planes + (tiled
          ? ((size_t)(r >> 5) * (ldp >> 4) + (f >> 4)) * 512 +
                ((r32 * 2 + (((f >> 3) & 1) ^ ((r32 >> 3) & 1))) << 3) + (f & 7)
          : (size_t)r * ldp + f);
      pp[0] = __builtin_bit_cast(unsigned short, h);
      pp[pstride] = __builtin_bit_cast(unsigned short, m);
      pp[2 * pstride] = __builtin_bit_cast(unsigned short, lo);
    }
  }
}


}  // namespace

static int chk(const char *k) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c: %s launch failed: %s\n", k, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

static s2c_lin_desc one_desc(int O, int I, const float *W, int ldw, const float *x,
                             int ldx, const float *bias, const float *add1, int ld1,
                             const float *add2, int ld2, const float *gate, int ldg,
                             int epi, float *out, int ldo) {
  s2c_lin_desc d;
  d.W = W; d.x = x; d.bias = bias; d.add1 = add1; d.add2 = add2; d.gate = gate;
  d.out = out; d.O = O; d.I = I; d.ldw = ldw; d.ldx = ldx; d.ld1 = ld1; d.ld2 = ld2;
  d.ldg = ldg; d.ldo = ldo; d.epi = epi;
  return d;
}
static bool bad_desc(const s2c_lin_desc *p) {
  return p->O <= 0 || p->I <= 0 || p->I > 1536 || (p->I & 3) || (p->ldw & 3) || (p->ldx & 3) ||
         !p->W || !p->x || (p->epi == 2 && !p->gate);
}

extern "C" int s2c_small_linear_pair(int R, const s2c_lin_desc *p1,
                                     const s2c_lin_desc *p2,
                                     const s2c_gru_bwd_desc *g1, void *stream) {
  if (R <= 0 || !p1 || bad_desc(p1) || (p2 && bad_desc(p2))) return -1;
  if (!p1->out && !g1) return -1;
  if (p2 && !p2->out) return -1;
  s2c_lin_desc q2 = p2 ? *p2 : *p1;
  s2c_gru_bwd_desc g;
  if (g1) g = *g1;
  else g.sr = g.sz = g.sn = g.sghn = g.hprev = nullptr, g.dgi = g.dgh = g.dh_direct = nullptr;
  const int groups1 = (p1->O + LIN_OB - 1) / LIN_OB;
  const int groups = groups1 + (p2 ? (p2->O + LIN_OB - 1) / LIN_OB : 0);
  hipLaunchKernelGGL(small_linear_kernel, dim3(groups, (R + RB - 1) / RB), dim3(256), 0,
                     (hipStream_t)stream, R, *p1, q2, g, g1 ? 1 : 0, groups1);
  return chk("small_linear");
}

extern "C" int s2c_small_linear(int R, int O, int I, const float *W, int ldw,
                                const float *x, int ldx, const float *bias,
                                const float *add1, int ld1, const float *add2,
                                int ld2, const float *gate, int ldg, int epi,
                                float *out, int ldo, void *stream) {
  const s2c_lin_desc d = one_desc(O, I, W, ldw, x, ldx, bias, add1, ld1, add2, ld2, gate,
                                  ldg, epi, out, ldo);
  return s2c_small_linear_pair(R, &d, nullptr, nullptr, stream);
}

extern "C" int s2c_gru_fwd(int R, int H, int I, const float *Wih, const float *Whh,
                           const float *bih, const float *bhh, const float *x,
                           int ldx, const float *h, float *hnew, float *sr,
                           float *sz, float *sn, float *sghn, void *stream) {
  if (R <= 0 || (H & 3) || (I & 3) || (ldx & 3) || H > 512 || I > 512) return -1;
  hipLaunchKernelGGL(gru_fwd_kernel, dim3((H + GRU_UB - 1) / GRU_UB, (R + RB - 1) / RB), dim3(256),
                     0, (hipStream_t)stream, R, H, I, Wih, Whh, bih, bhh, x, ldx, h,
                     hnew, sr, sz, sn, sghn);
  return chk("gru_fwd");
}

extern "C" int s2c_gru_gates_bwd(int R, int H, const float *dh1, const float *dh2,
                                 const float *sr, const float *sz, const float *sn,
                                 const float *sghn, const float *hprev, float *dgi,
                                 float *dgh, float *dh_direct, void *stream) {
  if (R <= 0 || H <= 0) return -1;
  s2c_gru_bwd_desc g;
  g.sr = sr; g.sz = sz; g.sn = sn; g.sghn = sghn; g.hprev = hprev;
  g.dgi = dgi; g.dgh = dgh; g.dh_direct = dh_direct;
  hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3((R * H + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, R, H, dh1, dh2, g);
  return chk("gru_gates_bwd");
}

extern "C" int s2c_attn_fwd(int R, int K, int H, int F, const float *M,
                            const float *q, int ldq, const float *wa,
                            const float *mask, const float *O, float *scores,
                            float *alpha, float *att, int lda, void *stream) {
  if (R <= 0 || K <= 0 || K > 1024 || (H & 3) || (ldq & 3)) return -1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_scores_kernel, dim3((unsigned)(((long long)R * K + 3) / 4)),
                     dim3(256), 0, st, R, K, H, M, q, ldq, wa, mask, scores);
  hipLaunchKernelGGL(attn_softmax_kernel, dim3((F + ATT_FC - 1) / ATT_FC, R), dim3(256),
                     0, st, K, F, scores, O, alpha, att, lda);
  return chk("attn_fwd");
}

extern "C" int s2c_attn_x2_fwd(int R, int K, int H, int F, int E, const float *M, const float *q,
                               int ldq, const float *wa, const float *mask, const float *O,
                               const float *Wl, int ldw, const float *bias, const float *add,
                               int ld_add, float *alpha, float *att, int lda, float *x2, int ldx2,
                               void *stream) {
  if (R <= 0 || K <= 0 || K > AX_MAXK || (H & 3) || (ldq & 3) || F <= 0 || F > AX_MAXF || (F & 3) ||
      (ldw & 3) || E <= 0 || !M || !q || !wa || !mask || !O || !Wl || !alpha || !att || !x2 ||
      (((uintptr_t)Wl | (uintptr_t)M | (uintptr_t)q | (uintptr_t)wa) & 15))
    return -1;
  hipLaunchKernelGGL(attn_x2_kernel, dim3((E + AX_OC - 1) / AX_OC, R), dim3(256), 0,
                     (hipStream_t)stream, K, H, F, E, M, q, ldq, wa, mask, O, Wl, ldw, bias, add,
                     ld_add, alpha, att, lda, x2, ldx2);
  return chk("attn_x2_fwd");
}

// dM (R x K x H) and dwa_rows (R x H) ACCUMULATE (the caller zeroes them once and
// sums dwa_rows over the rows at the end); dq (R x H) is overwritten.
extern "C" int s2c_attn_bwd(int R, int K, int H, int F, const float *datt, int ldd,
                            const float *att, int lda, const float *alpha,
                            const float *O, const float *M, const float *q, int ldq,
                            const float *wa, float *dM, float *dq, float *dwa_rows,
                            void *stream) {
  if (R <= 0 || K <= 0 || K > 1024 || F < 32 || F > 256 || (F & (F - 1)) || (H & 3) ||
      (ldq & 3) || (ldd & 3))
    return -1;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((H + ATT_HC - 1) / ATT_HC, R), dim3(256), 0,
                     (hipStream_t)stream, K, H, F, datt, ldd, att, lda, alpha, O, M, q,
                     ldq, wa, dM, dq, dwa_rows);
  return chk("attn_bwd");
}

// The attention backward of a step for K <= 32 keys with the map_lang^T product in front (see
// attn_bwd_x2_kernel): da2 (R x E, row stride ldda) = the gradient of map_lang's pre-activation,
// WlT = W_lang^T ((F + H) x E, rows f < F are read).  dM / dwa_rows accumulate, dq (row stride
// lddq) is overwritten.
extern "C" int s2c_attn_bwd_x2(int R, int K, int H, int F, int E, const float *da2, int ldda,
                               const float *WlT, int ldw, const float *att, int lda,
                               const float *alpha, const float *O, const float *M,
                               const float *q, int ldq, const float *wa, float *dM, float *dq,
                               int lddq, float *dwa_rows, void *stream) {
  if (R <= 0 || K <= 0 || K > AX_MAXK || F < 32 || F > AX_MAXF || (F & (F - 1)) || (H & 3) ||
      E <= 0 || E > 512 || (E & 3) || (ldq & 3) || (ldw & 3) || (lddq & 3) || !da2 || !WlT || !att ||
      !alpha || !O || !M || !q || !wa || !dM || !dq || !dwa_rows ||
      (((uintptr_t)WlT | (uintptr_t)O | (uintptr_t)M | (uintptr_t)q | (uintptr_t)wa |
        (uintptr_t)dM | (uintptr_t)dq | (uintptr_t)dwa_rows) & 15))
    return -1;
  hipLaunchKernelGGL(attn_bwd_x2_kernel, dim3(R), dim3(ABX_T), 0, (hipStream_t)stream, K, H, F, E, da2, ldda, WlT, ldw, att, lda, alpha, O, M,
                     q, ldq, wa, dM, dq, lddq, dwa_rows);
  return chk("attn_bwd_x2");
}

static int attn_local_launch(int R, int L, int H, int F, const float *mapped, const float *q,
                             int ldq, const float *wa, float ba, const float *valid,
                             const float *feats, float *alpha, float *att, int lda,
                             unsigned short *planes, long long pstride, int ldp, int tiled,
                             void *stream) {
  if (R <= 0 || L <= 0 || L > AL_MAXL || (H & 3) || (ldq & 3) || F <= 0 || !mapped || !q ||
      !wa || !feats || !alpha || (!att && !planes))
    return -1;
  if (L == 10)       // CONF default num_locals (scripts/train.py:332)
    hipLaunchKernelGGL(attn_local_kernel<10>, dim3((R + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, R, L, H, F, mapped, q, ldq, wa, ba, valid, feats,
                       alpha, att, lda, planes, pstride, ldp, tiled);
  else
    hipLaunchKernelGGL(attn_local_kernel<0>, dim3((R + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, R, L, H, F, mapped, q, ldq, wa, ba, valid, feats,
                       alpha, att, lda, planes, pstride, ldp, tiled);
  return chk("attn_local_fwd");
}

extern "C" int s2c_attn_local_fwd(int R, int L, int H, int F, const float *mapped,
                                  const float *q, int ldq, const float *wa, float ba,
                                  const float *valid, const float *feats, float *alpha,
                                  float *att, int lda, void *stream) {
  return attn_local_launch(R, L, H, F, mapped, q, ldq, wa, ba, valid, feats, alpha, att, lda,
                           nullptr, 0, 0, 0, stream);
}

extern "C" int s2c_attn_local_fwd_planes(int R, int L, int H, int F, const float *mapped,
                                         const float *q, int ldq, const float *wa, float ba,
                                         const float *valid, const float *feats, float *alpha,
                                         float *att, int lda, unsigned short *planes,
                                         long long pstride, int ldp, int tiled, void *stream) {
  if (planes == nullptr || ldp < F || (tiled && (ldp & 15))) return -1;
  return attn_local_launch(R, L, H, F, mapped, q, ldq, wa, ba, valid, feats, alpha, att, lda,
                           planes, pstride, ldp, tiled, stream);
}

