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
// Lane layout of the small-batch matrix-vector kernels: a wave owns one output
// feature; lane = (row r = lane & 7, chunk c = lane >> 3).  The 8 chunk lanes of a
// row read consecutive float4s of the weight row (128 B per load instruction, the
// 8 row lanes of a chunk share the address), each lane accumulates its row's
// partial dot product, and 3 xor-shuffles fold the chunks.  R <= 8 rows per pass.
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
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// partial dot product of one weight row with this lane's input row; the 32 chunk
// lanes of a row (8 per wave x 4 waves of the block) stride over the float4s.
// Two independent accumulators / unroll 4 keep several 16-byte loads in flight
// (these kernels are pure latency: ~1 wave per SIMD, a few KB per wave).
constexpr int NCHUNK = 32;
__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
// U float4 pairs per lane, every load issued before the first use: one memory
// round trip per call instead of one per loop iteration.
template <int U>
__device__ __forceinline__ void dot_row_u(const float4 *__restrict__ w,
                                          const float4 *__restrict__ x, int n4, int j0,
                                          float &acc0, float &acc1) {
  float4 a[U], b[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int j = j0 + u * NCHUNK;
    if (j < n4) {
      a[u] = w[j];
      b[u] = x[j];
    } else {
      a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      b[u] = a[u];
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (u & 1) acc1 += dot4(a[u], b[u]);
    else acc0 += dot4(a[u], b[u]);
  }
}
__device__ __forceinline__ float dot_row(const float *__restrict__ w,
                                         const float *__restrict__ x, int n4,
                                         int chunk) {
  const float4 *w4 = reinterpret_cast<const float4 *>(w);
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  float acc0 = 0.0f, acc1 = 0.0f;
  if (n4 <= 4 * NCHUNK) {
    dot_row_u<4>(w4, x4, n4, chunk, acc0, acc1);
  } else {
    for (int j = chunk; j < n4; j += 12 * NCHUNK) dot_row_u<12>(w4, x4, n4, j, acc0, acc1);
  }
  return acc0 + acc1;
}

// block-wide fold of per-lane partials: result valid in threads 0..7 (row = tid)
template <int NV>
__device__ __forceinline__ void block_fold(float (&v)[NV], float (*s_part)[4][RB]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    v[k] = fold_chunks(v[k]);
    if (lane < RB) s_part[k][wave][lane] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < RB) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
      v[k] = (s_part[k][0][threadIdx.x] + s_part[k][1][threadIdx.x]) +
             (s_part[k][2][threadIdx.x] + s_part[k][3][threadIdx.x]);
  }
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

__global__ __launch_bounds__(256) void small_linear_kernel(
    int R, s2c_lin_desc p1, s2c_lin_desc p2, s2c_gru_bwd_desc g1, int has_g1) {
  __shared__ float s_part[1][4][RB];
  const int r = threadIdx.x & 7, chunk = threadIdx.x >> 3;   // 32 chunks
  const bool second = (int)blockIdx.x >= p1.O;
  const s2c_lin_desc &p = second ? p2 : p1;
  const int o = second ? (int)blockIdx.x - p1.O : (int)blockIdx.x;
  const int row = blockIdx.y * RB + r;
  const int rowc = row < R ? row : R - 1;
  float v[1];
  v[0] = dot_row(p.W + (size_t)o * p.ldw, p.x + (size_t)rowc * p.ldx, p.I >> 2, chunk);
  block_fold<1>(v, s_part);
  if (threadIdx.x < RB && row < R) {
    float acc = v[0];
    if (p.bias) acc += p.bias[o];
    if (p.add1) acc += p.add1[(size_t)row * p.ld1 + o];
    if (p.add2) acc += p.add2[(size_t)row * p.ld2 + o];
    if (p.epi == 1) acc = fmaxf(acc, 0.0f);
    else if (p.epi == 2) acc = p.gate[(size_t)row * p.ldg + o] > 0.0f ? acc : 0.0f;
    if (p.out) p.out[(size_t)row * p.ldo + o] = acc;
    if (has_g1 && !second) gru_gates_bwd_one(g1, p1.O, row, o, acc);
  }
}

// ---------------------------------------------------------------------------
// GRUCell forward (torch.nn.GRUCell semantics):
//   gi = W_ih x + b_ih ; gh = W_hh h + b_hh          (gate order r, z, n)
//   r = sig(gi_r + gh_r) ; z = sig(gi_z + gh_z) ; n = tanh(gi_n + r * gh_n)
//   h' = (1 - z) * n + z * h
// One wave per hidden unit; saves r, z, n and gh_n (bias included) for BPTT.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru_fwd_kernel(
    int R, int H, int I, const float *__restrict__ Wih,
    const float *__restrict__ Whh, const float *__restrict__ bih,
    const float *__restrict__ bhh, const float *__restrict__ x, int ldx,
    const float *__restrict__ h, float *__restrict__ hnew,
    float *__restrict__ sr, float *__restrict__ sz, float *__restrict__ sn,
    float *__restrict__ sghn) {
  __shared__ float s_part[6][4][RB];
  const int r = threadIdx.x & 7, chunk = threadIdx.x >> 3;
  const int u = blockIdx.x;                      // one block per hidden unit
  const int row = blockIdx.y * RB + r;
  const int rowc = row < R ? row : R - 1;
  const float *xr = x + (size_t)rowc * ldx;
  const float *hr = h + (size_t)rowc * H;
  float g[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    g[k] = dot_row(Wih + (size_t)(k * H + u) * I, xr, I >> 2, chunk);
    g[3 + k] = dot_row(Whh + (size_t)(k * H + u) * H, hr, H >> 2, chunk);
  }
  block_fold<6>(g, s_part);
  if (threadIdx.x < RB && row < R) {
    const float gir = g[0] + bih[u], giz = g[1] + bih[H + u], gin = g[2] + bih[2 * H + u];
    const float ghr = g[3] + bhh[u], ghz = g[4] + bhh[H + u], ghn = g[5] + bhh[2 * H + u];
    const float rr = sigmoidf_(gir + ghr);
    const float zz = sigmoidf_(giz + ghz);
    const float nn = tanhf(gin + rr * ghn);
    const float hp = hr[u];
    const size_t e = (size_t)row * H + u;
    hnew[e] = (1.0f - zz) * nn + zz * hp;
    sr[e] = rr; sz[e] = zz; sn[e] = nn; sghn[e] = ghn;
  }
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
    acc += w.x * tanhf(a.x + b.x) + w.y * tanhf(a.y + b.y) +
           w.z * tanhf(a.z + b.z) + w.w * tanhf(a.w + b.w);
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

// Attention backward for one step, block = (chunk of ATT_KC keys, row):
//  (a) softmax backward without a row-wide pass:  sum_k alpha_k dalpha_k
//      = <datt, sum_k alpha_k O_k> = <datt, att>  (att is the saved forward output);
//      dalpha_k = <datt, O[r,k,:]> ; ds_k = alpha_k (dalpha_k - <datt, att>)
//  (b) dpre = ds * wa * (1 - c^2), c = tanh(M + q);
//      dM[r,k,:] += dpre ; dq[r,:] += sum_k dpre ; dwa[:] += sum_k ds * c
//      (k loop keeps the dq / dwa partial sums in registers: one atomic per (block, h)).
// dO = sum_t alpha_t (outer) datt_t has no recurrence and is one batched GEMM after
// the time loop (decoder_fused.py).
constexpr int ATT_KC = 16;
__global__ __launch_bounds__(256) void attn_bwd_kernel(
    int K, int H, int F, const float *__restrict__ datt, int ldd,
    const float *__restrict__ att, int lda, const float *__restrict__ alpha,
    const float *__restrict__ O, const float *__restrict__ M,
    const float *__restrict__ q, int ldq, const float *__restrict__ wa,
    float *__restrict__ dM, float *__restrict__ dq, float *__restrict__ dwa) {
  __shared__ float s_datt[512];
  __shared__ float s_red[4];
  __shared__ float s_ds[ATT_KC];
  const int row = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = blockIdx.x * ATT_KC;
  const int k1 = min(K, k0 + ATT_KC);
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
  {
    // 16 lanes per key
    const int kk = k0 + (tid >> 4), sub = tid & 15;
    float da = 0.0f;
    if (kk < k1) {
      const float *o = O + ((size_t)row * K + kk) * F;
      for (int f = sub; f < F; f += 16) da += s_datt[f] * o[f];
    }
    da += __shfl_xor(da, 1, 64);
    da += __shfl_xor(da, 2, 64);
    da += __shfl_xor(da, 4, 64);
    da += __shfl_xor(da, 8, 64);
    if (sub == 0 && kk < k1) {
      const float a = alpha[(size_t)row * K + kk];
      s_ds[kk - k0] = a * (da - dot);
    }
  }
  __syncthreads();
  for (int h = tid; h < H; h += 256) {
    const float qh = q[(size_t)row * ldq + h], wh = wa[h];
    float sq = 0.0f, sw = 0.0f;
    for (int k = k0; k < k1; ++k) {
      const float d = s_ds[k - k0];              // block-uniform
      if (d == 0.0f) continue;                   // masked (alpha == 0)
      const size_t e = ((size_t)row * K + k) * H + h;
      const float c = tanhf(M[e] + qh);
      const float dp = d * wh * (1.0f - c * c);
      dM[e] += dp;
      sq += dp;
      sw += d * c;
    }
    atomicAdd(dq + (size_t)row * H + h, sq);
    atomicAdd(dwa + h, sw);
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
  return p->O <= 0 || p->I <= 0 || (p->I & 3) || (p->ldw & 3) || (p->ldx & 3) ||
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
  const int O = p1->O + (p2 ? p2->O : 0);
  hipLaunchKernelGGL(small_linear_kernel, dim3(O, (R + RB - 1) / RB), dim3(256), 0,
                     (hipStream_t)stream, R, *p1, q2, g, g1 ? 1 : 0);
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
  if (R <= 0 || (H & 3) || (I & 3) || (ldx & 3)) return -1;
  hipLaunchKernelGGL(gru_fwd_kernel, dim3(H, (R + RB - 1) / RB), dim3(256),
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

// dM (R x K x H), dq (R x H) and dwa (H) ACCUMULATE (the caller zeroes them once).
extern "C" int s2c_attn_bwd(int R, int K, int H, int F, const float *datt, int ldd,
                            const float *att, int lda, const float *alpha,
                            const float *O, const float *M, const float *q, int ldq,
                            const float *wa, float *dM, float *dq, float *dwa,
                            void *stream) {
  if (R <= 0 || K <= 0 || K > 1024 || F > 512 || (H & 3)) return -1;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((K + ATT_KC - 1) / ATT_KC, R), dim3(256), 0,
                     (hipStream_t)stream, K, H, F, datt, ldd, att, lda, alpha, O, M, q,
                     ldq, wa, dM, dq, dwa);
  return chk("attn_bwd");
}
