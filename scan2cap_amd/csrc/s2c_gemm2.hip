// s2c_gemm2.hip -- streaming variant of the bf16x3 rows GEMM for the tall layers of the
// first set-abstraction stage (M ~ 1M rows, N <= 128, K <= 160):
//
//     Y[M x N] = pro(A)[M x K] * W^T  (+ per-column [sum | sumsq] partials)
//
// Same arithmetic as rows_gemm_x3_kernel (s2c_gemm.hip): every fp32 product is the six
// bf16 plane products with i + j <= 2 of a 3-way split, accumulated in fp32 in the same
// term order.  What changes is the skeleton.  The phase timeline of the tiled kernel
// (tools/prof_gemm.py) shows its workgroups stalled at the ISSUE of their loads and stores
// for 2/3 of their life: one tile per workgroup, barrier-separated phases, two workgroups
// per CU -- nothing is in flight while a workgroup computes or stores.  Here:
//
//   * persistent workgroups (one per CU, 8 waves), every WAVE an independent pipeline over
//     32-row tiles: no workgroup barrier after the weight staging;
//   * the operand never passes through VGPRs: `global_load_lds_dwordx4` (LDS-DMA) fills a
//     wave-private ring of 4 KB chunks (32 rows x 32 k, fp32), DEPTH chunks ahead, counted
//     with `s_waitcnt vmcnt(N)` -- loads of the next tile are in flight under the MFMAs and
//     the stores of the current one (tools/bench_stream.py: the bare ring copies
//     268 MB -> 268 MB at 5.1-5.3 TB/s);
//   * the DMA writes lane-linear, so the XOR swizzle that makes the operand reads
//     (ds_read_b128, one row per lane) conflict-free sits on the SOURCE address;
//   * W is split to bf16 planes once per workgroup and stays resident in LDS in operand
//     order ([plane][k-block of 8][column] x 16 B);
//   * the fp32 -> 3 x bf16 split of the activations happens on the operand read;
//   * the accumulator tile leaves through 4x4 DPP transposes as dwordx4 stores (8 rows x
//     128 B per instruction) -- no LDS round trip, no dword stores;
//   * PRO_GATHER: the ball-query grouping is fused as in s2c_gemm.hip, now entirely by DMA:
//     the neighbour ids of the next tile, the xyz rows (dword pieces) and the feature rows (only
//     4-byte aligned in the (B,N,3+C) cloud: LDS-DMA takes dword-aligned dwordx4 sources)
//     land in LDS without a VGPR load in the loop.  K is walked feature columns first, the
//     three centred coordinates last (W is permuted to match while it is staged).
//
// LDS (160 KB) decides the shapes this kernel takes: W planes 6 B per element plus one ring per
// wave -- 8 waves x 3 chunks when W is small, 4 waves x 4, 3 or 2 chunks for N = 128 with
// K ~ 128 (s2c_rows_stream_supported); everything else stays on the tiled kernel.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>
#include <stdlib.h>

using namespace s2c;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CHUNK_BYTES = 4096; // 32 rows x 32 floats
constexpr int AUX_BYTES = 1088;   // gather: idx[2][32] int, xyz[2][32][3], centres[2][2][3] float

enum { SPRO_NONE = 0, SPRO_BNRELU = 1, SPRO_GATHER = 2, SPRO_POOLBWD = 3 };
constexpr int PB_AUX_BYTES = 3072;   // pool-backward: arg[2][2][<=128] int16, dk[2][2][<=128] float

struct StreamArgs {
  long long M;
  int N, K;                 // K = columns of the logical operand (gather: 3 + C)
  const float *A; int lda;  // SPRO_NONE / SPRO_BNRELU
  // SPRO_BNRELU: operand = relu?(A * scale[k] + shift[k]) (the previous layer's BatchNorm +
  // ReLU, the arithmetic of bn_relu_kernel), written back to `side` (M x K, row stride lds_)
  // when side != nullptr -- the activation the backward pass needs leaves with the GEMM that
  // consumes it instead of through a separate pass over the pre-activation tensor
  const float *scale, *shift; int relu; float *side; int ld_side;
  // inference epilogue (ep_mean != nullptr): out = relu?(acc * sc[c] + sh[c]) with
  // sc = gamma / sqrt(var + eps), sh = beta - mean * sc (affine_epilogue of s2c_gemm.hip),
  // optionally max-pooled over groups of pool_ns consecutive rows; written to Y (row stride ldy)
  const float *ep_gamma, *ep_beta, *ep_mean, *ep_var; float ep_eps; int ep_relu, pool_ns;
  // training, pooled last layer (ext != nullptr): per (centre, column) the extremum of Y over
  // the pool_ns rows of the centre that the BatchNorm + ReLU + max-pool behind it will pick, and
  // its FIRST sample index: relu(y * scale + shift) is monotone in y, its maximum sits at max y
  // for scale >= 0 and at min y otherwise, and the sign of scale = gamma * invstd is the sign of
  // gamma, known BEFORE the statistics.  The sign is folded into the staged W planes (row n of
  // W times -1 where ext_sign[n] < 0: planes, products and sums negate exactly), so the
  // epilogue tracks ONE running maximum per column; ext = sign * max, the statistics partials
  // get their sign back when they are written.  Y itself need not be written (Y == nullptr).
  // (A first version tracked max AND min with their indices: the 32 x 128 tile's epilogue was
  // ~1100 VALU instructions against 156 in its four k-steps -- 45 % issue-active waves, two per
  // SIMD: the kernel was bound by the vector ALU, not by memory or the matrix cores.)
  float *ext; int *aext; const float *ext_sign;
  // EPI_NEXT (nY != nullptr): the output Y is the upstream gradient of a layer whose BatchNorm
  // backward starts with s1 = sum dz, s2 = sum dz (nY - nmean) ninvstd, dz = Y [nY nscale + nshift
  // > 0] (bn_bwd_stats_kernel, s2c_sa.hip): those sums go to `partial` instead of (sum, sumsq)
  const float *nY, *nscale, *nshift, *nmean, *ninvstd; int nrelu;
  // SPRO_POOLBWD: input gradient of a max-pooled BatchNorm layer WITHOUT its (M x C3) tensors.
  // With Y3 = A W3^T the layer's dY3 = dkrow - g (.) Y3 + e per channel (g = k0 k2 invstd,
  // e = g mean - k0 k1; dkrow = k0 * routed upstream gradient, one nonzero per centre and
  // channel), so dA = dY3 W3 = dkrow W3 - A G + cvec with G = W3^T diag(g) W3 (K x K).  The
  // operand is [A (DMA, KA = pb_ka columns) | dkrow (generated from arg / dk, pb_c3 columns)],
  // W = [-G^T | W3^T] (N x (KA + C3)), `bias` = cvec.  pb_arg / pb_dk: (J, C3), pb_ns rows per
  // centre.
  const short *pb_arg; const float *pb_dk; const float *bias; int pb_ka, pb_c3, pb_ns;
  const float *W; int ldw;
  float *Y; int ldy;
  float *partial; int partial_rows;
  // SPRO_GATHER
  const float *xyz, *new_xyz, *feats;
  const int *idx;
  long long frs, fbs;
  int n, m, ns, C;
  float radius; int normalize;
};

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

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

// 8 consecutive-k fp32 values -> the three bf16x8 operand planes
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

// NT: 32-column MFMA tiles per wave (N <= 32 NT); WAVES per workgroup (one workgroup per CU);
// SLOTS: ring slots per wave, SLOTS - 1 chunks requested ahead of the one being consumed.
// EPI: 0 = Y + statistics partials (training), 1 = raw per-centre extrema (+ statistics; pooled
// training layer), 2 = inference epilogue (BatchNorm + ReLU (+ max-pool)).  A template parameter
// because each keeps its own per-lane state: as run-time branches the three together spilled
// the NT = 4 kernels to scratch.
// EPI_NEXT: EPI_TRAIN whose two column sums are those of the NEXT BatchNorm backward (StreamArgs::nY)
enum { EPI_TRAIN = 0, EPI_RAW = 1, EPI_EVAL = 2, EPI_NEXT = 3 };

// Diagnostics (tools/prof_stream.py): wave 0 of workgroup g_sprof_block adds up the shader-clock
// cycles it spends per tile waiting for its ring chunks / in the k-steps / in the epilogue:
// g_sprof[0..3] = {waits, k-steps, epilogue, tiles}, [4] = life of the tile loop, [5] = prologue.
__device__ long long *g_sprof = nullptr;
__device__ int g_sprof_block = 0;
#define SP_NOW() ([&]() { __builtin_amdgcn_sched_barrier(0); long long t_ = (long long)__builtin_amdgcn_s_memtime(); \
                          __builtin_amdgcn_sched_barrier(0); return t_; }())
template <int NT, int PRO, int WAVES, int SLOTS, int EPI>
__global__ __launch_bounds__(64 * WAVES, (NT == 4 && WAVES == 4) ? 1 : 2) void rows_stream_gemm_kernel(StreamArgs p) {
  const long long sp_entry = g_sprof != nullptr ? (long long)__builtin_amdgcn_s_memtime() : 0;
  constexpr int NP = 32 * NT;
  constexpr int DEPTH = SLOTS - 1;
  constexpr int AUXB = PRO == SPRO_GATHER ? AUX_BYTES : (PRO == SPRO_POOLBWD ? PB_AUX_BYTES : 0);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const long long M = p.M;
  const int N = p.N, K = p.K;
  const int KA = PRO == SPRO_GATHER ? p.C : (PRO == SPRO_POOLBWD ? p.pb_ka : K);   // DMA columns
  const int KC = (KA + 31) >> 5;                 // DMA chunks per tile
  const int KS = (K + 15) >> 4;                  // k16 steps per tile
  const int KB = 2 * KS;                         // k-blocks of 8 in the W planes
  const int KCT = (KS + 1) >> 1;                 // chunk iterations (the last may be xyz only)

  unsigned char *wp = smem;
  const unsigned wbytes = 3u * KB * NP * 16u;
  const unsigned ssbytes = PRO == SPRO_BNRELU ? 2u * KB * 8u * 4u : 0u;   // scale | shift, KB*8 floats each
  float *s_scale = reinterpret_cast<float *>(smem + wbytes);
  float *s_shift = s_scale + KB * 8;
  unsigned char *mine = smem + wbytes + ssbytes + (unsigned)wave * (SLOTS * CHUNK_BYTES + AUXB);
  unsigned char *ring = mine;
  int *idxbuf = reinterpret_cast<int *>(mine + SLOTS * CHUNK_BYTES);          // [2][32]
  float *xyzbuf = reinterpret_cast<float *>(mine + SLOTS * CHUNK_BYTES + 256); // [2][32][3]
  short *pb_argbuf = reinterpret_cast<short *>(mine + SLOTS * CHUNK_BYTES);        // [2][2][128]
  float *pb_dkbuf = reinterpret_cast<float *>(mine + SLOTS * CHUNK_BYTES + 1024);  // [2][2][128]
  const unsigned pbarg_lds = (unsigned)(size_t)pb_argbuf, pbdk_lds = (unsigned)(size_t)pb_dkbuf;
  const unsigned ring_lds = (unsigned)(size_t)ring;
  float *ctrbuf = reinterpret_cast<float *>(mine + SLOTS * CHUNK_BYTES + 1024);  // [2][2][3]
  const unsigned idx_lds = (unsigned)(size_t)idxbuf, xyz_lds = (unsigned)(size_t)xyzbuf;
  const unsigned ctr_lds = (unsigned)(size_t)ctrbuf;

  // ---- stage W: split to planes, operand order, k permuted for the gather ----------------
  {
    const int quads = KB * 2;                    // k-quads per row of W'
    for (int e = tid; e < NP * quads; e += 64 * WAVES) {
      const int n = e / quads, k0 = (e - n * quads) * 4;
      float w[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = k0 + c;
        int src = -1;
        if (PRO == SPRO_GATHER) {
          if (k < KA) src = 3 + k; else if (k < KA + 3) src = k - KA;
        } else if (k < K) {
          src = k;
        }
        w[c] = (n < N && src >= 0) ? p.W[(long long)n * p.ldw + src] : 0.f;
      }
      if (EPI == EPI_RAW && p.ext_sign != nullptr && n < N && p.ext_sign[n] < 0.f) {
        w[0] = -w[0]; w[1] = -w[1]; w[2] = -w[2]; w[3] = -w[3];
      }
      unsigned h0, m0, l0, h1, m1, l1;
      split2((f32x2){w[0], w[1]}, h0, m0, l0);
      split2((f32x2){w[2], w[3]}, h1, m1, l1);
      unsigned char *d = wp + ((unsigned)(k0 >> 3) * NP + n) * 16u + (k0 & 7) * 2;
      *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
      *reinterpret_cast<uint2 *>(d + (unsigned)KB * NP * 16u) = make_uint2(m0, m1);
      *reinterpret_cast<uint2 *>(d + 2u * KB * NP * 16u) = make_uint2(l0, l1);
    }
    if (PRO == SPRO_BNRELU)
      for (int k = tid; k < KB * 8; k += 64 * WAVES) {
        s_scale[k] = k < K ? p.scale[k] : 1.f;
        s_shift[k] = k < K ? p.shift[k] : 0.f;
      }
    // ring + aux start as zeros: positions the DMA never writes (k >= KA of the last chunk)
    // only ever hold zeros or stale FINITE activations, and meet zero weights
    for (int e = lane; e < (SLOTS * CHUNK_BYTES + AUXB) / 16; e += 64)
      reinterpret_cast<uint4 *>(mine)[e] = make_uint4(0, 0, 0, 0);
    if (PRO == SPRO_POOLBWD && p.scale != nullptr) {
      // the operand's BatchNorm + ReLU constants (KA <= 64, pb_ns >= 32: the second centre's half of this
      // wave's dk buffer is never written by issue_pb) -- a copy per wave, same-wave program order
      pb_dkbuf[128 + lane] = lane < KA ? p.scale[lane] : 1.f;
      pb_dkbuf[192 + lane] = lane < KA ? p.shift[lane] : 0.f;
    }
  }
  __syncthreads();

  const long long tiles = (M + 31) >> 5;
  const long long wid = (long long)blockIdx.x * WAVES + wave, nw = (long long)gridDim.x * WAVES;

  // the n-th tile of this wave: tiles of one pooling group (pool_ns = 64: two tiles) stay
  // with one wave, groups are dealt round-robin over all waves of the grid
  const int TG = p.pool_ns == 64 ? 2 : 1;
  auto tile_at = [&](long long n) -> long long {
    return TG == 1 ? wid + n * nw : (wid + (n >> 1) * nw) * 2 + (n & 1);
  };

  // ---- issue cursor ---------------------------------------------------------------------
  long long it_n = 0, it_tile = tile_at(0);
  int it_c = 0, it_slot = 0, it_par = 0;
  const float *rp[4];                            // source rows of this lane's 4 DMA pieces
  const int dma_r0 = lane >> 3;                  // row of piece i: 8 i + dma_r0
  // piece i, lane: physical quad (lane & 7) of row r holds logical quad (lane & 7) ^ ((r >> 1) & 7)
  auto issue_idx = [&](long long t, int par) {   // neighbour ids of tile t -> idxbuf[par]
    if (lane < 32) {
      long long row = t * 32 + lane;
      if (row >= M) row = M - 1;
      glds4(p.idx + row, idx_lds + par * 128);
    }
  };
  auto setup_rows = [&](long long t, int par) {
    if (PRO == SPRO_GATHER) {
      // neighbour ids of tile t landed a tile ago (covered by the chunk waits since)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        long long row = t * 32 + 8 * i + dma_r0;
        if (row >= M) row = M - 1;
        const long long b = row / ((long long)p.m * p.ns);
        const int pt = idxbuf[par * 32 + 8 * i + dma_r0];
        rp[i] = p.feats + b * p.fbs + (long long)pt * p.frs;
      }
      if (lane < 32) {
        long long row = t * 32 + lane;
        if (row >= M) row = M - 1;
        const long long b = row / ((long long)p.m * p.ns);
        const int pt = idxbuf[par * 32 + lane];
        const float *src = p.xyz + (b * p.n + pt) * 3;      // -> xyzbuf[par][component][row]
        glds4(src, xyz_lds + par * 384);
        glds4(src + 1, xyz_lds + par * 384 + 128);
        glds4(src + 2, xyz_lds + par * 384 + 256);
      }
      if (lane < 2) {                            // the (one or two) centres of the tile
        long long row = t * 32 + 16 * lane;
        if (row >= M) row = M - 1;
        const float *src = p.new_xyz + (row / p.ns) * 3;    // -> ctrbuf[par][component][2]
        glds4(src, ctr_lds + par * 24);
        glds4(src + 1, ctr_lds + par * 24 + 8);
        glds4(src + 2, ctr_lds + par * 24 + 16);
      }
      const long long tn = tile_at(it_n + 1);
      if (tn < tiles) issue_idx(tn, par ^ 1);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        long long row = t * 32 + 8 * i + dma_r0;
        if (row >= M) row = M - 1;
        rp[i] = p.A + row * p.lda;
      }
    }
  };
  // SPRO_POOLBWD: arg / dk rows of a tile's centre (two centres when pb_ns = 16), 16 B per lane.
  // Requested at the START of the previous tile of this wave (two buffers by tile parity): the
  // issue cursor runs up to two tiles ahead of the MFMAs, so tying these to it would overwrite a
  // buffer whose tile has not reached its generated columns yet.
  auto issue_pb = [&](long long t, int par) {
    if (PRO == SPRO_POOLBWD) {
      const int c3 = p.pb_c3;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1 && p.pb_ns != 16) continue;
        long long row = t * 32 + 16 * h;
        if (row >= M) row = M - 1;
        const long long ctr = row / p.pb_ns;
        if (lane * 8 < c3) glds16(p.pb_arg + ctr * c3 + lane * 8, pbarg_lds + (par * 2 + h) * 256);
        if (lane * 4 < c3) glds16(p.pb_dk + ctr * c3 + lane * 4, pbdk_lds + (par * 2 + h) * 512);
      }
    }
  };
  auto issue_next = [&]() -> bool {
    if (it_tile >= tiles) return false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * i + dma_r0;
      const int q = (lane & 7) ^ ((r >> 1) & 7);
      const int k = it_c * 32 + q * 4;
      if (k < KA) glds16(rp[i] + k, ring_lds + it_slot * CHUNK_BYTES + i * 1024);
    }
    it_slot = it_slot + 1 == SLOTS ? 0 : it_slot + 1;
    if (++it_c == KC) {
      it_c = 0;
      it_tile = tile_at(++it_n);
      it_par ^= 1;
      if (it_tile < tiles) setup_rows(it_tile, it_par);
    }
    return true;
  };

  if (it_tile < tiles) {
    if (PRO == SPRO_GATHER) {
      issue_idx(it_tile, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    setup_rows(it_tile, 0);
    issue_pb(it_tile, 0);
  }
#pragma unroll 1
  for (int d = 0; d < DEPTH; ++d) issue_next();

  float s1[NT], s2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;

  // inference epilogue coefficients of this lane's columns
  constexpr int NE = EPI == EPI_EVAL ? NT : 1, NR = EPI == EPI_RAW ? NT : 1;
  float esc[NE], esh[NE], gmax[NE][2];
  if constexpr (EPI == EPI_EVAL) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = 32 * j + li;
      esc[j] = esh[j] = 0.f;
      if (col < N) {
        const float invstd = 1.0f / sqrtf(p.ep_var[col] + p.ep_eps);
        esc[j] = (p.ep_gamma ? p.ep_gamma[col] : 1.0f) * invstd;
        esh[j] = (p.ep_beta ? p.ep_beta[col] : 0.0f) - p.ep_mean[col] * esc[j];
      }
      gmax[j][0] = gmax[j][1] = -INFINITY;
    }
  }

  // SPRO_POOLBWD per-lane constants (kept in VGPRs: the scalar registers are over-subscribed
  // and a kernarg re-load inside the k loop stalls a wave that has no sibling on its SIMD)
  const int pb_srow0 = PRO == SPRO_POOLBWD ? (p.pb_ns == 16 ? (li & 15) : li) + 0 * lane : 0;
  const int pb_sodd = PRO == SPRO_POOLBWD ? (p.pb_ns == 64 ? 32 : 0) + 0 * lane : 0;
  const int pb_hoff = PRO == SPRO_POOLBWD ? (p.pb_ns == 16 ? (li >> 4) * 128 : 0) + 0 * lane : 0;
  const int pb_c3v = PRO == SPRO_POOLBWD ? p.pb_c3 + 0 * lane : 0;
  float pbias[NT];                               // SPRO_POOLBWD: cvec of this lane's columns
#pragma unroll
  for (int j = 0; j < NT; ++j)
    pbias[j] = (PRO == SPRO_POOLBWD && p.bias != nullptr && 32 * j + li < N) ? p.bias[32 * j + li] : 0.f;
  float rmax[NR][2], esgn[NR];                   // EPI_RAW: running maximum of sign * y
  int ramax[NR][2];                              // its row inside the centre, less 4 lk
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    esgn[j] = (EPI == EPI_RAW && p.ext_sign != nullptr && 32 * j + li < N &&
               p.ext_sign[32 * j + li] < 0.f) ? -1.f : 1.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) { rmax[j][h] = -INFINITY; ramax[j][h] = 0; }
  }

  constexpr int NX = EPI == EPI_NEXT ? NT : 1;
  float nsc[NX], nsh[NX], nmu[NX], nis[NX];
  if constexpr (EPI == EPI_NEXT) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = 32 * j + li;
      const bool ok = col < N;
      nsc[j] = ok ? p.nscale[col] : 0.f; nsh[j] = ok ? p.nshift[col] : 0.f;
      nmu[j] = ok ? p.nmean[col] : 0.f; nis[j] = ok ? p.ninvstd[col] : 0.f;
    }
  }
  const int swz = (li >> 1) & 7;
  int slot = 0, par = 0;
  long long *sprof = g_sprof;
  const bool sp_on = sprof != nullptr && (int)blockIdx.x == g_sprof_block && wave == 0;
  long long sp_wait = 0, sp_k = 0, sp_epi = 0, sp_tiles = 0, sp_t0 = 0, sp_life0 = 0;
  if (sp_on) sp_life0 = SP_NOW();
#pragma unroll 1
  for (long long n = 0; tile_at(n) < tiles; ++n) {
    const long long t = tile_at(n);
    if (sp_on) sp_t0 = SP_NOW();
    if (PRO == SPRO_POOLBWD && tile_at(n + 1) < tiles) issue_pb(tile_at(n + 1), par ^ 1);
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const long long r0 = t * 32;
    // EPI_NEXT: the next layer's pre-activations of this lane's accumulator elements, requested
    // BEFORE the k-steps (they are first used in the epilogue: requested there, every tile paid
    // one exposed memory round trip -- 87 us of the launch at SA1)
    float ny[NX][EPI == EPI_NEXT ? 16 : 1];
    if constexpr (EPI == EPI_NEXT) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const long long row = r0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
          const int col = 32 * j + li;
          ny[j][e] = (row < M && col < N) ? p.nY[row * (long long)N + col] : 0.f;
        }
    }

#pragma unroll 1
    for (int c = 0; c < KCT; ++c) {
      const unsigned char *sl = ring + slot * CHUNK_BYTES;
      if (c < KC) {
        long long sp_a = 0;
        if (sp_on) sp_a = SP_NOW();
        if (issue_next()) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * DEPTH) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (sp_on) sp_wait += SP_NOW() - sp_a;
        if (PRO == SPRO_BNRELU || (PRO == SPRO_POOLBWD && p.scale != nullptr)) {
          // (SPRO_POOLBWD with scale: the layer's input activation is recomputed from the previous
          // layer's pre-activation -- the forward kept no copy of it)
          const float *c_scale = PRO == SPRO_BNRELU ? s_scale : pb_dkbuf + 128;
          const float *c_shift = PRO == SPRO_BNRELU ? s_shift : pb_dkbuf + 192;
          // BatchNorm + ReLU applied in place on the landed chunk, every lane on the four
          // 16-byte pieces it requested (row-coalesced for the side store), before any
          // operand read of this wave (same wave: LDS operations stay in program order)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + dma_r0;
            const int q = (lane & 7) ^ ((r >> 1) & 7);
            const int k = c * 32 + q * 4;
            if (k < KA) {
              float4 *pos = reinterpret_cast<float4 *>(const_cast<unsigned char *>(sl) + i * 1024 + lane * 16);
              float4 v = *pos;
              const float4 sc = *reinterpret_cast<const float4 *>(c_scale + k);
              const float4 sh = *reinterpret_cast<const float4 *>(c_shift + k);
              v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y;
              v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
              if (p.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
              }
              *pos = v;
              const long long row = r0 + r;
              if (p.side != nullptr && row < M)
                *reinterpret_cast<float4 *>(p.side + row * p.ld_side + k) = v;
            }
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int ks = 2 * c + s;
        if (ks < KS) {
          float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
          if (c < KC) {
            const int qa = 4 * s + 2 * lk;
            va = *reinterpret_cast<const float4 *>(sl + li * 128 + ((qa ^ swz) << 4));
            vb = *reinterpret_cast<const float4 *>(sl + li * 128 + (((qa + 1) ^ swz) << 4));
          }
          if (PRO == SPRO_GATHER) {
            // the quad holding the centred coordinates: logical k = KA .. KA + 2
            const int kx = KA - ks * 16 - lk * 8;          // offset inside this lane's 8 values
            if (kx == 0 || kx == 4) {
              const float *cx = ctrbuf + par * 6 + (li >> 4);       // rows 0-15 / 16-31
              const float *px = xyzbuf + par * 96 + li;
              float x = px[0] - cx[0], y = px[32] - cx[2], z = px[64] - cx[4];
              if (p.normalize) { x = x / p.radius; y = y / p.radius; z = z / p.radius; }
              const float4 q = make_float4(x, y, z, 0.f);
              if (kx == 0) { va = q; vb = make_float4(0.f, 0.f, 0.f, 0.f); } else { vb = q; }
            } else if (kx < 0) {
              va = make_float4(0.f, 0.f, 0.f, 0.f); vb = va;
            }
          }
          if (PRO == SPRO_POOLBWD && ks * 16 >= KA) {
            // generated columns: dkrow[r][c] = dk[centre][c] where arg[centre][c] is this row
            const int cb = ks * 16 + lk * 8 - KA;
            const int srow = pb_srow0 + ((t & 1) ? pb_sodd : 0);
            const short *ab = pb_argbuf + par * 256 + pb_hoff + cb;
            const float *db = pb_dkbuf + par * 256 + pb_hoff + cb;
            const uint4 aw = *reinterpret_cast<const uint4 *>(ab);              // 8 x int16
            const float4 d0 = *reinterpret_cast<const float4 *>(db), d1 = *reinterpret_cast<const float4 *>(db + 4);
            const unsigned sr = cb < pb_c3v ? (unsigned)srow : 0xffffu;          // 0xffff never matches
            va.x = (aw.x & 0xffffu) == sr ? d0.x : 0.f; va.y = (aw.x >> 16) == sr ? d0.y : 0.f;
            va.z = (aw.y & 0xffffu) == sr ? d0.z : 0.f; va.w = (aw.y >> 16) == sr ? d0.w : 0.f;
            vb.x = (aw.z & 0xffffu) == sr ? d1.x : 0.f; vb.y = (aw.z >> 16) == sr ? d1.y : 0.f;
            vb.z = (aw.w & 0xffffu) == sr ? d1.z : 0.f; vb.w = (aw.w >> 16) == sr ? d1.w : 0.f;
          }
          bf16x8 a[3];
          split8(va, vb, a);
          bf16x8 b[NT][3];
          const unsigned char *wk = wp + ((unsigned)(2 * ks + lk) * NP + li) * 16u;
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              b[j][pl] = *reinterpret_cast<const bf16x8 *>(wk + (unsigned)pl * KB * NP * 16u + j * 512);
          // the 6 plane products with i + j <= 2, small terms first (as rows_gemm_x3_kernel)
          constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
          for (int tt = 0; tt < 6; ++tt)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[tt]], b[j][TB[tt]], acc[j], 0, 0, 0);
        }
      }
      if (c < KC) slot = slot + 1 == SLOTS ? 0 : slot + 1;
    }
    par ^= 1;
    long long sp_t1 = 0;
    if (sp_on) { sp_t1 = SP_NOW(); sp_k += sp_t1 - sp_t0; }

    if constexpr (EPI == EPI_EVAL) {
    if (p.pool_ns > 0) {
      // ---- inference: BN + ReLU + max over the pool_ns rows of a centre ----------------------
      // lane (li, lk) holds rows 4 lk + {0..3, 8..11, 16..19, 24..27} of column li: e < 8 are
      // rows 0..15, e >= 8 rows 16..31
      const int ns = p.pool_ns;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[j][e] * esc[j] + esh[j];
          if (p.ep_relu) v = fmaxf(v, 0.f);
          const int h = ns == 16 ? (e >> 3) : 0;
          if (h == 0) gmax[j][0] = fmaxf(gmax[j][0], v); else gmax[j][1] = fmaxf(gmax[j][1], v);
        }
      }
      const bool flush = ns != 64 || (n & 1);
      if (flush) {
        const long long centres = M / ns;
        const long long c0 = ns == 64 ? (t >> 1) : (ns == 32 ? t : 2 * t);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = 32 * j + li;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (h == 1 && ns != 16) continue;
            const float m = fmaxf(gmax[j][h], __shfl_xor(gmax[j][h], 32, 64));
            if (lk == 0 && col < N && c0 + h < centres) p.Y[(c0 + h) * p.ldy + col] = m;
            gmax[j][h] = -INFINITY;
          }
        }
      }
      continue;
    }
    }
    if constexpr (EPI == EPI_EVAL) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[j][e] * esc[j] + esh[j];
          if (p.ep_relu) v = fmaxf(v, 0.f);
          acc[j][e] = v;
        }
    }
    if (PRO == SPRO_POOLBWD) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] += pbias[j];
    }
    if constexpr (EPI == EPI_RAW) {
      // ---- training, pooled layer: running maximum of sign * y (+ first index) per centre --
      const int ns = p.pool_ns;
      const int tb = (ns == 64 && (n & 1)) ? 32 : 0;              // wave-uniform row base of the tile
      const bool full = r0 + 32 <= M;                             // (M % ns == 0: only a last tile is ragged)
      // ascending rows of this lane, strict compares: the first extremum is kept.  The index is
      // tracked without the lane's own 4 lk (a wave-uniform operand: no add per element)
      if (ns != 16) {
        if (full) {
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const float v = acc[j][e];
              if (v > rmax[j][0]) { rmax[j][0] = v; ramax[j][0] = tb + (e & 3) + 8 * (e >> 2); }
            }
        } else {
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int ro = (e & 3) + 8 * (e >> 2);
              const float v = acc[j][e];
              if (r0 + ro + 4 * lk < M && v > rmax[j][0]) { rmax[j][0] = v; ramax[j][0] = tb + ro; }
            }
        }
      } else {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {          // rows 0-15 (e < 8) / 16-31: two centres
            const int ro = (e & 3) + 8 * (e >> 2);
            const float v = acc[j][e];
            if ((full || r0 + ro + 4 * lk < M) && v > rmax[j][e >> 3]) {
              rmax[j][e >> 3] = v; ramax[j][e >> 3] = ro & 15;
            }
          }
      }
      if (ns != 64 || (n & 1)) {
        const long long centres = M / ns;
        const long long c0 = ns == 64 ? (t >> 1) : (ns == 32 ? t : 2 * t);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = 32 * j + li;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (h == 1 && ns != 16) continue;
            float v1 = rmax[j][h];
            int a1 = ramax[j][h] + 4 * lk;
            const float o1 = __shfl_xor(v1, 32, 64);
            const int b1 = __shfl_xor(a1, 32, 64);
            if (o1 > v1 || (o1 == v1 && b1 < a1)) { v1 = o1; a1 = b1; }
            if (lk == 0 && col < N && c0 + h < centres) {
              const long long o = (c0 + h) * N + col;
              p.ext[o] = v1 * esgn[j]; p.aext[o] = a1;
            }
            rmax[j][h] = -INFINITY; ramax[j][h] = 0;
          }
        }
      }
    }
    // ---- epilogue: statistics, 4x4 DPP transposes, dwordx4 stores -------------------------
    // C/D layout of 32x32: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = 32 * j + li;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if constexpr (EPI == EPI_NEXT) {
          float a4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = acc[j][4 * g + i];
            a4[i] = v;
            const float y = ny[j][4 * g + i];
            float dz = v;
            if (p.nrelu && !(y * nsc[j] + nsh[j] > 0.f)) dz = 0.f;
            if (r0 + 8 * g + 4 * lk + i < M) { s1[j] += dz; s2[j] += dz * ((y - nmu[j]) * nis[j]); }
          }
          quad_transpose(a4, lane);
          const long long row = r0 + 8 * g + 4 * lk + (lane & 3);
          const int c0 = 32 * j + (li & ~3);
          if (row < M) {
            float *dst = p.Y + row * p.ldy + c0;
            if (c0 + 3 < N) {
              *reinterpret_cast<float4 *>(dst) = make_float4(a4[0], a4[1], a4[2], a4[3]);
            } else {
              if (c0 < N) dst[0] = a4[0];
              if (c0 + 1 < N) dst[1] = a4[1];
              if (c0 + 2 < N) dst[2] = a4[2];
            }
          }
          continue;
        }
        // (columns >= N meet zero weight planes: they add exact zeros, no predicate; rows are
        // checked in a ragged last tile only -- per element the 64-bit compare and the select
        // tripled the vector-ALU work of this loop)
        float a4[4];
        if (r0 + 32 <= M) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = acc[j][4 * g + i];
            a4[i] = v;
            s1[j] += v; s2[j] += v * v;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = acc[j][4 * g + i];
            a4[i] = v;
            if (r0 + 8 * g + 4 * lk + i < M) { s1[j] += v; s2[j] += v * v; }
          }
        }
        if (p.Y == nullptr) continue;             // statistics only (pooled training layer)
        if constexpr (EPI == EPI_RAW) {           // the products carry the folded sign
#pragma unroll
          for (int i = 0; i < 4; ++i) a4[i] *= esgn[j];
        }
        quad_transpose(a4, lane);
        const long long row = r0 + 8 * g + 4 * lk + (lane & 3);
        const int c0 = 32 * j + (li & ~3);
        if (row < M) {
          float *dst = p.Y + row * p.ldy + c0;
          if (c0 + 3 < N) {
            *reinterpret_cast<float4 *>(dst) = make_float4(a4[0], a4[1], a4[2], a4[3]);
          } else {
            if (c0 < N) dst[0] = a4[0];
            if (c0 + 1 < N) dst[1] = a4[1];
            if (c0 + 2 < N) dst[2] = a4[2];
          }
        }
      }
    }
    if (sp_on) { sp_epi += SP_NOW() - sp_t1; ++sp_tiles; }
  }

  if (sp_on && lane == 0) {
    sprof[0] = sp_wait; sprof[1] = sp_k - sp_wait; sprof[2] = sp_epi; sprof[3] = sp_tiles;
    sprof[4] = SP_NOW() - sp_life0;
    sprof[5] = sp_life0 - sp_entry;              // prologue: W staging, ring set-up, first issues
  }
  if (p.partial != nullptr) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float t1 = (s1[j] + __shfl_xor(s1[j], 32, 64)) * (EPI == EPI_RAW ? esgn[EPI == EPI_RAW ? j : 0] : 1.f);
      const float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
      const int col = 32 * j + li;
      if (lk == 0 && col < N && wid < p.partial_rows) {
        p.partial[wid * 2 * N + col] = t1;
        p.partial[wid * 2 * N + N + col] = t2;
      }
    }
    // rows of the partial table this grid does not own stay zero
    for (long long r = wid + nw; r < p.partial_rows; r += nw)
      for (int c = lane; c < 2 * N; c += 64) p.partial[r * 2 * N + c] = 0.f;
  }
}

size_t stream_lds_bytes(int NT, int K, int waves, int slots, int pro) {
  const int KS = (K + 15) / 16;
  return (size_t)3 * 2 * KS * 32 * NT * 16 + (pro == SPRO_BNRELU ? (size_t)2 * 2 * KS * 8 * 4 : 0) +
         (size_t)waves * (slots * CHUNK_BYTES +
                          (pro == SPRO_GATHER ? AUX_BYTES : (pro == SPRO_POOLBWD ? PB_AUX_BYTES : 0)));
}

int g_stream_on = -1, g_stream_grid = 0;
bool stream_on() {
  if (g_stream_on < 0) {
    const char *e = getenv("S2C_GEMM_STREAM");
    g_stream_on = e ? atoi(e) : 1;
    const char *g = getenv("S2C_GEMM_STREAM_GRID");
    g_stream_grid = g ? atoi(g) : 240;
    if (g_stream_grid <= 0) g_stream_grid = 240;
  }
  return g_stream_on != 0;
}

constexpr long long STREAM_MIN_ROWS = 131072;
constexpr size_t LDS_LIMIT = 160 * 1024;

// Workgroup shapes in order of preference: more waves (latency hiding on the compute side),
// then more ring slots (bytes in flight).  LDS is what decides.
struct StreamCfg { int waves, slots; };
constexpr StreamCfg CFGS[5] = {{8, 3}, {8, 2}, {4, 4}, {4, 3}, {4, 2}};   // (8,4) measured slower than (8,3)
// (8,2) only for the pool-backward mode: its generated columns are pure MFMA + VALU work, which
// one wave per SIMD cannot overlap with anything (4 x 4: 337 us)

int pick_cfg(long long M, int N, int K, int pro) {
  if (M < STREAM_MIN_ROWS || N <= 0 || N > 128 || K <= 0) return -1;
  const int NT = N <= 64 ? 2 : 4;
  for (int c = 0; c < 5; ++c) {
    if (c == 1 && pro != SPRO_POOLBWD) continue;
    if (stream_lds_bytes(NT, K, CFGS[c].waves, CFGS[c].slots, pro) <= LDS_LIMIT) return c;
  }
  return -1;
}

template <int NT, int PRO, int WAVES, int SLOTS, int EPI>
int launch_one(const StreamArgs &a, int blocks, size_t lds, hipStream_t st) {
  // opt-in dynamic LDS size: set (and checked) once per DEVICE and kernel instance; a device
  // that refuses it makes the shape "not taken" (-2): the dispatch falls back to the tiled kernel
  static int attr_state[64];                   // 0 unknown, 1 ok, -1 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (attr_state[dev] == 0)
    attr_state[dev] =
        hipFuncSetAttribute((const void *)rows_stream_gemm_kernel<NT, PRO, WAVES, SLOTS, EPI>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT) == hipSuccess
            ? 1 : -1;
  if (attr_state[dev] < 0) {
    (void)hipGetLastError();
    return -2;
  }
  hipLaunchKernelGGL((rows_stream_gemm_kernel<NT, PRO, WAVES, SLOTS, EPI>), dim3(blocks), dim3(64 * WAVES),
                     lds, st, a);
  return 0;
}

template <int NT, int PRO, int EPI>
int launch_cfg(const StreamArgs &a, int cfg, int blocks, size_t lds, hipStream_t st) {
  switch (cfg) {
    case 0: return launch_one<NT, PRO, 8, 3, EPI>(a, blocks, lds, st);
    case 1:
      if constexpr (PRO == SPRO_POOLBWD) return launch_one<NT, PRO, 8, 2, EPI>(a, blocks, lds, st);
      else return -1;
    case 2: return launch_one<NT, PRO, 4, 4, EPI>(a, blocks, lds, st);
    case 3: return launch_one<NT, PRO, 4, 3, EPI>(a, blocks, lds, st);
    default: return launch_one<NT, PRO, 4, 2, EPI>(a, blocks, lds, st);
  }
}

// the (prologue, epilogue) pairs that exist: plain / BN+ReLU prologue x {train, raw}; plain /
// gather x eval; gather x train; pool-backward x train
template <int NT, int PRO>
int launch_epi(const StreamArgs &a, int cfg, int blocks, size_t lds, hipStream_t st) {
  const int epi = a.ext != nullptr ? EPI_RAW : (a.ep_mean != nullptr ? EPI_EVAL : (a.nY != nullptr ? EPI_NEXT : EPI_TRAIN));
  if (epi == EPI_TRAIN) return launch_cfg<NT, PRO, EPI_TRAIN>(a, cfg, blocks, lds, st);
  if constexpr (PRO == SPRO_POOLBWD)
    if (epi == EPI_NEXT) return launch_cfg<NT, PRO, EPI_NEXT>(a, cfg, blocks, lds, st);
  if constexpr (PRO == SPRO_NONE || PRO == SPRO_BNRELU)
    if (epi == EPI_RAW) return launch_cfg<NT, PRO, EPI_RAW>(a, cfg, blocks, lds, st);
  if constexpr (PRO == SPRO_NONE || PRO == SPRO_GATHER)
    if (epi == EPI_EVAL) return launch_cfg<NT, PRO, EPI_EVAL>(a, cfg, blocks, lds, st);
  return -1;
}

template <int PRO>
int launch_stream(const StreamArgs &a, hipStream_t st) {
  const int NT = a.N <= 64 ? 2 : 4;
  const int cfg = pick_cfg(a.M, a.N, a.K, PRO);
  if (cfg < 0) return -2;
  const int waves = CFGS[cfg].waves;
  const size_t lds = stream_lds_bytes(NT, a.K, waves, CFGS[cfg].slots, PRO);
  int blocks = g_stream_grid;
  if (a.partial != nullptr && blocks * waves > a.partial_rows) blocks = a.partial_rows / waves;
  if (blocks < 1) return -2;
  const int rc = NT == 2 ? launch_epi<2, PRO>(a, cfg, blocks, lds, st)
                         : launch_epi<4, PRO>(a, cfg, blocks, lds, st);
  if (rc != 0) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_rows_stream_gemm launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // namespace

extern "C" int s2c_rows_gemm_c64_bn_relu_side(long long M, int N, int K, const float *A, int lda,
                                              const float *scale, const float *shift,
                                              float *side, int ld_side, const float *W, int ldw,
                                              float *Y, int ldy, float *partial, void *stream);
extern "C" int s2c_rows_gemm_c64_supported(long long M, int N, int K);
extern "C" int s2c_rows_gemm_c64_pool_ext(long long M, int N, int K, const float *A, int lda,
                                          const float *scale, const float *shift, float *side,
                                          int ld_side, const float *W, int ldw, int pool_ns,
                                          const float *gamma, float *ext, int *aext, float *Y,
                                          int ldy, float *partial, void *stream);

// 1 when s2c_rows_gemm / s2c_sa_gather_gemm hand this shape to the streaming kernel
// (plain operand: K, lda multiples of 4 and 16-byte aligned A; gather: C a multiple of 4,
// C >= 100, ns in {16, 32, 64}).
extern "C" int s2c_rows_stream_supported(long long M, int N, int K, int gather) {
  if (!stream_on() || pick_cfg(M, N, K, gather ? SPRO_GATHER : SPRO_BNRELU) < 0) return 0;
  if (gather) return (K - 3) % 4 == 0 && K - 3 >= 100;
  return K % 4 == 0;
}

// Diagnostics: see g_sprof.  prof == NULL switches it off.
extern "C" int s2c_gemm_stream_set_profile(long long *prof, int block) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_sprof), &prof, sizeof(prof)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_sprof_block), &block, sizeof(block)) != hipSuccess) return -1;
  return 0;
}

// Workgroups of the persistent grid (default 240 of the 256 CUs; S2C_GEMM_STREAM_GRID).  A
// persistent workgroup that finds no free CU at launch starts when another one has FINISHED,
// i.e. doubles the kernel's time -- so the grid must leave out the CUs held by kernels running
// beside it (the geometry stage on the side stream: one FPS workgroup per scene for
// milliseconds).  scan2cap_amd.pipeline.GeometrySlots sets 256 - scenes per pass - 8.
// Returns the previous value.
extern "C" int s2c_gemm_set_stream_grid(int workgroups) {
  stream_on();
  const int old = g_stream_grid;
  if (workgroups > 0) g_stream_grid = workgroups > 1024 ? 1024 : workgroups;
  return old;
}

// Inference variants (BatchNorm + ReLU (+ max-pool) in the epilogue): the contracts of
// s2c_rows_gemm_bn_eval / s2c_sa_gather_gemm_bn_eval.  -2: shape not taken.
extern "C" int s2c_rows_stream_gemm_bn_eval(long long M, int N, int K, const float *A, int lda,
                                            const float *W, int ldw, const float *gamma,
                                            const float *beta, const float *mean, const float *var,
                                            float eps, int relu, int pool_ns, float *out, int ldo,
                                            void *stream) {
  if (!s2c_rows_stream_supported(M, N, K, 0) || (lda & 3) || ((uintptr_t)A & 15) ||
      (ldo & 3) || ((uintptr_t)out & 15))
    return -2;
  StreamArgs a = {};
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.Y = out; a.ldy = ldo;
  a.ep_gamma = gamma; a.ep_beta = beta; a.ep_mean = mean; a.ep_var = var; a.ep_eps = eps;
  a.ep_relu = relu; a.pool_ns = pool_ns;
  return launch_stream<SPRO_NONE>(a, (hipStream_t)stream);
}

extern "C" int s2c_sa_gather_stream_gemm_bn_eval(int b, int n, int m, int ns, int C,
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
  if (!s2c_rows_stream_supported(M, N, K, 1) || !(ns == 16 || ns == 32 || ns == 64) ||
      (ldo & 3) || ((uintptr_t)out & 15) || ((uintptr_t)feats & 3))
    return -2;
  StreamArgs a = {};
  a.M = M; a.N = N; a.K = K; a.W = W; a.ldw = ldw; a.Y = out; a.ldy = ldo;
  a.xyz = xyz; a.new_xyz = new_xyz; a.feats = feats; a.idx = idx;
  a.frs = feat_row_stride; a.fbs = feat_batch_stride;
  a.n = n; a.m = m; a.ns = ns; a.C = C; a.radius = radius; a.normalize = normalize;
  a.ep_gamma = gamma; a.ep_beta = beta; a.ep_mean = mean; a.ep_var = var; a.ep_eps = eps;
  a.ep_relu = relu; a.pool_ns = pool_ns;
  return launch_stream<SPRO_GATHER>(a, (hipStream_t)stream);
}

// 1: tall eligible shapes run on the streaming kernel (default), 0: everything on the tiled
// kernel.  Returns the previous setting.  (Also: environment S2C_GEMM_STREAM at first use.)
extern "C" int s2c_gemm_set_stream(int on) {
  const int old = stream_on() ? 1 : 0;
  g_stream_on = on ? 1 : 0;
  return old;
}

// Internal entry points used by s2c_gemm.hip's dispatch (same contracts as s2c_rows_gemm /
// s2c_sa_gather_gemm; `partial_rows` = s2c_rows_gemm_blocks(M, N)).  Return -2: shape not taken.
extern "C" int s2c_rows_stream_gemm(long long M, int N, int K, const float *A, int lda,
                                    const float *W, int ldw, float *Y, int ldy, float *partial,
                                    int partial_rows, void *stream) {
  if (!s2c_rows_stream_supported(M, N, K, 0) || (lda & 3) || ((uintptr_t)A & 15) ||
      (ldy & 3) || ((uintptr_t)Y & 15))
    return -2;
  StreamArgs a = {};
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.Y = Y; a.ldy = ldy;
  a.partial = partial; a.partial_rows = partial_rows;
  return launch_stream<SPRO_NONE>(a, (hipStream_t)stream);
}

// The per-POINT product of a set-abstraction stage's first layer, P (M x N) = X (M x K) W^T, on the
// streaming kernel (round 6).  X = the feature columns of the (B,N,3+C) cloud read IN PLACE: rows of
// 3 + C floats, so neither lda nor the row starts are multiples of 16 bytes -- the ring is filled by
// `global_load_lds_dwordx4`, which takes dword-aligned sources (as the gather prologue's feature rows
// always were).  K % 4 == 0, M >= the streaming threshold; -2: shape not taken (csrc/s2c_pgemm.hip).
extern "C" int s2c_point_gemm_stream(long long M, int N, int K, const float *A, long long lda,
                                     const float *W, int ldw, float *P, int ldp, void *stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !P || lda < K || ldw < K || ldp < N) return -1;
  if (!s2c_rows_stream_supported(M, N, K, 0) || ((uintptr_t)A & 3) || lda >= (1ll << 31) ||
      (ldp & 3) || ((uintptr_t)P & 15))
    return -2;
  StreamArgs a = {};
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = (int)lda; a.W = W; a.ldw = ldw; a.Y = P; a.ldy = ldp;
  return launch_stream<SPRO_NONE>(a, (hipStream_t)stream);
}

// Y = relu?(A * scale + shift) W^T with the activated operand written to `side` (M x K, may be
// NULL): one pass over the pre-activation tensor A instead of s2c_bn_relu + s2c_rows_gemm.
// Streaming kernel only: returns -2 when the shape is not taken (call the two separately).
extern "C" int s2c_rows_gemm_bn_relu_side(long long M, int N, int K, const float *A, int lda,
                                          const float *scale, const float *shift, int relu,
                                          float *side, int ld_side, const float *W, int ldw,
                                          float *Y, int ldy, float *partial, void *stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !Y || !scale || !shift || lda < K || ldw < K)
    return -1;
  if (!s2c_rows_stream_supported(M, N, K, 0) || (lda & 3) || ((uintptr_t)A & 15) ||
      (ldy & 3) || ((uintptr_t)Y & 15) || (side && ((ld_side & 3) || ((uintptr_t)side & 15)))) {
    // mid-size / wide layers: the 64-k-chunk kernel of s2c_gemm.hip (ReLU prologue only)
    if (relu && ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0)
      return s2c_rows_gemm_c64_bn_relu_side(M, N, K, A, lda, scale, shift, side, ld_side, W, ldw,
                                            Y, ldy, partial, stream);
    return -2;
  }
  StreamArgs a = {};
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.Y = Y; a.ldy = ldy;
  a.partial = partial; a.partial_rows = s2c_rows_gemm_blocks(M, N);
  a.scale = scale; a.shift = shift; a.relu = relu; a.side = side; a.ld_side = ld_side;
  return launch_stream<SPRO_BNRELU>(a, (hipStream_t)stream);
}

// 1 when s2c_rows_gemm_bn_relu_side takes (M, N, K) with a ReLU prologue on one of its kernels
extern "C" int s2c_rows_gemm_side_supported(long long M, int N, int K) {
  return s2c_rows_stream_supported(M, N, K, 0) || s2c_rows_gemm_c64_supported(M, N, K);
}

// The pooled LAST layer of a training stack: products as s2c_rows_gemm_bn_relu_side (scale ==
// NULL: plain operand, no side output), statistics partials as s2c_rows_gemm, and instead of
// (or, Y != NULL, besides) Y the per-centre extremum the pooled BatchNorm + ReLU will select
// (StreamArgs::ext; `gamma` = the BatchNorm weight of THIS layer, NULL = all positive):
// s2c_pool_select reads J x N values, not M x N.  -2: shape not taken by the streaming kernel.
extern "C" int s2c_rows_gemm_pool_raw(long long M, int N, int K, const float *A, int lda,
                                      const float *scale, const float *shift, int relu,
                                      float *side, int ld_side, const float *W, int ldw,
                                      int pool_ns, const float *gamma, float *ext, int *aext,
                                      float *Y, int ldy, float *partial, void *stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !ext || !aext ||
      lda < K || ldw < K || !(pool_ns == 16 || pool_ns == 32 || pool_ns == 64) || M % pool_ns)
    return -1;
  if (!s2c_rows_stream_supported(M, N, K, 0) || (lda & 3) || ((uintptr_t)A & 15) ||
      (Y && ((ldy & 3) || ((uintptr_t)Y & 15))) ||
      (side && ((ld_side & 3) || ((uintptr_t)side & 15)))) {
    // wide layers with Y materialised: the 64-k-chunk kernel of s2c_gemm.hip
    if (Y != nullptr && (scale == nullptr || relu))
      return s2c_rows_gemm_c64_pool_ext(M, N, K, A, lda, scale, shift, side, ld_side, W, ldw,
                                        pool_ns, gamma, ext, aext, Y, ldy, partial, stream);
    return -2;
  }
  StreamArgs a = {};
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.Y = Y; a.ldy = ldy;
  a.partial = partial; a.partial_rows = s2c_rows_gemm_blocks(M, N);
  a.scale = scale; a.shift = shift; a.relu = relu; a.side = side; a.ld_side = ld_side;
  a.pool_ns = pool_ns; a.ext = ext; a.aext = aext; a.ext_sign = gamma;
  if (scale != nullptr) return launch_stream<SPRO_BNRELU>(a, (hipStream_t)stream);
  return launch_stream<SPRO_NONE>(a, (hipStream_t)stream);
}

// 1 when s2c_pool_bwd_input_grad takes (M, N, KA, C3) (the exact test of its dispatch)
extern "C" int s2c_pool_bwd_supported(long long M, int N, int KA, int C3) {
  return stream_on() && KA > 0 && C3 > 0 && !(KA & 15) && !(C3 & 7) && C3 <= 128 &&
         pick_cfg(M, N, KA + C3, SPRO_POOLBWD) >= 0;
}

// pscale / pshift (both or neither): the operand's first KA columns are relu?(A pscale + pshift), the
// layer's input recomputed from the previous layer's pre-activation A.  The constants live in the half of
// a wave's dk buffer that only ns = 16 uses, 64 floats each.
static bool pb_act_ok(const float *pscale, const float *pshift, int KA, int ns) {
  if (!pscale && !pshift) return true;
  return pscale && pshift && KA <= 64 && ns >= 32;
}

// dA (M x N) = [A | dkrow] [-G^T | W3^T]^T + cvec  (see StreamArgs::pb_arg): the input gradient
// of a max-pooled BatchNorm(+ReLU) layer from the layer's INPUT activation A (M x KA), the
// routed pooled gradient dk / arg (J x C3) and Wcat (N x (KA + C3)), bias cvec (N).  KA % 16 == 0,
// C3 % 8 == 0, C3 <= 128, ns in {16, 32, 64}.  -2: shape not taken.
extern "C" int s2c_pool_bwd_input_grad(long long M, int N, int KA, int C3, int ns, const float *A,
                                       int lda, const short *arg, const float *dk,
                                       const float *Wcat, int ldw, const float *cvec, float *dA,
                                       int ldd, const float *pscale, const float *pshift, int prelu,
                                       void *stream) {
  if (M <= 0 || N <= 0 || KA <= 0 || C3 <= 0 || !A || !arg || !dk || !Wcat || !dA ||
      lda < KA || ldw < KA + C3 || !(ns == 16 || ns == 32 || ns == 64) || M % ns)
    return -1;
  if (!stream_on() || (KA & 15) || (C3 & 7) || C3 > 128 || (lda & 3) || ((uintptr_t)A & 15) ||
      (ldd & 3) || ((uintptr_t)dA & 15) || ((uintptr_t)arg & 15) || ((uintptr_t)dk & 15) ||
      pick_cfg(M, N, KA + C3, SPRO_POOLBWD) < 0 || !pb_act_ok(pscale, pshift, KA, ns))
    return -2;
  StreamArgs a = {};
  a.M = M; a.N = N; a.K = KA + C3; a.A = A; a.lda = lda; a.W = Wcat; a.ldw = ldw; a.Y = dA; a.ldy = ldd;
  a.pb_arg = arg; a.pb_dk = dk; a.bias = cvec; a.pb_ka = KA; a.pb_c3 = C3; a.pb_ns = ns;
  a.scale = pscale; a.shift = pshift; a.relu = prelu;
  return launch_stream<SPRO_POOLBWD>(a, (hipStream_t)stream);
}

// ... with the column sums of the PREVIOUS layer's BatchNorm backward (its upstream gradient is
// this dA) out of the epilogue: nY (M x N contiguous) etc. = that layer; npartial =
// s2c_rows_gemm_blocks(M, N) rows of [s1 | s2] for s2c_bn_bwd_finalize_partials.  ldd == N.
extern "C" int s2c_pool_bwd_input_grad_next_stats(long long M, int N, int KA, int C3, int ns,
                                                  const float *A, int lda, const short *arg,
                                                  const float *dk, const float *Wcat, int ldw,
                                                  const float *cvec, float *dA, int ldd,
                                                  const float *nY, const float *nscale,
                                                  const float *nshift, const float *nmean,
                                                  const float *ninvstd, int nrelu,
                                                  float *npartial, const float *pscale,
                                                  const float *pshift, int prelu, void *stream) {
  if (M <= 0 || N <= 0 || KA <= 0 || C3 <= 0 || !A || !arg || !dk || !Wcat || !dA ||
      lda < KA || ldw < KA + C3 || !(ns == 16 || ns == 32 || ns == 64) || M % ns || ldd != N ||
      !nY || !nscale || !nshift || !nmean || !ninvstd || !npartial)
    return -1;
  if (!stream_on() || (KA & 15) || (C3 & 7) || C3 > 128 || (lda & 3) || ((uintptr_t)A & 15) ||
      (ldd & 3) || ((uintptr_t)dA & 15) || ((uintptr_t)arg & 15) || ((uintptr_t)dk & 15) ||
      pick_cfg(M, N, KA + C3, SPRO_POOLBWD) < 0 || !pb_act_ok(pscale, pshift, KA, ns))
    return -2;
  StreamArgs a = {};
  a.M = M; a.N = N; a.K = KA + C3; a.A = A; a.lda = lda; a.W = Wcat; a.ldw = ldw; a.Y = dA; a.ldy = ldd;
  a.pb_arg = arg; a.pb_dk = dk; a.bias = cvec; a.pb_ka = KA; a.pb_c3 = C3; a.pb_ns = ns;
  a.nY = nY; a.nscale = nscale; a.nshift = nshift; a.nmean = nmean; a.ninvstd = ninvstd;
  a.nrelu = nrelu; a.partial = npartial; a.partial_rows = s2c_rows_gemm_blocks(M, N);
  a.scale = pscale; a.shift = pshift; a.relu = prelu;
  return launch_stream<SPRO_POOLBWD>(a, (hipStream_t)stream);
}

extern "C" int s2c_sa_gather_stream_gemm(int b, int n, int m, int ns, int C,
                                         long long feat_row_stride, long long feat_batch_stride,
                                         float radius, int normalize, const float *xyz,
                                         const float *new_xyz, const float *feats, const int *idx,
                                         int N, const float *W, int ldw, float *Y, int ldy,
                                         float *partial, int partial_rows, void *stream) {
  const long long M = (long long)b * m * ns;
  const int K = 3 + C;
  if (!s2c_rows_stream_supported(M, N, K, 1) || !(ns == 16 || ns == 32 || ns == 64) ||
      (ldy & 3) || ((uintptr_t)Y & 15) || ((uintptr_t)feats & 3))
    return -2;
  StreamArgs a = {};
  a.M = M; a.N = N; a.K = K; a.W = W; a.ldw = ldw; a.Y = Y; a.ldy = ldy;
  a.partial = partial; a.partial_rows = partial_rows;
  a.xyz = xyz; a.new_xyz = new_xyz; a.feats = feats; a.idx = idx;
  a.frs = feat_row_stride; a.fbs = feat_batch_stride;
  a.n = n; a.m = m; a.ns = ns; a.C = C; a.radius = radius; a.normalize = normalize;
  return launch_stream<SPRO_GATHER>(a, (hipStream_t)stream);
}
