// s2c_dwstream.hip -- tall weight gradients  dW[C x N] = dY^T A  (dY: M x C, A: M x N, M ~ 1e5..1e6)
// as a STREAMING kernel (round 5).  (reference: autograd of the 1x1 Conv2d layers of
// lib/pointnet2/pytorch_utils.py:67-120 inside pointnet2_modules.py:251-257.)
//
// The reduction index is the row index of both operands, the output is tiny: the kernel is a
// pure read of (C + N) * 4 bytes per row.  The split-K library product runs on the fp32 matrix
// instruction (MI16x16x1) and is bound by it at the wide shapes ((262144,256,128): 141 us =
// 122 TF of 157); the register-fed kernel of s2c_dw.hip is bound by its 4-byte loads.  Here:
//
//   * the output is cut into 64 x 64 tiles; a WAVE owns one tile and one k-group (every kg-th
//     16-row chunk of the workgroup's share) and streams ITS 64 columns of dY and of A through a
//     private ring of LDS slots filled by LDS-DMA (`global_load_lds_dwordx4`, SGPR base per chunk +
//     a per-lane 32-bit offset that never changes): no workgroup barrier in the loop, the waves of
//     a SIMD drift out of phase by themselves; column blocks that several tiles share come out of
//     L2 / the vector L1 again (a shared ring with one barrier per chunk was built first: the
//     barrier puts the two waves of a SIMD in phase and both pipes serialise -- 1.1-1.4x slower);
//   * the transposed operand costs nothing: a lane's 8 consecutive-k values of
//     v_mfma_f32_32x32x16_bf16 are 8 ROWS of one column -- `ds_read2_b32` down the columns of the
//     row-major chunk (lanes on consecutive columns);
//   * fp32 -> 3 x bf16 split in registers by truncation (full-rate bit masks), the 6 plane
//     products with i + j <= 2 (fp32-accurate); the products LAG one k-step behind the splits:
//     program order is one MFMA, one pair of values split, ... pinned with sched_barrier, so the
//     matrix pipe and the VALU of a SIMD overlap inside one wave;
//   * the loop is free of run-time bookkeeping: ring depth, DMA count and the `s_waitcnt vmcnt`
//     immediate are template constants (a generic version with run-time dispatch spent 1.2 us per
//     chunk in scalar code with everything else switched off);
//   * k-groups meet through LDS once at the end; one partial (C x N) tile per workgroup, added by
//     the caller's `s2c_multi_colsum` launch (kernel-boundary reduction: deterministic);
//   * N need not be a multiple of 4 and A's rows need only dword alignment (the (B,N,3+C) cloud's
//     own rows as the operand: [xyz | features] in ONE product); the chunk a DMA piece could
//     overrun (the tensor's last row, a ragged end) is filled by guarded loads;
//   * A == dY (the Gram matrix of the pooled-layer algebra): the chunk is loaded once.
// Measured on MI355X (rocprofv3, us, this kernel | library split-K bmm): (1M,64,64) 94 | 93,
// Gram (1M,64) 71 | 80, (262144,256,128) 133 | 142, (262144,128,128) 68 | 71, (320000,64,132) 74 | 88,
// the cloud rows (320000,64,135) 83 | 88 + 29 for the separate 3-column product, (320000,64,3)
// 27 | 29, (65536,256,128) 48 | 39.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr size_t LDS_BUDGET = 160 * 1024 - 2048;

struct DwsArgs {
  long long M;
  int C, N;
  const float *dY; long long ldy;
  const float *A; long long lda;
  float *part;             // [grid][C][N]
  int tcw, tnw, kg;        // waves = tcw * tnw tiles of 64 x 64, kg k-groups each
  int S, ni;               // ring slots per wave, DMA instructions per chunk
  int npad;                // N rounded up to whole 16-byte pieces
  int same;                // A is dY: one region
  int slot_bytes;
  long long nd;            // chunks 0 .. nd-1 are filled by DMA, nd .. nchunks-1 by guarded loads
  long long nchunks;       // 16-row chunks
  // PRO instances: the operand is relu?(A pscale[n] + pshift[n]) -- a layer's input activation recomputed
  // from the previous layer's pre-activation A (the arithmetic of the forward GEMM's prologue), so that
  // the forward need not write the activation out for this product alone
  const float *pscale, *pshift; int prelu;
};

struct Planes { bf16x8 p[3]; };

// fp32 -> hi + mid + lo bf16 planes by TRUNCATION (bit masks, full-rate VALU): hi = x & 0xffff0000,
// mid = (x - hi) & 0xffff0000, lo = the upper half of x - hi - mid.  Both residuals are exact in
// fp32, the three planes hold 24 leading bits of x: |x - (hi + mid + lo)| <= 2^-24 |x|, as accurate
// as the round-to-nearest split of s2c_gemm.hip for the six products with i + j <= 2.  Why not
// v_cvt_pk_bf16_f32 here: BOTH operands of every product are split in this kernel (6 MFMAs per
// split instead of 12+ in the forward GEMMs) and the conversions made it VALU-bound.
// one pair of consecutive-k values -> dword q of the three planes
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
  const unsigned b0 = __builtin_bit_cast(unsigned, x0), b1 = __builtin_bit_cast(unsigned, x1);
  const float r0 = x0 - __builtin_bit_cast(float, b0 & 0xffff0000u);
  const float r1 = x1 - __builtin_bit_cast(float, b1 & 0xffff0000u);
  const unsigned c0 = __builtin_bit_cast(unsigned, r0), c1 = __builtin_bit_cast(unsigned, r1);
  const float s0 = r0 - __builtin_bit_cast(float, c0 & 0xffff0000u);
  const float s1 = r1 - __builtin_bit_cast(float, c1 & 0xffff0000u);
  h = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
  m = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
  l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
}

// Single-tile outputs with C = 64 (SA1's second layer, the Gram matrix of the pooled-layer
// algebra, the coordinate columns): every WAVE is its own pipeline -- a private ring of 16-row
// chunks (one k-step), no workgroup barrier in the loop, the waves of a SIMD drift out of phase by
// themselves.  Chunk c goes to wave c % (grid * 8).  Everything the loop needs is a compile-time
// constant (NIA: DMA instructions of the A region -- 4: up to 64 columns at LDS stride 64, 1: up
// to 16 columns at stride 16, 0: A is dY; S ring slots): the generic first version spent 1.2 us
// per chunk in scalar bookkeeping (run-time vmcnt dispatch, per-piece pointer selects, spilled
// SGPRs) with nothing else switched on.  DMA addresses are an SGPR base per chunk + a per-lane
// 32-bit offset that never changes.
__device__ __forceinline__ void glds16s(const void *sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int NIA, int S, bool PRO = false>
__global__ __launch_bounds__(512, 1) void dw_private_kernel(DwsArgs p) {
  constexpr int C = 64, NIY = 4, NI = NIY + NIA;
  constexpr int NP = NIA == 1 ? 16 : 64;           // LDS row stride of the A region (floats)
  constexpr int SLOT = NI * 1024 + (NIA == 1 ? 1024 : 0);   // + slack for the 64-column reads
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  // wave -> (output tile, k-group): wider outputs are tiles of 64 x 64, every tile-wave streams ITS
  // 64 columns of dY and of A through its own ring (the column blocks other tiles share with it come
  // out of L2 / the vector L1 again: no workgroup barrier, no shared slot to hand over)
  const int ntile = p.tcw * p.tnw;
  const int tile = wave % ntile, kgid = wave / ntile;
  const int tc = tile / p.tnw, tn = tile % p.tnw;
  const int N = p.N;
  const int ncv = N - tn * 64 < 64 ? N - tn * 64 : 64;          // this tile's columns of A
  const int ncvp = p.npad - tn * 64 < 64 ? p.npad - tn * 64 : 64;
  const long long M = p.M;
  const long long GW = (long long)gridDim.x * p.kg, gw = (long long)blockIdx.x * p.kg + kgid;
  const float *dYt = p.dY + tc * 64, *At = p.A + tn * 64;

  // this lane's pieces: dY region 16 rows x 16 pieces (4 instructions), A region 16 rows x NP/4
  unsigned offy[NIY], offa[NIA > 0 ? NIA : 1];
#pragma unroll
  for (int j = 0; j < NIY; ++j) {
    const int pc = j * 64 + lane, row = pc >> 4, q = pc & 15;
    offy[j] = (unsigned)(row * (int)p.ldy + q * 4) * 4u;
  }
#pragma unroll
  for (int j = 0; j < NIA; ++j) {
    const int pc = j * 64 + lane, row = pc / (NP / 4), q = pc % (NP / 4);
    const int col = q * 4 < ncvp ? q * 4 : 0;     // columns past the operand: any valid address
    offa[j] = (unsigned)(row * (int)p.lda + col) * 4u;
  }
  // PRO: this lane's two columns of A (li, li + 32 of its tile) and their constants
  float psc[2] = {0.f, 0.f}, psh[2] = {0.f, 0.f};
  if (PRO) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = tn * 64 + li + 32 * j;
      if (col < N) { psc[j] = p.pscale[col]; psh[j] = p.pshift[col]; }
    }
  }
  const int prelu = p.prelu;
  unsigned char *ring = smem + (size_t)wave * S * SLOT;
  const unsigned ring_lds = (unsigned)(size_t)ring;
  auto issue = [&](long long chunk, int slot) {
    const float *by = dYt + chunk * 16 * p.ldy;
    const unsigned dst = ring_lds + (unsigned)slot * (unsigned)SLOT;
#pragma unroll
    for (int j = 0; j < NIY; ++j) glds16s(by, offy[j], dst + (unsigned)j * 1024u);
    if (NIA > 0) {
      const float *ba = At + chunk * 16 * p.lda;
#pragma unroll
      for (int j = 0; j < NIA; ++j) glds16s(ba, offa[j], dst + (unsigned)(NIY + j) * 1024u);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
  Planes ca[2], cb[2];
  {
    const bf16x8 z = __builtin_bit_cast(bf16x8, (u32x4){0u, 0u, 0u, 0u});
#pragma unroll
    for (int t = 0; t < 3; ++t) { ca[0].p[t] = z; ca[1].p[t] = z; cb[0].p[t] = z; cb[1].p[t] = z; }
  }
  // one k-step: read the chunk, split it into the next planes while the PREVIOUS planes are
  // multiplied; after the first products (which cover the LDS round trip) the slot, now in
  // registers, takes chunk `next` (>= 0).  Program order = issue order (sched_barrier pins it): one
  // MFMA, then one pair of values split into its plane dwords (11 full-rate VALU instructions).
  auto step = [&](int slot, long long next) {
    const float *sy = reinterpret_cast<const float *>(ring + (size_t)slot * SLOT);
    const float *sa = NIA == 0 ? sy : sy + 16 * C;
    const float *y0 = sy + (8 * lk) * C + li;
    const float *a0 = sa + (8 * lk) * NP + li;
    float va[2][8], vb[2][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      va[0][r] = y0[r * C];
      va[1][r] = y0[r * C + 32];
      vb[0][r] = a0[r * NP];
      vb[1][r] = a0[r * NP + 32];
    }
    u32x4 nh[4], nm[4], nl[4];                     // fragments a0, a1, b0, b1
#pragma unroll
    for (int t = 0; t < 24; ++t) {
      const int q = t >> 2, i = (t >> 1) & 1, j = t & 1;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[i].p[TA[q]], cb[j].p[TB[q]],
                                                          acc[i][j], 0, 0, 0);
      if (t == 5 && next >= 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot's values are in registers
        issue(next, slot);
      }
      if (t >= 6 && t < 22) {
        const int f = (t - 6) >> 2, d = (t - 6) & 3;
        float x0 = f == 0 ? va[0][2 * d] : f == 1 ? va[1][2 * d] : f == 2 ? vb[0][2 * d] : vb[1][2 * d];
        float x1 = f == 0 ? va[0][2 * d + 1] : f == 1 ? va[1][2 * d + 1]
                 : f == 2 ? vb[0][2 * d + 1] : vb[1][2 * d + 1];
        if (PRO && (f >= 2 || NIA == 0)) {       // NIA == 0 (A is dY, the Gram matrix): both operands
          x0 = x0 * psc[f & 1] + psh[f & 1];
          x1 = x1 * psc[f & 1] + psh[f & 1];
          if (prelu) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
        }
        unsigned h, m, l;
        split_pair(x0, x1, h, m, l);
        asm volatile("" : "+v"(h), "+v"(m), "+v"(l));    // formed HERE, not sunk behind the MFMAs
        nh[f][d] = h; nm[f][d] = m; nl[f][d] = l;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      ca[f].p[0] = __builtin_bit_cast(bf16x8, nh[f]);
      ca[f].p[1] = __builtin_bit_cast(bf16x8, nm[f]);
      ca[f].p[2] = __builtin_bit_cast(bf16x8, nl[f]);
      cb[f].p[0] = __builtin_bit_cast(bf16x8, nh[2 + f]);
      cb[f].p[1] = __builtin_bit_cast(bf16x8, nm[2 + f]);
      cb[f].p[2] = __builtin_bit_cast(bf16x8, nl[2 + f]);
    }
  };

  const long long n_my = p.nd > gw ? (p.nd - gw + GW - 1) / GW : 0;
  // + the chunks the DMA must not touch (ragged end, a piece that would overrun the tensor): the
  //   last wave fills its slot with guarded loads
  const long long n_all = n_my + (gw == GW - 1 ? p.nchunks - p.nd : 0);
#pragma unroll
  for (int j = 0; j < S; ++j)
    if (j < n_my) issue(gw + (long long)j * GW, j);
  int slot = 0;
#pragma unroll 1
  for (long long i = 0; i < n_all; ++i) {
    if (i < n_my) {
      // issued so far: chunks 0 .. min(n_my - 1, i + S - 1); those after i may stay in flight
      if (i + S - 1 <= n_my - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 1) * NI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      float *sy = reinterpret_cast<float *>(ring + (size_t)slot * SLOT);
      float *sa = NIA == 0 ? sy : sy + 16 * C;
      const long long r0 = (p.nd + (i - n_my)) * 16;
      for (int e = lane; e < 16 * C; e += 64) {
        const int row = e / C, col = e - row * C;
        sy[e] = r0 + row < M ? dYt[(r0 + row) * p.ldy + col] : 0.f;
      }
      if (NIA > 0)
        for (int e = lane; e < 16 * NP; e += 64) {
          const int row = e / NP, col = e - row * NP;
          sa[e] = (r0 + row < M && col < ncv) ? At[(r0 + row) * p.lda + col] : 0.f;
        }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    step(slot, i + S < n_my ? gw + (i + S) * GW : -1);
    slot = slot + 1 == S ? 0 : slot + 1;
  }
#pragma unroll
  for (int q = 0; q < 6; ++q)                    // the lagging product
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[i].p[TA[q]], cb[j].p[TB[q]],
                                                            acc[i][j], 0, 0, 0);

  // ---- the k-groups meet in LDS, one partial (C x N) tile per workgroup -----------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float *red = reinterpret_cast<float *>(smem);
  {
    float *dst = red + (size_t)wave * 4096;        // wave = kgid * ntile + tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[((i * 2 + j) * 16 + e) * 64 + lane] = acc[i][j][e];
  }
  __syncthreads();
  {
    // every wave sums a slice of the ntile x 4096 output elements over the k-groups
    float *out = p.part + (size_t)blockIdx.x * p.C * N;
    const int W = ntile * p.kg, total = ntile * 4096;
    for (int x = wave * 64 + lane; x < total; x += W * 64) {
      const int t = x >> 12, idx = x & 4095;
      float sacc = 0.f;
      for (int k = 0; k < p.kg; ++k) sacc += red[(size_t)(k * ntile + t) * 4096 + idx];
      const int ln = idx & 63, e = (idx >> 6) & 15, ij = idx >> 10;
      const int i = ij >> 1, j = ij & 1;
      // C/D layout of 32x32: col (b operand) = ln & 31, row (a operand) = (e&3) + 8(e>>2) + 4 (ln>>5)
      const int co = (t / p.tnw) * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5);
      const int ci = (t % p.tnw) * 64 + 32 * j + (ln & 31);
      if (ci < N) out[(size_t)co * N + ci] = sacc;
    }
  }
}

int g_dws_grid = 0;

// -> false: shape not taken
bool dws_plan(long long M, int C, int N, const float *dY, long long ldy, const float *A,
              long long lda, DwsArgs *o) {
  if (M < 1024 || C <= 0 || N <= 0 || C % 64 != 0 || ldy % 4 != 0 || ldy < C || lda < N) return false;
  if (((size_t)dY & 15) != 0 || ((size_t)A & 3) != 0) return false;
  DwsArgs a{};
  a.M = M; a.C = C; a.N = N; a.dY = dY; a.ldy = ldy; a.A = A; a.lda = lda;
  a.tcw = C / 64;
  a.tnw = (N + 63) / 64;
  const int nt = a.tcw * a.tnw;
  if (nt > 8) return false;
  a.same = (nt == 1 && A == dY && lda == ldy && C == N) ? 1 : 0;
  a.kg = 8 / nt;                         // 8, 4, 2 (6 waves for 3 tiles), 2, 1, ...
  a.npad = (N + 3) / 4 * 4;
  a.ni = a.same ? 4 : ((nt == 1 && N <= 16) ? 5 : 8);
  a.slot_bytes = a.ni * 1024 + (a.ni == 5 ? 1024 : 0);
  a.S = a.ni == 8 ? 2 : 3;
  if ((long long)16 * (ldy > lda ? ldy : lda) * 4 >= (1ll << 31)) return false;   // 32-bit lane offsets
  a.nchunks = (M + 15) / 16;
  // a DMA piece is 16 bytes: with N % 4 != 0 the last piece of a row runs into the next row, the
  // tensor's last row past its end -> the chunk that holds it is filled by guarded loads
  a.nd = (N % 4 == 0) ? M / 16 : (M - 1) / 16;
  *o = a;
  return true;
}

int dws_grid(const DwsArgs &a) {
  if (g_dws_grid <= 0) g_dws_grid = 240;      // (changed through s2c_weight_grad_stream_set_grid only)
  long long gsz = g_dws_grid;
  const long long units = (a.nchunks + a.kg - 1) / a.kg;
  if (gsz > units) gsz = units;
  return (int)gsz;
}

}  // namespace

// Number of (C x N) partial tiles s2c_weight_grad_stream writes (= its grid), 0: shape not taken.
extern "C" int s2c_weight_grad_stream_parts(long long M, int C, int N, const float *dY,
                                            long long ldy, const float *A, long long lda) {
  DwsArgs a;
  if (!dws_plan(M, C, N, dY, ldy, A, lda, &a)) return 0;
  return dws_grid(a);
}

// Workgroups of the persistent grid (default 240: the CUs the geometry stage holds on the side
// stream are left out, as for the streaming GEMM).  Returns the previous value.
extern "C" int s2c_weight_grad_stream_set_grid(int workgroups) {
  const int old = g_dws_grid ? g_dws_grid : 240;
  if (workgroups > 0) g_dws_grid = workgroups > 1024 ? 1024 : workgroups;
  return old;
}

static int dws_launch(long long M, int C, int N, const float *dY, long long ldy, const float *A,
                      long long lda, const float *pscale, const float *pshift, int prelu, float *part,
                      void *stream);

// dW partials: part[grid][C][N], every tile complete (the caller sums over the grid).
extern "C" int s2c_weight_grad_stream(long long M, int C, int N, const float *dY, long long ldy,
                                      const float *A, long long lda, float *part, void *stream) {
  return dws_launch(M, C, N, dY, ldy, A, lda, nullptr, nullptr, 0, part, stream);
}

// The same product with the operand relu?(A pscale[n] + pshift[n]) formed on the way (a layer's input
// activation from the previous layer's pre-activation A: the forward keeps no copy of it); shapes as
// s2c_weight_grad_stream_parts says, except N <= 16 single tiles (-2); A == dY: the activation's Gram matrix.
extern "C" int s2c_weight_grad_stream_act(long long M, int C, int N, const float *dY, long long ldy,
                                          const float *A, long long lda, const float *pscale,
                                          const float *pshift, int prelu, float *part, void *stream) {
  if (!pscale || !pshift) return -2;
  return dws_launch(M, C, N, dY, ldy, A, lda, pscale, pshift, prelu, part, stream);
}

static int dws_launch(long long M, int C, int N, const float *dY, long long ldy, const float *A,
                      long long lda, const float *pscale, const float *pshift, int prelu, float *part,
                      void *stream) {
  DwsArgs a;
  if (!part || !dws_plan(M, C, N, dY, ldy, A, lda, &a)) return -2;
  if (pscale && a.ni != 8 && a.ni != 4) return -2;      // (ni 4: A == dY, the Gram matrix of the activation)
  // the zero-filled pad rows of a ragged last chunk would be transformed to relu(pshift) as well: with
  // A != dY the dY rows beside them are zero and nothing is added, on the Gram path (A == dY) both
  // operands are transformed and each pad row would add relu(pshift)^T relu(pshift) -> not taken
  if (pscale && a.same && M % 16 != 0) return -2;
  a.pscale = pscale; a.pshift = pshift; a.prelu = prelu;
  a.part = part;
  const int grid = dws_grid(a);
  // the dynamic-LDS cap is a per-DEVICE function attribute: one flag per device of the process
  static bool attr_dev[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -2;
  if (!attr_dev[dev]) {
    const int cap = 160 * 1024;
    if (hipFuncSetAttribute((const void *)dw_private_kernel<4, 2>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
        hipFuncSetAttribute((const void *)dw_private_kernel<0, 3>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
        hipFuncSetAttribute((const void *)dw_private_kernel<1, 3>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
        hipFuncSetAttribute((const void *)dw_private_kernel<4, 2, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
        hipFuncSetAttribute((const void *)dw_private_kernel<0, 3, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess) {
      (void)hipGetLastError();
      return -2;
    }
    attr_dev[dev] = true;
  }
  const int PW = a.tcw * a.tnw * a.kg;
  size_t lds = (size_t)PW * a.S * a.slot_bytes + 1024;
  if (lds < (size_t)PW * 16384) lds = (size_t)PW * 16384;     // the waves' accumulators meet here
  const dim3 blk(64 * PW);
  if (a.ni == 8 && pscale)
    hipLaunchKernelGGL((dw_private_kernel<4, 2, true>), dim3(grid), blk, lds, (hipStream_t)stream, a);
  else if (a.ni == 8)
    hipLaunchKernelGGL((dw_private_kernel<4, 2>), dim3(grid), blk, lds, (hipStream_t)stream, a);
  else if (a.ni == 4 && pscale)
    hipLaunchKernelGGL((dw_private_kernel<0, 3, true>), dim3(grid), blk, lds, (hipStream_t)stream, a);
  else if (a.ni == 4)
    hipLaunchKernelGGL((dw_private_kernel<0, 3>), dim3(grid), blk, lds, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((dw_private_kernel<1, 3>), dim3(grid), blk, lds, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_weight_grad_stream launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
