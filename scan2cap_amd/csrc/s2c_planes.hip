// s2c_planes.hip -- fp32-accurate GEMMs on PRE-SPLIT bf16x3 planes: the greedy caption decoder
// (models/caption_module.py:502-592 `_forward_scene_batch`: R = B*K rows x 29 tokens, every
// product of `_step` :250-292 and the classifier :553) as hand-written MFMA kernels.
//
// Why planes.  The rows GEMMs of s2c_gemm.hip split each fp32 operand into hi + mid + lo bf16
// planes on the VALU after every fragment read (x = hi + mid + lo, 3 x 8 mantissa bits; the six
// plane products with i + j <= 2 on v_mfma_f32_32x32x16_bf16 are an fp32-accurate product at
// 2.7x the fp32 MFMA rate).  That is right for layers that stream a tall activation once.  The
// decoder's GEMMs are compute-bound (R x 812 x 300 ... R x 512 x 3500, R = 2048..8192) and every
// operand is re-read by 3..28 column tiles: here the weights are split ONCE per call and every
// activation is split ONCE, by the epilogue of the kernel that produces it, so the main loop is a
// pure matrix-core loop: 12 `ds_read_b128` feed 24 MFMAs (the three planes of an A / W fragment
// pair serve six products), no VALU work between them.
//
// Tile.  Workgroup = 4 waves on 128 rows x 128 columns, wave w = rows 32 w .. 32 w + 31 x all 128
// columns (4 accumulator tiles of 32 x 32): the four gate pre-activations (r, z, n_i, n_h) of
// a GRU unit then sit in the SAME lane and register of the four tiles and the GRU cell is the
// GEMM's epilogue.  K walks in chunks of 32; a chunk of both operands (3 planes x 128 rows x 64 B
// x 2 = 48 KB) is fetched by LDS-DMA (`global_load_lds_dwordx4`, no VGPR staging), rows of 64 B
// with the 16-byte slot XOR-swizzled by (row >> 2) & 3 on the SOURCE address (the DMA writes
// lane-linear), which makes the one-row-per-lane fragment reads conflict-free.  Single-buffered:
// three workgroups per CU (145 KB of LDS, <= 168 VGPRs) overlap one's DMA with another's MFMAs.
// Workgroup ids are XCD-aware: the 8 row tiles r = x, x + 8, ... of XCD x walk the column tiles
// together, so both operand tiles of a workgroup are L2 hits for all but the first toucher.
//
// The A operand is up to two K segments ([x | h] of a GRU cell, [word | h2] of map_topdown:
// no concatenation is ever materialised), each with an optional row map -- the greedy feedback
// `embeddings[argmax(logits)]` (caption_module.py:559-566) is a row map of the embedding table's
// planes, resolved in the consumer's prologue from the per-column-tile arg-max keys the classifier's
// epilogue leaves (first maximum, like torch.argmax): the logits are written once and never re-read.
//
// Epilogues (accumulators -> wave-private 4 KB LDS patch -> 8 consecutive columns of a row per lane):
// bias, row addend, ReLU, fp32 store (16-byte), bf16x3 plane store (16-byte per plane), arg-max key;
// GRU: r, z, n and h' = n + z (h - n) (the arithmetic of ATen's fused GRU cell) in the accumulator
// layout first.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

using namespace s2c;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PG_BM = 128, PG_BN = 128, PG_BK = 32;
constexpr unsigned PG_PLANE_BYTES = 128 * 64;                 // one plane of one operand tile
constexpr unsigned PG_W_BASE = 3 * PG_PLANE_BYTES;            // A planes, then W planes
constexpr unsigned PG_TOK_BASE = 6 * PG_PLANE_BYTES;          // 128 ints: resolved row map
constexpr unsigned PG_LDS_BYTES = PG_TOK_BASE + 128 * 4;

__device__ __forceinline__ void pg_glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void pg_split2(f32x2 v, unsigned &hi, unsigned &mid, unsigned &lo) {
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
  const bf16x2 m = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
  const bf16x2 l = __builtin_convertvector(r2, bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  mid = __builtin_bit_cast(unsigned, m);
  lo = __builtin_bit_cast(unsigned, l);
}

// 8 consecutive fp32 values -> one 16-byte piece of each plane
__device__ __forceinline__ void pg_split8(const float (&v)[8], uint4 &h, uint4 &m, uint4 &l) {
  pg_split2((f32x2){v[0], v[1]}, h.x, m.x, l.x);
  pg_split2((f32x2){v[2], v[3]}, h.y, m.y, l.y);
  pg_split2((f32x2){v[4], v[5]}, h.z, m.z, l.z);
  pg_split2((f32x2){v[6], v[7]}, h.w, m.w, l.w);
}

// order-preserving key of (value, column): larger value wins, equal values -> smaller column
// (torch.argmax: the first maximum)
__device__ __forceinline__ u64 pg_key(float v, int col) {
  const u32 b = __builtin_bit_cast(u32, v);
  const u32 o = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ((u64)o << 32) | (u64)(0xFFFFFFFFu - (u32)col);
}

__device__ __forceinline__ float pg_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// GRU = true: N = hidden units; column tile c = units 32 c .. 32 c + 31 as the four 32-column
// groups [r | z | n_i | n_h] (W rows 128 c ..), segment 0 = the cell's input x (groups r, z, n_i
// multiply), segment 1 = the previous hidden state (groups r, z, n_h).
template <bool GRU>
__global__ __launch_bounds__(256, 3) void planes_gemm_kernel(s2c_planes_gemm_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pg_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const int M = a.M, N = a.N;

  // ---- XCD-aware tile id: XCD x owns row tiles x, x + 8, ... and walks the column tiles ----
  const int nrt = (M + PG_BM - 1) / PG_BM;
  const int nct = GRU ? (N + 31) / 32 : (N + PG_BN - 1) / PG_BN;
  const int RT = (nrt + 7) >> 3;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ct = idx / RT, rt = (idx % RT) * 8 + xcd;
  if (rt >= nrt || ct >= nct) return;
  const int m0 = rt * PG_BM;
  const int cbase = GRU ? ct * 32 : ct * PG_BN;       // first output column (GRU: first unit)
  const long long wrow0 = (long long)ct * PG_BN;      // first W row of the tile

  // live 32-column groups (generic: those that start below N rounded up to 32)
  int live = 0xF;
  if (!GRU) {
    live = 0;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      if (cbase + 32 * nt < ((N + 31) & ~31)) live |= 1 << nt;
  }

  const unsigned lds0 = (unsigned)(size_t)pg_smem;
  int *s_tok = reinterpret_cast<int *>(pg_smem + PG_TOK_BASE);

  // ---- row map of segment 0 from the classifier's arg-max keys (greedy feedback) ----------
  if (a.tokkeys != nullptr) {
    if (tid < PG_BM) {
      const int row = m0 + tid < M ? m0 + tid : M - 1;
      const u64 *kp = a.tokkeys + (long long)row * a.ntokkeys;
      u64 best = 0;
      for (int j = 0; j < a.ntokkeys; ++j) { const u64 k = kp[j]; best = k > best ? k : best; }
      s_tok[tid] = (int)(0xFFFFFFFFu - (u32)best);
    }
    __syncthreads();
  }

  // ---- staging map: wave w fetches tile rows 32 w .. 32 w + 31 of both operands -------------
  // piece j = 16 rows x 64 B = one LDS-DMA instruction per plane; lane -> (row lane >> 2, slot
  // lane & 3), the slot holds source chunk slot ^ ((row >> 2) & 3)
  const int sr = lane >> 2, sslot = lane & 3;
  long long aoff[2][2];          // [segment][piece]: element offset of (source row, chunk col)
  int scol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rr = 32 * wave + 16 * j + sr;
    scol[j] = 8 * (sslot ^ ((rr >> 2) & 3));
    const int row = m0 + rr < M ? m0 + rr : M - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      long long src = row;
      if (s < a.nseg) {
        const s2c_planes_seg &sg = a.seg[s];
        if (s == 0 && a.tokkeys != nullptr) src = s_tok[rr];
        else if (sg.rowmap != nullptr) src = sg.rowmap[row];
        else if (sg.rowdiv > 0) src = row / sg.rowdiv;
        aoff[s][j] = src * sg.ld + scol[j];
      } else {
        aoff[s][j] = 0;
      }
    }
  }
  long long woff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
    woff[j] = (wrow0 + 32 * wave + 16 * j + sr) * (long long)a.ldw + scol[j];

  f32x16 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

  // fragment addresses: row li of the wave's A rows / of W group nt, 16-byte slot (2 s + lk) ^ f
  const unsigned fsw = (unsigned)((li >> 2) & 3);
  const unsigned fa_base = lds0 + (unsigned)(32 * wave + li) * 64u;
  const unsigned fw_base = lds0 + PG_W_BASE + (unsigned)li * 64u;

  int wchunk = 0;                                   // chunk index along W's K
#pragma unroll
  for (int s = 0; s < 2; ++s) {                     // (unrolled: aoff[s] stays in registers)
    if (s >= a.nseg) break;
    const s2c_planes_seg sg = a.seg[s];
    const int mask = GRU ? (s == 0 ? 0x7 : 0xB) : live;
    const bool stage_w = (mask >> wave) & 1;
    for (int c = 0; c < sg.kc; ++c, ++wchunk) {
      // ---- fetch the chunk ----
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned short *ap = sg.p + aoff[s][j] + 32 * c;
        const unsigned dst = lds0 + (unsigned)(32 * wave + 16 * j) * 64u;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          pg_glds16(ap + pl * sg.pstride, dst + pl * PG_PLANE_BYTES);
        if (stage_w) {
          const unsigned short *wp = a.W + woff[j] + 32 * wchunk;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            pg_glds16(wp + pl * a.wpstride, dst + PG_W_BASE + pl * PG_PLANE_BYTES);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // ---- 2 k-steps of 16: 3 A fragments, per live group 3 W fragments and 6 products ----
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned slot = ((unsigned)(2 * ks + lk) ^ fsw) * 16u;
        bf16x8 fa[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          fa[pl] = *reinterpret_cast<const bf16x8 *>(
              pg_smem + (fa_base - lds0) + pl * PG_PLANE_BYTES + slot);
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          if (!((mask >> nt) & 1)) continue;
          bf16x8 fb[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            fb[pl] = *reinterpret_cast<const bf16x8 *>(
                pg_smem + (fw_base - lds0) + pl * PG_PLANE_BYTES + (unsigned)nt * 2048u + slot);
#pragma unroll
          for (int q = 0; q < 6; ++q)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[q]], fb[TB[q]], acc[nt], 0, 0, 0);
        }
      }
      __syncthreads();                               // the tiles are overwritten next
    }
  }

  // ------------------------------------------------------------------------------------------
  // epilogue.  C/D layout of 32x32: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
  // Through a wave-private 32 x 32 fp32 patch (the staging tiles are dead: every wave passed the
  // barrier above) each lane gets 8 consecutive columns of rows (lane >> 2) and (lane >> 2) + 16.
  float *patch = reinterpret_cast<float *>(pg_smem) + wave * 1024;
  const int prow = lane >> 2, pcol = 8 * (lane & 3);

  if (GRU) {
    const int u = cbase + li;
    const bool uok = u < N;
    const float br = uok ? a.bias[u] : 0.f, bz = uok ? a.bias[N + u] : 0.f;
    const float bni = uok ? a.bias[2 * N + u] : 0.f, bnh = uok ? a.bias[3 * N + u] : 0.f;
    float hp[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * lk;
      hp[e] = (row < M && uok) ? a.hprev[(long long)row * a.ldh + u] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float r = pg_sigmoid(acc[0][e] + br);
      const float z = pg_sigmoid(acc[1][e] + bz);
      const float n = tanhf((acc[2][e] + bni) + r * (acc[3][e] + bnh));
      acc[0][e] = n + z * (hp[e] - n);
    }
  }

  u64 best[2] = {0, 0};
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    if (GRU ? nt > 0 : !((live >> nt) & 1)) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      patch[((e & 3) + 8 * (e >> 2) + 4 * lk) * 32 + li] = acc[nt][e];
    __builtin_amdgcn_wave_barrier();
    const int col0 = cbase + 32 * nt + pcol;        // first of the lane's 8 columns
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = m0 + 32 * wave + prow + 16 * p;
      const float4 v0 = *reinterpret_cast<const float4 *>(patch + (prow + 16 * p) * 32 + pcol);
      const float4 v1 = *reinterpret_cast<const float4 *>(patch + (prow + 16 * p) * 32 + pcol + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      if (row >= M) continue;
      if (!GRU) {
        const bool full = col0 + 8 <= N;
        if (a.bias != nullptr) {
          if (full) {
            const float4 b0 = *reinterpret_cast<const float4 *>(a.bias + col0);
            const float4 b1 = *reinterpret_cast<const float4 *>(a.bias + col0 + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) if (col0 + i < N) v[i] += a.bias[col0 + i];
          }
        }
        if (a.add != nullptr) {
          const float *ad = a.add + (long long)row * a.ldadd + col0;
          if (full && (a.ldadd & 3) == 0) {
            const float4 b0 = *reinterpret_cast<const float4 *>(ad);
            const float4 b1 = *reinterpret_cast<const float4 *>(ad + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) if (col0 + i < N) v[i] += ad[i];
          }
        }
        if (a.relu) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) if (col0 + i >= N) v[i] = 0.f;   // plane padding stays finite
        if (a.amax != nullptr) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (col0 + i < N) { const u64 k = pg_key(v[i], col0 + i); best[p] = k > best[p] ? k : best[p]; }
        }
      }
      if (a.C != nullptr) {
        float *cp = a.C + (long long)row * a.ldc + col0;
        if (col0 + 8 <= N && (a.ldc & 3) == 0) {
          *reinterpret_cast<float4 *>(cp) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4 *>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) if (col0 + i < N) cp[i] = v[i];
        }
      }
      if (a.P != nullptr && col0 < a.ldp) {          // ldp: a multiple of 32 >= N
        uint4 h, m, l;
        pg_split8(v, h, m, l);
        unsigned short *pp = a.P + (long long)row * a.ldp + col0;
        *reinterpret_cast<uint4 *>(pp) = h;
        *reinterpret_cast<uint4 *>(pp + a.ppstride) = m;
        *reinterpret_cast<uint4 *>(pp + 2 * a.ppstride) = l;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (!GRU && a.amax != nullptr) {
    // a row's 128 columns sit in the 4 lanes of a quad: fold, lane 0 of the quad writes
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      u64 k = best[p];
      k = umax64(k, dpp_mov_u64<DPP_QUAD_1032>(k));
      k = umax64(k, dpp_mov_u64<DPP_QUAD_2301>(k));
      const int row = m0 + 32 * wave + prow + 16 * p;
      if ((lane & 3) == 0 && row < M) a.amax[(long long)row * a.namax + ct] = k;
    }
  }
}

// fp32 (rows_in x K, row stride ldx) -> planes (3 x rows_out x ldp) bf16, zero beyond the matrix
__global__ __launch_bounds__(256) void planes_split_kernel(
    long long rows_in, int K, const float *__restrict__ X, long long ldx, long long rows_out,
    int ldp, unsigned short *__restrict__ P, long long pstride) {
  const int per_row = ldp >> 3;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows_out * per_row) return;
  const long long r = i / per_row;
  const int k = 8 * (int)(i % per_row);
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = 0.f;
  if (r < rows_in) {
    const float *src = X + r * ldx + k;
    if (k + 8 <= K && (ldx & 3) == 0 && (((uintptr_t)X) & 15) == 0) {
      const float4 a0 = *reinterpret_cast<const float4 *>(src);
      const float4 a1 = *reinterpret_cast<const float4 *>(src + 4);
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w;
      v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) if (k + q < K) v[q] = src[q];
    }
  }
  uint4 h, m, l;
  pg_split8(v, h, m, l);
  unsigned short *pp = P + r * ldp + k;
  *reinterpret_cast<uint4 *>(pp) = h;
  *reinterpret_cast<uint4 *>(pp + pstride) = m;
  *reinterpret_cast<uint4 *>(pp + 2 * pstride) = l;
}

int pg_chk(const char *k) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c: %s launch failed: %s\n", k, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

template <bool GRU>
int pg_launch(const s2c_planes_gemm_args &a, hipStream_t st) {
  static int attr_state[64];                   // per device: 0 unknown, 1 ok, -1 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (attr_state[dev] == 0)
    attr_state[dev] = hipFuncSetAttribute((const void *)planes_gemm_kernel<GRU>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)PG_LDS_BYTES) == hipSuccess ? 1 : -1;
  if (attr_state[dev] < 0) return -3;
  const int nrt = (a.M + PG_BM - 1) / PG_BM;
  const int nct = GRU ? (a.N + 31) / 32 : (a.N + PG_BN - 1) / PG_BN;
  const int RT = (nrt + 7) / 8;
  hipLaunchKernelGGL(planes_gemm_kernel<GRU>, dim3(8 * RT * nct), dim3(256), PG_LDS_BYTES, st, a);
  return pg_chk("planes_gemm");
}

}  // namespace

extern "C" int s2c_planes_gemm(const s2c_planes_gemm_args *a, void *stream) {
  if (a == nullptr || a->M <= 0 || a->N <= 0 || a->nseg < 1 || a->nseg > 2 || a->W == nullptr ||
      (a->ldw & 31))
    return -1;
  for (int s = 0; s < a->nseg; ++s)
    if (a->seg[s].p == nullptr || a->seg[s].kc <= 0 || (a->seg[s].ld & 7)) return -1;
  if (a->P != nullptr && ((a->ldp & 31) || a->ldp < a->N)) return -1;
  if (a->tokkeys != nullptr && a->ntokkeys <= 0) return -1;
  if (a->gru) {
    if (a->nseg != 2 || a->bias == nullptr || a->hprev == nullptr || (a->N & 31)) return -1;
    return pg_launch<true>(*a, (hipStream_t)stream);
  }
  if (a->amax != nullptr && a->namax < (a->N + PG_BN - 1) / PG_BN) return -1;
  return pg_launch<false>(*a, (hipStream_t)stream);
}

extern "C" int s2c_planes_split(long long rows_in, int K, const float *X, long long ldx,
                                long long rows_out, int ldp, unsigned short *P,
                                long long pstride, void *stream) {
  if (rows_out <= 0 || rows_in < 0 || rows_in > rows_out || K < 0 || (ldp & 7) || ldp < K ||
      P == nullptr || (rows_in > 0 && X == nullptr))
    return -1;
  const long long n = rows_out * (ldp >> 3);
  hipLaunchKernelGGL(planes_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, rows_in, K, X, ldx, rows_out, ldp, P, pstride);
  return pg_chk("planes_split");
}

// sizeof the argument structs (0: s2c_planes_gemm_args, 1: s2c_planes_seg) -- for bindings to
// check their layout
extern "C" long long s2c_planes_args_sizeof(int which) {
  return which == 0 ? (long long)sizeof(s2c_planes_gemm_args) : (long long)sizeof(s2c_planes_seg);
}
