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
// Tile.  Workgroup = 128 rows x 128 columns; a wave = rows 32 w .. 32 w + 31 x all 128
// columns (4 accumulator tiles of 32 x 32): the four gate pre-activations (r, z, n_i, n_h) of
// a GRU unit then sit in the SAME lane and register of the four tiles and the GRU cell is the
// GEMM's epilogue.  K walks in chunks of 32; a chunk of both operands (3 planes x 128 rows x 64 B
// x 2 = 48 KB) is fetched by LDS-DMA (`global_load_lds_dwordx4`, no VGPR staging), rows of 64 B
// with the 16-byte slot XOR-swizzled by (row >> 2) & 3 on the SOURCE address (the DMA writes
// lane-linear), which makes the one-row-per-lane fragment reads conflict-free.  The chunks go
// through a ring of THREE stages (144 KB: one workgroup per CU) requested two chunks ahead, counted
// with `s_waitcnt vmcnt(N)` and ONE raw `s_barrier` per chunk: the matrix pipe of every SIMD runs
// back to back while 96 KB per CU are in flight.  (First version: one stage, three workgroups per
// CU covering each other's DMA latency -- 0.36 of the bf16 roof on the classifier, 0.05-0.16 on
// the narrow products whose grid does not fill the chip three times; tools/bench_planes.py.)
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
#include <stdlib.h>
#include <type_traits>

using namespace s2c;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// TILED plane layout: a matrix is stored as blocks of 32 rows x 16 k = 1 KB, each block the exact
// lane-linear image one LDS-DMA instruction lands in LDS (row r32 at 32 r32 bytes, its two 16-byte
// halves swapped where (r32 >> 3) & 1), blocks ordered [row block][k block].  A DMA piece is then
// 8 full cache lines instead of 32 half-used ones: the DMA path pays per line it touches (the
// loop without products, classifier shape at R = 8192: 80 us with 64-byte row pieces, 137 us with
// 32-byte ones, both from row-major planes).
__device__ __forceinline__ long long pg_tiled_off(long long r, int k, int ld) {
  const int r32 = (int)(r & 31);
  return ((r >> 5) * (ld >> 4) + (k >> 4)) * 512 +
         ((r32 * 2 + (((k >> 3) & 1) ^ ((r32 >> 3) & 1))) << 3) + (k & 7);
}

__device__ __forceinline__ float pg_fast_sigmoid(float x) {   // hardware exp2 / rcp: ~2e-7 absolute
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float pg_fast_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.8853900817779268f) + 1.0f);
}

// Geometry of a workgroup: WM = 4 row groups x WN column halves of waves; a wave = RT row tiles of
// 32 x 4 column tiles of 32 (its 128 columns: the four gate groups of 32 GRU units).
//   small (RT 1, WN 1): 4 waves, 128 x 128, stage 24 KB, two workgroups per CU
//   big   (RT 2, WN 2): 8 waves, 256 x 256, stage 48 KB, one workgroup per CU
// ST = stages of the LDS-DMA ring = 3.  (A lone 128 x 128 workgroup on its CU -- grids of <= 256
// tiles -- takes 0.85 us per 16-k chunk, twice its MFMA time; a SIX-stage ring, five chunks in
// flight, was measured and is 15-25 % SLOWER: what a lone wave per SIMD pays is the serial issue of
// its six DMA pieces (~110 cycles each) in front of its 24 MFMAs, not the DMA latency.)
// Either way a chunk of 16 k is (4 RT + 4 WN) blocks of 32 rows per operand plane = ONE LDS-DMA
// instruction per wave, operand and plane (6 per wave and chunk).
template <int RT, int WN, int ST>
struct PgGeo {
  static constexpr int WM = 4, NW = WM * WN;
  static constexpr int BM = 32 * RT * WM, BN = 128 * WN;
  static constexpr int ABLK = RT * WM, WBLK = 4 * WN;            // 32-row blocks per operand
  static constexpr unsigned A_PLANE = 1024u * ABLK, W_PLANE = 1024u * WBLK;
  static constexpr unsigned W_BASE = 3 * A_PLANE;
  static constexpr unsigned STAGE = 3 * (A_PLANE + W_PLANE);
  static constexpr unsigned TOK_BASE = ST * STAGE;              // BM ints: resolved row map
  static constexpr unsigned LDS = TOK_BASE + 4 * BM;
};

// one k-step of 16 of the staged chunk for the accumulator tiles in MASK (compile-time: straight-
// line code, the W fragments of the next tile are requested before the products of this one)
// `dma(k)`, k = 0..5, requests the wave's k-th LDS-DMA piece of the chunk two ahead: the pieces are
// spread over the products (two per accumulator-tile group) instead of issued as one burst -- an
// LDS-DMA instruction holds the wave's issue slot for ~110 cycles, and a burst of six at the top of
// the chunk in BOTH waves of a SIMD left the matrix pipe idle for a third of the 256 x 256 loop.
template <int MASK, int RT, unsigned A_PLANE, unsigned W_PLANE, typename DMA>
__device__ __forceinline__ void pg_kstep(f32x16 (&acc)[RT][4], const unsigned char *stage,
                                         unsigned fa_off, unsigned fw_off, DMA dma) {
  bf16x8 fa[RT][3], fb[2][3];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      fa[i][pl] = *reinterpret_cast<const bf16x8 *>(stage + fa_off + pl * A_PLANE + i * 1024u);
  constexpr int first = (MASK & 1) ? 0 : (MASK & 2) ? 1 : (MASK & 4) ? 2 : 3;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl)
    fb[0][pl] = *reinterpret_cast<const bf16x8 *>(stage + fw_off + pl * W_PLANE + first * 1024u);
  constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
  int cur = 0, piece = 0;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    if (!((MASK >> nt) & 1)) continue;
    int nxt = -1;
#pragma unroll
    for (int m = 3; m > nt; --m) if ((MASK >> m) & 1) nxt = m;
    if (nxt >= 0) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        fb[cur ^ 1][pl] = *reinterpret_cast<const bf16x8 *>(stage + fw_off + pl * W_PLANE + nxt * 1024u);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
      for (int i = 0; i < RT; ++i)
        acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][TA[q]], fb[cur][TB[q]], acc[i][nt], 0, 0, 0);
      if (q == 1 || q == 3) {
        __builtin_amdgcn_sched_barrier(0);
        dma(piece++);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    cur ^= 1;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (k >= piece) dma(k);
}

// GRU = true: N = hidden units; a wave's 128 columns = units 32 c .. 32 c + 31 as the four
// 32-column groups [r | z | n_i | n_h] (W rows 128 c ..), segment 0 = the cell's input x (groups r,
// z, n_i multiply), segment 1 = the previous hidden state (groups r, z, n_h).
//
// What shaped it (classifier shape R = 8192 x 512 -> 3500, tools/bench_planes.py + S2C_PLANES_DBG):
// the LDS-DMA path delivers ~35 B/clk/CU whatever the layout of the source (row-major 64-byte or
// 32-byte row pieces, or the tiled 1 KB blocks: ~110 cycles of a SIMD per instruction), i.e.
// ~20 TB/s chip-wide, and a 128 x 128 tile needs 1 byte per 128 flop -- exactly the ratio of the
// bf16 roof to that rate: DMA and MFMA time are equal (~75 us each) and every 128 x 128 variant
// landed at 195-257 us (one 32-k stage x 3 workgroups per CU: 195; 3-stage ring, 1 workgroup per
// CU: 257, nothing overlapped; the same as 8 waves = 2 k-halves per row group: 226; 3-stage ring
// of 16-k chunks, 2 workgroups per CU: 233).  Hence the 256 x 256 tile (half the bytes per flop,
// two waves per SIMD: one's DMA issue and fragment reads under the other's MFMAs) wherever the
// grid still fills the chip, and the 128 x 128 tile with two workgroups per CU elsewhere.
template <bool GRU, int RT, int WN, int ST>
__global__ __launch_bounds__(64 * 4 * WN, (WN == 1 && ST == 3) ? 2 : 1) void planes_gemm_kernel(s2c_planes_gemm_args a) {
  typedef PgGeo<RT, WN, ST> G;
  constexpr int D = ST - 1;                     // chunks requested ahead of the one being multiplied
  extern __shared__ __attribute__((aligned(16))) unsigned char pg_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lk = lane >> 5;
  const int M = a.M, N = a.N;

  // ---- XCD-aware tile id: XCD x owns row tiles x, x + 8, ... and walks the column tiles ----
  const int nrt = (M + G::BM - 1) / G::BM;
  const int nct = GRU ? (N + 32 * WN - 1) / (32 * WN) : (N + G::BN - 1) / G::BN;
  const int RTX = (nrt + 7) >> 3;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ct = idx / RTX, rt = (idx % RTX) * 8 + xcd;
  if (rt >= nrt || ct >= nct) return;
  const int m0 = rt * G::BM;
  // first output column of the WAVE (GRU: its first unit), first W row of the workgroup
  const int cbase = GRU ? (ct * WN + wn) * 32 : ct * G::BN + 128 * wn;
  const long long wrow0 = (long long)ct * G::BN;

  // two outputs side by side (nsplit > 0): W rows [0, nsplit) -> C (n1 valid columns, bias, arg-max
  // keys, planes), W rows [nsplit, N) -> C2 (N - nsplit columns, row addend `add`)
  const bool part2 = !GRU && a.nsplit > 0 && cbase >= a.nsplit;
  const int obase = part2 ? cbase - a.nsplit : cbase;         // the wave's first OUTPUT column
  const int NV = GRU ? N : (part2 ? N - a.nsplit : (a.nsplit > 0 ? a.n1 : N));   // valid output columns
  float *const Cp = part2 ? a.C2 : a.C;
  const int ldcp = part2 ? a.ldc2 : a.ldc;
  const float *const biasp = part2 ? nullptr : a.bias;
  const float *const addp = (a.nsplit > 0 && !part2) ? nullptr : a.add;
  unsigned long long *const amaxp = part2 ? nullptr : a.amax;
  unsigned short *const Pp = part2 ? nullptr : a.P;

  // live 32-column groups of the wave (generic: those that start below NV rounded up to 32)
  int live = 0xF;
  if (!GRU) {
    live = 0;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      if (obase + 32 * nt < ((NV + 31) & ~31)) live |= 1 << nt;
  } else if (cbase >= N) {
    live = 0;
  }

  const unsigned lds0 = (unsigned)(size_t)pg_smem;
  int *s_tok = reinterpret_cast<int *>(pg_smem + G::TOK_BASE);

  // ---- row map of segment 0 from the classifier's arg-max keys (greedy feedback) ----------
  if (a.tokkeys != nullptr) {
    for (int t = tid; t < G::BM; t += 64 * G::NW) {
      const int row = m0 + t < M ? m0 + t : M - 1;
      const u64 *kp = a.tokkeys + (long long)row * a.ntokkeys;
      u64 best = 0;
      for (int j = 0; j < a.ntokkeys; ++j) { const u64 k = kp[j]; best = k > best ? k : best; }
      s_tok[t] = (int)(0xFFFFFFFFu - (u32)best);
    }
    __syncthreads();
  }

  // ---- staging map: wave w fetches 32-row block w of both operands (of RT WM resp. 4 WN) -----
  // a block = 32 rows x 32 B (16 k) = one LDS-DMA instruction per plane; lane -> (row lane >> 1,
  // 16-byte slot lane & 1), the slot holds source half slot ^ ((row >> 3) & 1): the one-row-per-
  // lane fragment reads (ds_read_b128, 16 lanes per LDS cycle) then touch every bank once.  Tiled
  // planes (include/s2c_fused.h) hold exactly this image: the DMA is a linear 1 KB copy.
  const int kc0 = 2 * a.seg[0].kc, kct = kc0 + (a.nseg > 1 ? 2 * a.seg[1].kc : 0);   // chunks of 16 k
  const bool stage_a = wave < G::ABLK, stage_wb = wave < G::WBLK;
  long long aoff0, aoff1, woff;      // element offsets of the wave's first piece of a segment
  int astep0 = 16, astep1 = 16;      // ... and from one 16-k chunk to the next
  {
    const int rr = 32 * wave + (lane >> 1);            // tile row of this lane's piece
    const int scol = 8 * ((lane & 1) ^ ((rr >> 3) & 1));
    const int row = m0 + rr < M ? m0 + rr : M - 1;
    const long long rb = (m0 >> 5) + wave;             // the wave's 32-row block of A
    long long src = row;
    if (a.tokkeys != nullptr) src = s_tok[stage_a ? rr : 0];
    else if (a.seg[0].rowmap != nullptr) src = a.seg[0].rowmap[row];
    else if (a.seg[0].rowdiv > 0) src = row / a.seg[0].rowdiv;
    aoff0 = src * a.seg[0].ld + scol;
    if (a.seg[0].tiled) { aoff0 = rb * (a.seg[0].ld >> 4) * 512 + lane * 8; astep0 = 512; }
    src = row;
    aoff1 = 0;
    if (a.nseg > 1) {
      if (a.seg[1].rowmap != nullptr) src = a.seg[1].rowmap[row];
      else if (a.seg[1].rowdiv > 0) src = row / a.seg[1].rowdiv;
      aoff1 = src * a.seg[1].ld + scol;
      if (a.seg[1].tiled) { aoff1 = rb * (a.seg[1].ld >> 4) * 512 + lane * 8; astep1 = 512; }
    }
    woff = ((wrow0 >> 5) + wave) * (long long)(a.ldw >> 4) * 512 + lane * 8;     // W: always tiled
  }
  const unsigned short *p0 = a.seg[0].p, *p1 = a.seg[1].p;
  const long long ps0 = a.seg[0].pstride, ps1 = a.seg[1].pstride;
  const unsigned piece_off = (unsigned)wave * 1024u;

  f32x16 acc[RT][4];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][nt][e] = 0.f;

  // fragment addresses: row li of the wave's A blocks RT wm + i / of W block 4 wn + nt, 16-byte
  // slot lk ^ f
  const unsigned slot = ((unsigned)lk ^ (unsigned)((li >> 3) & 1)) * 16u;
  const unsigned fa_off = (unsigned)(RT * wm) * 1024u + (unsigned)li * 32u + slot;
  const unsigned fw_off = G::W_BASE + (unsigned)(4 * wn) * 1024u + (unsigned)li * 32u + slot;

  // ---- K loop: a ring of three stages filled by LDS-DMA two chunks ahead ---------------------
  // chunk g lives in stage g % 3.  Iteration c: wait until this wave's pieces of chunk c have
  // landed (only the pieces of chunk c + 1 may still be in flight), ONE barrier (everybody's
  // pieces landed; everybody is done reading stage (c + 2) % 3 = the stage of chunk c - 1),
  // request chunk c + 2, multiply chunk c.  The barrier is the raw instruction: a
  // __syncthreads() would also wait for the DMA just issued.
  const int dbg = a.dbg;       // bench only: bit 0 = no MFMA, bit 1 = no DMA
  // W blocks whose products are skipped are not fetched either (GRU: n_h in segment 0, n_i in 1)
  auto chunk_wmask = [&](int g) { return GRU ? (g < kc0 ? 0x7 : 0xB) : 0xF; };
  // the (up to) six pieces of chunk g: 0..2 the A block's planes, 3..5 the W block's
  const unsigned short *dma_a = nullptr, *dma_w = nullptr;
  long long dma_pst = 0;
  unsigned dma_dst = 0;
  auto prep = [&](int g, unsigned stage_base) -> int {       // returns the DMA count of the chunk
    dma_a = dma_w = nullptr;
    if (g >= kct || (dbg & 2)) return 0;
    const bool s1 = g >= kc0;
    const int lc = s1 ? g - kc0 : g;
    dma_pst = s1 ? ps1 : ps0;
    dma_dst = stage_base + piece_off;
    int n = 0;
    if (stage_a) {
      dma_a = s1 ? p1 + aoff1 + (long long)astep1 * lc : p0 + aoff0 + (long long)astep0 * lc;
      n += 3;
    }
    if (stage_wb && ((chunk_wmask(g) >> (wave & 3)) & 1)) {
      dma_w = a.W + woff + 512ll * g;
      n += 3;
    }
    return n;
  };
  auto piece = [&](int k) {
    if (k < 3) {
      if (dma_a != nullptr) pg_glds16(dma_a + k * dma_pst, dma_dst + k * G::A_PLANE);
    } else if (k < 6 && dma_w != nullptr) {
      pg_glds16(dma_w + (k - 3) * a.wpstride, dma_dst + G::W_BASE + (k - 3) * G::W_PLANE);
    }
  };
  auto issue = [&](int g, unsigned stage_base) -> int {
    const int n = prep(g, stage_base);
#pragma unroll
    for (int k = 0; k < 6; ++k) piece(k);
    return n;
  };
  // pieces this wave requests for chunk g (the vmcnt bookkeeping below must match `prep`)
  auto nof = [&](int g) -> int {
    if (g >= kct || (dbg & 2)) return 0;
    return (stage_a ? 3 : 0) + ((stage_wb && ((chunk_wmask(g) >> (wave & 3)) & 1)) ? 3 : 0);
  };
  // chunk g lives in stage g % ST; chunks 0 .. D - 1 are requested up front
#pragma unroll
  for (int g = 0; g < D; ++g) issue(g, lds0 + g * G::STAGE);
  // `allowed` = pieces of the chunks c + 1 .. c + D - 1: what may still be in flight when chunk c
  // is needed (LDS-DMA completes in order)
  int allowed = 0;
#pragma unroll
  for (int g = 1; g < D; ++g) allowed += nof(g);
  unsigned st_c = 0, st_p = ST - 1;                             // stage of chunk c / of chunk c - 1
  // the chunk loop with the set of live accumulator tiles as a compile-time constant (as a run-
  // time test inside the loop every tile's products sat in their own basic block: no fragment
  // read ahead of a product, and the accumulators were copied between the blocks' registers)
  auto run = [&](auto mask_c, int c_begin, int c_end) {
    constexpr int MASK = decltype(mask_c)::value;
    for (int c = c_begin; c < c_end; ++c) {
      switch (allowed / 3) {       // s_waitcnt takes an immediate
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;   // ST <= 7
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // chunk c + D goes where chunk c - 1 was: everybody is past the barrier, i.e. done with it
      prep(c + D, lds0 + st_p * G::STAGE);
      allowed += nof(c + D) - nof(c + 1);
      if (MASK != 0 && !(dbg & 1)) {
        pg_kstep<MASK == 0 ? 1 : MASK, RT, G::A_PLANE, G::W_PLANE>(acc, pg_smem + st_c * G::STAGE, fa_off,
                                                                   fw_off, piece);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) piece(k);
      }
      st_p = st_c;
      st_c = st_c == ST - 1 ? 0 : st_c + 1;
    }
  };
  if (GRU) {
    if (live) {
      run(std::integral_constant<int, 0x7>(), 0, kc0);
      run(std::integral_constant<int, 0xB>(), kc0, kct);
    } else {
      run(std::integral_constant<int, 0>(), 0, kct);
    }
  } else if (live == 0xF) {
    run(std::integral_constant<int, 0xF>(), 0, kct);
  } else if (live == 0x7) {
    run(std::integral_constant<int, 0x7>(), 0, kct);
  } else if (live == 0x3) {
    run(std::integral_constant<int, 0x3>(), 0, kct);
  } else if (live == 0x1) {
    run(std::integral_constant<int, 0x1>(), 0, kct);
  } else {
    run(std::integral_constant<int, 0>(), 0, kct);             // a wave beyond N: DMA and barriers only
  }
  __syncthreads();                                   // the staging tiles are dead from here on
  if (live == 0) return;

  // ------------------------------------------------------------------------------------------
  // epilogue.  C/D layout of 32x32: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
  // Through a wave-private 32 x 32 fp32 patch (the staging tiles are dead) each lane gets 8
  // consecutive columns of rows (lane >> 2), (lane >> 2) + 16.
  float *patch = reinterpret_cast<float *>(pg_smem) + wave * 1024;
  const int prow = lane >> 2, pcol = 8 * (lane & 3);
  const int mrow0 = m0 + 32 * RT * wm;                 // first row of the wave

  if (GRU) {
    const int u = cbase + li;
    const bool uok = u < N;
    const float br = uok ? a.bias[u] : 0.f, bz = uok ? a.bias[N + u] : 0.f;
    const float bni = uok ? a.bias[2 * N + u] : 0.f, bnh = uok ? a.bias[3 * N + u] : 0.f;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      float hp[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mrow0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lk;
        hp[e] = (row < M && uok) ? a.hprev[(long long)row * a.ldh + u] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float r = pg_fast_sigmoid(acc[i][0][e] + br);
        const float z = pg_fast_sigmoid(acc[i][1][e] + bz);
        const float n = pg_fast_tanh((acc[i][2][e] + bni) + r * (acc[i][3][e] + bnh));
        acc[i][0][e] = n + z * (hp[e] - n);
      }
    }
  }

  // bias of all live tiles requested before the first patch round trip (a load per tile inside
  // the loop below was a chain of L2 latencies)
  float bv[GRU ? 1 : 4][8];
  if (!GRU) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int col0 = obase + 32 * nt + pcol;
#pragma unroll
      for (int i = 0; i < 8; ++i) bv[nt][i] = 0.f;
      if (biasp != nullptr && ((live >> nt) & 1)) {
        if (col0 + 8 <= NV) {
          const float4 b0 = *reinterpret_cast<const float4 *>(biasp + col0);
          const float4 b1 = *reinterpret_cast<const float4 *>(biasp + col0 + 4);
          bv[nt][0] = b0.x; bv[nt][1] = b0.y; bv[nt][2] = b0.z; bv[nt][3] = b0.w;
          bv[nt][4] = b1.x; bv[nt][5] = b1.y; bv[nt][6] = b1.z; bv[nt][7] = b1.w;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) if (col0 + i < NV) bv[nt][i] = biasp[col0 + i];
        }
      }
    }
  }

#pragma unroll
  for (int i = 0; i < RT; ++i) {
    u64 best[2] = {0, 0};
    // the row addends of this row tile, all live column tiles at once
    float av[GRU ? 1 : 4][GRU ? 1 : 2][8];
    if (!GRU) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int col0 = obase + 32 * nt + pcol;
          const int row = mrow0 + 32 * i + prow + 16 * p;
#pragma unroll
          for (int q = 0; q < 8; ++q) av[nt][p][q] = 0.f;
          if (addp != nullptr && ((live >> nt) & 1)) {
            const float *ad = addp + (long long)(row < M ? row : M - 1) * a.ldadd + col0;
            if (col0 + 8 <= NV && (a.ldadd & 3) == 0) {
              const float4 b0 = *reinterpret_cast<const float4 *>(ad);
              const float4 b1 = *reinterpret_cast<const float4 *>(ad + 4);
              av[nt][p][0] = b0.x; av[nt][p][1] = b0.y; av[nt][p][2] = b0.z; av[nt][p][3] = b0.w;
              av[nt][p][4] = b1.x; av[nt][p][5] = b1.y; av[nt][p][6] = b1.z; av[nt][p][7] = b1.w;
            } else {
#pragma unroll
              for (int q = 0; q < 8; ++q) if (col0 + q < NV) av[nt][p][q] = ad[q];
            }
          }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      if (GRU ? nt > 0 : !((live >> nt) & 1)) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        patch[((e & 3) + 8 * (e >> 2) + 4 * lk) * 32 + li] = acc[i][nt][e];
      __builtin_amdgcn_wave_barrier();
      const int col0 = obase + 32 * nt + pcol;        // first of the lane's 8 columns
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = mrow0 + 32 * i + prow + 16 * p;
        const float4 v0 = *reinterpret_cast<const float4 *>(patch + (prow + 16 * p) * 32 + pcol);
        const float4 v1 = *reinterpret_cast<const float4 *>(patch + (prow + 16 * p) * 32 + pcol + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (row >= M) continue;
        if (!GRU) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = (v[q] + bv[GRU ? 0 : nt][q]) + av[GRU ? 0 : nt][GRU ? 0 : p][q];
          if (a.relu) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) if (col0 + q >= NV) v[q] = 0.f;   // plane padding stays finite
          if (amaxp != nullptr) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (col0 + q < NV) { const u64 k = pg_key(v[q], col0 + q); best[p] = k > best[p] ? k : best[p]; }
          }
        }
        if (Cp != nullptr) {
          float *cp = Cp + (long long)row * ldcp + col0;
          if (col0 + 8 <= NV && (ldcp & 3) == 0) {
            *reinterpret_cast<float4 *>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) if (col0 + q < NV) cp[q] = v[q];
          }
        }
        if (Pp != nullptr && col0 < a.ldp) {          // ldp: a multiple of 32 >= NV
          uint4 h, m, l;
          pg_split8(v, h, m, l);
          unsigned short *pp = Pp + (a.ptiled ? pg_tiled_off(row, col0, a.ldp) : (long long)row * a.ldp + col0);
          *reinterpret_cast<uint4 *>(pp) = h;
          *reinterpret_cast<uint4 *>(pp + a.ppstride) = m;
          *reinterpret_cast<uint4 *>(pp + 2 * a.ppstride) = l;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (!GRU && amaxp != nullptr) {
      // a row's 128 columns of this wave sit in the 4 lanes of a quad: fold, lane 0 of the quad
      // writes the key of 128-column tile cbase / 128
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        u64 k = best[p];
        k = umax64(k, dpp_mov_u64<DPP_QUAD_1032>(k));
        k = umax64(k, dpp_mov_u64<DPP_QUAD_2301>(k));
        const int row = mrow0 + 32 * i + prow + 16 * p;
        if ((lane & 3) == 0 && row < M) amaxp[(long long)row * a.namax + (cbase >> 7)] = k;
      }
    }
  }
}

// fp32 (rows_in x K, row stride ldx) -> planes (3 x rows_out x ldp) bf16, zero beyond the matrix
__global__ __launch_bounds__(256) void planes_split_kernel(
    long long rows_in, int K, const float *__restrict__ X, long long ldx, long long rows_out,
    int ldp, unsigned short *__restrict__ P, long long pstride, int tiled) {
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
  unsigned short *pp = P + (tiled ? pg_tiled_off(r, k, ldp) : r * ldp + k);
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

template <bool GRU, int RT, int WN, int ST>
int pg_launch(const s2c_planes_gemm_args &a, hipStream_t st) {
  typedef PgGeo<RT, WN, ST> G;
  static int attr_state[64];                   // per device: 0 unknown, 1 ok, -1 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (attr_state[dev] == 0)
    attr_state[dev] = hipFuncSetAttribute((const void *)planes_gemm_kernel<GRU, RT, WN, ST>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)G::LDS) == hipSuccess ? 1 : -1;
  if (attr_state[dev] < 0) return -3;
  const int nrt = (a.M + G::BM - 1) / G::BM;
  const int nct = GRU ? (a.N + 32 * WN - 1) / (32 * WN) : (a.N + G::BN - 1) / G::BN;
  const int RTX = (nrt + 7) / 8;
  hipLaunchKernelGGL((planes_gemm_kernel<GRU, RT, WN, ST>), dim3(8 * RTX * nct), dim3(64 * G::NW), G::LDS,
                     st, a);
  return pg_chk("planes_gemm");
}

int g_pg_big = -1;       // s2c_planes_set_big: -1 auto, 0 never, 1 always (where the operands allow)

}  // namespace

extern "C" int s2c_planes_gemm(const s2c_planes_gemm_args *a, void *stream) {
  if (a == nullptr || a->M <= 0 || a->N <= 0 || a->nseg < 1 || a->nseg > 2 || a->W == nullptr ||
      (a->ldw & 31))
    return -1;
  for (int s = 0; s < a->nseg; ++s) {
    const s2c_planes_seg &g = a->seg[s];
    if (g.p == nullptr || g.kc <= 0 || (g.ld & 7)) return -1;
    // a tiled segment is read block-wise: no row map, whole 16-k blocks
    if (g.tiled && ((g.ld & 15) || g.rowmap != nullptr || g.rowdiv > 0 ||
                    (s == 0 && a->tokkeys != nullptr)))
      return -1;
  }
  if (a->nsplit != 0 && (a->gru || (a->nsplit & 127) || a->nsplit >= a->N || a->n1 <= 0 ||
                         a->n1 > a->nsplit || a->C2 == nullptr))
    return -1;
  const int nfirst = a->nsplit ? a->n1 : a->N;               // columns of the first output
  if (a->P != nullptr && ((a->ldp & 31) || a->ldp < nfirst)) return -1;
  if (a->tokkeys != nullptr && a->ntokkeys <= 0) return -1;
  if (g_pg_big == -1) g_pg_big = 1;                // 1 auto, 2 never, 3 always (s2c_planes_set_big)
  // 256 x 256 tiles where their grid still covers the chip (>= 192 workgroups); they read whole
  // 256-row / 256-column groups of blocks: the caller allocates operands to multiples of 256
  // (a->big_ok) -- models/greedy_fused.py does
  const long long tiles_big = (long long)((a->M + 255) / 256) *
                              (a->gru ? (a->N + 63) / 64 : (a->N + 255) / 256);
  const bool big = a->big_ok && (g_pg_big == 3 || (g_pg_big == 1 && tiles_big >= 192));
  if (a->gru) {
    if (a->nseg != 2 || a->bias == nullptr || a->hprev == nullptr || (a->N & 31)) return -1;
    return big ? pg_launch<true, 2, 2, 3>(*a, (hipStream_t)stream)
               : pg_launch<true, 1, 1, 3>(*a, (hipStream_t)stream);
  }
  if (a->amax != nullptr && a->namax < (nfirst + 127) / 128) return -1;
  return big ? pg_launch<false, 2, 2, 3>(*a, (hipStream_t)stream)
             : pg_launch<false, 1, 1, 3>(*a, (hipStream_t)stream);
}

// -1: by grid size (default), 0: never, 1: wherever the operands allow -- the 256 x 256 tile kernel
extern "C" void s2c_planes_set_big(int mode) { g_pg_big = mode < 0 ? 1 : (mode == 0 ? 2 : 3); }
extern "C" int s2c_planes_split(long long rows_in, int K, const float *X, long long ldx,
                                long long rows_out, int ldp, unsigned short *P,
                                long long pstride, int tiled, void *stream) {
  if (rows_out <= 0 || rows_in < 0 || rows_in > rows_out || K < 0 || (ldp & 7) || ldp < K ||
      P == nullptr || (rows_in > 0 && X == nullptr) || (tiled && ((rows_out & 31) || (ldp & 15))))
    return -1;
  const long long n = rows_out * (ldp >> 3);
  hipLaunchKernelGGL(planes_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, rows_in, K, X, ldx, rows_out, ldp, P, pstride, tiled);
  return pg_chk("planes_split");
}

// sizeof the argument structs (0: s2c_planes_gemm_args, 1: s2c_planes_seg) -- for bindings to
// check their layout
extern "C" long long s2c_planes_args_sizeof(int which) {
  return which == 0 ? (long long)sizeof(s2c_planes_gemm_args) : (long long)sizeof(s2c_planes_seg);
}
