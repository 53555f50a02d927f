// s2c_fps_cells.hip -- exact furthest point sampling on wave-owned grid cells.
//
// Same result, bit for bit, as the brute-force kernel in s2c_ops.hip (and hence as the
// reference, sampling_gpu.cu:69-173) and as s2c_fps_bucket.hip, whose pruning rule it
// keeps: points are counting-sorted into <= 1024 grid cells with tight bounding boxes; a
// pick can lower a min-distance in a cell only if a conservative lower bound of
// dist2(pick, box) does not exceed the cell's current maximum, so a round re-evaluates
// ~15 cells (~600 points) instead of 40000, with exactly the reference's float expression.
//
// What changed is the ROUND, which is a latency chain (2047 of them per SA1 call):
// s2c_fps_bucket.hip builds a list of active cells with LDS atomics, deals the list out to
// the 16 waves and needs three workgroup barriers per round (2.1 us).  Here every cell is
// OWNED by one wave -- cell c by wave c % NW, so that the 27 neighbours of a pick spread
// over all waves -- and a wave tests, re-evaluates and keeps the keys of its own cells in
// registers without talking to anybody: no list, no atomics, and ONE barrier per round
// (the cross-wave arg-max, double-buffered like s2c_fps_small.hip).
//
//   prep    (1024 threads / scene) bounding box, grid, counting sort of {x,y,z,d2} + rank
//           records, per-cell start / count / tight box -> workspace
//   rounds  (NW waves / scene) per round and wave: bound test of its SL = 1024/(64 NW) cells
//           per lane -> ballot -> its active cells, the 16-byte point records of up to four
//           of them requested from L2 at once, each evaluated by all 64 lanes, DPP arg-max,
//           result handed to the owning lane by readlane -> arg-max over own keys (DPP) ->
//           LDS slot -> barrier -> arg-max over the NW slots, pivot coordinates ride along.
#include "s2c_common.h"
#include "../../include/s2c_ops.h"

#include <math.h>
#include <stdio.h>

using namespace s2c;

namespace {

constexpr int PT = 1024;          // prep threads
constexpr int PNW = PT / 64;
constexpr int MAXC = 1024;        // cells
constexpr int TAB_WORDS = 9;      // start, cnt, cand, box lo[3], box hi[3]

__device__ __forceinline__ u32 bitrev_n(u32 v, int nbits) {
  return nbits == 0 ? 0u : (__builtin_bitreverse32(v) >> (32 - nbits));
}
__device__ __forceinline__ u32 ord_of(float f) {
  const u32 b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_inv(u32 o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

struct __attribute__((aligned(16))) Pt { float x, y, z, d2; };
struct __attribute__((aligned(16))) Slot { u64 key; float x, y, z, pad; };

__host__ __device__ inline size_t cells_scene_bytes(int n) {
  // Pt[n] | rank[n] | table[TAB_WORDS][MAXC]
  return (((size_t)n * (sizeof(Pt) + 4) + 15) & ~(size_t)15) + (size_t)TAB_WORDS * MAXC * 4;
}

// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(PT) void fps_cells_prep_kernel(
    int n, int bs, int log2bs, int target_cells, const float *__restrict__ xyz,
    char *__restrict__ ws, size_t stride) {
  __shared__ int s_cnt[MAXC], s_cursor[MAXC], s_cand[MAXC];
  __shared__ u32 s_bb[6][MAXC];
  __shared__ u32 s_red[6][PNW];
  __shared__ float s_grid[6];
  __shared__ int s_dims[4];
  __shared__ int s_wsum[PNW];
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  ws += (size_t)b * stride;
  Pt *spt = (Pt *)ws;
  u32 *srank = (u32 *)(ws + (size_t)n * sizeof(Pt));
  u32 *tab = (u32 *)(ws + (((size_t)n * (sizeof(Pt) + 4) + 15) & ~(size_t)15));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  {  // scene bounding box
    u32 lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (int k = tid; k < n; k += PT) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const u32 o = ord_of(xyz[k * 3 + a]);
        lo[a] = min(lo[a], o);
        hi[a] = max(hi[a], o);
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        lo[a] = min(lo[a], (u32)__shfl_xor((int)lo[a], off, 64));
        hi[a] = max(hi[a], (u32)__shfl_xor((int)hi[a], off, 64));
      }
      if (lane == 0) { s_red[a][wave] = lo[a]; s_red[3 + a][wave] = hi[a]; }
    }
    s_cnt[tid] = 0; s_cand[tid] = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) { s_bb[a][tid] = 0xFFFFFFFFu; s_bb[3 + a][tid] = 0u; }
    __syncthreads();
    if (tid == 0) {
      float flo[3], fhi[3], ext[3];
      for (int a = 0; a < 3; ++a) {
        u32 l = 0xFFFFFFFFu, h = 0u;
        for (int w = 0; w < PNW; ++w) { l = min(l, s_red[a][w]); h = max(h, s_red[3 + a][w]); }
        flo[a] = ord_inv(l); fhi[a] = ord_inv(h);
        ext[a] = fmaxf(fhi[a] - flo[a], 1e-6f);
      }
      // cells of equal edge over the axes the cloud really extends along: a planar or linear
      // cloud (one extent ~0) would otherwise drive the edge towards 0 and end in the 1x1x1
      // fallback -- a single cell = a single wave re-evaluating all n points every round
      const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
      bool live[3];
      int nlive = 0;
      float vol = 1.0f;
      for (int a = 0; a < 3; ++a) {
        live[a] = ext[a] > 1e-3f * emax;
        if (live[a]) { ++nlive; vol *= ext[a]; }
      }
      int g[3] = {1, 1, 1};
      if (nlive > 0) {
        float e = nlive == 3 ? cbrtf(vol / (float)target_cells)
                             : powf(vol / (float)target_cells, 1.0f / (float)nlive);
        for (int it = 0; it < 400; ++it) {
          for (int a = 0; a < 3; ++a)
            g[a] = live[a] ? max(1, min(1024, (int)(ext[a] / e) + 1)) : 1;
          if ((long long)g[0] * g[1] * g[2] <= MAXC) break;
          e *= 1.05f;
        }
        if ((long long)g[0] * g[1] * g[2] > MAXC) { g[0] = g[1] = g[2] = 1; }
      }
      for (int a = 0; a < 3; ++a) {
        s_grid[a] = flo[a];
        s_grid[3 + a] = (float)g[a] / ext[a];
        s_dims[a] = g[a];
      }
      s_dims[3] = g[0] * g[1] * g[2];
    }
    __syncthreads();
  }
  const float glx = s_grid[0], gly = s_grid[1], glz = s_grid[2];
  const float gix = s_grid[3], giy = s_grid[4], giz = s_grid[5];
  const int gx = s_dims[0], gy = s_dims[1], gz = s_dims[2];
  auto cell_of = [&](float x, float y, float z) {
    const int ix = min(gx - 1, max(0, (int)((x - glx) * gix)));
    const int iy = min(gy - 1, max(0, (int)((y - gly) * giy)));
    const int iz = min(gz - 1, max(0, (int)((z - glz) * giz)));
    return ix + gx * (iy + gy * iz);
  };

  for (int k = tid; k < n; k += PT)
    atomicAdd(&s_cnt[cell_of(xyz[k * 3], xyz[k * 3 + 1], xyz[k * 3 + 2])], 1);
  __syncthreads();
  int my_start;
  {
    const int v = s_cnt[tid];
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
    my_start = base + inc - v;
    s_cursor[tid] = my_start;
  }
  __syncthreads();
  for (int k = tid; k < n; k += PT) {
    const float x = xyz[k * 3], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
    const int c = cell_of(x, y, z);
    const int pos = atomicAdd(&s_cursor[c], 1);
    const float mag = sq3(x, y, z);
    const bool skip = (double)mag <= 1e-3;  // sampling_gpu.cu:100-101
    Pt p; p.x = x; p.y = y; p.z = z; p.d2 = skip ? -1.0f : 1e10f;
    spt[pos] = p;
    srank[pos] = (bitrev_n((u32)k & (u32)(bs - 1), log2bs) << 22) | ((u32)k >> log2bs);
    if (!skip) {
      atomicAdd(&s_cand[c], 1);
      atomicMin(&s_bb[0][c], ord_of(x)); atomicMax(&s_bb[3][c], ord_of(x));
      atomicMin(&s_bb[1][c], ord_of(y)); atomicMax(&s_bb[4][c], ord_of(y));
      atomicMin(&s_bb[2][c], ord_of(z)); atomicMax(&s_bb[5][c], ord_of(z));
    }
  }
  __syncthreads();
  // cell table (SoA, one coalesced row per field)
  const bool has = s_cand[tid] > 0;
  tab[0 * MAXC + tid] = (u32)my_start;
  tab[1 * MAXC + tid] = (u32)s_cnt[tid];
  tab[2 * MAXC + tid] = (u32)s_cand[tid];
#pragma unroll
  for (int a = 0; a < 6; ++a)
    tab[(3 + a) * MAXC + tid] = has ? __float_as_uint(ord_inv(s_bb[a][tid])) : 0u;
}

// ---------------------------------------------------------------------------------
template <int NW, bool PROFILE = false>
__global__ __launch_bounds__(NW * 64) void fps_cells_rounds_kernel(
    int n, int m, int log2bs, const float *__restrict__ xyz, char *__restrict__ ws,
    size_t stride, int *__restrict__ idx, long long *__restrict__ prof = nullptr) {
  // PROFILE: cycle sums per phase (s_memtime) of every wave of scene 0 -> prof[wave][8]
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int SL = MAXC / (64 * NW);      // cells per lane
  __shared__ Slot s_slot[2][NW];
  if (m <= 0) return;
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  ws += (size_t)b * stride;
  idx += (size_t)b * m;
  Pt *spt = (Pt *)ws;
  const u32 *srank = (const u32 *)(ws + (size_t)n * sizeof(Pt));
  const u32 *tab = (const u32 *)(ws + (((size_t)n * (sizeof(Pt) + 4) + 15) & ~(size_t)15));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- own cells: c = wave + NW * (lane + 64 s) ------------------------------------------
  int c_start[SL], c_cnt[SL];
  bool own[SL];
  float blx[SL], bly[SL], blz[SL], bhx[SL], bhy[SL], bhz[SL];
  u64 key[SL];
  float kx[SL], ky[SL], kz[SL];
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    const int c = wave + NW * (lane + 64 * s);
    c_start[s] = (int)tab[0 * MAXC + c];
    c_cnt[s] = (int)tab[1 * MAXC + c];
    own[s] = tab[2 * MAXC + c] > 0u;        // cells with no candidate never compete
    blx[s] = __uint_as_float(tab[3 * MAXC + c]); bly[s] = __uint_as_float(tab[4 * MAXC + c]);
    blz[s] = __uint_as_float(tab[5 * MAXC + c]); bhx[s] = __uint_as_float(tab[6 * MAXC + c]);
    bhy[s] = __uint_as_float(tab[7 * MAXC + c]); bhz[s] = __uint_as_float(tab[8 * MAXC + c]);
    key[s] = own[s] ? ((u64)(__float_as_uint(1e10f) + 1u) << 32) : 0ull;
    kx[s] = 0.f; ky[s] = 0.f; kz[s] = 0.f;
  }
  // pivot coordinates travel with the arg-max through LDS; point 0 = first pivot and the
  // "nothing selectable" case (the reference's besti = 0 initialisation, :90)
  const float x0 = xyz[0], y0 = xyz[1], z0 = xyz[2];
  float px = x0, py = y0, pz = z0;
  if (tid == 0) idx[0] = 0;

  for (int j = 1; j < m; ++j) {
    const int par = j & 1;
    long long t0 = 0;
    if (PROFILE) t0 = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      // ---- which of my cells can change? --------------------------------------------------
      bool active = false;
      if (own[s]) {
        const float ddx = fmaxf(fmaxf(blx[s] - px, px - bhx[s]), 0.0f);
        const float ddy = fmaxf(fmaxf(bly[s] - py, py - bhy[s]), 0.0f);
        const float ddz = fmaxf(fmaxf(blz[s] - pz, pz - bhz[s]), 0.0f);
        const float lb = (ddx * ddx + ddy * ddy + ddz * ddz) * 0.99999f;
        const float cmax = __uint_as_float((u32)(key[s] >> 32) - 1u);
        active = !(lb > cmax);
      }
      u64 amask = __ballot(active);
      if (PROFILE) pc[4] += (long long)__builtin_popcountll(amask);
      // ---- re-evaluate them: the whole wave on one cell at a time (a ~40-point cell is ONE
      // 64-lane pass), but the point records of up to four cells are requested before the
      // first one is looked at -- one L2 round trip per group instead of one per cell, and
      // the min-distance write-backs are issued behind all the loads
      while (amask) {
        int L[4], st[4], nc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          L[r] = amask ? (int)__builtin_ctzll(amask) : -1;
          amask = amask ? (amask & (amask - 1)) : 0ull;
          const int src = L[r] < 0 ? 0 : L[r];
          st[r] = __builtin_amdgcn_readlane(c_start[s], src);
          nc[r] = L[r] < 0 ? 0 : __builtin_amdgcn_readlane(c_cnt[s], src);
        }
        Pt pa[4], pb[4];
        u32 ra[4], rb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (nc[r] > 0) {                       // wave-uniform
            const int qa = st[r] + min(lane, nc[r] - 1);
            const int qb = st[r] + min(lane + 64, nc[r] - 1);
            pa[r] = spt[qa]; ra[r] = srank[qa];
            pb[r] = spt[qb]; rb[r] = srank[qb];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (nc[r] <= 0) continue;              // wave-uniform
          u64 best = 0ull;
          float bx = 0.f, by = 0.f, bz = 0.f;
          auto visit = [&](const Pt &p, u32 rk, int q) {
            const float d = sq3(p.x - px, p.y - py, p.z - pz);
            const float d2 = fminf(d, p.d2);
            if (d2 != p.d2) spt[st[r] + q].d2 = d2;
            const u64 k = d2 < 0.0f ? 0ull
                                    : ((u64)(__float_as_uint(d2) + 1u) << 32) |
                                          (u64)(0xFFFFFFFFu - rk);
            if (k > best) { best = k; bx = p.x; by = p.y; bz = p.z; }
          };
          if (lane < nc[r]) visit(pa[r], ra[r], lane);
          if (lane + 64 < nc[r]) visit(pb[r], rb[r], lane + 64);
          for (int q = lane + 128; q < nc[r]; q += 64) {      // crowded cells (rare)
            const Pt p = spt[st[r] + q];
            visit(p, srank[st[r] + q], q);
          }
          const u64 wbest = wave_max_u64_2x32(best);
          // a cell with candidates always has a non-zero key; ranks make the winner unique
          const int src = (int)__builtin_ctzll(__ballot(best == wbest));
          const float wx = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(bx), src));
          const float wy = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(by), src));
          const float wz = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(bz), src));
          if (lane == L[r]) { key[s] = wbest; kx[s] = wx; ky[s] = wy; kz[s] = wz; }
        }
      }
    }
    // ---- arg-max over my cells, over the wave, over the workgroup -------------------------
    long long t1 = 0;
    if (PROFILE) { t1 = (long long)__builtin_amdgcn_s_memtime(); pc[0] += t1 - t0; }
    u64 best = key[0];
    float bx = kx[0], by = ky[0], bz = kz[0];
#pragma unroll
    for (int s = 1; s < SL; ++s) {
      const bool gt = key[s] > best;
      best = gt ? key[s] : best;
      bx = gt ? kx[s] : bx; by = gt ? ky[s] : by; bz = gt ? kz[s] : bz;
    }
    const u64 wmax = wave_max_u64_2x32(best);
    Slot *slots = s_slot[par];
    if (wmax == 0ull) {
      if (lane == 0) slots[wave].key = 0ull;
    } else if (best == wmax) {
      Slot sl; sl.key = wmax; sl.x = bx; sl.y = by; sl.z = bz; sl.pad = 0.f;
      slots[wave] = sl;
    }
    long long t2 = 0;
    if (PROFILE) { t2 = (long long)__builtin_amdgcn_s_memtime(); pc[1] += t2 - t1; }
    __syncthreads();
    long long t3 = 0;
    if (PROFILE) { t3 = (long long)__builtin_amdgcn_s_memtime(); pc[2] += t3 - t2; }
    u64 v = lane < NW ? slots[lane].key : 0ull;
    const u64 mine = v;
    v = row16_max_u64_2x32(v);
    const u64 gkey = readlane_u64(v, 0);
    int old = 0;
    if ((gkey >> 32) == 0ull) {
      px = x0; py = y0; pz = z0;
    } else {
      const u32 r = 0xFFFFFFFFu - (u32)gkey;
      old = (int)(((r & 0x3FFFFFu) << log2bs) | bitrev_n(r >> 22, log2bs));
      const u64 wl = __ballot(lane < NW && mine == gkey);
      const int w = (int)__builtin_ctzll(wl);
      px = slots[w].x; py = slots[w].y; pz = slots[w].z;
    }
    if (tid == 0) idx[j] = old;
    if (PROFILE) pc[3] += (long long)__builtin_amdgcn_s_memtime() - t3;
  }
  if (PROFILE && prof && b == 0 && lane == 0) {
    for (int q = 0; q < 8; ++q) prof[wave * 8 + q] = pc[q];
  }
}

// ---------------------------------------------------------------------------------
// Round 3 experiment (waves = 17; NOT the default): the same one-pick round with the arg-max
// over the wave's idle cells moved under the L2 round trip.  tools/prof_fps.py on the kernel
// above: nearly every wave has ONE active cell per round (0.93 per wave-round at N = 40000,
// volume), so a round is, for every wave, L2 round trip (~1000 cycles) -> 64-lane evaluation ->
// 64-bit wave arg-max -> 64-bit arg-max over its own 64 keys (477) -> barrier -> decode (830).
// Variants built and measured on MI355X (us per round, N = 40000 volume / surface; the kernel
// above: 1.84 / 1.66):
//  (a) four active cells at once, one per 16-lane row, one row-wise DPP arg-max for all four,
//      idle-cell arg-max meanwhile, slots read whole + coordinates by readlane: 2.23 / 2.94 --
//      with one active cell per wave a 16-lane row needs three dependent passes over a 40-point
//      cell where 64 lanes need one;
//  (b) this kernel: 64-lane evaluation as above, idle-cell arg-max between the load issue and
//      the first use (sched_barrier; without it the scheduler hoists the block above the loads),
//      re-evaluated cells merged as wave-uniform scalars, 32-bit-first arg-max, one publishing
//      lane: 1.92 / 1.87 -- the moved reduction does not leave the chain (cells phase 2400
//      cycles vs 1563 + 477), and loads made conditional on the cell size serialise the four
//      cells' round trips (a vmcnt(0) in front of every cell's loads);
//  (c) (b) with the slots read whole and the winner's coordinates by readlane: decode 1116 vs
//      832 cycles.
// Lesson kept: the round is bound by ONE L2 round trip + three dependent cross-lane reductions +
// one barrier; none of the re-orderings shortens it, and the 2047 rounds stay off the critical
// path of the train step (side stream).  Same keys, same pruning rule: bit-identical picks.

// arg-max of a u64 key over the wave; returns the (wave-uniform) maximum, `src` = its lane
// (keys are unique when non-zero; src = lowest lane when every key is 0)
__device__ __forceinline__ u64 wave_argmax_u64(u64 v, int &src) {
  const u32 hi = (u32)(v >> 32);
  const u32 mhi = wave_umax32(hi);
  u64 cand = __ballot(hi == mhi);
  if (__builtin_popcountll(cand) > 1) {
    const u32 mlo = wave_umax32(hi == mhi ? (u32)v : 0u);
    cand = __ballot(hi == mhi && (u32)v == mlo);
  }
  src = (int)__builtin_ctzll(cand);
  return readlane_u64(v, src);
}

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
}

template <bool PROFILE>
__global__ __launch_bounds__(1024) void fps_cells_rounds4_kernel(
    int n, int m, int log2bs, const float *__restrict__ xyz, char *__restrict__ ws,
    size_t stride, int *__restrict__ idx, long long *__restrict__ prof = nullptr) {
  constexpr int NW = 16;
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  __shared__ Slot s_slot[2][NW];
  if (m <= 0) return;
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  ws += (size_t)b * stride;
  idx += (size_t)b * m;
  Pt *spt = (Pt *)ws;
  const u32 *srank = (const u32 *)(ws + (size_t)n * sizeof(Pt));
  const u32 *tab = (const u32 *)(ws + (((size_t)n * (sizeof(Pt) + 4) + 15) & ~(size_t)15));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // own cell: c = wave + 16 * lane
  const int c = wave + NW * lane;
  const int c_start = (int)tab[0 * MAXC + c], c_cnt = (int)tab[1 * MAXC + c];
  const bool own = tab[2 * MAXC + c] > 0u;       // cells with no candidate never compete
  const float blx = __uint_as_float(tab[3 * MAXC + c]), bly = __uint_as_float(tab[4 * MAXC + c]);
  const float blz = __uint_as_float(tab[5 * MAXC + c]), bhx = __uint_as_float(tab[6 * MAXC + c]);
  const float bhy = __uint_as_float(tab[7 * MAXC + c]), bhz = __uint_as_float(tab[8 * MAXC + c]);
  u64 key = own ? ((u64)(__float_as_uint(1e10f) + 1u) << 32) : 0ull;
  float kx = 0.f, ky = 0.f, kz = 0.f;

  const float x0 = xyz[0], y0 = xyz[1], z0 = xyz[2];
  float px = x0, py = y0, pz = z0;
  if (tid == 0) idx[0] = 0;

  for (int j = 1; j < m; ++j) {
    const int par = j & 1;
    long long t0 = 0;
    if (PROFILE) t0 = (long long)__builtin_amdgcn_s_memtime();
    // ---- which of my cells can change? ----------------------------------------------------
    bool active = false;
    if (own) {
      const float ddx = fmaxf(fmaxf(blx - px, px - bhx), 0.0f);
      const float ddy = fmaxf(fmaxf(bly - py, py - bhy), 0.0f);
      const float ddz = fmaxf(fmaxf(blz - pz, pz - bhz), 0.0f);
      const float lb = (ddx * ddx + ddy * ddy + ddz * ddz) * 0.99999f;
      const float cmax = __uint_as_float((u32)(key >> 32) - 1u);
      active = !(lb > cmax);
    }
    u64 amask = __ballot(active);
    if (PROFILE) pc[4] += (long long)__builtin_popcountll(amask);
    // wave-uniform best of this round: the idle cells' arg-max, then every re-evaluated cell
    u64 bk = 0ull;
    float bx = x0, by = y0, bz = z0;
    bool first = true;
    do {
      // ---- up to four active cells: request their records (one L2 round trip) --------------
      int L[4], st[4], nc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        L[r] = amask ? (int)__builtin_ctzll(amask) : -1;
        amask = amask ? (amask & (amask - 1)) : 0ull;
        const int src = L[r] < 0 ? 0 : L[r];
        st[r] = __builtin_amdgcn_readlane(c_start, src);
        nc[r] = L[r] < 0 ? 0 : __builtin_amdgcn_readlane(c_cnt, src);
      }
      Pt pa[4], pb[4];
      u32 ra[4], rb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (nc[r] > 0) {                       // wave-uniform
          const int qa = st[r] + min(lane, nc[r] - 1);
          pa[r] = spt[qa]; ra[r] = srank[qa];
          if (nc[r] > 64) {
            const int qb = st[r] + min(lane + 64, nc[r] - 1);
            pb[r] = spt[qb]; rb[r] = srank[qb];
          }
        }
      }
      // ---- meanwhile: arg-max over the cells that keep their key this round ---------------
      // (sched_barrier: the machine scheduler otherwise hoists this load-independent block
      // above the loads -- measured: no overlap, +240 cycles per round)
      __builtin_amdgcn_sched_barrier(0);
      if (first) {
        first = false;
        int src;
        const u64 ik = wave_argmax_u64(active ? 0ull : key, src);
        if (ik != 0ull) {
          bk = ik;
          bx = readlane_f(kx, src); by = readlane_f(ky, src); bz = readlane_f(kz, src);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (nc[r] <= 0) continue;              // wave-uniform
        u64 best = 0ull;
        float cx = 0.f, cy = 0.f, cz = 0.f;
        auto visit = [&](const Pt &p, u32 rk, int q) {
          const float d = sq3(p.x - px, p.y - py, p.z - pz);
          const float d2 = fminf(d, p.d2);
          if (d2 != p.d2) spt[st[r] + q].d2 = d2;
          const u64 k = d2 < 0.0f ? 0ull
                                  : ((u64)(__float_as_uint(d2) + 1u) << 32) |
                                        (u64)(0xFFFFFFFFu - rk);
          if (k > best) { best = k; cx = p.x; cy = p.y; cz = p.z; }
        };
        if (lane < nc[r]) visit(pa[r], ra[r], lane);
        if (nc[r] > 64) {
          if (lane + 64 < nc[r]) visit(pb[r], rb[r], lane + 64);
          for (int q = lane + 128; q < nc[r]; q += 64) {      // crowded cells (rare)
            const Pt p = spt[st[r] + q];
            visit(p, srank[st[r] + q], q);
          }
        }
        // a cell with candidates always has a non-zero key; ranks make the winner unique
        int src;
        const u64 wbest = wave_argmax_u64(best, src);
        const float wx = readlane_f(cx, src), wy = readlane_f(cy, src), wz = readlane_f(cz, src);
        if (lane == L[r]) { key = wbest; kx = wx; ky = wy; kz = wz; }
        if (wbest > bk) { bk = wbest; bx = wx; by = wy; bz = wz; }
      }
    } while (amask);
    long long t1 = 0;
    if (PROFILE) { t1 = (long long)__builtin_amdgcn_s_memtime(); pc[0] += t1 - t0; }
    // ---- publish (uniform values: one lane writes) -------------------------------------------
    Slot *slots = s_slot[par];
    if (lane == 0) {
      Slot sl; sl.key = bk; sl.x = bx; sl.y = by; sl.z = bz; sl.pad = 0.f;
      slots[wave] = sl;
    }
    long long t2 = 0;
    if (PROFILE) { t2 = (long long)__builtin_amdgcn_s_memtime(); pc[1] += t2 - t1; }
    __syncthreads();
    long long t3 = 0;
    if (PROFILE) { t3 = (long long)__builtin_amdgcn_s_memtime(); pc[2] += t3 - t2; }
    // ---- decode (as in the kernel above: it measured faster than reading the slots whole and
    // taking the coordinates by readlane, 832 vs 1116 cycles) ---------------------------------
    u64 v = lane < NW ? slots[lane].key : 0ull;
    const u64 mine = v;
    v = row16_max_u64_2x32(v);
    const u64 gkey = readlane_u64(v, 0);
    int old = 0;
    if ((gkey >> 32) == 0ull) {
      px = x0; py = y0; pz = z0;
    } else {
      const u32 r = 0xFFFFFFFFu - (u32)gkey;
      old = (int)(((r & 0x3FFFFFu) << log2bs) | bitrev_n(r >> 22, log2bs));
      const u64 wl = __ballot(lane < NW && mine == gkey);
      const int w = (int)__builtin_ctzll(wl);
      px = slots[w].x; py = slots[w].y; pz = slots[w].z;
    }
    if (tid == 0) idx[j] = old;
    if (PROFILE) pc[3] += (long long)__builtin_amdgcn_s_memtime() - t3;
  }
  if (PROFILE && prof && b == 0 && lane == 0) {
    for (int q = 0; q < 8; ++q) prof[wave * 8 + q] = pc[q];
  }
}

// ---------------------------------------------------------------------------------
// Several exact picks per round.
//
// A round of the kernel above is ~4000 cycles of dependent latency (L2 round trip for the
// active cells, three cross-lane reductions, one barrier) for ONE pick.  But the arg-max the
// round computes anyway usually determines the next few picks too.  Let c_1 > c_2 > ... be the
// 16 waves' best points (keys are unique: they carry the tie-break rank) and theta the
// largest key of any OTHER point (per wave: its second-best cell key and the runner-up
// inside its best cell).  c_1 is this round's pick.  Then c_t (t > 1) is exactly the t-th
// pick iff (a) key(c_t) > theta and (b) no pick accepted before it in this round lowers its
// min-distance, i.e. !(dist2(c_s, c_t) < d2(c_t)) for s < t with the reference's float
// expression: every other point's key can only fall, c_t's does not, so c_t is the maximum
// the reference's next round would find, tie rule included.  The min-distance updates of
// the accepted picks are applied together (min is associative: the same values as the
// reference's one-pick rounds).  Simulated and measured: 3.7-3.9 picks per round at N=40000.
// NOT the default (see the launcher): bit-exact, 3.9x fewer rounds, but each round is 4x as
// expensive -- the scene's single CU is instruction-issue bound once 16 waves each run the
// 8-pick update and the candidate selection (tools/prof_fps.py).
struct __attribute__((aligned(16))) MSlot { u64 key; float x, y, z; int pad; u64 bound; u64 pad2; };

// MAXP: most picks accepted per round (8: 3.9 per round on average but ~57 active cells to update;
// 2-3: fewer picks, a round close to the one-pick round's cost)
template <bool LDSD2, int MAXP>
__global__ __launch_bounds__(1024) void fps_cells_multi_kernel(
    int n, int m, int log2bs, const float *__restrict__ xyz, char *__restrict__ ws,
    size_t stride, int *__restrict__ idx, long long *__restrict__ prof = nullptr) {
  // LDSD2 (n <= ~40000): the running min-distances live in LDS (4 n bytes of the 160 KB) and
  // the sorted records carry the tie-break rank in their 4th word: ONE 16-byte read-only load
  // per point and no global store in the round loop at all -- vmcnt counts stores too on
  // CDNA4, so a min-distance write-back in front of the next group's loads put the store's
  // acknowledgement on the critical path of every group.
  constexpr int NW = 16;
  long long n_cells = 0, n_rounds = 0, t_upd = 0, t_cand = 0;
  __shared__ MSlot s_slot[2][NW];
  extern __shared__ __attribute__((aligned(16))) float s_d2[];
  if (m <= 0) return;
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  ws += (size_t)b * stride;
  idx += (size_t)b * m;
  Pt *spt = (Pt *)ws;
  const u32 *srank = (const u32 *)(ws + (size_t)n * sizeof(Pt));
  const u32 *tab = (const u32 *)(ws + (((size_t)n * (sizeof(Pt) + 4) + 15) & ~(size_t)15));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // own cell: c = wave + 16 * lane
  const int c = wave + NW * lane;
  const int c_start = (int)tab[0 * MAXC + c], c_cnt = (int)tab[1 * MAXC + c];
  const bool own = tab[2 * MAXC + c] > 0u;
  const float blx = __uint_as_float(tab[3 * MAXC + c]), bly = __uint_as_float(tab[4 * MAXC + c]);
  const float blz = __uint_as_float(tab[5 * MAXC + c]), bhx = __uint_as_float(tab[6 * MAXC + c]);
  const float bhy = __uint_as_float(tab[7 * MAXC + c]), bhz = __uint_as_float(tab[8 * MAXC + c]);
  u64 key1 = own ? ((u64)(__float_as_uint(1e10f) + 1u) << 32) : 0ull, key2 = key1;
  float kx = 0.f, ky = 0.f, kz = 0.f;

  if (LDSD2) {
    for (int q = tid; q < n; q += 1024) s_d2[q] = spt[q].d2;
    __syncthreads();
  }
  const float x0 = xyz[0], y0 = xyz[1], z0 = xyz[2];
  // accepted picks of the current round live in lanes 0 .. P-1 of every wave
  float ax = x0, ay = y0, az = z0;
  int P = 1;                        // round 0: the reference's first pick, point 0
  if (tid == 0) idx[0] = 0;
  int j = 1;                        // picks written so far

  for (int round = 0;; ++round) {
    const int par = round & 1;
    // ---- update: which of my cells can any of the P picks change? ------------------------
    bool active = false;
    if (own) {
      const float cmax = __uint_as_float((u32)(key1 >> 32) - 1u);
      for (int s = 0; s < P; ++s) {
        const float px = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(ax), s));
        const float py = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(ay), s));
        const float pz = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(az), s));
        const float ddx = fmaxf(fmaxf(blx - px, px - bhx), 0.0f);
        const float ddy = fmaxf(fmaxf(bly - py, py - bhy), 0.0f);
        const float ddz = fmaxf(fmaxf(blz - pz, pz - bhz), 0.0f);
        const float lb = (ddx * ddx + ddy * ddy + ddz * ddz) * 0.99999f;
        active = active || !(lb > cmax);
      }
    }
    u64 amask = __ballot(active);
    long long tp0 = 0;
    if (prof) { n_cells += __builtin_popcountll(amask); ++n_rounds; tp0 = (long long)__builtin_amdgcn_s_memtime(); }
    // four active cells at a time, one per 16-lane row: three 16-point chunks of each cell are
    // requested before the first is looked at (one L2 round trip per group of four cells), the
    // per-cell arg-max and runner-up are DPP row reductions (no cross-row traffic)
    const int row = lane >> 4, rl = lane & 15;
    while (amask) {
      int L[4], st = 0, nc = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        L[r] = amask ? (int)__builtin_ctzll(amask) : -1;
        amask = amask ? (amask & (amask - 1)) : 0ull;
        const int src = L[r] < 0 ? 0 : L[r];
        const int a = __builtin_amdgcn_readlane(c_start, src);
        const int q = __builtin_amdgcn_readlane(c_cnt, src);
        if (row == r) { st = a; nc = L[r] < 0 ? 0 : q; }
      }
      Pt pp[3];
      u32 rr[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int q = st + min(rl + 16 * u, max(nc - 1, 0));
        pp[u] = spt[q];
        rr[u] = srank[q];
        if (LDSD2) pp[u].d2 = s_d2[q];
      }
      u64 b1 = 0ull, b2 = 0ull;          // this lane's best and second-best key
      float bx = 0.f, by = 0.f, bz = 0.f;
      auto visit = [&](const Pt &p, u32 rk, int q) {
        float d2 = p.d2;
        for (int s = 0; s < P; ++s) {
          const float px = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(ax), s));
          const float py = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(ay), s));
          const float pz = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(az), s));
          const float d = sq3(p.x - px, p.y - py, p.z - pz);
          d2 = fminf(d, d2);
        }
        if (d2 != p.d2) {
          if (LDSD2) s_d2[st + q] = d2; else spt[st + q].d2 = d2;
        }
        const u64 k = d2 < 0.0f ? 0ull
                                : ((u64)(__float_as_uint(d2) + 1u) << 32) |
                                      (u64)(0xFFFFFFFFu - rk);
        if (k > b1) { b2 = b1; b1 = k; bx = p.x; by = p.y; bz = p.z; }
        else if (k > b2) { b2 = k; }
      };
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (rl + 16 * u < nc) visit(pp[u], rr[u], rl + 16 * u);
      for (int q = rl + 48; q < nc; q += 16) {        // crowded cells
        Pt p = spt[st + q];
        if (LDSD2) p.d2 = s_d2[st + q];
        visit(p, srank[st + q], q);
      }
      const u64 w1 = row16_max_u64_2x32(b1);
      const bool win = (b1 == w1) && (w1 != 0ull);
      const u64 w2 = row16_max_u64_2x32(win ? b2 : b1);
      const u64 winners = __ballot(win);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (L[r] < 0) continue;                         // wave-uniform
        const u64 wr = (winners >> (16 * r)) & 0xFFFFull;
        // a cell with candidates always yields a non-zero key; ranks make the winner unique
        const int src = 16 * r + (int)__builtin_ctzll(wr | 0x10000ull);
        const u64 k1 = readlane_u64(w1, 16 * r), k2 = readlane_u64(w2, 16 * r);
        const float wx = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(bx), src & 63));
        const float wy = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(by), src & 63));
        const float wz = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(bz), src & 63));
        if (lane == L[r]) { key1 = k1; key2 = k2; kx = wx; ky = wy; kz = wz; }
      }
    }
    if (j >= m) break;
    long long tp1 = 0;
    if (prof) { tp1 = (long long)__builtin_amdgcn_s_memtime(); t_upd += tp1 - tp0; }

    // ---- candidates: my wave's best point, and a bound on all its other points ------------
    const u64 wmax = wave_max_u64_2x32(key1);
    MSlot *slots = s_slot[par];
    if (wmax == 0ull) {
      if (lane == 0) { slots[wave].key = 0ull; slots[wave].bound = 0ull; }
    } else {
      const int src = (int)__builtin_ctzll(__ballot(key1 == wmax));
      const u64 bound = wave_max_u64_2x32(lane == src ? key2 : key1);
      if (lane == src) {
        MSlot sl; sl.key = wmax; sl.x = kx; sl.y = ky; sl.z = kz; sl.pad = 0; sl.bound = bound;
        sl.pad2 = 0ull;
        slots[wave] = sl;
      }
    }
    __syncthreads();
    // ---- every wave derives the same pick sequence from the 16 slots ----------------------
    u64 ckey = 0ull, cbound = 0ull;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (lane < NW) {
      const MSlot sl = slots[lane];
      ckey = sl.key; cbound = sl.bound; cx = sl.x; cy = sl.y; cz = sl.z;
    }
    const u64 theta = readlane_u64(row16_max_u64_2x32(cbound), 0);
    int rank = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) rank += (readlane_u64(ckey, q) > ckey) ? 1 : 0;
    const int room = min(MAXP, m - j);
    u64 akey = 0ull;                 // lane s: key of accepted pick s
    P = 0;
    for (int t = 0; t < NW && P < room; ++t) {
      const u64 who = __ballot(lane < NW && rank == t && ckey != 0ull);
      if (who == 0ull) break;
      const int src = (int)__builtin_ctzll(who);
      const u64 kt = readlane_u64(ckey, src);
      // a candidate with d2 == 0 (key >> 32 == 1) is an already-picked point: later picks of
      // this round (whose keys fall to d2 == 0 too) need not stay below it
      if (t > 0 && (!(kt > theta) || (kt >> 32) <= 1ull)) break;
      const float px = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(cx), src));
      const float py = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(cy), src));
      const float pz = __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(cz), src));
      const float d2t = __uint_as_float((u32)(kt >> 32) - 1u);
      // would an earlier pick of this round lower its min-distance?  (x2 = candidate, x1 = pick)
      const float d = sq3(px - ax, py - ay, pz - az);
      if (__ballot(lane < P && d < d2t) != 0ull) break;
      if (lane == P) { ax = px; ay = py; az = pz; akey = kt; }
      ++P;
      if ((kt >> 32) <= 1ull) {
        // the maximum has d2 == 0: picking it changes nothing, so the reference picks this
        // same point in every following round -- fill the rest of the round with it
        if (lane >= 1 && lane < room) { ax = px; ay = py; az = pz; akey = kt; }
        P = room;
        break;
      }
    }
    if (P == 0) {                    // nothing selectable: the reference's besti = 0 (:90)
      if (lane == 0) { ax = x0; ay = y0; az = z0; akey = 0ull; }
      P = 1;
    }
    if (wave == 0 && lane < P) {
      int old = 0;
      if ((akey >> 32) != 0ull) {
        const u32 r = 0xFFFFFFFFu - (u32)akey;
        old = (int)(((r & 0x3FFFFFu) << log2bs) | bitrev_n(r >> 22, log2bs));
      }
      idx[j + lane] = old;
    }
    j += P;
    if (prof) t_cand += (long long)__builtin_amdgcn_s_memtime() - tp1;
    if (j >= m) break;               // the last picks need no update
  }
  if (prof && b == 0 && lane == 0) {
    prof[wave * 8 + 0] = n_rounds; prof[wave * 8 + 1] = n_cells;
    prof[wave * 8 + 2] = t_upd; prof[wave * 8 + 3] = t_cand;
  }
}

// n * 4 bytes of min-distances + the static slots must fit the 160 KB of one CU
template <int MAXP>
void launch_multi_p(int b, int n, int m, int log2bs, const float *xyz, void *workspace,
                    size_t stride, int *idx, long long *prof, hipStream_t st) {
  const size_t lds = (size_t)n * 4;
  if (lds + 4096 <= 160 * 1024) {
    static bool attr_dev[64] = {};             // per device (the attribute is the device's)
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_dev[dev]) {
      (void)hipFuncSetAttribute((const void *)fps_cells_multi_kernel<true, MAXP>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
      attr_dev[dev] = true;
    }
    hipLaunchKernelGGL((fps_cells_multi_kernel<true, MAXP>), dim3(b), dim3(1024), lds, st, n, m,
                       log2bs, xyz, (char *)workspace, stride, idx, prof);
  } else {
    hipLaunchKernelGGL((fps_cells_multi_kernel<false, MAXP>), dim3(b), dim3(1024), 0, st, n, m,
                       log2bs, xyz, (char *)workspace, stride, idx, prof);
  }
}

// n * 4 bytes of min-distances + the static slots must fit the 160 KB of one CU.
// waves = -16: up to 8 picks per round; -2 / -3 / -4: up to 2 / 3 / 4
void launch_multi(int b, int n, int m, int log2bs, const float *xyz, void *workspace,
                  size_t stride, int *idx, long long *prof, hipStream_t st, int waves = -16) {
  switch (waves) {
    case -2: launch_multi_p<2>(b, n, m, log2bs, xyz, workspace, stride, idx, prof, st); break;
    case -3: launch_multi_p<3>(b, n, m, log2bs, xyz, workspace, stride, idx, prof, st); break;
    case -4: launch_multi_p<4>(b, n, m, log2bs, xyz, workspace, stride, idx, prof, st); break;
    default: launch_multi_p<8>(b, n, m, log2bs, xyz, workspace, stride, idx, prof, st); break;
  }
}

}  // namespace

// Diagnostics: the 16-wave rounds kernel with per-phase cycle counters of scene 0
// (prof[16][8]: cells phase, own arg-max, barrier wait, decode, active cells; device memory).
extern "C" int s2c_fps_cells_profile(int b, int n, int m, const float *xyz, void *workspace,
                                     int *idx, int waves, long long *prof, s2c_stream_t stream) {
  const int pow_2 = (int)(log((double)n) / log(2.0));
  int bs = 1 << pow_2;
  if (bs > 512) bs = 512;
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  int target = n / 40;
  if (target > MAXC) target = MAXC;
  hipStream_t st = (hipStream_t)stream;
  const size_t stride = cells_scene_bytes(n);
  hipLaunchKernelGGL(fps_cells_prep_kernel, dim3(b), dim3(PT), 0, st, n, bs, log2bs, target,
                     xyz, (char *)workspace, stride);
  if (waves == -16 || (waves <= -2 && waves >= -4))
    launch_multi(b, n, m, log2bs, xyz, workspace, stride, idx, prof, st, waves);
  else if (waves == 17)
    hipLaunchKernelGGL((fps_cells_rounds4_kernel<true>), dim3(b), dim3(1024), 0, st, n, m,
                       log2bs, xyz, (char *)workspace, stride, idx, prof);
  else if (waves == 4)
    hipLaunchKernelGGL((fps_cells_rounds_kernel<4, true>), dim3(b), dim3(256), 0, st, n, m,
                       log2bs, xyz, (char *)workspace, stride, idx, prof);
  else if (waves == 8)
    hipLaunchKernelGGL((fps_cells_rounds_kernel<8, true>), dim3(b), dim3(512), 0, st, n, m,
                       log2bs, xyz, (char *)workspace, stride, idx, prof);
  else
    hipLaunchKernelGGL((fps_cells_rounds_kernel<16, true>), dim3(b), dim3(1024), 0, st, n, m,
                       log2bs, xyz, (char *)workspace, stride, idx, prof);
  return (int)hipGetLastError();
}

static thread_local char g_err5[256] = "";
extern "C" const char *s2c_fps_cells_last_error_string(void) { return g_err5; }

extern "C" long long s2c_fps_cells_workspace_bytes(int b, int n) {
  if (b <= 0 || n <= 0) return 16;
  return (long long)b * (long long)cells_scene_bytes(n);
}

// Same contract as s2c_furthest_point_sampling_bucketed; `waves` = workgroup size of the
// rounds kernel in waves (4, 8 or 16; 0 = library default).
extern "C" int s2c_furthest_point_sampling_cells(int b, int n, int m, const float *xyz,
                                                 void *workspace, int *idx, int waves,
                                                 s2c_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0 || !xyz || !idx || !workspace ||
      ((uintptr_t)workspace & 15)) {
    snprintf(g_err5, sizeof(g_err5), "s2c: fps_cells: invalid argument");
    return S2C_EINVAL;
  }
  if (b == 0 || m == 0) return 0;
  const int pow_2 = (int)(log((double)n) / log(2.0));  // cuda_utils.h:13-19
  int bs = 1 << pow_2;
  if (bs > 512) bs = 512;
  if (bs < 1) bs = 1;
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  int target = n / 40;
  if (target > MAXC) target = MAXC;
  if (target < 1) target = 1;
  // 0: default = one pick per round on 16 waves (1.84 us/round at N = 40000; 8 waves 2.23,
  // 4 waves 3.27); -16 = several exact picks per round: 3.9 picks per round, but a round then
  // costs 19k cycles instead of 4.7k -- one CU issues every instruction of the scene, and the
  // 16 waves x ~1500 instructions of a multi-pick round are issue-bound (2.04 us per pick)
  // 17 = the round-3 variant (idle-cell arg-max under the load latency, scalar merge, one
  // publishing lane): bit-exact, but 1.92 vs 1.84 us per round -- NOT the default (see above)
  if (waves == 0) waves = 16;
  hipStream_t st = (hipStream_t)stream;
  const size_t stride = cells_scene_bytes(n);
  hipLaunchKernelGGL(fps_cells_prep_kernel, dim3(b), dim3(PT), 0, st, n, bs, log2bs, target,
                     xyz, (char *)workspace, stride);
  switch (waves) {
    case 4:
      hipLaunchKernelGGL((fps_cells_rounds_kernel<4>), dim3(b), dim3(256), 0, st, n, m, log2bs,
                         xyz, (char *)workspace, stride, idx, nullptr);
      break;
    case 8:
      hipLaunchKernelGGL((fps_cells_rounds_kernel<8>), dim3(b), dim3(512), 0, st, n, m, log2bs,
                         xyz, (char *)workspace, stride, idx, nullptr);
      break;
    case 16:
      hipLaunchKernelGGL((fps_cells_rounds_kernel<16>), dim3(b), dim3(1024), 0, st, n, m,
                         log2bs, xyz, (char *)workspace, stride, idx, nullptr);
      break;
    case 17:
      hipLaunchKernelGGL((fps_cells_rounds4_kernel<false>), dim3(b), dim3(1024), 0, st, n, m,
                         log2bs, xyz, (char *)workspace, stride, idx, nullptr);
      break;
    case -16: case -2: case -3: case -4:
      launch_multi(b, n, m, log2bs, xyz, workspace, stride, idx, nullptr, st, waves);
      break;
    default:
      snprintf(g_err5, sizeof(g_err5), "s2c: fps_cells: waves must be 4, 8, 16, 17 or -16");
      return S2C_EINVAL;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err5, sizeof(g_err5), "s2c: fps_cells launch failed: %s",
             hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
