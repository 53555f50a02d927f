// s2c_fps_small.hip -- register-resident FPS for point sets up to 8192 points
// (the SA2..SA4 and vote-aggregation stages): latency-optimised round.
//
// Per round (no global memory traffic at all after the prologue):
//   1. every thread updates the min-distances of its <= 8 register-resident
//      points and keeps its best 64-bit key ((bits(d2)+1) << 32 | ~rank, see
//      s2c_ops.hip for why this reproduces the reference's tie rule);
//   2. DPP wave arg-max; the lane that owns the wave's winner publishes
//      {key, x, y, z} to a double-buffered LDS slot;
//   3. ONE barrier; every thread reads the <= 16 slots, reduces them with a DPP
//      row arg-max and picks up the winner's coordinates from LDS -- the next
//      round's pivot never comes from HBM/L2 (the previous kernel paid a dependent
//      ~300-cycle scalar load per round for it).
// A single-wave configuration (n <= 512) needs neither LDS nor barriers.
#include "s2c_common.h"
#include "../../include/s2c_ops.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

using namespace s2c;

namespace {

__device__ __forceinline__ u32 bitrev_n(u32 v, int nbits) {
  return nbits == 0 ? 0u : (__builtin_bitreverse32(v) >> (32 - nbits));
}

struct __attribute__((aligned(16))) Slot { u64 key; float x, y, z, pad; };

// `verified` (optional): per-scene flags of the prefix verification below.  0 = the picks of
// this scene were PROVEN to be 0, 1, ..., m-1 -- they are written and the rounds never run.
template <int T, int PPT>
__global__ __launch_bounds__(T) void fps_small_kernel(int n, int m, int bs, int log2bs,
                                                      const float *__restrict__ xyz,
                                                      int *__restrict__ idx,
                                                      const int *__restrict__ verified) {
  constexpr int NW = T / 64;
  __shared__ Slot s_slot[2][NW];
  if (m <= 0) return;
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  idx += (size_t)b * m;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (verified != nullptr && verified[b] == 0) {
    for (int j = t; j < m; j += T) idx[j] = j;
    return;
  }

  float px[PPT], py[PPT], pz[PPT], mind[PPT];
  u32 nrank[PPT];  // ~rank
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = t + i * T;
    const int kk = k < n ? k : n - 1;
    const float x = xyz[kk * 3 + 0], y = xyz[kk * 3 + 1], z = xyz[kk * 3 + 2];
    const float mag = sq3(x, y, z);
    const bool skip = ((double)mag <= 1e-3) || (k >= n);   // sampling_gpu.cu:100-101
    mind[i] = skip ? -1.0f : 1e10f;
    px[i] = x; py[i] = y; pz[i] = z;
    const u32 rank = (bitrev_n((u32)kk & (u32)(bs - 1), log2bs) << 22) | ((u32)kk >> log2bs);
    nrank[i] = 0xFFFFFFFFu - rank;
  }
  // pivot of round 1 = point 0; also the fallback when nothing is selectable
  const float x0 = xyz[0], y0 = xyz[1], z0 = xyz[2];
  float cx = x0, cy = y0, cz = z0;
  if (t == 0) idx[0] = 0;

  for (int j = 1; j < m; ++j) {
    u64 best = 0ull;
    float bx = x0, by = y0, bz = z0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float d = sq3(px[i] - cx, py[i] - cy, pz[i] - cz);
      const float d2 = fminf(d, mind[i]);
      mind[i] = d2;
      const u64 key = d2 < 0.0f ? 0ull
                                : ((u64)(__float_as_uint(d2) + 1u) << 32) | (u64)nrank[i];
      const bool gt = key > best;
      best = gt ? key : best;
      bx = gt ? px[i] : bx; by = gt ? py[i] : by; bz = gt ? pz[i] : bz;
    }
    const u64 wmax = wave_max_u64(best);
    u64 key;
    if (NW == 1) {
      key = wmax;
      // winner lane is unique when key != 0 (ranks are unique)
      const u64 wl = __ballot(best == wmax && wmax != 0ull);
      if (wl) {
        const int src = (int)__builtin_ctzll(wl);
        cx = __shfl(bx, src, 64); cy = __shfl(by, src, 64); cz = __shfl(bz, src, 64);
      } else {
        cx = x0; cy = y0; cz = z0;
      }
    } else {
      Slot *slots = s_slot[j & 1];
      if (wmax == 0ull) {
        if (lane == 0) { slots[wave].key = 0ull; }
      } else if (best == wmax) {
        Slot s; s.key = wmax; s.x = bx; s.y = by; s.z = bz; s.pad = 0.f;
        slots[wave] = s;
      }
      __syncthreads();
      u64 v = lane < NW ? slots[lane].key : 0ull;
      const u64 mine = v;
      v = row16_max_u64(v);
      key = readlane_u64(v, 0);
      if (key == 0ull) {
        cx = x0; cy = y0; cz = z0;
      } else {
        const u64 wl = __ballot(lane < NW && mine == key);
        const int w = (int)__builtin_ctzll(wl);
        cx = slots[w].x; cy = slots[w].y; cz = slots[w].z;
      }
    }
    if (t == 0) {
      int old = 0;
      if ((key >> 32) != 0ull) {
        const u32 r = 0xFFFFFFFFu - (u32)key;
        old = (int)(((r & 0x3FFFFFu) << log2bs) | bitrev_n(r >> 22, log2bs));
      }
      idx[j] = old;
    }
  }
}

// ---- up to 2048 points: four waves, 32-bit reductions, ONE LDS exchange per round ------------
// The vote aggregation samples 256 of 1024 votes INSIDE the train / inference step (it cannot run
// ahead on the geometry stream), 255 strictly serial rounds.  fps_small_kernel spends a round on
// latency, not work (0.70 us at 1024 points whatever the thread count): 64-bit DPP reductions
// (two moves, a 64-bit compare and two selects per step, twice per round), an LDS slot write, the
// barrier, an LDS read, a second reduction, and ANOTHER LDS read for the winner's coordinates.
// Same selection rule here -- max over (bits(d2) + 1, ~rank), i.e. the reference's first-maximum
// tree order (sampling_gpu.cu:58-173) -- as two 32-bit passes with a DPP operand (v_max_u32 per
// step), the winner's coordinates by v_readlane, and one exchange: every wave publishes
// {hi, lo, x, y, z}, every lane reads the four slots (broadcast) and picks in registers.
// Diagnostics (PROF instance only): thread 0 of workgroup 0 adds up the cycles of a round's three
// phases in g_qprof ({own points + wave reductions, publish + barrier, read + pick, rounds}).
// First reading at 1024 points: 843 + 220 + 672 cycles -- every dependent VALU / DPP / readlane
// step costs 10-20 cycles with one wave per SIMD, so the round is its dependent-instruction count:
// the second reduction runs only when two lanes tie on the distance, and the decoding of the
// winner's index (bit reversal, shifts, a global store: 150-200 cycles in the wave every other
// wave then waits for at the barrier) is deferred to one parallel pass after the last round.
__device__ long long *g_qprof = nullptr;
#define QP_NOW() ([&]() { __builtin_amdgcn_sched_barrier(0); long long t_ = (long long)__builtin_amdgcn_s_memtime(); \
                          __builtin_amdgcn_sched_barrier(0); return t_; }())
constexpr int QUAD_MAX_M = 2048;
template <int PPT, bool PROF>
__global__ __launch_bounds__(256) void fps_quad_kernel(int n, int m, int bs, int log2bs,
                                                       const float *__restrict__ xyz,
                                                       int *__restrict__ idx,
                                                       const int *__restrict__ verified) {
  __shared__ uint4 s_a[2][4];      // {hi, lo, bits(x), bits(y)} per wave, double-buffered
  __shared__ float s_z[2][4];
  __shared__ u32 s_pick[QUAD_MAX_M];   // ~rank of every round's winner (0: none), decoded at the end
  if (m <= 0) return;
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  idx += (size_t)b * m;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (verified != nullptr && verified[b] == 0) {
    for (int j = t; j < m; j += 256) idx[j] = j;
    return;
  }
  float px[PPT], py[PPT], pz[PPT], mind[PPT];
  u32 nrank[PPT];  // ~rank
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = t + i * 256;
    const int kk = k < n ? k : n - 1;
    const float x = xyz[kk * 3 + 0], y = xyz[kk * 3 + 1], z = xyz[kk * 3 + 2];
    const float mag = sq3(x, y, z);
    const bool skip = ((double)mag <= 1e-3) || (k >= n);   // sampling_gpu.cu:100-101
    mind[i] = skip ? -1.0f : 1e10f;
    px[i] = x; py[i] = y; pz[i] = z;
    const u32 rank = (bitrev_n((u32)kk & (u32)(bs - 1), log2bs) << 22) | ((u32)kk >> log2bs);
    nrank[i] = 0xFFFFFFFFu - rank;
  }
  const float x0 = xyz[0], y0 = xyz[1], z0 = xyz[2];
  float cx = x0, cy = y0, cz = z0;
  long long *qp = g_qprof;
  const bool qon = PROF && qp != nullptr && blockIdx.x == 0 && wave == 0;
  long long qa = 0, qb = 0, qc = 0, q0 = 0, q1 = 0, q2 = 0;

  for (int j = 1; j < m; ++j) {
    if (PROF && qon) q0 = QP_NOW();
    u32 hi[PPT];
    u32 hl = 0u;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float d = sq3(px[i] - cx, py[i] - cy, pz[i] - cz);
      const float d2 = fminf(d, mind[i]);
      mind[i] = d2;
      hi[i] = d2 < 0.0f ? 0u : __float_as_uint(d2) + 1u;    // skipped points never win
      hl = hi[i] > hl ? hi[i] : hl;
    }
    const u32 mhi = wave_umax32(hl);
    // this lane's candidate among its own points: the largest ~rank at the maximum distance
    u32 ll = 0u;
    float bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const bool w = hi[i] == mhi && nrank[i] > ll;
      ll = w ? nrank[i] : ll;
      bx = w ? px[i] : bx; by = w ? py[i] : by; bz = w ? pz[i] : bz;
    }
    u64 wl = __ballot(hl == mhi);
    if (__builtin_popcountll(wl) > 1) {      // (wave-uniform, rare) lanes tie on the distance
      const u32 mlo = wave_umax32(hl == mhi ? ll : 0u);
      wl = __ballot(hl == mhi && ll == mlo);
    }
    const int src = (int)__builtin_ctzll(wl);       // (mhi == 0: every lane ties with ll = 0)
    const u32 wlo = (u32)__builtin_amdgcn_readlane((int)ll, src);
    const float wx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bx), src));
    const float wy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, by), src));
    const float wz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bz), src));
    if (PROF && qon) q1 = QP_NOW();
    if (lane == 0) {
      s_a[j & 1][wave] = make_uint4(mhi, mhi ? wlo : 0u, __float_as_uint(wx), __float_as_uint(wy));
      s_z[j & 1][wave] = wz;
    }
    __syncthreads();
    if (PROF && qon) q2 = QP_NOW();
    // (component-wise selects: a ternary on the whole uint4 went through scratch memory)
    u32 b_hi, b_lo, b_x, b_y;
    float b_z;
    {
      const uint4 c0 = s_a[j & 1][0];
      b_hi = c0.x; b_lo = c0.y; b_x = c0.z; b_y = c0.w; b_z = s_z[j & 1][0];
    }
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const uint4 c = s_a[j & 1][w];
      const float czz = s_z[j & 1][w];
      const bool gt = c.x > b_hi || (c.x == b_hi && c.y > b_lo);
      b_hi = gt ? c.x : b_hi; b_lo = gt ? c.y : b_lo;
      b_x = gt ? c.z : b_x; b_y = gt ? c.w : b_y;
      b_z = gt ? czz : b_z;
    }
    if (b_hi == 0u) {                        // nothing selectable: the reference keeps index 0
      cx = x0; cy = y0; cz = z0;
    } else {
      cx = __uint_as_float(b_x); cy = __uint_as_float(b_y); cz = b_z;
    }
    if (t == 255) s_pick[j] = b_hi != 0u ? b_lo : 0u;      // (~rank is never 0: rank < 2^31)
    if (PROF && qon) { const long long q3 = QP_NOW(); qa += q1 - q0; qb += q2 - q1; qc += q3 - q2; }
  }
  if (PROF && qon && lane == 0) { qp[0] = qa; qp[1] = qb; qp[2] = qc; qp[3] = m - 1; }
  __syncthreads();
  for (int j = t; j < m; j += 256) {
    int old = 0;
    if (j > 0 && s_pick[j] != 0u) {
      const u32 r = 0xFFFFFFFFu - s_pick[j];
      old = (int)(((r & 0x3FFFFFu) << log2bs) | bitrev_n(r >> 22, log2bs));
    }
    idx[j] = old;
  }
}

// ---- "is the answer 0, 1, ..., m-1 ?" ------------------------------------------------------
// FPS of a point set that is already in FPS pick order returns arange(m):
// SA2 samples SA1's centres, SA3 SA2's, SA4 SA3's (backbone_module.py:106-115), so three of
// the reference's five FPS calls are strictly serial round chains (sampling_gpu.cu:69-173)
// whose result is known -- unless a tie or a skipped point breaks the property.  Instead of
// trusting it, PROVE it per scene, in parallel, under the reference's exact rule:
//   picks == arange(m)  <=>  for every round j = 1..m-1 the arg-max over ALL points k of
//   key(k, D_j(k)),  D_j(k) = min_{i<j} d(p_k, p_i),  is point j (and its key is not 0),
// with d the reference's float expression, the |p|^2 <= 1e-3 skip and the 64-bit key whose
// low word carries the reference's thread-layout tie rule (same key as fps_small_kernel).
// Kernel 1: star[j] = key(j, D_j(j)).  Kernel 2: every point k walks the rounds with its
// running minimum and raises the scene's flag if its key ever beats star[j].  2 n m pair
// tests, no per-round synchronisation; the real FPS kernel then starts with "flag clear ->
// write arange, return" (no host round trip).
__device__ __forceinline__ float fps_d2(float px, float py, float pz, float cx, float cy,
                                        float cz) {
  return sq3(px - cx, py - cy, pz - cz);
}

// Both kernels: workgroup = 64 points (lane) x PFX_SEG segments of the pivot range (wave).  The
// running minimum of a point is a prefix-min over the pivots, so the rounds are cut into
// PFX_SEG segments that run in parallel: every wave first reduces ITS segment to one minimum
// per point (LDS), the prefix over the earlier segments is the state at the segment's first
// round, and the segment is walked a second time from that state with the key comparison.
// 2 x m / PFX_SEG dependent iterations per thread instead of m (one thread per point walking
// all rounds took 100 us per call: 0.25 waves per SIMD, every LDS broadcast latency exposed).
// Pivots and star keys are wave-uniform reads (scalar loads from the constant cache).
constexpr int PFX_SEG = 16;

__global__ __launch_bounds__(64 * PFX_SEG) void fps_prefix_star_kernel(
    int n, int m, int bs, int log2bs, const float *__restrict__ xyz, u64 *__restrict__ star,
    int *__restrict__ flag) {
  __shared__ float s_min[PFX_SEG][64];
  const int b = blockIdx.y;
  xyz += (size_t)b * n * 3;
  star += (size_t)b * m;
  const int lane = threadIdx.x & 63;
  const int seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = blockIdx.x * 64 + lane;
  if (blockIdx.x == 0 && threadIdx.x == 0) flag[b] = 0;   // (kernel 2 runs after this kernel)
  const int jj = j < m ? j : m - 1;
  const float x = xyz[jj * 3 + 0], y = xyz[jj * 3 + 1], z = xyz[jj * 3 + 2];
  // pivots i < j, i in this wave's share of [0, jhi): jhi = pivots the workgroup needs
  const int jhi = min(m, (int)(blockIdx.x + 1) * 64);
  const int L = (jhi + PFX_SEG - 1) / PFX_SEG;
  const int i0 = seg * L, i1 = min(jhi, i0 + L);
  float D = 1e10f;
#pragma unroll 8
  for (int i = i0; i < i1; ++i) {
    const float cx = xyz[i * 3 + 0], cy = xyz[i * 3 + 1], cz = xyz[i * 3 + 2];
    const float d = fminf(fps_d2(x, y, z, cx, cy, cz), D);
    D = i < jj ? d : D;
  }
  s_min[seg][lane] = D;
  __syncthreads();
  if (seg == 0 && j < m) {
#pragma unroll
    for (int s2 = 1; s2 < PFX_SEG; ++s2) D = fminf(D, s_min[s2][lane]);
    const float mag = sq3(x, y, z);
    const bool skip = (double)mag <= 1e-3;
    const u32 rank = (bitrev_n((u32)j & (u32)(bs - 1), log2bs) << 22) | ((u32)j >> log2bs);
    star[j] = skip ? 0ull
                   : ((u64)(__float_as_uint(D) + 1u) << 32) | (u64)(0xFFFFFFFFu - rank);
  }
}

__global__ __launch_bounds__(64 * PFX_SEG) void fps_prefix_check_kernel(
    int n, int m, int bs, int log2bs, const float *__restrict__ xyz,
    const u64 *__restrict__ star, int *__restrict__ flag) {
  __shared__ float s_min[PFX_SEG][64];
  const int b = blockIdx.y;
  xyz += (size_t)b * n * 3;
  star += (size_t)b * m;
  const int lane = threadIdx.x & 63;
  const int seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = blockIdx.x * 64 + lane;
  const int kk = k < n ? k : n - 1;
  const float x = xyz[kk * 3 + 0], y = xyz[kk * 3 + 1], z = xyz[kk * 3 + 2];
  const float mag = sq3(x, y, z);
  // a skipped point's key is 0 in every round: it can never beat star[j] (which must be > 0)
  const bool live = k < n && !((double)mag <= 1e-3);
  const u32 rank = (bitrev_n((u32)kk & (u32)(bs - 1), log2bs) << 22) | ((u32)kk >> log2bs);
  const u32 nrank = 0xFFFFFFFFu - rank;
  // round j = i + 1 uses pivot i: rounds 1 .. m-1  <=>  pivots i = 0 .. m-2
  const int L = (m - 1 + PFX_SEG - 1) / PFX_SEG;
  const int i0 = min(m - 1, seg * L), i1 = min(m - 1, i0 + L);
  float D = 1e10f;
#pragma unroll 8
  for (int i = i0; i < i1; ++i)
    D = fminf(fps_d2(x, y, z, xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2]), D);
  s_min[seg][lane] = D;
  __syncthreads();
  D = 1e10f;                                  // state at this segment's first round
  for (int s2 = 0; s2 < seg; ++s2) D = fminf(D, s_min[s2][lane]);
  bool beats = false, dead = false;
#pragma unroll 8
  for (int i = i0; i < i1; ++i) {
    const u64 st = star[i + 1];               // wave-uniform
    D = fminf(fps_d2(x, y, z, xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2]), D);
    const u64 key = ((u64)(__float_as_uint(D) + 1u) << 32) | (u64)nrank;
    // keys are unique (the rank is), so key == star[j] means k == j
    beats |= key > st;
    dead |= st == 0ull;                        // point j itself is not selectable
  }
  const bool bad = (live && beats) || dead;
  if (__ballot(bad) != 0ull && lane == 0) atomicOr(flag + b, 1);
}

__global__ void fps_prefix_set_kernel(int *flag, int b, int v) {
  for (int i = threadIdx.x; i < b; i += blockDim.x) flag[i] = v;
}

bool g_qprof_host = false;      // host mirror of g_qprof != nullptr (picks the PROF instance)

int ref_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));  // cuda_utils.h:13-19
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

}  // namespace

#define FPS_SMALL(T_, P_)                                                         \
  hipLaunchKernelGGL((fps_small_kernel<T_, P_>), dim3(b), dim3(T_), 0, st, n, m, bs, \
                     log2bs, xyz, idx, verified)

extern "C" int s2c_fps_small_limit(void) { return 8192; }

// diagnostics of fps_quad_kernel (see g_qprof); prof == NULL switches it off
extern "C" int s2c_fps_quad_set_profile(long long *prof) {
  g_qprof_host = prof != nullptr;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_qprof), &prof, sizeof(prof)) == hipSuccess ? 0 : -1;
}

// threads = 0: heuristic.  Otherwise one of 64/128/256/512/1024 (ceil(n/threads)
// must be <= 8).
static int fps_small_launch(int b, int n, int m, const float *xyz, int *idx, int threads,
                            const int *verified, hipStream_t st);

extern "C" int s2c_furthest_point_sampling_small(int b, int n, int m, const float *xyz,
                                                 int *idx, int threads,
                                                 s2c_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0 || n > 8192 || !xyz || !idx) return S2C_EINVAL;
  if (b == 0 || m == 0) return 0;
  return fps_small_launch(b, n, m, xyz, idx, threads, nullptr, (hipStream_t)stream);
}

// workspace layout: int flag[b] (padded to 16 bytes) | u64 star[b][m]
static size_t pfx_flag_bytes(int b) { return (((size_t)b * 4) + 15) & ~(size_t)15; }

extern "C" long long s2c_fps_prefix_workspace_bytes(int b, int m) {
  if (b < 0 || m < 0) return S2C_EINVAL;
  return (long long)(pfx_flag_bytes(b) + (size_t)b * m * 8);
}

// FPS for inputs that are EXPECTED to be in FPS pick order already (the centres of the
// previous set-abstraction stage): the result is identical to s2c_furthest_point_sampling for
// EVERY input -- the expectation is verified on the device per scene, scenes that fail it run
// the real rounds.  After the call workspace[0..b) (int) holds 1 for the scenes that fell back.
extern "C" int s2c_furthest_point_sampling_prefix(int b, int n, int m, const float *xyz,
                                                  void *workspace, int *idx, int threads,
                                                  s2c_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0 || n > 8192 || !xyz || !idx || !workspace) return S2C_EINVAL;
  if (b == 0 || m == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int *flag = (int *)workspace;
  if (m > n) {   // cannot be a prefix: every scene runs the rounds (flags say so)
    hipLaunchKernelGGL(fps_prefix_set_kernel, dim3(1), dim3(64), 0, st, flag, b, 1);
    return fps_small_launch(b, n, m, xyz, idx, threads, flag, st);
  }
  u64 *star = (u64 *)((char *)workspace + pfx_flag_bytes(b));
  const int bs = ref_opt_n_threads(n);
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  hipLaunchKernelGGL(fps_prefix_star_kernel, dim3((m + 63) / 64, b), dim3(64 * PFX_SEG), 0,
                     st, n, m, bs, log2bs, xyz, star, flag);
  hipLaunchKernelGGL(fps_prefix_check_kernel, dim3((n + 63) / 64, b), dim3(64 * PFX_SEG), 0,
                     st, n, m, bs, log2bs, xyz, (const u64 *)star, flag);
  return fps_small_launch(b, n, m, xyz, idx, threads, flag, st);
}

static int fps_small_launch(int b, int n, int m, const float *xyz, int *idx, int threads,
                            const int *verified, hipStream_t st) {
  const int bs = ref_opt_n_threads(n);
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  int T = threads;
  if (T == 0) {
    // the four-wave kernel with 32-bit reductions up to 2048 points (an explicit `threads` selects
    // the register-resident kernel: tests / tools)
    if (n <= 2048 && m <= QUAD_MAX_M) {
      const int ppt = (n + 255) / 256;
      const bool prof = g_qprof_host;
#define FPS_QUAD(P_)                                                                          \
  do {                                                                                        \
    if (prof) hipLaunchKernelGGL((fps_quad_kernel<P_, true>), dim3(b), dim3(256), 0, st, n, m, bs, \
                                 log2bs, xyz, idx, verified);                                 \
    else hipLaunchKernelGGL((fps_quad_kernel<P_, false>), dim3(b), dim3(256), 0, st, n, m, bs,     \
                            log2bs, xyz, idx, verified);                                      \
  } while (0)
      if (ppt <= 1) FPS_QUAD(1);
      else if (ppt <= 2) FPS_QUAD(2);
      else if (ppt <= 4) FPS_QUAD(4);
      else FPS_QUAD(8);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) {
        fprintf(stderr, "s2c: fps_quad launch failed: %s\n", hipGetErrorString(e));
        return (int)e;
      }
      return 0;
    }
    // measured on MI355X (us/round): a single wave (no LDS, no barrier) wins up to
    // 512 points (0.53), 512 threads for 1k..4k points (0.68 / 0.83), 1024 above
    T = n <= 512 ? 64 : (n <= 4096 ? 512 : 1024);
  }
  while (T < 1024 && (n + T - 1) / T > 8) T *= 2;
  const int ppt = (n + T - 1) / T;
  if (ppt > 8) return S2C_EINVAL;
#define FPS_SMALL_T(T_)                                    \
  if (ppt <= 1) FPS_SMALL(T_, 1);                          \
  else if (ppt <= 2) FPS_SMALL(T_, 2);                     \
  else if (ppt <= 4) FPS_SMALL(T_, 4);                     \
  else FPS_SMALL(T_, 8);
  switch (T) {
    case 64: FPS_SMALL_T(64) break;
    case 128: FPS_SMALL_T(128) break;
    case 256: FPS_SMALL_T(256) break;
    case 512: FPS_SMALL_T(512) break;
    case 1024: FPS_SMALL_T(1024) break;
    default: return S2C_EINVAL;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c: fps_small launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
