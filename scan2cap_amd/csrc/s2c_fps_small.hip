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

template <int T, int PPT>
__global__ __launch_bounds__(T) void fps_small_kernel(int n, int m, int bs, int log2bs,
                                                      const float *__restrict__ xyz,
                                                      int *__restrict__ idx) {
  constexpr int NW = T / 64;
  __shared__ Slot s_slot[2][NW];
  if (m <= 0) return;
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  idx += (size_t)b * m;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  float px[PPT], py[PPT], pz[PPT], mind[PPT];
  u32 nrank[PPT];  // ~rank
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = t + i * T;
    const int kk = k < n ? k : n - 1;
    const float x = xyz[kk * 3 + 0], y = xyz[kk * 3 + 1], z = xyz[kk * 3 + 2];
    const float mag = (x * x) + (y * y) + (z * z);
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
      const float d = (px[i] - cx) * (px[i] - cx) + (py[i] - cy) * (py[i] - cy) +
                      (pz[i] - cz) * (pz[i] - cz);
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
                     log2bs, xyz, idx)

extern "C" int s2c_fps_small_limit(void) { return 8192; }

// threads = 0: heuristic.  Otherwise one of 64/128/256/512/1024 (ceil(n/threads)
// must be <= 8).
extern "C" int s2c_furthest_point_sampling_small(int b, int n, int m, const float *xyz,
                                                 int *idx, int threads,
                                                 s2c_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0 || n > 8192 || !xyz || !idx) return S2C_EINVAL;
  if (b == 0 || m == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int bs = ref_opt_n_threads(n);
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  int T = threads;
  if (T == 0) {
    const char *e = getenv("S2C_FPS_T");
    if (e) T = atoi(e);
  }
  if (T == 0) {
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
