// s2c_ops.hip -- gfx950 (MI355X) kernels + C-ABI launchers for the nine
// point-cloud operators of the Scan2Cap hot path (include/s2c_ops.h).
//
// Written for CDNA4: 64-lane wavefronts, DPP/ballot wave primitives, LDS-staged
// point tiles, grids sized to fill 256 CUs (the reference launches one block per
// scene, e.g. ball_query_gpu.cu:50).  Semantics (tie rules, padding, skip rule)
// follow the reference kernels cited at each kernel; results are bit-exact with
// oracle/s2c_oracle.c under the canonical arithmetic (no FMA contraction).
#include "s2c_common.h"
#include "../../include/s2c_ops.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

using namespace s2c;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_err[256] = "";

static int fail_args(const char *what) {
  snprintf(g_err, sizeof(g_err), "s2c: invalid argument: %s", what);
  return S2C_EINVAL;
}

static int check_launch(const char *kernel) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "s2c: %s launch failed: %s", kernel,
             hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int s2c_abi_version(void) { return S2C_ABI_VERSION; }
extern "C" const char *s2c_last_error_string(void) { return g_err; }

static inline unsigned cdiv(long long a, long long b) {
  return (unsigned)((a + b - 1) / b);
}

// ===========================================================================
// 1. Furthest point sampling  (sampling_gpu.cu:59-173)
// ===========================================================================
// One workgroup per scene (the m-1 rounds are inherently serial).  Thread t owns
// points k = t + i*T; their running min-distances live in registers for the whole
// kernel (the reference round-trips a global `temp` array every round,
// sampling_gpu.cu:106-107).  Each round: per-thread scan -> DPP wave arg-max ->
// one LDS exchange (double-buffered, ONE barrier per round; the reference needs
// ten) -> uniform decode of the winner.
//
// Exact emulation of the reference's winner among bit-equal maxima: the
// reference block has bs = opt_n_threads(n) threads; thread (k mod bs) scans its
// points in ascending k with a strict '>' (earliest k wins), and the shared
// memory tree keeps the LOWER slot on ties at strides bs/2 .. 1
// (__update, :59-65) -- i.e. the winner is the candidate whose thread id has the
// smallest bit-reversal.  Both rules collapse into a static per-point priority
//     rank(k) = bitrev_log2(bs)(k mod bs) << 22  |  (k div bs)      (smaller wins)
// so a single unsigned 64-bit max over  (bits(d2)+1) << 32 | ~rank  reproduces
// the reference for ANY thread geometry.  d2 >= 0 makes its bit pattern
// monotone; key 0 means "no candidate" (all points skipped) and decodes to
// index 0, as the reference's besti = 0 initialisation does (:90).
// Points with |p|^2 <= 1e-3 are skipped (:100-101) by pinning their min-distance
// to -1, which can never beat best = -1 under the strict compare.
__device__ __forceinline__ u32 bitrev_n(u32 v, int nbits) {
  return nbits == 0 ? 0u : (__builtin_bitreverse32(v) >> (32 - nbits));
}

template <int T, int PPT, bool XYZ_IN_REGS>
__global__ __launch_bounds__(T) void fps_kernel(int n, int m, int bs,
                                                int log2bs,
                                                const float *__restrict__ xyz,
                                                int *__restrict__ idx) {
  constexpr int NW = T / kWave;
  __shared__ u64 s_key[2][NW > 1 ? NW : 1];
  if (m <= 0) return;
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  idx += (size_t)b * m;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;

  float mind[PPT];
  float px[XYZ_IN_REGS ? PPT : 1], py[XYZ_IN_REGS ? PPT : 1],
      pz[XYZ_IN_REGS ? PPT : 1];
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(xyz), 0, n * 12, 0x00020000);
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = t + i * T;
    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, t * 12, i * T * 12, 0);
    const float x = __uint_as_float(v.x), y = __uint_as_float(v.y),
                z = __uint_as_float(v.z);
    const float mag = sq3(x, y, z);
    const bool skip = ((double)mag <= 1e-3) || (k >= n);
    mind[i] = skip ? -1.0f : 1e10f;
    if (XYZ_IN_REGS) { px[i] = x; py[i] = y; pz[i] = z; }
    if (!XYZ_IN_REGS && (i % 4) == 3) __builtin_amdgcn_sched_barrier(0);
  }

  int old = 0;
  if (t == 0) idx[0] = 0;

  for (int j = 1; j < m; ++j) {
    const int so = __builtin_amdgcn_readfirstlane(old);
    const float x1 = xyz[so * 3 + 0], y1 = xyz[so * 3 + 1],
                z1 = xyz[so * 3 + 2];
    float best = -1.0f;
    int besti = 0;
    if (XYZ_IN_REGS) {
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const float x2 = px[i], y2 = py[i], z2 = pz[i];
        const float d = sq3(x2 - x1, y2 - y1, z2 - z1);
        const float d2 = fminf(d, mind[i]);
        mind[i] = d2;
        // T is a multiple of bs, so within a thread rank(k) grows with i and
        // the strict compare keeps the highest-priority point among equals.
        const bool gt = d2 > best;
        besti = gt ? i : besti;
        best = gt ? d2 : best;
      }
    } else {
      // coordinates stream from L2 in chunks of CH points per thread; the
      // scheduling barrier keeps the compiler from hoisting every load of the
      // round to the top (which would spill the min-distance registers).
      constexpr int CH = 4;
      static_assert(XYZ_IN_REGS || PPT % CH == 0, "PPT must be a multiple of 4");
#pragma unroll
      for (int c0 = 0; c0 < PPT; c0 += CH) {
        // buffer loads: one VGPR offset (t*12) + an SGPR/literal point offset,
        // hardware bounds check returns 0 past the scene (those lanes are
        // pinned to -1 and never compete) -- no per-point address registers.
        float x2[CH], y2[CH], z2[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(
              rsrc, t * 12, (c0 + u) * T * 12, 0);
          x2[u] = __uint_as_float(v.x);
          y2[u] = __uint_as_float(v.y);
          z2[u] = __uint_as_float(v.z);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const float d = sq3(x2[u] - x1, y2[u] - y1, z2[u] - z1);
          const float d2 = fminf(d, mind[c0 + u]);
          mind[c0 + u] = d2;
          const bool gt = d2 > best;
          besti = gt ? (c0 + u) : besti;
          best = gt ? d2 : best;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const u32 k = (u32)(t + besti * T);
    const u32 rank = (bitrev_n(k & (u32)(bs - 1), log2bs) << 22) | (k >> log2bs);
    u64 key = best < 0.0f ? 0ull
                          : ((u64)(__float_as_uint(best) + 1u) << 32) |
                                (u64)(0xFFFFFFFFu - rank);
    key = wave_max_u64(key);
    if (NW > 1) {
      if (lane == 0) s_key[j & 1][wave] = key;
      __syncthreads();
      u64 v = lane < NW ? s_key[j & 1][lane] : 0ull;
      v = row16_max_u64(v);  // NW <= 16: one DPP row
      key = readlane_u64(v, 0);
    }
    if ((key >> 32) == 0ull) {
      old = 0;
    } else {
      const u32 r = 0xFFFFFFFFu - (u32)key;
      old = (int)(((r & 0x3FFFFFu) << log2bs) | bitrev_n(r >> 22, log2bs));
    }
    if (t == 0) idx[j] = old;
  }
}

// Fallback for point sets that do not fit the register-resident variants:
// min-distances live in the caller's `temp` buffer (same contract as the
// reference's scratch, sampling.cpp:74-76).
template <int T>
__global__ __launch_bounds__(T) void fps_kernel_spill(
    int n, int m, int bs, int log2bs, const float *__restrict__ xyz,
    float *__restrict__ temp, int *__restrict__ idx) {
  constexpr int NW = T / kWave;
  __shared__ u64 s_key[2][NW];
  if (m <= 0) return;
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  temp += (size_t)b * n;
  idx += (size_t)b * m;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  for (int k = t; k < n; k += T) {
    const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
    const float mag = sq3(x, y, z);
    temp[k] = ((double)mag <= 1e-3) ? -1.0f : 1e10f;
  }
  int old = 0;
  if (t == 0) idx[0] = 0;
  for (int j = 1; j < m; ++j) {
    const int so = __builtin_amdgcn_readfirstlane(old);
    const float x1 = xyz[so * 3 + 0], y1 = xyz[so * 3 + 1],
                z1 = xyz[so * 3 + 2];
    float best = -1.0f;
    int bestk = 0;
    for (int k = t; k < n; k += T) {
      const float x2 = xyz[k * 3 + 0], y2 = xyz[k * 3 + 1],
                  z2 = xyz[k * 3 + 2];
      const float d = sq3(x2 - x1, y2 - y1, z2 - z1);
      const float d2 = fminf(d, temp[k]);
      temp[k] = d2;
      const bool gt = d2 > best;
      bestk = gt ? k : bestk;
      best = gt ? d2 : best;
    }
    const u32 k = (u32)bestk;
    const u32 rank = (bitrev_n(k & (u32)(bs - 1), log2bs) << 22) | (k >> log2bs);
    u64 key = best < 0.0f ? 0ull
                          : ((u64)(__float_as_uint(best) + 1u) << 32) |
                                (u64)(0xFFFFFFFFu - rank);
    key = wave_max_u64(key);
    if (lane == 0) s_key[j & 1][wave] = key;
    __syncthreads();
    u64 v = lane < NW ? s_key[j & 1][lane] : 0ull;
    v = row16_max_u64(v);
    key = readlane_u64(v, 0);
    if ((key >> 32) == 0ull) {
      old = 0;
    } else {
      const u32 r = 0xFFFFFFFFu - (u32)key;
      old = (int)(((r & 0x3FFFFFu) << log2bs) | bitrev_n(r >> 22, log2bs));
    }
    if (t == 0) idx[j] = old;
  }
}

// cuda_utils.h:13-19 restated (same double-precision expression).
static int ref_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

constexpr int kFpsResidentLimit = 1024 * 96;
extern "C" int s2c_fps_resident_limit(void) { return kFpsResidentLimit; }

#define S2C_FPS_LAUNCH(T_, PPT_, XR_)                                        \
  hipLaunchKernelGGL((fps_kernel<T_, PPT_, XR_>), dim3(b), dim3(T_), 0, st,  \
                     n, m, bs, log2bs, xyz, idx)

extern "C" int s2c_furthest_point_sampling(int b, int n, int m,
                                           const float *xyz, float *temp,
                                           int *idx, s2c_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0) return fail_args("fps: b>=0, n>0, m>=0");
  if (b == 0 || m == 0) return 0;
  if (!xyz || !idx) return fail_args("fps: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int bs = ref_opt_n_threads(n);
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  if (n < 1024) {
    // T = max(64, bs) is a multiple of bs and n < 2*bs  =>  <= 2 points/thread
    const int T = bs < 64 ? 64 : bs;
    const int ppt = (n + T - 1) / T;
    if (ppt > 2) return fail_args("fps: internal geometry");
    switch (T) {
      case 64: S2C_FPS_LAUNCH(64, 2, true); break;
      case 128: S2C_FPS_LAUNCH(128, 2, true); break;
      case 256: S2C_FPS_LAUNCH(256, 2, true); break;
      default: S2C_FPS_LAUNCH(512, 2, true); break;
    }
  } else {
    const int ppt = (n + 1023) / 1024;
    if (ppt <= 1) S2C_FPS_LAUNCH(1024, 1, true);
    else if (ppt <= 2) S2C_FPS_LAUNCH(1024, 2, true);
    else if (ppt <= 4) S2C_FPS_LAUNCH(1024, 4, true);
    else if (ppt <= 8) S2C_FPS_LAUNCH(1024, 8, true);
    else if (ppt <= 16) S2C_FPS_LAUNCH(1024, 16, true);
    else if (ppt <= 24) S2C_FPS_LAUNCH(1024, 24, false);
    else if (ppt <= 40) S2C_FPS_LAUNCH(1024, 40, false);
    else if (ppt <= 64) S2C_FPS_LAUNCH(1024, 64, false);
    else if (ppt <= 96) S2C_FPS_LAUNCH(1024, 96, false);
    else {
      if (!temp) return fail_args("fps: n above resident limit needs temp");
      hipLaunchKernelGGL((fps_kernel_spill<1024>), dim3(b), dim3(1024), 0, st,
                         n, m, bs, log2bs, xyz, temp, idx);
    }
  }
  return check_launch("furthest_point_sampling");
}

// ===========================================================================
// 2/3. gather_points (+grad)  (sampling_gpu.cu:8-20, :34-47)
// ===========================================================================
__global__ __launch_bounds__(256) void gather_points_kernel(
    int c, int n, int m, long long total, const float *__restrict__ points,
    const int *__restrict__ idx, float *__restrict__ out) {
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const int j = (int)(e % m);
    const long long bc = e / m;  // b*c + l
    const long long bi = bc / c;
    const int a = idx[bi * m + j];
    out[e] = points[bc * n + a];
  }
}

__global__ __launch_bounds__(256) void gather_points_grad_kernel(
    int c, int n, int m, long long total, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_points) {
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const int j = (int)(e % m);
    const long long bc = e / m;
    const long long bi = bc / c;
    const int a = idx[bi * m + j];
    atomicAdd(grad_points + bc * n + a, grad_out[e]);
  }
}

static unsigned grid_for(long long total, int per_block) {
  long long g = (total + per_block - 1) / per_block;
  if (g > 256 * 32) g = 256 * 32;  // grid-stride beyond 32 blocks per CU
  if (g < 1) g = 1;
  return (unsigned)g;
}

extern "C" int s2c_gather_points(int b, int c, int n, int npoints,
                                 const float *points, const int *idx,
                                 float *out, s2c_stream_t stream) {
  if (b < 0 || c < 0 || n <= 0 || npoints < 0) return fail_args("gather_points sizes");
  const long long total = (long long)b * c * npoints;
  if (total == 0) return 0;
  if (!points || !idx || !out) return fail_args("gather_points: null pointer");
  hipLaunchKernelGGL(gather_points_kernel, dim3(grid_for(total, 256)), dim3(256),
                     0, (hipStream_t)stream, c, n, npoints, total, points, idx,
                     out);
  return check_launch("gather_points");
}

extern "C" int s2c_gather_points_grad(int b, int c, int n, int npoints,
                                      const float *grad_out, const int *idx,
                                      float *grad_points, s2c_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0) return fail_args("gather_points_grad sizes");
  hipStream_t st = (hipStream_t)stream;
  const long long nout = (long long)b * c * n;
  if (nout == 0) return 0;
  if (!grad_points) return fail_args("gather_points_grad: null pointer");
  hipError_t e = zero_async(grad_points, sizeof(float) * nout, st);
  if (e != hipSuccess) return (int)e;
  const long long total = (long long)b * c * npoints;
  if (total == 0) return 0;
  if (!grad_out || !idx) return fail_args("gather_points_grad: null pointer");
  hipLaunchKernelGGL(gather_points_grad_kernel, dim3(grid_for(total, 256)),
                     dim3(256), 0, st, c, n, npoints, total, grad_out, idx,
                     grad_points);
  return check_launch("gather_points_grad");
}

// ===========================================================================
// 4. ball_query  (ball_query_gpu.cu:9-44)
// ===========================================================================
// One WAVE per centre, 8 centres per workgroup sharing LDS-staged SoA tiles of
// the scene's points (the reference: one THREAD per centre, one block per scene,
// AoS global reads).  64 candidates are tested per step; __ballot + mbcnt give the
// ordered compaction, so hits land in ascending point index exactly as the
// reference's serial scan produces them (:27-41), including "first hit fills the
// whole row" (:34-38) and the early exit at nsample hits (:27).
constexpr int BQ_WAVES = 8;
constexpr int BQ_TILE = 2048;

__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(
    int n, int m, float radius2, int nsample,
    const float *__restrict__ new_xyz, const float *__restrict__ xyz,
    int *__restrict__ idx) {
  __shared__ float s_xyz[3][BQ_TILE];
  extern __shared__ int s_rows[];  // [BQ_WAVES][nsample]
  const int b = blockIdx.y;
  xyz += (size_t)b * n * 3;
  new_xyz += (size_t)b * m * 3;
  idx += (size_t)b * m * nsample;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blockIdx.x * BQ_WAVES + wave;
  int *row = s_rows + wave * nsample;
  for (int l = lane; l < nsample; l += 64) row[l] = 0;  // ball_query.cpp:19-21

  float cx = 0.f, cy = 0.f, cz = 0.f;
  bool done = j >= m;
  if (!done) {
    cx = new_xyz[j * 3 + 0];
    cy = new_xyz[j * 3 + 1];
    cz = new_xyz[j * 3 + 2];
  }
  int cnt = 0;
  for (int base = 0; base < n; base += BQ_TILE) {
    const int tn = min(BQ_TILE, n - base);
    // coalesced flat read of the AoS tile, de-interleaved into SoA in LDS
    for (int f = tid; f < tn * 3; f += BQ_WAVES * 64) {
      const float v = xyz[(size_t)base * 3 + f];
      const int p = f / 3;
      s_xyz[f - p * 3][p] = v;
    }
    __syncthreads();
    if (!done) {
      for (int p0 = 0; p0 < tn; p0 += 64) {
        const int p = p0 + lane;
        const int pc = p < tn ? p : tn - 1;
        const float x = s_xyz[0][pc], y = s_xyz[1][pc], z = s_xyz[2][pc];
        const float d2 = sq3(cx - x, cy - y, cz - z);
        const bool hit = (p < tn) && (d2 < radius2);
        const u64 mask = __ballot(hit);
        if (mask) {
          if (cnt == 0) {
            const int k0 = base + p0 + (int)__builtin_ctzll(mask);
            for (int l = lane; l < nsample; l += 64) row[l] = k0;
          }
          const int pos = cnt + mask_rank_below(mask);
          if (hit && pos < nsample) row[pos] = base + p;
          cnt += (int)__builtin_popcountll(mask);
          if (cnt >= nsample) break;
        }
      }
      done = cnt >= nsample;
    }
    // barrier before the tile is overwritten + block-wide early exit
    if (__syncthreads_and(done)) break;
  }
  if (j < m)
    for (int l = lane; l < nsample; l += 64) idx[(size_t)j * nsample + l] = row[l];
}

extern "C" int s2c_ball_query(int b, int n, int m, float radius, int nsample,
                              const float *new_xyz, const float *xyz, int *idx,
                              s2c_stream_t stream) {
  if (b < 0 || n < 0 || m < 0 || nsample < 0) return fail_args("ball_query sizes");
  if (b == 0 || m == 0 || nsample == 0) return 0;
  if (!new_xyz || !idx || (n > 0 && !xyz)) return fail_args("ball_query: null pointer");
  if (b > 65535) return fail_args("ball_query: b > 65535");
  const size_t dyn = sizeof(int) * (size_t)BQ_WAVES * nsample;
  if (dyn > 96 * 1024) return fail_args("ball_query: nsample too large");
  const float radius2 = radius * radius;  // ball_query_gpu.cu:22
  hipLaunchKernelGGL(ball_query_kernel, dim3(cdiv(m, BQ_WAVES), b),
                     dim3(BQ_WAVES * 64), dyn, (hipStream_t)stream, n, m,
                     radius2, nsample, new_xyz, xyz, idx);
  return check_launch("ball_query");
}

// ===========================================================================
// 5/6. group_points (+grad)  (group_points_gpu.cu:8-28, :43-64)
// ===========================================================================
// Output-stationary: consecutive lanes own consecutive (j,k) of one channel row,
// so index loads and the (large) output stores are fully coalesced; the gathered
// source row (n floats of one channel) is served from L2.
template <int V>
__global__ __launch_bounds__(256) void group_points_kernel(
    int c, int n, long long mk, long long total_v,
    const float *__restrict__ points, const int *__restrict__ idx,
    float *__restrict__ out) {
  // total_v = b*c*mk/V work items; item e covers out[e*V .. e*V+V)
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total_v;
       e += (long long)gridDim.x * 256) {
    const long long o = e * V;
    const long long bc = o / mk;
    const long long r = o - bc * mk;  // j*nsample + k
    const long long bi = bc / c;
    const float *src = points + bc * n;
    const int *ip = idx + bi * mk + r;
    if (V == 4) {
      const int4 ii = *reinterpret_cast<const int4 *>(ip);
      float4 v;
      v.x = src[ii.x]; v.y = src[ii.y]; v.z = src[ii.z]; v.w = src[ii.w];
      *reinterpret_cast<float4 *>(out + o) = v;
    } else {
      out[o] = src[ip[0]];
    }
  }
}

template <int V>
__global__ __launch_bounds__(256) void group_points_grad_kernel(
    int c, int n, long long mk, long long total_v,
    const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points) {
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total_v;
       e += (long long)gridDim.x * 256) {
    const long long o = e * V;
    const long long bc = o / mk;
    const long long r = o - bc * mk;
    const long long bi = bc / c;
    float *dst = grad_points + bc * n;
    const int *ip = idx + bi * mk + r;
    if (V == 4) {
      const int4 ii = *reinterpret_cast<const int4 *>(ip);
      const float4 g = *reinterpret_cast<const float4 *>(grad_out + o);
      atomicAdd(dst + ii.x, g.x);
      atomicAdd(dst + ii.y, g.y);
      atomicAdd(dst + ii.z, g.z);
      atomicAdd(dst + ii.w, g.w);
    } else {
      atomicAdd(dst + ip[0], grad_out[o]);
    }
  }
}

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int s2c_group_points(int b, int c, int n, int npoints, int nsample,
                                const float *points, const int *idx, float *out,
                                s2c_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0)
    return fail_args("group_points sizes");
  const long long mk = (long long)npoints * nsample;
  const long long total = (long long)b * c * mk;
  if (total == 0) return 0;
  if (!points || !idx || !out) return fail_args("group_points: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if ((mk % 4) == 0 && aligned16(idx) && aligned16(out)) {
    hipLaunchKernelGGL((group_points_kernel<4>), dim3(grid_for(total / 4, 256)),
                       dim3(256), 0, st, c, n, mk, total / 4, points, idx, out);
  } else {
    hipLaunchKernelGGL((group_points_kernel<1>), dim3(grid_for(total, 256)),
                       dim3(256), 0, st, c, n, mk, total, points, idx, out);
  }
  return check_launch("group_points");
}

extern "C" int s2c_group_points_grad(int b, int c, int n, int npoints,
                                     int nsample, const float *grad_out,
                                     const int *idx, float *grad_points,
                                     s2c_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0)
    return fail_args("group_points_grad sizes");
  hipStream_t st = (hipStream_t)stream;
  const long long nout = (long long)b * c * n;
  if (nout == 0) return 0;
  if (!grad_points) return fail_args("group_points_grad: null pointer");
  hipError_t e = zero_async(grad_points, sizeof(float) * nout, st);
  if (e != hipSuccess) return (int)e;
  const long long mk = (long long)npoints * nsample;
  const long long total = (long long)b * c * mk;
  if (total == 0) return 0;
  if (!grad_out || !idx) return fail_args("group_points_grad: null pointer");
  if ((mk % 4) == 0 && aligned16(idx) && aligned16(grad_out)) {
    hipLaunchKernelGGL((group_points_grad_kernel<4>),
                       dim3(grid_for(total / 4, 256)), dim3(256), 0, st, c, n,
                       mk, total / 4, grad_out, idx, grad_points);
  } else {
    hipLaunchKernelGGL((group_points_grad_kernel<1>),
                       dim3(grid_for(total, 256)), dim3(256), 0, st, c, n, mk,
                       total, grad_out, idx, grad_points);
  }
  return check_launch("group_points_grad");
}

// ===========================================================================
// 7. three_nn  (interpolate_gpu.cu:9-59)
// ===========================================================================
// Thread per unknown point; the known set is streamed through LDS tiles (SoA,
// uniform broadcast reads).  The strict '<' cascade of the reference (:34-49) is
// kept verbatim, so equal distances keep the earlier k in the better slot.  The
// reference's double 1e40 sentinels (:27) compare against a float d exactly like
// +inf does, and store as +inf after the double->float conversion (:52-54).
constexpr int NN_TILE = 1024;

__global__ __launch_bounds__(256) void three_nn_kernel(
    int n, int m, const float *__restrict__ unknown,
    const float *__restrict__ known, float *__restrict__ dist2,
    int *__restrict__ idx) {
  __shared__ float s_k[3][NN_TILE];
  const int b = blockIdx.y;
  unknown += (size_t)b * n * 3;
  known += (size_t)b * m * 3;
  dist2 += (size_t)b * n * 3;
  idx += (size_t)b * n * 3;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int jc = j < n ? j : n - 1;
  const float ux = unknown[jc * 3 + 0], uy = unknown[jc * 3 + 1],
              uz = unknown[jc * 3 + 2];
  float best1 = INFINITY, best2 = INFINITY, best3 = INFINITY;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int base = 0; base < m; base += NN_TILE) {
    const int tn = min(NN_TILE, m - base);
    __syncthreads();
    for (int f = threadIdx.x; f < tn * 3; f += 256) {
      const float v = known[(size_t)base * 3 + f];
      const int p = f / 3;
      s_k[f - p * 3][p] = v;
    }
    __syncthreads();
    for (int p = 0; p < tn; ++p) {
      const float x = s_k[0][p], y = s_k[1][p], z = s_k[2][p];
      const float d =
          sq3(ux - x, uy - y, uz - z);
      const int k = base + p;
      if (d < best1) {
        best3 = best2; besti3 = besti2;
        best2 = best1; besti2 = besti1;
        best1 = d;     besti1 = k;
      } else if (d < best2) {
        best3 = best2; besti3 = besti2;
        best2 = d;     besti2 = k;
      } else if (d < best3) {
        best3 = d;     besti3 = k;
      }
    }
  }
  if (j < n) {
    dist2[j * 3 + 0] = best1; dist2[j * 3 + 1] = best2; dist2[j * 3 + 2] = best3;
    idx[j * 3 + 0] = besti1;  idx[j * 3 + 1] = besti2;  idx[j * 3 + 2] = besti3;
  }
}

extern "C" int s2c_three_nn(int b, int n, int m, const float *unknown,
                            const float *known, float *dist2, int *idx,
                            s2c_stream_t stream) {
  if (b < 0 || n < 0 || m < 0) return fail_args("three_nn sizes");
  if (b == 0 || n == 0) return 0;
  if (!unknown || !dist2 || !idx || (m > 0 && !known))
    return fail_args("three_nn: null pointer");
  if (b > 65535) return fail_args("three_nn: b > 65535");
  hipLaunchKernelGGL(three_nn_kernel, dim3(cdiv(n, 256), b), dim3(256), 0,
                     (hipStream_t)stream, n, m, unknown, known, dist2, idx);
  return check_launch("three_nn");
}

// ===========================================================================
// 8/9. three_interpolate (+grad)  (interpolate_gpu.cu:72-101, :116-143)
// ===========================================================================
__global__ __launch_bounds__(256) void three_interpolate_kernel(
    int c, int m, int n, long long total, const float *__restrict__ points,
    const int *__restrict__ idx, const float *__restrict__ weight,
    float *__restrict__ out) {
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const int j = (int)(e % n);
    const long long bc = e / n;
    const long long bi = bc / c;
    const float *w = weight + (bi * n + j) * 3;
    const int *ii = idx + (bi * n + j) * 3;
    const float *p = points + bc * m;
    // sum order of interpolate_gpu.cu:98-99
    out[e] = p[ii[0]] * w[0] + p[ii[1]] * w[1] + p[ii[2]] * w[2];
  }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    int c, int n, int m, long long total, const float *__restrict__ grad_out,
    const int *__restrict__ idx, const float *__restrict__ weight,
    float *__restrict__ grad_points) {
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const int j = (int)(e % n);
    const long long bc = e / n;
    const long long bi = bc / c;
    const float *w = weight + (bi * n + j) * 3;
    const int *ii = idx + (bi * n + j) * 3;
    float *gp = grad_points + bc * m;
    const float g = grad_out[e];
    atomicAdd(gp + ii[0], g * w[0]);
    atomicAdd(gp + ii[1], g * w[1]);
    atomicAdd(gp + ii[2], g * w[2]);
  }
}

extern "C" int s2c_three_interpolate(int b, int c, int m, int n,
                                     const float *points, const int *idx,
                                     const float *weight, float *out,
                                     s2c_stream_t stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0) return fail_args("three_interpolate sizes");
  const long long total = (long long)b * c * n;
  if (total == 0) return 0;
  if (!points || !idx || !weight || !out)
    return fail_args("three_interpolate: null pointer");
  hipLaunchKernelGGL(three_interpolate_kernel, dim3(grid_for(total, 256)),
                     dim3(256), 0, (hipStream_t)stream, c, m, n, total, points,
                     idx, weight, out);
  return check_launch("three_interpolate");
}

extern "C" int s2c_three_interpolate_grad(int b, int c, int n, int m,
                                          const float *grad_out, const int *idx,
                                          const float *weight,
                                          float *grad_points,
                                          s2c_stream_t stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0)
    return fail_args("three_interpolate_grad sizes");
  hipStream_t st = (hipStream_t)stream;
  const long long nout = (long long)b * c * m;
  if (nout == 0) return 0;
  if (!grad_points) return fail_args("three_interpolate_grad: null pointer");
  hipError_t e = zero_async(grad_points, sizeof(float) * nout, st);
  if (e != hipSuccess) return (int)e;
  const long long total = (long long)b * c * n;
  if (total == 0) return 0;
  if (!grad_out || !idx || !weight)
    return fail_args("three_interpolate_grad: null pointer");
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(grid_for(total, 256)),
                     dim3(256), 0, st, c, n, m, total, grad_out, idx, weight,
                     grad_points);
  return check_launch("three_interpolate_grad");
}
