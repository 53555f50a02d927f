// s2c_bq_grid.hip -- ball query on a uniform grid (large point sets).
//
// Same result, bit for bit, as ball_query_kernel in s2c_ops.hip and hence as the
// reference (ball_query_gpu.cu:9-44: the first `nsample` points in ASCENDING point
// index with d2 < r2, the first hit pads the row, no hit leaves the zero-initialised
// row of ball_query.cpp:19-21) -- but a centre only looks at the points of the <= 27
// grid cells its ball can touch instead of all n (SA1 of the BASELINE workload:
// ~90 candidates instead of 40000, 655 M distance tests -> 1.5 M per step).
//
//   build  one 1024-thread workgroup per scene: bounding box -> cubic cells of edge
//          r * (1 + 1e-4) (so |dx| < r can never skip a cell, whatever the rounding of
//          the cell index) -> LDS histogram (<= 32768 cells = 128 KB) -> scan ->
//          counting sort of {x, y, z, index} records (16 B) into cell order.  Cells are
//          linearised x-fastest: the 3 x-neighbours of a cell row are ONE contiguous
//          range of the sorted records.
//   query  one WAVE per centre: 9 ranges (dy, dz in -1..1), candidates flattened over
//          the ranges so that 64 lanes test 64 candidates per step (a candidate is one
//          16-byte load), exactly the reference's float expression, hits compacted
//          with ballot + mbcnt into a wave-private LDS list.  The grid visits points
//          in cell order, the reference in index order: the hits are ranked by index
//          afterwards (h <= 64: rank by counting over readlane broadcasts; more: the
//          nsample-th smallest index by bisection with ballot counts, then the same).
//          More hits than the list holds (a ball that swallows > 1024 points) falls
//          back to the reference's own ascending scan, which then exits after a few
//          hundred points.
#include "s2c_common.h"
#include "../../include/s2c_ops.h"

#include <math.h>
#include <stdio.h>

using namespace s2c;

namespace {

constexpr int BT = 1024;          // build: threads per scene
constexpr int BNW = BT / 64;
constexpr int MAXC = 32768;       // cells per scene (LDS histogram of the build)
constexpr int MAXAXIS = 256;      // cells per axis (keeps the cell-index rounding inside the margin)
constexpr int QW = 4;             // query: waves (centres) per workgroup
constexpr int CAP = 1024;         // query: hit list entries per wave
constexpr int HDR_BYTES = 64;
constexpr int U = 8;              // build: points per thread in flight
// LDS histogram index of cell c: one pad word per 32 cells, so that the scan -- thread t
// walks cells 32 t .. 32 t + 31 -- hits bank (t + q) % 32 instead of bank q for all 64
// lanes (a 64-way conflict on every access: 30 us of a 60 us kernel)
__device__ __forceinline__ int hslot(int c) { return c + (c >> 5); }
constexpr int HIST_WORDS = MAXC + (MAXC >> 5);

struct __attribute__((aligned(16))) Rec { float x, y, z; int k; };

struct GridHdr {                  // first HDR_BYTES of a scene's workspace
  float lo[3];
  float inv[3];
  int dim[3];
  int ncell;
  int pad[6];
};
static_assert(sizeof(GridHdr) == HDR_BYTES, "header layout");

__host__ __device__ inline size_t scene_bytes(int n) {
  // header | start[MAXC + 4] | records[n]
  return (size_t)HDR_BYTES + (size_t)(MAXC + 4) * 4 + (size_t)n * sizeof(Rec);
}

__device__ __forceinline__ u32 ord_of(float f) {
  const u32 b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_inv(u32 o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

__device__ __forceinline__ int cell_axis(float v, float lo, float inv, int g) {
  const int i = (int)floorf((v - lo) * inv);
  return min(g - 1, max(0, i));
}

// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(BT) void bq_grid_build_kernel(
    int n, float radius, const float *__restrict__ xyz, char *__restrict__ ws,
    size_t stride) {
  extern __shared__ __attribute__((aligned(16))) int s_cnt[];   // [HIST_WORDS]
  __shared__ u32 s_red[6][BNW];
  __shared__ int s_wsum[BNW];
  __shared__ GridHdr s_hdr;
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  ws += (size_t)b * stride;
  GridHdr *hdr = (GridHdr *)ws;
  int *start = (int *)(ws + HDR_BYTES);
  Rec *rec = (Rec *)(ws + HDR_BYTES + (size_t)(MAXC + 4) * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- bounding box ---------------------------------------------------------------
  {
    u32 lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    // U points per thread in flight: the three passes over the cloud are chains of
    // dependent-latency loads otherwise (one workgroup per scene: nothing else hides them)
    for (int k0 = tid; k0 < n; k0 += BT * U) {
      float p[U][3];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = min(k0 + u * BT, n - 1);     // clamped: a repeated point is harmless
#pragma unroll
        for (int a = 0; a < 3; ++a) p[u][a] = xyz[k * 3 + a];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const u32 o = ord_of(p[u][a]);
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
    for (int c = tid; c < HIST_WORDS; c += BT) s_cnt[c] = 0;
    __syncthreads();
    if (tid == 0) {
      float flo[3], ext[3];
      for (int a = 0; a < 3; ++a) {
        u32 l = 0xFFFFFFFFu, h = 0u;
        for (int w = 0; w < BNW; ++w) { l = min(l, s_red[a][w]); h = max(h, s_red[3 + a][w]); }
        flo[a] = ord_inv(l);
        ext[a] = fmaxf(ord_inv(h) - flo[a], 0.0f);
      }
      // edge >= r (1 + 1e-4): two points closer than r along an axis are at most one
      // cell apart whatever the rounding of (v - lo) * inv (relative error ~1.2e-7 x the
      // cell index: the 1e-4 margin covers indices up to MAXAXIS = 256 with a factor 3 to
      // spare, so no axis gets more cells than that -- a corridor-like cloud, extent / r in
      // the hundreds, gets coarser cells instead of a missed neighbour); grown until the
      // grid fits the LDS histogram
      float e = fmaxf(radius * 1.0001f, 1e-12f);
      int g[3];
      for (int it = 0; it < 200; ++it) {
        long long prod = 1;
        int gmax = 0;
        for (int a = 0; a < 3; ++a) {
          const float q = ext[a] / e;
          g[a] = q < 4.0e6f ? (int)q + 1 : 4000001;
          prod *= g[a];
          gmax = max(gmax, g[a]);
        }
        if (prod <= MAXC && gmax <= MAXAXIS) break;
        e *= 1.1f;
      }
      if ((long long)g[0] * g[1] * g[2] > MAXC || max(g[0], max(g[1], g[2])) > MAXAXIS) {
        g[0] = g[1] = g[2] = 1;
      }
      for (int a = 0; a < 3; ++a) {
        s_hdr.lo[a] = flo[a];
        s_hdr.inv[a] = 1.0f / e;
        s_hdr.dim[a] = g[a];
      }
      s_hdr.ncell = g[0] * g[1] * g[2];
      *hdr = s_hdr;
      start[s_hdr.ncell] = n;                // (the scan below covers cells < MAXC only)
    }
    __syncthreads();
  }
  const float lx = s_hdr.lo[0], ly = s_hdr.lo[1], lz = s_hdr.lo[2];
  const float ix = s_hdr.inv[0], iy = s_hdr.inv[1], iz = s_hdr.inv[2];
  const int gx = s_hdr.dim[0], gy = s_hdr.dim[1], gz = s_hdr.dim[2];
  const int ncell = s_hdr.ncell;
  auto cell_of = [&](float x, float y, float z) {
    return cell_axis(x, lx, ix, gx) +
           gx * (cell_axis(y, ly, iy, gy) + gy * cell_axis(z, lz, iz, gz));
  };

  // ---- histogram -> exclusive scan -> scatter ------------------------------------------
  for (int k0 = tid; k0 < n; k0 += BT * U) {
    float p[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + u * BT, n - 1);
#pragma unroll
      for (int a = 0; a < 3; ++a) p[u][a] = xyz[k * 3 + a];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k0 + u * BT < n) atomicAdd(&s_cnt[hslot(cell_of(p[u][0], p[u][1], p[u][2]))], 1);
  }
  __syncthreads();
  {
    constexpr int PER = MAXC / BT;  // 32 consecutive cells per thread
    int local[PER];
    int sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { local[q] = s_cnt[hslot(tid * PER + q)]; sum += local[q]; }
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    int base = inc - sum;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int c = tid * PER + q;
      s_cnt[hslot(c)] = base;                // cursor
      if (c <= ncell) start[c] = base;       // start[ncell] = n
      base += local[q];
    }
  }
  __syncthreads();
  for (int k0 = tid; k0 < n; k0 += BT * U) {
    float p[U][3];
    int pos[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + u * BT, n - 1);
#pragma unroll
      for (int a = 0; a < 3; ++a) p[u][a] = xyz[k * 3 + a];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      pos[u] = (k0 + u * BT < n)
                   ? atomicAdd(&s_cnt[hslot(cell_of(p[u][0], p[u][1], p[u][2]))], 1) : -1;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (pos[u] >= 0) {
        Rec r; r.x = p[u][0]; r.y = p[u][1]; r.z = p[u][2]; r.k = k0 + u * BT;
        rec[pos[u]] = r;
      }
  }
}

// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(QW * 64) void ball_query_grid_kernel(
    int b_total, int n, int m, float radius2, int nsample, int blocks_per_scene,
    const float *__restrict__ new_xyz, const float *__restrict__ xyz,
    const char *__restrict__ ws, size_t stride, int *__restrict__ idx) {
  __shared__ int s_list[QW][CAP];
  __shared__ int s_rstart[QW][16], s_rpre[QW][16];
  // blockIdx -> (scene, block): with b a multiple of 8 the blocks of scene s run on XCD
  // s % 8 (the dispatcher places block i on XCD i % 8), so that a scene's records and cell
  // table are cached by ONE L2 instead of all eight -- speed only
  int b, blk;
  if ((b_total & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    b = xcd + 8 * (slot / blocks_per_scene);
    blk = slot % blocks_per_scene;
  } else {
    b = blockIdx.x / blocks_per_scene;
    blk = blockIdx.x % blocks_per_scene;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blk * QW + wave;
  if (j >= m) return;                      // waves are independent: no block barrier
  ws += (size_t)b * stride;
  const GridHdr *hdr = (const GridHdr *)ws;
  const int *start = (const int *)(ws + HDR_BYTES);
  const Rec *rec = (const Rec *)(ws + HDR_BYTES + (size_t)(MAXC + 4) * 4);
  xyz += (size_t)b * n * 3;
  new_xyz += (size_t)b * m * 3;
  idx += ((size_t)b * m + j) * nsample;
  int *list = s_list[wave];

  const float cx = new_xyz[j * 3 + 0], cy = new_xyz[j * 3 + 1], cz = new_xyz[j * 3 + 2];
  const int gx = hdr->dim[0], gy = hdr->dim[1], gz = hdr->dim[2];
  const int icx = cell_axis(cx, hdr->lo[0], hdr->inv[0], gx);
  const int icy = cell_axis(cy, hdr->lo[1], hdr->inv[1], gy);
  const int icz = cell_axis(cz, hdr->lo[2], hdr->inv[2], gz);

  // ---- the 9 candidate ranges (lanes 0..8) ---------------------------------------------
  int len = 0, rs = 0;
  if (lane < 9) {
    const int yy = icy + (lane % 3) - 1, zz = icz + (lane / 3) - 1;
    if (yy >= 0 && yy < gy && zz >= 0 && zz < gz) {
      const int rowc = gx * (yy + gy * zz);
      const int c0 = rowc + max(icx - 1, 0), c1 = rowc + min(icx + 1, gx - 1);
      rs = start[c0];
      len = start[c1 + 1] - rs;
    }
  }
  int incl = len;
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  const int total = __builtin_amdgcn_readlane(incl, 8);
  if (lane < 9) { s_rstart[wave][lane] = rs; s_rpre[wave][lane] = incl - len; }
  int pre[9];
#pragma unroll
  for (int r = 1; r < 9; ++r) pre[r] = __builtin_amdgcn_readlane(incl, r - 1);

  // ---- candidates -> hit list ----------------------------------------------------------
  int hits = 0;
  for (int c0 = 0; c0 < total; c0 += 64) {
    const int cand = c0 + lane;
    const bool act = cand < total;
    int r = 0;
#pragma unroll
    for (int q = 1; q < 9; ++q) r += (cand >= pre[q]) ? 1 : 0;
    const int pos = act ? s_rstart[wave][r] + (cand - s_rpre[wave][r]) : 0;
    const Rec p = rec[pos];
    const float d2 = sq3(cx - p.x, cy - p.y, cz - p.z);
    const bool hit = act && (d2 < radius2);
    const u64 mask = __ballot(hit);
    if (mask) {
      const int w = hits + mask_rank_below(mask);
      if (hit && w < CAP) list[w] = p.k;
      hits += (int)__builtin_popcountll(mask);
    }
  }

  int nout = 0;        // valid entries of the sorted prefix, left in list[0 .. nout)
  if (hits > CAP) {
    // ---- a ball holding more points than the list: the reference's own ascending scan
    int cnt = 0;
    for (int p0 = 0; p0 < n && cnt < nsample; p0 += 64) {
      const int p = p0 + lane;
      const int pc = p < n ? p : n - 1;
      const float x = xyz[pc * 3], y = xyz[pc * 3 + 1], z = xyz[pc * 3 + 2];
      const float d2 = sq3(cx - x, cy - y, cz - z);
      const bool hit = (p < n) && (d2 < radius2);
      const u64 mask = __ballot(hit);
      if (mask) {
        const int w = cnt + mask_rank_below(mask);
        if (hit && w < nsample) list[w] = p;
        cnt += (int)__builtin_popcountll(mask);
      }
    }
    nout = min(cnt, nsample);
  } else if (hits > 0) {
    int h = hits;
    if (h > 64) {
      // the nsample-th smallest index T by bisection (indices are distinct), then the
      // entries <= T compacted to the front of the list: exactly nsample of them
      const int want = min(nsample, h);
      int lo = 0, hi = n - 1;
      while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        int cnt = 0;
        for (int i0 = 0; i0 < h; i0 += 64) {
          const int i = i0 + lane;
          cnt += (int)__builtin_popcountll(__ballot(i < h && list[i] <= mid));
        }
        if (cnt >= want) hi = mid; else lo = mid + 1;
      }
      int kept = 0;
      for (int i0 = 0; i0 < h; i0 += 64) {
        const int i = i0 + lane;
        const int v = i < h ? list[i] : 0;
        const bool keep = i < h && v <= lo;
        const u64 mask = __ballot(keep);
        const int w = kept + mask_rank_below(mask);
        // in-place compaction is safe: w <= i for every kept entry, and a wave's LDS
        // accesses execute in order (this chunk is read before anything is written)
        if (keep) list[w] = v;
        kept += (int)__builtin_popcountll(mask);
      }
      h = kept;                                  // == want <= 64
    }
    // rank by counting: h <= 64 values, one per lane
    const int v = lane < h ? list[lane] : 0x7FFFFFFF;
    int rank = 0;
    for (int t = 0; t < h; ++t) rank += (__builtin_amdgcn_readlane(v, t) < v) ? 1 : 0;
    if (lane < h) list[rank] = v;
    nout = min(h, nsample);
  }
  for (int l = lane; l < nsample; l += 64)
    idx[l] = nout == 0 ? 0 : list[l < nout ? l : 0];
}

}  // namespace

static thread_local char g_err4[256] = "";
extern "C" const char *s2c_bq_grid_last_error_string(void) { return g_err4; }

extern "C" long long s2c_ball_query_workspace_bytes(int b, int n) {
  if (b <= 0 || n <= 0) return 16;
  return (long long)b * (long long)((scene_bytes(n) + 15) & ~(size_t)15);
}

extern "C" int s2c_ball_query_grid_max_nsample(void) { return 64; }

// Same contract as s2c_ball_query plus a caller-owned, 16-byte aligned scratch of
// s2c_ball_query_workspace_bytes(b, n) bytes; nsample <= s2c_ball_query_grid_max_nsample().
extern "C" int s2c_ball_query_grid(int b, int n, int m, float radius, int nsample,
                                   const float *new_xyz, const float *xyz, void *workspace,
                                   int *idx, s2c_stream_t stream) {
  if (b < 0 || n < 0 || m < 0 || nsample < 0 || nsample > 64 || !(radius > 0.0f)) {
    snprintf(g_err4, sizeof(g_err4), "s2c: ball_query_grid: invalid argument");
    return S2C_EINVAL;
  }
  if (b == 0 || m == 0 || nsample == 0) return 0;
  if (!new_xyz || !idx || !xyz || n == 0 || !workspace || ((uintptr_t)workspace & 15)) {
    snprintf(g_err4, sizeof(g_err4), "s2c: ball_query_grid: null / unaligned pointer");
    return S2C_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t stride = (scene_bytes(n) + 15) & ~(size_t)15;
  // the build kernel's LDS histogram needs the opt-in dynamic LDS size: set (and checked) once
  // per DEVICE; a device / context that refuses it gets S2C_ENOSUP and the caller falls back
  // to the brute-force kernel (s2c_ball_query)
  {
    static int attr_state[64];                 // 0 unknown, 1 ok, -1 refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (attr_state[dev] == 0)
      attr_state[dev] = hipFuncSetAttribute((const void *)bq_grid_build_kernel,
                                            hipFuncAttributeMaxDynamicSharedMemorySize,
                                            HIST_WORDS * 4) == hipSuccess ? 1 : -1;
    if (attr_state[dev] < 0) {
      (void)hipGetLastError();
      snprintf(g_err4, sizeof(g_err4), "s2c: ball_query_grid: %d bytes of LDS not available",
               HIST_WORDS * 4);
      return S2C_ENOSUP;
    }
  }
  hipLaunchKernelGGL(bq_grid_build_kernel, dim3(b), dim3(BT), HIST_WORDS * 4, st, n, radius, xyz,
                     (char *)workspace, stride);
  const int bps = (m + QW - 1) / QW;
  const float radius2 = radius * radius;  // ball_query_gpu.cu:22
  hipLaunchKernelGGL(ball_query_grid_kernel, dim3((unsigned)bps * (unsigned)b), dim3(QW * 64),
                     0, st, b, n, m, radius2, nsample, bps, new_xyz, xyz,
                     (const char *)workspace, stride, idx);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err4, sizeof(g_err4), "s2c: ball_query_grid launch failed: %s",
             hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
