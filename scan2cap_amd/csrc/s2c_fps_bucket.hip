// s2c_fps_bucket.hip -- exact furthest point sampling with spatial buckets.
//
// Same result, bit for bit, as the brute-force kernel in s2c_ops.hip (and hence
// as the reference, sampling_gpu.cu:69-173), but each round only touches the
// points whose running min-distance can change.
//
// A pick p lowers temp[k] only if d(p,k) < temp[k].  Points are counting-sorted
// into <= 1024 grid cells; each cell keeps its TIGHT bounding box (from the
// actual coordinates) and the arg-max key of its points.  If a float lower bound
// of the squared distance from p to the box, shrunk by 1e-5 (>> the ~4 ulp
// rounding slack of the distance expression), already exceeds the cell's
// maximum min-distance, no temp[] in the cell can change and the cell keeps its
// key.  Otherwise the cell's points are re-evaluated with EXACTLY the reference's
// float expression.  The global winner is the max over the <= 1024 cell keys,
// which carry the reference's tie-break priority (see s2c_ops.hip), so the
// answer does not depend on the cell order, the order inside a cell or the
// thread geometry.
//
// One 1024-thread workgroup per scene (the rounds are serial).  Per round:
//   A  every thread tests its own cell (box in registers) and appends active
//      cells to an LDS list (one LDS atomic per wave);
//   B  64 rows of 16 lanes each walk the active cells: 16-B {x,y,z,d2} loads of
//      the sorted points (L2 resident), 4-step DPP row arg-max, one LDS write;
//   C  1024-way arg-max of the cell keys (DPP + one LDS exchange).
#include "s2c_common.h"
#include "../../include/s2c_ops.h"

#include <math.h>
#include <stdio.h>

using namespace s2c;

namespace {

constexpr int T = 1024;
constexpr int NW = T / 64;
constexpr int MAXC = 1024;  // cells, one per thread

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

__global__ __launch_bounds__(T) void fps_bucket_kernel(
    int n, int m, int bs, int log2bs, int target_cells,
    const float *__restrict__ xyz, Pt *__restrict__ spt, u32 *__restrict__ srank,
    int *__restrict__ idx) {
  __shared__ int s_cnt[MAXC], s_start[MAXC], s_cursor[MAXC], s_cand[MAXC];
  __shared__ u32 s_bb[6][MAXC];
  __shared__ u64 s_cellkey[MAXC];
  __shared__ float s_cellxyz[3][MAXC];   // coordinates of each cell's arg-max point
  __shared__ unsigned short s_list[MAXC];
  __shared__ int s_nactive[2];
  __shared__ u64 s_wkey[2][NW];
  __shared__ float s_wxyz[2][NW][4];
  __shared__ u32 s_red[6][NW];
  __shared__ float s_grid[9];  // lo[3], inv cell[3]; dims as ints below
  __shared__ int s_dims[4];
  __shared__ int s_wsum[NW];
  if (m <= 0) return;
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  spt += (size_t)b * n;
  srank += (size_t)b * n;
  idx += (size_t)b * m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- S1: scene bounding box --------------------------------------------
  {
    u32 lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (int k = tid; k < n; k += T) {
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
    if (tid < 2) s_nactive[tid] = 0;
    __syncthreads();
    if (tid == 0) {
      float flo[3], fhi[3], ext[3];
      for (int a = 0; a < 3; ++a) {
        u32 l = 0xFFFFFFFFu, h = 0u;
        for (int w = 0; w < NW; ++w) { l = min(l, s_red[a][w]); h = max(h, s_red[3 + a][w]); }
        flo[a] = ord_inv(l); fhi[a] = ord_inv(h);
        ext[a] = fmaxf(fhi[a] - flo[a], 1e-6f);
      }
      // cubic cells of edge e with ~target_cells cells in the box
      const float vol = ext[0] * ext[1] * ext[2];
      float e = cbrtf(vol / (float)target_cells);
      int g[3];
      for (int it = 0; it < 64; ++it) {
        for (int a = 0; a < 3; ++a) g[a] = max(1, min(1024, (int)(ext[a] / e) + 1));
        if ((long long)g[0] * g[1] * g[2] <= MAXC) break;
        e *= 1.05f;
      }
      if ((long long)g[0] * g[1] * g[2] > MAXC) { g[0] = g[1] = g[2] = 1; }
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
  const int gx = s_dims[0], gy = s_dims[1], gz = s_dims[2], ncells = s_dims[3];
  auto cell_of = [&](float x, float y, float z) {
    const int ix = min(gx - 1, max(0, (int)((x - glx) * gix)));
    const int iy = min(gy - 1, max(0, (int)((y - gly) * giy)));
    const int iz = min(gz - 1, max(0, (int)((z - glz) * giz)));
    return ix + gx * (iy + gy * iz);
  };

  // ---- S2: histogram -> scan -> scatter (counting sort by cell) ------------
  for (int k = tid; k < n; k += T)
    atomicAdd(&s_cnt[cell_of(xyz[k * 3], xyz[k * 3 + 1], xyz[k * 3 + 2])], 1);
  __syncthreads();
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
    s_start[tid] = base + inc - v;
    s_cursor[tid] = base + inc - v;
  }
  __syncthreads();
  for (int k = tid; k < n; k += T) {
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

  // ---- per-thread cell state ------------------------------------------------
  const bool own = tid < ncells && s_cand[tid] > 0;  // cells with no candidate never compete
  float blx = 0, bly = 0, blz = 0, bhx = 0, bhy = 0, bhz = 0;
  if (own) {
    blx = ord_inv(s_bb[0][tid]); bly = ord_inv(s_bb[1][tid]); blz = ord_inv(s_bb[2][tid]);
    bhx = ord_inv(s_bb[3][tid]); bhy = ord_inv(s_bb[4][tid]); bhz = ord_inv(s_bb[5][tid]);
  }
  u64 mykey = own ? ((u64)(__float_as_uint(1e10f) + 1u) << 32) : 0ull;
  s_cellkey[tid] = mykey;

  int old = 0;
  if (tid == 0) idx[0] = 0;
  s_cellxyz[0][tid] = 0.f; s_cellxyz[1][tid] = 0.f; s_cellxyz[2][tid] = 0.f;
  __syncthreads();

  const int row = tid >> 4, rl = tid & 15;
  // pivot coordinates travel with the arg-max through LDS (no dependent global
  // load per round); x0.. is point 0 = first pivot and the "nothing selectable" case
  const float x0 = xyz[0], y0 = xyz[1], z0 = xyz[2];
  float px = x0, py = y0, pz = z0;
  for (int j = 1; j < m; ++j) {
    const int par = j & 1;
    // ---- A: which cells can change? ---------------------------------------
    bool active = false;
    if (own) {
      const float ddx = fmaxf(fmaxf(blx - px, px - bhx), 0.0f);
      const float ddy = fmaxf(fmaxf(bly - py, py - bhy), 0.0f);
      const float ddz = fmaxf(fmaxf(blz - pz, pz - bhz), 0.0f);
      const float lb = (ddx * ddx + ddy * ddy + ddz * ddz) * 0.99999f;
      const float cmax = __uint_as_float((u32)(mykey >> 32) - 1u);
      active = !(lb > cmax);
    }
    const u64 amask = __ballot(active);
    if (amask) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_nactive[par], (int)__builtin_popcountll(amask));
      base = __builtin_amdgcn_readfirstlane(base);
      if (active) s_list[base + mask_rank_below(amask)] = (unsigned short)tid;
    }
    __syncthreads();
    // ---- B: re-evaluate the active cells --------------------------------------
    // few active cells (the common case: ~15 of 1024): one WAVE per cell, so a
    // ~40-point cell is a single 64-lane pass; many: one 16-lane row per cell.
    const int na = s_nactive[par];
    if (tid == 0) s_nactive[par ^ 1] = 0;
    if (na <= NW) {
      if (wave < na) {
        const int c = s_list[wave];
        const int st = s_start[c], nc = s_cnt[c];
        u64 best = 0ull;
        float bx = 0.f, by = 0.f, bz = 0.f;
        for (int q = lane; q < nc; q += 64) {
          const Pt p = spt[st + q];
          const u32 rk = srank[st + q];
          const float d = sq3(p.x - px, p.y - py, p.z - pz);
          const float d2 = fminf(d, p.d2);
          if (d2 != p.d2) spt[st + q].d2 = d2;
          const u64 key = d2 < 0.0f ? 0ull
                                    : ((u64)(__float_as_uint(d2) + 1u) << 32) |
                                          (u64)(0xFFFFFFFFu - rk);
          if (key > best) { best = key; bx = p.x; by = p.y; bz = p.z; }
        }
        const u64 wbest = wave_max_u64(best);
        if (best == wbest && wbest != 0ull) {   // unique lane (ranks are unique)
          s_cellkey[c] = wbest;
          s_cellxyz[0][c] = bx; s_cellxyz[1][c] = by; s_cellxyz[2][c] = bz;
        }
      }
    } else {
      for (int e = row; e < na; e += T / 16) {
        const int c = s_list[e];
        const int st = s_start[c], nc = s_cnt[c];
        u64 best = 0ull;
        float bx = 0.f, by = 0.f, bz = 0.f;
        for (int q = rl; q < nc; q += 16) {
          const Pt p = spt[st + q];
          const u32 rk = srank[st + q];
          const float d = sq3(p.x - px, p.y - py, p.z - pz);
          const float d2 = fminf(d, p.d2);
          if (d2 != p.d2) spt[st + q].d2 = d2;
          const u64 key = d2 < 0.0f ? 0ull
                                    : ((u64)(__float_as_uint(d2) + 1u) << 32) |
                                          (u64)(0xFFFFFFFFu - rk);
          if (key > best) { best = key; bx = p.x; by = p.y; bz = p.z; }
        }
        const u64 rbest = row16_max_u64(best);
        if (best == rbest && rbest != 0ull) {
          s_cellkey[c] = rbest;
          s_cellxyz[0][c] = bx; s_cellxyz[1][c] = by; s_cellxyz[2][c] = bz;
        }
      }
    }
    __syncthreads();
    // ---- C: arg-max over the cell keys (coordinates ride along) ---------------
    mykey = s_cellkey[tid];
    u64 key = wave_max_u64(mykey);
    if (mykey == key && key != 0ull) {
      s_wkey[par][wave] = key;
      s_wxyz[par][wave][0] = s_cellxyz[0][tid];
      s_wxyz[par][wave][1] = s_cellxyz[1][tid];
      s_wxyz[par][wave][2] = s_cellxyz[2][tid];
    } else if (key == 0ull && lane == 0) {
      s_wkey[par][wave] = 0ull;
    }
    __syncthreads();
    u64 v = lane < NW ? s_wkey[par][lane] : 0ull;
    const u64 mine = v;
    v = row16_max_u64(v);
    key = readlane_u64(v, 0);
    if ((key >> 32) == 0ull) {
      old = 0;
      px = x0; py = y0; pz = z0;
    } else {
      const u32 r = 0xFFFFFFFFu - (u32)key;
      old = (int)(((r & 0x3FFFFFu) << log2bs) | bitrev_n(r >> 22, log2bs));
      const u64 wl = __ballot(lane < NW && mine == key);
      const int w = (int)__builtin_ctzll(wl);
      px = s_wxyz[par][w][0]; py = s_wxyz[par][w][1]; pz = s_wxyz[par][w][2];
    }
    if (tid == 0) idx[j] = old;
  }
}

}  // namespace

static thread_local char g_err3[256] = "";
extern "C" const char *s2c_fps_last_error_string(void) { return g_err3; }

extern "C" long long s2c_fps_workspace_bytes(int b, int n) {
  return (long long)b * n * (long long)(sizeof(Pt) + sizeof(u32));
}

// Same contract as s2c_furthest_point_sampling, plus a caller-provided scratch
// of s2c_fps_workspace_bytes(b, n) bytes (16-byte aligned).
extern "C" int s2c_furthest_point_sampling_bucketed(int b, int n, int m,
                                                    const float *xyz,
                                                    void *workspace, int *idx,
                                                    s2c_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0 || !xyz || !idx || !workspace ||
      ((uintptr_t)workspace & 15)) {
    snprintf(g_err3, sizeof(g_err3), "s2c: fps_bucketed: invalid argument");
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
  Pt *spt = (Pt *)workspace;
  u32 *srank = (u32 *)((char *)workspace + (size_t)b * n * sizeof(Pt));
  hipLaunchKernelGGL(fps_bucket_kernel, dim3(b), dim3(T), 0, (hipStream_t)stream, n,
                     m, bs, log2bs, target, xyz, spt, srank, idx);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err3, sizeof(g_err3), "s2c: fps_bucketed launch failed: %s",
             hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
