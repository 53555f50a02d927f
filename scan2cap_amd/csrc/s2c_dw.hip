// s2c_dw.hip -- weight gradient of a rows x channels layer, dW = dY^T A:
//     dW[co, ci] = sum_m dY[m, co] * A[m, ci],   m up to ~1e6, Cout x Cin <= 259 x 512.
// (reference: autograd of the 1x1 Conv2d / Conv1d / Linear layers, e.g.
// lib/pointnet2/pytorch_utils.py:11-120; here the backward of pointnet2/fused.py::_MLPRows.)
//
// A library GEMM sees one small output tile and a million-deep reduction; the previous
// path split the rows into slabs (strided-batched GEMM) and summed the partial products
// with a second kernel.  This kernel is built for the shape instead:
//  * the reduction index m is the ROW index of both operands, so the 8 consecutive-k values
//    a lane feeds to v_mfma_f32_32x32x16_bf16 are 8 rows of one column: each lane fetches
//    them with 8 plain dword loads (a load instruction covers two 128-byte row segments),
//    splits them into bf16 hi/mid/lo planes in registers and issues the 6 plane products
//    (fp32-accurate, see s2c_gemm.hip) -- no LDS staging, no transposition, every element
//    of dY and A is read once per output tile;
//  * a workgroup = 4 waves on the same 64x64 output tile and interleaved 16-row steps of a
//    slab of rows; the four accumulators meet in LDS;
//  * slabs meet through two levels of "last workgroup to arrive adds up" (fixed order =>
//    deterministic, no float atomics, no second launch): 16 slabs -> group partial,
//    groups -> dW.  Counters live in a caller-provided workspace that the kernel leaves
//    zeroed again.
// Block index -> (slab, tile) keeps the tiles of one slab on one XCD (shared L2).
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int DW_GROUP = 16;     // slabs per first-level group

struct Planes { bf16x8 p[3]; };   // hi, mid, lo of 8 consecutive-k values

__device__ __forceinline__ Planes split8(const float (&v)[8]) {
  u32x4 h, m, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 x = {v[2 * q], v[2 * q + 1]};
    const bf16x2 hb = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(hb, f32x2);
    const bf16x2 mb = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(mb, f32x2);
    const bf16x2 lb = __builtin_convertvector(r2, bf16x2);
    h[q] = __builtin_bit_cast(unsigned, hb);
    m[q] = __builtin_bit_cast(unsigned, mb);
    l[q] = __builtin_bit_cast(unsigned, lb);
  }
  Planes o;
  o.p[0] = __builtin_bit_cast(bf16x8, h);
  o.p[1] = __builtin_bit_cast(bf16x8, m);
  o.p[2] = __builtin_bit_cast(bf16x8, l);
  return o;
}

// 8 rows (m0 + r) of column `col`, zero outside the matrix
__device__ __forceinline__ void load8(float (&v)[8], const float *__restrict__ X, long long ld,
                                      long long m0, long long M, int col, int ncol) {
  const bool colok = col < ncol;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const long long m = m0 + r;
    v[r] = (colok && m < M) ? X[m * ld + col] : 0.0f;
  }
}

__device__ __forceinline__ void dw_x3_body(
    int bid, long long M, int Cout, int Cin, const float *__restrict__ dY, long long ldy,
    const float *__restrict__ A, long long lda, int rows_per_slab, int nslab, int ntci,
    int ntiles, float *__restrict__ part1, float *__restrict__ part2,
    unsigned *__restrict__ cnt, float *__restrict__ dW, int lddw) {
  __shared__ float s_acc[2][4 * 16 * 64];      // two waves' 64x64 accumulators (32 KB)
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  // XCD-aware mapping: blocks with equal (bid % 8) share an L2
  const int xcd = bid & 7;
  const int t = (bid >> 3) % ntiles;
  const int slab = xcd + 8 * ((bid >> 3) / ntiles);
  if (slab >= nslab) return;
  const int co0 = (t / ntci) * 64, ci0 = (t % ntci) * 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const long long row0 = (long long)slab * rows_per_slab;
  const long long row1 = row0 + rows_per_slab < M ? row0 + rows_per_slab : M;
  // wave w takes the 16-row steps w, w+4, ... of the slab
  float va[2][8], vb[2][8], na[2][8], nb[2][8];
  long long m = row0 + 16 * wave;
  if (m < row1) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      load8(va[c], dY, ldy, m + 8 * lk, row1, co0 + 32 * c + li, Cout);
      load8(vb[c], A, lda, m + 8 * lk, row1, ci0 + 32 * c + li, Cin);
    }
  }
  for (; m < row1; m += 64) {
    const long long mn = m + 64;
    if (mn < row1) {           // next step's rows are requested before this step's MFMAs
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        load8(na[c], dY, ldy, mn + 8 * lk, row1, co0 + 32 * c + li, Cout);
        load8(nb[c], A, lda, mn + 8 * lk, row1, ci0 + 32 * c + li, Cin);
      }
    }
    Planes pa[2], pb[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      pa[c] = split8(va[c]);
      pb[c] = split8(vb[c]);
    }
    // x*y ~= the 6 plane products with i + j <= 2, small terms first, accumulators in turn
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i].p[TA[q]], pb[j].p[TB[q]],
                                                              acc[i][j], 0, 0, 0);
    if (mn < row1) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          va[c][r] = na[c][r];
          vb[c][r] = nb[c][r];
        }
    }
  }

  // ---- the four waves' accumulators meet in LDS: 2,3 -> 0,1 then 1 -> 0 -----------------
  auto put = [&](float *dst) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[((i * 2 + j) * 16 + e) * 64 + lane] = acc[i][j][e];
  };
  auto add = [&](const float *src) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += src[((i * 2 + j) * 16 + e) * 64 + lane];
  };
  if (wave >= 2) put(s_acc[wave - 2]);
  __syncthreads();
  if (wave < 2) add(s_acc[wave]);
  __syncthreads();
  if (wave == 1) put(s_acc[0]);
  __syncthreads();
  if (wave == 0) add(s_acc[0]);

  // C/D layout of 32x32: col (b operand) = lane & 31, row (a operand) = (e&3) + 8(e>>2) + 4 lk
  const size_t tile_elems = (size_t)Cout * Cin;
  auto store_tile = [&](float *dst, long long ld) {   // wave 0 only
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = co0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lk;
          const int ci = ci0 + 32 * j + li;
          if (co < Cout && ci < Cin) dst[(long long)co * ld + ci] = acc[i][j][e];
        }
  };
  if (nslab == 1) {
    if (wave == 0) store_tile(dW, lddw);
    return;
  }
  if (wave == 0) store_tile(part1 + (size_t)slab * tile_elems, Cin);
  if (cnt == nullptr) return;        // partials only: the caller reduces them

  // ---- level 1: the last workgroup of a group of DW_GROUP slabs adds them up ------------
  const int ngroups = (nslab + DW_GROUP - 1) / DW_GROUP;
  const int group = slab / DW_GROUP;
  const int g0 = group * DW_GROUP;
  const int gsize = min(DW_GROUP, nslab - g0);
  unsigned *cnt1 = cnt + (size_t)t * (ngroups + 1);
  unsigned *cnt2 = cnt1 + ngroups;
  __threadfence();
  __syncthreads();
  if (tid == 0)
    s_ticket = __hip_atomic_fetch_add(cnt1 + group, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_ticket != (unsigned)(gsize - 1)) return;
  __threadfence();
  if (tid == 0) cnt1[group] = 0;                       // leave the workspace clean
  float *dst1 = (ngroups == 1) ? dW : part2 + (size_t)group * tile_elems;
  const long long ld1 = (ngroups == 1) ? lddw : Cin;
  // 64 x 64 tile, 256 threads: thread -> (row = tid / 4 + 0..., 16 columns)
  for (int idx = tid; idx < 64 * 64; idx += 256) {
    const int co = co0 + idx / 64, ci = ci0 + (idx & 63);
    if (co >= Cout || ci >= Cin) continue;
    float s = 0.f;
    for (int k = 0; k < gsize; ++k)
      s += part1[(size_t)(g0 + k) * tile_elems + (size_t)co * Cin + ci];
    dst1[(long long)co * ld1 + ci] = s;
  }
  if (ngroups == 1) return;

  // ---- level 2: the last group adds the group partials ---------------------------------
  __threadfence();
  __syncthreads();
  if (tid == 0)
    s_ticket = __hip_atomic_fetch_add(cnt2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_ticket != (unsigned)(ngroups - 1)) return;
  __threadfence();
  if (tid == 0) *cnt2 = 0;
  for (int idx = tid; idx < 64 * 64; idx += 256) {
    const int co = co0 + idx / 64, ci = ci0 + (idx & 63);
    if (co >= Cout || ci >= Cin) continue;
    float s = 0.f;
    for (int k = 0; k < ngroups; ++k)
      s += part2[(size_t)k * tile_elems + (size_t)co * Cin + ci];
    dW[(long long)co * lddw + ci] = s;
  }
}

__global__ __launch_bounds__(256) void dw_x3_kernel(
    long long M, int Cout, int Cin, const float *__restrict__ dY, long long ldy,
    const float *__restrict__ A, long long lda, int rows_per_slab, int nslab, int ntci,
    int ntiles, float *__restrict__ part1, float *__restrict__ part2,
    unsigned *__restrict__ cnt, float *__restrict__ dW, int lddw) {
  dw_x3_body(blockIdx.x, M, Cout, Cin, dY, ldy, A, lda, rows_per_slab, nslab, ntci, ntiles, part1,
             part2, cnt, dW, lddw);
}

// Several independent weight gradients in ONE launch (the layers of a stack: 17 us launches that
// each fill a fraction of the chip): job j owns the blocks [first[j], first[j + 1]); partial tiles
// only (the caller's multi_colsum launch adds the slabs), one slab: straight into dW.
struct DwJobPlan { int first, rows_per_slab, nslab, ntci, ntiles; };
struct DwMulti {
  s2c_dw_jobs j;
  DwJobPlan plan[S2C_DW_MAX_JOBS];
};
__global__ __launch_bounds__(256) void dw_x3_multi_kernel(DwMulti a) {
  int j = 0;
  const int bid = blockIdx.x;
#pragma unroll 1
  while (j + 1 < a.j.n_jobs && bid >= a.plan[j + 1].first) ++j;
  const s2c_dw_job &jb = a.j.job[j];
  const DwJobPlan &pl = a.plan[j];
  dw_x3_body(bid - pl.first, jb.M, jb.Cout, jb.Cin, jb.dY, jb.ldy, jb.A, jb.lda, pl.rows_per_slab,
             pl.nslab, pl.ntci, pl.ntiles, jb.part, nullptr, nullptr, jb.dW, jb.lddw);
}

int chk6(const char *k) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c: %s launch failed: %s\n", k, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

void dw_plan(long long M, int Cout, int Cin, int *rows_per_slab, int *nslab, int *ntiles) {
  const int nt = ((Cout + 63) / 64) * ((Cin + 63) / 64);
  // aim at ~512 workgroups, slabs of 256 .. 4096 rows (multiples of 64)
  long long want = (512 + nt - 1) / nt;
  long long rps = (M + want - 1) / want;
  rps = (rps + 63) / 64 * 64;
  if (rps < 256) rps = 256;
  if (rps > 4096) rps = 4096;
  *rows_per_slab = (int)rps;
  *nslab = (int)((M + rps - 1) / rps);
  *ntiles = nt;
}

}  // namespace

extern "C" long long s2c_weight_grad_workspace_bytes(long long M, int Cout, int Cin) {
  if (M <= 0 || Cout <= 0 || Cin <= 0) return 0;
  int rps, nslab, nt;
  dw_plan(M, Cout, Cin, &rps, &nslab, &nt);
  const long long ngroups = (nslab + DW_GROUP - 1) / DW_GROUP;
  return 4ll * ((long long)nslab + ngroups) * Cout * Cin;
}

// number of row slabs = partial (Cout x Cin) tiles the kernel writes into `workspace` when
// called with counters == NULL (the caller adds them up, e.g. s2c_multi_colsum)
extern "C" int s2c_weight_grad_slabs(long long M, int Cout, int Cin) {
  if (M <= 0 || Cout <= 0 || Cin <= 0) return 0;
  int rps, nslab, nt;
  dw_plan(M, Cout, Cin, &rps, &nslab, &nt);
  return nslab;
}

extern "C" long long s2c_weight_grad_counter_bytes(long long M, int Cout, int Cin) {
  if (M <= 0 || Cout <= 0 || Cin <= 0) return 0;
  int rps, nslab, nt;
  dw_plan(M, Cout, Cin, &rps, &nslab, &nt);
  const long long ngroups = (nslab + DW_GROUP - 1) / DW_GROUP;
  return 4ll * nt * (ngroups + 1);
}

extern "C" int s2c_weight_grad(long long M, int Cout, int Cin, const float *dY, long long ldy,
                               const float *A, long long lda, float *dW, int lddw,
                               void *workspace, void *counters, void *stream) {
  if (M <= 0 || Cout <= 0 || Cin <= 0 || !dY || !A || !dW || ldy < Cout || lda < Cin ||
      lddw < Cin)
    return -1;
  int rps, nslab, nt;
  dw_plan(M, Cout, Cin, &rps, &nslab, &nt);
  if (nslab > 1 && !workspace) return -1;
  const long long ngroups = (nslab + DW_GROUP - 1) / DW_GROUP;
  float *part1 = (float *)workspace;
  float *part2 = part1 ? part1 + (size_t)nslab * Cout * Cin : nullptr;
  (void)ngroups;
  const int nblocks = 8 * nt * ((nslab + 7) / 8);
  hipLaunchKernelGGL(dw_x3_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, M, Cout,
                     Cin, dY, ldy, A, lda, rps, nslab, (Cin + 63) / 64, nt, part1, part2,
                     (unsigned *)counters, dW, lddw);
  return chk6("weight_grad");
}

// The weight gradients of several layers in one launch (at most S2C_DW_MAX_JOBS): job j writes
// s2c_weight_grad_slabs(M, Cout, Cin) partial tiles into job[j].part (one slab: dW itself, part may
// be NULL); the caller adds the slabs up (s2c_multi_colsum).
extern "C" int s2c_weight_grad_multi(const s2c_dw_jobs *jobs, void *stream) {
  if (!jobs || jobs->n_jobs <= 0 || jobs->n_jobs > S2C_DW_MAX_JOBS) return -1;
  DwMulti a;
  a.j = *jobs;
  int first = 0;
  for (int j = 0; j < jobs->n_jobs; ++j) {
    const s2c_dw_job &jb = jobs->job[j];
    if (jb.M <= 0 || jb.Cout <= 0 || jb.Cin <= 0 || !jb.dY || !jb.A || !jb.dW || jb.ldy < jb.Cout ||
        jb.lda < jb.Cin || jb.lddw < jb.Cin)
      return -1;
    int rps, nslab, nt;
    dw_plan(jb.M, jb.Cout, jb.Cin, &rps, &nslab, &nt);
    if (nslab > 1 && !jb.part) return -1;
    a.plan[j].first = first;
    a.plan[j].rows_per_slab = rps;
    a.plan[j].nslab = nslab;
    a.plan[j].ntci = (jb.Cin + 63) / 64;
    a.plan[j].ntiles = nt;
    first += 8 * nt * ((nslab + 7) / 8);
  }
  hipLaunchKernelGGL(dw_x3_multi_kernel, dim3(first), dim3(256), 0, (hipStream_t)stream, a);
  return chk6("weight_grad_multi");
}
