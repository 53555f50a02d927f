// s2c_dw32.hip -- weight gradient of a tall rows x channels layer on the fp32 matrix cores,
//     dW[c, k] = sum_m dY[m, c] * X[m, k],      M = 32768 .. 1M rows, C x K <= 256 x 288,
// (backward of lib/pointnet2/pytorch_utils.py:11-120's 1x1 convs; pointnet2/fused.py::_MLPRows),
// with X optionally the GATHERED operand of a set-abstraction stage's first layer
//     X[(b, j, s), :] = [ (xyz[b, idx] - new_xyz[b, j]) (/ radius) | feats[b, idx, :] ]
// (pointnet2_utils.py:347-359) read in place: the backward no longer re-materialises the
// (rows x (3 + C)) tensor (`s2c_sa_gather_rows` + a write and a read of it per stage).
//
// Why the fp32 MFMA.  The reduction index m is the ROW index of both operands.  The bf16
// matrix instructions want 8 consecutive reduction elements per lane -- 8 rows of one column,
// i.e. a transposed read (s2c_dw.hip: eight 4-byte loads per operand and a VALU split to bf16x3
// planes per 16 rows; a library GEMM: transposing tile loads).  `v_mfma_f32_32x32x2_f32` takes
// ONE fp32 per lane and operand, lane = (i or j = lane & 31, k = lane >> 5): lanes 0-31 hold 32
// consecutive columns of row 2p, lanes 32-63 of row 2p + 1 -- exactly what a coalesced load of two
// 128-byte row segments of a row-major matrix delivers.  No LDS, no transposition, no split, exact
// fp32 products (an fp32 FMA chain in row order), and the matrix pipe is not the limit: 64 cycles
// per 2 rows x (32 x 32) -- the kernel needs 2 M C K flops at 157 TF against (C + K) 4 M bytes of
// HBM, i.e. it is HBM-bound up to C K / (C + K) ~ 40 (64 x 64: 32, 128 x 128: 64 -> matrix-bound).
//
// Wave = a 64 x 64 block of dW (2 x 2 accumulator tiles), workgroup = all blocks of dW x RS row
// splits; a workgroup walks a slab of rows in pairs, eight pairs of operand loads in flight per
// wave; row splits meet in LDS (ds_add_f32); one partial (C x K) per workgroup, summed by the
// caller's multi_colsum launch (a kernel boundary, not an in-kernel hand-off: DESIGN 4.3).
//
// STATUS (round 4): correct (tests/test_dw32_gpu.py) and NOT the default (S2C_DW32=1 opts in):
// measured 171 us against the library's 104 us at 1M x 64 x 64, 179 against 80 at 262144 x 128 x
// 128 (tools/bench_dw32.py).  One 4-byte load per lane and operand tile moves 256 B per wave
// instruction; with 4 such loads per 4 MFMAs the wave's issue slots, not HBM, bound it (3.1 TB/s).
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int D32_PD = 8;          // row pairs of operand loads in flight per wave

struct D32Args {
  long long M;
  int C, K;
  const float *dY; long long ldy;
  const float *X; long long ldx;         // dense operand (gather.idx == nullptr)
  s2c_dw_gather g;
  float *part;                           // (slabs, C, K)
  long long rows_per_slab;               // a multiple of 2 * RS
  int nbc, nbk, RS;                      // 64-wide blocks of C and of K, row splits
};

template <bool GATHER>
__global__ __launch_bounds__(768) void dw32_kernel(D32Args a) {
  extern __shared__ float d32_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = a.nbc * a.nbk;
  const int blk = wave % nblk, rs = wave / nblk;        // output block, row split
  const int bc = blk / a.nbk, bk = blk % a.nbk;
  const int li = lane & 31, lk = lane >> 5;
  const long long s0 = (long long)blockIdx.x * a.rows_per_slab;
  long long s1 = s0 + a.rows_per_slab;
  if (s1 > a.M) s1 = a.M;

  // the lane's columns: tile t of the block = columns 64 b + 32 t + li
  int cc[2], kk[2];
  bool cok[2], kok[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    cc[t] = 64 * bc + 32 * t + li; cok[t] = cc[t] < a.C;
    kk[t] = 64 * bk + 32 * t + li; kok[t] = kk[t] < a.K;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // row pair q of this wave: rows s0 + 2 (rs + RS q) + lk.  nfull pairs lie entirely inside the slab
  // (both rows): they run unmasked; at most one more pair has only its first row inside.
  const long long step = 2LL * a.RS;
  const long long first = s0 + 2LL * rs + lk;
  const long long span = s1 - s0 - 2LL * rs;                       // rows from the wave's first pair on
  const int npairs = span > 0 ? (int)((span + step - 1) / step) : 0;
  const int nfull = span > 1 ? (int)((span - 2) / step) + 1 : 0;   // pairs whose second row is < s1

  // Every load of the main loop is UNCONDITIONAL and its address a running pointer (one 64-bit add
  // per operand tile and pair).  A first version recomputed row * ld (64-bit multiplies), clamped
  // rows and selected zeros per pair: its loop WITHOUT loads and products already took 100 us at
  // 1M x 64 x 64 (S2C_DW32_DBG=3) -- ~150 VALU instructions per pair against 4 MFMAs.  Columns
  // beyond C / K read a valid column instead: they only reach accumulator rows / columns that are
  // never stored.
  int ccl[2], kkl[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    ccl[t] = cok[t] ? cc[t] : a.C - 1;
    kkl[t] = kok[t] ? kk[t] : a.K - 1;
  }
  const long long m0w = first < a.M ? first : a.M - 1;             // (an empty wave never loads)
  const float *pa[2] = {a.dY + m0w * a.ldy + ccl[0], a.dY + m0w * a.ldy + ccl[1]};
  const float *pb[2] = {a.X, a.X};
  if (!GATHER) { pb[0] = a.X + m0w * a.ldx + kkl[0]; pb[1] = a.X + m0w * a.ldx + kkl[1]; }
  const long long inc_a = step * a.ldy, inc_b = step * a.ldx;
  // gather: running (row, centre, scene) state instead of divisions per row
  long long m_cur = m0w;
  int bj = 0, bsc = 0, r_in = 0, j_in = 0;      // centre index m / ns, scene, m % ns, centre % g.m
  if (GATHER) {
    bj = (int)(m_cur / a.g.ns); r_in = (int)(m_cur % a.g.ns);
    bsc = bj / a.g.m; j_in = bj % a.g.m;
  }
  auto load = [&](float (&A)[2], float (&B)[2]) {
    A[0] = *pa[0]; A[1] = *pa[1];
    pa[0] += inc_a; pa[1] += inc_a;
    if (!GATHER) {
      B[0] = *pb[0]; B[1] = *pb[1];
      pb[0] += inc_b; pb[1] += inc_b;
    } else {
      const int p = a.g.idx[m_cur];
      const float *f = a.g.feats + (long long)bsc * a.g.fbs + (long long)p * a.g.frs;
      const float *px = a.g.xyz + ((long long)bsc * a.g.n + p) * 3;
      const float *pc = a.g.new_xyz + (long long)bj * 3;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int k = kkl[t];
        if (k >= 3) {                               // (lane-constant over the loop)
          B[t] = f[k - 3];
        } else {
          float vx = px[k] - pc[k];
          if (a.g.normalize) vx = vx / a.g.radius;
          B[t] = vx;
        }
      }
      m_cur += step;
      r_in += (int)step;
      while (r_in >= a.g.ns) { r_in -= a.g.ns; ++bj; ++j_in; }
      while (j_in >= a.g.m) { j_in -= a.g.m; ++bsc; }
    }
  };
  auto mma = [&](float a0, float a1, float b0, float b1) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  };

  // Steady state without a single branch: groups of D32_PD full pairs, every slot consumed and
  // refilled; the last group drains; what is left (< D32_PD pairs) runs one pair at a time.
  float ra[D32_PD][2], rb[D32_PD][2];
  const int ngroups = nfull / D32_PD;
  if (ngroups > 0) {
#pragma unroll
    for (int d = 0; d < D32_PD; ++d) load(ra[d], rb[d]);
    for (int g = 0; g + 1 < ngroups; ++g) {
#pragma unroll
      for (int d = 0; d < D32_PD; ++d) {
        const float a0 = ra[d][0], a1 = ra[d][1], b0 = rb[d][0], b1 = rb[d][1];
        load(ra[d], rb[d]);                             // refill the slot: D32_PD pairs ahead
        mma(a0, a1, b0, b1);
      }
    }
#pragma unroll
    for (int d = 0; d < D32_PD; ++d) mma(ra[d][0], ra[d][1], rb[d][0], rb[d][1]);
  }
  for (int r = ngroups * D32_PD; r < nfull; ++r) {
    float ta[2], tb[2];
    load(ta, tb);
    mma(ta[0], ta[1], tb[0], tb[1]);
  }
  if (npairs > nfull) {
    // the slab's last row, alone in its pair: the lanes of the missing second row re-read the first
    // row (a valid address) and contribute zeros
    if (lk) {
      pa[0] -= a.ldy; pa[1] -= a.ldy;
      if (!GATHER) { pb[0] -= a.ldx; pb[1] -= a.ldx; } else { m_cur -= 1; r_in -= 1; if (r_in < 0) { r_in += a.g.ns; --bj; if (--j_in < 0) { j_in += a.g.m; --bsc; } } }
    }
    float ta[2], tb[2];
    load(ta, tb);
    mma(lk ? 0.f : ta[0], lk ? 0.f : ta[1], lk ? 0.f : tb[0], lk ? 0.f : tb[1]);
  }

  // ---- row splits meet in LDS, one partial per workgroup ----
  // C/D layout: column j = lane & 31 (X column), row i = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (dY column)
  float *part = a.part + (long long)blockIdx.x * a.C * a.K;
  if (a.RS > 1) {
    float *tile = d32_lds + (size_t)blk * 4096;          // 64 x 64 floats per output block
    for (int i = tid; i < nblk * 4096; i += blockDim.x) d32_lds[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * lk, j = 32 * tj + li;
          atomicAdd(&tile[i * 64 + j], acc[ti][tj][e]);
        }
    __syncthreads();
    if (rs == 0) {
      for (int x = lane; x < 4096; x += 64) {
        const int c = 64 * bc + (x >> 6), k = 64 * bk + (x & 63);
        if (c < a.C && k < a.K) part[(long long)c * a.K + k] = tile[x];
      }
    }
  } else {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int c = 64 * bc + 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * lk;
          const int k = 64 * bk + 32 * tj + li;
          if (c < a.C && k < a.K) part[(long long)c * a.K + k] = acc[ti][tj][e];
        }
  }
}

struct D32Plan { int nbc, nbk, RS, waves, nslab; long long rows_per_slab; };

D32Plan d32_plan(long long M, int C, int K) {
  D32Plan p;
  p.nbc = (C + 63) / 64;
  p.nbk = (K + 63) / 64;
  const int nblk = p.nbc * p.nbk;
  p.RS = nblk >= 5 ? 1 : 8 / nblk;                       // 6 .. 12 waves per workgroup
  p.waves = nblk * p.RS;
  // ~2 waves per SIMD over the chip, at least 256 rows per slab (partials stay small)
  long long want = (2048 + p.waves - 1) / p.waves;
  long long rows = (M + want - 1) / want;
  if (rows < 256) rows = 256;
  const long long gran = 2LL * p.RS;
  rows = (rows + gran - 1) / gran * gran;
  p.rows_per_slab = rows;
  p.nslab = (int)((M + rows - 1) / rows);
  return p;
}

}  // namespace

extern "C" int s2c_weight_grad_f32_slabs(long long M, int C, int K) {
  if (M <= 0 || C <= 0 || K <= 0 || ((C + 63) / 64) * ((K + 63) / 64) > 12) return -1;
  return d32_plan(M, C, K).nslab;
}

extern "C" int s2c_weight_grad_f32(long long M, int C, int K, const float *dY, long long ldy,
                                   const float *X, long long ldx, const s2c_dw_gather *g,
                                   float *part, void *stream) {
  if (M <= 0 || C <= 0 || K <= 0 || dY == nullptr || part == nullptr ||
      ((C + 63) / 64) * ((K + 63) / 64) > 12)
    return -1;
  const bool gather = g != nullptr && g->idx != nullptr;
  if (!gather && X == nullptr) return -1;
  if (gather && (g->xyz == nullptr || g->new_xyz == nullptr || g->ns <= 0 || g->m <= 0 ||
                 (K > 3 && g->feats == nullptr)))
    return -1;
  const D32Plan p = d32_plan(M, C, K);
  D32Args a;
  a.M = M; a.C = C; a.K = K; a.dY = dY; a.ldy = ldy; a.X = X; a.ldx = ldx;
  if (gather) a.g = *g; else { a.g = s2c_dw_gather(); a.g.idx = nullptr; }
  a.part = part; a.rows_per_slab = p.rows_per_slab; a.nbc = p.nbc; a.nbk = p.nbk; a.RS = p.RS;
  const size_t lds = p.RS > 1 ? (size_t)p.nbc * p.nbk * 4096 * sizeof(float) : 0;
  static int attr_done[2];
  if (lds > 48 * 1024 && !attr_done[gather]) {
    const void *fn = gather ? (const void *)dw32_kernel<true> : (const void *)dw32_kernel<false>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
      return -3;
    attr_done[gather] = 1;
  }
  if (gather)
    hipLaunchKernelGGL(dw32_kernel<true>, dim3(p.nslab), dim3(64 * p.waves), lds, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(dw32_kernel<false>, dim3(p.nslab), dim3(64 * p.waves), lds, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c: weight_grad_f32 launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
