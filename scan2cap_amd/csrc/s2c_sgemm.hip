// s2c_sgemm.hip -- the SMALL products of the layer stacks (2048 .. 32768 rows, K, N <= 512):
//
//     Y[M x N] = A[M x K] B (+ bias),   B = W (K x N, row-major)          -- input gradients dX = dY W
//                                       B = W^T, W (N x K, row-major)      -- forward y = x W^T + b
//
// (reference: the FP / vote / proposal-head / SA3-SA4 layers of lib/pointnet2/pytorch_utils.py:67-120
// and their autograd; models/voting_module.py:40-56, models/proposal_module.py:48-77.)
//
// These launches last 10-25 us in the library (fp32 matrix instruction, 13 dX products + 2 biased
// forward layers per cfg3 step, tools/lib_gemm_census.py) and the tall-layer kernels of s2c_gemm*.hip
// lose to it here: 128-row tiles leave most of the 256 CUs idle at 2048-8192 rows and a transposed
// copy of W preceded every dX.  This kernel is built for the shape:
//   * 64 x 64 output tiles, one workgroup of 4 waves per tile, two workgroups per CU (64 KB of LDS
//     each): 512 tiles at (8192, 256) fill the chip in one round;
//   * the four waves split K (k-steps of 16 dealt round-robin), each holds the whole 64 x 64 tile in
//     4 accumulators -- every operand fragment is split to bf16 planes exactly once -- and they meet
//     through LDS at the end (coalesced row stores, bias added there);
//   * both operand tiles of a 128-deep stage land in LDS by LDS-DMA (no VGPR staging): the A tile
//     (and W in the forward form) row-major with the 16-byte pieces of row r XOR-swizzled by r & 31 on
//     the SOURCE address, so a lane-per-row `ds_read_b128` fragment read is conflict-free; W in the
//     dX form stays row-major (K x 64) and is read DOWN its columns (`ds_read2_b32`): the transposed
//     operand needs no transposed copy;
//   * fp32 -> 3 x bf16 planes by truncation, 6 plane products with i + j <= 2 (fp32-accurate,
//     s2c_dwstream.hip); the products lag one k-step behind the splits, MFMA and VALU interleaved in
//     program order;
//   * K need not be a multiple of 4 (259 = 3 + 256 after a concatenation, 97): the ragged columns
//     are filled by guarded loads, the rest of the stage by zeros.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 128;                       // k columns per stage
constexpr int TILE_BYTES = 64 * KC * 4;       // one operand tile of a stage (32 KB)

// index -> element offset: (i / div) * hi + (i % div) * lo (div == 0: i * lo) -- a row of a (T,R,.)
// tensor addressed by (r, t), a plain row stride, ...
struct SgMap { int div; int hi, lo; };
__device__ __forceinline__ unsigned sg_off(const SgMap &m, long long i) {
  if (m.div == 0) return (unsigned)((int)i * m.lo);
  const int q = (int)(i / m.div);
  return (unsigned)(q * m.hi + ((int)i - q * m.div) * m.lo);
}

struct SgArgs {
  long long M;
  int N, K;
  const float *A, *B;
  float *Y;
  const float *bias;
  SgMap arow, brow, crow;   // rows of A (m; AT: k), of B (BT: n; else k), of Y (m)
  int ntn;                  // column tiles
  int ksplit;               // > 1: the k-stages are dealt to gridDim.y workgroups, Y += by float atomics
};

struct Planes { bf16x8 p[3]; };

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
  const unsigned b0 = __builtin_bit_cast(unsigned, x0), b1 = __builtin_bit_cast(unsigned, x1);
  const float r0 = x0 - __builtin_bit_cast(float, b0 & 0xffff0000u);
  const float r1 = x1 - __builtin_bit_cast(float, b1 & 0xffff0000u);
  const unsigned c0 = __builtin_bit_cast(unsigned, r0), c1 = __builtin_bit_cast(unsigned, r1);
  const float s0 = r0 - __builtin_bit_cast(float, c0 & 0xffff0000u);
  const float s1 = r1 - __builtin_bit_cast(float, c1 & 0xffff0000u);
  h = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
  m = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
  l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
}

__device__ __forceinline__ void glds16s(const void *sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// BT: B is W (N x K) row-major, Y = A W^T (forward); else B is (K x N) row-major.
// AT: A is given k-major, (K x M) row-major (Y = A^T B: a weight gradient with a short reduction);
//     its tile is staged like the (K x 64) B tile and read down its columns.  (AT && BT: not built.)
template <bool AT, bool BT>
__global__ __launch_bounds__(256, 2) void sgemm_kernel(SgArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *sA = reinterpret_cast<float *>(smem);                     // [64 rows][KC] swizzled; AT: [KC][64]
  float *sB = reinterpret_cast<float *>(smem + TILE_BYTES);        // BT: [64 n][KC] swizzled; else [KC][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const long long M = p.M;
  const int N = p.N, K = p.K;
  // tiles of one row block run back to back (shared A rows stay in L2)
  const int tn = blockIdx.x % p.ntn;
  const long long m0 = (long long)(blockIdx.x / p.ntn) * 64;
  const int n0 = tn * 64;
  const int Kq = K & ~3;                     // whole 16-byte pieces
  const unsigned smem_lds = (unsigned)(size_t)smem;

  // ---- this lane's DMA pieces: 8 instructions per operand tile and wave ---------------------
  // row-major swizzled tile: piece P = row * 32 + pos holds the row's logical quad pos ^ (row & 31)
  // k-major tile [KC][64]: piece P = k * 16 + c: row k of the stage, columns 4c .. 4c + 3 of the tile
  unsigned a_row[8], a_q[8];                 // byte offset of the source row | stage row, logical quad | column bytes
  unsigned b_row[8], b_q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int P = (wave + 4 * j) * 64 + lane;
    const int r = P >> 5, pos = P & 31;
    const int kk = P >> 4, c = P & 15;
    if (AT) {
      a_row[j] = (unsigned)kk;
      a_q[j] = (unsigned)((m0 + 4 * c < M ? (int)m0 + 4 * c : 0) * 4);
    } else {
      long long row = m0 + r;
      if (row >= M) row = M - 1;
      a_row[j] = sg_off(p.arow, row) * 4u;
      a_q[j] = (unsigned)(pos ^ (r & 31));
    }
    if (BT) {
      int n = n0 + r;
      if (n >= N) n = N - 1;
      b_row[j] = sg_off(p.brow, n) * 4u;
      b_q[j] = (unsigned)(pos ^ (r & 31));
    } else {
      b_row[j] = (unsigned)kk;               // stage row; the byte offset depends on the stage
      b_q[j] = (unsigned)((n0 + 4 * c < N ? n0 + 4 * c : 0) * 4);
    }
  }
  const float *Abase = p.A, *Bbase = p.B;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
  Planes ca[2], cb[2];
  bool have = false;

  auto products = [&]() {
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[i].p[TA[q]], cb[j].p[TB[q]],
                                                              acc[i][j], 0, 0, 0);
  };
  // operands of k-step s of the staged tiles -> registers
  auto read_raw = [&](int s, float (&va)[2][8], float (&vb)[2][8]) {
    const int q0 = 4 * s + 2 * lk;
    if (AT) {
      const float *a0 = sA + (16 * s + 8 * lk) * 64 + li;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        va[0][r] = a0[r * 64];
        va[1][r] = a0[r * 64 + 32];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float *row = sA + (i * 32 + li) * KC;
        const float4 x = *reinterpret_cast<const float4 *>(row + ((q0 ^ li) << 2));
        const float4 y = *reinterpret_cast<const float4 *>(row + (((q0 + 1) ^ li) << 2));
        va[i][0] = x.x; va[i][1] = x.y; va[i][2] = x.z; va[i][3] = x.w;
        va[i][4] = y.x; va[i][5] = y.y; va[i][6] = y.z; va[i][7] = y.w;
      }
    }
    if (BT) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float *row = sB + (j * 32 + li) * KC;
        const float4 x = *reinterpret_cast<const float4 *>(row + ((q0 ^ li) << 2));
        const float4 y = *reinterpret_cast<const float4 *>(row + (((q0 + 1) ^ li) << 2));
        vb[j][0] = x.x; vb[j][1] = x.y; vb[j][2] = x.z; vb[j][3] = x.w;
        vb[j][4] = y.x; vb[j][5] = y.y; vb[j][6] = y.z; vb[j][7] = y.w;
      }
    } else {
      const float *b0 = sB + (16 * s + 8 * lk) * 64 + li;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        vb[0][r] = b0[r * 64];
        vb[1][r] = b0[r * 64 + 32];
      }
    }
  };
  auto set_planes = [&](const u32x4 (&nh)[4], const u32x4 (&nm)[4], const u32x4 (&nl)[4]) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      ca[f].p[0] = __builtin_bit_cast(bf16x8, nh[f]);
      ca[f].p[1] = __builtin_bit_cast(bf16x8, nm[f]);
      ca[f].p[2] = __builtin_bit_cast(bf16x8, nl[f]);
      cb[f].p[0] = __builtin_bit_cast(bf16x8, nh[2 + f]);
      cb[f].p[1] = __builtin_bit_cast(bf16x8, nm[2 + f]);
      cb[f].p[2] = __builtin_bit_cast(bf16x8, nl[2 + f]);
    }
  };
#define S2C_SG_PICK(u, d) ((u) == 0 ? va[0][d] : (u) == 1 ? va[1][d] : (u) == 2 ? vb[0][d] : vb[1][d])
  // the first k-step of a wave: split only
  auto step_first = [&](int s) {
    float va[2][8], vb[2][8];
    read_raw(s, va, vb);
    u32x4 nh[4], nm[4], nl[4];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int f = u >> 2, d = u & 3;
      unsigned h, m, l;
      split_pair(S2C_SG_PICK(f, 2 * d), S2C_SG_PICK(f, 2 * d + 1), h, m, l);
      nh[f][d] = h; nm[f][d] = m; nl[f][d] = l;
    }
    set_planes(nh, nm, nl);
  };
  // every later one: its splits interleaved with the previous k-step's 24 MFMAs (program order =
  // issue order, pinned by sched_barrier; the first products cover the LDS round trip)
  auto step_lag = [&](int s) {
    float va[2][8], vb[2][8];
    read_raw(s, va, vb);
    u32x4 nh[4], nm[4], nl[4];
#pragma unroll
    for (int t = 0; t < 24; ++t) {
      const int q = t >> 2, i = (t >> 1) & 1, j = t & 1;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[i].p[TA[q]], cb[j].p[TB[q]],
                                                          acc[i][j], 0, 0, 0);
      if (t >= 6 && t < 22) {
        const int f = (t - 6) >> 2, d = (t - 6) & 3;
        unsigned h, m, l;
        split_pair(S2C_SG_PICK(f, 2 * d), S2C_SG_PICK(f, 2 * d + 1), h, m, l);
        asm volatile("" : "+v"(h), "+v"(m), "+v"(l));
        nh[f][d] = h; nm[f][d] = m; nl[f][d] = l;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    set_planes(nh, nm, nl);
  };
#undef S2C_SG_PICK

  const int nstage = (K + KC - 1) / KC;
  // ksplit > 1: this workgroup's share of the stages
  const int per = (nstage + (p.ksplit > 1 ? p.ksplit : 1) - 1) / (p.ksplit > 1 ? p.ksplit : 1);
  const int st_lo = p.ksplit > 1 ? (int)blockIdx.y * per : 0;
  const int st_hi = st_lo + per < nstage ? st_lo + per : nstage;
#pragma unroll 1
  for (int st = st_lo; st < st_hi; ++st) {
    const int kb = st * KC;
    if (st > st_lo) __syncthreads();         // the previous stage's operands are in registers
    // ---- DMA: pieces wholly inside the operand come from memory, the others from a valid dummy address ----
    {
      const unsigned dstA = smem_lds + (unsigned)wave * 1024u;
      const unsigned dstB = smem_lds + (unsigned)TILE_BYTES + (unsigned)wave * 1024u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (AT) {
          const unsigned k = (unsigned)kb + a_row[j];
          glds16s(Abase, sg_off(p.arow, k < (unsigned)K ? k : 0u) * 4u + a_q[j], dstA + (unsigned)j * 4096u);
        } else {
          const unsigned k = (unsigned)kb + 4u * a_q[j];
          glds16s(Abase, a_row[j] + (k < (unsigned)Kq ? k * 4u : 0u), dstA + (unsigned)j * 4096u);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (BT) {
          const unsigned k = (unsigned)kb + 4u * b_q[j];
          glds16s(Bbase, b_row[j] + (k < (unsigned)Kq ? k * 4u : 0u), dstB + (unsigned)j * 4096u);
        } else {
          const unsigned k = (unsigned)kb + b_row[j];
          glds16s(Bbase, sg_off(p.brow, k < (unsigned)K ? k : 0u) * 4u + b_q[j], dstB + (unsigned)j * 4096u);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- ragged K: columns [Kq, K) by guarded loads, [K, stage end) zeros -----------------------
    const int kend = kb + KC;
    if (kend > Kq) {
      __syncthreads();                       // every wave's dummy pieces have landed before they are fixed
      const int k_lo = Kq > kb ? Kq : kb;
      const int span = kend - k_lo;          // columns to fix in this stage
      if (!AT || BT)
        for (int e = tid; e < 64 * span; e += 256) {
          const int r = e / span, k = k_lo + (e - r * span);
          const int q = (k - kb) >> 2, el = (k - kb) & 3;
          if (!AT) {
            long long row = m0 + r;
            if (row >= M) row = M - 1;
            sA[r * KC + (((q ^ (r & 31)) << 2) | el)] = k < K ? p.A[sg_off(p.arow, row) + k] : 0.f;
          }
          if (BT) {
            int n = n0 + r;
            if (n >= N) n = N - 1;
            sB[r * KC + (((q ^ (r & 31)) << 2) | el)] = k < K ? p.B[sg_off(p.brow, n) + k] : 0.f;
          }
        }
      if (kend > K) {                        // rows k >= K of the k-major tiles
        const int r_lo = K > kb ? K - kb : 0;
        if (AT) for (int e = tid; e < (KC - r_lo) * 64; e += 256) sA[r_lo * 64 + e] = 0.f;
        if (!BT) for (int e = tid; e < (KC - r_lo) * 64; e += 256) sB[r_lo * 64 + e] = 0.f;
      }
    }
    __syncthreads();
    const int kvalid = K - kb < KC ? K - kb : KC;
    const int nks = (kvalid + 15) >> 4;
#pragma unroll 1
    for (int s = wave; s < nks; s += 4) {
      if (have) step_lag(s);
      else { step_first(s); have = true; }
    }
  }
  if (have) products();                      // the lagging product

  // ---- the four waves meet in LDS; rows leave coalesced ------------------------------------------
  __syncthreads();
  float *red = reinterpret_cast<float *>(smem);
  {
    float *dst = red + (size_t)wave * 4096;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[((i * 2 + j) * 16 + e) * 64 + lane] = acc[i][j][e];
  }
  __syncthreads();
  {
    const int col = tid & 63;
    const bool colok = n0 + col < N;
    const float bv = (p.bias != nullptr && colok) ? p.bias[n0 + col] : 0.f;
    const int j = col >> 5, ln_lo = col & 31;
#pragma unroll 4
    for (int t = 0; t < 16; ++t) {
      const int row = (tid >> 6) + 4 * t;
      // C/D layout of 32x32: row = (e & 3) + 8 (e >> 2) + 4 (ln >> 5), col = ln & 31
      const int i = row >> 5, rr = row & 31;
      const int e = ((rr >> 3) << 2) | (rr & 3), h = (rr >> 2) & 1;
      const int idx = ((i * 2 + j) * 16 + e) * 64 + h * 32 + ln_lo;
      float v = red[idx] + red[4096 + idx] + red[8192 + idx] + red[12288 + idx];
      if (colok && m0 + row < M) {
        float *dst = p.Y + sg_off(p.crow, m0 + row) + n0 + col;
        if (p.ksplit > 1) atomicAdd(dst, blockIdx.y == 0 ? v + bv : v);
        else *dst = v + bv;
      }
    }
  }
}

}  // namespace

static unsigned long long map_reach(const SgMap &m, long long n) {
  if (n <= 0) return 0;
  if (m.div == 0) return (unsigned long long)(n - 1) * (unsigned long long)(m.lo < 0 ? 0 : m.lo);
  const long long q = (n - 1) / m.div;
  const long long r = (n - 1 < m.div - 1) ? n - 1 : m.div - 1;
  return (unsigned long long)(q * m.hi + r * m.lo);
}

static int launch_sgemm(const SgArgs &a, int at, int bt, hipStream_t st) {
  const long long tiles = ((a.M + 63) / 64) * a.ntn;
  if (tiles <= 0 || tiles > 0x7fffffff) return -2;
  const size_t lds = 2 * TILE_BYTES;
  const dim3 grid((unsigned)tiles, (unsigned)(a.ksplit > 1 ? a.ksplit : 1));
  if (at) hipLaunchKernelGGL((sgemm_kernel<true, false>), grid, dim3(256), lds, st, a);
  else if (bt) hipLaunchKernelGGL((sgemm_kernel<false, true>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((sgemm_kernel<false, false>), grid, dim3(256), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_small_gemm launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// 1: the shape is taken.  (4 <= K, N % 4 == 0 for the (K x N) form, dword-aligned operands with
// row offsets below 2^29 floats.)
extern "C" int s2c_small_gemm_supported(long long M, int N, int K, long long lda, long long ldb,
                                        int b_transposed) {
  if (M <= 0 || N <= 0 || K < 4 || K > 65536 || lda < K) return 0;
  if (b_transposed ? ldb < K : (ldb < N || N % 4 != 0 || ldb % 4 != 0)) return 0;
  if (M * lda >= (1ll << 29) || (long long)(b_transposed ? N : K) * ldb >= (1ll << 29)) return 0;
  return 1;
}

// Y (M x N, row stride ldy) = A (M x K, row stride lda) B (+ bias): b_transposed = 0: B is (K x N)
// row-major (row stride ldb) -- dX = dY W with W as stored; 1: B is W (N x K) row-major, Y = A W^T.
extern "C" int s2c_small_gemm(long long M, int N, int K, const float *A, long long lda,
                              const float *B, long long ldb, int b_transposed, const float *bias,
                              float *Y, long long ldy, void *stream) {
  if (!A || !B || !Y || !s2c_small_gemm_supported(M, N, K, lda, ldb, b_transposed)) return -2;
  if (((size_t)A & 3) || ((size_t)B & (b_transposed ? 3 : 15)) || M * ldy >= (1ll << 29)) return -2;
  SgArgs a{};
  a.M = M; a.N = N; a.K = K; a.A = A; a.B = B; a.Y = Y; a.bias = bias;
  a.arow = SgMap{0, 0, (int)lda};
  a.brow = SgMap{0, 0, (int)ldb};
  a.crow = SgMap{0, 0, (int)ldy};
  a.ntn = (N + 63) / 64;
  a.ksplit = 1;
  return launch_sgemm(a, 0, b_transposed, (hipStream_t)stream);
}

// The general form (csrc/s2c_sgemm.hip): operand rows through index maps, three layouts, optional
// split of K over workgroups with float atomics into a ZEROED Y (a long reduction behind few tiles).
//   form 0: Y = A B,    A (M x K) rows by arow(m), k contiguous;  B (K x N) rows by brow(k), N % 4 == 0
//   form 1: Y = A W^T,  A as above;                                W (N x K) rows by brow(n)
//   form 2: Y = A^T B,  A (K x M) rows by arow(k), M % 4 == 0;     B (K x N) rows by brow(k), N % 4 == 0
// Y rows by crow(m), N contiguous columns.
extern "C" int s2c_small_gemm_ex(const s2c_sgemm_args *g, void *stream) {
  if (!g || !g->A || !g->B || !g->Y || g->M <= 0 || g->N <= 0 || g->K < 4 || g->form < 0 || g->form > 2)
    return -2;
  const int at = g->form == 2, bt = g->form == 1;
  if ((!bt && (g->N % 4)) || (at && (g->M % 4))) return -2;
  if (((size_t)g->A & (at ? 15 : 3)) || ((size_t)g->B & (bt ? 3 : 15))) return -2;
  SgArgs a{};
  a.M = g->M; a.N = g->N; a.K = g->K; a.A = g->A; a.B = g->B; a.Y = g->Y; a.bias = g->bias;
  a.arow = SgMap{g->arow.div, g->arow.hi, g->arow.lo};
  a.brow = SgMap{g->brow.div, g->brow.hi, g->brow.lo};
  a.crow = SgMap{g->crow.div, g->crow.hi, g->crow.lo};
  // 32-bit byte offsets: every reachable element below 2^29
  const unsigned long long lim = 1ull << 29;
  if (map_reach(a.arow, at ? g->K : g->M) + (at ? g->M : g->K) >= lim ||
      map_reach(a.brow, bt ? g->N : g->K) + (bt ? g->K : g->N) >= lim ||
      map_reach(a.crow, g->M) + g->N >= lim)
    return -2;
  if (at && (a.arow.lo % 4 || a.arow.hi % 4)) return -2;       // 16-byte pieces along M
  if (!bt && (a.brow.lo % 4 || a.brow.hi % 4)) return -2;
  a.ntn = (g->N + 63) / 64;
  a.ksplit = g->ksplit > 1 ? g->ksplit : 1;
  const int nstage = (g->K + KC - 1) / KC;
  if (a.ksplit > nstage) a.ksplit = nstage;
  return launch_sgemm(a, at, bt, (hipStream_t)stream);
}
