// s2c_bnbwd_fused.hip -- the backward of a 64 -> 64 BatchNorm(+ReLU) layer BETWEEN two others in ONE
// pass over its three row tensors (round 5; SA1's second layer: M = B * npoint * nsample = 1M rows).
// (reference: autograd of lib/pointnet2/pytorch_utils.py:67-120 inside pointnet2_modules.py:251-257.)
//
//     dY  = BatchNorm(+ReLU)-backward(dA, Y)            never written
//     dX  = dY W                      (M x 64)          the layer's input gradient
//     dW  = dY^T relu?(nY nscale + nshift)  (64 x 64)   its weight gradient: the layer's input is the
//                                                       previous layer's activation, recomputed from nY
//     [s1 | s2] = the previous layer's BatchNorm-backward column sums over (dX, nY)
//
// Until now: s2c_bn_bwd_gemm_next_stats wrote dY (268 MB at SA1) for s2c_weight_grad_stream to read back
// together with the activation (another 268 MB, written by the forward as a side output): 1340 + 536 MB
// in 324 + 91 us.  Here every row is read once (dA, Y, nY: 768 B) and written once (dX: 256 B):
//
//   * persistent workgroups of 8 waves, every WAVE its own pipeline over 16-row chunks (chunk c to
//     wave c % (8 grid)): the three row blocks of a chunk land in the wave's private LDS slot by
//     LDS-DMA (12 x `global_load_lds_dwordx4`, SGPR base + a per-lane offset that never changes, the
//     16-byte pieces XOR-swizzled by row on the SOURCE address), are read into registers in the
//     layouts the products need, and the slot takes the wave's next chunk while this one is
//     multiplied -- 8 x 12 KB in flight per CU, no workgroup barrier in the loop;
//   * dX on v_mfma_f32_16x16x32_bf16 (16 rows x 64 columns, K = the 64 channels; W's bf16x3 planes
//     resident in LDS in operand order, W read as stored: no transposed copy), dW on
//     v_mfma_f32_32x32x16_bf16 (K = the chunk's 16 rows: a lane's 8 consecutive-k values are 8 ROWS
//     of one column, read down the row-major chunk); fp32-accurate bf16x3 products (6 plane products,
//     truncation split as in s2c_dwstream.hip); dY is formed ONCE, in the column layout (a lane = one
//     channel x 8 rows: its per-channel constants sit in registers), and crosses to dX's row layout
//     through a wave-private 4 KB LDS transposer (forming it per layout re-read 28 x 16 B of constants
//     per lane and chunk from LDS: 308 -> 256 us);
//   * the dX tile leaves through the same transposer as 256-byte rows (dwordx4 stores);
//   * one (64 x 64) dW partial and one [s1 | s2] row per WORKGROUP (the waves meet in LDS once, at
//     the end); the caller's s2c_multi_colsum / s2c_bn_bwd_finalize_partials add them up
//     (kernel-boundary reductions: deterministic).
// The arithmetic of dY and of the column sums is that of bn_bwd_apply_kernel / bn_bwd_stats_kernel
// (s2c_sa.hip) term by term.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>
#include <stdlib.h>

extern "C" int s2c_weight_grad_stream_set_grid(int workgroups);

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NWAVE = 8;
constexpr int SLOT = 3 * 4096;                     // dA | Y | nY, 16 rows x 64 floats each
constexpr int OFF_W = NWAVE * SLOT;                // 98304: W planes, operand order [j][s][plane][lane] x 16 B
constexpr int OFF_OUT = OFF_W + 4 * 2 * 3 * 1024;  // 122880: dX transposer, 4 KB per wave
constexpr int OFF_PAR = OFF_OUT + NWAVE * 4096;    // 155648: 11 x 64 per-channel values
constexpr int LDS_BYTES = OFF_PAR + 11 * 64 * 4;   // 158464 (161792 is what a workgroup can have)
constexpr int OFF_STAT = NWAVE * 16384;            // after the loop: [wave][2][64] column sums behind the dW tiles

struct FbArgs {
  long long M, nchunks;
  const float *dA, *Y, *nY;
  const float *scale, *shift, *mean, *invstd, *coef;
  const float *nscale, *nshift, *nmean, *ninvstd;
  const float *W;
  float *dX, *dWpart, *npartial;
  int ldw, relu, nrelu;
};

__device__ __forceinline__ void glds16s(const void *sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// one pair of consecutive-k values -> dword of the hi / mid / lo planes (truncation split: the three
// planes hold the 24 leading bits of x, both residuals are exact; s2c_dwstream.hip)
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

struct Planes { bf16x8 p[3]; };

__device__ __forceinline__ Planes split8(const float (&v)[8]) {
  u32x4 h, m, l;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    unsigned a, b, c;
    split_pair(v[2 * d], v[2 * d + 1], a, b, c);
    h[d] = a; m[d] = b; l[d] = c;
  }
  Planes o;
  o.p[0] = __builtin_bit_cast(bf16x8, h);
  o.p[1] = __builtin_bit_cast(bf16x8, m);
  o.p[2] = __builtin_bit_cast(bf16x8, l);
  return o;
}

// bn_bwd_apply_kernel (s2c_sa.hip): dz = dA [Y scale + shift > 0], dY = k0 (dz - k1 - ((Y - mean) invstd) k2)
__device__ __forceinline__ float bn_bwd(float g, float y, float sc, float sh, float mu, float is,
                                        float k0, float k1, float k2, int relu) {
  if (relu && !(y * sc + sh > 0.f)) g = 0.f;
  return k0 * (g - k1 - ((y - mu) * is) * k2);
}

// float offset of element (row, col) of a swizzled 16 x 64 block
__device__ __forceinline__ int sw(int row, int col) {
  return row * 64 + ((((col >> 2) ^ row) & 15) << 2) + (col & 3);
}

__global__ __launch_bounds__(512, 1) void bn_bwd_dx_dw64_kernel(FbArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;       // 32x32x16 operand coordinates
  const int r16 = lane & 15, g16 = lane >> 4;     // 16x16x32 operand coordinates
  float *par = reinterpret_cast<float *>(smem + OFF_PAR);

  // ---- once per workgroup: per-channel values and W's planes in operand order ------------------
  if (tid < 64) {
    par[0 * 64 + tid] = p.scale[tid];  par[1 * 64 + tid] = p.shift[tid];
    par[2 * 64 + tid] = p.mean[tid];   par[3 * 64 + tid] = p.invstd[tid];
    par[4 * 64 + tid] = p.coef[tid];   par[5 * 64 + tid] = p.coef[64 + tid];
    par[6 * 64 + tid] = p.coef[128 + tid];
    par[7 * 64 + tid] = p.nscale[tid]; par[8 * 64 + tid] = p.nshift[tid];
    par[9 * 64 + tid] = p.nmean[tid];  par[10 * 64 + tid] = p.ninvstd[tid];
  }
  {
    // dX[r, n] = sum_c dY[r, c] W[c, n]: B operand of block j (columns 16 j ..), k-step s (channels
    // 32 s ..): lane (n = lane & 15, kb = lane >> 4) holds W[32 s + 8 kb + e][16 j + n], e = 0 .. 7
    const int js = tid >> 6, j = js >> 1, s = js & 1;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = p.W[(size_t)(32 * s + 8 * g16 + e) * p.ldw + 16 * j + r16];
    const Planes pl = split8(v);
    bf16x8 *dst = reinterpret_cast<bf16x8 *>(smem + OFF_W) + (size_t)js * 3 * 64 + lane;
    dst[0] = pl.p[0]; dst[64] = pl.p[1]; dst[128] = pl.p[2];
  }
  __syncthreads();

  // per-lane constants of the column layout: channel / column li + 32 i
  float c_sc[2], c_sh[2], c_mu[2], c_is[2], c_k0[2], c_k1[2], c_k2[2], b_sc[2], b_sh[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = li + 32 * i;
    c_sc[i] = par[c]; c_sh[i] = par[64 + c]; c_mu[i] = par[128 + c]; c_is[i] = par[192 + c];
    c_k0[i] = par[256 + c]; c_k1[i] = par[320 + c]; c_k2[i] = par[384 + c];
    b_sc[i] = par[448 + c]; b_sh[i] = par[512 + c];
  }
  const int relu = p.relu, nrelu = p.nrelu;

  // DMA: instruction q of a block, lane l -> 16-byte piece P = 64 q + l of the LDS block = (row P >> 4,
  // position P & 15), which holds the row's piece (P & 15) ^ row
  unsigned voff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int P = 64 * q + lane, row = P >> 4, pos = P & 15;
    voff[q] = (unsigned)(row * 64 + ((pos ^ row) & 15) * 4) * 4u;
  }
  unsigned char *slot = smem + (size_t)wave * SLOT;
  const unsigned slot_lds = (unsigned)(size_t)slot;
  auto issue = [&](long long chunk) {
    const float *b0 = p.dA + chunk * 1024, *b1 = p.Y + chunk * 1024, *b2 = p.nY + chunk * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16s(b0, voff[q], slot_lds + (unsigned)q * 1024u);
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16s(b1, voff[q], slot_lds + 4096u + (unsigned)q * 1024u);
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16s(b2, voff[q], slot_lds + 8192u + (unsigned)q * 1024u);
  };
  const float *sA = reinterpret_cast<const float *>(slot);
  const float *sY = sA + 1024, *sN = sA + 2048;
  float *sOut = reinterpret_cast<float *>(smem + OFF_OUT + (size_t)wave * 4096);
  const bf16x8 *Wl = reinterpret_cast<const bf16x8 *>(smem + OFF_W);

  f32x16 adw[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) adw[i][j][e] = 0.f;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};

  const long long GW = (long long)gridDim.x * NWAVE, gw = (long long)blockIdx.x * NWAVE + wave;
  if (gw < p.nchunks) issue(gw);
#pragma unroll 1
  for (long long chunk = gw; chunk < p.nchunks; chunk += GW) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- the chunk into registers: column layout (both MFMA operands of dW; dY is formed there and
    // handed to the dX product through the wave's transposer) and dX's accumulator layout of nY --------
    float ga[2][8], ya[2][8], nb[2][8];             // columns li + 32 i, rows 8 lk + e
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int o = sw(8 * lk + e, li + 32 * i);
        ga[i][e] = sA[o]; ya[i][e] = sY[o]; nb[i][e] = sN[o];
      }
    float nacc[4][4];                               // column 16 j + (lane & 15), row 4 g16 + t
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) nacc[j][t] = sN[sw(4 * g16 + t, 16 * j + r16)];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (chunk + GW < p.nchunks) issue(chunk + GW);  // the slot is in registers: its next chunk

    // ---- dY (once), dW += dY^T A_in --------------------------------------------------------------
    {
      Planes fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v[8], w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = bn_bwd(ga[i][e], ya[i][e], c_sc[i], c_sh[i], c_mu[i], c_is[i], c_k0[i], c_k1[i],
                        c_k2[i], relu);
          sOut[sw(8 * lk + e, li + 32 * i)] = v[e];
          const float a = nb[i][e] * b_sc[i] + b_sh[i];
          w[e] = nrelu ? fmaxf(a, 0.f) : a;
        }
        fa[i] = split8(v);
        fb[i] = split8(w);
      }
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            adw[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].p[TA[q]], fb[j].p[TB[q]],
                                                                adw[i][j], 0, 0, 0);
    }
    // ---- dX = dY W: row lane & 15, channels 32 s + 8 g16 + (0 .. 7) back out of the transposer -----
    f32x4 adx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) adx[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Planes fx[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float v[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 d = *reinterpret_cast<const float4 *>(
            sOut + r16 * 64 + (((8 * s + 2 * g16 + h) ^ r16) & 15) * 4);
        v[4 * h + 0] = d.x; v[4 * h + 1] = d.y; v[4 * h + 2] = d.z; v[4 * h + 3] = d.w;
      }
      fx[s] = split8(v);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 *wp = Wl + (size_t)(j * 2 + s) * 3 * 64 + lane;
        const bf16x8 w0 = wp[0], w1 = wp[64], w2 = wp[128];
        const bf16x8 wpl[3] = {w0, w1, w2};
#pragma unroll
        for (int q = 0; q < 6; ++q)
          adx[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[s].p[TA[q]], wpl[TB[q]], adx[j], 0, 0, 0);
      }
    // ---- the previous layer's column sums, dX out through the transposer -------------------------
    // C/D layout of 16x16: column = lane & 15, row = 4 (lane >> 4) + t
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 16 * j + r16;
      const float n_sc = par[448 + c], n_sh = par[512 + c], n_mu = par[576 + c], n_is = par[640 + c];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float v = adx[j][t], y = nacc[j][t];
        float dz = v;
        if (nrelu && !(y * n_sc + n_sh > 0.f)) dz = 0.f;
        s1[j] += dz;
        s2[j] += dz * ((y - n_mu) * n_is);
        sOut[sw(4 * g16 + t, 16 * j + r16)] = v;
      }
    }
    {
      float *dst = p.dX + chunk * 1024;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 4 * q + g16;
        const float4 v = *reinterpret_cast<const float4 *>(sOut + row * 64 + ((r16 ^ row) & 15) * 4);
        *reinterpret_cast<float4 *>(dst + row * 64 + r16 * 4) = v;
      }
    }
  }

  // ---- the waves meet: one dW partial and one [s1 | s2] row per workgroup -----------------------
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s1[j] += __shfl_xor(s1[j], 16, 64); s1[j] += __shfl_xor(s1[j], 32, 64);
    s2[j] += __shfl_xor(s2[j], 16, 64); s2[j] += __shfl_xor(s2[j], 32, 64);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                 // every wave is out of its loop: the LDS is free
  float *stat = reinterpret_cast<float *>(smem + OFF_STAT) + wave * 128;
  if (g16 == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { stat[16 * j + r16] = s1[j]; stat[64 + 16 * j + r16] = s2[j]; }
  }
  float *red = reinterpret_cast<float *>(smem);
  {
    float *dst = red + (size_t)wave * 4096;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[((i * 2 + j) * 16 + e) * 64 + lane] = adw[i][j][e];
  }
  __syncthreads();
  {
    float *out = p.dWpart + (size_t)blockIdx.x * 4096;
#pragma unroll 1
    for (int x = tid; x < 4096; x += 512) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < NWAVE; ++k) a += red[(size_t)k * 4096 + x];
      const int ln = x & 63, e = (x >> 6) & 15, ij = x >> 10;
      // C/D layout of 32x32: column (B operand) = ln & 31, row (A operand) = (e & 3) + 8 (e >> 2) + 4 (ln >> 5)
      const int c = 32 * (ij >> 1) + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5);
      const int n = 32 * (ij & 1) + (ln & 31);
      out[c * 64 + n] = a;
    }
    if (tid < 128) {
      const float *st = reinterpret_cast<const float *>(smem + OFF_STAT);
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < NWAVE; ++k) a += st[k * 128 + tid];
      p.npartial[(size_t)blockIdx.x * 128 + tid] = a;
    }
  }
}

int fb_grid(long long M) {
  const int g = s2c_weight_grad_stream_set_grid(0);     // the persistent grid beside the geometry stage
  const long long units = (M / 16 + NWAVE - 1) / NWAVE;
  return (int)(g < units ? g : units);
}

bool fb_takes(long long M) { return M >= 4096 && M % 16 == 0 && M * 64 * 4 < (1ll << 40); }

}  // namespace

// Rows of the two partial tables (= the grid); 0: shape not taken (M % 16, fewer than 4096 rows).
extern "C" int s2c_bn_bwd_dx_dw64_parts(long long M) { return fb_takes(M) ? fb_grid(M) : 0; }

extern "C" int s2c_bn_bwd_dx_dw64(long long M, const float *dA, const float *Y, const float *scale,
                                  const float *shift, const float *mean, const float *invstd,
                                  const float *coef, int relu, const float *W, int ldw, float *dX,
                                  const float *nY, const float *nscale, const float *nshift,
                                  const float *nmean, const float *ninvstd, int nrelu,
                                  float *dWpart, float *npartial, void *stream) {
  if (!fb_takes(M) || !dA || !Y || !nY || !W || !dX || !dWpart || !npartial || ldw < 64) return -2;
  if ((((size_t)dA | (size_t)Y | (size_t)nY | (size_t)dX) & 15) != 0) return -2;
  // the dynamic-LDS cap is a per-DEVICE function attribute: one flag per device of the process
  static bool attr_dev[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -2;
  if (!attr_dev[dev]) {
    if (hipFuncSetAttribute((const void *)bn_bwd_dx_dw64_kernel,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      (void)hipGetLastError();
      return -2;
    }
    attr_dev[dev] = true;
  }
  FbArgs a{};
  a.M = M; a.nchunks = M / 16;
  a.dA = dA; a.Y = Y; a.nY = nY;
  a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.coef = coef;
  a.nscale = nscale; a.nshift = nshift; a.nmean = nmean; a.ninvstd = ninvstd;
  a.W = W; a.ldw = ldw; a.dX = dX; a.dWpart = dWpart; a.npartial = npartial;
  a.relu = relu; a.nrelu = nrelu;
  hipLaunchKernelGGL(bn_bwd_dx_dw64_kernel, dim3(fb_grid(M)), dim3(512), LDS_BYTES,
                     (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_bn_bwd_dx_dw64 launch failed: %s\n", hipGetErrorString(e));
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void *)bn_bwd_dx_dw64_kernel) == hipSuccess)
      fprintf(stderr, "  grid %d lds %d | static %zu local %zu maxthr %d regs %d maxdyn %d\n", fb_grid(M),
              LDS_BYTES, fa.sharedSizeBytes, fa.localSizeBytes, fa.maxThreadsPerBlock, fa.numRegs,
              fa.maxDynamicSharedSizeBytes);
    return (int)e;
  }
  return 0;
}
