// s2c_common.h -- shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Canonical arithmetic: IEEE binary32 in source order, no FMA
// contraction.  Enforced both here and with -ffp-contract=off on the command line.
#pragma clang fp contract(off)

namespace s2c {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x3 __attribute__((ext_vector_type(3)));

constexpr int kWave = 64;

// a*a + b*b + c*c of the index-producing ops (FPS distance and |p|^2 skip test, ball query,
// three_nn: sampling_gpu.cu:100-104, ball_query_gpu.cu:31-32, interpolate_gpu.cu:36-37).
// Default (0) = the canonical arithmetic of DESIGN.md section 2: source order, every product and
// sum rounded separately.  The reference's own build is nvcc, whose default --fmad=true
// contracts a product feeding an add into one fused multiply-add; which of the three products
// stays a plain multiply is the CUDA compiler's choice and cannot be observed here (no CUDA
// device, no nvcc), so both LLVM-style contractions are provided for a holder of real CUDA
// outputs to check the <= 1-ulp near-tie class against:
//   1:  fma(c, c, fma(a, a, b*b))     (the left product fused first, then the outer add)
//   2:  fma(c, c, fma(b, b, a*a))
// Selected at BUILD time (-DS2C_NVCC_CONTRACT=1|2 -> libs2c_hip_nvcc<k>.so,
// `S2C_NVCC_CONTRACT=k python -m scan2cap_amd.build`); the oracle has the same switch at run
// time (s2c_oracle_set_contract).  Features (floats) are unaffected: only xyz enters these.
#ifndef S2C_NVCC_CONTRACT
#define S2C_NVCC_CONTRACT 0
#endif
__device__ __forceinline__ float sq3(float a, float b, float c) {
#if S2C_NVCC_CONTRACT == 1
  return __builtin_fmaf(c, c, __builtin_fmaf(a, a, b * b));
#elif S2C_NVCC_CONTRACT == 2
  return __builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a));
#else
  return (a * a + b * b) + c * c;
#endif
}

// DPP controls (gfx9 encoding)
constexpr int DPP_QUAD_1032 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_2301 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // row_half_mirror
constexpr int DPP_ROW_MIRROR = 0x140;      // row_mirror

template <int CTRL>
__device__ __forceinline__ u64 dpp_mov_u64(u64 v) {
  const int lo = (int)(u32)v, hi = (int)(u32)(v >> 32);
  const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return ((u64)(u32)hi2 << 32) | (u64)(u32)lo2;
}

__device__ __forceinline__ u64 umax64(u64 a, u64 b) { return a > b ? a : b; }

// After this every lane of each 16-lane row holds that row's maximum.
__device__ __forceinline__ u64 row16_max_u64(u64 v) {
  v = umax64(v, dpp_mov_u64<DPP_QUAD_1032>(v));
  v = umax64(v, dpp_mov_u64<DPP_QUAD_2301>(v));
  v = umax64(v, dpp_mov_u64<DPP_ROW_HALF_MIRROR>(v));
  v = umax64(v, dpp_mov_u64<DPP_ROW_MIRROR>(v));
  return v;
}

__device__ __forceinline__ u64 readlane_u64(u64 v, int lane) {
  const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, lane);
  const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), lane);
  return ((u64)hi << 32) | lo;
}

// Wave-uniform maximum of a u64 across all 64 lanes.
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
  v = row16_max_u64(v);
  const u64 a = readlane_u64(v, 0), b = readlane_u64(v, 16);
  const u64 c = readlane_u64(v, 32), d = readlane_u64(v, 48);
  return umax64(umax64(a, b), umax64(c, d));
}

// ---- the same reductions as two 32-bit passes -------------------------------------------
// A 64-bit max step costs two DPP moves, a 64-bit compare and two selects; a 32-bit one is
// a single v_max_u32 with a DPP operand.  max over (hi, lo) pairs = max over hi, then max
// over lo among the lanes that hold that hi.
template <int CTRL>
__device__ __forceinline__ u32 dpp_umax32(u32 v) {
  const u32 o = (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
  return v > o ? v : o;
}
__device__ __forceinline__ u32 row16_umax32(u32 v) {
  v = dpp_umax32<DPP_QUAD_1032>(v);
  v = dpp_umax32<DPP_QUAD_2301>(v);
  v = dpp_umax32<DPP_ROW_HALF_MIRROR>(v);
  v = dpp_umax32<DPP_ROW_MIRROR>(v);
  return v;
}
__device__ __forceinline__ u32 wave_umax32(u32 v) {
  v = row16_umax32(v);
  const u32 a = (u32)__builtin_amdgcn_readlane((int)v, 0), b = (u32)__builtin_amdgcn_readlane((int)v, 16);
  const u32 c = (u32)__builtin_amdgcn_readlane((int)v, 32), d = (u32)__builtin_amdgcn_readlane((int)v, 48);
  const u32 ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
__device__ __forceinline__ u64 wave_max_u64_2x32(u64 v) {
  const u32 hi = (u32)(v >> 32);
  const u32 mhi = wave_umax32(hi);
  const u32 mlo = wave_umax32(hi == mhi ? (u32)v : 0u);
  return ((u64)mhi << 32) | mlo;
}
// every lane of each 16-lane row gets that row's maximum
__device__ __forceinline__ u64 row16_max_u64_2x32(u64 v) {
  const u32 hi = (u32)(v >> 32);
  const u32 mhi = row16_umax32(hi);
  const u32 mlo = row16_umax32(hi == mhi ? (u32)v : 0u);
  return ((u64)mhi << 32) | mlo;
}

// number of set bits of `mask` strictly below the calling lane
__device__ __forceinline__ int mask_rank_below(u64 mask) {
  return (int)__builtin_amdgcn_mbcnt_hi(
      (u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Zero fill as an ordinary KERNEL launch.  Not hipMemsetAsync: inside a captured hipGraph a
// memset node followed by the kernel that accumulates into the buffer (float atomics) was
// observed NOT to be ordered on ROCm 7.2 / gfx950 -- replays of the training step returned
// non-finite gradients in 88 % of the steps (tools/stress_nan.py), eager launches never;
// a kernel node behind a kernel node is ordered.
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
      0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// a[i] = element (row i, column = lane & 3 of the quad)  ->  a[c] = element (row = lane & 3,
// column c): a 4x4 transpose inside every quad of lanes (2 x 2 exchanges).
__device__ __forceinline__ void quad_transpose(float (&a)[4], int lane) {
  const bool o1 = lane & 1, o2 = lane & 2;
#pragma unroll
  for (int i = 0; i < 4; i += 2) {
    const float x = o1 ? a[i] : a[i + 1];
    const float y = dpp_quad<0xB1>(x);              // quad_perm [1,0,3,2]
    if (o1) a[i] = y; else a[i + 1] = y;
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float x = o2 ? a[k] : a[k + 2];
    const float y = dpp_quad<0x4E>(x);              // quad_perm [2,3,0,1]
    if (o2) a[k] = y; else a[k + 2] = y;
  }
}


static __global__ void __launch_bounds__(256) zero_fill_kernel(uint4 *__restrict__ p16, size_t n16,
                                                        u32 *__restrict__ tail, int ntail) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t k = i; k < n16; k += stride) p16[k] = make_uint4(0u, 0u, 0u, 0u);
  if (i < (size_t)ntail) tail[i] = 0u;
}

// `bytes` a multiple of 4, `p` 4-byte aligned (16-byte aligned bulk, dword head/tail).
inline hipError_t zero_async(void *p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  uintptr_t a = (uintptr_t)p;
  size_t head = (16 - (a & 15)) & 15;             // bytes up to 16-byte alignment
  if (head > bytes) head = bytes;
  if (head) {                                      // rare: unaligned start
    hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(256), 0, st, (uint4 *)nullptr, (size_t)0,
                       (u32 *)p, (int)(head / 4));
    a += head; bytes -= head;
  }
  const size_t n16 = bytes / 16;
  const int ntail = (int)((bytes - n16 * 16) / 4);
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint4 *)a, n16,
                     (u32 *)(a + n16 * 16), ntail);
  return hipGetLastError();
}

}  // namespace s2c
