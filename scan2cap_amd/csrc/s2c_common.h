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

// number of set bits of `mask` strictly below the calling lane
__device__ __forceinline__ int mask_rank_below(u64 mask) {
  return (int)__builtin_amdgcn_mbcnt_hi(
      (u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

}  // namespace s2c
