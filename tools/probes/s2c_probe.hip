// s2c_probe.hip -- measurement-only kernels (tools/bench_stream.py): what a streaming pass
// over (M x 64) fp32 rows can reach on this chip with (0) plain dwordx4 loads/stores and
// (1) a wave-private ring of LDS-DMA loads (global_load_lds_dwordx4, no VGPR staging).
// Not on the product path.
#include "../../scan2cap_amd/csrc/s2c_common.h"

namespace {

__global__ __launch_bounds__(256) void probe_copy_plain(long long n4, const float4 *__restrict__ a,
                                                        float4 *__restrict__ y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4;
       i += (long long)gridDim.x * 256) {
    float4 v = a[i];
    v.x += 1.f;
    y[i] = v;
  }
}

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// One wave = one pipeline: tiles of 32 rows x 64 floats (8 KB = 8 DMA instructions), ring of
// SLOTS tiles per wave, DEPTH tiles requested ahead.
template <int SLOTS, int DEPTH>
__global__ __launch_bounds__(512, 2) void probe_copy_dma(long long M, const float *__restrict__ A,
                                                         float *__restrict__ Y, int misalign) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long tiles = M / 32;
  const long long wid = (long long)blockIdx.x * 8 + wave, nw = (long long)gridDim.x * 8;
  unsigned char *ring = smem + (size_t)wave * SLOTS * 8192;
  const unsigned ring_lds = (unsigned)(size_t)ring;   // LDS byte address (low 32 bits of the flat address)
  auto issue = [&](long long t, int slot) {
    const float *src = A + t * 32 * 64 + misalign;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      glds16(src + i * 256 + lane * 4, ring_lds + slot * 8192 + i * 1024);
  };
  long long t = wid;
  int slot = 0;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (t + d * nw < tiles) issue(t + d * nw, d % SLOTS);
  for (; t < tiles; t += nw) {
    const long long tn = t + DEPTH * nw;
    if (tn < tiles) {
      issue(tn, (slot + DEPTH) % SLOTS);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * DEPTH) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const float4 *s = reinterpret_cast<const float4 *>(ring + slot * 8192);
    float4 *dst = reinterpret_cast<float4 *>(Y + t * 32 * 64);
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = s[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i].x += 1.f; dst[i * 64 + lane] = v[i]; }
    slot = (slot + 1) % SLOTS;
  }
}

}  // namespace

extern "C" int s2c_probe_copy(int mode, long long M, const float *A, float *Y, int blocks,
                              int misalign, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    hipLaunchKernelGGL(probe_copy_plain, dim3(blocks), dim3(256), 0, st, M * 16,
                       reinterpret_cast<const float4 *>(A), reinterpret_cast<float4 *>(Y));
  } else if (mode == 1) {
    const int lds = 8 * 2 * 8192;
    hipFuncSetAttribute((const void *)probe_copy_dma<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((probe_copy_dma<2, 1>), dim3(blocks), dim3(512), lds, st, M, A, Y, misalign);
  } else if (mode == 2) {
    const int lds = 8 * 2 * 8192;
    hipFuncSetAttribute((const void *)probe_copy_dma<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((probe_copy_dma<2, 1>), dim3(blocks), dim3(512), lds, st, M, A, Y, misalign);
  }
  return (int)hipGetLastError();
}


// A kernel that just HOLDS compute units: `blocks` workgroups of `threads` threads, each requesting
// `lds_bytes` of LDS (one per CU above 80 KB), spin for `cycles` shader-clock cycles.  For the test
// that provokes the persistent decoder's give-up path (tests/test_fused_gpu.py): with part of the
// chip held, some of its workgroups cannot become resident and the resident ones poll in vain.
namespace {
__global__ void probe_hog_kernel(long long cycles, int *sink) {
  extern __shared__ int hog_lds[];
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  while ((long long)__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(64);
  if (cycles < 0) sink[0] = hog_lds[threadIdx.x];       // never: keeps the LDS request alive
}
}  // namespace

extern "C" int s2c_probe_hog(int blocks, int threads, int lds_bytes, long long cycles, void *stream) {
  if (blocks <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024) return -1;
  if (hipFuncSetAttribute((const void *)probe_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                          lds_bytes) != hipSuccess)
    return -3;
  hipLaunchKernelGGL(probe_hog_kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream,
                     cycles, (int *)nullptr);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
