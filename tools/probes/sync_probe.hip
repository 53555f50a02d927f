// sync_probe.hip -- what does a grid-wide exchange cost INSIDE a kernel on gfx950?
// (tools/probe_sync.py; DESIGN 4.4: the decoder's 300 dependent launches vs a persistent kernel)
//   mode 0: one counter, atomicAdd arrive + spin on an agent-scope load
//   mode 1: flag array: workgroup w stores the epoch to its own 64-byte line, 256 lanes poll G flags
//   mode 2: tagged data ("LL"): every value travels as {bits, epoch} in one 8-byte store; the
//           consumers poll the DATA -- no separate flag, one memory-side round trip per phase
//   group > 1 (modes 1, 2): only workgroups with the same (blockIdx.x % group) exchange
//           (group = 8: XCD-local under the round-robin dispatch)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define AG __HIP_MEMORY_SCOPE_AGENT

extern "C" __global__ __launch_bounds__(256) void sync_probe_kernel(
    int mode, int iters, int V, int group, unsigned *cnt, unsigned *flags,
    unsigned long long *data0, unsigned long long *data1, float *out) {
  const int G = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
  __shared__ float s_red[4];
  float carry = 1.0f;
  for (int it = 0; it < iters; ++it) {
    const unsigned epoch = it + 1;
    if (mode == 0) {
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, AG);
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, AG) < (unsigned)G * epoch) {}
      }
      __syncthreads();
    } else if (mode == 1) {
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + w * 16, epoch, __ATOMIC_RELEASE, AG);
      const int ng = G / group;
      for (int j = tid; j < ng; j += 256) {
        const int src = j * group + (w % group);
        while (__hip_atomic_load(flags + src * 16, __ATOMIC_ACQUIRE, AG) < epoch) {}
      }
      __syncthreads();
    } else if (mode == 3 || mode == 4) {
      // as mode 2, but every lane keeps ALL its polled 16-byte loads (two tagged values each) in
      // flight and re-polls only while some tag is stale: the decoder's 8 x 512 input vector
      const int ng = G / group, me = w / group, set = w % group;
      unsigned long long *buf = ((it & 1) ? data1 : data0) + (size_t)set * V;
      const int per = V / ng;
      if (tid < per) {
        const float val = carry * 0.5f + (float)(me * per + tid) * 1e-6f;
        const unsigned long long pk =
            ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(val);
        if (mode == 4)   // system-scope store (sc0 sc1) instead of agent scope (sc1)
          __hip_atomic_store(buf + me * per + tid, pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else
          __hip_atomic_store(buf + me * per + tid, pk, __ATOMIC_RELAXED, AG);
      }
      constexpr int U = 8;     // 8 x 16 bytes x 256 lanes = 4096 tagged values
      uint4 v[U];
      const uint4 *b4 = reinterpret_cast<const uint4 *>(buf);
      bool stale;
      do {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = tid + u * 256;
          if (2 * j < V)
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[u]) : "v"(b4 + j) : "memory");
          else v[u] = make_uint4(0u, epoch, 0u, epoch);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stale = false;
#pragma unroll
        for (int u = 0; u < U; ++u) stale |= (v[u].y != epoch) | (v[u].w != epoch);
      } while (stale);
      float acc = 0.0f;
#pragma unroll
      for (int u = 0; u < U; ++u) acc += __uint_as_float(v[u].x) + __uint_as_float(v[u].z);
      for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if ((tid & 63) == 0) s_red[tid >> 6] = acc;
      __syncthreads();
      carry = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)V;
      __syncthreads();
    } else {
      // members of this workgroup's exchange set: ng workgroups; the set's vector has V values,
      // member m produces V / ng of them
      const int ng = G / group, me = w / group, set = w % group;
      unsigned long long *buf = ((it & 1) ? data1 : data0) + (size_t)set * V;
      const int per = V / ng;
      if (tid < per) {
        const float val = carry * 0.5f + (float)(me * per + tid) * 1e-6f;
        const unsigned long long pk =
            ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(val);
        __hip_atomic_store(buf + me * per + tid, pk, __ATOMIC_RELAXED, AG);
      }
      float acc = 0.0f;
      for (int j = tid; j < V; j += 256) {
        unsigned long long pk;
        do {
          pk = __hip_atomic_load(buf + j, __ATOMIC_RELAXED, AG);
        } while ((unsigned)(pk >> 32) != epoch);
        acc += __uint_as_float((unsigned)pk);
      }
      for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if ((tid & 63) == 0) s_red[tid >> 6] = acc;
      __syncthreads();
      carry = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)V;
      __syncthreads();
    }
  }
  if (tid == 0) out[w] = carry;
}

extern "C" float sync_probe(int mode, int G, int iters, int V, int group, void *scratch) {
  unsigned *cnt = (unsigned *)scratch;
  unsigned *flags = cnt + 64;
  unsigned long long *d0 = (unsigned long long *)(flags + 16 * 1024);
  unsigned long long *d1 = d0 + 8 * 8192;
  float *out = (float *)(d1 + 8 * 8192);
  (void)hipMemsetAsync(scratch, 0, 4u << 20, 0);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipEventRecord(a, 0);
  hipLaunchKernelGGL(sync_probe_kernel, dim3(G), dim3(256), 0, 0, mode, iters, V, group, cnt,
                     flags, d0, d1, out);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return ms * 1000.0f / iters;
}
