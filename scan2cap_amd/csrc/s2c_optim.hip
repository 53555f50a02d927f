// s2c_optim.hip -- the Adam update of EVERY parameter tensor of the model in one launch.
//
// Reference: scripts/train.py:138 `optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.wd)`
// stepped once per batch (lib/solver.py:293-302).  torch's fused multi-tensor Adam needs 3 launches of
// ~40 us + a `_foreach_add_` for the step counters for the model's ~125 tensors (6.2 M parameters:
// 0.12 ms of a 7.5 ms step, at 0.8 TB/s of the 174 MB it moves -- a chunk of 64K elements per block
// table entry, tensors of 64 .. 512 elements each a block of their own inside it).
//
// One launch here: the kernel argument holds, per tensor, {parameter, gradient, element count, offset
// of its moments}; the first and second moments of ALL tensors live in two flat buffers the optimizer
// owns (scan2cap_amd/optim.py), so an entry is 24 bytes and 128 tensors fit the 4 KB argument block.
// A workgroup takes one 4096-element chunk of one tensor (block -> tensor by a prefix table in the
// argument), 16-byte accesses where the three pointers allow it.  The update is torch.optim.Adam's
// (torch/optim/adam.py `_single_tensor_adam`, amsgrad / maximize off, L2 weight decay added to the
// gradient), evaluated in fp32 like its fused kernel:
//     g' = g + wd p;  m = m + (1 - b1) (g' - m);  v = b2 v + (1 - b2) g' g'
//     p = p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The step counts t (one per tensor, as torch keeps them: a parameter without a gradient is skipped
// and its count stays) live in device memory (a replayed hipGraph cannot change a host scalar): every
// workgroup reads its tensor's count when it starts, and the LAST workgroup to finish (an agent-scope
// counter) stores t + 1 for every updated tensor and clears the counter -- every other workgroup has
// read by then, and the next launch is ordered behind this one by the stream.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"
#include "../../include/s2c_ops.h"

#include <stdio.h>

namespace {

constexpr int CHUNK = 4096;       // elements per workgroup
constexpr int THREADS = 256;

__global__ __launch_bounds__(THREADS) void adam_multi_kernel(s2c_adam_args a) {
  __shared__ float s_step_size, s_bc2_sqrt;
  __shared__ int s_t;
  const int tid = threadIdx.x;
  if (tid == 0) {
    // block -> tensor: the last tensor whose first block is <= blockIdx.x
    int lo = 0, hi = a.n_tensors - 1;
    const int blk = (int)blockIdx.x;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (a.first_block[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    s_t = lo;
    // the bias corrections in float64, as torch forms them (beta^t in fp32 loses 1 - beta2^t to
    // cancellation: 1.3e-5 relative at t = 1 from rounding 0.999 alone)
    const double step = (double)a.step[lo] + 1.0;
    const double bc1 = 1.0 - pow(a.beta1, step);
    const double bc2 = 1.0 - pow(a.beta2, step);
    s_step_size = (float)(a.lr / bc1);
    s_bc2_sqrt = (float)sqrt(bc2);
  }
  __syncthreads();
  const int t = s_t;
  const float step_size = s_step_size, bc2_sqrt = s_bc2_sqrt;
  const float b1 = (float)a.beta1, b2 = (float)a.beta2;
  const float eps = (float)a.eps, wd = (float)a.weight_decay;
  const float omb1 = (float)(1.0 - a.beta1), omb2 = (float)(1.0 - a.beta2);

  float *__restrict__ p = a.t[t].param;
  const float *__restrict__ g = a.t[t].grad;
  float *__restrict__ m = a.exp_avg + a.t[t].offset;
  float *__restrict__ v = a.exp_avg_sq + a.t[t].offset;
  const int n = a.t[t].numel;
  const int e0 = ((int)blockIdx.x - a.first_block[t]) * CHUNK;
  const int e1 = min(n, e0 + CHUNK);

  auto upd = [&](float &pp, float gg, float &mm, float &vv) {
    const float gr = gg + wd * pp;
    mm = mm + omb1 * (gr - mm);
    vv = b2 * vv + omb2 * gr * gr;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp = pp - step_size * (mm / denom);
  };

  if (g != nullptr) {
    const bool vec = ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0);
    if (vec) {
      const int nv = (e1 - e0) >> 2;
      float4 *p4 = reinterpret_cast<float4 *>(p + e0);
      const float4 *g4 = reinterpret_cast<const float4 *>(g + e0);
      float4 *m4 = reinterpret_cast<float4 *>(m + e0);
      float4 *v4 = reinterpret_cast<float4 *>(v + e0);
      // CHUNK / 4 / THREADS = 4 float4 per thread: all loads in flight before the first use
      float4 pp[4], gg[4], mm[4], vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + u * THREADS;
        if (i < nv) { pp[u] = p4[i]; gg[u] = g4[i]; mm[u] = m4[i]; vv[u] = v4[i]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + u * THREADS;
        if (i < nv) {
          upd(pp[u].x, gg[u].x, mm[u].x, vv[u].x);
          upd(pp[u].y, gg[u].y, mm[u].y, vv[u].y);
          upd(pp[u].z, gg[u].z, mm[u].z, vv[u].z);
          upd(pp[u].w, gg[u].w, mm[u].w, vv[u].w);
          p4[i] = pp[u]; m4[i] = mm[u]; v4[i] = vv[u];
        }
      }
      for (int i = e0 + (nv << 2) + tid; i < e1; i += THREADS) {
        float pp1 = p[i], mm1 = m[i], vv1 = v[i];
        upd(pp1, g[i], mm1, vv1);
        p[i] = pp1; m[i] = mm1; v[i] = vv1;
      }
    } else {
      for (int i = e0 + tid; i < e1; i += THREADS) {
        float pp1 = p[i], mm1 = m[i], vv1 = v[i];
        upd(pp1, g[i], mm1, vv1);
        p[i] = pp1; m[i] = mm1; v[i] = vv1;
      }
    }
  }
  // the last workgroup to finish advances the step counts of the tensors that had a gradient (every
  // workgroup read its tensor's count when it started)
  __shared__ unsigned s_done;
  __syncthreads();
  if (tid == 0)
    s_done = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_done == gridDim.x - 1) {
    for (int i = tid; i < a.n_tensors; i += THREADS)
      if (a.t[i].grad != nullptr)
        __hip_atomic_store(a.step + i, a.step[i] + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static_assert(sizeof(s2c_adam_args) <= 4096, "the argument block of a launch");

}  // namespace

extern "C" int s2c_adam_chunk(void) { return CHUNK; }

// a->first_block[i] = number of workgroups of the tensors before i (the caller fills it with
// ceil(numel / s2c_adam_chunk()) prefix sums); a->first_block[n_tensors] = the grid.
extern "C" int s2c_adam_multi(const s2c_adam_args *a, void *stream) {
  if (!a || a->n_tensors < 0 || a->n_tensors > S2C_ADAM_MAX_TENSORS) return S2C_EINVAL;
  if (a->n_tensors == 0) return 0;
  if (!a->exp_avg || !a->exp_avg_sq || !a->step || !a->counter) return S2C_EINVAL;
  const int grid = a->first_block[a->n_tensors];
  if (grid <= 0) return S2C_EINVAL;
  for (int i = 0; i < a->n_tensors; ++i) {
    const int nb = a->first_block[i + 1] - a->first_block[i];
    if (!a->t[i].param || a->t[i].numel <= 0 || a->t[i].offset < 0 ||
        nb != (a->t[i].numel + CHUNK - 1) / CHUNK)
      return S2C_EINVAL;
  }
  hipLaunchKernelGGL(adam_multi_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, *a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_adam_multi launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
