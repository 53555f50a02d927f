// s2c_loss.hip -- the detection part of get_scene_cap_loss as two forward and one
// backward kernel.
//
// Reference: lib/loss_helper.py:24-69 (vote loss), :71-111 (objectness), :113-187
// (box + semantic class), :381-491 (weights), utils/nn_distance.py:13-59.  The
// reference (and the op-by-op restatement in scan2cap_amd/loss_helper.py) spends
// ~170 forward + ~250 autograd micro-kernels on it, each a few microseconds of
// launch latency for a few KB of data; the actual arithmetic is ~1 MFLOP.
//
// One block per scene.  All terms are sums over proposals / seeds / GT boxes divided
// by global counts, so the forward is (1) per-scene partial sums + per-element
// labels/arg-mins, (2) a tiny finalize kernel (fixed-order double sums over scenes:
// deterministic); the backward is analytic and needs only the saved arg-mins.
// Tie rules: first minimum / first maximum, like torch.min / torch.argmax.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

// row stride of the score arrays: dense (n) or the head-output row (a.ld_scores)
#define LD(n) ((size_t)(a.ld_scores > 0 ? a.ld_scores : (n)))

namespace {

constexpr int NPART = 14;
// One scene = NSEC workgroups (blockIdx.y): sections 0 .. NSEC - 3 share the seeds of the vote
// loss, NSEC - 2 takes the proposals, NSEC - 1 the ground-truth boxes -- the three loops are
// independent, and one workgroup per scene ran them one after the other (56 us of dependent
// global loads on 8 CUs for a few hundred KB of inputs).  partial = (B, NSEC, NPART).
constexpr int NSEC = 6;
constexpr int MAXG = 256, MAXK = 1024;

__device__ __forceinline__ float huber1(float e) {   // nn_distance.py:13-30, delta = 1
  const float a = fabsf(e);
  const float q = fminf(a, 1.0f);
  return 0.5f * q * q + (a - q);
}
__device__ __forceinline__ float huber1_grad(float e) { return fminf(fmaxf(e, -1.0f), 1.0f); }

// log-sum-exp of n logits (n small)
__device__ __forceinline__ float lse_of(const float *x, int n) {
  float m = x[0];
  for (int i = 1; i < n; ++i) m = fmaxf(m, x[i]);
  float s = 0.0f;
  for (int i = 0; i < n; ++i) s += expf(x[i] - m);
  return m + logf(s);
}

__device__ __forceinline__ void block_sums(float (&v)[NPART], float *s_red /*[4][NPART]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NPART; ++k) {
    float x = v[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    if (lane == 0) s_red[wave * NPART + k] = x;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void detloss_fwd_kernel(s2c_detloss_args a) {
  __shared__ float s_gt[MAXG * 3];
  __shared__ float s_c[MAXK * 3];
  __shared__ float s_red[4 * NPART];
  const int b = blockIdx.x, tid = threadIdx.x, sec = blockIdx.y;
  const int K = a.K, G = a.G;
  if (sec >= NSEC - 2) {
    for (int i = tid; i < G * 3; i += 256)
      s_gt[i] = a.center_label[((size_t)b * G + i / 3) * a.ld_center_label + i % 3];
    for (int i = tid; i < K * 3; i += 256) s_c[i] = a.center[(size_t)b * K * 3 + i];
  }
  __syncthreads();
  float acc[NPART];
#pragma unroll
  for (int i = 0; i < NPART; ++i) acc[i] = 0.0f;

  // ---- vote loss (loss_helper.py:24-69): seeds of this scene
  if (sec < NSEC - 2)
  for (int s = sec * 256 + tid; s < a.S; s += 256 * (NSEC - 2)) {
    const int idx = a.seed_inds[(size_t)b * a.S + s];
    const float m = (float)a.vote_label_mask[(size_t)b * a.N + idx];
    const float *sx = a.seed_xyz + ((size_t)b * a.S + s) * 3;
    const float *vl = a.vote_label + ((size_t)b * a.N + idx) * 9;
    float best = 0.0f;
    int barg = 0;
    for (int j = 0; j < 3; ++j) {
      const float gx = vl[3 * j] + sx[0], gy = vl[3 * j + 1] + sx[1], gz = vl[3 * j + 2] + sx[2];
      float dj = 0.0f;
      int ij = 0;
      for (int i = 0; i < a.VF; ++i) {
        const float *v = a.vote_xyz + (((size_t)b * a.S + s) * a.VF + i) * 3;
        const float d = (fabsf(v[0] - gx) + fabsf(v[1] - gy)) + fabsf(v[2] - gz);
        if (i == 0 || d < dj) { dj = d; ij = i; }
      }
      if (j == 0 || dj < best) { best = dj; barg = ij * 3 + j; }
    }
    a.vote_arg[(size_t)b * a.S + s] = barg;
    acc[0] += best * m;
    acc[1] += m;
  }

  // ---- per proposal: objectness (:71-111) and box / class terms (:113-187)
  if (sec == NSEC - 2)
  for (int k = tid; k < K; k += 256) {
    const size_t bk = (size_t)b * K + k;
    const float *ax = a.agg_xyz + bk * 3;
    float d1 = 0.0f;
    int g1 = 0;
    for (int g = 0; g < G; ++g) {
      const float dx = ax[0] - s_gt[3 * g], dy = ax[1] - s_gt[3 * g + 1], dz = ax[2] - s_gt[3 * g + 2];
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (g == 0 || d < d1) { d1 = d; g1 = g; }
    }
    const float eu = sqrtf(d1 + 1e-6f);
    const bool near = eu < a.near_threshold;
    const float mask = (near || eu > a.far_threshold) ? 1.0f : 0.0f;
    const float label = near ? 1.0f : 0.0f;
    a.objectness_label[bk] = near ? 1 : 0;
    a.objectness_mask[bk] = mask;
    a.object_assignment[bk] = g1;
    const float *os = a.objectness_scores + bk * LD(2);
    const float w = near ? a.obj_w1 : a.obj_w0;
    acc[2] += w * (lse_of(os, 2) - os[near ? 1 : 0]) * mask;
    acc[3] += mask;
    acc[4] += label;
    const int pred = os[1] > os[0] ? 1 : 0;           // argmax, first maximum on ties
    acc[13] += (pred == (near ? 1 : 0)) ? mask : 0.0f;

    // centre: nearest GT of the predicted centre
    const float cx = s_c[3 * k], cy = s_c[3 * k + 1], cz = s_c[3 * k + 2];
    float c1 = 0.0f;
    int cg = 0;
    for (int g = 0; g < G; ++g) {
      const float dx = cx - s_gt[3 * g], dy = cy - s_gt[3 * g + 1], dz = cz - s_gt[3 * g + 2];
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (g == 0 || d < c1) { c1 = d; cg = g; }
    }
    a.center_g1[bk] = cg;
    acc[5] += c1 * label;

    const size_t bg = (size_t)b * G + g1;
    // heading
    const int hc = (int)a.heading_class_label[bg];
    const float *hs = a.heading_scores + bk * LD(a.NH);
    acc[8] += (lse_of(hs, a.NH) - hs[hc]) * label;
    const float hres = a.heading_residual_label[bg] / (3.14159265358979323846f / (float)a.NH);
    acc[9] += huber1(a.heading_res_norm[bk * LD(a.NH) + hc] - hres) * label;
    // size
    const int sc = (int)a.size_class_label[bg];
    const float *ss = a.size_scores + bk * LD(a.NS);
    acc[10] += (lse_of(ss, a.NS) - ss[sc]) * label;
    float sr = 0.0f;
    for (int c = 0; c < 3; ++c) {
      const float lab = a.size_residual_label[bg * 3 + c] / a.mean_size_arr[sc * 3 + c];
      sr += huber1(a.size_res_norm[bk * LD(a.NS * 3) + sc * 3 + c] - lab);
    }
    acc[11] += (sr / 3.0f) * label;
    // semantic class
    const int cc = (int)a.sem_cls_label[bg];
    const float *cs = a.sem_cls_scores + bk * LD(a.NC);
    acc[12] += (lse_of(cs, a.NC) - cs[cc]) * label;
  }

  // ---- per GT box: nearest predicted centre (second chamfer direction)
  if (sec == NSEC - 1)
  for (int g = tid; g < G; g += 256) {
    const float gx = s_gt[3 * g], gy = s_gt[3 * g + 1], gz = s_gt[3 * g + 2];
    float d2 = 0.0f;
    int k2 = 0;
    for (int k = 0; k < K; ++k) {
      const float dx = s_c[3 * k] - gx, dy = s_c[3 * k + 1] - gy, dz = s_c[3 * k + 2] - gz;
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (k == 0 || d < d2) { d2 = d; k2 = k; }
    }
    a.center_k2[(size_t)b * G + g] = k2;
    const float blm = a.box_label_mask[(size_t)b * G + g];
    acc[6] += d2 * blm;
    acc[7] += blm;
  }

  block_sums(acc, s_red);
  if (tid < NPART)
    a.partial[((size_t)b * NSEC + sec) * NPART + tid] =
        (s_red[tid] + s_red[NPART + tid]) + (s_red[2 * NPART + tid] + s_red[3 * NPART + tid]);
}

// stats: [0..8] vote, objectness, center, heading_cls, heading_reg, size_cls,
// size_reg, sem_cls, box ; [9] detection total ; [10] pos_ratio ; [11] neg_ratio ;
// [12] obj_acc ; [13..16] denominators (votes, objectness mask, objectness label,
// box label mask)
__global__ void detloss_finalize_kernel(int B, int K, const float *__restrict__ partial,
                                        float *__restrict__ stats) {
  __shared__ double s_p[NPART];
  if (threadIdx.x < NPART) {
    double s = 0.0;
    for (int b = 0; b < B * NSEC; ++b) s += (double)partial[(size_t)b * NPART + threadIdx.x];
    s_p[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const float den_v = (float)s_p[1] + 1e-6f, den_o = (float)s_p[3] + 1e-6f;
  const float den_l = (float)s_p[4] + 1e-6f, den_b = (float)s_p[7] + 1e-6f;
  const float vote = (float)s_p[0] / den_v;
  const float obj = (float)s_p[2] / den_o;
  const float center = (float)s_p[5] / den_l + (float)s_p[6] / den_b;
  const float hcls = (float)s_p[8] / den_l, hreg = (float)s_p[9] / den_l;
  const float scls = (float)s_p[10] / den_l, sreg = (float)s_p[11] / den_l;
  const float sem = (float)s_p[12] / den_l;
  const float box = center + 0.1f * hcls + hreg + 0.1f * scls + sreg;   // :424-425
  stats[0] = vote; stats[1] = obj; stats[2] = center; stats[3] = hcls; stats[4] = hreg;
  stats[5] = scls; stats[6] = sreg; stats[7] = sem; stats[8] = box;
  stats[9] = (vote + 0.5f * obj + box + 0.1f * sem) * 10.0f;             // :466-468
  const float total = (float)B * (float)K;
  stats[10] = (float)s_p[4] / total;
  stats[11] = (float)s_p[3] / total - stats[10];
  stats[12] = (float)s_p[13] / den_o;
  stats[13] = den_v; stats[14] = den_o; stats[15] = den_l; stats[16] = den_b;
}

// gradient of stats[9] (times the upstream scalar *gup) w.r.t. every float input
__global__ __launch_bounds__(256) void detloss_bwd_kernel(s2c_detloss_args a,
                                                          s2c_detloss_grads d,
                                                          const float *__restrict__ gup) {
  __shared__ float s_gt[MAXG * 3];
  __shared__ float s_blm[MAXG];
  __shared__ int s_k2[MAXG];
  const int b = blockIdx.x, tid = threadIdx.x, sec = blockIdx.y;
  const int K = a.K, G = a.G;
  // sections NSEC - 2 .. NSEC + 1: the proposals, one group of output tensors each (objectness +
  // centre | heading | size | semantic class): a thread per proposal wrote ~100 strided floats
  const int part = sec - (NSEC - 2);
  if (part == 0) {
    for (int i = tid; i < G * 3; i += 256)
      s_gt[i] = a.center_label[((size_t)b * G + i / 3) * a.ld_center_label + i % 3];
    for (int i = tid; i < G; i += 256) {
      s_blm[i] = a.box_label_mask[(size_t)b * G + i];
      s_k2[i] = a.center_k2[(size_t)b * G + i];
    }
  }
  __syncthreads();
  const float up = gup[0] * 10.0f;
  const float den_v = a.stats[13], den_o = a.stats[14], den_l = a.stats[15], den_b = a.stats[16];

  if (sec < NSEC - 2)
  for (int s = sec * 256 + tid; s < a.S; s += 256 * (NSEC - 2)) {
    const int idx = a.seed_inds[(size_t)b * a.S + s];
    const float m = (float)a.vote_label_mask[(size_t)b * a.N + idx];
    const int arg = a.vote_arg[(size_t)b * a.S + s];
    const int is = arg / 3, js = arg - 3 * is;
    const float *sx = a.seed_xyz + ((size_t)b * a.S + s) * 3;
    const float *vl = a.vote_label + ((size_t)b * a.N + idx) * 9 + 3 * js;
    const float coef = up * m / den_v;
    for (int i = 0; i < a.VF; ++i) {
      const size_t e = (((size_t)b * a.S + s) * a.VF + i) * 3;
      for (int c = 0; c < 3; ++c) {
        const float diff = a.vote_xyz[e + c] - (vl[c] + sx[c]);
        const float sg = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
        d.vote_xyz[e + c] = i == is ? coef * sg : 0.0f;
      }
    }
  }

  if (part >= 0)
  for (int k = tid; k < K; k += 256) {
    const size_t bk = (size_t)b * K + k;
    const float label = (float)a.objectness_label[bk];
    const float mask = a.objectness_mask[bk];
    const int g1 = (int)a.object_assignment[bk];
    const size_t bg = (size_t)b * G + g1;
    // objectness scores (weight 0.5)
    if (part == 0) {
      const float *os = a.objectness_scores + bk * LD(2);
      const int y = label > 0.0f ? 1 : 0;
      const float w = y ? a.obj_w1 : a.obj_w0;
      const float l = lse_of(os, 2);
      const float coef = 0.5f * up * w * mask / den_o;
      d.objectness_scores[bk * LD(2) + 0] = coef * (expf(os[0] - l) - (y == 0 ? 1.0f : 0.0f));
      d.objectness_scores[bk * LD(2) + 1] = coef * (expf(os[1] - l) - (y == 1 ? 1.0f : 0.0f));
    }
    const float cl = up * label / den_l;
    // centre (both chamfer directions)
    if (part == 0) {
      const float *c = a.center + bk * 3;
      const int cg = a.center_g1[bk];
      float gx = 2.0f * (c[0] - s_gt[3 * cg]) * (label / den_l);
      float gy = 2.0f * (c[1] - s_gt[3 * cg + 1]) * (label / den_l);
      float gz = 2.0f * (c[2] - s_gt[3 * cg + 2]) * (label / den_l);
      for (int g = 0; g < G; ++g) {
        if (s_k2[g] != k) continue;
        const float wgt = s_blm[g] / den_b;
        gx += 2.0f * (c[0] - s_gt[3 * g]) * wgt;
        gy += 2.0f * (c[1] - s_gt[3 * g + 1]) * wgt;
        gz += 2.0f * (c[2] - s_gt[3 * g + 2]) * wgt;
      }
      d.center[bk * LD(3) + 0] = up * gx;
      d.center[bk * LD(3) + 1] = up * gy;
      d.center[bk * LD(3) + 2] = up * gz;
    }
    // heading (class weight 0.1, residual weight 1)
    if (part == 1) {
      const int hc = (int)a.heading_class_label[bg];
      const float *hs = a.heading_scores + bk * LD(a.NH);
      const float l = lse_of(hs, a.NH);
      const float hres = a.heading_residual_label[bg] / (3.14159265358979323846f / (float)a.NH);
      const float e = a.heading_res_norm[bk * LD(a.NH) + hc] - hres;
      for (int h = 0; h < a.NH; ++h) {
        d.heading_scores[bk * LD(a.NH) + h] = 0.1f * cl * (expf(hs[h] - l) - (h == hc ? 1.0f : 0.0f));
        d.heading_res_norm[bk * LD(a.NH) + h] = h == hc ? cl * huber1_grad(e) : 0.0f;
      }
    }
    // size (class weight 0.1, residual weight 1, mean over 3 axes)
    if (part == 2) {
      const int sc = (int)a.size_class_label[bg];
      const float *ss = a.size_scores + bk * LD(a.NS);
      const float l = lse_of(ss, a.NS);
      for (int s = 0; s < a.NS; ++s) {
        d.size_scores[bk * LD(a.NS) + s] = 0.1f * cl * (expf(ss[s] - l) - (s == sc ? 1.0f : 0.0f));
        for (int c = 0; c < 3; ++c) {
          float g = 0.0f;
          if (s == sc) {
            const float lab = a.size_residual_label[bg * 3 + c] / a.mean_size_arr[sc * 3 + c];
            g = cl * huber1_grad(a.size_res_norm[bk * LD(a.NS * 3) + sc * 3 + c] - lab) / 3.0f;
          }
          d.size_res_norm[bk * LD(a.NS * 3) + s * 3 + c] = g;
        }
      }
    }
    // semantic class (weight 0.1)
    if (part == 3) {
      const int cc = (int)a.sem_cls_label[bg];
      const float *cs = a.sem_cls_scores + bk * LD(a.NC);
      const float l = lse_of(cs, a.NC);
      for (int c = 0; c < a.NC; ++c)
        d.sem_cls_scores[bk * LD(a.NC) + c] = 0.1f * cl * (expf(cs[c] - l) - (c == cc ? 1.0f : 0.0f));
    }
  }
}

}  // namespace

static int chk3(const char *k) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c: %s launch failed: %s\n", k, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

static bool bad_args(const s2c_detloss_args *a) {
  return !a || a->B <= 0 || a->S <= 0 || a->VF <= 0 || a->K <= 0 || a->K > MAXK ||
         a->G <= 0 || a->G > MAXG || a->NH <= 0 || a->NS <= 0 || a->NC <= 0 ||
         a->NH > 64 || a->NS > 64 || a->NC > 64 || a->ld_center_label < 3;
}

extern "C" int s2c_detection_loss_partial_floats(void) { return NPART * NSEC; }

extern "C" int s2c_detection_loss_fwd(const s2c_detloss_args *a, void *stream) {
  if (bad_args(a)) return -1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(detloss_fwd_kernel, dim3(a->B, NSEC), dim3(256), 0, st, *a);
  hipLaunchKernelGGL(detloss_finalize_kernel, dim3(1), dim3(64), 0, st, a->B, a->K,
                     a->partial, a->stats);
  return chk3("detection_loss_fwd");
}

extern "C" int s2c_detection_loss_bwd(const s2c_detloss_args *a,
                                      const s2c_detloss_grads *d, const float *gup,
                                      void *stream) {
  if (bad_args(a) || !d || !gup) return -1;
  hipLaunchKernelGGL(detloss_bwd_kernel, dim3(a->B, NSEC + 2), dim3(256), 0, (hipStream_t)stream,
                     *a, *d, gup);
  return chk3("detection_loss_bwd");
}

// ---------------------------------------------------------------------------------------
// Caption loss (lib/loss_helper.py:189-230): masked cross-entropy of the teacher-forced
// logits + word accuracy, forward in two launches and backward in one instead of ~40
// framework kernels (log_softmax, nll, masks, sums, arg-max, ...).
//   ce[r]    = -log_softmax(pred[r])[target[r]]           (0 where target == 0: ignore_index)
//   cap_loss = sum_r ce[r] * good[b(r)] / (sum_r good[b(r)] + 1e-6)
//   cap_acc  = #(argmax == target, target != 0, good) / #(target != 0, good)   (0 if none)
namespace {

constexpr int CAP_T = 256;

__device__ __forceinline__ float block_reduce_sum(float v, float *s) {
  const int t = threadIdx.x;
  s[t] = v;
  __syncthreads();
  for (int k = CAP_T / 2; k > 0; k >>= 1) {
    if (t < k) s[t] += s[t + k];
    __syncthreads();
  }
  const float r = s[0];
  __syncthreads();
  return r;
}

// block = one (sample, word) row of V logits
__global__ __launch_bounds__(CAP_T) void caploss_rows_kernel(
    int V, int T, const float *__restrict__ pred, const long long *__restrict__ target,
    long long target_stride, const unsigned char *__restrict__ good,
    float *__restrict__ row_lse, float *__restrict__ row_stats) {
  __shared__ float s_v[CAP_T];
  __shared__ int s_i[CAP_T];
  const int r = blockIdx.x, t = threadIdx.x;
  const int b = r / T, w = r - b * T;
  const float *x = pred + (size_t)r * V;
  const long long tg = target[(size_t)b * target_stride + w];
  float m = -INFINITY;
  int am = 0x7fffffff;
  for (int v = t; v < V; v += CAP_T) {
    const float xv = x[v];
    if (xv > m || (xv != xv && m == m)) { m = xv; am = v; }
  }
  s_v[t] = m;
  s_i[t] = am;
  __syncthreads();
  for (int k = CAP_T / 2; k > 0; k >>= 1) {
    if (t < k) {
      const float o = s_v[t + k];
      const int oi = s_i[t + k];
      if (o > s_v[t] || (o == s_v[t] && oi < s_i[t])) { s_v[t] = o; s_i[t] = oi; }
    }
    __syncthreads();
  }
  m = s_v[0];
  am = s_i[0];
  __syncthreads();
  float s = 0.f;
  for (int v = t; v < V; v += CAP_T) s += expf(x[v] - m);
  s = block_reduce_sum(s, s_v);
  if (t == 0) {
    const float ls = logf(s);
    const bool g = good[b] != 0, live = tg != 0;
    const float ce = live ? -((x[tg] - m) - ls) : 0.0f;
    row_lse[r] = m + ls;
    float *o = row_stats + (size_t)r * 4;
    o[0] = g ? ce : 0.0f;
    o[1] = g ? 1.0f : 0.0f;
    o[2] = (g && live && (long long)am == tg) ? 1.0f : 0.0f;
    o[3] = (g && live) ? 1.0f : 0.0f;
  }
}

// out[0] = cap_loss, out[1] = cap_acc, out[2] = 1 / (sum good + 1e-6)
__global__ __launch_bounds__(CAP_T) void caploss_finalize_kernel(
    int rows, const float *__restrict__ row_stats, float *__restrict__ out) {
  __shared__ double s_d[4][CAP_T];
  const int t = threadIdx.x;
  double a[4] = {0, 0, 0, 0};
  for (int r = t; r < rows; r += CAP_T)
    for (int k = 0; k < 4; ++k) a[k] += (double)row_stats[(size_t)r * 4 + k];
  for (int k = 0; k < 4; ++k) s_d[k][t] = a[k];
  __syncthreads();
  for (int k = CAP_T / 2; k > 0; k >>= 1) {
    if (t < k)
      for (int q = 0; q < 4; ++q) s_d[q][t] += s_d[q][t + k];
    __syncthreads();
  }
  if (t == 0) {
    const float s1 = (float)s_d[0][0], s2 = (float)s_d[1][0];
    const float hits = (float)s_d[2][0], n = (float)s_d[3][0];
    out[0] = s1 / (s2 + 1e-6f);
    out[1] = n > 0.0f ? hits / fmaxf(n, 1.0f) : 0.0f;
    out[2] = 1.0f / (s2 + 1e-6f);
  }
}

// d pred[r, v] = gup * good * [target != 0] / (sum good + 1e-6) * (softmax(pred[r])[v] - [v == target])
__global__ __launch_bounds__(CAP_T) void caploss_bwd_kernel(
    int V, int T, const float *__restrict__ pred, const long long *__restrict__ target,
    long long target_stride, const unsigned char *__restrict__ good,
    const float *__restrict__ row_lse, const float *__restrict__ fwd_out,
    const float *__restrict__ gup, float *__restrict__ dpred) {
  const int r = blockIdx.x, t = threadIdx.x;
  const int b = r / T, w = r - b * T;
  const long long tg = target[(size_t)b * target_stride + w];
  const float coef = (good[b] != 0 && tg != 0) ? gup[0] * fwd_out[2] : 0.0f;
  const float *x = pred + (size_t)r * V;
  float *d = dpred + (size_t)r * V;
  const float lse = row_lse[r];
  for (int v = t; v < V; v += CAP_T) {
    float g = 0.0f;
    if (coef != 0.0f) g = coef * (expf(x[v] - lse) - (v == tg ? 1.0f : 0.0f));
    d[v] = g;
  }
}

}  // namespace

extern "C" int s2c_caption_loss_fwd(int B, int T, int V, const float *pred,
                                    const long long *target, long long target_stride,
                                    const unsigned char *good, float *row_lse,
                                    float *row_stats, float *out, void *stream) {
  if (B <= 0 || T <= 0 || V <= 0 || !pred || !target || !good || !row_lse || !row_stats || !out)
    return -1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(caploss_rows_kernel, dim3(B * T), dim3(CAP_T), 0, st, V, T, pred, target,
                     target_stride, good, row_lse, row_stats);
  hipLaunchKernelGGL(caploss_finalize_kernel, dim3(1), dim3(CAP_T), 0, st, B * T, row_stats, out);
  return chk3("caption_loss_fwd");
}

extern "C" int s2c_caption_loss_bwd(int B, int T, int V, const float *pred,
                                    const long long *target, long long target_stride,
                                    const unsigned char *good, const float *row_lse,
                                    const float *fwd_out, const float *gup, float *dpred,
                                    void *stream) {
  if (B <= 0 || T <= 0 || V <= 0 || !pred || !target || !good || !row_lse || !fwd_out || !gup ||
      !dpred)
    return -1;
  hipLaunchKernelGGL(caploss_bwd_kernel, dim3(B * T), dim3(CAP_T), 0, (hipStream_t)stream, V, T,
                     pred, target, target_stride, good, row_lse, fwd_out, gup, dpred);
  return chk3("caption_loss_bwd");
}
