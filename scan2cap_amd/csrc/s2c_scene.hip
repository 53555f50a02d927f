// s2c_scene.hip -- on-device assembly of training items from HBM-resident scenes
// (reference: lib/dataset.py:320-540 in numpy on DataLoader workers).
//
// The whole ScanNet training set (1201 scenes x ~150k vertices x (9 + 128) floats ~ 100 GB)
// fits in one MI355X's 288 GB, so a batch is gathered by the GPU from resident scenes
// instead of being assembled on the host and copied (173 MB/step at cfg3): the only
// per-step host traffic is the random draws (B*N vertex indices + 32 doubles per item).
//
// Arithmetic follows numpy's exactly: float32 data, float64 rotation matrices applied as
// a dgemm (fused multiply-add chain over k = 0,1,2) and rounded back to float32 after
// each rotation, float64 translation, float32 votes.
#include "s2c_common.h"
#include "../../include/s2c_scene.h"

#include <stdio.h>

namespace {

constexpr int MAXOBJ = S2C_SCENE_MAX_NUM_OBJ;
constexpr int MAXINST = S2C_SCENE_MAX_INSTANCE;

// order-preserving float <-> unsigned maps
__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ---------------------------------------------------------------------------------
// np.percentile(z, 0.99): radix select (11 + 11 + 10 bits) of the two neighbouring order
// statistics, then numpy's float32 interpolation (numpy/lib/_function_base_impl.py:
// _QuantileMethods["linear"], _get_gamma, _lerp).  One workgroup; runs once per scene.
__global__ __launch_bounds__(1024) void floor_height_kernel(long long nv,
                                                            const float *__restrict__ verts,
                                                            int cols, float *__restrict__ out) {
  __shared__ unsigned hist[2048];
  __shared__ unsigned s_prefix, s_mask, s_eq, s_next;
  __shared__ long long s_rank;
  const int t = threadIdx.x;
  const float q = 0.99f / 100.0f;
  const float vi = (float)(nv - 1) * q;   // method "linear": (n - 1) * quantiles
  float prevf = floorf(vi);
  long long k = (long long)prevf;
  bool clamp_hi = vi >= (float)(nv - 1);
  if (clamp_hi) k = nv - 1;
  if (t == 0) { s_prefix = 0; s_mask = 0; s_rank = k; }
  const int shifts[3] = {21, 10, 0};
  const int nbits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    for (int i = t; i < 2048; i += 1024) hist[i] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix, mask = s_mask;
    const int sh = shifts[pass];
    const unsigned bm = (1u << nbits[pass]) - 1u;
    for (long long i = t; i < nv; i += 1024) {
      const unsigned key = f2key(verts[i * cols + 2]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> sh) & bm], 1u);
    }
    __syncthreads();
    if (t == 0) {
      long long r = s_rank;
      unsigned bin = 0;
      for (; bin < bm; ++bin) {
        const unsigned c = hist[bin];
        if (r < (long long)c) break;
        r -= c;
      }
      s_rank = r;
      s_prefix = prefix | (bin << sh);
      s_mask = mask | (bm << sh);
      s_eq = hist[bin];
    }
    __syncthreads();
  }
  const unsigned key_k = s_prefix;
  // (k+1)-th order statistic: the same value if it repeats, else the smallest larger key
  if (t == 0) s_next = 0xFFFFFFFFu;
  __syncthreads();
  const bool repeated = s_rank + 1 < (long long)s_eq;
  if (!repeated && !clamp_hi) {
    unsigned best = 0xFFFFFFFFu;
    for (long long i = t; i < nv; i += 1024) {
      const unsigned key = f2key(verts[i * cols + 2]);
      if (key > key_k && key < best) best = key;
    }
    atomicMin(&s_next, best);
  }
  __syncthreads();
  if (t == 0) {
    const float a = key2f(key_k);
    const float b = (repeated || clamp_hi) ? a : key2f(s_next);
    const float g = clamp_hi ? 0.0f : vi - prevf;
    const float diff = b - a;
    float r = a + diff * g;
    if (g >= 0.5f) r = b - diff * (1.0f - g);
    out[0] = r;
  }
}

// ---------------------------------------------------------------------------------
// Vertex sampling on the device (utils/pc_utils.py:32-40: N distinct vertices of Nv in random
// order; with replacement only when Nv < N).  choices[i] = pi(i), i < N, where pi is a keyed
// pseudo-random PERMUTATION of [0, Nv): a 6-round balanced Feistel network over 2h >= log2(Nv)
// bits, restricted to [0, Nv) by cycle walking (<= 4 applications on average).  O(N) work,
// no sort, distinct by construction.
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void scene_sample_kernel(
    int N, const long long *__restrict__ vert_off, const int *__restrict__ scene_ids,
    const unsigned long long *__restrict__ seeds, long long *__restrict__ choices) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int scene = scene_ids[b];
  const unsigned long long nv = (unsigned long long)(vert_off[scene + 1] - vert_off[scene]);
  const unsigned long long seed = seeds[b];
  const unsigned k0 = mix32((unsigned)seed), k1 = mix32((unsigned)(seed >> 32) ^ 0x9e3779b9u);
  long long out;
  if (nv < (unsigned long long)N) {
    // with replacement: independent uniform draws
    const unsigned r = mix32(mix32((unsigned)i ^ k0) + k1);
    out = (long long)(((unsigned long long)r * nv) >> 32);
  } else {
    int h = 1;
    while ((1ull << (2 * h)) < nv) ++h;        // nv < 2^31 => h <= 16
    const unsigned mask = (1u << h) - 1u;
    unsigned x = (unsigned)i;
    do {
      unsigned L = x >> h, R = x & mask;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const unsigned key = (r & 1 ? k1 : k0) + 0x9e3779b9u * (unsigned)(r + 1);
        const unsigned t = L ^ (mix32(R ^ key) & mask);
        L = R;
        R = t;
      }
      x = (L << h) | R;
    } while ((unsigned long long)x >= nv);
    out = (long long)x;
  }
  choices[(size_t)b * N + i] = out;
}

// ---------------------------------------------------------------------------------
// one rotation of numpy's `np.dot(xyz_f32, R.T)` assigned back into a float32 array
__device__ __forceinline__ void rot_f32(float &x, float &y, float &z, const double *R) {
  const double dx = x, dy = y, dz = z;
  const float nx = (float)fma(dz, R[2], fma(dy, R[1], dx * R[0]));
  const float ny = (float)fma(dz, R[5], fma(dy, R[4], dx * R[3]));
  const float nz = (float)fma(dz, R[8], fma(dy, R[7], dx * R[6]));
  x = nx; y = ny; z = nz;
}

constexpr int GROWS = 32;   // cloud rows per workgroup

// The workgroup's GROWS output rows are one contiguous span of the cloud: they are put
// together in LDS (row thread: xyz chain + the small channels; all threads: the multiview
// row as 16-byte loads) and leave as 16-byte coalesced stores.
__global__ __launch_bounds__(256) void scene_gather_kernel(
    int N, int cols, int Cm, int use_color, int use_normal, int use_height, int augment,
    int Cout, const float *__restrict__ verts, const float *__restrict__ mv,
    const long long *__restrict__ vert_off, const float *__restrict__ floor_h,
    const int *__restrict__ scene_ids, const long long *__restrict__ choices,
    const double *__restrict__ aug, float *__restrict__ cloud) {
  extern __shared__ __align__(16) float s_tile[];   // GROWS x Cout
  __shared__ long long s_src[GROWS];
  const int b = blockIdx.y, t = threadIdx.x;
  const int row0 = blockIdx.x * GROWS;
  const int nrows = min(GROWS, N - row0);
  const int scene = scene_ids[b];
  const long long voff = vert_off[scene];
  const int c_nrm = 3 + (use_color ? 3 : 0), c_mv = c_nrm + (use_normal ? 3 : 0);
  const int c_h = c_mv + Cm;
  if (t < nrows) {
    const long long v = voff + choices[(size_t)b * N + row0 + t];
    const float *p = verts + v * cols;
    float x = p[0], y = p[1], z = p[2];
    float *row = s_tile + t * Cout;
    s_src[t] = v;
    if (use_color) {
      row[3] = (float)(((double)p[3] - 109.8) / 256.0);   // MEAN_COLOR_RGB, lib/dataset.py:28
      row[4] = (float)(((double)p[4] - 97.2) / 256.0);
      row[5] = (float)(((double)p[5] - 83.8) / 256.0);
    }
    if (use_normal) {
      row[c_nrm] = p[6]; row[c_nrm + 1] = p[7]; row[c_nrm + 2] = p[8];
    }
    if (use_height) row[c_h] = z - floor_h[scene];
    if (augment) {
      const double *A = aug + (size_t)b * 32;
      if (A[0] != 0.0) x = -1.0f * x;
      if (A[1] != 0.0) y = -1.0f * y;
      rot_f32(x, y, z, A + 2);
      rot_f32(x, y, z, A + 11);
      rot_f32(x, y, z, A + 20);
      x = (float)((double)x + A[29]);
      y = (float)((double)y + A[30]);
      z = (float)((double)z + A[31]);
    }
    row[0] = x; row[1] = y; row[2] = z;
  }
  __syncthreads();
  if (Cm > 0) {
    if ((Cm & 3) == 0) {
      const int q4 = Cm >> 2, nf4 = nrows * q4;
#pragma unroll 4
      for (int f = t; f < nf4; f += 256) {
        const int r = f / q4, q = f - r * q4;
        const float4 v = *reinterpret_cast<const float4 *>(mv + s_src[r] * Cm + 4 * q);
        float *o = s_tile + r * Cout + c_mv + 4 * q;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
      }
    } else {
      const int tot = nrows * Cm;
      for (int f = t; f < tot; f += 256) {
        const int r = f / Cm, q = f - r * Cm;
        s_tile[r * Cout + c_mv + q] = mv[s_src[r] * Cm + q];
      }
    }
  }
  __syncthreads();
  float *dst = cloud + ((size_t)b * N + row0) * Cout;
  const int total = nrows * Cout;
  if ((((size_t)dst & 15) == 0) && (total & 3) == 0) {
    const float4 *s4 = reinterpret_cast<const float4 *>(s_tile);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (int e = t; e < (total >> 2); e += 256) d4[e] = s4[e];
  } else {
    for (int e = t; e < total; e += 256) dst[e] = s_tile[e];
  }
}

// ---------------------------------------------------------------------------------
// votes, pass 1: per-instance min / max / first sampled point.  Grid (VCHUNKS, B): every
// workgroup reduces its slice of the item in LDS and merges the instances it met into the
// item's table in global memory (tab_lo (B,MAXINST,4) = min x,y,z keys + first index,
// memset to 0xFF; tab_hi (B,MAXINST,3) = max keys, memset to 0).
constexpr int VCHUNKS = 32;

__global__ __launch_bounds__(512) void scene_votes_minmax_kernel(
    int N, int Cout, const float *__restrict__ cloud, const int *__restrict__ ins,
    const long long *__restrict__ vert_off, const int *__restrict__ scene_ids,
    const long long *__restrict__ choices, unsigned *__restrict__ tab_lo,
    unsigned *__restrict__ tab_hi) {
  __shared__ unsigned s_lo[MAXINST][4];
  __shared__ unsigned s_hi[MAXINST][3];
  const int b = blockIdx.y, t = threadIdx.x;
  const long long voff = vert_off[scene_ids[b]];
  const long long *ch = choices + (size_t)b * N;
  const float *pc = cloud + (size_t)b * N * Cout;
  for (int i = t; i < MAXINST; i += 512) {
    s_lo[i][0] = s_lo[i][1] = s_lo[i][2] = s_lo[i][3] = 0xFFFFFFFFu;
    s_hi[i][0] = s_hi[i][1] = s_hi[i][2] = 0u;
  }
  __syncthreads();
  const int per = (N + VCHUNKS - 1) / VCHUNKS;
  const int i0 = blockIdx.x * per, i1 = min(N, i0 + per);
  for (int i = i0 + t; i < i1; i += 512) {
    const int inst = ins[voff + ch[i]] & (MAXINST - 1);
    const float *p = pc + (size_t)i * Cout;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned key = f2key(p[c]);
      atomicMin(&s_lo[inst][c], key);
      atomicMax(&s_hi[inst][c], key);
    }
    atomicMin(&s_lo[inst][3], (unsigned)i);
  }
  __syncthreads();
  unsigned *glo = tab_lo + (size_t)b * MAXINST * 4;
  unsigned *ghi = tab_hi + (size_t)b * MAXINST * 3;
  for (int i = t; i < MAXINST; i += 512) {
    if (s_lo[i][3] == 0xFFFFFFFFu) continue;      // instance not in this slice
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      atomicMin(&glo[i * 4 + c], s_lo[i][c]);
      atomicMax(&ghi[i * 3 + c], s_hi[i][c]);
    }
    atomicMin(&glo[i * 4 + 3], s_lo[i][3]);
  }
}

// votes, pass 2: thread per point
__global__ __launch_bounds__(256) void scene_votes_apply_kernel(
    int N, int Cout, const float *__restrict__ cloud, const int *__restrict__ ins,
    const int *__restrict__ sem, const long long *__restrict__ vert_off,
    const int *__restrict__ scene_ids, const long long *__restrict__ choices,
    unsigned long long vote_mask, const unsigned *__restrict__ tab_lo,
    const unsigned *__restrict__ tab_hi, float *__restrict__ vote_label,
    long long *__restrict__ vote_label_mask) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const long long voff = vert_off[scene_ids[b]];
  const long long *ch = choices + (size_t)b * N;
  const int inst = ins[voff + ch[i]] & (MAXINST - 1);
  const unsigned *lo = tab_lo + ((size_t)b * MAXINST + inst) * 4;
  const unsigned *hi = tab_hi + ((size_t)b * MAXINST + inst) * 3;
  const unsigned s = (unsigned)sem[voff + ch[lo[3]]];
  const bool votes = s < 64u && ((vote_mask >> s) & 1ull);
  const float *p = cloud + ((size_t)b * N + i) * Cout;
  float v[3] = {0.0f, 0.0f, 0.0f};
  if (votes) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float centre = 0.5f * (key2f(lo[c]) + key2f(hi[c]));
      v[c] = centre - p[c];
    }
  }
  float *o = vote_label + ((size_t)b * N + i) * 9;
#pragma unroll
  for (int rep = 0; rep < 3; ++rep) {
    o[rep * 3 + 0] = v[0]; o[rep * 3 + 1] = v[1]; o[rep * 3 + 2] = v[2];
  }
  vote_label_mask[(size_t)b * N + i] = votes ? 1 : 0;
}

// ---------------------------------------------------------------------------------
// box labels: workgroup per item, thread per box slot
__device__ __forceinline__ void rot_box(double *bx, const double *R, int axis) {
  // model_util_scannet.py:47-79
  const double c0 = bx[0], c1 = bx[1], c2 = bx[2];
  const int ia = (axis == 0) ? 1 : 0, ib = (axis == 2) ? 1 : 2;
  const double ha = bx[3 + ia] / 2.0, hb = bx[3 + ib] / 2.0;
  double ea = 0.0, eb = 0.0;
  const double sa[4] = {-1, 1, 1, -1}, sb[4] = {-1, -1, 1, 1};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double v0 = sa[k] * ha, v1 = sb[k] * hb;
    const double o0 = fma(0.0, R[2], fma(v1, R[1], v0 * R[0]));
    const double o1 = fma(0.0, R[5], fma(v1, R[4], v0 * R[3]));
    ea = (k == 0) ? o0 : fmax(ea, o0);
    eb = (k == 0) ? o1 : fmax(eb, o1);
  }
  bx[0] = fma(c2, R[2], fma(c1, R[1], c0 * R[0]));
  bx[1] = fma(c2, R[5], fma(c1, R[4], c0 * R[3]));
  bx[2] = fma(c2, R[8], fma(c1, R[7], c0 * R[6]));
  bx[3 + ia] = 2.0 * ea;
  bx[3 + ib] = 2.0 * eb;
}

__global__ __launch_bounds__(MAXOBJ) void scene_box_labels_kernel(
    int augment, const double *__restrict__ boxes, const int *__restrict__ box_off,
    const float *__restrict__ box_rot, const unsigned char *__restrict__ box_rot_mask,
    const int *__restrict__ scene_ids, const long long *__restrict__ object_ids,
    const double *__restrict__ aug, const int *__restrict__ class_of,
    const double *__restrict__ mean_size, s2c_scene_labels o) {
  __shared__ int s_ref;
  const int b = blockIdx.x, i = threadIdx.x;
  const int scene = scene_ids[b];
  const int boff = box_off[scene];
  const int nb = min(box_off[scene + 1] - boff, MAXOBJ);
  const bool live = i < nb;
  if (i == 0) s_ref = -1;
  __syncthreads();
  const double *src = boxes + (size_t)(boff + i) * 8;
  double bx[6] = {0, 0, 0, 0, 0, 0};
  double nyu = 0.0, oid = 0.0;
  if (live) {
#pragma unroll
    for (int c = 0; c < 6; ++c) bx[c] = src[c];
    nyu = src[6];
    oid = src[7];
  }
  if (augment) {
    const double *A = aug + (size_t)b * 32;
    if (A[0] != 0.0) bx[0] = -1.0 * bx[0];
    if (A[1] != 0.0) bx[1] = -1.0 * bx[1];
    rot_box(bx, A + 2, 0);
    rot_box(bx, A + 11, 1);
    rot_box(bx, A + 20, 2);
    bx[0] += A[29]; bx[1] += A[30]; bx[2] += A[31];
  }
  int cls = 0;
  double res[3] = {0, 0, 0}, size[3] = {0, 0, 0};
  if (live) {
    const int id = (int)nyu;
    cls = (id >= 0 && id <= 40) ? class_of[id] : -1;
    if (cls < 0) cls = 0;   // rejected by the host when the scene is registered
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      res[c] = bx[3 + c] - mean_size[cls * 3 + c];
      size[c] = mean_size[cls * 3 + c] + res[c];
    }
  }
  const bool is_ref = live && oid == (double)object_ids[b];
  if (is_ref) atomicMax(&s_ref, i);
  const size_t bi = (size_t)b * MAXOBJ + i;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o.center_label[bi * 3 + c] = (float)bx[c];
    o.size_residual_label[bi * 3 + c] = (float)res[c];
  }
  o.size_class_label[bi] = cls;
  o.sem_cls_label[bi] = cls;
  o.scene_object_ids[bi] = live ? (long long)oid : 0;
  o.gt_box_object_ids[bi] = live ? (long long)oid : 0;
  o.box_label_mask[bi] = live ? 1.0f : 0.0f;
  o.gt_box_masks[bi] = live ? 1 : 0;
  o.ref_box_label[bi] = is_ref ? 1 : 0;
  const bool has_rot = live && box_rot_mask && box_rot_mask[boff + i];
  o.scene_object_rotation_masks[bi] = has_rot ? 1 : 0;
#pragma unroll
  for (int c = 0; c < 9; ++c)
    o.scene_object_rotations[bi * 9 + c] = has_rot ? box_rot[(size_t)(boff + i) * 9 + c] : 0.0f;
  // corners: utils/box_util.py:350-357 at heading 0
  const double sx[8] = {1, 1, -1, -1, 1, 1, -1, -1};
  const double sy[8] = {1, -1, -1, 1, 1, -1, -1, 1};
  const double sz[8] = {1, 1, 1, 1, -1, -1, -1, -1};
  double corner[24];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    corner[k * 3 + 0] = live ? sx[k] * (size[0] / 2) + bx[0] : 0.0;
    corner[k * 3 + 1] = live ? sy[k] * (size[1] / 2) + bx[1] : 0.0;
    corner[k * 3 + 2] = live ? sz[k] * (size[2] / 2) + bx[2] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < 24; ++k) o.gt_box_corner_label[bi * 24 + k] = corner[k];
  __syncthreads();
  const int ref = s_ref;
  if (i == 0) o.num_bbox[b] = nb;
  if (ref < 0) {
    if (i == 0) {
      for (int c = 0; c < 3; ++c) {
        o.ref_center_label[b * 3 + c] = 0.0f;
        o.ref_size_residual_label[b * 3 + c] = 0.0f;
      }
      o.ref_size_class_label[b] = 0;
      for (int k = 0; k < 24; ++k) o.ref_box_corner_label[(size_t)b * 24 + k] = 0.0;
    }
  } else if (i == ref) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      o.ref_center_label[b * 3 + c] = (float)bx[c];
      o.ref_size_residual_label[b * 3 + c] = (float)res[c];
    }
    o.ref_size_class_label[b] = cls;
#pragma unroll
    for (int k = 0; k < 24; ++k) o.ref_box_corner_label[(size_t)b * 24 + k] = corner[k];
  }
}

int chk5(const char *k) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c: %s launch failed: %s\n", k, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // namespace

extern "C" int s2c_scene_floor_height(long long nv, const float *verts, int vert_cols,
                                      float *floor, void *stream) {
  if (nv <= 0 || !verts || vert_cols < 3 || !floor) return -1;
  hipLaunchKernelGGL(floor_height_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, nv,
                     verts, vert_cols, floor);
  return chk5("scene_floor_height");
}

extern "C" int s2c_scene_sample(int B, int N, const long long *vert_off, const int *scene_ids,
                                const unsigned long long *seeds, long long *choices,
                                void *stream) {
  if (B <= 0 || N <= 0 || !vert_off || !scene_ids || !seeds || !choices) return -1;
  hipLaunchKernelGGL(scene_sample_kernel, dim3((N + 255) / 256, B), dim3(256), 0,
                     (hipStream_t)stream, N, vert_off, scene_ids, seeds, choices);
  return chk5("scene_sample");
}

extern "C" int s2c_scene_gather(int B, int N, int vert_cols, int Cm, int use_color,
                                int use_normal, int use_multiview, int use_height,
                                int augment, const float *verts, const float *mv,
                                const long long *vert_off, const float *floor,
                                const int *scene_ids, const long long *choices,
                                const double *aug, float *cloud, void *stream) {
  if (B <= 0 || N <= 0 || !verts || !vert_off || !scene_ids || !choices || !cloud) return -1;
  if (vert_cols < 3 || (use_color && vert_cols < 6) || (use_normal && vert_cols < 9)) return -1;
  if (use_multiview && (!mv || Cm <= 0)) return -1;
  if (use_height && !floor) return -1;
  if (augment && !aug) return -1;
  const int cm = use_multiview ? Cm : 0;
  const int Cout = 3 + (use_color ? 3 : 0) + (use_normal ? 3 : 0) + cm + (use_height ? 1 : 0);
  const size_t lds = sizeof(float) * (size_t)GROWS * Cout;
  if (lds > 60 * 1024) return -1;
  hipLaunchKernelGGL(scene_gather_kernel, dim3((N + GROWS - 1) / GROWS, B), dim3(256), lds,
                     (hipStream_t)stream, N, vert_cols, cm, use_color, use_normal, use_height,
                     augment, Cout, verts, mv, vert_off, floor, scene_ids, choices, aug, cloud);
  return chk5("scene_gather");
}

extern "C" long long s2c_scene_votes_workspace_bytes(int B) {
  return B > 0 ? (long long)B * MAXINST * 7 * 4 : 0;
}

extern "C" int s2c_scene_votes(int B, int N, int Cout, const float *cloud, const int *ins,
                               const int *sem, const long long *vert_off,
                               const int *scene_ids, const long long *choices,
                               unsigned long long vote_id_mask, void *workspace,
                               float *vote_label, long long *vote_label_mask, void *stream) {
  if (B <= 0 || N <= 0 || Cout < 3 || !cloud || !ins || !sem || !vert_off || !scene_ids ||
      !choices || !workspace || !vote_label || !vote_label_mask)
    return -1;
  hipStream_t st = (hipStream_t)stream;
  unsigned *tab_lo = (unsigned *)workspace;
  unsigned *tab_hi = tab_lo + (size_t)B * MAXINST * 4;
  if (hipMemsetAsync(tab_lo, 0xFF, sizeof(unsigned) * (size_t)B * MAXINST * 4, st) != hipSuccess ||
      hipMemsetAsync(tab_hi, 0, sizeof(unsigned) * (size_t)B * MAXINST * 3, st) != hipSuccess)
    return -1;
  hipLaunchKernelGGL(scene_votes_minmax_kernel, dim3(VCHUNKS, B), dim3(512), 0, st, N, Cout,
                     cloud, ins, vert_off, scene_ids, choices, tab_lo, tab_hi);
  hipLaunchKernelGGL(scene_votes_apply_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, N,
                     Cout, cloud, ins, sem, vert_off, scene_ids, choices, vote_id_mask, tab_lo,
                     tab_hi, vote_label, vote_label_mask);
  return chk5("scene_votes");
}

extern "C" int s2c_scene_box_labels(int B, int augment, const double *boxes,
                                    const int *box_off, const float *box_rot,
                                    const unsigned char *box_rot_mask, const int *scene_ids,
                                    const long long *object_ids, const double *aug,
                                    const int *class_of_nyu40, const double *mean_size,
                                    s2c_scene_labels out, void *stream) {
  if (B <= 0 || !boxes || !box_off || !scene_ids || !object_ids || !class_of_nyu40 ||
      !mean_size || (augment && !aug))
    return -1;
  const void *need[] = {out.center_label, out.size_class_label, out.size_residual_label,
                        out.sem_cls_label, out.scene_object_ids, out.scene_object_rotations,
                        out.scene_object_rotation_masks, out.box_label_mask,
                        out.ref_box_label, out.gt_box_corner_label, out.gt_box_masks,
                        out.gt_box_object_ids, out.num_bbox, out.ref_center_label,
                        out.ref_size_class_label, out.ref_size_residual_label,
                        out.ref_box_corner_label};
  for (const void *p : need)
    if (!p) return -1;
  hipLaunchKernelGGL(scene_box_labels_kernel, dim3(B), dim3(MAXOBJ), 0, (hipStream_t)stream,
                     augment, boxes, box_off, box_rot, box_rot_mask, scene_ids, object_ids, aug,
                     class_of_nyu40, mean_size, out);
  return chk5("scene_box_labels");
}
