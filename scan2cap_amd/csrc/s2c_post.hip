// s2c_post.hip -- device side of `parse_predictions` (lib/ap_helper.py:40-178):
// the empty-box test and the greedy 3-D / 2-D NMS.
//
// Reference: for every predicted box a scipy Delaunay hull membership test over
// all N points on the CPU (ap_helper.py:92-103 -> model_util_scannet.py:13-22), then
// a numpy greedy NMS per scene (utils/nms.py:72-151), after D2H copies of every
// head output.  Boxes are cuboids rotated about the Y axis (utils/box_util.py:340-358),
// so hull membership is three interval tests in the box frame.
// All arithmetic in float64 like numpy's.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>

namespace {

constexpr int PIB_TILE = 2048;   // points per block

// counts[b,k] += #points of the block's tile inside box k (closed intervals).
// grid (ceil(n / PIB_TILE), b); thread = box (strided when K > 256).
__global__ __launch_bounds__(256) void boxes_count_points_kernel(
    int n, int K, const float *__restrict__ pts, long long pt_stride, long long pt_bstride,
    const double *__restrict__ center, const double *__restrict__ size,
    const double *__restrict__ angle, int *__restrict__ counts) {
  __shared__ float s_p[PIB_TILE * 3];
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * PIB_TILE;
  const int np = min(PIB_TILE, n - p0);
  for (int i = threadIdx.x; i < np * 3; i += 256) {
    const int p = i / 3, c = i - 3 * p;
    s_p[i] = pts[(long long)b * pt_bstride + (long long)(p0 + p) * pt_stride + c];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256) {
    const size_t bk = (size_t)b * K + k;
    const double cx = center[bk * 3], cy = center[bk * 3 + 1], cz = center[bk * 3 + 2];
    const double hl = 0.5 * size[bk * 3], hw = 0.5 * size[bk * 3 + 1], hh = 0.5 * size[bk * 3 + 2];
    const double a = angle[bk];
    const double ca = cos(a), sa = sin(a);
    int cnt = 0;
    for (int p = 0; p < np; ++p) {
      const double dx = (double)s_p[3 * p] - cx, dy = (double)s_p[3 * p + 1] - cy,
                   dz = (double)s_p[3 * p + 2] - cz;
      // corners = roty(a) * local + center, roty = [[c,0,s],[0,1,0],[-s,0,c]]
      // => local = roty(a)^T * d
      const double lx = ca * dx - sa * dz;
      const double lz = sa * dx + ca * dz;
      cnt += (fabs(lx) <= hl && fabs(dy) <= hw && fabs(lz) <= hh) ? 1 : 0;
    }
    if (cnt) atomicAdd(counts + bk, cnt);
  }
}

// Greedy NMS of one scene (utils/nms.py:13-151).  boxes (K x 6) = [x1,y1,z1,x2,y2,z2]
// (for the 2-D variant the caller passes the x / z extents in slots 0,1,3,4 and
// z1 = 0, z2 = 1), score (K), cls (K) or NULL, valid (K) 0/1 -> keep (K) 0/1.
// Order = descending score, ties: higher index first (np.argsort ascending, picks
// taken from the end).  Block = scene, thread = box (K <= 1024).
__global__ __launch_bounds__(1024) void nms_kernel(
    int K, const double *__restrict__ boxes, const double *__restrict__ score,
    const long long *__restrict__ cls, const unsigned char *__restrict__ valid,
    double thresh, int old_type, int add_eps, unsigned char *__restrict__ keep) {
  __shared__ double s_sc[1024];
  __shared__ int s_order[1024];
  __shared__ unsigned char s_alive[1024];
  const int b = blockIdx.x, t = threadIdx.x;
  const double *bx = boxes + (size_t)b * K * 6;
  double x1 = 0, y1 = 0, z1 = 0, x2 = 0, y2 = 0, z2 = 0, area = 0, sc = 0;
  long long mycls = 0;
  bool live = false;
  if (t < K) {
    x1 = bx[t * 6]; y1 = bx[t * 6 + 1]; z1 = bx[t * 6 + 2];
    x2 = bx[t * 6 + 3]; y2 = bx[t * 6 + 4]; z2 = bx[t * 6 + 5];
    area = (x2 - x1) * (y2 - y1) * (z2 - z1);
    sc = score[(size_t)b * K + t];
    if (cls) mycls = cls[(size_t)b * K + t];
    live = valid[(size_t)b * K + t] != 0;
    s_sc[t] = sc;
    s_alive[t] = live ? 1 : 0;
    keep[(size_t)b * K + t] = 0;
  }
  __syncthreads();
  if (t < K) {
    // position in the descending order among ALL boxes (invalid ones are skipped
    // through s_alive)
    int r = 0;
    for (int j = 0; j < K; ++j) {
      const double sj = s_sc[j];
      r += (sj > sc || (sj == sc && j > t)) ? 1 : 0;
    }
    s_order[r] = t;
  }
  __syncthreads();
  for (int r = 0; r < K; ++r) {
    const int i = s_order[r];
    if (!s_alive[i]) continue;            // block-uniform
    __syncthreads();                      // everyone has read s_alive[i]
    if (t == i) {
      keep[(size_t)b * K + i] = 1;
      s_alive[i] = 0;
    } else if (t < K && s_alive[t]) {
      const double ix1 = bx[i * 6], iy1 = bx[i * 6 + 1], iz1 = bx[i * 6 + 2];
      const double ix2 = bx[i * 6 + 3], iy2 = bx[i * 6 + 4], iz2 = bx[i * 6 + 5];
      const double iarea = (ix2 - ix1) * (iy2 - iy1) * (iz2 - iz1);
      const double l = fmax(0.0, fmin(ix2, x2) - fmax(ix1, x1));
      const double w = fmax(0.0, fmin(iy2, y2) - fmax(iy1, y1));
      const double h = fmax(0.0, fmin(iz2, z2) - fmax(iz1, z1));
      const double inter = l * w * h;
      double o;
      if (old_type) o = inter / area;
      else o = inter / (iarea + area - inter + (add_eps ? 1e-8 : 0.0));
      if (cls && cls[(size_t)b * K + i] != mycls) o = 0.0;
      if (o > thresh) s_alive[t] = 0;
    }
    __syncthreads();
  }
}

}  // namespace

static int chk4(const char *k) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c: %s launch failed: %s\n", k, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int s2c_boxes_count_points(int b, int n, int K, const float *pts,
                                      long long pt_stride, long long pt_batch_stride,
                                      const double *center, const double *size,
                                      const double *angle, int *counts, void *stream) {
  if (b <= 0 || n <= 0 || K <= 0 || !pts || !center || !size || !angle || !counts ||
      pt_stride < 3)
    return -1;
  hipStream_t st = (hipStream_t)stream;
  if (s2c::zero_async(counts, sizeof(int) * (size_t)b * K, st) != hipSuccess) return -1;
  hipLaunchKernelGGL(boxes_count_points_kernel, dim3((n + PIB_TILE - 1) / PIB_TILE, b),
                     dim3(256), 0, st, n, K, pts, pt_stride, pt_batch_stride, center, size,
                     angle, counts);
  return chk4("boxes_count_points");
}

extern "C" int s2c_nms(int b, int K, const double *boxes, const double *score,
                       const long long *cls, const unsigned char *valid, double thresh,
                       int old_type, int add_eps, unsigned char *keep, void *stream) {
  if (b <= 0 || K <= 0 || K > 1024 || !boxes || !score || !valid || !keep) return -1;
  hipLaunchKernelGGL(nms_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, K, boxes,
                     score, cls, valid, thresh, old_type, add_eps, keep);
  return chk4("nms");
}
