// s2c_decoder_persist.hip -- the teacher-forced decoder's forward recurrence as ONE kernel
// (models/caption_module.py:250-292 `_step`, T sequential steps; s2c_decoder.hip runs the same
// arithmetic as 5 dependent launches per step).
//
// Why: a step is five mat-vec stages, each needing the complete output of the one before
// (map_topdown+ReLU -> GRUCell 1 -> [map_hidd | map_lang's h block] -> attention + map_lang+ReLU
// -> GRUCell 2).  As launches that chain costs 5.8-7.7 us per stage whatever the kernel does (launch
// ramp + two dependent L2 round trips + drain); measured inside one kernel (tools/probe_sync.py) a
// grid-wide exchange of an 8 x 512 vector costs 2.2 us IF the data carries its own validity:
//   * every value travels as an 8-byte {bits, tag} pair written with ONE agent-scope store; the
//     consumers poll the DATA (16-byte sc0 sc1 loads, all of a lane's loads in flight, re-issued
//     only while a tag is stale).  No counter, no flag, no fence: a counter barrier over 256
//     workgroups measured 7.4 us, a flag array 17 us.
//   * tag = (launch nonce << 6) + step + 1; two buffers by step parity.  A buffer is rewritten at
//     step t + 2 only by a workgroup that has seen ALL of step t + 1, which every workgroup
//     publishes after it has finished reading step t -- no reader can be overtaken.
//   * the nonce lives in device memory (kernel arguments are frozen in a replayed hipGraph) and is
//     advanced by workgroup 0 once every workgroup has read it.
// 128 workgroups x 512 threads, one per CU (<= 256 VGPRs, ~100 KB LDS at the benchmark shapes): the
// geometry stream's FPS kernel holds 8 CUs for milliseconds at a time and a grid that needed all
// 256 CUs would wait for it -- spinning.  The host refuses the kernel unless the occupancy query
// says the whole grid is co-resident with CUs to spare.
//
// Work split (R <= 8 rows).  Wave = (unit pair up, row quad rq, part): part-0 lanes hold the
// h-operand (float4 slots lane, lane + 64 of rows 4 rq .. 4 rq + 3), part-1 lanes the x-operand of
// the GRU cells.  A workgroup owns ceil(H / 128) hidden units of both cells, two per wave (their
// 3 x (E + H) weight rows live in REGISTERS for the whole kernel: 96 VGPRs), ceil(E / 128) outputs
// of map_topdown and ceil((H + E) / 128) of the q / map_lang-h product (weights in LDS), and -- for
// the attention stage, which is row-local -- the (row w % 8, slice w / 8) block of map_lang's E
// outputs; every workgroup of a row redoes the row's K x H tanh scores (10 per thread), as
// attn_x2_kernel does.  The h-part of a GRU cell depends on the previous step only: the part-0
// waves form it while the part-1 waves are still polling the x operand.
#include "s2c_common.h"
#include "../../include/s2c_fused.h"

#include <stdio.h>
#include <stdlib.h>

using namespace s2c;

namespace {

constexpr int PG = 128;        // workgroups
constexpr int PT = 512;        // threads per workgroup
constexpr int PW = PT / 64;    // waves
constexpr int P_MAXK = 32;     // keys
constexpr int P_NSL = PG / 8;  // map_lang output slices per row
constexpr int P_MAXOC = 32;    // map_lang outputs per slice  (E <= 512)
constexpr int P_UB = 2;        // hidden units per WAVE (two unit pairs per workgroup, H <= 512)
constexpr int P_OB1 = 2;       // map_topdown outputs per wave (E <= 512)
constexpr int P_OB3 = 4;       // q / lang-h outputs per wave  (H + E <= 1024)
#define S2C_AG __HIP_MEMORY_SCOPE_AGENT
// cache policy of the polled loads: sc0 | sc1 (system scope: never served from a stale L2 / L1
// line) | bit 31 = volatile for the compiler (a poll must not be hoisted out of its loop)
constexpr int P_AUX = (int)0x80000011u;
// A poll that has not seen its data after this many passes (~1 us each) gives up: the workgroup
// raises a.fail, stops polling (garbage results, no hang) and the host reports the error.
constexpr int P_SPIN_MAX = 1 << 20;

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL,
                                                    0xF, 0xF, false));
}
// every lane of a 16-lane row receives the row's sum
__device__ __forceinline__ float row16_sum(float v) {
  v += dppf<DPP_QUAD_1032>(v);
  v += dppf<DPP_QUAD_2301>(v);
  v += dppf<DPP_ROW_HALF_MIRROR>(v);
  v += dppf<DPP_ROW_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float fdot4(const float4 a, const float4 b, float acc) {
  acc = __builtin_fmaf(a.x, b.x, acc);
  acc = __builtin_fmaf(a.y, b.y, acc);
  acc = __builtin_fmaf(a.z, b.z, acc);
  return __builtin_fmaf(a.w, b.w, acc);
}
__device__ __forceinline__ float p_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float p_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ void st_tag(u64 *p, float v, u32 tag) {
  __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, S2C_AG);
}

// offsets (in pairs) of the five exchanged vectors inside one parity buffer
struct XOff {
  int x1, h1, ql, x2, h2, total;
};
__device__ __host__ __forceinline__ XOff xoff(int H, int E) {
  XOff o;
  o.x1 = 0;
  o.h1 = 8 * E;
  o.ql = o.h1 + 8 * H;
  o.x2 = o.ql + 8 * (H + E);
  o.h2 = o.x2 + 8 * E;
  o.total = o.h2 + 8 * H;
  return o;
}

// Poll rows r0 .. r0+3 (those < R) of a tagged (8 x I) vector: this lane's float4 slots lane and
// lane + 64 (those < n4).  All sixteen 16-byte loads are in flight together; a row's base goes
// through the scalar offset, the slot through the instruction's immediate.
__device__ __forceinline__ void poll_rows(__amdgpu_buffer_rsrc_t rs, int base_pairs, int I, int n4,
                                          int r0, int R, u32 tag, float4 (&x)[4][2],
                                          volatile int *s_dead, int backoff) {
  const int lane = threadIdx.x & 63;
  const int voff = lane * 32;
  u32x4 raw[4][2][2];
  bool stale;
  int spins = 0;
  do {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int soff = __builtin_amdgcn_readfirstlane((base_pairs + (r0 + rr) * I) * 8);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if ((lane + 64 * s < n4) && (r0 + rr < R)) {
          raw[rr][s][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 2048 * s, soff, P_AUX);
          raw[rr][s][1] =
              __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 2048 * s + 16, soff, P_AUX);
        } else {
          raw[rr][s][0] = raw[rr][s][1] = (u32x4){0u, tag, 0u, tag};
        }
      }
    }
    stale = false;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        stale |= (raw[rr][s][0].y != tag) | (raw[rr][s][0].w != tag) | (raw[rr][s][1].y != tag) |
                 (raw[rr][s][1].w != tag);
    if (stale && (++spins > P_SPIN_MAX || *s_dead)) {
      *s_dead = 1;
      break;
    }
    if (stale)                                  // polling waves must not saturate the fabric
      for (int i = 0; i < backoff; ++i) __builtin_amdgcn_s_sleep(1);
  } while (stale);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int s = 0; s < 2; ++s)
      x[rr][s] = make_float4(__uint_as_float(raw[rr][s][0].x), __uint_as_float(raw[rr][s][0].z),
                             __uint_as_float(raw[rr][s][1].x), __uint_as_float(raw[rr][s][1].z));
}

// phase stamps (s_memtime, shader cycles) of workgroup 0's waves: prof[(wave * T + t) * 16 + slot]
#define P_STAMP(slot)                                                                  \
  do {                                                                                 \
    if (a.prof && w == 0 && lane == 0)                                                 \
      a.prof[((size_t)wv * T + t) * 16 + (slot)] = __builtin_readcyclecounter();       \
  } while (0)

// workgroup-local step flags: the polling wave of a row quad raises flag = step + 1 once the
// operand is in the LDS stash; the other waves of the quad wait on LDS instead of polling memory
__device__ __forceinline__ void flag_raise(volatile int *f, int v) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63) == 0) *f = v;
}
__device__ __forceinline__ void flag_wait(volatile int *f, int v, volatile int *s_dead) {
  int spins = 0;
  while (*f < v) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > P_SPIN_MAX || *s_dead) {
      *s_dead = 1;
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct GruW {
  float4 w[P_UB][3][2];
};

// this lane's slots of the GRU weight rows of the wave's units (part 0: W_hh, part 1: W_ih)
__device__ __forceinline__ void load_gru_w(GruW &g, const float *W, int I, int H, int u0) {
  const int lane = threadIdx.x & 63, n4 = I >> 2;
#pragma unroll
  for (int j = 0; j < P_UB; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int u = min(u0 + j, H - 1), q = lane + 64 * s;
        g.w[j][k][s] = q < n4 ? reinterpret_cast<const float4 *>(W + (size_t)(k * H + u) * I)[q]
                              : make_float4(0.f, 0.f, 0.f, 0.f);
      }
}

// partial gate sums of 4 rows x P_UB units x 3 gates -> the 16-lane-row totals in s_red[v][0..3]
__device__ __forceinline__ void gru_partials(const GruW &g, const float4 (&x)[4][2],
                                             float (*s_red)[4]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < P_UB; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        float a = fdot4(g.w[j][k][0], x[rr][0], 0.0f);
        a = fdot4(g.w[j][k][1], x[rr][1], a);
        a = row16_sum(a);
        if ((lane & 15) == 0) s_red[(j * 3 + k) * 4 + rr][lane >> 4] = a;
      }
}
// the same with the weight rows in LDS (sW: [unit j][gate k][I floats] of this wave's part / unit
// pair): GRU cell 2 -- both cells' rows in registers do not fit 256 VGPRs next to a poll in flight
__device__ __forceinline__ void gru_partials_lds(const float *sW, int I, const float4 (&x)[4][2],
                                                 float (*s_red)[4]) {
  const int lane = threadIdx.x & 63, n4 = I >> 2;
#pragma unroll
  for (int j = 0; j < P_UB; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 *wr = reinterpret_cast<const float4 *>(sW + (size_t)(j * 3 + k) * I);
      float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
      if (lane < n4) w0 = wr[lane];
      if (lane + 64 < n4) w1 = wr[lane + 64];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        float a = fdot4(w0, x[rr][0], 0.0f);
        a = fdot4(w1, x[rr][1], a);
        a = row16_sum(a);
        if ((lane & 15) == 0) s_red[(j * 3 + k) * 4 + rr][lane >> 4] = a;
      }
    }
}
__device__ __forceinline__ float red4(const float (*s_red)[4], int v) {
  return (s_red[v][0] + s_red[v][1]) + (s_red[v][2] + s_red[v][3]);
}

struct GruOut {
  float *h, *sr, *sz, *sn, *sghn;   // plain (R x H) destinations of this step
};

// GRUCell epilogue on the item lanes of a part-0 wave (lane = unit j * 4 + row rr); `bias` =
// this cell's [ih | hh][gate][unit of the workgroup] table in LDS, jw = the wave's first unit
// inside the workgroup
__device__ __forceinline__ void gru_epilogue(const float (*s_h)[4], const float (*s_x)[4], int H,
                                             int R, int r0, int u0, int jw, int ub,
                                             const float (*bias)[3][4], float &hp,
                                             const GruOut &o, u64 *xb, u32 tag) {
  const int lane = threadIdx.x & 63;
  const int j = lane >> 2, rr = lane & 3, row = r0 + rr, ju = jw + j, u = u0 + ju;
  if (lane < 4 * P_UB && ju < ub && row < R && u < H) {
    const int v = j * 12 + rr;
    const float gir = red4(s_x, v) + bias[0][0][ju], giz = red4(s_x, v + 4) + bias[0][1][ju],
                gin = red4(s_x, v + 8) + bias[0][2][ju];
    const float ghr = red4(s_h, v) + bias[1][0][ju], ghz = red4(s_h, v + 4) + bias[1][1][ju],
                ghn = red4(s_h, v + 8) + bias[1][2][ju];
    const float r = p_sigmoid(gir + ghr), z = p_sigmoid(giz + ghz);
    const float n = p_tanh(gin + r * ghn);
    const float hn = (1.0f - z) * n + z * hp;
    hp = hn;
    st_tag(xb + (size_t)row * H + u, hn, tag);
    const size_t e = (size_t)row * H + u;
    o.h[e] = hn; o.sr[e] = r; o.sz[e] = z; o.sn[e] = n; o.sghn[e] = ghn;
  }
}

// out[k2][rr] partials of `nk` LDS weight rows against the 4 x 2 operand slots of this lane
template <int NK>
__device__ __forceinline__ void lds_rows_partials(const float *sW, int H, int n4,
                                                  const float4 (&x)[4][2], float (*so)[4]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k2 = 0; k2 < NK; ++k2) {
    float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
    if (lane < n4) w0 = reinterpret_cast<const float4 *>(sW + (size_t)k2 * H)[lane];
    if (lane + 64 < n4) w1 = reinterpret_cast<const float4 *>(sW + (size_t)k2 * H)[lane + 64];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      float acc = fdot4(w0, x[rr][0], 0.0f);
      acc = fdot4(w1, x[rr][1], acc);
      acc = row16_sum(acc);
      if ((lane & 15) == 0) so[k2 * 4 + rr][lane >> 4] = acc;
    }
  }
}

__global__ __launch_bounds__(PT, 2) void decoder_fwd_persist_kernel(s2c_dec_fwd_args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float s_red[2][PW][24][4];    // [GRU cell][wave][value][16-lane row]
  __shared__ float s_one[2][PW][16][4];    // P1 / P3 partials (the same wave writes and reads)
  __shared__ float s_bias[2][2][3][4];     // [cell][ih | hh][gate][unit of the workgroup]
  __shared__ float s_sc[P_MAXK][PT / 16], s_s[P_MAXK], s_add[P_MAXOC];
  static_assert(PT / 16 == 32, "score partials: one 32-lane group per key");
  __shared__ __attribute__((aligned(16))) float s_att[256];
  __shared__ u32 s_nonce;
  __shared__ int s_dead;
  __shared__ int s_flag[3][2];             // [h2 of P1 | h1 of P3 | x1 published][row quad]
  const int R = a.R, K = a.K, H = a.H, E = a.E, F = a.F, T = a.T;
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int part = wv & 1, rq = (wv >> 1) & 1, r0 = 4 * rq, up = wv >> 2;
  const int n4h = H >> 2, n4e = E >> 2;
  const int ub = (H + PG - 1) / PG, ob1 = (E + PG - 1) / PG, ob3 = (H + E + PG - 1) / PG;
  const int u0 = w * ub;                       // first hidden unit of the workgroup
  const int row4 = w & 7, slice = w >> 3;
  const int oc4 = (E + P_NSL - 1) / P_NSL;
  // dynamic LDS carve-up (floats)
  float *sW1 = smem;                           // 2 P_OB1 x H  map_topdown's h2 block
  float *sW3 = sW1 + 2 * P_OB1 * H;            // 2 P_OB3 x H  [map_hidd ; map_lang's h block]
  float *sH1 = sW3 + 2 * P_OB3 * H;            // 8 x H        h1 of the current step
  float *sH2 = sH1 + 8 * H;                    // 8 x H        h2
  float *sM = sH2 + 8 * H;                     // K x H        map_feat(obj_feats) of row4
  float *sO = sM + K * H;                      // K x F        obj_feats of row4
  float *sWl = sO + K * F;                     // P_MAXOC x F  map_lang's attended-feature block
  float *sG2 = sWl + P_MAXOC * F;              // GRU cell 2: [unit pair][hh: 6 x H | ih: 6 x E]
  const float *sG2w = sG2 + (size_t)up * 6 * (H + E) + (part ? 6 * H : 0);
  // ---- launch nonce -------------------------------------------------------------------
  if (tid == 0) {
    s_dead = 0;
    s_flag[0][0] = s_flag[0][1] = s_flag[1][0] = s_flag[1][1] = s_flag[2][0] = s_flag[2][1] = 0;
    s_nonce = __hip_atomic_load(a.nonce, __ATOMIC_RELAXED, S2C_AG);
    __hip_atomic_fetch_add(a.started, 1u, __ATOMIC_RELAXED, S2C_AG);
  }
  // ---- resident operands ----------------------------------------------------------------
  GruW g1;
  load_gru_w(g1, part ? a.W_ih1 : a.W_hh1, part ? E : H, H, u0 + P_UB * up);
  for (int c = 0; c < 4; ++c) {                // (unit pair, part) blocks of cell 2
    const int cup = c >> 1, cpart = c & 1, I = cpart ? E : H, n4 = I >> 2;
    const float *W = cpart ? a.W_ih2 : a.W_hh2;
    float *dst = sG2 + (size_t)cup * 6 * (H + E) + (cpart ? 6 * H : 0);
    for (int i = tid; i < 6 * n4; i += PT) {
      const int jk = i / n4, q = i - jk * n4, j = jk / 3, k = jk - 3 * j;
      const int u = min(u0 + P_UB * cup + j, H - 1);
      reinterpret_cast<float4 *>(dst)[i] =
          reinterpret_cast<const float4 *>(W + (size_t)(k * H + u) * I)[q];
    }
  }
  for (int i = tid; i < 2 * P_OB1 * n4h; i += PT) {
    const int k = i / n4h, q = i - k * n4h, o = min(w * ob1 + k, E - 1);
    reinterpret_cast<float4 *>(sW1)[i] =
        reinterpret_cast<const float4 *>(a.W_td_h2 + (size_t)o * a.ldtd)[q];
  }
  for (int i = tid; i < 2 * P_OB3 * n4h; i += PT) {
    const int k = i / n4h, q = i - k * n4h, o = min(w * ob3 + k, H + E - 1);
    reinterpret_cast<float4 *>(sW3)[i] = reinterpret_cast<const float4 *>(a.Wqh + (size_t)o * H)[q];
  }
  if (row4 < R) {
    for (int i = tid; i < K * n4h; i += PT)
      reinterpret_cast<float4 *>(sM)[i] =
          reinterpret_cast<const float4 *>(a.M + (size_t)row4 * K * H)[i];
    for (int i = tid; i < K * (F >> 2); i += PT)
      reinterpret_cast<float4 *>(sO)[i] =
          reinterpret_cast<const float4 *>(a.O + (size_t)row4 * K * F)[i];
    for (int i = tid; i < oc4 * (F >> 2); i += PT) {
      const int ol = i / (F >> 2), q = i - ol * (F >> 2), o = min(slice * oc4 + ol, E - 1);
      reinterpret_cast<float4 *>(sWl)[i] =
          reinterpret_cast<const float4 *>(a.W_lang + (size_t)o * a.ldlang)[q];
    }
  }
  if (tid < 48) {
    const int c = tid / 24, ih = (tid / 12) & 1, k = (tid / 4) % 3, ju = tid & 3;
    const int u = min(u0 + ju, H - 1);
    const float *b = c == 0 ? (ih == 0 ? a.b_ih1 : a.b_hh1) : (ih == 0 ? a.b_ih2 : a.b_hh2);
    s_bias[c][ih][k][ju] = b[k * H + u];
  }
  // attention constant of this thread: hidden unit tid
  const float wa0 = tid < H ? a.wa[tid] : 0.0f;
  float hp1 = 0.f, hp2 = 0.f;                  // h of the item lanes' (row, unit)
  // h1, h2 of step 0 are zero
  for (int i = tid; i < 2 * 8 * n4h; i += PT)
    reinterpret_cast<float4 *>(sH1)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const u32 base = (s_nonce << 6) + 1u;
  const XOff xo = xoff(H, E);
  __amdgpu_buffer_rsrc_t rs[2];
  rs[0] = __builtin_amdgcn_make_buffer_rsrc((void *)a.xbuf, 0, xo.total * 8, 0x00020000);
  rs[1] = __builtin_amdgcn_make_buffer_rsrc((void *)(a.xbuf + xo.total), 0, xo.total * 8,
                                            0x00020000);
  const size_t RH = (size_t)R * H, RE = (size_t)R * E;

  for (int t = 0; t < T; ++t) {
    const int par = t & 1;
    const u32 tag = base + (u32)t;
    u64 *xb = a.xbuf + (size_t)par * xo.total;
    // ================= P1: x1 = relu(W_td[:, h2 block] h2 + Pw[:, t] + Ptf) ==================
    P_STAMP(0);
    if (part == 0) {
      const int k = lane >> 2, rr = lane & 3, row = r0 + rr, ko = P_OB1 * up + k,
                o = w * ob1 + ko;
      const bool item = lane < 4 * P_OB1 && ko < ob1 && row < R && o < E;
      float e_add = 0.0f;
      if (item) e_add = a.Pw[((size_t)row * T + t) * E + o] + a.Ptf[(size_t)row * E + o];
      float4 x[4][2];
      if (t > 0 && up == 0) {
        poll_rows(rs[par ^ 1], xo.h2, H, n4h, r0, R, tag - 1u, x, &s_dead, a.backoff);
        P_STAMP(1);
#pragma unroll
        for (int rr2 = 0; rr2 < 4; ++rr2)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            if (lane + 64 * s < n4h)
              reinterpret_cast<float4 *>(sH2 + (size_t)(r0 + rr2) * H)[lane + 64 * s] = x[rr2][s];
        flag_raise(&s_flag[0][rq], t + 1);
      } else if (t > 0) {
        flag_wait(&s_flag[0][rq], t + 1, &s_dead);
#pragma unroll
        for (int rr2 = 0; rr2 < 4; ++rr2)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            x[rr2][s] = lane + 64 * s < n4h
                            ? reinterpret_cast<const float4 *>(sH2 + (size_t)(r0 + rr2) * H)[lane + 64 * s]
                            : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
#pragma unroll
        for (int rr2 = 0; rr2 < 4; ++rr2)
#pragma unroll
          for (int s = 0; s < 2; ++s) x[rr2][s] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float (*so)[4] = s_one[0][wv];
      lds_rows_partials<P_OB1>(sW1 + (size_t)P_OB1 * up * H, H, n4h, x, so);
      __builtin_amdgcn_wave_barrier();
      if (item) {
        const float v = fmaxf(red4(so, k * 4 + rr) + e_add, 0.0f);
        st_tag(xb + xo.x1 + (size_t)row * E + o, v, tag);
        a.X1[(size_t)t * RE + (size_t)row * E + o] = v;
      }
      if (up == 1) flag_raise(&s_flag[2][rq], t + 1);
      P_STAMP(2);
    }
    // ================= P2: GRUCell 1 =====================================================
    {
      float4 x[4][2];
      if (part == 0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            x[rr][s] = lane + 64 * s < n4h
                           ? reinterpret_cast<const float4 *>(sH1 + (size_t)(r0 + rr) * H)[lane + 64 * s]
                           : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        // x1 cannot exist before this workgroup's own share has been published (all workgroups
        // run in step): do not load the fabric with polls until then
        flag_wait(&s_flag[2][rq], t + 1, &s_dead);
        poll_rows(rs[par], xo.x1, E, n4e, r0, R, tag, x, &s_dead, a.backoff);
      }
      P_STAMP(3);
      gru_partials(g1, x, s_red[0][wv]);
      P_STAMP(4);
      __syncthreads();
      P_STAMP(5);
      if (part == 0) {
        GruOut o = {a.H1 + (size_t)(t + 1) * RH, a.S1[0] + (size_t)t * RH, a.S1[1] + (size_t)t * RH,
                    a.S1[2] + (size_t)t * RH, a.S1[3] + (size_t)t * RH};
        gru_epilogue(s_red[0][wv], s_red[0][wv + 1], H, R, r0, u0, P_UB * up, ub, s_bias[0], hp1,
                     o, xb + xo.h1, tag);
      }
      P_STAMP(6);
    }
    // ================= P3: [q | lang-h] = [W_h ; W_lang[:, F:]] h1 ===========================
    if (part == 0) {
      float4 x[4][2];
      if (up == 0) {
        poll_rows(rs[par], xo.h1, H, n4h, r0, R, tag, x, &s_dead, a.backoff);
        P_STAMP(7);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            if (lane + 64 * s < n4h)
              reinterpret_cast<float4 *>(sH1 + (size_t)(r0 + rr) * H)[lane + 64 * s] = x[rr][s];
        flag_raise(&s_flag[1][rq], t + 1);
      } else {
        flag_wait(&s_flag[1][rq], t + 1, &s_dead);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            x[rr][s] = lane + 64 * s < n4h
                           ? reinterpret_cast<const float4 *>(sH1 + (size_t)(r0 + rr) * H)[lane + 64 * s]
                           : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float (*so)[4] = s_one[1][wv];
      lds_rows_partials<P_OB3>(sW3 + (size_t)P_OB3 * up * H, H, n4h, x, so);
      __builtin_amdgcn_wave_barrier();
      const int k = lane >> 2, rr = lane & 3, row = r0 + rr, ko = P_OB3 * up + k,
                o = w * ob3 + ko;
      if (lane < 4 * P_OB3 && ko < ob3 && row < R && o < H + E) {
        const float v = red4(so, k * 4 + rr);
        st_tag(xb + xo.ql + (size_t)row * (H + E) + o, v, tag);
        a.QL[(size_t)t * R * (H + E) + (size_t)row * (H + E) + o] = v;
      }
      P_STAMP(8);
    }
    // ================= P4: attention of row4 + this slice of map_lang =======================
    if (row4 < R) {
      // q[row4, tid] and the slice's lang-h addends: one tagged value each
      const int o_add = slice * oc4 + tid;
      const bool has_q = tid < H, has_add = tid < oc4 && o_add < E;
      const u64 *pq = xb + xo.ql + (size_t)row4 * (H + E) + (has_q ? tid : 0);
      const u64 *pa = xb + xo.ql + (size_t)row4 * (H + E) + H + (has_add ? o_add : 0);
      u64 kq = (u64)tag << 32, ka = (u64)tag << 32;
      bool stale;
      int spins = 0;
      do {
        if (has_q) kq = __hip_atomic_load(pq, __ATOMIC_RELAXED, S2C_AG);
        if (has_add) ka = __hip_atomic_load(pa, __ATOMIC_RELAXED, S2C_AG);
        stale = ((u32)(kq >> 32) != tag) | ((u32)(ka >> 32) != tag);
        if (stale && (++spins > P_SPIN_MAX || *(volatile int *)&s_dead)) {
          *(volatile int *)&s_dead = 1;
          break;
        }
      } while (stale);
      P_STAMP(9);
      const float q0 = __uint_as_float((u32)kq);
      if (tid < oc4) s_add[tid] = __uint_as_float((u32)ka);
      for (int k = 0; k < K; ++k) {
        float p = 0.0f;
        if (has_q) p = wa0 * p_tanh(sM[(size_t)k * H + tid] + q0);
        p = row16_sum(p);
        if ((lane & 15) == 0) s_sc[k][tid >> 4] = p;
      }
      __syncthreads();
      for (int i = tid; i < 32 * K; i += PT) {   // 32 row partials per key: one more DPP pass
        const int k = i >> 5;
        float v = row16_sum(s_sc[k][i & 31]);
        v += __shfl_xor(v, 16, 64);
        if ((i & 31) == 0) s_s[k] = a.mask[(size_t)row4 * K + k] == 0.0f ? -1e30f : v;
      }
      __syncthreads();
      {
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, s_s[k]);
        // one exponential per key and thread: the weighted sum and the normaliser in one pass
        const int f = tid < F ? tid : 0;
        float sum = 0.0f, acc = 0.0f;
        for (int k = 0; k < K; ++k) {
          const float e = __expf(s_s[k] - mx);
          sum += e;
          acc = __builtin_fmaf(e, sO[(size_t)k * F + f], acc);
        }
        const float inv = 1.0f / sum;
        if (slice == 0 && tid < K)
          a.ALPHA[(size_t)t * R * K + (size_t)row4 * K + tid] = __expf(s_s[tid] - mx) * inv;
        if (tid < F) {
          s_att[tid] = acc * inv;
          if (slice == 0) a.ATT[(size_t)t * R * F + (size_t)row4 * F + tid] = acc * inv;
        }
      }
      __syncthreads();
      {
        const int ol = tid >> 4, pr = tid & 15, o = slice * oc4 + ol;
        float acc = 0.0f;
        if (ol < oc4)
          for (int f4 = pr; f4 < (F >> 2); f4 += 16)
            acc = fdot4(reinterpret_cast<const float4 *>(sWl + (size_t)ol * F)[f4],
                        reinterpret_cast<const float4 *>(s_att)[f4], acc);
        acc = row16_sum(acc);
        if (pr == 0 && ol < oc4 && o < E) {
          const float v = fmaxf(acc + a.b_lang[o] + s_add[ol], 0.0f);
          st_tag(xb + xo.x2 + (size_t)row4 * E + o, v, tag);
          a.X2[(size_t)t * RE + (size_t)row4 * E + o] = v;
        }
      }
      P_STAMP(10);
    }
    // ================= P5: GRUCell 2 =====================================================
    {
      float4 x[4][2];
      if (part == 0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            x[rr][s] = lane + 64 * s < n4h
                           ? reinterpret_cast<const float4 *>(sH2 + (size_t)(r0 + rr) * H)[lane + 64 * s]
                           : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        poll_rows(rs[par], xo.x2, E, n4e, r0, R, tag, x, &s_dead, a.backoff);
      }
      P_STAMP(11);
      gru_partials_lds(sG2w, part ? E : H, x, s_red[1][wv]);
      __syncthreads();
      if (part == 0) {
        GruOut o = {a.H2 + (size_t)(t + 1) * RH, a.S2[0] + (size_t)t * RH, a.S2[1] + (size_t)t * RH,
                    a.S2[2] + (size_t)t * RH, a.S2[3] + (size_t)t * RH};
        gru_epilogue(s_red[1][wv], s_red[1][wv + 1], H, R, r0, u0, P_UB * up, ub, s_bias[1], hp2,
                     o, xb + xo.h2, tag);
      }
      P_STAMP(12);
    }
  }
  __syncthreads();
  if (tid == 0 && s_dead) __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, S2C_AG);
  // ---- advance the nonce once every workgroup has read it -----------------------------------
  if (w == 0 && tid == 0) {
    int spins = 0;
    while (__hip_atomic_load(a.started, __ATOMIC_RELAXED, S2C_AG) < (u32)PG && ++spins < P_SPIN_MAX) {}
    __hip_atomic_store(a.started, 0u, __ATOMIC_RELAXED, S2C_AG);
    __hip_atomic_store(a.nonce, s_nonce + 1u, __ATOMIC_RELAXED, S2C_AG);
  }
}

size_t persist_lds_bytes(int K, int H, int E, int F) {
  return sizeof(float) * ((size_t)(2 * P_OB1 + 2 * P_OB3 + 16) * H + (size_t)K * H + (size_t)K * F +
                          (size_t)P_MAXOC * F + (size_t)12 * (H + E));
}

int g_persist = -1;

}  // namespace

extern "C" void s2c_decoder_persist_set(int on) { g_persist = on; }

extern "C" long long s2c_decoder_fwd_persist_xbuf_pairs(int H, int E) {
  return 2LL * xoff(H, E).total;
}

// 1 = the persistent kernel can take these shapes on the current device (whole grid co-resident)
extern "C" int s2c_decoder_fwd_persist_supported(int R, int K, int H, int E, int F, int T) {
  if (g_persist < 0) {
    const char *e = getenv("S2C_DECODER_PERSIST");
    g_persist = e ? atoi(e) : 1;
  }
  if (!g_persist) return 0;
  if (R < 1 || R > 8 || K < 1 || K > P_MAXK || T < 1 || T > 62) return 0;
  if (H % 4 || E % 4 || F % 4 || H < 4 || H > 512 || E < 4 || E > 512 || F < 4 || F > 256) return 0;
  const size_t lds = persist_lds_bytes(K, H, E, F);
  static int state[64];           // per device: 0 unknown, 1 ok, -1 refused
  static size_t lds_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (state[dev] == 0 || lds > lds_set[dev]) {
    state[dev] = -1;
    if (hipFuncSetAttribute((const void *)decoder_fwd_persist_kernel,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    lds_set[dev] = lds;
    state[dev] = 1;
  }
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)decoder_fwd_persist_kernel,
                                                   PT, lds) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  // every workgroup must be resident at once, with a margin of CUs for kernels of other streams
  return (long long)per_cu * (cus - 16) >= PG ? 1 : 0;
}

extern "C" int s2c_decoder_fwd_persist(const s2c_dec_fwd_args *a, void *stream) {
  if (!a || !s2c_decoder_fwd_persist_supported(a->R, a->K, a->H, a->E, a->F, a->T)) return -2;
  if (a->ldtd % 4 || a->ldlang % 4) return -2;
  const size_t lds = persist_lds_bytes(a->K, a->H, a->E, a->F);
  hipLaunchKernelGGL(decoder_fwd_persist_kernel, dim3(PG), dim3(PT), lds, (hipStream_t)stream, *a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_decoder_fwd_persist launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
