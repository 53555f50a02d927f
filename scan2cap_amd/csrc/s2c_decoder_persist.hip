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

// old = 0 with bound_ctrl: lets the DPP combiner fold the move into the add (v_add_f32_dpp); every
// control used here has a valid source lane for every lane
template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
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
template <int NR>
__device__ __forceinline__ void poll_rows(__amdgpu_buffer_rsrc_t rs, int base_pairs, int I, int n4,
                                          int r0, int R, u32 tag, float4 (&x)[NR][2],
                                          volatile int *s_dead, int backoff) {
  const int lane = threadIdx.x & 63;
  const int voff = lane * 32;
  u32x4 raw[NR][2][2];
#pragma unroll
  for (int rr = 0; rr < NR; ++rr)
#pragma unroll
    for (int s = 0; s < 2; ++s)              // slots this lane does not own: valid from the start
      raw[rr][s][0] = raw[rr][s][1] = (u32x4){0u, tag, 0u, tag};
  bool stale;
  int spins = 0;
  do {
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      const int soff = __builtin_amdgcn_readfirstlane((base_pairs + (r0 + rr) * I) * 8);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if ((lane + 64 * s < n4) && (r0 + rr < R)) {
          raw[rr][s][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 2048 * s, soff, P_AUX);
          raw[rr][s][1] =
              __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 2048 * s + 16, soff, P_AUX);
        }
      }
    }
    // a slot's tags only ever grow, so a stale tag is SMALLER than `tag` and cannot contain all of
    // its bits: the AND of all tags equals `tag` iff every one of them does (one compare per pass)
    u32 m = tag;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        m &= (raw[rr][s][0].y & raw[rr][s][0].w) & (raw[rr][s][1].y & raw[rr][s][1].w);
    stale = m != tag;
    if (stale && (++spins > P_SPIN_MAX || *s_dead)) {
      *s_dead = 1;
      break;
    }
    if (stale)                                  // polling waves must not saturate the fabric
      for (int i = 0; i < backoff; ++i) __builtin_amdgcn_s_sleep(1);
  } while (stale);
#pragma unroll
  for (int rr = 0; rr < NR; ++rr)
#pragma unroll
    for (int s = 0; s < 2; ++s)
      x[rr][s] = make_float4(__uint_as_float(raw[rr][s][0].x), __uint_as_float(raw[rr][s][0].z),
                             __uint_as_float(raw[rr][s][1].x), __uint_as_float(raw[rr][s][1].z));
}

// phase stamps (s_memtime, shader cycles) of workgroup 0's waves: prof[(wave * T + t) * 16 + slot]
#define P_STAMP(slot)                                                                  \
  do {                                                                                 \
    if (PROF) {                                                                        \
      if (a.prof && w == 0 && lane == 0)                                               \
        a.prof[((size_t)wv * T + t) * 16 + (slot)] = __builtin_readcyclecounter();     \
    } else {                                                                           \
      /* measured: without the stamps the scheduler moves code across the phases and the   \
         forward kernel loses 6 % (636 -> 674 us); keep the phase boundaries */            \
      __builtin_amdgcn_sched_barrier(0);                                               \
    }                                                                                  \
  } while (0)

// Workgroup barrier of the step loops.  Measured twice: an LDS-only variant (s_waitcnt lgkmcnt(0) +
// s_barrier, i.e. without __syncthreads()'s wait for the wave's outstanding global stores) is SLOWER
// (fwd 565 -> 588 us, bwd 642 -> 649 us), although the same relaxation of the LDS flags below gained
// (fwd 580 -> 565 us): what a wave waits for at these barriers is the other row quad, which runs
// ~1800 cycles behind through every exchange, not its own stores.
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }

// workgroup-local step flags: the polling wave of a row quad raises flag = step + 1 once the
// operand is in the LDS stash; the other waves of the quad wait on LDS instead of polling memory
// (LDS is one in-order memory: the stash writes of this wave precede its flag write; a
// workgroup-scope release fence would also wait for the wave's outstanding GLOBAL stores)
__device__ __forceinline__ void flag_raise(volatile int *f, int v) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) *f = v;
}
// NAP: sleep between looks -- for the long wait of the x-part waves (thousands of cycles, next to
// h-part waves that are computing on the same SIMDs); the operand flags are waited for hot
template <bool NAP = false>
__device__ __forceinline__ void flag_wait(volatile int *f, int v, volatile int *s_dead) {
  int spins = 0;
  while (*f < v) {
    if (NAP) __builtin_amdgcn_s_sleep(2);
    if (++spins > P_SPIN_MAX || *s_dead) {
      *s_dead = 1;
      break;
    }
  }
  asm volatile("" ::: "memory");
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

// partial gate sums of 4 rows x P_UB units x 3 gates -> the 16-lane-row totals in s_red[v][0..3].
// Per unit: all dot products first, then the DPP folds (twelve independent chains the scheduler
// can interleave), then the stores under ONE exec mask.
__device__ __forceinline__ void gru_partials(const GruW &g, const float4 (&x)[4][2],
                                             float (*s_red)[4]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < P_UB; ++j) {
    float acc[12];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        acc[k * 4 + rr] = fdot4(g.w[j][k][1], x[rr][1], fdot4(g.w[j][k][0], x[rr][0], 0.0f));
#pragma unroll
    for (int v = 0; v < 12; ++v) acc[v] = row16_sum(acc[v]);
    if ((lane & 15) == 0) {
#pragma unroll
      for (int v = 0; v < 12; ++v) s_red[j * 12 + v][lane >> 4] = acc[v];
    }
  }
}
// the same with the weight rows in LDS (sW: [unit j][gate k][I floats] of this wave's part / unit
// pair): GRU cell 2 -- both cells' rows in registers do not fit 256 VGPRs next to a poll in flight
__device__ __forceinline__ void gru_partials_lds(const float *sW, int I, const float4 (&x)[4][2],
                                                 float (*s_red)[4]) {
  const int lane = threadIdx.x & 63, n4 = I >> 2;
#pragma unroll
  for (int j = 0; j < P_UB; ++j) {
    float acc[12];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 *wr = reinterpret_cast<const float4 *>(sW + (size_t)(j * 3 + k) * I);
      float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
      if (lane < n4) w0 = wr[lane];
      if (lane + 64 < n4) w1 = wr[lane + 64];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[k * 4 + rr] = fdot4(w1, x[rr][1], fdot4(w0, x[rr][0], 0.0f));
    }
#pragma unroll
    for (int v = 0; v < 12; ++v) acc[v] = row16_sum(acc[v]);
    if ((lane & 15) == 0) {
#pragma unroll
      for (int v = 0; v < 12; ++v) s_red[j * 12 + v][lane >> 4] = acc[v];
    }
  }
}
__device__ __forceinline__ float red4(const float (*s_red)[4], int v) {
  const float4 q = *reinterpret_cast<const float4 *>(s_red[v]);    // one 16-byte LDS read
  return (q.x + q.y) + (q.z + q.w);
}

struct GruOut {
  float *h, *sr, *sz, *sn, *sghn;   // plain (R x H) destinations of this step
  float *c[4];                      // NULL or: gate-gradient coefficients (see s2c_dec_bwd_args)
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
    st_tag(xb + (size_t)row * H + u, hn, tag);
    const size_t e = (size_t)row * H + u;
    o.h[e] = hn; o.sr[e] = r; o.sz[e] = z; o.sn[e] = n; o.sghn[e] = ghn;
    if (o.c[0]) {
      // d(pre-activations) = dh' * c:  dgi = dh' [cr | cz | cn],  dgh = dh' [cr | cz | cnr]
      const float cn = (1.0f - z) * (1.0f - n * n);
      o.c[0][e] = cn * ghn * (r * (1.0f - r));
      o.c[1][e] = (hp - n) * (z * (1.0f - z));
      o.c[2][e] = cn;
      o.c[3][e] = cn * r;
    }
    hp = hn;
  }
}

// out[k2][rr] partials of `nk` LDS weight rows against the 4 x 2 operand slots of this lane
template <int NK>
__device__ __forceinline__ void lds_rows_partials(const float *sW, int H, int n4,
                                                  const float4 (&x)[4][2], float (*so)[4]) {
  const int lane = threadIdx.x & 63;
  float acc[NK * 4];
#pragma unroll
  for (int k2 = 0; k2 < NK; ++k2) {
    float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
    if (lane < n4) w0 = reinterpret_cast<const float4 *>(sW + (size_t)k2 * H)[lane];
    if (lane + 64 < n4) w1 = reinterpret_cast<const float4 *>(sW + (size_t)k2 * H)[lane + 64];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[k2 * 4 + rr] = fdot4(w1, x[rr][1], fdot4(w0, x[rr][0], 0.0f));
  }
#pragma unroll
  for (int v = 0; v < NK * 4; ++v) acc[v] = row16_sum(acc[v]);
  if ((lane & 15) == 0) {
#pragma unroll
    for (int v = 0; v < NK * 4; ++v) so[v][lane >> 4] = acc[v];
  }
}

// Two grids.  <128, 512> (default): two unit pairs per workgroup, the second pair's waves read the
// polled operands from the LDS stash of the first.  <256, 256>: at most 256 VGPRs and <= 80 KB of
// LDS, so two fit one CU when another stream's kernel holds some; its h-part waves re-poll h1 /
// h2 instead of keeping a stash.  Taken when the first grid's LDS request does not fit.
template <int PGc, int PTc, bool PROF>
__global__ __launch_bounds__(PTc, 2) void decoder_fwd_persist_kernel(s2c_dec_fwd_args a) {
  constexpr int PG = PGc, PT = PTc, PW = PT / 64, UPS = PT / 256, P_NSL = PG / 8,
                P_MAXOC = 512 / P_NSL, NH = 512 / PT, NP = PT / 16;
  constexpr bool STASH = UPS == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ __attribute__((aligned(16))) float s_red[2][PW][24][4];   // [GRU cell][wave][value][16-lane row]
  __shared__ __attribute__((aligned(16))) float s_one[2][PW][16][4];    // P1 / P3 partials (the same wave writes and reads)
  __shared__ float s_bias[2][2][3][4];     // [cell][ih | hh][gate][unit of the workgroup]
  __shared__ float s_sc[P_MAXK][PT / 16], s_s[P_MAXK], s_add[P_MAXOC], s_mask[P_MAXK];
  __shared__ __attribute__((aligned(16))) float s_att[256];
  __shared__ u32 s_nonce;
  __shared__ int s_dead;
  __shared__ int s_flag[3][2];             // [h2 of P1 | h1 of P3 | x1 published][row quad]
  const int R = a.R, K = a.K, H = a.H, E = a.E, F = a.F, T = a.T;
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // waves wv and wv + 4 share a SIMD: give them DIFFERENT parts -- the h-part of a GRU cell is
  // formed while the x-part polls, and P1 / P3 are h-part work only, so the two never compete for
  // issue slots outside the attention stage
  const int part = wv / (PW / 2), rq = wv & 1, r0 = 4 * rq, up = (wv >> 1) & (UPS - 1);
  const int n4h = H >> 2, n4e = E >> 2;
  const int ub = (H + PG - 1) / PG, ob1 = (E + PG - 1) / PG, ob3 = (H + E + PG - 1) / PG;
  const int u0 = w * ub;                       // first hidden unit of the workgroup
  const int row4 = w & 7, slice = w >> 3;
  const int oc4 = (E + P_NSL - 1) / P_NSL;
  // dynamic LDS carve-up (floats)
  float *sW1 = smem;                           // 2 P_OB1 x H  map_topdown's h2 block
  float *sW3 = sW1 + UPS * P_OB1 * H;          // UPS P_OB3 x H  [map_hidd ; map_lang's h block]
  float *sH1 = sW3 + UPS * P_OB3 * H;          // 8 x H        h1 of the current step (STASH)
  float *sH2 = sH1 + (STASH ? 8 * H : 0);      // 8 x H        h2                     (STASH)
  float *sM = sH2 + (STASH ? 8 * H : 0);       // K x H        map_feat(obj_feats) of row4
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
  for (int c = 0; c < 2 * UPS; ++c) {          // (unit pair, part) blocks of cell 2
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
  for (int i = tid; i < UPS * P_OB1 * n4h; i += PT) {
    const int k = i / n4h, q = i - k * n4h, o = min(w * ob1 + k, E - 1);
    reinterpret_cast<float4 *>(sW1)[i] =
        reinterpret_cast<const float4 *>(a.W_td_h2 + (size_t)o * a.ldtd)[q];
  }
  for (int i = tid; i < UPS * P_OB3 * n4h; i += PT) {
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
  // attention constants of this thread: hidden unit tid; the bias of its map_lang output
  float wa_[NH];
#pragma unroll
  for (int m = 0; m < NH; ++m) wa_[m] = tid + PT * m < H ? a.wa[tid + PT * m] : 0.0f;
  float bl = 0.0f;
  if (row4 < R) {
    const int o = slice * oc4 + (tid >> 4);
    if ((tid >> 4) < oc4 && o < E) bl = a.b_lang[o];
    if (tid < K) s_mask[tid] = a.mask[(size_t)row4 * K + tid];
  }
  float hp1 = 0.f, hp2 = 0.f;                  // h of the item lanes' (row, unit)
  // h1, h2 of step 0 are zero
  if (STASH)
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
        poll_rows<4>(rs[par ^ 1], xo.h2, H, n4h, r0, R, tag - 1u, x, &s_dead, a.backoff);
        P_STAMP(1);
        if (STASH) {
#pragma unroll
          for (int rr2 = 0; rr2 < 4; ++rr2)
#pragma unroll
            for (int s = 0; s < 2; ++s)
              if (lane + 64 * s < n4h)
                reinterpret_cast<float4 *>(sH2 + (size_t)(r0 + rr2) * H)[lane + 64 * s] = x[rr2][s];
          flag_raise(&s_flag[0][rq], t + 1);
        }
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
      if (up == UPS - 1) flag_raise(&s_flag[2][rq], t + 1);
      P_STAMP(2);
    }
    // ================= P2: GRUCell 1 =====================================================
    {
      float4 x[4][2];
      if (part == 0) {
        if (STASH) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int s = 0; s < 2; ++s)
              x[rr][s] = lane + 64 * s < n4h
                             ? reinterpret_cast<const float4 *>(sH1 + (size_t)(r0 + rr) * H)[lane + 64 * s]
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (t > 0) {                      // h1 of the previous step: complete, one pass
          poll_rows<4>(rs[par ^ 1], xo.h1, H, n4h, r0, R, tag - 1u, x, &s_dead, a.backoff);
        } else {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int s = 0; s < 2; ++s) x[rr][s] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
        // x1 cannot exist before this workgroup's own share has been published (all workgroups
        // run in step): do not load the fabric with polls until then
        flag_wait<true>(&s_flag[2][rq], t + 1, &s_dead);
        poll_rows<4>(rs[par], xo.x1, E, n4e, r0, R, tag, x, &s_dead, a.backoff);
      }
      P_STAMP(3);
      gru_partials(g1, x, s_red[0][wv]);
      P_STAMP(4);
      lds_barrier();
      P_STAMP(5);
      if (part == 0) {
        float *S = a.S + (size_t)t * RH, *C = a.C ? a.C + (size_t)t * RH : nullptr;
        const size_t TRH = (size_t)T * RH;
        GruOut o = {a.H1 + (size_t)(t + 1) * RH, S, S + TRH, S + 2 * TRH, S + 3 * TRH,
                    {C, C + TRH, C + 2 * TRH, C + 3 * TRH}};
        gru_epilogue(s_red[0][wv], s_red[0][wv + PW / 2], H, R, r0, u0, P_UB * up, ub, s_bias[0], hp1,
                     o, xb + xo.h1, tag);
      }
      P_STAMP(6);
    }
    // ================= P3: [q | lang-h] = [W_h ; W_lang[:, F:]] h1 ===========================
    if (part == 0) {
      float4 x[4][2];
      if (up == 0) {
        poll_rows<4>(rs[par], xo.h1, H, n4h, r0, R, tag, x, &s_dead, a.backoff);
        P_STAMP(7);
        if (STASH) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int s = 0; s < 2; ++s)
              if (lane + 64 * s < n4h)
                reinterpret_cast<float4 *>(sH1 + (size_t)(r0 + rr) * H)[lane + 64 * s] = x[rr][s];
          flag_raise(&s_flag[1][rq], t + 1);
        }
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
      // q[row4, tid (+ PT)] and the slice's lang-h addends: one tagged value each
      const int o_add = slice * oc4 + tid;
      const bool has_add = tid < oc4 && o_add < E;
      const u64 *pa = xb + xo.ql + (size_t)row4 * (H + E) + H + (has_add ? o_add : 0);
      u64 kq[NH], ka = (u64)tag << 32;
      bool stale;
      int spins = 0;
      do {
        stale = false;
#pragma unroll
        for (int m = 0; m < NH; ++m) {
          const int h = tid + PT * m;
          kq[m] = h < H ? __hip_atomic_load(xb + xo.ql + (size_t)row4 * (H + E) + h, __ATOMIC_RELAXED,
                                            S2C_AG)
                        : (u64)tag << 32;
        }
        if (has_add) ka = __hip_atomic_load(pa, __ATOMIC_RELAXED, S2C_AG);
#pragma unroll
        for (int m = 0; m < NH; ++m) stale |= (u32)(kq[m] >> 32) != tag;
        stale |= (u32)(ka >> 32) != tag;
        if (stale && (++spins > P_SPIN_MAX || *(volatile int *)&s_dead)) {
          *(volatile int *)&s_dead = 1;
          break;
        }
      } while (stale);
      P_STAMP(9);
      if (tid < oc4) s_add[tid] = __uint_as_float((u32)ka);
      for (int k0 = 0; k0 < K; k0 += 4) {        // four independent chains in flight
        float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NH; ++m) {
          const int h = tid + PT * m;
          const float qm = __uint_as_float((u32)kq[m]);
          float mv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) mv[j] = (h < H && k0 + j < K) ? sM[(size_t)(k0 + j) * H + h] : 0.0f;
#pragma unroll
          for (int j = 0; j < 4; ++j) p[j] += h < H ? wa_[m] * p_tanh(mv[j] + qm) : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = row16_sum(p[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((lane & 15) == 0 && k0 + j < K) s_sc[k0 + j][tid >> 4] = p[j];
      }
      P_STAMP(13);
      lds_barrier();
      for (int i = tid; i < NP * K; i += PT) {   // NP row partials per key: one more DPP pass
        const int k = i / NP;
        float v = row16_sum(s_sc[k][i % NP]);
        if (NP == 32) v += __shfl_xor(v, 16, 64);
        if (i % NP == 0) s_s[k] = s_mask[k] == 0.0f ? -1e30f : v;
      }
      lds_barrier();
      P_STAMP(14);
      {
        float mx = -INFINITY;
        for (int k0 = 0; k0 < K; k0 += 4) {      // four LDS reads in flight
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = k0 + j < K ? s_s[k0 + j] : -INFINITY;
          mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
        }
        // one exponential per key and thread: the weighted sum and the normaliser in one pass
        const int f = tid < F ? tid : 0;
        float sum = 0.0f, acc = 0.0f;
        for (int k0 = 0; k0 < K; k0 += 4) {
          float v[4], o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = k0 + j < K ? s_s[k0 + j] : -INFINITY;
            o[j] = k0 + j < K ? sO[(size_t)(k0 + j) * F + f] : 0.0f;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float e = __expf(v[j] - mx);
            sum += e;
            acc = __builtin_fmaf(e, o[j], acc);
          }
        }
        const float inv = 1.0f / sum;
        if (slice == 0 && tid < K)
          a.ALPHA[(size_t)t * R * K + (size_t)row4 * K + tid] = __expf(s_s[tid] - mx) * inv;
        if (tid < F) {
          s_att[tid] = acc * inv;
          if (slice == 0) a.ATT[(size_t)t * R * F + (size_t)row4 * F + tid] = acc * inv;
        }
      }
      P_STAMP(15);
      lds_barrier();
      {
        const int ol = tid >> 4, pr = tid & 15, o = slice * oc4 + ol;
        float acc = 0.0f;
        if (ol < oc4)
          for (int f4 = pr; f4 < (F >> 2); f4 += 16)
            acc = fdot4(reinterpret_cast<const float4 *>(sWl + (size_t)ol * F)[f4],
                        reinterpret_cast<const float4 *>(s_att)[f4], acc);
        acc = row16_sum(acc);
        if (pr == 0 && ol < oc4 && o < E) {
          const float v = fmaxf(acc + bl + s_add[ol], 0.0f);
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
        if (STASH) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int s = 0; s < 2; ++s)
              x[rr][s] = lane + 64 * s < n4h
                             ? reinterpret_cast<const float4 *>(sH2 + (size_t)(r0 + rr) * H)[lane + 64 * s]
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (t > 0) {                      // h2 of the previous step: complete, one pass
          poll_rows<4>(rs[par ^ 1], xo.h2, H, n4h, r0, R, tag - 1u, x, &s_dead, a.backoff);
        } else {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int s = 0; s < 2; ++s) x[rr][s] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
        poll_rows<4>(rs[par], xo.x2, E, n4e, r0, R, tag, x, &s_dead, a.backoff);
      }
      P_STAMP(11);
      gru_partials_lds(sG2w, part ? E : H, x, s_red[1][wv]);
      lds_barrier();
      if (part == 0) {
        const size_t TRH = (size_t)T * RH;
        float *S = a.S + 4 * TRH + (size_t)t * RH, *C = a.C ? a.C + 4 * TRH + (size_t)t * RH : nullptr;
        GruOut o = {a.H2 + (size_t)(t + 1) * RH, S, S + TRH, S + 2 * TRH, S + 3 * TRH,
                    {C, C + TRH, C + 2 * TRH, C + 3 * TRH}};
        gru_epilogue(s_red[1][wv], s_red[1][wv + PW / 2], H, R, r0, u0, P_UB * up, ub, s_bias[1], hp2,
                     o, xb + xo.h2, tag);
      }
      P_STAMP(12);
    }
  }
  __syncthreads();
  if (tid == 0 && s_dead) {                    // loud without a host check: the last hidden state
    __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, S2C_AG);      // turns NaN, and with it the loss
    // an element this workgroup OWNS (row 0 of its first hidden unit; its own last store precedes the
    // barrier above, nobody else writes it): a NaN in somebody else's element could be overwritten by
    // its owner's ordinary store afterwards
    a.H2[(size_t)T * R * H + (u0 < H ? u0 : 0)] = __builtin_nanf("");
  }
  // ---- advance the nonce once every workgroup has read it -----------------------------------
  if (w == 0 && tid == 0) {
    int spins = 0;
    while (__hip_atomic_load(a.started, __ATOMIC_RELAXED, S2C_AG) < (u32)PG && ++spins < P_SPIN_MAX) {}
    __hip_atomic_store(a.started, 0u, __ATOMIC_RELAXED, S2C_AG);
    __hip_atomic_store(a.nonce, s_nonce + 1u, __ATOMIC_RELAXED, S2C_AG);
  }
}

// ============================================================================================
// Back-propagation through time of the same recurrence as ONE kernel (the launch chain:
// 5 dependent launches per step, decoder_fused.py).  Per step t = T-1 .. 0, five exchanges:
//   B1  gates of cell 2 from dh2'(t) -> da2 = (W_ih2^T dgi) * (x2 > 0),  dh2_part = W_hh2^T dgh + dh2' z
//   B2  attention backward of row w % 8 (row-local) from da2 -> dq; dM / dwa accumulate in registers
//   B3  dh1'(t) = [W_h^T | W_lang[:, F:]^T] [dq | da2] + dh1c
//   B4  gates of cell 1 from dh1' -> da1 = (W_ih1^T dgi) * (x1 > 0),  dh1c = W_hh1^T dgh + dh1' z
//   B5  dh2'(t-1) = W_td[:, h2 block]^T da1 + dh2_part + dH2[t-1]
// What makes the exchanged vectors small: a consumer needs the gate gradients dgi / dgh (3 H
// values per row) only as dh' * c with c = the coefficient arrays the forward kernel saved
// (s2c_dec_fwd_args.C1 / C2) -- so dh' (H values) travels and every workgroup forms the products
// itself; and the attention backward needs datt = W_lang[:, :F]^T da2 only inside two inner
// products, <datt, O_k> = <da2, P_k> and <datt, att_t> = <da2, Latt_t> with P = O W_lang[:, :F]^T and
// Latt = ATT W_lang[:, :F]^T formed by two GEMMs before the loop.
// Wave = (row quad rq, job).  In B1 / B4 job g takes gate block g of the 3 H reduction (r, z, n
// for dgi, n*r for dgh) for all of the workgroup's outputs, the blocks are summed through LDS; in
// B3 / B5 job j takes hidden unit j of the workgroup's four.  One wave per row quad polls, the
// others wait on an LDS flag and read the operand from the LDS stash.
// ============================================================================================
constexpr int B_OBE = 4;   // e-outputs (da1 / da2 columns) per workgroup  (E <= 512)
constexpr int B_UB = 4;    // hidden units per workgroup                   (H <= 512)

struct BOff {
  int dh2, da2, dq, dh1, da1, total;
};
__device__ __host__ __forceinline__ BOff boff(int H, int E) {
  BOff o;
  o.dh2 = 0;
  o.da2 = 8 * H;
  o.dq = o.da2 + 8 * E;
  o.dh1 = o.dq + 8 * H;
  o.da1 = o.dh1 + 8 * H;
  o.total = o.da1 + 8 * E;
  return o;
}

template <int NR>
__device__ __forceinline__ void stash_put(float *sX, int I, int n4, int r0, const float4 (&x)[NR][2]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int rr = 0; rr < NR; ++rr)
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (lane + 64 * s < n4) reinterpret_cast<float4 *>(sX + (size_t)(r0 + rr) * I)[lane + 64 * s] = x[rr][s];
}
__device__ __forceinline__ void stash_get(const float *sX, int I, int n4, int r0, float4 (&x)[4][2]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int s = 0; s < 2; ++s)
      x[rr][s] = lane + 64 * s < n4
                     ? reinterpret_cast<const float4 *>(sX + (size_t)(r0 + rr) * I)[lane + 64 * s]
                     : make_float4(0.f, 0.f, 0.f, 0.f);
}

// Wave -> (row quad, job).  Waves wv and wv + 4 share a SIMD; in the gate phases jobs 0, 1 (gate
// blocks r, z: all seven outputs) carry twice the dot products of jobs 2, 3 (block n: the e- or the
// h-outputs only), so every SIMD gets one of each.
__device__ __forceinline__ int bwd_job(int wv) { return (wv >> 2) == 0 ? (wv & 1) : 2 + (wv & 1); }
__device__ __forceinline__ int bwd_rq(int wv) {
  return (wv >> 2) == 0 ? ((wv >> 1) & 1) : 1 - ((wv >> 1) & 1);
}

struct GateIO {
  const float *C[4];          // coefficient arrays of the cell at step t (R x H each)
  const float *Z;             // update gate z of the cell at step t (R x H)
  const float *X;             // the ReLU output whose sign gates da (R x E at step t)
  float *dgi, *dgh;           // plain (R x 3H) at step t
  float *da;                  // plain da destination of step t, row stride ld_da
  int ld_da;
  u64 *xb_da;                 // tagged da destination (8 x E)
  const float *plain_dh;      // step T-1 of cell 2: dh' = dH2[T-1] (R x H), else NULL
  int src_off;                // tagged dh' source (pairs) inside rs_src
  u32 src_tag, tag;
  u64 *prof;                  // NULL or this wave's stamp slots of the step (3 used from prof_slot)
  int prof_slot;
};
#define G_STAMP(k)                                                               \
  do {                                                                           \
    if (PROF) {                                                                  \
      if (io.prof && lane == 0) io.prof[io.prof_slot + (k)] = __builtin_readcyclecounter(); \
    } else {                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                         \
    }                                                                            \
  } while (0)

// B1 / B4.  The weight rows of the e-outputs are in LDS (sWe: [output][3H]); those of the
// h-outputs in registers (REGS: this wave's gate block in Wh) or in LDS (sWh).
template <bool REGS, bool PROF>
__device__ __forceinline__ void gate_phase(const GateIO &io, __amdgpu_buffer_rsrc_t rs_src,
                                           const float4 (&Wh)[B_UB][2], const float *sWe,
                                           const float *sWh,
                                           int R, int H, int E, int obe, int ub, int e0, int u0,
                                           float *sD, float (*s_red)[32][4], float (*s_dhp)[4],
                                           volatile int *flag, int flagv, volatile int *s_dead) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, job = bwd_job(wv), r0 = 4 * bwd_rq(wv);
  const int sl = 4 * bwd_rq(wv);             // s_red rows of this row quad, indexed by job
  const int n4h = H >> 2;
  const float *Cg = io.C[job];
  // (1) this wave's coefficient slice, (2) the item lanes' coefficients, (3) epilogue operands
  float4 c[4][2];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int s = 0; s < 2; ++s)
      c[rr][s] = (lane + 64 * s < n4h && r0 + rr < R)
                     ? reinterpret_cast<const float4 *>(Cg + (size_t)(r0 + rr) * H)[lane + 64 * s]
                     : make_float4(0.f, 0.f, 0.f, 0.f);
  const int ij = (lane & 15) >> 2, irr = lane & 3, irow = r0 + irr;
  const bool item = lane < 16 && ij < ub && irow < R && u0 + ij < H;
  const float ci = item ? Cg[(size_t)irow * H + u0 + ij] : 0.0f;
  const bool eok = job == 0 && lane < 16 && ij < obe && irow < R && e0 + ij < E;
  const bool hok = job == 0 && lane >= 16 && lane < 32 && ij < ub && irow < R && u0 + ij < H;
  float e_op = 0.0f;
  if (eok) e_op = io.X[(size_t)irow * E + e0 + ij];
  if (hok) e_op = io.Z[(size_t)irow * H + u0 + ij];
  // (4) operand dh'
  float4 x[4][2];
  if (job < 2) {                               // jobs 0, 1 fetch two rows each (2, 3: measured no better)
    float4 y[2][2];
    const int ry = r0 + 2 * job;
    if (io.plain_dh) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          y[rr][s] = (lane + 64 * s < n4h && ry + rr < R)
                         ? reinterpret_cast<const float4 *>(io.plain_dh + (size_t)(ry + rr) * H)[lane + 64 * s]
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      poll_rows<2>(rs_src, io.src_off, H, n4h, ry, R, io.src_tag, y, s_dead, 0);
    }
    stash_put<2>(sD, H, n4h, ry, y);
    flag_raise(flag + job, flagv);
  }
  flag_wait(flag, flagv, s_dead);
  flag_wait(flag + 1, flagv, s_dead);
  stash_get(sD, H, n4h, r0, x);
  G_STAMP(0);
  // (5) gate gradients of this block
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      x[rr][s].x *= c[rr][s].x; x[rr][s].y *= c[rr][s].y;
      x[rr][s].z *= c[rr][s].z; x[rr][s].w *= c[rr][s].w;
    }
  // (6) this block's share of every output, two outputs (eight independent fold chains) at a time
  const int gblk = (job < 3 ? job : 2) * H;
#pragma unroll
  for (int o2 = 0; o2 < B_OBE + B_UB; o2 += 2) {
    float acc[8];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int o8 = o2 + d;
      const bool isE = o8 < B_OBE;
      const int k = isE ? o8 : o8 - B_OBE;
      const bool use = isE ? (k < obe && job != 3) : (k < ub && job != 2);   // wave-uniform
      float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
      if (use) {
        if (REGS && !isE) {
          w0 = Wh[k][0]; w1 = Wh[k][1];
        } else {
          const float4 *wr =
              reinterpret_cast<const float4 *>((isE ? sWe : sWh) + (size_t)k * 3 * H + gblk);
          if (lane < n4h) w0 = wr[lane];
          if (lane + 64 < n4h) w1 = wr[lane + 64];
        }
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[d * 4 + rr] = fdot4(w1, x[rr][1], fdot4(w0, x[rr][0], 0.0f));
    }
#pragma unroll
    for (int v = 0; v < 8; ++v) acc[v] = row16_sum(acc[v]);
    if ((lane & 15) == 0) {
#pragma unroll
      for (int v = 0; v < 8; ++v) s_red[sl + job][o2 * 4 + v][lane >> 4] = acc[v];
    }
  }
  G_STAMP(1);
  // (7) the gate gradients of the workgroup's own units, for the weight-gradient GEMMs
  if (item) {
    const int u = u0 + ij;
    const float v = sD[(size_t)irow * H + u] * ci;
    float *gi = io.dgi + (size_t)irow * 3 * H, *gh = io.dgh + (size_t)irow * 3 * H;
    if (job == 0) { gi[u] = v; gh[u] = v; }
    else if (job == 1) { gi[H + u] = v; gh[H + u] = v; }
    else if (job == 2) gi[2 * H + u] = v;
    else gh[2 * H + u] = v;
  }
  lds_barrier();
  G_STAMP(2);
  // (8) sum of the three blocks; da is published, the h-path stays in the workgroup
  if (eok) {
    const int v8 = ij * 4 + irr;
    float v = (red4(s_red[sl], v8) + red4(s_red[sl + 1], v8)) + red4(s_red[sl + 2], v8);
    v = e_op > 0.0f ? v : 0.0f;
    st_tag(io.xb_da + (size_t)irow * E + e0 + ij, v, io.tag);
    io.da[(size_t)irow * io.ld_da + e0 + ij] = v;
  }
  if (hok) {
    const int v8 = (B_OBE + ij) * 4 + irr;
    const float v = (red4(s_red[sl], v8) + red4(s_red[sl + 1], v8)) + red4(s_red[sl + 3], v8);
    s_dhp[irow][ij] = v + sD[(size_t)irow * H + u0 + ij] * e_op;
  }
}

template <bool PROF>
__global__ __launch_bounds__(PT, 2) void decoder_bwd_persist_kernel(s2c_dec_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ __attribute__((aligned(16))) float s_red[2][PW][32][4];    // [cell][wave][output x row][16-lane row]
  __shared__ __attribute__((aligned(16))) float s_one[2][PW][4][4];      // B3 / B5 partials (the same wave writes and reads)
  __shared__ float s_dhp[2][8][B_UB];       // [dh2_part | dh1c][row][unit of the workgroup]
  __shared__ float s_sc[P_MAXK + 1][PT / 16], s_dal[P_MAXK + 1], s_alpha[P_MAXK];
  __shared__ float s_dpre[P_MAXK][32], s_dsc[P_MAXK][32];
  __shared__ int s_flag[5][2][2];           // [dh2' | dq | da2 | dh1' | da1 stash ready][row quad][half]
  __shared__ u32 s_nonce;
  __shared__ int s_dead;
  const int R = a.R, K = a.K, H = a.H, E = a.E, T = a.T;
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int job = bwd_job(wv), rq = bwd_rq(wv), r0 = 4 * rq;
  const int n4h = H >> 2, n4e = E >> 2, HE = H + E;
  const int obe = (E + PG - 1) / PG, ub = (H + PG - 1) / PG;
  const int e0 = w * obe, u0 = w * ub;
  const int row4 = w & 7, hs0 = 32 * (w >> 3);
  float *sW4 = smem;                           // (obe + ub) x 3H  cell 1: W_ih1^T / W_hh1^T rows
  float *sWe2 = sW4 + (size_t)(obe + ub) * 3 * H;  // (obe + ub) x 3H  cell 2: W_ih2^T / W_hh2^T rows
  float *sW3 = sWe2 + (size_t)(obe + ub) * 3 * H;  // ub x (H + E)  [W_h^T | W_lang[:, F:]^T] rows
  float *sP = sW3 + (size_t)ub * HE;           // K x E    O[row4] W_lang[:, :F]^T
  float *sMs = sP + (size_t)K * E;             // K x 32   map_feat(obj_feats)[row4, :, h slice]
  // operand stashes.  dh' (B1, B4) and dq (B3) share one, da2 (B3) and da1 (B5) the other: a stash
  // is rewritten only after a poll that cannot complete before every wave of this workgroup has
  // published what it computed from the previous content
  float *sD = sMs + (size_t)K * 32;            // 8 x H
  float *sDQ = sD;
  float *sDA2 = sD + 8 * H;                    // 8 x E
  float *sDA1 = sDA2;
  if (tid == 0) {
    s_dead = 0;
    s_nonce = __hip_atomic_load(a.nonce, __ATOMIC_RELAXED, S2C_AG);
    __hip_atomic_fetch_add(a.started, 1u, __ATOMIC_RELAXED, S2C_AG);
  }
  if (tid < 20) (&s_flag[0][0][0])[tid] = 0;
  if (tid < 2 * 8 * B_UB) (&s_dhp[0][0][0])[tid] = 0.0f;
  // ---- resident operands ----------------------------------------------------------------
  for (int i = tid; i < (obe + ub) * 3 * n4h; i += PT) {
    const int o = i / (3 * n4h), q = i - o * 3 * n4h;
    const float *row = o < obe ? a.WT_ih2 + (size_t)min(e0 + o, E - 1) * 3 * H
                               : a.WT_hh2 + (size_t)min(u0 + o - obe, H - 1) * 3 * H;
    reinterpret_cast<float4 *>(sWe2)[i] = reinterpret_cast<const float4 *>(row)[q];
  }
  float4 Wtd[2];                               // B5: W_td[:, h2 block]^T row of unit `job`
#pragma unroll
  for (int s = 0; s < 2; ++s)
    Wtd[s] = (job < ub && lane + 64 * s < n4e)
                 ? reinterpret_cast<const float4 *>(a.WT_td + (size_t)min(u0 + job, H - 1) * E)[lane + 64 * s]
                 : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < (obe + ub) * 3 * n4h; i += PT) {
    const int o = i / (3 * n4h), q = i - o * 3 * n4h;
    const float *row = o < obe ? a.WT_ih1 + (size_t)min(e0 + o, E - 1) * 3 * H
                               : a.WT_hh1 + (size_t)min(u0 + o - obe, H - 1) * 3 * H;
    reinterpret_cast<float4 *>(sW4)[i] = reinterpret_cast<const float4 *>(row)[q];
  }
  for (int i = tid; i < ub * (HE >> 2); i += PT) {
    const int j = i / (HE >> 2), q = i - j * (HE >> 2);
    reinterpret_cast<float4 *>(sW3)[i] =
        reinterpret_cast<const float4 *>(a.WT_hl + (size_t)min(u0 + j, H - 1) * HE)[q];
  }
  if (row4 < R) {
    for (int i = tid; i < K * n4e; i += PT)
      reinterpret_cast<float4 *>(sP)[i] = reinterpret_cast<const float4 *>(a.P + (size_t)row4 * K * E)[i];
    for (int i = tid; i < K * 32; i += PT) {
      const int k = i >> 5, h = hs0 + (i & 31);
      sMs[i] = h < H ? a.M[((size_t)row4 * K + k) * H + h] : 0.0f;
    }
  }
  // attention items of this thread: (key, hidden unit of the slice) i = tid, tid + PT
  bool iok[2];
  float wah[2], dMacc[2] = {0.f, 0.f}, dwa_acc = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int i = tid + PT * m, h = hs0 + (i & 31);
    iok[m] = row4 < R && i < 32 * K && h < H;
    wah[m] = iok[m] ? a.wa[h] : 0.0f;
  }
  __syncthreads();
  const u32 base = (s_nonce << 6) + 1u;
  const BOff bo = boff(H, E);
  __amdgpu_buffer_rsrc_t rs[2];
  rs[0] = __builtin_amdgcn_make_buffer_rsrc((void *)a.xbuf, 0, bo.total * 8, 0x00020000);
  rs[1] = __builtin_amdgcn_make_buffer_rsrc((void *)(a.xbuf + bo.total), 0, bo.total * 8,
                                            0x00020000);
  const size_t RH = (size_t)R * H, RE = (size_t)R * E;
  const float4 nowh[B_UB][2] = {};

  for (int st = 0; st < T; ++st) {
    const int t = T - 1 - st, par = st & 1;
    const u32 tag = base + (u32)st;
    u64 *xb = a.xbuf + (size_t)par * bo.total;
    P_STAMP(0);
    // ================= B1: cell 2 ==========================================================
    {
      GateIO io;
#pragma unroll
      for (int g = 0; g < 4; ++g) io.C[g] = a.C + (size_t)(4 + g) * T * RH + (size_t)t * RH;
      io.Z = a.S + (size_t)5 * T * RH + (size_t)t * RH;
      io.X = a.X2 + (size_t)t * RE;
      io.dgi = a.DG + (size_t)(2 * T + t) * 3 * RH;
      io.dgh = a.DG + (size_t)(3 * T + t) * 3 * RH;
      io.da = a.DQA + (size_t)t * R * HE + H;
      io.ld_da = HE;
      io.xb_da = xb + bo.da2;
      io.plain_dh = st == 0 ? a.dH2 + (size_t)t * RH : nullptr;
      io.src_off = bo.dh2;
      io.src_tag = tag - 1u;
      io.tag = tag;
      io.prof = (a.prof && w == 0) ? a.prof + ((size_t)wv * T + t) * 16 : nullptr;
      io.prof_slot = 9;
      gate_phase<false, PROF>(io, rs[par ^ 1], nowh, sWe2, sWe2 + (size_t)obe * 3 * H, R, H, E, obe, ub, e0, u0, sD, s_red[0],
                       s_dhp[0], &s_flag[0][rq][0], st + 1, &s_dead);
    }
    P_STAMP(1);
    // ================= B2: attention backward of row4, hidden slice hs0 .. hs0 + 31 ==========
    if (row4 < R) {
      const size_t tr = (size_t)t * R + row4;
      const float latt = tid < E ? a.Latt[tr * E + tid] : 0.0f;
      float qv[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
        qv[m] = iok[m] ? a.QL[tr * HE + hs0 + ((tid + PT * m) & 31)] : 0.0f;
      if (tid < K) s_alpha[tid] = a.ALPHA[tr * K + tid];
      u64 kd = (u64)tag << 32;
      {
        const u64 *pd = xb + bo.da2 + (size_t)row4 * E + (tid < E ? tid : 0);
        int spins = 0;
        bool stale;
        do {
          if (tid < E) kd = __hip_atomic_load(pd, __ATOMIC_RELAXED, S2C_AG);
          stale = (u32)(kd >> 32) != tag;
          if (stale && (++spins > P_SPIN_MAX || *(volatile int *)&s_dead)) {
            *(volatile int *)&s_dead = 1;
            break;
          }
        } while (stale);
      }
      P_STAMP(2);
      const float d = tid < E ? __uint_as_float((u32)kd) : 0.0f;
      for (int k0 = 0; k0 <= K; k0 += 4) {       // <da2, P_k> and (k = K) <da2, Latt_t>
        float p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + j;
          p[j] = k < K ? (tid < E ? d * sP[(size_t)k * E + tid] : 0.0f) : (k == K ? d * latt : 0.0f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = row16_sum(p[j]);
        if ((lane & 15) == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (k0 + j <= K) s_sc[k0 + j][tid >> 4] = p[j];
        }
      }
      lds_barrier();
      for (int i = tid; i < 32 * (K + 1); i += PT) {
        float v = row16_sum(s_sc[i >> 5][i & 31]);
        v += __shfl_xor(v, 16, 64);
        if ((i & 31) == 0) s_dal[i >> 5] = v;
      }
      lds_barrier();
      const float c0 = s_dal[K];
#pragma unroll
      for (int m = 0; m < 2; ++m)
        if (iok[m]) {
          const int i = tid + PT * m, k = i >> 5, hh = i & 31;
          const float ds = s_alpha[k] * (s_dal[k] - c0);
          const float c = p_tanh(sMs[i] + qv[m]);
          const float dpre = ds * wah[m] * (1.0f - c * c);
          dMacc[m] += dpre;
          s_dpre[k][hh] = dpre;
          s_dsc[k][hh] = ds * c;
        }
      lds_barrier();
      if (tid < 32 && hs0 + tid < H) {
        float dq = 0.0f, dw = 0.0f;
        for (int k0 = 0; k0 < K; k0 += 4) {        // eight LDS reads in flight
          float u[4], v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            u[j] = k0 + j < K ? s_dpre[k0 + j][tid] : 0.0f;
            v[j] = k0 + j < K ? s_dsc[k0 + j][tid] : 0.0f;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) { dq += u[j]; dw += v[j]; }
        }
        dwa_acc += dw;
        st_tag(xb + bo.dq + (size_t)row4 * H + hs0 + tid, dq, tag);
        a.DQA[tr * HE + hs0 + tid] = dq;
      }
    }
    P_STAMP(3);
    // ================= B3: dh1' = [W_h^T | W_lang[:, F:]^T] [dq | da2] + dh1c ==================
    {
      float4 xq[4][2], xa[4][2];
      {                                          // jobs 0, 1: dq rows; jobs 2, 3: da2 rows
        float4 y[2][2];
        const int half = job & 1, ry = r0 + 2 * half;
        if (job < 2) {
          poll_rows<2>(rs[par], bo.dq, H, n4h, ry, R, tag, y, &s_dead, 0);
          stash_put<2>(sDQ, H, n4h, ry, y);
          flag_raise(&s_flag[1][rq][half], st + 1);
        } else {
          poll_rows<2>(rs[par], bo.da2, E, n4e, ry, R, tag, y, &s_dead, 0);
          stash_put<2>(sDA2, E, n4e, ry, y);
          flag_raise(&s_flag[2][rq][half], st + 1);
        }
      }
      flag_wait(&s_flag[1][rq][0], st + 1, &s_dead);
      flag_wait(&s_flag[1][rq][1], st + 1, &s_dead);
      flag_wait(&s_flag[2][rq][0], st + 1, &s_dead);
      flag_wait(&s_flag[2][rq][1], st + 1, &s_dead);
      stash_get(sDQ, H, n4h, r0, xq);
      stash_get(sDA2, E, n4e, r0, xa);
      P_STAMP(4);
      const float4 *wr = reinterpret_cast<const float4 *>(sW3 + (size_t)min(job, ub - 1) * HE);
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 wq0 = lane < n4h ? wr[lane] : z4, wq1 = lane + 64 < n4h ? wr[lane + 64] : z4;
      const float4 wa0 = lane < n4e ? wr[n4h + lane] : z4,
                   wa1 = lane + 64 < n4e ? wr[n4h + lane + 64] : z4;
      float acc[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        acc[rr] = fdot4(wa1, xa[rr][1], fdot4(wa0, xa[rr][0], fdot4(wq1, xq[rr][1], fdot4(wq0, xq[rr][0], 0.0f))));
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[rr] = row16_sum(acc[rr]);
      if ((lane & 15) == 0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) s_one[0][wv][rr][lane >> 4] = acc[rr];
      }
      __builtin_amdgcn_wave_barrier();
      const int row = r0 + lane, u = u0 + job;
      if (lane < 4 && job < ub && u < H && row < R)
        st_tag(xb + bo.dh1 + (size_t)row * H + u, red4(s_one[0][wv], lane) + s_dhp[1][row][job], tag);
    }
    P_STAMP(5);
    // ================= B4: cell 1 ==========================================================
    {
      GateIO io;
#pragma unroll
      for (int g = 0; g < 4; ++g) io.C[g] = a.C + (size_t)g * T * RH + (size_t)t * RH;
      io.Z = a.S + (size_t)T * RH + (size_t)t * RH;
      io.X = a.X1 + (size_t)t * RE;
      io.dgi = a.DG + (size_t)t * 3 * RH;
      io.dgh = a.DG + (size_t)(T + t) * 3 * RH;
      io.da = a.DA1 + (size_t)t * RE;
      io.ld_da = E;
      io.xb_da = xb + bo.da1;
      io.plain_dh = nullptr;
      io.src_off = bo.dh1;
      io.src_tag = tag;
      io.tag = tag;
      io.prof = (a.prof && w == 0) ? a.prof + ((size_t)wv * T + t) * 16 : nullptr;
      io.prof_slot = 12;
      gate_phase<false, PROF>(io, rs[par], nowh, sW4, sW4 + (size_t)obe * 3 * H, R, H, E, obe, ub, e0, u0, sD, s_red[1], s_dhp[1],
                        &s_flag[3][rq][0], st + 1, &s_dead);
    }
    P_STAMP(6);
    // ================= B5: dh2'(t-1) = W_td[:, h2 block]^T da1 + dh2_part + dH2[t-1] ==========
    if (t > 0) {
      const int row = r0 + lane, u = u0 + job;
      const bool ok = lane < 4 && job < ub && u < H && row < R;
      const float dprev = ok ? a.dH2[(size_t)(t - 1) * RH + (size_t)row * H + u] : 0.0f;
      float4 x[4][2];
      if (job < 2) {
        float4 y[2][2];
        poll_rows<2>(rs[par], bo.da1, E, n4e, r0 + 2 * job, R, tag, y, &s_dead, 0);
        stash_put<2>(sDA1, E, n4e, r0 + 2 * job, y);
        flag_raise(&s_flag[4][rq][job], st + 1);
      }
      flag_wait(&s_flag[4][rq][0], st + 1, &s_dead);
      flag_wait(&s_flag[4][rq][1], st + 1, &s_dead);
      stash_get(sDA1, E, n4e, r0, x);
      P_STAMP(7);
      float acc[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[rr] = fdot4(Wtd[1], x[rr][1], fdot4(Wtd[0], x[rr][0], 0.0f));
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[rr] = row16_sum(acc[rr]);
      if ((lane & 15) == 0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) s_one[1][wv][rr][lane >> 4] = acc[rr];
      }
      __builtin_amdgcn_wave_barrier();
      if (ok)
        st_tag(xb + bo.dh2 + (size_t)row * H + u,
               (red4(s_one[1][wv], lane) + s_dhp[0][row][job]) + dprev, tag);
      P_STAMP(8);
    }
  }
  // ---- accumulated attention gradients ----------------------------------------------------
#pragma unroll
  for (int m = 0; m < 2; ++m)
    if (iok[m]) {
      const int i = tid + PT * m;
      a.dM[((size_t)row4 * K + (i >> 5)) * H + hs0 + (i & 31)] = dMacc[m];
    }
  if (row4 < R && tid < 32 && hs0 + tid < H) a.dwa_rows[(size_t)row4 * H + hs0 + tid] = dwa_acc;
  __syncthreads();
  if (tid == 0 && s_dead) {                    // loud without a host check: a NaN gradient
    __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, S2C_AG);
    // in an element this workgroup owns where it owns one (stored above, before the barrier)
    a.dwa_rows[(row4 < R && hs0 < H) ? (size_t)row4 * H + hs0 : 0] = __builtin_nanf("");
  }
  if (w == 0 && tid == 0) {
    int spins = 0;
    while (__hip_atomic_load(a.started, __ATOMIC_RELAXED, S2C_AG) < (u32)PG && ++spins < P_SPIN_MAX) {}
    __hip_atomic_store(a.started, 0u, __ATOMIC_RELAXED, S2C_AG);
    __hip_atomic_store(a.nonce, s_nonce + 1u, __ATOMIC_RELAXED, S2C_AG);
  }
}

size_t persist_bwd_lds_bytes(int K, int H, int E) {
  const int obe = (E + PG - 1) / PG, ub = (H + PG - 1) / PG;
  return sizeof(float) * ((size_t)2 * (obe + ub) * 3 * H + (size_t)ub * (H + E) + (size_t)K * E +
                          (size_t)K * 32 + 8 * (size_t)H + 8 * (size_t)E);
}

size_t persist_lds_bytes(int K, int H, int E, int F, int grid) {
  const int ups = grid == 256 ? 1 : 2, maxoc = grid == 256 ? 16 : 32;
  return sizeof(float) * ((size_t)ups * (P_OB1 + P_OB3) * H + (ups == 2 ? 16 * (size_t)H : 0) +
                          (size_t)K * H + (size_t)K * F + (size_t)maxoc * F +
                          (size_t)ups * 6 * (H + E));
}

int g_persist = -1;

}  // namespace

extern "C" void s2c_decoder_persist_set(int on) { g_persist = on; }

// sizeof the argument structs (0: forward, 1: backward) -- for bindings to check their layout
extern "C" long long s2c_decoder_persist_args_sizeof(int which) {
  return which == 0 ? (long long)sizeof(s2c_dec_fwd_args) : (long long)sizeof(s2c_dec_bwd_args);
}

extern "C" long long s2c_decoder_fwd_persist_xbuf_pairs(int H, int E) {
  return 2LL * xoff(H, E).total;
}

// grid of the forward kernel for these shapes on the current device: 256 (x 256 threads), 128
// (x 512 threads) or 0 = not taken.  S2C_DECODER_PERSIST_GRID=128|256 forces one.
static int fwd_persist_grid(int R, int K, int H, int E, int F, int T) {
  if (g_persist < 0) {
    const char *e = getenv("S2C_DECODER_PERSIST");
    g_persist = e ? atoi(e) : 1;
  }
  if (!g_persist) return 0;
  if (R < 1 || R > 8 || K < 1 || K > P_MAXK || T < 1 || T > 62) return 0;
  if (H % 4 || E % 4 || F % 4 || H < 4 || H > 512 || E < 4 || E > 512 || F < 4 || F > 256) return 0;
  static int forced = -1;
  if (forced < 0) {
    const char *e = getenv("S2C_DECODER_PERSIST_GRID");
    forced = e ? atoi(e) : 0;
  }
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  static size_t lds_set[2][64];
  // 128 x 512 first: measured 778 vs 780 us alone and 9.35 vs 9.40 ms per training step -- the
  // phases are no faster with one wave per SIMD (forced by padding the LDS request: the same
  // 2340 cycles for a GRU cell's partial sums), so the second grid only buys headroom in LDS
  for (int cc = 0; cc < 2; ++cc) {
    const int c = 1 - cc;
    const int grid = c == 0 ? 256 : 128;
    if (forced && forced != grid) continue;
    const void *fn = c == 0 ? (const void *)decoder_fwd_persist_kernel<256, 256, false>
                            : (const void *)decoder_fwd_persist_kernel<128, 512, false>;
    const void *fnp = c == 0 ? (const void *)decoder_fwd_persist_kernel<256, 256, true>
                             : (const void *)decoder_fwd_persist_kernel<128, 512, true>;
    const size_t lds = persist_lds_bytes(K, H, E, F, grid);
    if (lds > lds_set[c][dev]) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
          hipFuncSetAttribute(fnp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        continue;
      }
      lds_set[c][dev] = lds;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, c == 0 ? 256 : 512, lds) !=
        hipSuccess) {
      (void)hipGetLastError();
      continue;
    }
    // every workgroup resident at once, with a margin of CUs for the kernels of other streams
    if ((long long)per_cu * (cus - 16) >= grid) return grid;
  }
  return 0;
}

extern "C" int s2c_decoder_fwd_persist_supported(int R, int K, int H, int E, int F, int T) {
  return fwd_persist_grid(R, K, H, E, F, T) ? 1 : 0;
}

extern "C" int s2c_decoder_fwd_persist(const s2c_dec_fwd_args *a, void *stream) {
  if (!a || a->ldtd % 4 || a->ldlang % 4) return -2;
  const int grid = fwd_persist_grid(a->R, a->K, a->H, a->E, a->F, a->T);
  if (!grid) return -2;
  const size_t lds = persist_lds_bytes(a->K, a->H, a->E, a->F, grid);
  if (grid == 256 && a->prof)
    hipLaunchKernelGGL((decoder_fwd_persist_kernel<256, 256, true>), dim3(256), dim3(256), lds,
                       (hipStream_t)stream, *a);
  else if (grid == 256)
    hipLaunchKernelGGL((decoder_fwd_persist_kernel<256, 256, false>), dim3(256), dim3(256), lds,
                       (hipStream_t)stream, *a);
  else if (a->prof)
    hipLaunchKernelGGL((decoder_fwd_persist_kernel<128, 512, true>), dim3(128), dim3(512), lds,
                       (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL((decoder_fwd_persist_kernel<128, 512, false>), dim3(128), dim3(512), lds,
                       (hipStream_t)stream, *a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_decoder_fwd_persist launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" long long s2c_decoder_bwd_persist_xbuf_pairs(int H, int E) {
  return 2LL * boff(H, E).total;
}

extern "C" int s2c_decoder_bwd_persist_supported(int R, int K, int H, int E, int T) {
  if (g_persist < 0) {
    const char *e = getenv("S2C_DECODER_PERSIST");
    g_persist = e ? atoi(e) : 1;
  }
  static int g_bwd = -1;
  if (g_bwd < 0) {
    const char *e = getenv("S2C_DECODER_PERSIST_BWD");
    g_bwd = e ? atoi(e) : 1;
  }
  if (!g_persist || !g_bwd) return 0;
  if (R < 1 || R > 8 || K < 1 || K > P_MAXK || T < 1 || T > 62) return 0;
  if (H % 4 || E % 4 || H < 4 || H > 512 || E < 4 || E > 512) return 0;
  const size_t lds = persist_bwd_lds_bytes(K, H, E);
  static int state[64];
  static size_t lds_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (state[dev] == 0 || lds > lds_set[dev]) {
    state[dev] = -1;
    if (hipFuncSetAttribute((const void *)decoder_bwd_persist_kernel<false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)decoder_bwd_persist_kernel<true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    lds_set[dev] = lds;
    state[dev] = 1;
  }
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)decoder_bwd_persist_kernel<false>,
                                                   PT, lds) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return (long long)per_cu * (cus - 16) >= PG ? 1 : 0;
}

extern "C" int s2c_decoder_bwd_persist(const s2c_dec_bwd_args *a, void *stream) {
  if (!a || !s2c_decoder_bwd_persist_supported(a->R, a->K, a->H, a->E, a->T)) return -2;
  const size_t lds = persist_bwd_lds_bytes(a->K, a->H, a->E);
  if (a->prof)
    hipLaunchKernelGGL(decoder_bwd_persist_kernel<true>, dim3(PG), dim3(PT), lds, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL(decoder_bwd_persist_kernel<false>, dim3(PG), dim3(PT), lds, (hipStream_t)stream, *a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "s2c_decoder_bwd_persist launch failed: %s\n", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
