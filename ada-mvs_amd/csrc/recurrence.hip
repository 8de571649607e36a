// Software-pipelined recurrence of one cascade stage.
//
//   for d in range(D): reg_cost, state1, state2 = SliceCostRegNetRED(cost_d, state1, state2)
//   reference models/adamvs.py:495-527 (the loop), 415-424 (one step), models/module.py:24-52 (ConvGRUCell)
//
// One step is a chain  gates1 -> cand1 -> conv2 -> gates2 -> cand2 -> decoder  of six dependent tile loops.  Only two
// of the dependencies cross steps -- level 1 of step t+1 needs h1[t], level 2 of step t+1 needs h2[t] -- so the steps
// are skewed: while level 1 works on hypothesis t, level 2 works on t-1 and the decoder on t-2, and roles that do
// not depend on each other share a launch ("slot"): workgroups [0, n0) of the grid walk the tiles of one role,
// [n0, n0+n1) those of the next, ... (k_slot).  Three dependent launches per hypothesis instead of six, each with
// matrix-bound and memory-bound tile loops resident side by side on every CU:
//
//   fp32     slot A(t): gates1(t) | conv2(t-1)      slot B(t): cand1(t) | gates2(t-1)     slot C(t): cand2(t-1) | decoder(t-2)
//   two launches (schedule 3, small stages):  A(t): gates1(t) | conv2+gates2(t-1) in two halves    B(t): cand1(t) | cand2(t-1) | decoder(t-2)
//   bf16x3   both GRU levels are one kernel each (gru1, gru2):
//            slot A(t): gru1(t) | conv2(t-1)        slot B(t): gru1(t)' | gru2(t-1) | decoder(t-2)
//            (the level-1 kernel has no dependant inside its step: its tiles are dealt to both launches)
//            or ONE launch per hypothesis (schedule 5, the smallest stages):  gru1(t) | conv2(t-1) | gru2(t-2) | decoder(t-3)
//
// State rings: h1 of step t lives in h1[t % 4] (the decoder still reads h1[t-2] or h1[t-3] while cand1 writes h1[t]),
// h2 and conv2's output of step t in h2[t % 2], c2[t % 2].  The arithmetic of every tile is that of the one-role kernels (slice_roles.h): the pipelined stage
// is bit-identical to the sequential one, which tests/test_hip_parity.py asserts.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "persistent.h"
#include "slice_roles.h"
#include "slice_roles_bx3.h"
#include "slice_roles_fused.h"
#include "slice_roles_wino.h"

namespace adamvs {

struct NopRole {
  struct Args {};
  static constexpr size_t LDS_BYTES = 0;
  static int tiles_x(const Args&) { return 1; }
  static int tiles_y(const Args&) { return 1; }
  static __device__ __forceinline__ void run(const Args&, const TileGrid&, TileRange, int, int, float*) {}
};

template <class R0, class R1, class R2>
struct SlotArgs {
  typename R0::Args a0; TileGrid g0; TileRange r0; int n0;     // role 0: workgroups [0, n0)
  typename R1::Args a1; TileGrid g1; TileRange r1; int n1;     // role 1: workgroups [n0, n0 + n1)
  typename R2::Args a2; TileGrid g2; TileRange r2;             // role 2: the rest of the grid
};

// The grid is divided between the roles (workgroups [0, n0) run role 0, ...) in proportion to their estimated work.
// (Tried and dropped: every workgroup walking its share of every role, one role after the other -- no cost model, but
// 20-30 % slower: each workgroup then pays every role's prologue and holds no role's weights for long.)
// Workgroups per CU the register budget of a launch is held to: the fused split-bf16 GRU roles are written for two
// (256 registers); without the bound the compiler lets a launch that contains them grow past that.
template <class R> struct min_blocks { static constexpr int value = 1; };
template <> struct min_blocks<Gru1FusedBx3Role> { static constexpr int value = 2; };
template <> struct min_blocks<Gru2FusedBx3Role> { static constexpr int value = 2; };
// the same for the fp32 roles that sit at the edge of two workgroups per CU: a launch that contains them is held to 256 registers
// (accumulator registers included: the decoder's five accumulators pushed the one-launch schedule to 260 and one workgroup per CU,
// stage 1 of cfg4's share 4.8 -> 6.4 ms, before this bound)
template <> struct min_blocks<Gru2FusedRole<4>> { static constexpr int value = 2; };
template <> struct min_blocks<ConvWinoRole<16, 16, 2, EPI_GATES>> { static constexpr int value = 2; };
// (Gru1FusedRole<8, 2> held to three workgroups per CU -- 168 registers, 18 of its 194 spilled -- is slower than at two:
//  stage 2 of cfg3 at 32 tiles 28.3 against 27.8 ms, the separate kernels 26.5)
constexpr int imax3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

template <class R0, class R1, class R2>
__global__ __launch_bounds__(256, imax3(min_blocks<R0>::value, min_blocks<R1>::value, min_blocks<R2>::value)) void k_slot(SlotArgs<R0, R1, R2> s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wg = blockIdx.x;                  // workgroup-uniform: the role dispatch is a scalar branch
  if (wg < s.n0) R0::run(s.a0, s.g0, s.r0, wg, s.n0, lds);
  else if (wg < s.n0 + s.n1) R1::run(s.a1, s.g1, s.r1, wg - s.n0, s.n1, lds);
  else R2::run(s.a2, s.g2, s.r2, wg - s.n0 - s.n1, (int)gridDim.x - s.n0 - s.n1, lds);
}

constexpr size_t max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

// A role instance of a slot: its arguments (null = absent), the fraction [f0, f1) of its tiles this slot takes and its
// cost per tile relative to the other roles (what the grid is divided by).
template <class R>
struct RoleUse {
  const typename R::Args* args;
  float cost;
  float f0, f1;
};

// Divide `total` workgroups over the roles in proportion to their work, at least one and at most one per tile.
template <int NR>
static void split_grid(int total, const long (&tiles)[NR], const double (&work)[NR], int (&n)[NR]) {
  double W = 0;
  for (int i = 0; i < NR; ++i) W += tiles[i] ? work[i] : 0;
  int sum = 0;
  for (int i = 0; i < NR; ++i) {
    n[i] = 0;
    if (!tiles[i]) continue;
    long v = (long)(total * (work[i] / W) + 0.5);
    n[i] = (int)(v < 1 ? 1 : (v > tiles[i] ? tiles[i] : v));
    sum += n[i];
  }
  while (sum > total) {                       // rounding: take from the role with the most workgroups
    int k = -1;
    for (int i = 0; i < NR; ++i) if (n[i] > 1 && (k < 0 || n[i] > n[k])) k = i;
    if (k < 0) break;
    --n[k]; --sum;
  }
  while (sum < total) {                       // give to the role with the most work per workgroup that can still grow
    int k = -1;
    for (int i = 0; i < NR; ++i)
      if (tiles[i] && n[i] < tiles[i] && (k < 0 || work[i] / n[i] > work[k] / n[k])) k = i;
    if (k < 0) break;
    ++n[k]; ++sum;
  }
}

template <class R0, class R1, class R2>
static int launch_slot(const RoleUse<R0>& u0, const RoleUse<R1>& u1, const RoleUse<R2>& u2, int B, hipStream_t st, const char* name) {
  constexpr size_t lds = max3(R0::LDS_BYTES, R1::LDS_BYTES, R2::LDS_BYTES);
  static_assert(lds <= 160 * 1024, "slot exceeds the LDS of a CU");
  auto kern = k_slot<R0, R1, R2>;
  static const int capacity = [&] {                                  // per instantiation; once, thread-safely (magic static)
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return resident_blocks(kern, 256, lds);
  }();
  SlotArgs<R0, R1, R2> s;
  memset(&s, 0, sizeof(s));
  long tiles[3] = {0, 0, 0};
  double work[3] = {0, 0, 0};
  int rc;
  auto prep = [&](auto* args, auto& dst_args, TileGrid& g, TileRange& r, float cost, float f0, float f1, int i, auto role) -> int {
    typedef decltype(role) R;
    if (!args) return 0;
    dst_args = *args;
    if (int e = make_tile_grid(g, R::tiles_x(*args), R::tiles_y(*args), B)) return e;
    r.begin = (int)((double)g.ntiles * f0 + 0.5);
    r.end = f1 >= 1.0f ? g.ntiles : (int)((double)g.ntiles * f1 + 0.5);
    tiles[i] = r.end - r.begin;
    work[i] = (double)tiles[i] * cost;
    return 0;
  };
  if ((rc = prep(u0.args, s.a0, s.g0, s.r0, u0.cost, u0.f0, u0.f1, 0, R0()))) return rc;
  if ((rc = prep(u1.args, s.a1, s.g1, s.r1, u1.cost, u1.f0, u1.f1, 1, R1()))) return rc;
  if ((rc = prep(u2.args, s.a2, s.g2, s.r2, u2.cost, u2.f0, u2.f1, 2, R2()))) return rc;
  const long all = tiles[0] + tiles[1] + tiles[2];
  if (all == 0) return 0;
  int n[3];
  split_grid<3>((int)(all < capacity ? all : capacity), tiles, work, n);
  s.n0 = n[0];
  s.n1 = n[1];
  hipLaunchKernelGGL(kern, dim3(n[0] + n[1] + n[2]), dim3(256), lds, st, s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "%s: %s", name, hipGetErrorString(e));
  return 0;
}

// ---- a launch of four roles: when both ConvGRU levels are one kernel each, nothing inside a step depends on anything else
// once the levels are skewed by one hypothesis each -- gru1(t) | conv2(t-1) | gru2(t-2) | decoder(t-3) -- and a
// hypothesis costs ONE dependent launch.
template <class R0, class R1, class R2, class R3>
struct SlotArgs4 {
  typename R0::Args a0; TileGrid g0; TileRange r0; int n0;
  typename R1::Args a1; TileGrid g1; TileRange r1; int n1;
  typename R2::Args a2; TileGrid g2; TileRange r2; int n2;
  typename R3::Args a3; TileGrid g3; TileRange r3;
};
constexpr int imax4(int a, int b, int c, int d) { return imax3(imax3(a, b, c), d, d); }
constexpr size_t max4(size_t a, size_t b, size_t c, size_t d) { return max3(max3(a, b, c), d, d); }

template <class R0, class R1, class R2, class R3>
__global__ __launch_bounds__(256, imax4(min_blocks<R0>::value, min_blocks<R1>::value, min_blocks<R2>::value, min_blocks<R3>::value))
void k_slot4(SlotArgs4<R0, R1, R2, R3> s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wg = blockIdx.x;
  if (wg < s.n0) R0::run(s.a0, s.g0, s.r0, wg, s.n0, lds);
  else if (wg < s.n0 + s.n1) R1::run(s.a1, s.g1, s.r1, wg - s.n0, s.n1, lds);
  else if (wg < s.n0 + s.n1 + s.n2) R2::run(s.a2, s.g2, s.r2, wg - s.n0 - s.n1, s.n2, lds);
  else R3::run(s.a3, s.g3, s.r3, wg - s.n0 - s.n1 - s.n2, (int)gridDim.x - s.n0 - s.n1 - s.n2, lds);
}

template <class R0, class R1, class R2, class R3>
static int launch_slot4(const RoleUse<R0>& u0, const RoleUse<R1>& u1, const RoleUse<R2>& u2, const RoleUse<R3>& u3, int B, hipStream_t st,
                        const char* name) {
  constexpr size_t lds = max4(R0::LDS_BYTES, R1::LDS_BYTES, R2::LDS_BYTES, R3::LDS_BYTES);
  static_assert(lds <= 64 * 1024, "slot exceeds the default dynamic LDS limit");
  auto kern = k_slot4<R0, R1, R2, R3>;
  static const int capacity = resident_blocks(kern, 256, lds);
  SlotArgs4<R0, R1, R2, R3> s;
  memset(&s, 0, sizeof(s));
  long tiles[4] = {0, 0, 0, 0};
  double work[4] = {0, 0, 0, 0};
  int rc;
  auto prep = [&](auto* args, auto& dst_args, TileGrid& g, TileRange& r, float cost, float f0, float f1, int i, auto role) -> int {
    typedef decltype(role) R;
    if (!args) return 0;
    dst_args = *args;
    if (int e = make_tile_grid(g, R::tiles_x(*args), R::tiles_y(*args), B)) return e;
    r.begin = (int)((double)g.ntiles * f0 + 0.5);
    r.end = f1 >= 1.0f ? g.ntiles : (int)((double)g.ntiles * f1 + 0.5);
    tiles[i] = r.end - r.begin;
    work[i] = (double)tiles[i] * cost;
    return 0;
  };
  if ((rc = prep(u0.args, s.a0, s.g0, s.r0, u0.cost, u0.f0, u0.f1, 0, R0()))) return rc;
  if ((rc = prep(u1.args, s.a1, s.g1, s.r1, u1.cost, u1.f0, u1.f1, 1, R1()))) return rc;
  if ((rc = prep(u2.args, s.a2, s.g2, s.r2, u2.cost, u2.f0, u2.f1, 2, R2()))) return rc;
  if ((rc = prep(u3.args, s.a3, s.g3, s.r3, u3.cost, u3.f0, u3.f1, 3, R3()))) return rc;
  const long all = tiles[0] + tiles[1] + tiles[2] + tiles[3];
  if (all == 0) return 0;
  int n[4];
  split_grid<4>((int)(all < capacity ? all : capacity), tiles, work, n);
  s.n0 = n[0]; s.n1 = n[1]; s.n2 = n[2];
  hipLaunchKernelGGL(kern, dim3(n[0] + n[1] + n[2] + n[3]), dim3(256), lds, st, s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "%s: %s", name, hipGetErrorString(e));
  return 0;
}

// ---- the roles of one step ------------------------------------------------------------------------------------
typedef ConvSmallRole<8, 8, 1, 1, EPI_GATES> Gates1;
typedef Cand1TwoRowRole Cand1;
typedef ConvSmallRole<8, 0, 1, 2, EPI_RELU> Conv2;
typedef ConvSmallRole<16, 16, 2, 1, EPI_GATES> Gates2;
typedef ConvSmallRole<16, 16, 1, 1, EPI_CAND> Cand2;
typedef ConvWinoRole<8, 8, 1, EPI_GATES> Gates1W;      // the same convolutions in the F(2x2, 3x3) form (slice_roles_wino.h), 8 x 32 tiles
typedef ConvWinoRole<16, 16, 2, EPI_GATES> Gates2W;
typedef ConvWinoRole<16, 16, 1, EPI_CAND> Cand2W;
typedef Gru1FusedRole<4, 2> Gru1S;      // fp32, both levels fused: 4 x 30 / 4 x 14 tiles for stages with few tiles per CU
typedef Gru2FusedRole<4> Gru2S;
typedef Gru1FusedBx3Role Gru1Bx;
typedef ConvSmallBx3Role<8, 0, 1, 2, BXE_RELU> Conv2Bx;
typedef Gru2FusedBx3Role Gru2Bx;

// Relative cost per tile (microseconds of a whole-chip launch per tile, measured at cfg2's stage-1 shape; only the
// ratios inside a slot matter).  ADAMVS_RECUR_COSTS="g1,c1,v2,g2,c2,dec,k1bx,v2bx,g2bx,c2bx" overrides (tuning).
// fa, fb: the fused bf16x3 level-1 kernel runs tiles [0,fa) in slot A, [fa,fb) in B, [fb,1) in C; fd: the decoder of the
// bf16x3 schedule runs [0,fd) in slot B and the rest in C.
struct RoleCosts { float g1, c1, v2, g2, c2, dec, k1bx, v2bx, g2bx, c2bx, fa, fb, fd; };
// the fused fp32 roles (slice_roles_fused.h), same unit: per tile, from their MFMA counts at the rate of the roles above
// (4 x 30 level 1: 624 MFMAs; 8 x 30: 1104; 4 x 14 level 2: 1152).  ADAMVS_RECUR_COSTS_FUSED="k1s,k1l,k2s" overrides.
struct FusedCosts { float k1s, k1l, k2s; };
static const FusedCosts& fused_costs() {
  static const FusedCosts c = [] {
    FusedCosts v = {10.0f, 23.0f, 15.0f};        // (13, ., 20 by MFMA count; swept at 1 / 2 / 4 tiles: a plateau over 8-11, 12-16, cfg4's share 17.05 -> 16.97 ms)
    if (const char* e = getenv("ADAMVS_RECUR_COSTS_FUSED")) {
      float a, b, d;
      if (sscanf(e, "%f,%f,%f", &a, &b, &d) == 3 && a > 0.f && b > 0.f && d > 0.f && a < 1e6f && b < 1e6f && d < 1e6f) v = FusedCosts{a, b, d};
      else fprintf(stderr, "adamvs: ADAMVS_RECUR_COSTS_FUSED ignored (three positive values)\n");
    }
    return v;
  }();
  return c;
}
static RoleCosts parse_role_costs() {
  RoleCosts c = {2.85f, 4.07f, 2.39f, 9.98f, 5.53f, 5.0f, 11.3f, 3.2f, 12.0f, 5.0f, 0.6f, 0.8f, 0.42f};       // (dec: 8 x 30 tiles since round 4; k1bx 13.8 until the two-row candidate of round 5)
  if (const char* e = getenv("ADAMVS_RECUR_COSTS")) {
    float v[13];
    bool ok = sscanf(e, "%f,%f,%f,%f,%f,%f,%f,%f,%f,%f,%f,%f,%f", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7, v + 8, v + 9, v + 10,
                     v + 11, v + 12) == 13;
    for (int i = 0; ok && i < 10; ++i) ok = v[i] > 0.f && v[i] < 1e6f;          // a zero cost would divide the grid by zero work
    for (int i = 10; ok && i < 13; ++i) ok = v[i] >= 0.f && v[i] <= 1.f;         // tile fractions
    if (ok) c = RoleCosts{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12]};
    else fprintf(stderr, "adamvs: ADAMVS_RECUR_COSTS ignored (13 comma-separated values: 10 positive costs, 3 fractions in [0,1])\n");
  }
  return c;
}
// initialised once, thread-safely (C++11 magic static): nn.DataParallel calls the stage from one thread per device
static const RoleCosts& role_costs() {
  static const RoleCosts c = parse_role_costs();
  return c;
}

// option recur_mode: 0 = six launches per step, states updated in place; 1 / 3 / 5 = software-pipelined slots, schedule
// 1 / 3 / 5 (recurrence_lags); -1 (the default) = by size.  Measured on MI355X (profiles/r02_recurrence_schedules.txt): sharing launches does
// not make the roles faster -- a slot takes the sum of its roles' standalone times, the decoder more -- so what the
// pipeline buys is three launch latencies per hypothesis, which pays while a step is latency-bound (few tiles per CU:
// cfg4's 4 tiles per GPU 24.3 -> 22.0 ms) and costs 2-3 % once every role fills the chip several times over.
int recurrence_mode(int precision, long pixels) {
  const int forced = opt(OPT_RECUR_MODE);
  if (forced == 0 || forced == 1 || forced == 3 || forced == 5) return forced;
  // B * h * w of the stage.  Measured (profiles/r02_recurrence_schedules.txt): fp32 -- two launches per hypothesis win up
  // to ~200k pixels (cfg4's 4 tiles per GPU at stage 1: 7.8 -> 5.9 -> 5.1 ms), three up to ~800k, six beyond;
  // bf16x3 (both GRU levels are one kernel each: four launches per hypothesis, two, or one) -- one up to ~300k pixels
  // (cfg4's share at stage 1: 5.27 / 3.75 / 3.49 ms with four / two / one; at 295k 9.95 / 9.40 / 8.99), two up to ~1M (at
  // 590k 15.6 / 15.1 / 15.5; at 1.18M a tie), four beyond (at 2.36M 51.3 against 55.2).
  // (round 4, after the fused levels' gate rows were interleaved: at 295k -- stage 2 of cfg4's share -- 2.77 ms with one launch per
  // hypothesis, 2.67 with two; at 74k 3.45 against 3.69)
  if (precision != PRECISION_FP32) return pixels <= 250000 ? 5 : (pixels <= 1000000 ? 3 : 0);
  // round 4, fp32 with both ConvGRU levels one kernel each (slice_roles_fused.h): ONE launch per hypothesis pays on the
  // smallest stages only -- cfg4's share at stage 1 (74k pixels) 5.01 -> 4.79 ms; at 295k 4.44 -> 5.28, at 1.18M 1.72 -> 2.50:
  // the fused tiles execute 36 % more MFMAs and the stage is not latency- but throughput-bound as soon as every CU has a few tiles.
  // (Level 1 fused on 8 x 30 tiles with a launch per role -- schedule 6 of round 4 -- lost at every size: cfg3 at 32 tiles 23.4 / 26.5 /
  // 13.2 ms per stage -> 25.4 / 27.8 / 13.4; cfg2 at 128 tiles 79.6 -> 86.5.  Removed in round 6; docs/history/tried_without_gain.md.)
  // With the gate convolutions in the F(2x2, 3x3) form (slice_roles_wino.h) in both: cfg3 at 32 tiles, stage 1 (590k pixels)
  // 22.2 ms with three launches per hypothesis, 22.7 with six (direct kernels: 23.3 / 23.9); at 2.36M 27.3 against 22.4.
  return pixels <= 100000 ? 5 : (pixels <= 200000 ? 3 : (pixels <= 800000 ? 1 : 0));
}

template <class R> static RoleUse<R> use(const typename R::Args* a, float cost, float f0 = 0.f, float f1 = 1.f) {
  return RoleUse<R>{a, cost, f0, f1};
}
template <class R> static RoleUse<R> none() { return RoleUse<R>{nullptr, 0.f, 0.f, 1.f}; }

// How far level-2's candidate and the decoder run behind level 1 (in hypotheses) under a schedule.
//   schedule 1 (the default)         A: gates1(t) | conv2(t-1)    B: cand1(t) | gates2(t-1)              C: cand2(t-1) | decoder(t-2)
//   schedule 3                       A: gates1(t) | conv2+gates2(t-1)   B: cand1(t) | cand2(t-1) | decoder(t-2)
//                                    two dependent launches per hypothesis (Conv2Gates2Role fuses conv2 into the gate kernel);
//                                    bf16x3: the fused level-1 kernel takes the place of gates1 and cand1, its tiles dealt to both
// (Schedule 2 of round 2 -- gates2, the one role that needs 234 registers, in a launch of its own -- was never chosen by size and
// was removed in round 6.)
//   schedule 5 (both levels one kernel each)   gru1(t) | conv2(t-1) | gru2(t-2) | decoder(t-3): ONE launch per hypothesis
RecurLags recurrence_lags(int schedule, int precision) {
  if (schedule == 5) return RecurLags{2, 3};       // one launch: gru2 two, the decoder three behind
  return RecurLags{1, 2};
}

// Pipeline step t of a stage, 0 <= t < D + lag_dec: level 1 of hypothesis t, conv2 / gates2 of t-1, cand2 of t-lag_c2,
// decoder of t-lag_dec (each only while its hypothesis exists).  c1_t = conv1 output of hypothesis t; vol_dec = base of
// the cost-volume chunk that receives the decoded hypothesis as its plane d_dec of D_vol.
int launch_recur_pipeline_step(const GruStateRing& rb, const FuseWeights& fw, int B, int h, int w, int D, int t, const float* c1_t,
                               float* vol_dec, int D_vol, int d_dec, int in_up, int precision, int schedule, hipStream_t st) {
  const int h2 = h / 2, w2 = w / 2;
  const RecurLags lag = recurrence_lags(schedule, precision);
  const int s2 = t - 1, sc = t - lag.c2, sd = t - lag.dec;            // hypotheses of conv2/gates2, cand2, decoder
  const bool l1 = t < D, l2 = s2 >= 0 && s2 < D, lc = sc >= 0 && sc < D, dec = sd >= 0 && sd < D;
  // ring slots: s may be as low as -1 - lag.dec while the pipeline fills (roles of hypotheses that do not exist yet are
  // disabled, but their argument structs are still built): a true modulo, never a negative index
  auto H1 = [&](int s) { return rb.h1[((s % 4) + 4) % 4]; };
  auto H2 = [&](int s) { return rb.h2[((s % 2) + 2) % 2]; };
  auto C2 = [&](int s) { return rb.c2[((s % 2) + 2) % 2]; };
  const RoleCosts& k = role_costs();
  int rc;
  DecoderArgs da{H2(sd), H1(sd), fw.upconv1, fw.upconv1_b, fw.final_w, vol_dec, h, w, D_vol, d_dec};

  if (precision == PRECISION_BF16X3) {
    // both GRU levels are one kernel each: two launches per hypothesis whatever the schedule number
    //   A: gru1(t) [0,f) | conv2(t-1)        B: gru1(t) [f,1) | gru2(t-1) | decoder(t-2)
    // (the level-1 kernel has no dependant inside its step: its tiles are dealt to both launches so that they carry equal work)
    Gru1Args g1{c1_t, H1(t - 1), H1(t), (const bf16x8*)fw.gates1, fw.gates1_b, (const bf16x8*)fw.cand1, fw.cand1_b, h, w};
    SmallConvArgsBx v2{H1(s2), nullptr, (const bf16x8*)fw.conv2, nullptr, C2(s2), nullptr, nullptr, h, w, h2, w2, 16, nullptr};
    if (schedule == 5) {
      // ONE launch per hypothesis: gru1(t) | conv2(t-1) | gru2(t-2) | decoder(t-3) -- every role reads only what earlier launches wrote
      Gru2Args g2s{C2(sc), H2(sc - 1), H2(sc), (const bf16x8*)fw.gates2, fw.gates2_b, (const bf16x8*)fw.cand2, fw.cand2_b, h2, w2};
      if (in_up)
        return launch_slot4<Gru1Bx, Conv2Bx, Gru2Bx, DecoderRole<true>>(
            l1 ? use<Gru1Bx>(&g1, k.k1bx) : none<Gru1Bx>(), l2 ? use<Conv2Bx>(&v2, k.v2bx) : none<Conv2Bx>(), lc ? use<Gru2Bx>(&g2s, k.g2bx) : none<Gru2Bx>(),
            dec ? use<DecoderRole<true>>(&da, k.dec) : none<DecoderRole<true>>(), B, st, "recurrence, one launch (bf16x3)");
      return launch_slot4<Gru1Bx, Conv2Bx, Gru2Bx, DecoderRole<false>>(
          l1 ? use<Gru1Bx>(&g1, k.k1bx) : none<Gru1Bx>(), l2 ? use<Conv2Bx>(&v2, k.v2bx) : none<Conv2Bx>(), lc ? use<Gru2Bx>(&g2s, k.g2bx) : none<Gru2Bx>(),
          dec ? use<DecoderRole<false>>(&da, k.dec) : none<DecoderRole<false>>(), B, st, "recurrence, one launch (bf16x3)");
    }
    Gru2Args g2{C2(s2), H2(s2 - 1), H2(s2), (const bf16x8*)fw.gates2, fw.gates2_b, (const bf16x8*)fw.cand2, fw.cand2_b, h2, w2};
    // per level-1 tile (8 x 30 pixels): 0.94 conv2 tiles (4 x 16 at half resolution), 0.54 gru2 tiles (8 x 14), 1 decoder tile (8 x 30)
    float f = 0.5f + ((0.54f * k.g2bx + 1.0f * k.dec) - 0.94f * k.v2bx) / (2.f * k.k1bx);
    f = !l2 ? 1.0f : (f < 0.1f ? 0.1f : (f > 1.f ? 1.f : f));
    if (l1 || l2)
      if ((rc = launch_slot<Gru1Bx, Conv2Bx, NopRole>(l1 ? use<Gru1Bx>(&g1, k.k1bx, 0.f, f) : none<Gru1Bx>(),
                                                      l2 ? use<Conv2Bx>(&v2, k.v2bx) : none<Conv2Bx>(), none<NopRole>(), B, st,
                                                      "recurrence slot A (bf16x3)")))
        return rc;
    const bool rest = l1 && f < 1.f;
    if (!(rest || l2 || dec)) return 0;
    if (in_up)
      return launch_slot<Gru1Bx, Gru2Bx, DecoderRole<true>>(rest ? use<Gru1Bx>(&g1, k.k1bx, f, 1.f) : none<Gru1Bx>(),
                                                            l2 ? use<Gru2Bx>(&g2, k.g2bx) : none<Gru2Bx>(),
                                                            dec ? use<DecoderRole<true>>(&da, k.dec) : none<DecoderRole<true>>(), B, st,
                                                            "recurrence slot B (bf16x3)");
    return launch_slot<Gru1Bx, Gru2Bx, DecoderRole<false>>(rest ? use<Gru1Bx>(&g1, k.k1bx, f, 1.f) : none<Gru1Bx>(),
                                                           l2 ? use<Gru2Bx>(&g2, k.g2bx) : none<Gru2Bx>(),
                                                           dec ? use<DecoderRole<false>>(&da, k.dec) : none<DecoderRole<false>>(), B, st,
                                                           "recurrence slot B (bf16x3)");
  }

  if (schedule == 5) {
    // fp32 with both ConvGRU levels one kernel each (slice_roles_fused.h): one launch per hypothesis, as in bf16x3 above
    const FusedCosts& fk = fused_costs();
    Gru1F32Args f1{c1_t, H1(t - 1), H1(t), fw.gates1, fw.gates1_b, fw.cand1, fw.cand1_b, h, w};
    SmallConvArgs v2s{H1(s2), nullptr, fw.conv2, nullptr, C2(s2), nullptr, h, w, h2, w2, 16, nullptr};
    Gru2F32Args f2{C2(sc), H2(sc - 1), H2(sc), fw.gates2, fw.gates2_b, fw.cand2, fw.cand2_b, h2, w2};
    if (in_up)
      return launch_slot4<Gru1S, Conv2, Gru2S, DecoderRole<true>>(
          l1 ? use<Gru1S>(&f1, fk.k1s) : none<Gru1S>(), l2 ? use<Conv2>(&v2s, k.v2) : none<Conv2>(), lc ? use<Gru2S>(&f2, fk.k2s) : none<Gru2S>(),
          dec ? use<DecoderRole<true>>(&da, k.dec) : none<DecoderRole<true>>(), B, st, "recurrence, one launch (fp32)");
    return launch_slot4<Gru1S, Conv2, Gru2S, DecoderRole<false>>(
        l1 ? use<Gru1S>(&f1, fk.k1s) : none<Gru1S>(), l2 ? use<Conv2>(&v2s, k.v2) : none<Conv2>(), lc ? use<Gru2S>(&f2, fk.k2s) : none<Gru2S>(),
        dec ? use<DecoderRole<false>>(&da, k.dec) : none<DecoderRole<false>>(), B, st, "recurrence, one launch (fp32)");
  }
  SmallConvArgs g1{c1_t, H1(t - 1), fw.gates1, fw.gates1_b, rb.rh1, rb.u1, h, w, h, w, 16, nullptr};
  SmallConvArgs c1{c1_t, rb.rh1, fw.cand1, fw.cand1_b, H1(t), rb.u1, h, w, h, w, 8, H1(t - 1)};
  SmallConvArgs v2{H1(s2), nullptr, fw.conv2, nullptr, C2(s2), nullptr, h, w, h2, w2, 16, nullptr};
  SmallConvArgs g2{C2(s2), H2(s2 - 1), fw.gates2, fw.gates2_b, rb.rh2, rb.u2, h2, w2, h2, w2, 32, nullptr};
  SmallConvArgs c2{C2(sc), rb.rh2, fw.cand2, fw.cand2_b, H2(sc), rb.u2, h2, w2, h2, w2, 16, H2(sc - 1)};
  if (schedule == 3) {
    Conv2Gates2Args vg{H1(s2), H2(s2 - 1), fw.conv2, fw.gates2, fw.gates2_b, C2(s2), rb.rh2, rb.u2, h, w, h2, w2};
    if (l1 || l2)
      if ((rc = launch_slot<Gates1, Conv2Gates2Role<0>, Conv2Gates2Role<1>>(
               l1 ? use<Gates1>(&g1, k.g1) : none<Gates1>(), l2 ? use<Conv2Gates2Role<0>>(&vg, 0.5f * k.g2 + k.v2) : none<Conv2Gates2Role<0>>(),
               l2 ? use<Conv2Gates2Role<1>>(&vg, 0.5f * k.g2 + k.v2) : none<Conv2Gates2Role<1>>(), B, st, "recurrence slot A (schedule 3)")))
        return rc;
    if (!(l1 || lc || dec)) return 0;
    if (in_up)
      return launch_slot<Cand1, Cand2, DecoderRole<true>>(l1 ? use<Cand1>(&c1, k.c1) : none<Cand1>(), lc ? use<Cand2>(&c2, k.c2) : none<Cand2>(),
                                                          dec ? use<DecoderRole<true>>(&da, k.dec) : none<DecoderRole<true>>(), B, st,
                                                          "recurrence slot B (schedule 3)");
    return launch_slot<Cand1, Cand2, DecoderRole<false>>(l1 ? use<Cand1>(&c1, k.c1) : none<Cand1>(), lc ? use<Cand2>(&c2, k.c2) : none<Cand2>(),
                                                         dec ? use<DecoderRole<false>>(&da, k.dec) : none<DecoderRole<false>>(), B, st,
                                                         "recurrence slot B (schedule 3)");
  }
  if (schedule == 1 && (gru_wino_mask() & 7) == 7 && fw.gates1_w && fw.gates2_w && fw.cand2_w) {
    // schedule 1 with the gate convolutions of both levels and the level-2 candidate in the F(2x2, 3x3) form (8 x 32 tiles; cost
    // per tile = four 4 x 16 tiles of the direct role times the measured ratio of the one-role launches: 0.76 / 0.69 / 0.89)
    SmallConvArgs g1w = g1, g2w = g2, c2w = c2;
    g1w.wpk = fw.gates1_w; g2w.wpk = fw.gates2_w; c2w.wpk = fw.cand2_w;
    const float kg1 = 4.f * 0.76f * k.g1, kg2 = 4.f * 0.69f * k.g2, kc2 = 4.f * 0.89f * k.c2;
    if (l1 || l2)
      if ((rc = launch_slot<Gates1W, Conv2, NopRole>(l1 ? use<Gates1W>(&g1w, kg1) : none<Gates1W>(), l2 ? use<Conv2>(&v2, k.v2) : none<Conv2>(),
                                                     none<NopRole>(), B, st, "recurrence slot A (F(2x2,3x3))")))
        return rc;
    if (l1 || l2)
      if ((rc = launch_slot<Cand1, Gates2W, NopRole>(l1 ? use<Cand1>(&c1, k.c1) : none<Cand1>(), l2 ? use<Gates2W>(&g2w, kg2) : none<Gates2W>(),
                                                     none<NopRole>(), B, st, "recurrence slot B (F(2x2,3x3))")))
        return rc;
    if (lc || dec) {
      if (in_up)
        rc = launch_slot<Cand2W, DecoderRole<true>, NopRole>(lc ? use<Cand2W>(&c2w, kc2) : none<Cand2W>(),
                                                             dec ? use<DecoderRole<true>>(&da, k.dec) : none<DecoderRole<true>>(),
                                                             none<NopRole>(), B, st, "recurrence slot C (F(2x2,3x3))");
      else
        rc = launch_slot<Cand2W, DecoderRole<false>, NopRole>(lc ? use<Cand2W>(&c2w, kc2) : none<Cand2W>(),
                                                              dec ? use<DecoderRole<false>>(&da, k.dec) : none<DecoderRole<false>>(),
                                                              none<NopRole>(), B, st, "recurrence slot C (F(2x2,3x3))");
      if (rc) return rc;
    }
    return 0;
  }
  if (l1 || l2)
    if ((rc = launch_slot<Gates1, Conv2, NopRole>(l1 ? use<Gates1>(&g1, k.g1) : none<Gates1>(), l2 ? use<Conv2>(&v2, k.v2) : none<Conv2>(),
                                                  none<NopRole>(), B, st, "recurrence slot A")))
      return rc;
  if (l1 || l2)
    if ((rc = launch_slot<Cand1, Gates2, NopRole>(l1 ? use<Cand1>(&c1, k.c1) : none<Cand1>(), l2 ? use<Gates2>(&g2, k.g2) : none<Gates2>(),
                                                  none<NopRole>(), B, st, "recurrence slot B")))
      return rc;
  if (lc || dec) {
    if (in_up)
      rc = launch_slot<Cand2, DecoderRole<true>, NopRole>(lc ? use<Cand2>(&c2, k.c2) : none<Cand2>(),
                                                          dec ? use<DecoderRole<true>>(&da, k.dec) : none<DecoderRole<true>>(),
                                                          none<NopRole>(), B, st, "recurrence slot C");
    else
      rc = launch_slot<Cand2, DecoderRole<false>, NopRole>(lc ? use<Cand2>(&c2, k.c2) : none<Cand2>(),
                                                           dec ? use<DecoderRole<false>>(&da, k.dec) : none<DecoderRole<false>>(),
                                                           none<NopRole>(), B, st, "recurrence slot C");
    if (rc) return rc;
  }
  return 0;
}

// ---------------------------------------------------------------------------
// Soft-argmin over the regularised cost slices of one chunk of hypotheses (reference models/adamvs.py:516-531):
// p = exp(cost) (no max subtraction), E += p, M = max(M, p) (initial 0, strict <), A += depth_d p in hypothesis order;
// after the last chunk depth = A / (E + 1e-10), confidence = M / (E + 1e-10).  The running E, M, A of a pixel cross
// chunks through `acc` [3][B*Ho*Wo]; the sums are formed in the order of the single-pass kernel, so chunking does
// not change a bit.  When IN_UP the hypothesis plane is the 2x bilinear upsample (align_corners=False) of
// planes[b][d] (adamvs.py:521-522).  vol [B][vol_D][Ho*Wo] holds hypotheses d0 .. d0+nd-1 as its planes 0 .. nd-1.
template <bool IN_UP>
__global__ void k_soft_argmin_chunk(const float* __restrict__ vol, int vol_D, PlaneSrc planes, int D, int d0, int nd,
                                    float* __restrict__ acc, int first, int last, float* __restrict__ depth,
                                    float* __restrict__ conf, int h, int w, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int Ho = IN_UP ? 2 * h : h, Wo = IN_UP ? 2 * w : w;
  int X = (int)(i % Wo), Y = (int)((i / Wo) % Ho);
  size_t b = i / ((size_t)Wo * Ho);
  size_t hw = (size_t)h * w, HW = (size_t)Ho * Wo;
  int y0 = Y, y1 = Y, x0 = X, x1 = X; float ly = 0.f, lx = 0.f;
  if (IN_UP) {
    float sy = fmaxf(((float)Y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf(((float)X + 0.5f) * 0.5f - 0.5f, 0.f);
    y0 = (int)sy; x0 = (int)sx;
    y1 = y0 + (y0 < h - 1 ? 1 : 0); x1 = x0 + (x0 < w - 1 ? 1 : 0);
    ly = sy - (float)y0; lx = sx - (float)x0;
  }
  const float* v = vol + b * vol_D * HW + (size_t)Y * Wo + X;
  // the (up to) four plane lines the pixel blends; generated planes cost a multiply and an add per hypothesis
  const PlaneLine p00 = plane_line(planes, b, (size_t)y0 * w + x0, D, hw), p01 = plane_line(planes, b, (size_t)y0 * w + x1, D, hw);
  const PlaneLine p10 = plane_line(planes, b, (size_t)y1 * w + x0, D, hw), p11 = plane_line(planes, b, (size_t)y1 * w + x1, D, hw);
  float E = 0.f, M = 0.f, A = 0.f;
  if (!first) { E = acc[i]; M = acc[total + i]; A = acc[2 * total + i]; }
  // eight planes per trip: their loads are issued together (one load in flight per thread ran this kernel at 2.9 TB/s); the sums keep
  // their order
#pragma unroll 8
  for (int d = 0; d < nd; ++d) {
    float pr = __expf(v[(size_t)d * HW]);
    float dep;
    if (IN_UP) {
      float top = plane_at(planes, p00, d0 + d, hw) * (1.f - lx) + plane_at(planes, p01, d0 + d, hw) * lx;
      float bot = plane_at(planes, p10, d0 + d, hw) * (1.f - lx) + plane_at(planes, p11, d0 + d, hw) * lx;
      dep = top * (1.f - ly) + bot * ly;
    } else {
      dep = plane_at(planes, p00, d0 + d, hw);
    }
    M = (M < pr) ? pr : M;
    A = dep * pr + A;
    E = E + pr;
  }
  if (last) {
    float den = E + 1e-10f;
    depth[i] = A / den;
    conf[i] = M / den;
  } else {
    acc[i] = E; acc[total + i] = M; acc[2 * total + i] = A;
  }
}

int launch_soft_argmin_chunk(const float* vol, int vol_D, PlaneSrc planes, int D, int d0, int nd, float* acc, int first, int last,
                             float* depth, float* conf, int B, int h, int w, int in_up, hipStream_t st) {
  size_t total = (size_t)B * (in_up ? 4 : 1) * h * w;
  unsigned nb = (unsigned)((total + 255) / 256);
  if (in_up)
    hipLaunchKernelGGL((k_soft_argmin_chunk<true>), dim3(nb), dim3(256), 0, st, vol, vol_D, planes, D, d0, nd, acc, first, last, depth,
                       conf, h, w, total);
  else
    hipLaunchKernelGGL((k_soft_argmin_chunk<false>), dim3(nb), dim3(256), 0, st, vol, vol_D, planes, D, d0, nd, acc, first, last, depth,
                       conf, h, w, total);
  ADAMVS_CHECK_LAUNCH("soft_argmin (chunk)");
  return 0;
}

}  // namespace adamvs
