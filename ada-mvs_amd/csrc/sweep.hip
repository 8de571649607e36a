// Plane sweep with register-resident taps: weighted multi-view aggregation
// (reference models/adamvs.py:495-512) for a run of hypotheses.
//
// Consecutive hypotheses of a reference pixel sample almost the same source
// pixels (at stage 1 of the cascade 192 planes span about 1.5 px of disparity),
// so the sweep runs with the hypothesis loop INSIDE the thread: C/4 lanes own
// one reference pixel, keep the four bilinear taps of every source view in
// registers and reload them only when the projection enters another source
// cell.  The feature maps are then read about once per pixel instead of once
// per pixel per plane; per plane only the projection, the bilinear weights and
// the blend are recomputed.  Nothing is assumed about the planes (any order,
// any geometry): the cell test is per lane and per plane.
//
// The aggregated similarity of the chunk goes to a workspace [Dc][B][hw][C] and
// conv1 of SliceCostRegNetRED (adamvs.py:416) runs over it as an ordinary tiled
// convolution on the matrix cores (slice_red.hip), writing c1[d].
#include <limits.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "warp_math.h"

namespace adamvs {

// grid: (ceil(hw / (256/G)), 1, B); block 256.  sim [d1-d0][B][hw][C].
// MODE 1 (MS-REDNet, reference models/msrednet.py:396-412): the same sweep with the variance over the reference and
// the S warped views instead of the weighted sum -- two accumulators (sum, sum of squares) per plane, result
// -(E[x^2] - E[x]^2) into channels [0, C) of sim [d1-d0][B][hw][Da] and, when sim_b != null, of sim_b [..][Db]
// (vw and eps_num unused).
template <int C, int SV, int MODE = 0>
__global__ __launch_bounds__(256) void k_sweep_aggregate(const float* __restrict__ feat, const float* __restrict__ rt,
                                                         PlaneSrc planes, const float* __restrict__ vw,
                                                         float* __restrict__ sim, int B, int S, int D, int d0, int d1,
                                                         int h, int w, int eps_num, float* __restrict__ sim_b = nullptr,
                                                         int Da = C, int Db = C) {
  constexpr int G = C / 4, PPB = 256 / G;
  const int hw = h * w;
  const int tid = threadIdx.x, g = tid % G;
  const int pix = blockIdx.x * PPB + tid / G;
  const int b = blockIdx.z;
  const bool live = pix < hw;
  const int pc = live ? pix : hw - 1;
  const float x = (float)(pc % w), y = (float)(pc / w);
  const f32x4 ref4 = *(const f32x4*)(feat + ((size_t)b * hw + pc) * C + 4 * g);

  // The G lanes of a pixel share the projection work: lane g projects the pixel into views g, g+G, ... and hands the
  // packed cell and the four bilinear weights to the other lanes (ds_bpermute), so a projection is computed once
  // per pixel, view and plane instead of once per lane.
  constexpr int VPL = (SV + G - 1) / G;            // views a lane projects
  float ax[VPL], ay[VPL], az[VPL], tx[VPL], ty[VPL], tz[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int sc = min(g + k * G, S - 1);
    const float* r = rt + ((size_t)b * S + sc) * 12;
    ax[k] = r[0] * x + r[1] * y + r[2];              // rot_xyz = R.[x,y,1] (module.py:549)
    ay[k] = r[3] * x + r[4] * y + r[5];
    az[k] = r[6] * x + r[7] * y + r[8];
    tx[k] = r[9]; ty[k] = r[10]; tz[k] = r[11];
  }
  // per view (all lanes): reference feature x view weight / (1e-5 + sum of view weights), cached cell + its 4 taps.
  // (adamvs.py:497-512: sum_v w_v (warp_v ref) / (1e-5 + sum_v w_v); the weights do not depend on the plane, so the
  // normalisation is folded into the per-view reference vector once per pixel)
  float wv[SV];
  f32x4 refw[SV];
  int ccell[SV];
  f32x4 t00[SV], t01[SV], t10[SV], t11[SV];
  const float* src0 = feat + ((size_t)B + b) * (size_t)hw * C + 4 * g;      // view s: + s * vstride (uniform)
  const size_t vstride = (size_t)B * hw * C;
#pragma unroll
  for (int s = 0; s < SV; ++s) {
    const int sc = min(s, S - 1);
    wv[s] = (MODE == 0 && s < S) ? vw[((size_t)sc * B + b) * hw + pc] : 0.f;
    ccell[s] = -1;
    t00[s] = t01[s] = t10[s] = t11[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float eps_term;       // eps_num (train/test twin, adamvs.py:262-300): (1e-5 + sum) / sum_v w_v instead of sum / (1e-5 + sum_v w_v)
  {
    float wsum = eps_num ? 0.f : 1e-5f;                            // adamvs.py:497
#pragma unroll
    for (int s = 0; s < SV; ++s) wsum += wv[s];
    const float inv = 1.0f / wsum;
    eps_term = eps_num ? 1e-5f * inv : 0.f;
#pragma unroll
    for (int s = 0; s < SV; ++s) refw[s] = ref4 * (wv[s] * inv);
  }
  const int gbase = (threadIdx.x & 63) & ~(G - 1);   // first lane of this pixel's group
  const PlaneLine pl = plane_line(planes, b, pc, D, hw);
  const int OA = (MODE == 1) ? Da : C;
  float* out = sim + ((size_t)b * hw + pc) * OA + 4 * g;
  const size_t ostride = (size_t)B * hw * OA;
  float* outb = (MODE == 1 && sim_b) ? sim_b + ((size_t)b * hw + pc) * Db + 4 * g : nullptr;
  const size_t ostride_b = (size_t)B * hw * Db;
  const float inv_n = 1.0f / (float)(S + 1);

  // depths: lane g of a pixel fetches plane dg+g, the group reads them back lane by lane -- one load per G planes,
  // so the (rolled) plane loop has no load that would queue behind the previous plane's store (vmcnt is in issue order)
  // vmcnt retires in order and counts stores too: a tap reload issued after the store of the previous plane waits
  // for that store to complete (~1.5 us), and with 8 pixels x S views per wave some lane reloads on nearly every
  // plane.  The results of G planes are therefore parked in LDS (each thread reads back only what it wrote: no
  // barrier) and flushed as one burst of G stores, so the reloads of G - 1 of every G planes find no store in
  // front of them.  The plane loop stays rolled (unrolled, the scheduler hoists every plane's projection: spills).
  // The burst is 8 planes for every C (with C = 16 / 8 a pixel has 4 / 2 lanes: bursts of G planes would put a store in
  // front of the reloads of every fourth / second plane); lane g then fetches planes dg+g, dg+G+g, ...
  constexpr int PK = 8, NM = PK / G;
  __shared__ f32x4 park[PK][256];
  for (int dg = d0; dg < d1; dg += PK) {
  float mydepth[NM];
#pragma unroll
  for (int k = 0; k < NM; ++k) mydepth[k] = plane_at(planes, pl, min(dg + k * G + g, d1 - 1), hw);
#pragma unroll 1
  for (int j = 0; j < PK; ++j) {
    const int d = dg + j;
    if (d >= d1) break;
    float mine_d = mydepth[0];
#pragma unroll
    for (int k = 1; k < NM; ++k) mine_d = (j / G == k) ? mydepth[k] : mine_d;      // uniform select
    const float depth = __shfl(mine_d, gbase + (j % G), 64);
    PlaneTaps mine[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) mine[k] = plane_taps(ax[k], ay[k], az[k], tx[k], ty[k], tz[k], depth, h, w);
    f32x4 acc = {eps_term, eps_term, eps_term, eps_term};
    f32x4 acc2 = ref4 * ref4;
    if (MODE == 1) acc = ref4;
#pragma unroll
    for (int s = 0; s < SV; ++s) {
      if (s >= S) break;                                         // uniform
      const int from = gbase + (s % G);
      const PlaneTaps& m = mine[s / G];
      const int cell = __shfl(m.cell, from, 64);
      const float w00 = __shfl(m.w00, from, 64), w01 = __shfl(m.w01, from, 64);
      const float w10 = __shfl(m.w10, from, 64), w11 = __shfl(m.w11, from, 64);
      if (cell != -1 && cell != ccell[s]) {                      // entered another source cell: reload the 4 taps
        ccell[s] = cell;
        load_cell_taps(src0 + s * vstride, C, cell, h, w, t00[s], t01[s], t10[s], t11[s]);
      }
      f32x4 wrp = t00[s] * w00 + t01[s] * w01 + t10[s] * w10 + t11[s] * w11;     // all-zero weights when padding
      if (MODE == 1) { acc += wrp; acc2 += wrp * wrp; }          // msrednet.py:404-407
      else acc += wrp * refw[s];                                 // adamvs.py:504-512
    }
    if (MODE == 1) { const f32x4 m = acc * inv_n; acc = m * m - acc2 * inv_n; }     // -(E[x^2] - E[x]^2), msrednet.py:411
    park[j][tid] = acc;
  }
  if (live) {
    const int nd = min(PK, d1 - dg);
#pragma unroll 1
    for (int j = 0; j < nd; ++j) {
      *(f32x4*)(out + (size_t)(dg + j - d0) * ostride) = park[j][tid];
      if (MODE == 1 && outb) *(f32x4*)(outb + (size_t)(dg + j - d0) * ostride_b) = park[j][tid];
    }
  }
  }
}

// ---------------------------------------------------------------------------
// The weighted aggregation of the hot path (MODE 0 above), restructured around what the plane loop of k_sweep_aggregate
// spends its time on (round 3; the kernel above stays for the variance mode and for feature maps of 2 GiB and more per view):
//  * projections are handed around inside a quad with DPP moves (VALU, no LDS crossbar, no lgkmcnt wait) instead of five
//    ds_bpermute per view and plane: lane q of a quad projects views q, q + 4 (both quads of a C = 32 pixel hold all
//    views; with C = 8 a quad holds two pixels and lane q of a pair projects views q, q + 2, ...);
//  * the view weight (already normalised by 1e-5 + sum of weights) is folded into the four bilinear weights by the lane
//    that projects the view, so a view costs eight packed FMAs into ONE accumulator and the reference vector enters once
//    per plane:  sim = eps + ref * sum_v sum_t (w_vt wn_v) tap_vt   (reference adamvs.py:497-512, reassociated);
//  * tap reloads are buffer loads (uniform descriptor per view, 32-bit lane offset: no 64-bit address arithmetic), and
//    a plane runs in two phases with the NEXT plane's projection in between: phase 1 compares the cells of all views and
//    issues every reload of the plane, then the next projection is computed, then phase 2 blends -- the reloads of a
//    plane overlap each other and ~40 vector instructions instead of being waited for one view after the other;
//  * generated planes (the stage path) cost a multiply and an add per plane; explicit planes are staged through LDS per
//    group of 8, fetched before the previous group's store burst (vmcnt retires in order).
template <int CTRL> __device__ __forceinline__ int dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// the value lane q of the caller's quad holds (G >= 4), or lane q of the caller's pair (G == 2: a quad = two pixels)
template <int G> __device__ __forceinline__ int quad_bcast_i(int v, int q) {
  if (G == 2) return q == 0 ? dpp_mov_i<0xA0>(v) : dpp_mov_i<0xF5>(v);          // quad_perm [0,0,2,2] / [1,1,3,3]
  switch (q) {
    case 0: return dpp_mov_i<0x00>(v);
    case 1: return dpp_mov_i<0x55>(v);
    case 2: return dpp_mov_i<0xAA>(v);
    default: return dpp_mov_i<0xFF>(v);
  }
}
template <int G> __device__ __forceinline__ float quad_bcast_f(float v, int q) {
  return __builtin_bit_cast(float, quad_bcast_i<G>(__builtin_bit_cast(int, v), q));
}

// grid: (ceil(hw / PPB), 1, B); block 256.  sim [d1-d0][B][hw][C].  GEN: planes generated (uniform / window).
// HALVES = 1: S <= 4 source views, 256 / G pixels per workgroup.  HALVES = 2 (5 ... 8 views, BASELINE cfg5): waves 0, 1 take
// views 0-3 and waves 2, 3 views 4-7 of the SAME 128 / G pixels; each half parks its partial sum of a group of 8 planes in
// LDS and the halves meet at the flush (one barrier pair per 8 planes).  Eight views in one lane need 128 registers of taps
// (208 in all: two waves per SIMD); split, a lane keeps the 122 of the four-view kernel and four waves cover the reloads.
// More than 8 source views (the reference loops over any number, adamvs.py:501): one launch per group of 8 views, views
// [vbase, vbase + 8) each; the first writes its share of the sum (plus the eps term), the others add theirs to what is there.
template <int C, int HALVES, bool GEN>
__global__ __launch_bounds__(256) void k_sweep_blend(const float* __restrict__ feat, const float* __restrict__ rt, PlaneSrc planes,
                                                     const float* __restrict__ vw, float* __restrict__ sim, int B, int S, int D,
                                                     int d0, int d1, int h, int w, int eps_num, int vbase) {
  constexpr int SV = 4, G = C / 4, NT = 256 / HALVES, PPB = NT / G, NQ = G < 4 ? G : 4, VPL = (SV + NQ - 1) / NQ, PK = 8, NM = PK / G;
  const int hw = h * w;
  const int tid = threadIdx.x, half = __builtin_amdgcn_readfirstlane(tid / NT), ltid = tid % NT;      // half: wave-uniform
  const int g = ltid % G, gq = g % NQ, pib = ltid / G;
  const int pix = blockIdx.x * PPB + pib;
  const int b = blockIdx.z;
  const bool live = pix < hw;
  const int pc = live ? pix : hw - 1;
  const float x = (float)(pc % w), y = (float)(pc / w);
  const f32x4 ref4 = *(const f32x4*)(feat + ((size_t)b * hw + pc) * C + 4 * g);
  const int v0 = vbase + SV * half;                      // first view of this half
  const int Sm = min(S - v0, SV);                        // views of this half (>= 1: HALVES = 2 only when the group has > 4 views)
  const bool accum = vbase > 0;                          // uniform: a later view group adds to the first one's result

  // normalisation of the view weights (adamvs.py:497-512), per pixel: all lanes, all S views
  float winv, eps_term;
  {
    float wsum = eps_num ? 0.f : 1e-5f;
    for (int s = 0; s < S; ++s) wsum += vw[((size_t)s * B + b) * hw + pc];
    winv = 1.0f / wsum;
    eps_term = (eps_num && !accum) ? 1e-5f * winv : 0.f;       // train/test twin (adamvs.py:262-300): (1e-5 + sum) / sum_v w_v
  }
  // the views this lane projects: v0 + gq, v0 + gq + NQ, ... (views past S-1 repeat the last one; never consumed)
  float ax[VPL], ay[VPL], az[VPL], tx[VPL], ty[VPL], tz[VPL], wn[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int sc = min(v0 + gq + k * NQ, S - 1);
    const float* r = rt + ((size_t)b * S + sc) * 12;
    ax[k] = r[0] * x + r[1] * y + r[2];              // rot_xyz = R.[x,y,1] (module.py:549)
    ay[k] = r[3] * x + r[4] * y + r[5];
    az[k] = r[6] * x + r[7] * y + r[8];
    tx[k] = r[9]; ty[k] = r[10]; tz[k] = r[11];
    wn[k] = vw[((size_t)sc * B + b) * hw + pc] * winv;
  }
  // cached cell and its four taps (4 channels of this lane) per view.  Named variables, not arrays: an array of SV
  // float4 is promoted to ONE <16 x float> value, i.e. a 512-bit register tuple that is copied whole after every
  // conditional reload (measured: 188 registers and 32 v_mov_b64 per view and plane).
#define ADAMVS_EACH_VIEW(X) X(0) X(1) X(2) X(3)
#define ADAMVS_DECL_VIEW(i) int cc##i = -1; f32x4 ta##i = {0.f, 0.f, 0.f, 0.f}, tb##i = ta##i, tc##i = ta##i, td##i = ta##i;
  ADAMVS_EACH_VIEW(ADAMVS_DECL_VIEW)
#undef ADAMVS_DECL_VIEW
  const size_t vstride = (size_t)B * hw * C;
  const float* src1 = feat + ((size_t)B + b) * (size_t)hw * C + (size_t)v0 * vstride;      // view v0 + s: + s * vstride (uniform)
  const unsigned g16 = 16u * g;
  const unsigned rowpitch = (unsigned)w * (C * 4);
  const PlaneLine pl = plane_line(planes, b, pc, D, hw);
  const unsigned ooff = live ? (unsigned)((((size_t)b * hw + pc) * C + 4 * g) * 4) : BUF_OOB;
  const size_t ostride = (size_t)B * hw * C;

  __shared__ f32x4 park[PK][256];
  __shared__ float dstash[GEN ? 1 : PK][GEN ? 1 : PPB];

  auto project = [&](float depth, PlaneTaps (&o)[VPL]) {
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      o[k] = plane_taps_scaled(ax[k], ay[k], az[k], tx[k], ty[k], tz[k], depth, h, w, wn[k]);
    }
  };
  auto depth_of = [&](int d, int j) -> float {          // plane d = plane j of the current group
    if (GEN) return plane_value(pl.lo, pl.step, d);
    return dstash[GEN ? 0 : j][GEN ? 0 : pib];
  };

  float nd[NM];                                     // explicit planes: lane g fetches planes dg + g, dg + G + g, ... of the NEXT group
  if (!GEN) {
#pragma unroll
    for (int k = 0; k < NM; ++k) nd[k] = pl.q[(size_t)min(d0 + k * G + g, d1 - 1) * hw];
  }
  for (int dg = d0; dg < d1; dg += PK) {
    if (!GEN) {
      __syncthreads();                              // every lane is done with the previous group's planes
      if (half == 0) {
#pragma unroll
        for (int k = 0; k < NM; ++k) dstash[GEN ? 0 : k * G + g][GEN ? 0 : pib] = nd[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NM; ++k) nd[k] = pl.q[(size_t)min(dg + PK + k * G + g, d1 - 1) * hw];     // in flight during the group
    }
    PlaneTaps nxt[VPL];
    project(depth_of(dg, 0), nxt);
#pragma unroll 1
    for (int j = 0; j < PK; ++j) {
      const int d = dg + j;
      if (d >= d1) break;
      PlaneTaps cur[VPL];
#pragma unroll
      for (int k = 0; k < VPL; ++k) cur[k] = nxt[k];
      // ---- phase 1: the cells of this plane, every reload issued
      auto reload = [&](int s, int& cc, f32x4& ta, f32x4& tb, f32x4& tc, f32x4& td) {
        const int cell = quad_bcast_i<G>(cur[s / NQ].cell, s % NQ);
        if (cell != -1 && cell != cc) {                            // entered another source cell
          cc = cell;
          const buf_rsrc rs = make_rsrc(src1 + (size_t)s * vstride);
          const int ix = (cell & 0xFFFF) - 1, iy = (cell >> 16) - 1;
          const unsigned xa = (unsigned)max(ix, 0) * (C * 4) + g16, xb = (unsigned)min(ix + 1, w - 1) * (C * 4) + g16;
          const unsigned ya = (unsigned)max(iy, 0) * rowpitch, yb = (unsigned)min(iy + 1, h - 1) * rowpitch;
          ta = buf_load4(rs, ya + xa);
          tb = buf_load4(rs, ya + xb);
          tc = buf_load4(rs, yb + xa);
          td = buf_load4(rs, yb + xb);
        }
      };
#define ADAMVS_RELOAD_VIEW(i) if (i < Sm) reload(i, cc##i, ta##i, tb##i, tc##i, td##i);
      ADAMVS_EACH_VIEW(ADAMVS_RELOAD_VIEW)
#undef ADAMVS_RELOAD_VIEW
      __builtin_amdgcn_sched_barrier(0);
      // ---- the next plane's projection, under the reloads
      if (j + 1 < PK) project(depth_of(min(d + 1, d1 - 1), min(j + 1, PK - 1)), nxt);
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase 2: blend
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};       // even / odd views: two FMA chains
      auto blend = [&](int s, const f32x4& ta, const f32x4& tb, const f32x4& tc, const f32x4& td) {
        const PlaneTaps& m = cur[s / NQ];
        const float w00 = quad_bcast_f<G>(m.w00, s % NQ), w01 = quad_bcast_f<G>(m.w01, s % NQ);
        const float w10 = quad_bcast_f<G>(m.w10, s % NQ), w11 = quad_bcast_f<G>(m.w11, s % NQ);
        f32x4& a = acc[s & 1];                                     // all-zero weights when padding
        a = ta * w00 + a; a = tb * w01 + a; a = tc * w10 + a; a = td * w11 + a;
      };
#define ADAMVS_BLEND_VIEW(i) if (i < Sm) blend(i, ta##i, tb##i, tc##i, td##i);
      ADAMVS_EACH_VIEW(ADAMVS_BLEND_VIEW)
#undef ADAMVS_BLEND_VIEW
      // one half: the finished value; two halves: this half's partial sum (the halves meet at the flush)
      park[j][tid] = HALVES == 1 ? ref4 * (acc[0] + acc[1]) + eps_term : acc[0] + acc[1];
      // every reload of this plane has been consumed; stated explicitly (the builtin, so that the compiler's wait insertion
      // knows it): without it the reload branches of the next plane, which build their addresses in the dead tap
      // registers, each start with a conservative s_waitcnt vmcnt and the reloads of a plane serialise
      wait_vmem_all();
    }
    const int nd_ = min(PK, d1 - dg);
    if (HALVES == 1) {
#pragma unroll 1
      for (int j = 0; j < nd_; ++j) {
        const buf_rsrc ro = make_rsrc(sim + (size_t)(dg + j - d0) * ostride);
        f32x4 v = park[j][tid];
        if (accum) v += buf_load4(ro, ooff);
        buf_store4(ro, ooff, v);
      }
    } else {
      __syncthreads();                              // both halves have parked the group
#pragma unroll 1
      for (int j = half; j < nd_; j += 2) {         // half h finishes planes h, h + 2, ...: views 0-3 first, then 4-7 (fixed order)
        const buf_rsrc ro = make_rsrc(sim + (size_t)(dg + j - d0) * ostride);
        f32x4 v = ref4 * (park[j][ltid] + park[j][NT + ltid]) + eps_term;
        if (accum) v += buf_load4(ro, ooff);
        buf_store4(ro, ooff, v);
      }
      __syncthreads();                              // before the next group overwrites the parked sums
    }
  }
#undef ADAMVS_EACH_VIEW
}

// (Round 6 built this kernel with a STORE WAVE -- seven compute waves parking four planes at a time in LDS, an eighth wave of the
// workgroup doing nothing but the global stores, so that no compute wave ever has a store in its in-order vmcnt queue.  The timing
// builds had put the stores at 20 - 27 % of the aggregation + conv1 phase.  Bit-identical, and slower: cfg3 / bf16x3 / 32 tiles
// 8.65 / 7.48 / 2.92 -> 9.08 / 7.64 / 2.92 ms per stage, cfg2 / 256 tiles 75.5 -> 78.9 ms: the workgroup barrier that hands a
// buffer over couples the seven compute waves -- every reload stall of one is then a stall of all -- and an eighth of the compute
// waves is gone.  tools/experiments/sweep_store_wave/, profiles/r06_sweep_timing.txt.)
template <int C>
static int launch_sweep_variance_c(const float* feat, const float* rt, PlaneSrc planes, float* out_a, int Da, float* out_b,
                                   int Db, int B, int S, int D, int h, int w, hipStream_t st) {
  dim3 grid(cdiv(h * w, 256 / (C / 4)), 1, B);
  if (S <= 4)
    hipLaunchKernelGGL((k_sweep_aggregate<C, 4, 1>), grid, dim3(256), 0, st, feat, rt, planes, nullptr, out_a, B, S, D, 0, D, h, w, 0,
                       out_b, Da, Db);
  else
    hipLaunchKernelGGL((k_sweep_aggregate<C, 8, 1>), grid, dim3(256), 0, st, feat, rt, planes, nullptr, out_a, B, S, D, 0, D, h, w, 0,
                       out_b, Da, Db);
  ADAMVS_CHECK_LAUNCH("sweep_variance");
  return 0;
}

// -variance of (reference, S warped views) for all D planes with register-resident taps; S <= 8, C in {8, 16, 32}
int launch_sweep_variance(const float* feat, const float* rt, const float* planes, float* out_a, int Da, float* out_b, int Db,
                          int B, int S, int C, int D, int h, int w, hipStream_t st) {
  const PlaneSrc ps = explicit_planes(planes);
  if (C == 32) return launch_sweep_variance_c<32>(feat, rt, ps, out_a, Da, out_b, Db, B, S, D, h, w, st);
  if (C == 16) return launch_sweep_variance_c<16>(feat, rt, ps, out_a, Da, out_b, Db, B, S, D, h, w, st);
  if (C == 8) return launch_sweep_variance_c<8>(feat, rt, ps, out_a, Da, out_b, Db, B, S, D, h, w, st);
  return set_error(-1, "sweep_variance: C=%d unsupported (8, 16 or 32)", C);
}

template <int C>
static int launch_sweep_c(const float* feat, const float* rt, PlaneSrc planes, const float* vw, float* sim, int B, int S,
                          int D, int d0, int d1, int h, int w, int eps_num, hipStream_t st) {
  dim3 grid(cdiv(h * w, 256 / (C / 4)), 1, B);
  if ((size_t)B * h * w * C * 4 < 0x7fffffffu) {      // 32-bit lane offsets inside one view / one plane; larger batches: the kernel above
    const bool gen = planes.mode != PLANES_EXPLICIT;
    for (int vbase = 0; vbase < S; vbase += 8) {        // groups of 8 views; the later ones accumulate
      if (S - vbase <= 4) {
        if (gen) hipLaunchKernelGGL((k_sweep_blend<C, 1, true>), grid, dim3(256), 0, st, feat, rt, planes, vw, sim, B, S, D, d0, d1, h, w, eps_num, vbase);
        else hipLaunchKernelGGL((k_sweep_blend<C, 1, false>), grid, dim3(256), 0, st, feat, rt, planes, vw, sim, B, S, D, d0, d1, h, w, eps_num, vbase);
      } else {                    // 5 ... 8 views: two halves of the workgroup share 128 / G pixels
        const dim3 grid2(cdiv(h * w, 128 / (C / 4)), 1, B);
        if (gen) hipLaunchKernelGGL((k_sweep_blend<C, 2, true>), grid2, dim3(256), 0, st, feat, rt, planes, vw, sim, B, S, D, d0, d1, h, w, eps_num, vbase);
        else hipLaunchKernelGGL((k_sweep_blend<C, 2, false>), grid2, dim3(256), 0, st, feat, rt, planes, vw, sim, B, S, D, d0, d1, h, w, eps_num, vbase);
      }
      ADAMVS_CHECK_LAUNCH("sweep_blend");
    }
    return 0;
  }
  if (S > 8) return set_error(-1, "aggregate_conv1: S=%d source views need the blend sweep (maps of 2 GiB per view take at most 8)", S);
  if (S <= 4)
    hipLaunchKernelGGL((k_sweep_aggregate<C, 4>), grid, dim3(256), 0, st, feat, rt, planes, vw, sim, B, S, D, d0, d1, h, w, eps_num);
  else
    hipLaunchKernelGGL((k_sweep_aggregate<C, 8>), grid, dim3(256), 0, st, feat, rt, planes, vw, sim, B, S, D, d0, d1, h, w, eps_num);
  ADAMVS_CHECK_LAUNCH("sweep_aggregate");
  return 0;
}

// chunk of planes held in the similarity workspace between the sweep and conv1
int sweep_chunk_planes(int D) { return D < 32 ? D : 32; }

size_t sweep_workspace_floats(int B, int C, int D, int h, int w) {
  return (size_t)sweep_chunk_planes(D) * B * h * w * C;
}

// hypotheses [d0, d1) (at most sweep_chunk_planes(D)): weighted aggregation into sim_ws, conv1 of it into
// c1_chunk [d1-d0][B][hw][8]
int launch_sweep_conv1_chunk(const float* feat, const float* rt, PlaneSrc planes, const float* vw, const float* w1pk,
                             float* c1_chunk, float* sim_ws, int B, int S, int C, int D, int d0, int d1, int h, int w, int precision,
                             int eps_num, hipStream_t st) {
  if (S < 1) return set_error(-1, "aggregate_conv1: S=%d source views", S);
  int rc;
  if (C == 32) rc = launch_sweep_c<32>(feat, rt, planes, vw, sim_ws, B, S, D, d0, d1, h, w, eps_num, st);
  else if (C == 16) rc = launch_sweep_c<16>(feat, rt, planes, vw, sim_ws, B, S, D, d0, d1, h, w, eps_num, st);
  else if (C == 8) rc = launch_sweep_c<8>(feat, rt, planes, vw, sim_ws, B, S, D, d0, d1, h, w, eps_num, st);
  else return set_error(-1, "aggregate_conv1: C=%d unsupported (8, 16 or 32)", C);
  if (rc) return rc;
  // conv1 over the (d1-d0)*B similarity maps of the chunk; image n = dlocal*B + b lands in c1_chunk[dlocal][b]
  return launch_conv1(sim_ws, w1pk, c1_chunk, (d1 - d0) * B, C, h, w, precision, st);
}

int launch_sweep_conv1(const float* feat, const float* rt, PlaneSrc planes, const float* vw, const float* w1pk,
                       float* c1, float* sim_ws, int B, int S, int C, int D, int h, int w, int precision, int eps_num,
                       hipStream_t st) {
  const int dc = sweep_chunk_planes(D);
  for (int d0 = 0; d0 < D; d0 += dc) {
    const int d1 = (d0 + dc < D) ? d0 + dc : D;
    if (int rc = launch_sweep_conv1_chunk(feat, rt, planes, vw, w1pk, c1 + (size_t)d0 * B * h * w * 8, sim_ws, B, S, C, D, d0, d1, h, w,
                                          precision, eps_num, st))
      return rc;
  }
  return 0;
}

}  // namespace adamvs
