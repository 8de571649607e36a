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
  if (S > 8 || S < 1) return set_error(-1, "aggregate_conv1: S=%d source views unsupported (at most 8)", S);
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
