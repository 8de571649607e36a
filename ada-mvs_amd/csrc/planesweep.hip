// Plane-sweep kernels (gfx950).
//
//   k_pair_similarity   pass A of InferDepthNet0.forward, reference
//                       models/adamvs.py:464-478: per source view, per
//                       hypothesis d: sim[d] = mean_c(ref[c] * warp_d(src)[c])
//   (pass B, the weighted aggregation of adamvs.py:495-512, lives in sweep.hip)
//
// Feature maps are channel-last [view][B][h*w][C]; a bilinear tap of one
// pixel is one contiguous C*4-byte line, fetched by C/4 neighbouring lanes as
// one float4 each; the per-pixel channel reduction is a wavefront shuffle tree.
#include <limits.h>

#include "common.h"
#include "kernels.h"
#include "conv_frag.h"
#include "planes.h"
#include "warp_math.h"

namespace adamvs {

// Sum over the G = 2, 4 or 8 neighbouring lanes that share a pixel, on the VALU (DPP), result in every lane.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  if (G >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // [2,3,0,1]
  if (G >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}

// The value the LAST of the G neighbouring lanes of a pixel holds, in every lane of the group (DPP).
template <int G>
__device__ __forceinline__ int group_last(int v) {
  if (G == 2) return __builtin_amdgcn_update_dpp(0, v, 0xF5, 0xF, 0xF, true);                 // quad_perm [1,1,3,3]
  if (G == 4) return __builtin_amdgcn_update_dpp(0, v, 0xFF, 0xF, 0xF, true);                 // quad_perm [3,3,3,3]
  const int lo = __builtin_amdgcn_update_dpp(0, v, 0x157, 0xF, 0x3, true);                    // row_share:7 -> lanes 0-7 of every row
  return __builtin_amdgcn_update_dpp(lo, v, 0x15F, 0xF, 0xC, false);                          // row_share:15 -> lanes 8-15
}

// ---------------------------------------------------------------------------
// grid: (pixel groups, S, B); block 256.  sim [S][B][hw][Dw] (channel-last in d; Dw >= D: channels [D, Dw) are zeros -- the
// width CostRegNet2D runs at when D is not one its kernels are built for, kernels.h::costreg_width).
template <int C>
__global__ __launch_bounds__(256) void k_pair_similarity(const float* __restrict__ feat, const float* __restrict__ rt,
                                                         PlaneSrc planes, float* __restrict__ sim,
                                                         int B, int S, int D, int h, int w, int Dw) {
  constexpr int G = C / 4;          // lanes per pixel
  constexpr int PPB = 256 / G;      // pixels per block
  const int hw = h * w;
  const int tid = threadIdx.x;
  const int g = tid % G;
  const int pix = blockIdx.x * PPB + tid / G;
  const int s = blockIdx.y, b = blockIdx.z;
  const bool live = pix < hw;
  const int pc = live ? pix : hw - 1;
  const int y = pc / w, x = pc % w;
  const f32x4 ref4 = *(const f32x4*)(feat + ((size_t)b * hw + pc) * C + 4 * g);       // view 0
  const float* src = feat + ((size_t)(s + 1) * B + b) * (size_t)hw * C + 4 * g;
  const float* r = rt + ((size_t)b * S + s) * 12;
  const float fx = (float)x, fy = (float)y;
  // rot_xyz = R.[x,y,1] once per pixel (module.py:549); per plane: * depth + t, divide (550-553)
  const float ax = r[0] * fx + r[1] * fy + r[2], ay = r[3] * fx + r[4] * fy + r[5], az = r[6] * fx + r[7] * fy + r[8];
  const float tx = r[9], ty = r[10], tz = r[11];
  const PlaneLine pl = plane_line(planes, b, pc, D, hw);
  float* out = sim + (((size_t)s * B + b) * hw + pc) * Dw;
  // sim = mean_c ref[c] * sum_t w_t tap_t[c] = (1/C) sum_t w_t * dot(ref, tap_t): while consecutive planes fall into
  // the same source cell (192 planes span ~1.5 px at stage 1) the four tap . ref dot products do not change, so they
  // are what stays in registers -- reduced over the G lanes of the pixel once per cell -- and a plane costs its own
  // lane one projection and four FMAs.  The G lanes of a pixel project G different planes (lane g: plane d0+g); the
  // cells are passed around with ds_bpermute in plane order, so any sequence of cells is handled, one reload each.
  int ccell = -1;
  float d00 = 0.f, d01 = 0.f, d10 = 0.f, d11 = 0.f;
  const int gbase = (threadIdx.x & 63) & ~(G - 1);
  for (int d0 = 0; d0 < D; d0 += G) {
    float keep = 0.f;
    const PlaneTaps mine = plane_taps(ax, ay, az, tx, ty, tz, plane_at(planes, pl, min(d0 + g, D - 1), hw), h, w);
    // The cell of the group's LAST plane, on the VALU (DPP).  Planes are monotone in disparity nearly everywhere, so the G
    // planes of a group leave the cached cell at most once: every lane then sits in the cached cell or in the last
    // lane's -- ONE reload per group, no walk through the planes (the walk below costs a ds_bpermute round trip per plane).
    const int cn = group_last<G>(mine.cell);
    if (__all(mine.cell == ccell || mine.cell == -1)) {            // whole wave still inside its cached cells
      keep = (mine.w00 * d00 + mine.w01 * d01 + mine.w10 * d10 + mine.w11 * d11) * (1.0f / (float)C);
    } else if (__all(cn != -1 && (mine.cell == ccell || mine.cell == cn || mine.cell == -1))) {
      float n00 = d00, n01 = d01, n10 = d10, n11 = d11;
      if (cn != ccell) {                                           // the same decision in all G lanes of a pixel
        f32x4 t00, t01, t10, t11;
        load_cell_taps(src, C, cn, h, w, t00, t01, t10, t11);
        const f32x4 m00 = t00 * ref4, m01 = t01 * ref4, m10 = t10 * ref4, m11 = t11 * ref4;
        n00 = group_sum<G>((m00.x + m00.y) + (m00.z + m00.w));
        n01 = group_sum<G>((m01.x + m01.y) + (m01.z + m01.w));
        n10 = group_sum<G>((m10.x + m10.y) + (m10.z + m10.w));
        n11 = group_sum<G>((m11.x + m11.y) + (m11.z + m11.w));
      }
      const bool old = mine.cell == ccell;                         // a plane off the image has four zero weights: either set
      keep = (mine.w00 * (old ? d00 : n00) + mine.w01 * (old ? d01 : n01) + mine.w10 * (old ? d10 : n10) + mine.w11 * (old ? d11 : n11)) *
             (1.0f / (float)C);
      ccell = cn; d00 = n00; d01 = n01; d10 = n10; d11 = n11;
    } else
#pragma unroll
    for (int j = 0; j < G; ++j) {                                  // planes past D-1 repeat the last one, never stored
      const int cell = __shfl(mine.cell, gbase + j, 64);
      if (cell != -1 && cell != ccell) {
        ccell = cell;
        f32x4 t00, t01, t10, t11;
        load_cell_taps(src, C, cell, h, w, t00, t01, t10, t11);
        const f32x4 m00 = t00 * ref4, m01 = t01 * ref4, m10 = t10 * ref4, m11 = t11 * ref4;
        d00 = group_sum<G>((m00.x + m00.y) + (m00.z + m00.w));
        d01 = group_sum<G>((m01.x + m01.y) + (m01.z + m01.w));
        d10 = group_sum<G>((m10.x + m10.y) + (m10.z + m10.w));
        d11 = group_sum<G>((m11.x + m11.y) + (m11.z + m11.w));
      }
      // padding taps carry weight 0 (and a plane that misses the image has cell -1 and four zero weights)
      if (j == g) keep = (mine.w00 * d00 + mine.w01 * d01 + mine.w10 * d10 + mine.w11 * d11) * (1.0f / (float)C);
    }
    if (live && d0 + g < D) out[d0 + g] = keep;       // G lanes x 4 B contiguous per pixel
  }
  if (live)
    for (int d = D + g; d < Dw; d += G) out[d] = 0.f;
}

}  // namespace adamvs

using namespace adamvs;

int adamvs::launch_pair_similarity(const float* feat, const float* rt, PlaneSrc planes, float* sim, int B, int S, int C, int D, int h,
                                   int w, hipStream_t st, int Dw) {
  if (Dw <= 0) Dw = D;
  ADAMVS_CHECK_ARG(Dw >= D && S <= 65535 && B <= 65535, "pair_similarity: Dw=%d < D=%d, or S=%d / B=%d above 65535", Dw, D, S, B);
  ADAMVS_CHECK_ARG(feat && rt && planes.p && sim && B > 0 && S > 0 && D > 0 && h > 1 && w > 1,
                   "pair_similarity: bad arguments (B=%d S=%d D=%d h=%d w=%d)", B, S, D, h, w);
  ADAMVS_CHECK_ARG(C == 8 || C == 16 || C == 32, "pair_similarity: C=%d unsupported (8, 16 or 32)", C);
  int hw = h * w;
  if (C == 32)
    hipLaunchKernelGGL((k_pair_similarity<32>), dim3(cdiv(hw, 32), S, B), dim3(256), 0, st, feat, rt, planes, sim, B, S, D, h, w, Dw);
  else if (C == 16)
    hipLaunchKernelGGL((k_pair_similarity<16>), dim3(cdiv(hw, 64), S, B), dim3(256), 0, st, feat, rt, planes, sim, B, S, D, h, w, Dw);
  else
    hipLaunchKernelGGL((k_pair_similarity<8>), dim3(cdiv(hw, 128), S, B), dim3(256), 0, st, feat, rt, planes, sim, B, S, D, h, w, Dw);
  ADAMVS_CHECK_LAUNCH("pair_similarity");
  return 0;
}

extern "C" int adamvs_pair_similarity(const float* feat, const float* rt, const float* planes, float* sim, int B, int S,
                                      int C, int D, int h, int w, void* stream) {
  return launch_pair_similarity(feat, rt, explicit_planes(planes), sim, B, S, C, D, h, w, (hipStream_t)stream);
}

extern "C" size_t adamvs_aggregate_conv1_workspace_bytes(int B, int C, int D, int h, int w) {
  return sweep_workspace_floats(B, C, D, h, w) * sizeof(float);
}

extern "C" int adamvs_aggregate_conv1(const float* feat, const float* rt, const float* planes, const float* view_weight,
                                      const float* w1pk, float* c1, int B, int S, int C, int D, int h, int w, int precision,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  ADAMVS_CHECK_ARG(precision == PRECISION_FP32 || precision == PRECISION_BF16X3, "aggregate_conv1: precision=%d (0 fp32, 1 bf16x3)", precision);
  ADAMVS_CHECK_ARG(feat && rt && planes && view_weight && w1pk && c1 && B > 0 && S > 0 && D > 0 && h > 1 && w > 1,
                   "aggregate_conv1: bad arguments");
  ADAMVS_CHECK_ARG(C == 8 || C == 16 || C == 32, "aggregate_conv1: C=%d unsupported (8, 16 or 32)", C);
  ADAMVS_CHECK_ARG(workspace && workspace_bytes >= sweep_workspace_floats(B, C, D, h, w) * sizeof(float),
                   "aggregate_conv1: workspace too small (%zu < %zu bytes)", workspace_bytes,
                   sweep_workspace_floats(B, C, D, h, w) * sizeof(float));
  return launch_sweep_conv1(feat, rt, explicit_planes(planes), view_weight, w1pk, c1, (float*)workspace, B, S, C, D, h, w, precision, 0,
                            (hipStream_t)stream);
}
