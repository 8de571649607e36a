// Softmax over D in the epilogue of CostRegNet2D's last layer (reference models/adamvs.py:481-486: softmax of the scores,
// its maximum = the view weight, depth_regression = the pair depth), shared by the fp32 and the split-bf16 kernels: both
// leave a lane with channels 16 (wm MT + mt) + 4 q .. + 3 of pixel (row wn NTR + r, column p) of a BR x 16 block.
#pragma once
#include "common.h"
#include "planes.h"

namespace adamvs {

// A lane reduces the 4 MT scores it holds per row to (max, sum of exp, sum of exp * depth); the 4 WM partials of a pixel
// meet in LDS ([BR][16][4 WM][3] floats at `part`, which must not be in use: barrier before the call) and one thread per
// pixel merges them (the online-softmax merge): view weight = 1 / sum, pair depth = weighted sum / sum.  The score
// volume is never stored.  image n belongs to tile n % B (planes are per tile).
template <int MT, int WM, int NTR, int BR>
__device__ __forceinline__ void softmax_epilogue(const f32x4 (&acc)[MT][NTR], const float* __restrict__ bias, const PlaneSrc& planes,
                                                 int B, int n, int r0, int c0, int ho, int wo, int D, float* __restrict__ vw,
                                                 float* __restrict__ pd, float* part) {
  // D = the number of hypothesis planes (<= the channels the lanes hold: pad channels carry a score of -1e30)
  constexpr int NPART = 4 * WM, WN = 4 / WM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int p = lane & 15, q = lane >> 4;
  const int b = n % B;
  const size_t hw = (size_t)ho * wo;
  static_assert(NTR * WN == BR, "rows of the block");
#pragma unroll
  for (int r = 0; r < NTR; ++r) {
    const int row = wn * NTR + r;
    const int oy = min(r0 + row, ho - 1), ox = min(c0 + p, wo - 1);
    const PlaneLine pl = plane_line(planes, b, (size_t)oy * wo + ox, D, hw);
    f32x4 v[MT];
    float m = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      v[mt] = acc[mt][r] + *(const f32x4*)(bias + (wm * MT + mt) * 16 + 4 * q);
      m = fmaxf(m, fmaxf(fmaxf(v[mt].x, v[mt].y), fmaxf(v[mt].z, v[mt].w)));
    }
    float se = 0.f, sd = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int d = (wm * MT + mt) * 16 + 4 * q;
      const float e0 = __expf(v[mt].x - m), e1 = __expf(v[mt].y - m), e2 = __expf(v[mt].z - m), e3 = __expf(v[mt].w - m);
      se += (e0 + e1) + (e2 + e3);
      // explicit planes: four loads; generated: a multiply and an add each
      sd = __fmaf_rn(e3, plane_at_pad(planes, pl, d + 3, hw), __fmaf_rn(e2, plane_at_pad(planes, pl, d + 2, hw),
           __fmaf_rn(e1, plane_at_pad(planes, pl, d + 1, hw), __fmaf_rn(e0, plane_at_pad(planes, pl, d, hw), sd))));
    }
    float* o = part + ((row * 16 + p) * NPART + wm * 4 + q) * 3;
    o[0] = m; o[1] = se; o[2] = sd;
  }
  __syncthreads();
  if (tid < BR * 16) {
    const int row = tid >> 4, col = tid & 15;
    const float* pp = part + (row * 16 + col) * NPART * 3;
    float M = -INFINITY;
#pragma unroll
    for (int j = 0; j < NPART; ++j) M = fmaxf(M, pp[3 * j]);
    float Z = 0.f, P = 0.f;
#pragma unroll
    for (int j = 0; j < NPART; ++j) {
      const float sc = __expf(pp[3 * j] - M);
      Z = __fmaf_rn(pp[3 * j + 1], sc, Z);
      P = __fmaf_rn(pp[3 * j + 2], sc, P);
    }
    const int oy = r0 + row, ox = c0 + col;
    if (oy < ho && ox < wo) {
      const size_t opix = ((size_t)n * ho + oy) * wo + ox;
      vw[opix] = 1.0f / Z;                      // max_d softmax = exp(max - max) / sum
      pd[opix] = P / Z;
    }
  }
}

}  // namespace adamvs
