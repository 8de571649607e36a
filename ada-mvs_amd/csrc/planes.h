// Where the hypothesis planes of a stage come from (reference models/module.py:628-663).
//
// The reference materialises depth_values [B,D,h,w] for every stage; on the fused stage path (adamvs_depth_stage_forward)
// the planes are instead generated where they are used:
//   mode 0  explicit tensor [B][D][hw] (the op-level entry points, InferDepthNet0.forward with caller-made planes)
//   mode 1  uniform, get_depth_range_samples with a 2-D cur_depth (module.py:650-658): p = [B][2] = (min, max),
//           plane d = min + d * ((max - min) / (D - 1))
//   mode 2  window, get_cur_depth_range_samples (module.py:628-643): p = cur_depth [B][hw],
//           lo = cur - half_span, hi = cur + half_span, plane d = lo + d * ((hi - lo) / (D - 1)); half_span from the
//           descriptor or, when span_dev is set, from device memory
// with the reference's fp32 operation order (a rounded product, then a rounded sum: no fused multiply-add), so a
// generated plane equals the materialised one bit for bit.
#pragma once
#include "common.h"

namespace adamvs {

enum { PLANES_EXPLICIT = 0, PLANES_UNIFORM = 1, PLANES_WINDOW = 2 };

struct PlaneSrc {
  const float* p;
  int mode;
  float half_span;
  const float* span_dev;     // window mode: when non-null the half span is read from device memory (one float) instead -- a
                             // captured hipGraph then serves tiles of any depth range (adamvs_stage_desc.half_span_dev)
};
static inline PlaneSrc explicit_planes(const float* planes) { return PlaneSrc{planes, PLANES_EXPLICIT, 0.f, nullptr}; }

// the planes of one (tile b, pixel): a strided line of the tensor, or (lo, step)
struct PlaneLine {
  const float* q;
  float lo, step;
  int last;          // D - 1
};

// HIP's __fmul_rn / __fadd_rn are plain operators and hipcc contracts a * b + c into one fused multiply-add by default
// (-ffp-contract=fast): measured, the "rounded product, rounded sum" came out as the fma.  The pragma is what keeps the
// two roundings of the reference's tensor arithmetic.
__device__ __forceinline__ float plane_value(float lo, float step, int d) {
#pragma clang fp contract(off)
  const float prod = (float)d * step;
  return lo + prod;
}

__device__ __forceinline__ PlaneLine plane_line(const PlaneSrc& s, size_t b, size_t pix, int D, size_t hw) {
  PlaneLine l{nullptr, 0.f, 0.f, D - 1};
  if (s.mode == PLANES_EXPLICIT) {
    l.q = s.p + b * D * hw + pix;
  } else {
    float lo, hi;
    if (s.mode == PLANES_UNIFORM) {
      lo = s.p[2 * b];
      hi = s.p[2 * b + 1];
    } else {
      const float c = s.p[b * hw + pix];
      const float hs = s.span_dev ? *s.span_dev : s.half_span;       // uniform address: a scalar load
      lo = c - hs;
      hi = c + hs;
    }
    l.lo = lo;
    l.step = (hi - lo) / (float)(D - 1);          // single operations: nothing to contract
  }
  return l;
}

__device__ __forceinline__ float plane_at(const PlaneSrc& s, const PlaneLine& l, int d, size_t hw) {
  return s.mode == PLANES_EXPLICIT ? l.q[(size_t)d * hw] : plane_value(l.lo, l.step, d);
}
// For the softmax over CostRegNet2D's channels when the network runs wider than the D hypotheses (costreg_width, pad
// channels score -1e30 and weigh exactly 0): d may run past the last plane; any finite value will do there.
__device__ __forceinline__ float plane_at_pad(const PlaneSrc& s, const PlaneLine& l, int d, size_t hw) {
  return plane_at(s, l, min(d, l.last), hw);
}

}  // namespace adamvs
