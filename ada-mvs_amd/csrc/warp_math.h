// Plane-induced homography + bilinear tap set, shared by every kernel that
// warps source features (reference models/module.py:549-566).
//
//   X = (R.[x,y,1]^T) . d + t ;  (u,v) = (X0/X2, X1/X2)
//   sample at pixel coordinates (u,v): grid_sample(bilinear, padding zeros,
//   align_corners=True) applied to the reference's normalised grid lands on
//   exactly these pixel coordinates.  Taps outside [0,w-1]x[0,h-1] add 0.
//
// The reference has no guard for X2 <= 0 (module.py:553).  X2 < 0 (a plane behind the source camera) gives finite,
// mirrored coordinates, sampled like any others -- here too.  X2 == 0 gives inf (0/0: NaN) coordinates, and ATen's
// grid_sample then returns NaN in EVERY channel of that pixel: floor(inf) - inf = NaN bilinear weights times the zero it
// substitutes for a masked tap (pinned by tests/golden/op_warp_behind.npz, a run of the reference).  The tap sets below
// reproduce that: a non-finite coordinate poisons all four weights with NaN; finite coordinates outside the image keep 0.
#pragma once
#include "common.h"

namespace adamvs {

//
// WHICH backend of the reference is the contract: its CPU path (BASELINE.json: "maps match the reference CPU path"; every
// fixture is a CPU run).  ATen's CUDA grid_sample differs exactly here -- it skips out-of-bounds taps, so +-inf coordinates give 0
// and only NaN coordinates give NaN -- i.e. a user comparing against a GPU run of the reference sees 0 where this library (and the
// reference on CPU) returns NaN, for pixels exactly on a source camera's focal plane.  Either way the tile is garbage downstream.

// 0 when u and v are finite, NaN when either is inf or NaN (0 * inf = NaN); two instructions in the plane loops of the sweeps.
// The arithmetic form needs IEEE semantics for 0 * x: the library is never built with -ffast-math / -ffinite-math-only
// (ada-mvs_amd/build.py FLAGS), and a build that tries does not compile.
#if defined(__FAST_MATH__) || (defined(__FINITE_MATH_ONLY__) && __FINITE_MATH_ONLY__)
#error "warp_math.h: nonfinite_poison() relies on 0 * inf = NaN; do not build with -ffast-math / -ffinite-math-only"
#endif
__device__ __forceinline__ float nonfinite_poison(float u, float v) { return fmaf(0.f, u, 0.f * v); }

struct WarpTaps {
  float w00, w01, w10, w11;   // weights of (y0,x0) (y0,x1) (y1,x0) (y1,x1); 0 when the tap is out of range
  int o00, o01, o10, o11;     // pixel offsets y*w+x (0 when out of range)
};

__device__ __forceinline__ WarpTaps warp_taps(const float* __restrict__ rt, float x, float y, float d, int h, int w) {
  // rot_xyz = R.[x,y,1]; rot_depth_xyz = rot_xyz * d; proj_xyz = + t   (module.py:549-552)
  float a0 = rt[0] * x + rt[1] * y + rt[2];
  float a1 = rt[3] * x + rt[4] * y + rt[5];
  float a2 = rt[6] * x + rt[7] * y + rt[8];
  float X0 = a0 * d + rt[9];
  float X1 = a1 * d + rt[10];
  float X2 = a2 * d + rt[11];
  float u = X0 / X2;
  float v = X1 / X2;
  WarpTaps t;
  t.w00 = t.w01 = t.w10 = t.w11 = nonfinite_poison(u, v);
  t.o00 = t.o01 = t.o10 = t.o11 = 0;
  // fully outside: every tap is padding (weight 0; NaN for inf / NaN coordinates, as grid_sample returns them)
  if (!(u > -1.0f && u < (float)w && v > -1.0f && v < (float)h)) return t;
  float fx0 = floorf(u), fy0 = floorf(v);
  int x0 = (int)fx0, y0 = (int)fy0;
  float lx = u - fx0, ly = v - fy0;
  bool vx0 = x0 >= 0, vx1 = x0 + 1 <= w - 1;
  bool vy0 = y0 >= 0, vy1 = y0 + 1 <= h - 1;
  if (vy0 && vx0) { t.w00 = (1.f - lx) * (1.f - ly); t.o00 = y0 * w + x0; }
  if (vy0 && vx1) { t.w01 = lx * (1.f - ly);         t.o01 = y0 * w + x0 + 1; }
  if (vy1 && vx0) { t.w10 = (1.f - lx) * ly;         t.o10 = (y0 + 1) * w + x0; }
  if (vy1 && vx1) { t.w11 = lx * ly;                 t.o11 = (y0 + 1) * w + x0 + 1; }
  return t;
}

// 1/x to within ~1 ulp: v_rcp_f32 + one Newton step (3 instructions instead of the ~11 of an IEEE divide;
// the plane sweep does two divisions per view per plane).  |error| of u = X0 * rcp(X2) stays below 2e-5 px
// for coordinates up to a few hundred pixels.
__device__ __forceinline__ float rcp_nr(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.0f), r, r);
}

// Projection of one pixel onto one plane, in the form the register-resident sweeps pass between lanes:
// the source cell packed into one int ((ix+1) | (iy+1) << 16, or -1 when every tap is padding) and the four
// bilinear weights with the per-tap zero padding already folded in.
struct PlaneTaps {
  int cell;
  float w00, w01, w10, w11;
};

__device__ __forceinline__ PlaneTaps plane_taps(float ax, float ay, float az, float tx, float ty, float tz, float depth,
                                                int h, int w) {
  float X0 = ax * depth + tx, X1 = ay * depth + ty, X2 = az * depth + tz;      // module.py:550-552
  float rz = rcp_nr(X2);
  float u = X0 * rz, v = X1 * rz;                                              // module.py:553
  PlaneTaps t;
  t.cell = -1;
  t.w00 = t.w01 = t.w10 = t.w11 = nonfinite_poison(u, v);          // 0; NaN on the source camera's focal plane (X2 == 0)
  if (u > -1.0f && u < (float)w && v > -1.0f && v < (float)h) {
    float fx0 = floorf(u), fy0 = floorf(v);
    int ix = (int)fx0, iy = (int)fy0;
    float lx = u - fx0, ly = v - fy0;
    bool vx0 = ix >= 0, vx1 = ix + 1 <= w - 1, vy0 = iy >= 0, vy1 = iy + 1 <= h - 1;
    t.w00 = (vy0 && vx0) ? (1.f - lx) * (1.f - ly) : 0.f;
    t.w01 = (vy0 && vx1) ? lx * (1.f - ly) : 0.f;
    t.w10 = (vy1 && vx0) ? (1.f - lx) * ly : 0.f;
    t.w11 = (vy1 && vx1) ? lx * ly : 0.f;
    t.cell = (ix + 1) | ((iy + 1) << 16);
  }
  return t;
}

// The same projection without branches, with a scale folded into the weights (the sweep folds the normalised view weight
// in here): every quantity is computed for every lane and masked at the end -- in a rolled plane loop the two nested
// exec-masked regions of plane_taps cost a dozen register initialisations per plane.  Inside the image band ix lies in
// [-1, w-1] and iy in [-1, h-1], so tap x0 is padding only for ix == -1 and tap x1 only for ix == w-1 (likewise y).
__device__ __forceinline__ PlaneTaps plane_taps_scaled(float ax, float ay, float az, float tx, float ty, float tz, float depth,
                                                       int h, int w, float scale) {
  const float X0 = ax * depth + tx, X1 = ay * depth + ty, X2 = az * depth + tz;      // module.py:550-552
  const float rz = rcp_nr(X2);
  const float u = X0 * rz, v = X1 * rz;                                              // module.py:553
  const bool inside = u > -1.0f && u < (float)w && v > -1.0f && v < (float)h;        // false for NaN / inf as well
  const float fx0 = floorf(u), fy0 = floorf(v);
  const int ix = (int)fx0, iy = (int)fy0;
  const float lx = u - fx0, ly = v - fy0;
  // outside the image: x weights 0 (finite y weights times 0); inf / NaN coordinates: NaN in all four products, whatever
  // the y weights came out as (see the header of this file)
  const float pz = nonfinite_poison(u, v);
  const float wx0 = (inside && ix >= 0) ? 1.f - lx : pz, wx1 = (inside && ix < w - 1) ? lx : pz;
  const float wy0 = (iy >= 0 ? 1.f - ly : 0.f) * scale, wy1 = (iy < h - 1 ? ly : 0.f) * scale;
  PlaneTaps t;
  t.w00 = wx0 * wy0; t.w01 = wx1 * wy0; t.w10 = wx0 * wy1; t.w11 = wx1 * wy1;
  t.cell = inside ? ((ix + 1) | ((iy + 1) << 16)) : -1;
  return t;
}

// the four taps of a packed cell (clamped to the image; padding taps carry weight 0)
__device__ __forceinline__ void load_cell_taps(const float* __restrict__ src, int C, int cell, int h, int w, f32x4& t00,
                                               f32x4& t01, f32x4& t10, f32x4& t11) {
  int ix = (cell & 0xFFFF) - 1, iy = (cell >> 16) - 1;
  int xa = max(ix, 0), xb = min(ix + 1, w - 1), ya = max(iy, 0), yb = min(iy + 1, h - 1);
  t00 = *(const f32x4*)(src + ((size_t)ya * w + xa) * C);
  t01 = *(const f32x4*)(src + ((size_t)ya * w + xb) * C);
  t10 = *(const f32x4*)(src + ((size_t)yb * w + xa) * C);
  t11 = *(const f32x4*)(src + ((size_t)yb * w + xb) * C);
}

// 4 channels of a channel-last feature map [hw][C] gathered with a tap set.
__device__ __forceinline__ f32x4 gather4(const float* __restrict__ fea, int C, int c0, const WarpTaps& t) {
  const float* base = fea + c0;
  f32x4 a = *(const f32x4*)(base + (size_t)t.o00 * C);
  f32x4 b = *(const f32x4*)(base + (size_t)t.o01 * C);
  f32x4 c = *(const f32x4*)(base + (size_t)t.o10 * C);
  f32x4 d = *(const f32x4*)(base + (size_t)t.o11 * C);
  return a * t.w00 + b * t.w01 + c * t.w10 + d * t.w11;
}

}  // namespace adamvs
