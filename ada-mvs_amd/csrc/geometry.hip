// Geometry / resampling kernels of the Ada-MVS depth-inference path (gfx950).
//
//   relative transforms   reference models/module.py:539-541  (T = P_src . P_ref^-1)
//   homo_warp (NCHW op)   reference models/module.py:527-568  (homo_warping_float)
//   depth-range samples   reference models/module.py:628-663
//   2x bilinear upsample  reference models/adamvs.py:505, 522 (F.interpolate, align_corners=False)
//   depth_regression      reference models/module.py:617-625
//   NCHW -> channel-last feature packing (layout the plane-sweep kernels gather from)
#include "common.h"
#include "planes.h"
#include "warp_math.h"

namespace adamvs {

// ---------------------------------------------------------------------------
// T = P_src . P_ref^-1, one thread per (batch, source view); fp64 inside.
// out[b][s] = { R row-major (9), t (3) }.
__global__ void k_relative_transforms(const float* __restrict__ proj, float* __restrict__ rt, int B, int V) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int S = V - 1;
  if (idx >= B * S) return;
  int b = idx / S, s = idx % S;
  const float* pr = proj + ((size_t)b * V) * 16;
  const float* ps = proj + ((size_t)b * V + s + 1) * 16;
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { a[i][j] = pr[i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < 4; ++c) {           // Gauss-Jordan, partial pivoting
    int piv = c; double best = fabs(a[c][c]);
    for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
    if (piv != c) for (int j = 0; j < 8; ++j) { double tmp = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = tmp; }
    double inv = 1.0 / a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] *= inv;
    for (int r = 0; r < 4; ++r) if (r != c) {
      double f = a[r][c];
      for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
    }
  }
  float* o = rt + (size_t)idx * 12;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc += (double)ps[i * 4 + k] * a[k][4 + j];
      if (j < 3) o[i * 3 + j] = (float)acc; else o[9 + i] = (float)acc;
    }
  }
}

// ---------------------------------------------------------------------------
// [B][C][hw] -> [B][hw][C]; C % 4 == 0. One thread = one pixel x 4 channels.
__global__ void k_pack_nhwc(const float* __restrict__ in, float* __restrict__ out, int C, int hw, size_t total4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  int C4 = C >> 2;
  int g = (int)(i % C4);
  size_t bp = i / C4;            // b*hw + p
  size_t b = bp / hw, p = bp % hw;
  const float* src = in + (b * C + 4 * g) * hw + p;
  f32x4 v = {src[0], src[(size_t)hw], src[2 * (size_t)hw], src[3 * (size_t)hw]};
  *(f32x4*)(out + bp * C + 4 * g) = v;
}

// [B][hw][C] -> [B][C][hw]
__global__ void k_unpack_nchw(const float* __restrict__ in, float* __restrict__ out, int C, int hw, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  size_t p = i % hw;
  size_t bc = i / hw;
  size_t b = bc / C, c = bc % C;
  out[i] = in[(b * hw + p) * C + c];
}

// ---------------------------------------------------------------------------
// Hypothesis planes.  uniform: d_k = min + k (max-min)/(D-1) from depth_values[b] = {min,max}
// (module.py:650-658).  window: lo = cur - D/2 I, hi = cur + D/2 I, d_k = lo + k (hi-lo)/(D-1)
// (module.py:632-641), no clamping.
__global__ void k_depth_samples_uniform(const float* __restrict__ dv, float* __restrict__ out, int D, int hw, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  size_t bd = i / hw;
  int b = (int)(bd / D), d = (int)(bd % D);
  float dmin = dv[2 * b], dmax = dv[2 * b + 1];
  float step = (dmax - dmin) / (float)(D - 1);
  out[i] = plane_value(dmin, step, d);                    // module.py:651-656: a rounded product, then a rounded sum
}

__global__ void k_depth_samples_window(const float* __restrict__ cur, float* __restrict__ out, float half_span,
                                       int D, int hw, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  size_t p = i % hw;
  size_t bd = i / hw;
  int b = (int)(bd / D), d = (int)(bd % D);
  float c = cur[(size_t)b * hw + p];
  float lo = c - half_span, hi = c + half_span;
  float step = (hi - lo) / (float)(D - 1);
  out[i] = plane_value(lo, step, d);                      // module.py:632-641
}

// ---------------------------------------------------------------------------
// Generic bilinear resize, align_corners=False (ATen area_pixel_compute_source_index).
__device__ __forceinline__ void resize_taps(int dst, int n_in, float scale, int& i0, int& i1, float& l1) {
  float s = ((float)dst + 0.5f) * scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__global__ void k_resize_bilinear(const float* __restrict__ in, float* __restrict__ out, int hi, int wi, int ho, int wo,
                                  size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int x = (int)(i % wo);
  int y = (int)((i / wo) % ho);
  size_t n = i / ((size_t)wo * ho);
  int y0, y1, x0, x1; float ly, lx;
  resize_taps(y, hi, (float)hi / (float)ho, y0, y1, ly);
  resize_taps(x, wi, (float)wi / (float)wo, x0, x1, lx);
  const float* p = in + n * (size_t)hi * wi;
  float top = p[y0 * wi + x0] * (1.f - lx) + p[y0 * wi + x1] * lx;
  float bot = p[y1 * wi + x0] * (1.f - lx) + p[y1 * wi + x1] * lx;
  out[i] = top * (1.f - ly) + bot * ly;
}

// ---------------------------------------------------------------------------
// depth_regression: out[b][p] = sum_d prob[b][d][p] * depth(b,d,p).
// mode 0: depth_values [B][D]; mode 1: [B][D][hd][wd], bilinearly resized to [h][w].
__global__ void k_depth_regression(const float* __restrict__ prob, const float* __restrict__ dv, float* __restrict__ out,
                                   int mode, int D, int h, int w, int hd, int wd, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int x = (int)(i % w), y = (int)((i / w) % h);
  size_t b = i / ((size_t)w * h);
  size_t hw = (size_t)h * w;
  const float* pp = prob + b * D * hw + (size_t)y * w + x;
  float acc = 0.f;
  if (mode == 0) {
    for (int d = 0; d < D; ++d) acc += pp[d * hw] * dv[b * D + d];
  } else {
    int y0, y1, x0, x1; float ly, lx;
    resize_taps(y, hd, (float)hd / (float)h, y0, y1, ly);
    resize_taps(x, wd, (float)wd / (float)w, x0, x1, lx);
    size_t hwd = (size_t)hd * wd;
    for (int d = 0; d < D; ++d) {
      const float* q = dv + (b * D + d) * hwd;
      float top = q[y0 * wd + x0] * (1.f - lx) + q[y0 * wd + x1] * lx;
      float bot = q[y1 * wd + x0] * (1.f - lx) + q[y1 * wd + x1] * lx;
      acc += pp[d * hw] * (top * (1.f - ly) + bot * ly);
    }
  }
  out[i] = acc;
}

// ---------------------------------------------------------------------------
// homo_warping_float as a standalone op on the reference's own layouts:
// src [B][C][h][w], depth [B][Nd][h][w] -> out [B][C][Nd][h][w].  One thread per
// (b, d, pixel), loop over channels (taps of neighbouring lanes are neighbouring
// addresses of one channel plane).
__global__ void k_homo_warp_nchw(const float* __restrict__ src, const float* __restrict__ rt, const float* __restrict__ depth,
                                 float* __restrict__ out, int C, int Nd, int h, int w, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int x = (int)(i % w), y = (int)((i / w) % h);
  size_t bd = i / ((size_t)w * h);
  int d = (int)(bd % Nd);
  size_t b = bd / Nd;
  size_t hw = (size_t)h * w;
  WarpTaps tp = warp_taps(rt + b * 12, (float)x, (float)y, depth[i], h, w);
  const float* s = src + b * C * hw;
  float* o = out + (b * C * Nd + d) * hw + (size_t)y * w + x;
  for (int c = 0; c < C; ++c) {
    const float* sc = s + c * hw;
    float v = 0.f;
    if (tp.w00 != 0.f) v += tp.w00 * sc[tp.o00];
    if (tp.w01 != 0.f) v += tp.w01 * sc[tp.o01];
    if (tp.w10 != 0.f) v += tp.w10 * sc[tp.o10];
    if (tp.w11 != 0.f) v += tp.w11 * sc[tp.o11];
    o[(size_t)c * Nd * hw] = v;
  }
}

}  // namespace adamvs

// ===========================================================================
using namespace adamvs;

extern "C" int adamvs_relative_transforms(const float* proj, float* rt, int B, int V, void* stream) {
  ADAMVS_CHECK_ARG(proj && rt && B > 0 && V > 1, "relative_transforms: bad arguments (B=%d V=%d)", B, V);
  int n = B * (V - 1);
  hipLaunchKernelGGL(k_relative_transforms, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, proj, rt, B, V);
  ADAMVS_CHECK_LAUNCH("relative_transforms");
  return 0;
}

extern "C" int adamvs_pack_features(const float* nchw, float* nhwc, int B, int C, int h, int w, void* stream) {
  ADAMVS_CHECK_ARG(nchw && nhwc && B > 0 && C > 0 && (C % 4) == 0 && h > 0 && w > 0,
                   "pack_features: bad arguments (B=%d C=%d h=%d w=%d; C must be a multiple of 4)", B, C, h, w);
  size_t total4 = (size_t)B * h * w * (C / 4);
  hipLaunchKernelGGL(k_pack_nhwc, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nchw, nhwc, C,
                     h * w, total4);
  ADAMVS_CHECK_LAUNCH("pack_features");
  return 0;
}

extern "C" int adamvs_unpack_features(const float* nhwc, float* nchw, int B, int C, int h, int w, void* stream) {
  ADAMVS_CHECK_ARG(nchw && nhwc && B > 0 && C > 0 && h > 0 && w > 0, "unpack_features: bad arguments");
  size_t total = (size_t)B * h * w * C;
  hipLaunchKernelGGL(k_unpack_nchw, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nhwc, nchw, C,
                     h * w, total);
  ADAMVS_CHECK_LAUNCH("unpack_features");
  return 0;
}

extern "C" int adamvs_depth_range_samples_uniform(const float* depth_values, float* out, int B, int D, int h, int w,
                                                  void* stream) {
  ADAMVS_CHECK_ARG(depth_values && out && B > 0 && D > 1 && h > 0 && w > 0, "depth_range_samples_uniform: bad arguments (D=%d)", D);
  size_t total = (size_t)B * D * h * w;
  hipLaunchKernelGGL(k_depth_samples_uniform, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     depth_values, out, D, h * w, total);
  ADAMVS_CHECK_LAUNCH("depth_range_samples_uniform");
  return 0;
}

extern "C" int adamvs_depth_range_samples_window(const float* cur_depth, double depth_interval_pixel, float* out, int B, int D,
                                                 int h, int w, void* stream) {
  ADAMVS_CHECK_ARG(cur_depth && out && B > 0 && D > 1 && h > 0 && w > 0, "depth_range_samples_window: bad arguments (D=%d)", D);
  size_t total = (size_t)B * D * h * w;
  // reference: ndepth / 2 * depth_inteval_pixel, Python float arithmetic (module.py:632)
  float half_span = (float)((double)D / 2.0 * depth_interval_pixel);
  hipLaunchKernelGGL(k_depth_samples_window, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     cur_depth, out, half_span, D, h * w, total);
  ADAMVS_CHECK_LAUNCH("depth_range_samples_window");
  return 0;
}

extern "C" int adamvs_resize_bilinear(const float* in, float* out, int N, int hi, int wi, int ho, int wo, void* stream) {
  ADAMVS_CHECK_ARG(in && out && N > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0, "resize_bilinear: bad arguments");
  size_t total = (size_t)N * ho * wo;
  hipLaunchKernelGGL(k_resize_bilinear, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, hi,
                     wi, ho, wo, total);
  ADAMVS_CHECK_LAUNCH("resize_bilinear");
  return 0;
}

extern "C" int adamvs_depth_regression(const float* prob, const float* depth_values, float* out, int B, int D, int h, int w,
                                       int hd, int wd, void* stream) {
  ADAMVS_CHECK_ARG(prob && depth_values && out && B > 0 && D > 0 && h > 0 && w > 0 && hd >= 0 && wd >= 0,
                   "depth_regression: bad arguments");
  int mode = (hd > 0 && wd > 0) ? 1 : 0;
  size_t total = (size_t)B * h * w;
  hipLaunchKernelGGL(k_depth_regression, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, prob,
                     depth_values, out, mode, D, h, w, hd, wd, total);
  ADAMVS_CHECK_LAUNCH("depth_regression");
  return 0;
}

extern "C" int adamvs_homo_warp(const float* src_fea, const float* rt, const float* depth_values, float* out, int B, int C,
                                int Nd, int h, int w, void* stream) {
  ADAMVS_CHECK_ARG(src_fea && rt && depth_values && out && B > 0 && C > 0 && Nd > 0 && h > 1 && w > 1,
                   "homo_warp: bad arguments (B=%d C=%d Nd=%d h=%d w=%d)", B, C, Nd, h, w);
  size_t total = (size_t)B * Nd * h * w;
  hipLaunchKernelGGL(k_homo_warp_nchw, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src_fea, rt,
                     depth_values, out, C, Nd, h, w, total);
  ADAMVS_CHECK_LAUNCH("homo_warp");
  return 0;
}
