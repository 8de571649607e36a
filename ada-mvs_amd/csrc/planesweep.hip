// Plane-sweep kernels (gfx950).
//
//   k_pair_similarity   pass A of InferDepthNet0.forward, reference
//                       models/adamvs.py:464-478: per source view, per
//                       hypothesis d: sim[d] = mean_c(ref[c] * warp_d(src)[c])
//   k_aggregate_conv1   pass B, reference models/adamvs.py:495-512 + the first
//                       layer of SliceCostRegNetRED (adamvs.py:416):
//                       sim[c] = sum_v w_v warp_v[c] ref[c] / (1e-5 + sum_v w_v),
//                       c1 = ReLU(conv1(sim)); evaluated for EVERY hypothesis
//                       of the stage in one launch (it does not depend on the
//                       recurrent state), sim never leaves the CU.
//
// Feature maps are channel-last [view][B][h*w][C]; a bilinear tap of one
// pixel is one contiguous C*4-byte line, fetched by C/4 neighbouring lanes as
// one float4 each; the per-pixel channel reduction is a wavefront shuffle tree.
#include <limits.h>

#include "common.h"
#include "kernels.h"
#include "conv_frag.h"
#include "warp_math.h"

namespace adamvs {

// ---------------------------------------------------------------------------
// grid: (pixel groups, S, B); block 256.  sim [S][B][hw][D] (channel-last in d).
template <int C>
__global__ __launch_bounds__(256) void k_pair_similarity(const float* __restrict__ feat, const float* __restrict__ rt,
                                                         const float* __restrict__ planes, float* __restrict__ sim,
                                                         int B, int S, int D, int h, int w) {
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
  const float* pl = planes + (size_t)b * D * hw + pc;
  float* out = sim + (((size_t)s * B + b) * hw + pc) * D;
  // the 4 bilinear taps stay in registers while consecutive planes fall into the same source cell
  int cx = INT_MIN, cy = INT_MIN;
  f32x4 t00 = {0.f, 0.f, 0.f, 0.f}, t01 = t00, t10 = t00, t11 = t00;
  float depth = pl[0];
  for (int d0 = 0; d0 < D; d0 += G) {
    float keep = 0.f;
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int d = d0 + j;                                        // planes past D-1 repeat the last one, never stored
      float depth_next = pl[(size_t)min(d + 1, D - 1) * hw];
      float X0 = ax * depth + tx, X1 = ay * depth + ty, X2 = az * depth + tz;
      float u = X0 / X2, v = X1 / X2;
      f32x4 wrp = {0.f, 0.f, 0.f, 0.f};
      if (u > -1.0f && u < (float)w && v > -1.0f && v < (float)h) {
        float fx0 = floorf(u), fy0 = floorf(v);
        int ix = (int)fx0, iy = (int)fy0;
        if (ix != cx || iy != cy) {
          cx = ix; cy = iy;
          int xa = max(ix, 0), xb = min(ix + 1, w - 1), ya = max(iy, 0), yb = min(iy + 1, h - 1);
          t00 = *(const f32x4*)(src + ((size_t)ya * w + xa) * C);
          t01 = *(const f32x4*)(src + ((size_t)ya * w + xb) * C);
          t10 = *(const f32x4*)(src + ((size_t)yb * w + xa) * C);
          t11 = *(const f32x4*)(src + ((size_t)yb * w + xb) * C);
        }
        float lx = u - fx0, ly = v - fy0;
        bool vx0 = ix >= 0, vx1 = ix + 1 <= w - 1, vy0 = iy >= 0, vy1 = iy + 1 <= h - 1;
        float w00 = (vy0 && vx0) ? (1.f - lx) * (1.f - ly) : 0.f;
        float w01 = (vy0 && vx1) ? lx * (1.f - ly) : 0.f;
        float w10 = (vy1 && vx0) ? (1.f - lx) * ly : 0.f;
        float w11 = (vy1 && vx1) ? lx * ly : 0.f;
        wrp = t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11;
      }
      f32x4 m = wrp * ref4;
      float part = (m.x + m.y) + (m.z + m.w);
#pragma unroll
      for (int o = 1; o < G; o <<= 1) part += __shfl_xor(part, o, 64);
      if (j == g) keep = part * (1.0f / (float)C);
      depth = depth_next;
    }
    if (live && d0 + g < D) out[d0 + g] = keep;       // G lanes x 4 B contiguous per pixel
  }
}

// ---------------------------------------------------------------------------
// grid: (ceil(w/32), ceil(h/TR), B*D); block 256.
// c1 [D][B][hw][8]; view weights vw [S][B][hw]; w1pk = conv1 A-fragments [1][9][C/4][64].
template <int C, int TR>
__global__ __launch_bounds__(256) void k_aggregate_conv1(const float* __restrict__ feat, const float* __restrict__ rt,
                                                         const float* __restrict__ planes, const float* __restrict__ vw,
                                                         const float* __restrict__ w1pk, float* __restrict__ c1,
                                                         int B, int S, int D, int h, int w) {
  constexpr int G = C / 4, KC = C / 4;
  constexpr int LR = TR + 2, LC = 34;
  constexpr int PLANE = plane_pitch16(LR * LC);
  extern __shared__ float lds[];      // [C][PLANE]
  const int hw = h * w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z / D, d = blockIdx.z % D;
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * TR;

  float wf[1][9][KC];
  load_wfrag<1, KC>(wf, w1pk, lane);

  const float* refb = feat + (size_t)b * hw * C;
  const float* pl = planes + ((size_t)b * D + d) * hw;
  // UI items x US views are gathered together: UI*US*4 independent 16-byte taps in flight per lane
  constexpr int UI = 2, US = 2, NITEMS = LR * LC * G;
  for (int base = tid; base < NITEMS; base += 256 * UI) {
    int pix[UI], lofs[UI], g4v[UI];
    bool inimg[UI], live[UI];
    float depth[UI], wsum[UI], fx[UI], fy[UI];
    f32x4 ref4[UI], acc[UI];
#pragma unroll
    for (int u = 0; u < UI; ++u) {
      int i = base + u * 256;
      live[u] = i < NITEMS;
      int ic = live[u] ? i : 0;
      int g = ic % G, pp = ic / G;
      int ry = pp / LC, rx = pp % LC;
      int y = y0 - 1 + ry, x = x0 - 1 + rx;
      inimg[u] = live[u] && y >= 0 && y < h && x >= 0 && x < w;
      int yc = min(max(y, 0), h - 1), xc = min(max(x, 0), w - 1);
      pix[u] = yc * w + xc;
      fx[u] = (float)xc; fy[u] = (float)yc;
      lofs[u] = (4 * g) * PLANE + ry * LC + rx;
      g4v[u] = 4 * g;
      depth[u] = pl[pix[u]];
      ref4[u] = *(const f32x4*)(refb + (size_t)pix[u] * C + 4 * g);
      acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      wsum[u] = 1e-5f;                                               // adamvs.py:497
    }
    for (int s0 = 0; s0 < S; s0 += US) {
      f32x4 t00[UI][US], t01[UI][US], t10[UI][US], t11[UI][US];
      WarpTaps tp[UI][US];
      float wv[UI][US];
#pragma unroll
      for (int u = 0; u < UI; ++u)
#pragma unroll
        for (int v = 0; v < US; ++v) {
          int s = min(s0 + v, S - 1);
          float wgt = vw[((size_t)s * B + b) * hw + pix[u]];
          wv[u][v] = (s0 + v < S) ? wgt : 0.f;
          tp[u][v] = warp_taps(rt + ((size_t)b * S + s) * 12, fx[u], fy[u], depth[u], h, w);
          const float* src = feat + ((size_t)(s + 1) * B + b) * (size_t)hw * C + g4v[u];
          t00[u][v] = *(const f32x4*)(src + (size_t)tp[u][v].o00 * C);
          t01[u][v] = *(const f32x4*)(src + (size_t)tp[u][v].o01 * C);
          t10[u][v] = *(const f32x4*)(src + (size_t)tp[u][v].o10 * C);
          t11[u][v] = *(const f32x4*)(src + (size_t)tp[u][v].o11 * C);
        }
#pragma unroll
      for (int u = 0; u < UI; ++u)
#pragma unroll
        for (int v = 0; v < US; ++v) {
          f32x4 wrp = t00[u][v] * tp[u][v].w00 + t01[u][v] * tp[u][v].w01 + t10[u][v] * tp[u][v].w10 + t11[u][v] * tp[u][v].w11;
          acc[u] += (wrp * ref4[u]) * wv[u][v];                        // adamvs.py:504-508
          wsum[u] += wv[u][v];
        }
    }
#pragma unroll
    for (int u = 0; u < UI; ++u) {
      if (live[u]) {
        f32x4 simv = inimg[u] ? acc[u] / wsum[u] : f32x4{0.f, 0.f, 0.f, 0.f};   // adamvs.py:512; zero padding outside
        float* dl = lds + lofs[u];
        dl[0] = simv.x; dl[PLANE] = simv.y; dl[2 * PLANE] = simv.z; dl[3 * PLANE] = simv.w;
      }
    }
  }
  __syncthreads();

  const int p = lane & 15, q = lane >> 4;
  const float* xb = lds + q * PLANE + p;
  float* c1b = c1 + ((size_t)d * B + b) * (size_t)hw * 8;
  for (int run = wave; run < TR * 2; run += 4) {
    int row = run >> 1, col = (run & 1) * 16;
    f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    conv3x3_run<1, KC, 1, PLANE, LC>(acc, wf, xb, row, col);
    int y = y0 + row, x = x0 + col + p;
    if (q < 2 && y < h && x < w) {
      f32x4 o = acc[0];
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
      *(f32x4*)(c1b + ((size_t)y * w + x) * 8 + 4 * q) = o;
    }
  }
}

}  // namespace adamvs

using namespace adamvs;

extern "C" int adamvs_pair_similarity(const float* feat, const float* rt, const float* planes, float* sim, int B, int S,
                                      int C, int D, int h, int w, void* stream) {
  ADAMVS_CHECK_ARG(feat && rt && planes && sim && B > 0 && S > 0 && D > 0 && h > 1 && w > 1,
                   "pair_similarity: bad arguments (B=%d S=%d D=%d h=%d w=%d)", B, S, D, h, w);
  ADAMVS_CHECK_ARG(C == 8 || C == 16 || C == 32, "pair_similarity: C=%d unsupported (8, 16 or 32)", C);
  int hw = h * w;
  hipStream_t st = (hipStream_t)stream;
  if (C == 32)
    hipLaunchKernelGGL((k_pair_similarity<32>), dim3(cdiv(hw, 32), S, B), dim3(256), 0, st, feat, rt, planes, sim, B, S, D, h, w);
  else if (C == 16)
    hipLaunchKernelGGL((k_pair_similarity<16>), dim3(cdiv(hw, 64), S, B), dim3(256), 0, st, feat, rt, planes, sim, B, S, D, h, w);
  else
    hipLaunchKernelGGL((k_pair_similarity<8>), dim3(cdiv(hw, 128), S, B), dim3(256), 0, st, feat, rt, planes, sim, B, S, D, h, w);
  ADAMVS_CHECK_LAUNCH("pair_similarity");
  return 0;
}

namespace adamvs {
int launch_aggregate_conv1(const float* feat, const float* rt, const float* planes, const float* vw, const float* w1pk,
                           float* c1, int B, int S, int C, int D, int h, int w, hipStream_t st) {
  constexpr int TR = 8;
  dim3 grid(cdiv(w, 32), cdiv(h, TR), B * D);
  size_t lds = (size_t)C * plane_pitch16((TR + 2) * 34) * sizeof(float);
  if (C == 32)
    hipLaunchKernelGGL((k_aggregate_conv1<32, TR>), grid, dim3(256), lds, st, feat, rt, planes, vw, w1pk, c1, B, S, D, h, w);
  else if (C == 16)
    hipLaunchKernelGGL((k_aggregate_conv1<16, TR>), grid, dim3(256), lds, st, feat, rt, planes, vw, w1pk, c1, B, S, D, h, w);
  else if (C == 8)
    hipLaunchKernelGGL((k_aggregate_conv1<8, TR>), grid, dim3(256), lds, st, feat, rt, planes, vw, w1pk, c1, B, S, D, h, w);
  else
    return set_error(-1, "aggregate_conv1: C=%d unsupported (8, 16 or 32)", C);
  ADAMVS_CHECK_LAUNCH("aggregate_conv1");
  return 0;
}
}  // namespace adamvs

extern "C" size_t adamvs_aggregate_conv1_workspace_bytes(int B, int C, int D, int h, int w) {
  return sweep_workspace_floats(B, C, D, h, w) * sizeof(float);
}

extern "C" int adamvs_aggregate_conv1(const float* feat, const float* rt, const float* planes, const float* view_weight,
                                      const float* w1pk, float* c1, int B, int S, int C, int D, int h, int w, int algo,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  ADAMVS_CHECK_ARG(feat && rt && planes && view_weight && w1pk && c1 && B > 0 && S > 0 && D > 0 && h > 1 && w > 1,
                   "aggregate_conv1: bad arguments");
  ADAMVS_CHECK_ARG((size_t)B * D <= 65535, "aggregate_conv1: B*D=%d exceeds the grid z limit", B * D);
  if (algo == 0) {
    ADAMVS_CHECK_ARG(workspace && workspace_bytes >= sweep_workspace_floats(B, C, D, h, w) * sizeof(float),
                     "aggregate_conv1: workspace too small (%zu < %zu bytes)", workspace_bytes,
                     sweep_workspace_floats(B, C, D, h, w) * sizeof(float));
    return launch_sweep_conv1(feat, rt, planes, view_weight, w1pk, c1, (float*)workspace, B, S, C, D, h, w, (hipStream_t)stream);
  }
  return launch_aggregate_conv1(feat, rt, planes, view_weight, w1pk, c1, B, S, C, D, h, w, (hipStream_t)stream);
}
