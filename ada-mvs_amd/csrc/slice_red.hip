// Recurrent cost regularisation (one hypothesis slice per step) + soft-argmin.
//
//   SliceCostRegNetRED.forward   reference models/adamvs.py:415-424
//   ConvGRUCell.forward          reference models/module.py:24-52
//   online soft-argmin           reference models/adamvs.py:516-531
//
// Every map is channel-last [B][h*w][ch].  The 3x3 convolutions run on the
// fp32 matrix cores (conv_frag.h); gates / candidate / blend are fused into the
// epilogues, the two decoder layers are one kernel.
#include "common.h"
#include "conv_frag.h"
#include "kernels.h"
#include "persistent.h"
#include "slice_roles.h"
#include "slice_roles_wino.h"

namespace adamvs {

// A GRU convolution in the F(2x2, 3x3) form (slice_roles_wino.h), one role per launch
template <int CA, int CB, int NT, int EPI>
__global__ __launch_bounds__(256, 2) void k_conv_small_wino(SmallConvArgs a, TileGrid tg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  ConvWinoRole<CA, CB, NT, EPI>::run(a, tg, TileRange{0, tg.ntiles}, blockIdx.x, gridDim.x, lds);
}

// One role per launch (the one-step op adamvs_slice_reg_step, conv3x3_pair of MS-REDNet); the software-pipelined
// recurrence of a whole stage runs the same roles several per launch (recurrence.hip).
template <int CA, int CB, int NT, int STRIDE, int EPI, int TR, int RW>
__global__ __launch_bounds__(256) void k_conv_small(SmallConvArgs a, TileGrid tg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  ConvSmallRole<CA, CB, NT, STRIDE, EPI, TR, RW>::run(a, tg, TileRange{0, tg.ntiles}, blockIdx.x, gridDim.x, lds);
}
// MS-REDNet's shallow GRU levels with the elementwise kernels folded into the window fill (GruPro, kernels.h)
template <int CA, int CB, int NT>
__global__ __launch_bounds__(256) void k_conv_small_pro(SmallConvArgs a, TileGrid tg, GruPro pro) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  ConvSmallRole<CA, CB, NT, 1, EPI_LINEAR, 4, 1>::run(a, tg, TileRange{0, tg.ntiles}, blockIdx.x, gridDim.x, lds, pro);
}

// DEC_WAVES workgroups per CU the decoder is built for (A/B: -DDEC_WAVES=1 = no register cap, 133 registers = three)
#ifndef DEC_WAVES
#define DEC_WAVES 4
#endif
template <bool IN_UP>
__global__ __launch_bounds__(256, DEC_WAVES) void k_decoder(DecoderArgs a, TileGrid tg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  DecoderRole<IN_UP>::run(a, tg, TileRange{0, tg.ntiles}, blockIdx.x, gridDim.x, lds);
}

__global__ __launch_bounds__(256) void k_cand1_two_row(SmallConvArgs a, TileGrid tg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  Cand1TwoRowRole::run(a, tg, TileRange{0, tg.ntiles}, blockIdx.x, gridDim.x, lds);
}


// ---------------------------------------------------------------------------
// Soft-argmin over the regularised cost slices (adamvs.py:516-531):
// p = exp(cost) (no max subtraction), E = sum p, M = max p (initial 0), A = sum depth_d p,
// depth = A / (E + 1e-10), confidence = M / (E + 1e-10).  When IN_UP the hypothesis plane
// is the 2x bilinear upsample (align_corners=False) of planes[b][d] (adamvs.py:521-522).
template <bool IN_UP>
__global__ void k_soft_argmin(const float* __restrict__ vol, const float* __restrict__ planes, float* __restrict__ depth,
                              float* __restrict__ conf, int D, int h, int w, size_t total) {
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
  const float* v = vol + b * D * HW + (size_t)Y * Wo + X;
  const float* pl = planes + b * D * hw;
  float E = 0.f, M = 0.f, A = 0.f;
  for (int d = 0; d < D; ++d) {
    float pr = __expf(v[(size_t)d * HW]);
    const float* q = pl + (size_t)d * hw;
    float dep;
    if (IN_UP) {
      float top = q[y0 * w + x0] * (1.f - lx) + q[y0 * w + x1] * lx;
      float bot = q[y1 * w + x0] * (1.f - lx) + q[y1 * w + x1] * lx;
      dep = top * (1.f - ly) + bot * ly;
    } else {
      dep = q[y0 * w + x0];
    }
    M = (M < pr) ? pr : M;
    A = dep * pr + A;
    E = E + pr;
  }
  float den = E + 1e-10f;
  depth[i] = A / den;
  conf[i] = M / den;
}

// ---------------------------------------------------------------------------
// host-side launchers shared by the op-level entry point and the stage driver

template <int CA, int CB, int NT, int STRIDE, int EPI, int TR, int RW>
static int launch_small_shape(const SmallConvArgs& a, int B, hipStream_t st, const char* name) {
  constexpr int TC = 16 * RW;
  constexpr int LR = (STRIDE == 1) ? TR + 2 : 2 * TR + 1;
  constexpr int LC = (STRIDE == 1) ? TC + 2 : 2 * TC + 1;
  constexpr int PLANE = (STRIDE == 1) ? plane_pitch16(LR * LC) : ((LR * LC) | 1);
  constexpr size_t lds = (size_t)((CA + CB) / 4) * group_pitch(PLANE, (CA + CB) / 4) * sizeof(float);
  static_assert(lds <= 64 * 1024, "tile exceeds the default dynamic LDS limit");
  auto kern = k_conv_small<CA, CB, NT, STRIDE, EPI, TR, RW>;
  static const int capacity = resident_blocks(kern, 256, lds);      // once per instantiation, thread-safely (magic static)
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(a.wo, TC), cdiv(a.ho, TR), B)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, tg);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "%s: %s", name, hipGetErrorString(e));
  return 0;
}

template <int CA, int CB, int NT, int EPI>
static int launch_small_wino(const SmallConvArgs& a, int B, hipStream_t st, const char* name) {
  typedef ConvWinoRole<CA, CB, NT, EPI> Role;
  constexpr size_t lds = Role::LDS_BYTES;
  auto kern = k_conv_small_wino<CA, CB, NT, EPI>;
  static const int capacity = [&] {                      // once per instantiation, thread-safely (magic static)
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return resident_blocks(kern, 256, lds);
  }();
  TileGrid tg;
  if (int rc = make_tile_grid(tg, Role::tiles_x(a), Role::tiles_y(a), B)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, tg);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "%s: %s", name, hipGetErrorString(e));
  return 0;
}

// option gru_wino: bit mask of the GRU convolutions that run in the F(2x2, 3x3) form when a role has a launch of its own
// (1 gates1, 2 gates2, 4 cand2, 8 cand1; default 7; 0 = the direct kernels, which the pipelined schedules use: their maps
// equal the direct kernels' bit for bit, the F(2x2, 3x3) ones to ~1e-7).
// Measured at cfg2, 128 tiles (recurrence per step, ms): none 79.8; gates1 74.9; gates2 74.2; cand2 78.6; all three 68.3.
int gru_wino_mask() { return opt(OPT_GRU_WINO); }

// Tile = 4 rows x 16 columns: the smallest halo (6 x 18 input pixels for 64 outputs) of the one-run-per-wave shapes.
template <int CA, int CB, int NT, int STRIDE, int EPI>
static int launch_small(const SmallConvArgs& a, int B, hipStream_t st, const char* name) {
  return launch_small_shape<CA, CB, NT, STRIDE, EPI, 4, 1>(a, B, st, name);
}

// conv1 (C -> 8, ReLU; reference adamvs.py:416) in two-row form: the 16 MFMA rows are 8 output channels of
// output row y and the same 8 channels of row y+1, fed by the same input-row fragment (tap ky for the first
// half, ky-1 for the second), so no half of the tile is zero padding: 12 fragment passes per 2 rows instead of 18.
// src [N][hw][C] -> c1 [N][hw][8].  Persistent and pipelined like k_conv_small; tile 8 rows x 16 columns = four
// row pairs, one per wave (12 * C/4 MFMAs each).
template <int C>
__global__ __launch_bounds__(256) void k_conv1_two_row(const float* __restrict__ src, const float* __restrict__ wpk,
                                                       float* __restrict__ c1, int h, int w, TileGrid tg) {
  constexpr int KC = C / 4, G = C / 4, TR = 8, TC = 16, LR = TR + 2, LC = TC + 2, NPIX = LR * LC;
  constexpr int PLANE = plane_pitch16(NPIX), GP = group_pitch(PLANE, G);
  constexpr int NL = (NPIX * G + 255) / 256;
  extern __shared__ float lds[];           // [G][GP]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
  const int p = lane & 15, q = lane >> 4;
  float wf[12][KC];
#pragma unroll
  for (int t = 0; t < 12; ++t)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) wf[t][kc] = wpk[(t * KC + kc) * 64 + lane];

  // ---- per-lane constants
  unsigned goff[NL], lbyte[NL];
  int rc[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int j = min(tid + k * 256, NPIX * G - 1);          // surplus lanes repeat the last item
    const int g = j % G, pp = j / G, r = pp / LC, c = pp % LC;
    goff[k] = (unsigned)(((r * w + c) * C + 4 * g) * 4);
    lbyte[k] = (unsigned)((g * GP + r * LC + c) * 4);
    rc[k] = r | (c << 16);
    pin(goff[k]); pin(lbyte[k]); pin(rc[k]);
  }
  unsigned xbyte[KC];                                         // B-fragment origin of the wave's row pair
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    xbyte[kc] = (unsigned)((kc * GP + q * PLANE + (2 * wave) * LC + p) * 4);
    pin(xbyte[kc]);
  }
  const int orow = 2 * wave + (q >> 1);                       // lane's output pixel (orow, p), channels 4*(q&1)..
  unsigned ooff = (unsigned)(((orow * w + p) * 8 + 4 * (q & 1)) * 4);
  pin(ooff);

  auto load_tile = [&](f32x4 (&stage)[NL], int n, int tx, int ty) {
    const int ix0 = tx * TC - 1, iy0 = ty * TR - 1;
    const buf_rsrc rs = make_rsrc((const char*)src + (((long)n * h + iy0) * w + ix0) * (C * 4));
    if (iy0 >= 0 && ix0 >= 0 && iy0 + LR <= h && ix0 + LC <= w) {
#pragma unroll
      for (int k = 0; k < NL; ++k) stage[k] = buf_load4(rs, goff[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int iy = iy0 + (rc[k] & 0xffff), ix = ix0 + (rc[k] >> 16);
        stage[k] = buf_load4(rs, ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) ? goff[k] : BUF_OOB);
      }
    }
  };
  auto store_tile = [&](const f32x4 (&stage)[NL]) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      float* dl = (float*)((char*)lds + lbyte[k]);
      f32x4 v = stage[k];
      dl[0] = v.x; dl[PLANE] = v.y; dl[2 * PLANE] = v.z; dl[3 * PLANE] = v.w;
    }
  };

  int t = blockIdx.x;
  if (t >= tg.ntiles) return;
  int n, tx, ty;
  tile_coords(tg, t, n, tx, ty);
  f32x4 stage[NL];
  load_tile(stage, n, tx, ty);
  wait_vmem_all();
  store_tile(stage);
  __syncthreads();
  for (;;) {
    const int tn = t + gridDim.x;
    const bool more = tn < tg.ntiles;
    int nn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, nn, txn, tyn);
      load_tile(stage, nn, txn, tyn);               // in flight during the MFMA chain
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
          acc = mfma16(wf[rr * 3 + kx][kc], *(const float*)((const char*)lds + xbyte[kc] + (rr * LC + kx) * 4), acc);
    drain(acc);

    wait_vmem_all();
    __syncthreads();                                // every wave is done reading the tile
    if (more) store_tile(stage);

    const int y0 = ty * TR, x0 = tx * TC;
    const buf_rsrc ro = make_rsrc((char*)c1 + (((long)n * h + y0) * w + x0) * 32);
    unsigned oo = ooff;
    if (!(y0 + TR <= h && x0 + TC <= w)) oo = (y0 + orow < h && x0 + p < w) ? ooff : BUF_OOB;
    acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    buf_store4(ro, oo, acc);
    if (!more) break;
    __syncthreads();                                // next tile visible
    t = tn; n = nn; tx = txn; ty = tyn;
  }
}


static int launch_cand1_two_row(const SmallConvArgs& a, int B, hipStream_t st) {
  constexpr int G = 4;
  constexpr size_t lds = (size_t)G * group_pitch(plane_pitch16(10 * 18), G) * sizeof(float);
  static const int capacity = resident_blocks(k_cand1_two_row, 256, lds);      // once per instantiation, thread-safely (magic static)
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(a.wo, 16), cdiv(a.ho, 8), B)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(k_cand1_two_row, dim3(grid), dim3(256), lds, st, a, tg);
  ADAMVS_CHECK_LAUNCH("cand1 (two-row)");
  return 0;
}

// The same convolution for wide inputs (C = 32) with the contraction split over the waves.  Held in full, the
// A fragments are 12 * C/4 = 96 registers per lane and leave room for two waves per SIMD, too few to cover the
// load / LDS-fill / epilogue phases of each other.  Here wave k owns input channels 8k..8k+7 (24 fragment
// registers) and computes partial sums for all four row pairs of the tile; the partials meet in LDS, wave k adds
// up row pair k in a fixed order (deterministic) and writes it out.  Same MFMA count, half the registers.
template <int C>
__global__ __launch_bounds__(256, 3) void k_conv1_ksplit(const float* __restrict__ src, const float* __restrict__ wpk,
                                                      float* __restrict__ c1, int h, int w, TileGrid tg) {
  static_assert(C == 32, "four waves x two k-chunks");
  constexpr int KC = C / 4, G = C / 4, KW = KC / 4, TR = 8, TC = 16, LR = TR + 2, LC = TC + 2, NPIX = LR * LC;
  constexpr int PLANE = plane_pitch16(NPIX), GP = group_pitch(PLANE, G);
  constexpr int NL = (NPIX * G + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];           // tile [G][GP], then partials [12][64][4]
  float* red = lds + G * GP;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
  const int p = lane & 15, q = lane >> 4;
  float wf[12][KW];
#pragma unroll
  for (int t = 0; t < 12; ++t)
#pragma unroll
    for (int kc = 0; kc < KW; ++kc) wf[t][kc] = wpk[(t * KC + wave * KW + kc) * 64 + lane];

  unsigned goff[NL], lbyte[NL];
  int rc[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int j = min(tid + k * 256, NPIX * G - 1);
    const int g = j % G, pp = j / G, r = pp / LC, c = pp % LC;
    goff[k] = (unsigned)(((r * w + c) * C + 4 * g) * 4);
    lbyte[k] = (unsigned)((g * GP + r * LC + c) * 4);
    rc[k] = r | (c << 16);
    pin(goff[k]); pin(lbyte[k]); pin(rc[k]);
  }
  unsigned xbyte[KW];                                         // B-fragment origin of row pair 0 in the wave's channels
#pragma unroll
  for (int kc = 0; kc < KW; ++kc) {
    xbyte[kc] = (unsigned)(((wave * KW + kc) * GP + q * PLANE + p) * 4);
    pin(xbyte[kc]);
  }
  // partial of (source wave s, row pair rp != s) lives in slot s*3 + (rp > s ? rp - 1 : rp)
  unsigned wbyte = (unsigned)(((wave * 3) * 64 + lane) * 16);            // first slot this wave writes
  unsigned rbyte[3];                                                     // slots this wave reads, sources in order
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int s = i < wave ? i : i + 1;                                  // the other waves, ascending
    rbyte[i] = (unsigned)(((s * 3 + (wave > s ? wave - 1 : wave)) * 64 + lane) * 16);
    pin(rbyte[i]);
  }
  pin(wbyte);
  const int orow = 2 * wave + (q >> 1);
  unsigned ooff = (unsigned)(((orow * w + p) * 8 + 4 * (q & 1)) * 4);
  pin(ooff);

  auto load_tile = [&](f32x4 (&stage)[NL], int n, int tx, int ty) {
    const int ix0 = tx * TC - 1, iy0 = ty * TR - 1;
    const buf_rsrc rs = make_rsrc((const char*)src + (((long)n * h + iy0) * w + ix0) * (C * 4));
    if (iy0 >= 0 && ix0 >= 0 && iy0 + LR <= h && ix0 + LC <= w) {
#pragma unroll
      for (int k = 0; k < NL; ++k) stage[k] = buf_load4(rs, goff[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int iy = iy0 + (rc[k] & 0xffff), ix = ix0 + (rc[k] >> 16);
        stage[k] = buf_load4(rs, ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) ? goff[k] : BUF_OOB);
      }
    }
  };
  auto store_tile = [&](const f32x4 (&stage)[NL]) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      float* dl = (float*)((char*)lds + lbyte[k]);
      f32x4 v = stage[k];
      dl[0] = v.x; dl[PLANE] = v.y; dl[2 * PLANE] = v.z; dl[3 * PLANE] = v.w;
    }
  };

  int t = blockIdx.x;
  if (t >= tg.ntiles) return;
  int n, tx, ty;
  tile_coords(tg, t, n, tx, ty);
  f32x4 stage[NL];
  load_tile(stage, n, tx, ty);
  wait_vmem_all();
  store_tile(stage);
  __syncthreads();
  for (;;) {
    const int tn = t + gridDim.x;
    const bool more = tn < tg.ntiles;
    int nn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, nn, txn, tyn);
      load_tile(stage, nn, txn, tyn);               // in flight during the MFMA chains
    }
    f32x4 acc[4];
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) acc[rp] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kc = 0; kc < KW; ++kc)
#pragma unroll
          for (int rp = 0; rp < 4; ++rp)            // four independent chains
            acc[rp] = mfma16(wf[rr * 3 + kx][kc],
                             *(const float*)((const char*)lds + xbyte[kc] + ((2 * rp + rr) * LC + kx) * 4), acc[rp]);
    // partials of the row pairs the other waves finish
#pragma unroll
    for (int rp = 0; rp < 4; ++rp)
      if (rp != wave) *(f32x4*)((char*)red + wbyte + (rp > wave ? rp - 1 : rp) * 1024) = acc[rp];

    wait_vmem_all();
    __syncthreads();                                // partials visible; every wave is done reading the tile
    if (more) store_tile(stage);

    f32x4 own = wave == 0 ? acc[0] : (wave == 1 ? acc[1] : (wave == 2 ? acc[2] : acc[3]));
    f32x4 part[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) part[i] = *(const f32x4*)((const char*)red + rbyte[i]);
    // fixed order of the four channel slices: 0, 1, 2, 3
    f32x4 sum;
    if (wave == 0) sum = ((own + part[0]) + part[1]) + part[2];
    else if (wave == 1) sum = ((part[0] + own) + part[1]) + part[2];
    else if (wave == 2) sum = ((part[0] + part[1]) + own) + part[2];
    else sum = ((part[0] + part[1]) + part[2]) + own;

    const int y0 = ty * TR, x0 = tx * TC;
    const buf_rsrc ro = make_rsrc((char*)c1 + (((long)n * h + y0) * w + x0) * 32);
    unsigned oo = ooff;
    if (!(y0 + TR <= h && x0 + TC <= w)) oo = (y0 + orow < h && x0 + p < w) ? ooff : BUF_OOB;
    sum.x = fmaxf(sum.x, 0.f); sum.y = fmaxf(sum.y, 0.f); sum.z = fmaxf(sum.z, 0.f); sum.w = fmaxf(sum.w, 0.f);
    buf_store4(ro, oo, sum);
    if (!more) break;
    __syncthreads();                                // next tile visible; partials consumed
    t = tn; n = nn; tx = txn; ty = tyn;
  }
}

// The same convolution (C = 32) with the minimal-filtering form F(2, 3) ALONG x on top of the two-row form along y (round 5).
// Two neighbouring outputs of a row need four products per kernel row and channel instead of six:
//     d = the four window columns under an output pair;  v = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
//     U = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2) of the kernel row's three taps;  m_j = U_j v_j
//     out[2m] = m0 + m1 + m2,  out[2m + 1] = m1 - m2 - m3
// so a two-row pass (rows 0-7 of the MFMA tile: the 8 channels of output row y with kernel row rr, rows 8-15: of row y + 1 with
// kernel row rr - 1) issues 4 MFMAs per input row, 4 input channels and 32 output pixels where k_conv1_ksplit issues 6: 2 MFMAs per
// pixel instead of 3 (the direct one-row form: 4.5).  The columns of an MFMA are 16 output PAIRS; a lane reads its four window values
// as two 8-byte LDS reads and forms v with four vector instructions per 4 MFMAs.  U is formed from the streamed two-row fragments when
// the kernel starts (three adds and two multiplies per fragment triple): the blob keeps its twelve (rr, kx) fragments.
// Tile 8 x 32; the contraction is split over two wave pairs: wave (kh, rp2) owns input channels 16 kh .. + 15 (64 registers of U) and
// the row pairs 2 rp2, 2 rp2 + 1, reduces its sixteen sums to the four outputs per row pair BEFORE the partial sums meet in LDS
// (the output transform is linear), finishes row pair 2 rp2 + kh and hands the other to its partner: fixed order (channels 0-15 first).
template <int C>
__global__ __launch_bounds__(256, 2) void k_conv1_f23(const float* __restrict__ src, const float* __restrict__ wpk,
                                                   float* __restrict__ c1, int h, int w, TileGrid tg) {
  static_assert(C == 32, "two halves of four k-chunks");
  typedef float f32x2c __attribute__((ext_vector_type(2)));
  constexpr int KC = C / 4, G = C / 4, KH = KC / 2, TR = 8, TC = 32, LR = TR + 2, LC = TC + 2, NPIX = LR * LC;
  constexpr int PLANE = plane_pitch16(NPIX), GP = group_pitch(PLANE, G);
  static_assert((PLANE % 2) == 0 && (GP % 2) == 0 && (LC % 2) == 0, "8-byte aligned patch reads");
  constexpr int NL = (NPIX * G + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];           // tile [G][GP], then partials [4 waves][2][64] float4
  f32x4* red = (f32x4*)(lds + G * GP);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
  const int p = lane & 15, q = lane >> 4;
  const int kh = wave & 1, rp2 = wave >> 1;
  const f32x2c pm = {1.0f, -1.0f};

  float uf[4][4][KH];                                                   // [rr][j][kc]
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int kc = 0; kc < KH; ++kc) {
      const float g0 = wpk[((rr * 3 + 0) * KC + kh * KH + kc) * 64 + lane], g1 = wpk[((rr * 3 + 1) * KC + kh * KH + kc) * 64 + lane];
      const float g2 = wpk[((rr * 3 + 2) * KC + kh * KH + kc) * 64 + lane];
      uf[rr][0][kc] = g0;
      uf[rr][1][kc] = 0.5f * ((g0 + g2) + g1);
      uf[rr][2][kc] = 0.5f * ((g0 + g2) - g1);
      uf[rr][3][kc] = g2;
    }

  unsigned goff[NL], lbyte[NL];
  int rc[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int j = min(tid + k * 256, NPIX * G - 1);
    const int g = j % G, pp = j / G, r = pp / LC, c = pp % LC;
    goff[k] = (unsigned)(((r * w + c) * C + 4 * g) * 4);
    lbyte[k] = (unsigned)((g * GP + r * LC + c) * 4);
    rc[k] = r | (c << 16);
    pin(goff[k]); pin(lbyte[k]);
  }
  // the lane's patch of the wave's first row pair in the wave's first k-chunk: + k-chunk, row pair, input row as immediates
  unsigned xb = (unsigned)((kh * KH * GP + q * PLANE + (4 * rp2) * LC + 2 * p) * 4);
  pin(xb);
  const int mine = 2 * rp2 + kh;                                       // the row pair this wave finishes
  const int orow = 2 * mine + (q >> 1);                                // lane's output pixels (orow, 2 p), (orow, 2 p + 1), channels 4 (q & 1)..
  unsigned ooff = (unsigned)(((orow * w + 2 * p) * 8 + 4 * (q & 1)) * 4);
  pin(ooff);

  auto load_tile = [&](f32x4 (&stage)[NL], int n, int tx, int ty) {
    const int ix0 = tx * TC - 1, iy0 = ty * TR - 1;
    const buf_rsrc rs = make_rsrc((const char*)src + (((long)n * h + iy0) * w + ix0) * (C * 4));
    if (iy0 >= 0 && ix0 >= 0 && iy0 + LR <= h && ix0 + LC <= w) {
#pragma unroll
      for (int k = 0; k < NL; ++k) stage[k] = buf_load4(rs, goff[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int iy = iy0 + (rc[k] & 0xffff), ix = ix0 + (rc[k] >> 16);
        stage[k] = buf_load4(rs, ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) ? goff[k] : BUF_OOB);
      }
    }
  };
  auto store_tile = [&](const f32x4 (&stage)[NL]) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      float* dl = (float*)((char*)lds + lbyte[k]);
      f32x4 v = stage[k];
      dl[0] = v.x; dl[PLANE] = v.y; dl[2 * PLANE] = v.z; dl[3 * PLANE] = v.w;
    }
  };

  int t = blockIdx.x;
  if (t >= tg.ntiles) return;
  int n, tx, ty;
  tile_coords(tg, t, n, tx, ty);
  f32x4 stage[NL];
  load_tile(stage, n, tx, ty);
  wait_vmem_all();
  store_tile(stage);
  __syncthreads();
  for (;;) {
    const int tn = t + gridDim.x;
    const bool more = tn < tg.ntiles;
    int nn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, nn, txn, tyn);
      load_tile(stage, nn, txn, tyn);               // in flight during the MFMA chains
    }
    f32x4 m[2][4];                                  // [the wave's row pair][j]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) m[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int kc = 0; kc < KH; ++kc)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const char* at = (const char*)lds + xb + (kc * GP + (2 * a + rr) * LC) * 4;
          const f32x2c d01 = *(const f32x2c*)at, d23 = *(const f32x2c*)(at + 8);
          const f32x2c v03 = d01 - d23;                              // (d0 - d2, d1 - d3)
          // (v1, v2) = (d2 + d1, d2 - d1) as ONE packed FMA on the pairs the reads deliver (op_sel picks d1 twice, d2 twice)
          const f32x2c v12 = __builtin_elementwise_fma(__builtin_shufflevector(d01, d01, 1, 1), pm, __builtin_shufflevector(d23, d23, 0, 0));
          m[a][0] = mfma16(uf[rr][0][kc], v03.x, m[a][0]);
          m[a][1] = mfma16(uf[rr][1][kc], v12.x, m[a][1]);
          m[a][2] = mfma16(uf[rr][2][kc], v12.y, m[a][2]);
          m[a][3] = mfma16(uf[rr][3][kc], v03.y, m[a][3]);
        }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) drain(m[a][j]);
    // the two outputs of every pair, for both row pairs; the one the partner finishes goes to LDS
    f32x4 y[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      y[a][0] = (m[a][0] + m[a][1]) + m[a][2];
      y[a][1] = (m[a][1] - m[a][2]) - m[a][3];
    }
    red[(wave * 2 + 0) * 64 + lane] = kh ? y[0][0] : y[1][0];      // wave kh = 0 finishes its row pair 0 and hands over row pair 1; kh = 1 the reverse
    red[(wave * 2 + 1) * 64 + lane] = kh ? y[0][1] : y[1][1];

    wait_vmem_all();
    __syncthreads();                                // partials visible; every wave is done reading the tile
    if (more) store_tile(stage);

    const f32x4 p0 = red[((wave ^ 1) * 2 + 0) * 64 + lane], p1 = red[((wave ^ 1) * 2 + 1) * 64 + lane];
    f32x4 o0 = kh ? p0 + y[1][0] : y[0][0] + p0;    // fixed order: channels 0-15 (kh = 0), then 16-31
    f32x4 o1 = kh ? p1 + y[1][1] : y[0][1] + p1;
    const int y0 = ty * TR, x0 = tx * TC;
    const buf_rsrc ro = make_rsrc((char*)c1 + (((long)n * h + y0) * w + x0) * 32);
    unsigned oa = ooff, ob = ooff + 32;
    if (!(y0 + TR <= h && x0 + TC <= w)) {
      const bool rowok = y0 + orow < h;
      oa = (rowok && x0 + 2 * p < w) ? ooff : BUF_OOB;
      ob = (rowok && x0 + 2 * p + 1 < w) ? ooff + 32 : BUF_OOB;
    }
    o0.x = fmaxf(o0.x, 0.f); o0.y = fmaxf(o0.y, 0.f); o0.z = fmaxf(o0.z, 0.f); o0.w = fmaxf(o0.w, 0.f);
    o1.x = fmaxf(o1.x, 0.f); o1.y = fmaxf(o1.y, 0.f); o1.z = fmaxf(o1.z, 0.f); o1.w = fmaxf(o1.w, 0.f);
    buf_store4(ro, oa, o0);
    buf_store4(ro, ob, o1);
    if (!more) break;
    __syncthreads();                                // next tile visible; partials consumed
    t = tn; n = nn; tx = txn; ty = tyn;
  }
}

// The same form for the narrow inputs of stages 2 and 3 (C = 16, 8): no split of the contraction -- wave k owns row pair k of an
// 8 x 32 tile with all of U in registers (4 rr x 4 j x C/4: 64 / 32), four sums, the output transform in registers.
template <int C>
__global__ __launch_bounds__(256) void k_conv1_f23_rows(const float* __restrict__ src, const float* __restrict__ wpk,
                                                       float* __restrict__ c1, int h, int w, TileGrid tg) {
  typedef float f32x2c __attribute__((ext_vector_type(2)));
  constexpr int KC = C / 4, G = C / 4, TR = 8, TC = 32, LR = TR + 2, LC = TC + 2, NPIX = LR * LC;
  constexpr int PLANE = plane_pitch16(NPIX), GP = (group_pitch(PLANE, G) + 1) & ~1;
  static_assert((PLANE % 2) == 0 && (GP % 2) == 0 && (LC % 2) == 0, "8-byte aligned patch reads");
  constexpr int NL = (NPIX * G + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];           // tile [G][GP]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = row pair
  const int p = lane & 15, q = lane >> 4;
  const f32x2c pm = {1.0f, -1.0f};

  float uf[4][4][KC];                                                   // [rr][j][kc]
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const float g0 = wpk[((rr * 3 + 0) * KC + kc) * 64 + lane], g1 = wpk[((rr * 3 + 1) * KC + kc) * 64 + lane];
      const float g2 = wpk[((rr * 3 + 2) * KC + kc) * 64 + lane];
      uf[rr][0][kc] = g0;
      uf[rr][1][kc] = 0.5f * ((g0 + g2) + g1);
      uf[rr][2][kc] = 0.5f * ((g0 + g2) - g1);
      uf[rr][3][kc] = g2;
    }

  unsigned goff[NL], lbyte[NL];
  int rc[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int j = min(tid + k * 256, NPIX * G - 1);
    const int g = j % G, pp = j / G, r = pp / LC, c = pp % LC;
    goff[k] = (unsigned)(((r * w + c) * C + 4 * g) * 4);
    lbyte[k] = (unsigned)((g * GP + r * LC + c) * 4);
    rc[k] = r | (c << 16);
    pin(goff[k]); pin(lbyte[k]); pin(rc[k]);
  }
  unsigned xb = (unsigned)((q * PLANE + (2 * wave) * LC + 2 * p) * 4);
  pin(xb);
  const int orow = 2 * wave + (q >> 1);                                // lane's output pixels (orow, 2 p), (orow, 2 p + 1), channels 4 (q & 1)..
  unsigned ooff = (unsigned)(((orow * w + 2 * p) * 8 + 4 * (q & 1)) * 4);
  pin(ooff);

  auto load_tile = [&](f32x4 (&stage)[NL], int n, int tx, int ty) {
    const int ix0 = tx * TC - 1, iy0 = ty * TR - 1;
    const buf_rsrc rs = make_rsrc((const char*)src + (((long)n * h + iy0) * w + ix0) * (C * 4));
    if (iy0 >= 0 && ix0 >= 0 && iy0 + LR <= h && ix0 + LC <= w) {
#pragma unroll
      for (int k = 0; k < NL; ++k) stage[k] = buf_load4(rs, goff[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int iy = iy0 + (rc[k] & 0xffff), ix = ix0 + (rc[k] >> 16);
        stage[k] = buf_load4(rs, ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) ? goff[k] : BUF_OOB);
      }
    }
  };
  auto store_tile = [&](const f32x4 (&stage)[NL]) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      float* dl = (float*)((char*)lds + lbyte[k]);
      f32x4 v = stage[k];
      dl[0] = v.x; dl[PLANE] = v.y; dl[2 * PLANE] = v.z; dl[3 * PLANE] = v.w;
    }
  };

  int t = blockIdx.x;
  if (t >= tg.ntiles) return;
  int n, tx, ty;
  tile_coords(tg, t, n, tx, ty);
  f32x4 stage[NL];
  load_tile(stage, n, tx, ty);
  wait_vmem_all();
  store_tile(stage);
  __syncthreads();
  for (;;) {
    const int tn = t + gridDim.x;
    const bool more = tn < tg.ntiles;
    int nn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, nn, txn, tyn);
      load_tile(stage, nn, txn, tyn);               // in flight during the MFMA chain
    }
    f32x4 m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const char* at = (const char*)lds + xb + (kc * GP + rr * LC) * 4;
        const f32x2c d01 = *(const f32x2c*)at, d23 = *(const f32x2c*)(at + 8);
        const f32x2c v03 = d01 - d23;
        const f32x2c v12 = __builtin_elementwise_fma(__builtin_shufflevector(d01, d01, 1, 1), pm, __builtin_shufflevector(d23, d23, 0, 0));
        m[0] = mfma16(uf[rr][0][kc], v03.x, m[0]);
        m[1] = mfma16(uf[rr][1][kc], v12.x, m[1]);
        m[2] = mfma16(uf[rr][2][kc], v12.y, m[2]);
        m[3] = mfma16(uf[rr][3][kc], v03.y, m[3]);
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) drain(m[j]);
    f32x4 o0 = (m[0] + m[1]) + m[2], o1 = (m[1] - m[2]) - m[3];

    wait_vmem_all();
    __syncthreads();                                // every wave is done reading the tile
    if (more) store_tile(stage);

    const int y0 = ty * TR, x0 = tx * TC;
    const buf_rsrc ro = make_rsrc((char*)c1 + (((long)n * h + y0) * w + x0) * 32);
    unsigned oa = ooff, ob = ooff + 32;
    if (!(y0 + TR <= h && x0 + TC <= w)) {
      const bool rowok = y0 + orow < h;
      oa = (rowok && x0 + 2 * p < w) ? ooff : BUF_OOB;
      ob = (rowok && x0 + 2 * p + 1 < w) ? ooff + 32 : BUF_OOB;
    }
    o0.x = fmaxf(o0.x, 0.f); o0.y = fmaxf(o0.y, 0.f); o0.z = fmaxf(o0.z, 0.f); o0.w = fmaxf(o0.w, 0.f);
    o1.x = fmaxf(o1.x, 0.f); o1.y = fmaxf(o1.y, 0.f); o1.z = fmaxf(o1.z, 0.f); o1.w = fmaxf(o1.w, 0.f);
    buf_store4(ro, oa, o0);
    buf_store4(ro, ob, o1);
    if (!more) break;
    __syncthreads();                                // next tile visible
    t = tn; n = nn; tx = txn; ty = tyn;
  }
}

template <int C>
static int launch_conv1_f23_rows(const float* cost, const float* w, float* c1, int N, int h, int w_, hipStream_t st) {
  constexpr int G = C / 4;
  constexpr size_t lds = (size_t)G * ((group_pitch(plane_pitch16(10 * 34), G) + 1) & ~1) * sizeof(float);
  auto kern = k_conv1_f23_rows<C>;
  static const int capacity = resident_blocks(kern, 256, lds);      // once per instantiation, thread-safely (magic static)
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(w_, 32), cdiv(h, 8), N)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, cost, w, c1, h, w_, tg);
  ADAMVS_CHECK_LAUNCH("conv1 (F(2,3) along x)");
  return 0;
}

// option conv1_f23: bit 1 = C = 32 (stage 1), bit 2 = C = 16 / 8 (stages 2, 3) in the F(2, 3)-along-x form; 0 = k_conv1_ksplit /
// k_conv1_two_row, as in rounds 1-4 (A/B).  Measured at cfg2 (aggregation + conv1 per step): 43.75 -> 38.28 ms at 128 tiles, 85.7 -> 76.0 at 256.
static int conv1_f23() { return opt(OPT_CONV1_F23); }

static int launch_conv1_f23_32(const float* cost, const float* w, float* c1, int N, int h, int w_, hipStream_t st) {
  constexpr int C = 32;
  constexpr size_t lds = ((size_t)(C / 4) * group_pitch(plane_pitch16(10 * 34), C / 4) + 4 * 2 * 64 * 4) * sizeof(float);
  static_assert(lds <= 64 * 1024, "default dynamic LDS limit");
  auto kern = k_conv1_f23<C>;
  static const int capacity = resident_blocks(kern, 256, lds);      // once per instantiation, thread-safely (magic static)
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(w_, 32), cdiv(h, 8), N)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, cost, w, c1, h, w_, tg);
  ADAMVS_CHECK_LAUNCH("conv1 (F(2,3) along x)");
  return 0;
}

static int launch_conv1_ksplit32(const float* cost, const float* w, float* c1, int N, int h, int w_, hipStream_t st) {
  constexpr int C = 32;
  constexpr size_t lds = ((size_t)(C / 4) * group_pitch(plane_pitch16(10 * 18), C / 4) + 12 * 64 * 4) * sizeof(float);
  auto kern = k_conv1_ksplit<C>;
  static const int capacity = resident_blocks(kern, 256, lds);      // once per instantiation, thread-safely (magic static)
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(w_, 16), cdiv(h, 8), N)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, cost, w, c1, h, w_, tg);
  ADAMVS_CHECK_LAUNCH("conv1");
  return 0;
}

template <int C>
static int launch_conv1_c(const float* cost, const float* w, float* c1, int N, int h, int w_, hipStream_t st) {
  constexpr size_t lds = (size_t)(C / 4) * group_pitch(plane_pitch16(10 * 18), C / 4) * sizeof(float);
  auto kern = k_conv1_two_row<C>;
  static const int capacity = resident_blocks(kern, 256, lds);      // once per instantiation, thread-safely (magic static)
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(w_, 16), cdiv(h, 8), N)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, cost, w, c1, h, w_, tg);
  ADAMVS_CHECK_LAUNCH("conv1");
  return 0;
}

template <int CA, int CB, int NT>
static int launch_small_pro(const SmallConvArgs& a, const GruPro& pro, int B, hipStream_t st) {
  typedef ConvSmallRole<CA, CB, NT, 1, EPI_LINEAR, 4, 1> Role;
  auto kern = k_conv_small_pro<CA, CB, NT>;
  static const int capacity = resident_blocks(kern, 256, Role::LDS_BYTES);      // once per instantiation, thread-safely
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(a.wo, 16), cdiv(a.ho, 4), B)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Role::LDS_BYTES, st, a, tg, pro);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "conv_pair (folded): %s", hipGetErrorString(e));
  return 0;
}

// out = conv3x3(cat(srcA, srcB)) + bias on compact channel-last maps (the ConvGRUCell2 convolutions of MS-REDNet's two
// shallow levels: gate_conv / output_conv of reference models/module.py:62-67 before their GroupNorm)
// gn_part != null: GroupNorm partial sums of the output in the epilogue (SmallConvArgs); *gn_parts receives the number of
// partials per (sample, group), or 0 when the map has more tiles than the partial buffer holds (the caller then reduces the
// map with k_gn_partial as before).
int launch_conv_pair(const float* srcA, int CA, const float* srcB, int CB, const float* wpk, const float* bias, float* out,
                     int cout, int B, int h, int w, hipStream_t st, double* gn_part, int gn_hc, int gn_groups, int* gn_parts,
                     const GruPro* pro) {
  SmallConvArgs a{srcA, srcB, wpk, bias, out, nullptr, h, w, h, w, cout};
  if (gn_parts) *gn_parts = 0;
  const long parts = (long)cdiv(w, 16) * cdiv(h, 4) * 4;            // launch_small: tiles of 4 rows x 16 columns, one run per wave
  if (gn_part && gn_parts && gn_epilogue_partials(parts, B)) {
    a.gn_part = gn_part; a.gn_hc = gn_hc; a.gn_groups = gn_groups;
    *gn_parts = (int)parts;
  }
  if (pro) {
    if (CB == 8 && cout <= 16) {
      if (CA == 32) return launch_small_pro<32, 8, 1>(a, *pro, B, st);
      if (CA == 16) return launch_small_pro<16, 8, 1>(a, *pro, B, st);
      if (CA == 8) return launch_small_pro<8, 8, 1>(a, *pro, B, st);
    }
    if (CA == 16 && CB == 16 && cout <= 16) return launch_small_pro<16, 16, 1>(a, *pro, B, st);
    if (CA == 16 && CB == 16 && cout <= 32) return launch_small_pro<16, 16, 2>(a, *pro, B, st);
  }
  if (CB == 8 && cout <= 16) {
    if (CA == 32) return launch_small<32, 8, 1, 1, EPI_LINEAR>(a, B, st, "conv_pair");
    if (CA == 16) return launch_small<16, 8, 1, 1, EPI_LINEAR>(a, B, st, "conv_pair");
    if (CA == 8) return launch_small<8, 8, 1, 1, EPI_LINEAR>(a, B, st, "conv_pair");
  }
  if (CA == 16 && CB == 16 && cout <= 16) return launch_small<16, 16, 1, 1, EPI_LINEAR>(a, B, st, "conv_pair");
  if (CA == 16 && CB == 16 && cout <= 32) return launch_small<16, 16, 2, 1, EPI_LINEAR>(a, B, st, "conv_pair");
  return set_error(-1, "conv3x3_pair: (CA=%d, CB=%d, cout=%d) unsupported: (32|16|8, 8, <=16) or (16, 16, <=32)", CA, CB, cout);
}

bool conv_pair_epilogue_partials(int B, int h, int w) { return gn_epilogue_partials((long)cdiv(w, 16) * cdiv(h, 4) * 4, B); }

int launch_conv1(const float* cost, const float* w, float* c1, int N, int C, int h, int w_, int precision, hipStream_t st) {
  if (precision == PRECISION_BF16X3) return launch_conv1_bf16x3(cost, w, c1, N, C, h, w_, st);
  if (C == 32) return (conv1_f23() & 1) ? launch_conv1_f23_32(cost, w, c1, N, h, w_, st) : launch_conv1_ksplit32(cost, w, c1, N, h, w_, st);
  if (C == 16) return (conv1_f23() & 2) ? launch_conv1_f23_rows<16>(cost, w, c1, N, h, w_, st) : launch_conv1_c<16>(cost, w, c1, N, h, w_, st);
  if (C == 8) return (conv1_f23() & 2) ? launch_conv1_f23_rows<8>(cost, w, c1, N, h, w_, st) : launch_conv1_c<8>(cost, w, c1, N, h, w_, st);
  return set_error(-1, "conv1: C=%d unsupported (8, 16 or 32)", C);
}

// One recurrent step after conv1: c1 -> GRU1 -> conv2 -> GRU2 -> decoder -> vol[:, d].
int launch_slice_step(const float* c1, const FuseWeights& fw, const StepBuffers& sb, float* vol, int B, int h, int w, int D,
                      int d, int in_up, int precision, hipStream_t st, float** h1_now, float** h2_now) {
  int h2 = h / 2, w2 = w / 2, rc;
  float* h1 = sb.h1;
  float* h2s = sb.h2;
  if (precision == PRECISION_BF16X3) {
    if ((rc = launch_gru_convs_bf16x3(c1, fw, sb, B, h, w, d, &h1, &h2s, st))) return rc;
  } else {
  {  // GRU level 1: gates on cat(c1, h1), candidate on cat(c1, r*h1)
    SmallConvArgs g{c1, sb.h1, fw.gates1, fw.gates1_b, sb.rh1, sb.u1, h, w, h, w, 16};
    if ((gru_wino_mask() & 1) && fw.gates1_w) {
      g.wpk = fw.gates1_w;
      if ((rc = launch_small_wino<8, 8, 1, EPI_GATES>(g, B, st, "gates1 (F(2x2,3x3))"))) return rc;
    } else if ((rc = launch_small<8, 8, 1, 1, EPI_GATES>(g, B, st, "gates1"))) return rc;
    SmallConvArgs c{c1, sb.rh1, fw.cand1, fw.cand1_b, sb.h1, sb.u1, h, w, h, w, 8};     // cand1: two-row fragments
    if ((gru_wino_mask() & 8) && fw.cand1_w) {
      c.wpk = fw.cand1_w;
      if ((rc = launch_small_wino<8, 8, 1, EPI_CAND>(c, B, st, "cand1 (F(2x2,3x3))"))) return rc;
    } else if ((rc = launch_cand1_two_row(c, B, st))) return rc;
  }
  {  // conv2: 8 -> 16, stride 2, ReLU
    SmallConvArgs a{h1, nullptr, fw.conv2, nullptr, sb.c2, nullptr, h, w, h2, w2, 16};
    if ((rc = launch_small<8, 0, 1, 2, EPI_RELU>(a, B, st, "conv2"))) return rc;
  }
  {  // GRU level 2
    SmallConvArgs g{sb.c2, sb.h2, fw.gates2, fw.gates2_b, sb.rh2, sb.u2, h2, w2, h2, w2, 32};
    if ((gru_wino_mask() & 2) && fw.gates2_w) {
      g.wpk = fw.gates2_w;
      if ((rc = launch_small_wino<16, 16, 2, EPI_GATES>(g, B, st, "gates2 (F(2x2,3x3))"))) return rc;
    } else if ((rc = launch_small<16, 16, 2, 1, EPI_GATES>(g, B, st, "gates2"))) return rc;
    SmallConvArgs c{sb.c2, sb.rh2, fw.cand2, fw.cand2_b, sb.h2, sb.u2, h2, w2, h2, w2, 16};
    if ((gru_wino_mask() & 4) && fw.cand2_w) {
      c.wpk = fw.cand2_w;
      if ((rc = launch_small_wino<16, 16, 1, EPI_CAND>(c, B, st, "cand2 (F(2x2,3x3))"))) return rc;
    } else if ((rc = launch_small<16, 16, 1, 1, EPI_CAND>(c, B, st, "cand2"))) return rc;
  }
  }
  if (h1_now) *h1_now = h1;
  if (h2_now) *h2_now = h2s;
  DecoderArgs da{h2s, h1, fw.upconv1, fw.upconv1_b, fw.final_w, vol, h, w, D, d};
  TileGrid tg;
  if ((rc = make_tile_grid(tg, cdiv(w, DecoderRole<true>::TCI), cdiv(h, DecoderRole<true>::TRI), B))) return rc;
  constexpr size_t dlds = DecoderRole<true>::LDS_BYTES;
  if (in_up) {
    static const int cap_up = resident_blocks(k_decoder<true>, 256, dlds);      // once, thread-safely (magic static)
    hipLaunchKernelGGL((k_decoder<true>), dim3(tg.ntiles < cap_up ? tg.ntiles : cap_up), dim3(256), dlds, st, da, tg);
  } else {
    static const int cap_flat = resident_blocks(k_decoder<false>, 256, dlds);
    hipLaunchKernelGGL((k_decoder<false>), dim3(tg.ntiles < cap_flat ? tg.ntiles : cap_flat), dim3(256), dlds, st, da, tg);
  }
  ADAMVS_CHECK_LAUNCH("decoder");
  return 0;
}

int launch_soft_argmin(const float* vol, const float* planes, float* depth, float* conf, int B, int D, int h, int w,
                       int in_up, hipStream_t st) {
  size_t total = (size_t)B * (in_up ? 4 : 1) * h * w;
  unsigned nb = (unsigned)((total + 255) / 256);
  if (in_up) hipLaunchKernelGGL((k_soft_argmin<true>), dim3(nb), dim3(256), 0, st, vol, planes, depth, conf, D, h, w, total);
  else hipLaunchKernelGGL((k_soft_argmin<false>), dim3(nb), dim3(256), 0, st, vol, planes, depth, conf, D, h, w, total);
  ADAMVS_CHECK_LAUNCH("soft_argmin");
  return 0;
}

}  // namespace adamvs
