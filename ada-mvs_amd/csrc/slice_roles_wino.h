// The 3x3 convolutions of the ConvGRU cells (reference models/module.py:24-52) in the minimal-filtering form F(2x2, 3x3),
// fp32 throughout: 16 products per 2x2 outputs and channel pair instead of 36.
//
// Why here: the recurrence is matrix-bound on the fp32 pipe at 60-70 % utilisation (lesson 7) and every
// other way round that bound has been built and measured (fused levels, shared launches, deeper prefetch: none moves it).
// What is left is the number of multiplications.  CostRegNet2D's stride-1 layers already run this way (costreg2d_wino.hip);
// for the 16- and 32-channel convolutions of the GRU cells the same algebra needs a different mapping, because K is short
// and the weights of a whole layer fit the register file:
//   * workgroup = 4 waves = the 4 ROWS i of the transformed 4x4 patch (as in k_conv_wino): wave i reads two raw patch rows
//     (d0 - d2 | d1 + d2 | d2 - d1 | d1 - d3), forms its row of Bt d B with 8 vector instructions per 4 input channels and
//     feeds 4 NT MFMAs with them (positions (i, 0..3) x NT row tiles): one vector instruction per MFMA at NT = 2, no
//     transform computed twice;
//   * its 4 NT KC weight fragments U = G g Gt (host, double precision) stay in registers for the whole launch;
//   * a tile is 8 x 32 output pixels = 4 tile rows of 16 tiles (the 16 columns of an MFMA); per tile row the four patch rows
//     meet through LDS (Z_i[b] = sum_j M[i][j] At[b][j]; wave (a, b) forms Y[a][b] = sum_i At[a][i] Z_i[b]) and every wave
//     ends up with ONE pixel of each 2x2 tile for all NT row tiles: the GRU epilogues (sigmoid, r*h, u; tanh, blend) are
//     spread evenly over the lanes.
// MFMAs per 256 pixels and wave: gates2 (32 -> 32) 256 instead of 576, cand2 (32 -> 16) 128 / 288, gates1 (16 -> 16) 64 / 144.
#pragma once
#include "slice_roles.h"

// Timing builds only (tools/build_variant.py <name> -DWINO_GRU_EXP=<bits>; results are wrong): 1 no barrier per exchange round,
// 2 no transcendentals in the epilogue, 4 no input transform, 8 no exchange at all (every wave combines its own Z).
#ifndef WINO_GRU_EXP
#define WINO_GRU_EXP 0
#endif

namespace adamvs {

typedef float f32x2w __attribute__((ext_vector_type(2)));
typedef unsigned u32x2w __attribute__((ext_vector_type(2)));

// Round 5 (A/B: tools/build_variant.py <name> -DWINO_GRU_IL=0): the bias in accumulator position (1, 1) for every role, and for the
// level-1 gates the reset / update rows interleaved (see `IL` below)
#ifndef WINO_GRU_IL
#define WINO_GRU_IL 1
#endif
// Round 5 (A/B: -DWINO_GRU_RPR2=0): two tile rows per exchange round in the level-2 candidate (ConvWinoRole::RPR)
#ifndef WINO_GRU_RPR2
#define WINO_GRU_RPR2 1
#endif

// wpk: U fragments [NT][4 patch rows i][4 patch columns j][KC][64 lanes], lane l = U[i][j][cout = 16 nt + (l & 15)][cin = 4 kc + (l >> 4)]
// (packing.pack_small_conv_wino); the other fields of SmallConvArgs as for ConvSmallRole with the same EPI.
template <int CA, int CB, int NT, int EPI>
struct ConvWinoRole {
  typedef SmallConvArgs Args;
  static_assert(EPI == EPI_GATES || EPI == EPI_CAND, "the GRU epilogues");
  static constexpr int CIN = CA + CB, KC = CIN / 4, G = KC, GA = CA / 4, GB = CB / 4, HC = CB;
  static constexpr int TR = 8, TC = 32, LR = TR + 2, LC = TC + 2, NPIX = LR * LC;
  static constexpr int PLANE = plane_pitch16(NPIX), GP = group_pitch(PLANE, G);
  static_assert((PLANE % 2) == 0 && (GP % 2) == 0 && (LC % 2) == 0, "8-byte aligned patch reads");
  static constexpr int NA = (NPIX * GA + 255) / 256, NB = (NPIX * GB + 255) / 256, NL = NA + NB;
  static constexpr int Z0 = (G * GP + 3) & ~3;                         // floats: the exchange, two buffers of [RPR rows][4 waves][2 b][NT][64 lanes] float4
  static constexpr int ZBUF = 4 * 2 * NT * 64 * 4;
  // tile rows per exchange round.  2 for the level-2 candidate (round 5): half the barriers and eight independent sums per wave; its
  // window (47 KB) + 32 KB of exchange still fit twice on a CU.  The level-1 gates (20 + 16 KB: four workgroups per CU) and the
  // level-2 gates (NT = 2: 47 + 32 KB) would lose a workgroup per CU to it and keep 1.
  static constexpr int RPR = (WINO_GRU_RPR2 && EPI == EPI_CAND && NT == 1 && KC == 8) ? 2 : 1;
  static constexpr size_t LDS_BYTES = (size_t)(Z0 + 2 * RPR * ZBUF) * sizeof(float);
  static constexpr int TILE_W = TC, TILE_H = TR;
  static int tiles_x(const Args& a) { return cdiv(a.wo, TC); }
  static int tiles_y(const Args& a) { return cdiv(a.ho, TR); }

  static __device__ __forceinline__ void run(const Args& a, const TileGrid& tg, TileRange tr, int wg, int nwg, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = patch row i
  const int p = lane & 15, q = lane >> 4;

  // IL (the 16-row gate convolution of level 1, round 5): the gate rows as the lanes want them -- MFMA row 4 q + e = reset-gate channel
  // 2 q + e (e = 0, 1) | update-gate channel 2 q + e - 2 (e = 2, 3) -- so that every lane ends with two reset and two update values of
  // the same channels and no half of the wave sits out the r * h branch; a row permutation of A is a lane permutation of its
  // fragments, applied here: the blob keeps the reference's row order.
  constexpr bool IL = WINO_GRU_IL && EPI == EPI_GATES && NT == 1 && HC == 8;
  const int row_of_lane = IL ? ((lane & 2) ? 8 + 2 * ((lane & 15) >> 2) + (lane & 1) : 2 * ((lane & 15) >> 2) + (lane & 1)) : (lane & 15);
  const int src_lane = (lane & 48) | row_of_lane;
  float uf[NT][4][KC];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) uf[nt][j][kc] = a.wpk[((((nt * 4 + wave) * 4 + j) * KC) + kc) * 64 + src_lane];
  // The transformed filters carry the factor that makes the accumulator v_exp_f32's argument (-log2 e for the gates, 2 log2 e for the
  // candidate: packing.pack_small_conv_wino); the bias, shared with the direct kernels, is scaled here, once per launch.  It rides in
  // accumulator position (1, 1) (round 5): At E A is all ones for the unit element E11, so the sum that wave 1 starts from the bias
  // carries it into all four outputs of a tile (costreg2d_wino.hip does the same) and the epilogue adds nothing.
  constexpr float PRE = EPI == EPI_GATES ? -1.4426950408889634f : 2.8853900817779268f;
  f32x4 bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const f32x4 b = IL ? f32x4{a.bias[2 * q], a.bias[2 * q + 1], a.bias[8 + 2 * q], a.bias[8 + 2 * q + 1]} : *(const f32x4*)(a.bias + nt * 16 + 4 * q);
    bias[nt] = WINO_GRU_IL ? (wave == 1 ? b * PRE : f32x4{0.f, 0.f, 0.f, 0.f}) : b * PRE;
    if (WINO_GRU_IL) drain(bias[nt]);             // (kept in registers: the select is made once)
  }

  // ---- window fill: as ConvSmallRole (two sources, planar groups)
  unsigned goff[NL], lbyte[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const bool isA = k < NA;
    const int gs = isA ? GA : GB, cs = isA ? CA : CB;
    int j = tid + (isA ? k : k - NA) * 256;
    j = min(j, NPIX * gs - 1);
    const int g = j % gs, pp = j / gs, r = pp / LC, c = pp % LC;
    goff[k] = (unsigned)(((r * a.wi + c) * cs + 4 * g) * 4);
    lbyte[k] = (unsigned)((((isA ? 0 : GA) + g) * GP + r * LC + c) * 4);
    pin(goff[k]); pin(lbyte[k]);
  }
  // raw patch rows of the wave: T = rowA + sgn rowB  (i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3)
  const int rowA = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int rowB = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
  const float sgn = wave == 1 ? 1.0f : -1.0f;
  const f32x2w sgn2 = {sgn, sgn};
  unsigned pa = (unsigned)((q * PLANE + rowA * LC + 2 * p) * 4), pb = (unsigned)((q * PLANE + rowB * LC + 2 * p) * 4);
  pin(pa); pin(pb);
  // epilogue: wave (oa, ob) owns output pixel (2 t + oa, 2 p + ob) of every 2x2 tile of tile row t
  const int oa = wave >> 1, ob = wave & 1;
  const float os = oa ? -1.0f : 1.0f;
  const int CO = HC;                                                   // channels per pixel of both destinations
  unsigned ooff = (unsigned)(((oa * a.wo + 2 * p + ob) * CO + 4 * q) * 4);      // + 2 t rows; GATES: r*h and u share it (co4 or co4 - HC)
  pin(ooff);
  // GATES: the state channels 4 q .. of the wave's pixel, in the window (pixel (2 t + oa + 1, 2 p + ob + 1)); IL: channels 2 q, 2 q + 1
  const unsigned hbyte = IL ? (unsigned)(((GA + (q >> 1)) * GP + 2 * (q & 1) * PLANE + (oa + 1) * LC + 2 * p + ob + 1) * 4)
                            : (unsigned)(((GA + min(q, GB - 1)) * GP + (oa + 1) * LC + 2 * p + ob + 1) * 4);
  unsigned ooff2 = (unsigned)(((oa * a.wo + 2 * p + ob) * CO + 2 * q) * 4);     // IL: the lane's channel pair of both destinations
  if (IL) pin(ooff2);
  float* zl = lds + Z0;

  auto load_tile = [&](f32x4 (&stage)[NL], int b, int tx, int ty) {
    const int ix0 = tx * TC - 1, iy0 = ty * TR - 1;
    const long pix0 = ((long)b * a.hi + iy0) * a.wi + ix0;
    const buf_rsrc ra = make_rsrc((const char*)a.srcA + pix0 * (CA * 4));
    const buf_rsrc rb = make_rsrc((const char*)a.srcB + pix0 * (CB * 4));
    if (iy0 >= 0 && ix0 >= 0 && iy0 + LR <= a.hi && ix0 + LC <= a.wi) {
#pragma unroll
      for (int k = 0; k < NL; ++k) stage[k] = buf_load4(k < NA ? ra : rb, goff[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const bool isA = k < NA;
        const int gs = isA ? GA : GB;
        const int pp = min(tid + (isA ? k : k - NA) * 256, NPIX * gs - 1) / gs;
        const int iy = iy0 + pp / LC, ix = ix0 + pp % LC;
        stage[k] = buf_load4(isA ? ra : rb, ((unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi) ? goff[k] : BUF_OOB);
      }
    }
  };
  auto store_tile = [&](const f32x4 (&stage)[NL]) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      float* dl = (float*)((char*)lds + lbyte[k]);
      const f32x4 v = stage[k];
      dl[0] = v.x; dl[PLANE] = v.y; dl[2 * PLANE] = v.z; dl[3 * PLANE] = v.w;
    }
  };

  int t = tr.begin + wg;
  if (t >= tr.end) return;
  int b, tx, ty;
  tile_coords(tg, t, b, tx, ty);
  f32x4 stage[NL];
  load_tile(stage, b, tx, ty);
  wait_vmem_all();
  store_tile(stage);
  __syncthreads();
  for (;;) {
    const int oy0 = ty * TR, ox0 = tx * TC;
    const long opix0 = ((long)b * a.ho + oy0) * a.wo + ox0;
    const buf_rsrc r0 = make_rsrc((char*)a.dst0 + opix0 * (CO * 4));
    const buf_rsrc r1 = make_rsrc((char*)a.dst1 + opix0 * (CO * 4));
    const buf_rsrc rin = make_rsrc((const char*)(a.hin ? a.hin : a.dst0) + opix0 * (CO * 4));
    const bool full = oy0 + TR <= a.ho && ox0 + TC <= a.wo;
    // CAND: the epilogue operands of all four rounds first (vmcnt retires in order: requested behind the next window they
    // would make the first round wait for the whole window)
    f32x4 pre_u[EPI == EPI_CAND ? 4 : 1][NT], pre_h[EPI == EPI_CAND ? 4 : 1][NT];
    if (EPI == EPI_CAND) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        unsigned o = ooff + (unsigned)(2 * r * a.wo * CO * 4);
        if (!full && !(oy0 + 2 * r + oa < a.ho && ox0 + 2 * p + ob < a.wo)) o = BUF_OOB;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const unsigned on = (o != BUF_OOB && nt * 16 + 4 * q < HC) ? o + nt * 64 : BUF_OOB;
          pre_u[EPI == EPI_CAND ? r : 0][nt] = buf_load4(r1, on);
          pre_h[EPI == EPI_CAND ? r : 0][nt] = buf_load4(rin, on);
        }
      }
    }
    const int tn = t + nwg;
    const bool more = tn < tr.end;
    int bn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, bn, txn, tyn);
      load_tile(stage, bn, txn, tyn);                                  // in flight during the four tile rows
    }

#pragma unroll
    for (int rd = 0; rd < 4 / RPR; ++rd) {                             // exchange round: tile rows rd * RPR .. (output rows 2 tr4, 2 tr4 + 1 each)
      f32x4 m[RPR][NT][4];
#pragma unroll
      for (int rr = 0; rr < RPR; ++rr)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) m[rr][nt][j] = (WINO_GRU_IL && j == 1) ? bias[nt] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int rr = 0; rr < RPR; ++rr) {
          const int tr4 = rd * RPR + rr;
          const int off = (kc * GP + 2 * tr4 * LC) * 4;
          const f32x2w a01 = *(const f32x2w*)((const char*)lds + pa + off), a23 = *(const f32x2w*)((const char*)lds + pa + off + 8);
          const f32x2w b01 = *(const f32x2w*)((const char*)lds + pb + off), b23 = *(const f32x2w*)((const char*)lds + pb + off + 8);
          // T = A + sgn B as two packed FMAs on the pairs the LDS reads deliver; (v0, v3) = T01 - T23 as one packed subtraction
          // (written on the vector types: as scalars the compiler packs half of them and pays for it in register moves)
          const f32x2w t01 = __builtin_elementwise_fma(sgn2, b01, a01), t23 = __builtin_elementwise_fma(sgn2, b23, a23);
          const f32x2w v03 = t01 - t23;
          float v0 = v03.x, v3 = v03.y, v1 = t01.y + t23.x, v2 = t23.x - t01.y;
          if (WINO_GRU_EXP & 4) { v0 = a01.x; v1 = a01.y; v2 = b23.x; v3 = b23.y; }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            m[rr][nt][0] = mfma16(uf[nt][0][kc], v0, m[rr][nt][0]);
            m[rr][nt][1] = mfma16(uf[nt][1][kc], v1, m[rr][nt][1]);
            m[rr][nt][2] = mfma16(uf[nt][2][kc], v2, m[rr][nt][2]);
            m[rr][nt][3] = mfma16(uf[nt][3][kc], v3, m[rr][nt][3]);
          }
        }
#pragma unroll
      for (int rr = 0; rr < RPR; ++rr)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) drain(m[rr][nt][j]);
      // Z_i[b] = sum_j M[i][j] At[b][j] into the exchange buffer of this round
      f32x4* zb = (f32x4*)(zl + (rd & 1) * (RPR * ZBUF));
#pragma unroll
      for (int rr = 0; rr < RPR; ++rr)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (WINO_GRU_EXP & 8) break;
          zb[rr * (ZBUF / 4) + ((wave * 2 + 0) * NT + nt) * 64 + lane] = (m[rr][nt][0] + m[rr][nt][1]) + m[rr][nt][2];
          zb[rr * (ZBUF / 4) + ((wave * 2 + 1) * NT + nt) * 64 + lane] = (m[rr][nt][1] - m[rr][nt][2]) - m[rr][nt][3];
        }
      if (!(WINO_GRU_EXP & (1 | 8))) __syncthreads();                  // one barrier per round: the buffers alternate
#pragma unroll
      for (int rr = 0; rr < RPR; ++rr) {
      const int tr4 = rd * RPR + rr;
      unsigned oo = ooff + (unsigned)(2 * tr4 * a.wo * CO * 4);
      if (!full && !(oy0 + 2 * tr4 + oa < a.ho && ox0 + 2 * p + ob < a.wo)) oo = BUF_OOB;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 z0 = zb[rr * (ZBUF / 4) + (((oa + 0) * 2 + ob) * NT + nt) * 64 + lane];
        f32x4 z1 = zb[rr * (ZBUF / 4) + (((oa + 1) * 2 + ob) * NT + nt) * 64 + lane];
        f32x4 z2 = zb[rr * (ZBUF / 4) + (((oa + 2) * 2 + ob) * NT + nt) * 64 + lane];
        if (WINO_GRU_EXP & 8) { z0 = (m[rr][nt][0] + m[rr][nt][1]) + m[rr][nt][2]; z1 = z0; z2 = (m[rr][nt][1] - m[rr][nt][2]) - m[rr][nt][3]; }
        const f32x4 v = WINO_GRU_IL ? z0 + os * (z1 + z2) : (z0 + os * (z1 + z2)) + bias[nt];
        const int co4 = nt * 16 + 4 * q;
        if (IL) {                                                      // (r[2q], r[2q+1], u[2q], u[2q+1]) of the lane's pixel
          const f32x4 sg = (WINO_GRU_EXP & 2) ? v : f32x4{sigmoid_pre(v.x), sigmoid_pre(v.y), sigmoid_pre(v.z), sigmoid_pre(v.w)};
          const float* hl = (const float*)((const char*)lds + hbyte + 2 * tr4 * LC * 4);
          const float h0 = hl[0], h1 = hl[PLANE];
          unsigned o2 = ooff2 + (unsigned)(2 * tr4 * a.wo * CO * 4);
          if (oo == BUF_OOB) o2 = BUF_OOB;
          __builtin_amdgcn_raw_buffer_store_b64(u32x2w{__float_as_uint(sg.x * h0), __float_as_uint(sg.y * h1)}, r0, o2, 0, 0);   // r * h (module.py:35-41)
          __builtin_amdgcn_raw_buffer_store_b64(u32x2w{__float_as_uint(sg.z), __float_as_uint(sg.w)}, r1, o2, 0, 0);             // u
        } else if (EPI == EPI_GATES) {
          const f32x4 sg = (WINO_GRU_EXP & 2) ? v : f32x4{sigmoid_pre(v.x), sigmoid_pre(v.y), sigmoid_pre(v.z), sigmoid_pre(v.w)};
          if (co4 < HC) {                                              // reset-gate rows -> r * h (module.py:35-41)
            const float* hl = (const float*)((const char*)lds + hbyte + 2 * tr4 * LC * 4);
            const f32x4 hc = {hl[0], hl[PLANE], hl[2 * PLANE], hl[3 * PLANE]};
            buf_store4(r0, oo == BUF_OOB ? BUF_OOB : oo + nt * 64, sg * hc);
          } else if (co4 < 2 * HC) {                                   // update-gate rows -> u
            buf_store4(r1, oo == BUF_OOB ? BUF_OOB : oo + (unsigned)((co4 - HC - 4 * q) * 4), sg);
          }
        } else if (co4 < HC) {                                         // EPI_CAND (module.py:44-50)
          const f32x4 cnd = {tanh_pre(v.x), tanh_pre(v.y), tanh_pre(v.z), tanh_pre(v.w)};
          const f32x4 u4 = pre_u[EPI == EPI_CAND ? tr4 : 0][nt], h4 = pre_h[EPI == EPI_CAND ? tr4 : 0][nt];
          buf_store4(r0, oo == BUF_OOB ? BUF_OOB : oo + nt * 64, gru_blend(u4, h4, cnd));
        }
      }
      }
    }

    wait_vmem_all();                   // the next window (and this tile's stores)
    __syncthreads();                   // every wave is done with the window
    if (more) store_tile(stage);
    if (!more) break;
    __syncthreads();                   // next tile visible
    t = tn; b = bn; tx = txn; ty = tyn;
  }
  }
};

}  // namespace adamvs
