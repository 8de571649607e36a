// Both ConvGRU levels of SliceCostRegNetRED as ONE fp32 tile loop each (reference models/module.py:24-52: gates on
// cat(x, h), r * h, candidate on cat(x, r * h), blend with u), the fp32 twins of Gru1FusedBx3Role / Gru2FusedBx3Role.
//
// Why: the gate and the candidate convolution of a level are two DEPENDENT launches per hypothesis (the candidate needs
// r * h on a 3 x 3 neighbourhood), and they exchange r * h and u through memory and read x and h with their halo twice
// (level 1: 393 of the 720 B per pixel and step the fp32 recurrence moves).  Fused, a tile computes the gates on its own
// pixels plus a one-pixel ring, keeps r * h and u in LDS, and reads x and h once (halo 2) -- 173 B per pixel at level 1 --
// at the price of the ring's gate MFMAs (8 x 30 tile: 4.6 MFMAs per pixel against 3.75).  With both levels fused nothing
// inside a step depends on anything else once the levels are skewed by a hypothesis each, so a hypothesis costs ONE
// dependent launch (recurrence.hip, schedule 5): what a stage with few tiles per CU (BASELINE cfg4: four tiles per GPU)
// pays for is launches, not matrix time.
//
// The arithmetic is that of the separate kernels -- the same MFMA chains in the same order from the same fp32 operands
// (r * h and u never leave fp32 here either), the same epilogue expressions -- so every schedule stays bit-identical.
#pragma once
#include "slice_roles.h"

namespace adamvs {

struct Gru1F32Args {
  const float* x;        // c1 [B][h*w][8]
  const float* hin;      // state in  [B][h*w][8]
  float* hout;           // state out [B][h*w][8] (a different buffer: neighbouring tiles still read the old one)
  const float* wg; const float* bg;     // gates1 A fragments [1][9][4][64], bias [16]
  const float* wc; const float* bc;     // cand1 two-row A fragments [12][4][64], bias [16] (8 used)
  int h, w;
};

// Tile = TR rows x (16 NRW - 2) columns; gates on the tile grown by one pixel ((TR + 2) rows x NRW runs of 16), window of
// x and h grown by two.  LDS: six channel groups in the planar layout of ConvSmallRole -- x 0-3, x 4-7, h 0-3, h 4-7,
// r*h 0-3, r*h 4-7 -- then u [region pixel][8].  Wave k takes gate runs k, k + 4, ... and candidate row pairs k, k + 4, ...
// (the two-row form of TwoRowPairRole: MFMA rows 0-7 = the 8 channels of output row y, rows 8-15 = of row y + 1).
//   <8, 2>: 8 x 30, 5 gate runs + 2 row pairs per wave -- balanced, the large-batch tile
//   <4, 2>: 4 x 30, 3 + 1 per wave: twice the tiles with half the chain each, for stages with few tiles per CU
template <int TR, int NRW>
struct Gru1FusedRole {
  typedef Gru1F32Args Args;
  static_assert((TR % 2) == 0 && (NRW == 1 || NRW == 2 || NRW == 4), "row pairs; runs dealt to four waves");
  static constexpr int TC = 16 * NRW - 2, WR = TR + 4, WC = TC + 4, NPIX = WR * WC;
  static constexpr int RR = TR + 2, RCOLS = 16 * NRW;                  // gate region
  static constexpr int KC = 4, G = 6, PLANE = plane_pitch16(NPIX), GP = group_pitch(PLANE, G);
  static constexpr int U0 = (G * GP + 3) & ~3;                         // floats (16-byte aligned): u [RR * RCOLS][8]
  static constexpr int RS = 4 / NRW;                                   // region rows (row pairs) between a wave's consecutive runs
  static constexpr int NGR = RR * NRW, NG = (NGR + 3) / 4;             // gate runs in all / per wave
  static constexpr int NCR = (TR / 2) * NRW, NC = (NCR + 3) / 4;       // candidate pair-runs in all / per wave
  static constexpr int NS = (NPIX * 2 + 255) / 256;                    // float4 items per source and thread
  static constexpr size_t LDS_BYTES = (size_t)(U0 + RR * RCOLS * 8) * sizeof(float);
  static constexpr int TILE_W = TC, TILE_H = TR;
  static int tiles_x(const Args& a) { return cdiv(a.w, TC); }
  static int tiles_y(const Args& a) { return cdiv(a.h, TR); }

  static __device__ __forceinline__ void run(const Args& a, const TileGrid& tg, TileRange tr, int wg, int nwg, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int row0 = wave / NRW, cr = wave % NRW;                        // the wave's first run: region row (row pair), column run

  float wgf[1][9][KC];
  load_wfrag<1, KC>(wgf, a.wg, lane);
  float wcf[12][KC];
#pragma unroll
  for (int t = 0; t < 12; ++t)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) wcf[t][kc] = a.wc[(t * KC + kc) * 64 + lane];
  const f32x4 bias_g = *(const f32x4*)(a.bg + 4 * q);
  const f32x4 bias_c = *(const f32x4*)(a.bc + 4 * (q & 1));

  // ---- per-lane constants
  unsigned goff[NS], lbyte[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int j = min(tid + k * 256, NPIX * 2 - 1);                    // surplus lanes repeat the last item
    const int g = j & 1, pp = j >> 1, r = pp / WC, c = pp % WC;
    goff[k] = (unsigned)(((r * a.w + c) * 8 + 4 * g) * 4);
    lbyte[k] = (unsigned)((g * GP + r * WC + c) * 4);                  // x group g; h group g: + 2 GP
    pin(goff[k]); pin(lbyte[k]);
  }
  unsigned xg[KC], xc[KC];                                             // B-fragment origins of the wave's first gate / candidate run
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    xg[kc] = (unsigned)((kc * GP + q * PLANE + row0 * WC + cr * 16 + p) * 4);                               // cat(x, h): groups 0-3
    xc[kc] = (unsigned)(((kc < 2 ? kc : kc + 2) * GP + q * PLANE + (2 * row0 + 1) * WC + cr * 16 + 1 + p) * 4);   // cat(x, r*h): 0, 1, 4, 5
    pin(xg[kc]); pin(xc[kc]);
  }
  // gate epilogue: lanes q < 2 hold the reset gate of channels 4q.., lanes q >= 2 the update gate of channels 4(q-2)..,
  // of region pixel (row, cr*16 + p) = window pixel (row + 1, cr*16 + p + 1)
  const unsigned hbyte = (unsigned)(((2 + (q & 1)) * GP + (row0 + 1) * WC + cr * 16 + p + 1) * 4);          // h of that pixel; r*h: + 2 GP
  const unsigned ubyte_w = (unsigned)((U0 + (row0 * RCOLS + cr * 16 + p) * 8 + 4 * (q & 1)) * 4);
  // candidate epilogue: lane = inner pixel (2 pair + (q >> 1), cr*16 + p), channels 4 (q & 1)..: u of region (row + 1, col + 1),
  // h of window (row + 2, col + 2)
  const int orow0 = 2 * row0 + (q >> 1), ocol = cr * 16 + p;
  const unsigned ubyte_r = (unsigned)((U0 + ((orow0 + 1) * RCOLS + ocol + 1) * 8 + 4 * (q & 1)) * 4);
  const unsigned pbyte = (unsigned)(((2 + (q & 1)) * GP + (orow0 + 2) * WC + ocol + 2) * 4);
  unsigned ooff = ocol < TC ? (unsigned)(((orow0 * a.w + ocol) * 8 + 4 * (q & 1)) * 4) : BUF_OOB;
  const unsigned ostep = (unsigned)(2 * RS * a.w * 32);                // bytes between a wave's consecutive row pairs
  pin(ooff);

  auto load_tile = [&](f32x4 (&sx)[NS], f32x4 (&sh)[NS], int b, int tx, int ty) {
    const int ix0 = tx * TC - 2, iy0 = ty * TR - 2;
    const long pix0 = ((long)b * a.h + iy0) * a.w + ix0;
    const buf_rsrc rx = make_rsrc((const char*)a.x + pix0 * 32);
    const buf_rsrc rh = make_rsrc((const char*)a.hin + pix0 * 32);
    if (iy0 >= 0 && ix0 >= 0 && iy0 + WR <= a.h && ix0 + WC <= a.w) {
#pragma unroll
      for (int k = 0; k < NS; ++k) { sx[k] = buf_load4(rx, goff[k]); sh[k] = buf_load4(rh, goff[k]); }
    } else {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const int pp = min(tid + k * 256, NPIX * 2 - 1) >> 1;          // (edge tiles only: not worth a register per item)
        const int iy = iy0 + pp / WC, ix = ix0 + pp % WC;
        const unsigned o = ((unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w) ? goff[k] : BUF_OOB;      // zero padding
        sx[k] = buf_load4(rx, o); sh[k] = buf_load4(rh, o);
      }
    }
  };
  auto store_tile = [&](const f32x4 (&sx)[NS], const f32x4 (&sh)[NS]) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      float* dx = (float*)((char*)lds + lbyte[k]);
      float* dh = dx + 2 * GP;
      dx[0] = sx[k].x; dx[PLANE] = sx[k].y; dx[2 * PLANE] = sx[k].z; dx[3 * PLANE] = sx[k].w;
      dh[0] = sh[k].x; dh[PLANE] = sh[k].y; dh[2 * PLANE] = sh[k].z; dh[3 * PLANE] = sh[k].w;
    }
  };

  int t = tr.begin + wg;
  if (t >= tr.end) return;
  int b, tx, ty;
  tile_coords(tg, t, b, tx, ty);
  f32x4 sx[NS], sh[NS];
  load_tile(sx, sh, b, tx, ty);
  wait_vmem_all();
  store_tile(sx, sh);
  __syncthreads();
  for (;;) {
    const int oy0 = ty * TR, ox0 = tx * TC;
    const buf_rsrc rout = make_rsrc((char*)a.hout + (((long)b * a.h + oy0) * a.w + ox0) * 32);
    const bool full = oy0 + TR <= a.h && ox0 + TC <= a.w;
    const int tn = t + nwg;
    const bool more = tn < tr.end;
    int bn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, bn, txn, tyn);
      load_tile(sx, sh, bn, txn, tyn);                                 // in flight during both chains
    }

    // ---- gates on cat(x, h) (module.py:35-41), the chain of ConvSmallRole<8, 8, 1, 1, EPI_GATES>
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      if (4 * j + 3 >= NGR && wave + 4 * j >= NGR) break;             // uniform; only the ragged last round tests
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int kc = 0; kc < KC; ++kc)
            acc = mfma16(wgf[0][ky * 3 + kx][kc], *(const float*)((const char*)lds + xg[kc] + ((j * RS + ky) * WC + kx) * 4), acc);
      drain(acc);
      const f32x4 v = acc + bias_g;
      const f32x4 sg = {sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)};
      if (q < 2) {
        const float* hl = (const float*)((const char*)lds + hbyte + j * RS * WC * 4);
        const f32x4 hc = {hl[0], hl[PLANE], hl[2 * PLANE], hl[3 * PLANE]};
        const f32x4 rh = sg * hc;
        float* dl = (float*)((char*)lds + hbyte + j * RS * WC * 4) + 2 * GP;
        dl[0] = rh.x; dl[PLANE] = rh.y; dl[2 * PLANE] = rh.z; dl[3 * PLANE] = rh.w;
      } else {
        *(f32x4*)((char*)lds + ubyte_w + j * RS * RCOLS * 32) = sg;
      }
    }
    __syncthreads();                   // r * h and u visible

    // ---- candidate on cat(x, r * h) (module.py:44-50), the chain of TwoRowPairRole<TR_CAND>
    f32x4 out[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (4 * j + 3 >= NCR && wave + 4 * j >= NCR) break;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int kc = 0; kc < KC; ++kc)
            acc = mfma16(wcf[rr * 3 + kx][kc], *(const float*)((const char*)lds + xc[kc] + ((2 * j * RS + rr) * WC + kx) * 4), acc);
      drain(acc);
      const f32x4 pre_u = *(const f32x4*)((const char*)lds + ubyte_r + j * 2 * RS * RCOLS * 32);
      const float* hl = (const float*)((const char*)lds + pbyte + j * 2 * RS * WC * 4);
      const f32x4 pre_h = {hl[0], hl[PLANE], hl[2 * PLANE], hl[3 * PLANE]};
      const f32x4 v = acc + bias_c;
      const f32x4 cnd = {tanh_fast(v.x), tanh_fast(v.y), tanh_fast(v.z), tanh_fast(v.w)};
      out[j] = pre_u * pre_h + (1.0f - pre_u) * cnd;
    }

    wait_vmem_all();                   // the one wait point of the tile: the next window has had both chains to arrive
    __syncthreads();                   // every wave is done with the window
    if (more) store_tile(sx, sh);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (4 * j + 3 >= NCR && wave + 4 * j >= NCR) break;
      unsigned oo = ooff == BUF_OOB ? BUF_OOB : ooff + j * ostep;
      if (!full && !(oy0 + orow0 + 2 * j * RS < a.h && ox0 + ocol < a.w)) oo = BUF_OOB;
      buf_store4(rout, oo, out[j]);
    }
    if (!more) break;
    __syncthreads();                   // next tile visible
    t = tn; b = bn; tx = txn; ty = tyn;
  }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Level 2 (cat(conv2 output, state2): 32 channels, 16 hidden, half the stage resolution).  The gate convolution has 32
// output rows = two MFMA row tiles (144 registers of weights next to the candidate's 72): the waves specialise -- waves 0, 1
// the reset-gate rows, waves 2, 3 the update-gate rows, each over every second region row -- and all four share the
// candidate (one 16-pixel run per inner row), as in Gru2FusedBx3Role.
// Tile TR x 14.  LDS: twelve channel groups -- c2 0-15, h 0-15, r*h 0-15 -- then u [region pixel][16].
struct Gru2F32Args {
  const float* x;        // conv2 output [B][h*w][16]   (h, w: the level-2 size)
  const float* hin;      // state in  [B][h*w][16]
  float* hout;           // state out [B][h*w][16] (a different buffer)
  const float* wg; const float* bg;     // gates2 A fragments [2][9][8][64], bias [32]
  const float* wc; const float* bc;     // cand2 A fragments [1][9][8][64], bias [16]
  int h, w;
};

template <int TR>
struct Gru2FusedRole {
  typedef Gru2F32Args Args;
  static constexpr int TC = 14, WR = TR + 4, WC = TC + 4, NPIX = WR * WC;
  static constexpr int RR = TR + 2, RCOLS = 16;
  static constexpr int KC = 8, G = 12, PLANE = plane_pitch16(NPIX), GP = group_pitch(PLANE, G);
  static constexpr int U0 = (G * GP + 3) & ~3;                         // floats (16-byte aligned): u [RR * 16][16]
  static constexpr int NG = (RR + 1) / 2, NC = (TR + 3) / 4;           // gate runs per wave (rows w & 1, + 2, ...), candidate runs per wave
  static constexpr int NS = (NPIX * 4 + 255) / 256;
  static constexpr size_t LDS_BYTES = (size_t)(U0 + RR * RCOLS * 16) * sizeof(float);
  static_assert(LDS_BYTES <= 64 * 1024, "default dynamic LDS limit");
  static constexpr int TILE_W = TC, TILE_H = TR;
  static int tiles_x(const Args& a) { return cdiv(a.w, TC); }
  static int tiles_y(const Args& a) { return cdiv(a.h, TR); }

  static __device__ __forceinline__ void run(const Args& a, const TileGrid& tg, TileRange tr, int wg, int nwg, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int half = wave >> 1, rr0 = wave & 1;          // gate rows (0 reset, 1 update); first region row of the wave

  float wgf[1][9][KC];
  load_wfrag<1, KC>(wgf, a.wg + half * 9 * KC * 64, lane);             // fragment stream [nt][tap][kc][64]: nt = half
  float wcf[1][9][KC];
  load_wfrag<1, KC>(wcf, a.wc, lane);
  const f32x4 bias_g = *(const f32x4*)(a.bg + half * 16 + 4 * q);
  const f32x4 bias_c = *(const f32x4*)(a.bc + 4 * q);

  unsigned goff[NS], lbyte[NS];
  int rc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int j = min(tid + k * 256, NPIX * 4 - 1);
    const int g = j & 3, pp = j >> 2, r = pp / WC, c = pp % WC;
    goff[k] = (unsigned)(((r * a.w + c) * 16 + 4 * g) * 4);
    lbyte[k] = (unsigned)((g * GP + r * WC + c) * 4);                  // c2 group g; h group g: + 4 GP
    rc[k] = r | (c << 16);
    pin(goff[k]); pin(lbyte[k]); pin(rc[k]);
  }
  unsigned xg[KC], xc[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    xg[kc] = (unsigned)((kc * GP + q * PLANE + rr0 * WC + p) * 4);                                  // cat(c2, h): groups 0-7
    xc[kc] = (unsigned)(((kc < 4 ? kc : kc + 4) * GP + q * PLANE + (wave + 1) * WC + 1 + p) * 4);   // cat(c2, r*h): 0-3, 8-11
    pin(xg[kc]); pin(xc[kc]);
  }
  // gate epilogue: the lane holds channels 4q.. of its half's gate at region pixel (row, p) = window (row + 1, p + 1)
  const unsigned hbyte = (unsigned)(((4 + q) * GP + (rr0 + 1) * WC + p + 1) * 4);                  // h of that pixel; r*h: + 4 GP
  const unsigned ubyte_w = (unsigned)((U0 + (rr0 * RCOLS + p) * 16 + 4 * q) * 4);
  // candidate epilogue: inner pixel (wave + 4j, p), channels 4q..
  const unsigned ubyte_r = (unsigned)((U0 + ((wave + 1) * RCOLS + p + 1) * 16 + 4 * q) * 4);
  const unsigned pbyte = (unsigned)(((4 + q) * GP + (wave + 2) * WC + p + 2) * 4);
  unsigned ooff = p < TC ? (unsigned)(((wave * a.w + p) * 16 + 4 * q) * 4) : BUF_OOB;
  const unsigned ostep = (unsigned)(4 * a.w * 64);
  pin(ooff);

  auto load_tile = [&](f32x4 (&sx)[NS], f32x4 (&sh)[NS], int b, int tx, int ty) {
    const int ix0 = tx * TC - 2, iy0 = ty * TR - 2;
    const long pix0 = ((long)b * a.h + iy0) * a.w + ix0;
    const buf_rsrc rx = make_rsrc((const char*)a.x + pix0 * 64);
    const buf_rsrc rh = make_rsrc((const char*)a.hin + pix0 * 64);
    if (iy0 >= 0 && ix0 >= 0 && iy0 + WR <= a.h && ix0 + WC <= a.w) {
#pragma unroll
      for (int k = 0; k < NS; ++k) { sx[k] = buf_load4(rx, goff[k]); sh[k] = buf_load4(rh, goff[k]); }
    } else {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const int iy = iy0 + (rc[k] & 0xffff), ix = ix0 + (rc[k] >> 16);
        const unsigned o = ((unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w) ? goff[k] : BUF_OOB;
        sx[k] = buf_load4(rx, o); sh[k] = buf_load4(rh, o);
      }
    }
  };
  auto store_tile = [&](const f32x4 (&sx)[NS], const f32x4 (&sh)[NS]) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      float* dx = (float*)((char*)lds + lbyte[k]);
      float* dh = dx + 4 * GP;
      dx[0] = sx[k].x; dx[PLANE] = sx[k].y; dx[2 * PLANE] = sx[k].z; dx[3 * PLANE] = sx[k].w;
      dh[0] = sh[k].x; dh[PLANE] = sh[k].y; dh[2 * PLANE] = sh[k].z; dh[3 * PLANE] = sh[k].w;
    }
  };

  int t = tr.begin + wg;
  if (t >= tr.end) return;
  int b, tx, ty;
  tile_coords(tg, t, b, tx, ty);
  f32x4 sx[NS], sh[NS];
  load_tile(sx, sh, b, tx, ty);
  wait_vmem_all();
  store_tile(sx, sh);
  __syncthreads();
  for (;;) {
    const int oy0 = ty * TR, ox0 = tx * TC;
    const buf_rsrc rout = make_rsrc((char*)a.hout + (((long)b * a.h + oy0) * a.w + ox0) * 64);
    const bool full = oy0 + TR <= a.h && ox0 + TC <= a.w;
    const int tn = t + nwg;
    const bool more = tn < tr.end;
    int bn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, bn, txn, tyn);
      load_tile(sx, sh, bn, txn, tyn);
    }

    // ---- the wave's gate rows on cat(c2, h): the chain of ConvSmallRole<16, 16, 2, 1, EPI_GATES> for its row tile
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      if (2 * j + 1 >= RR && rr0 + 2 * j >= RR) break;                // uniform (RR is even: never taken)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int kc = 0; kc < KC; ++kc)
            acc = mfma16(wgf[0][ky * 3 + kx][kc], *(const float*)((const char*)lds + xg[kc] + ((2 * j + ky) * WC + kx) * 4), acc);
      drain(acc);
      const f32x4 v = acc + bias_g;
      const f32x4 sg = {sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)};
      if (half == 0) {
        const float* hl = (const float*)((const char*)lds + hbyte + 2 * j * WC * 4);
        const f32x4 hc = {hl[0], hl[PLANE], hl[2 * PLANE], hl[3 * PLANE]};
        const f32x4 rh = sg * hc;
        float* dl = (float*)((char*)lds + hbyte + 2 * j * WC * 4) + 4 * GP;
        dl[0] = rh.x; dl[PLANE] = rh.y; dl[2 * PLANE] = rh.z; dl[3 * PLANE] = rh.w;
      } else {
        *(f32x4*)((char*)lds + ubyte_w + 2 * j * RCOLS * 64) = sg;
      }
    }
    __syncthreads();                   // r * h and u visible

    // ---- candidate on cat(c2, r * h): the chain of ConvSmallRole<16, 16, 1, 1, EPI_CAND>
    f32x4 out[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (4 * j + 3 >= TR && wave + 4 * j >= TR) break;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int kc = 0; kc < KC; ++kc)
            acc = mfma16(wcf[0][ky * 3 + kx][kc], *(const float*)((const char*)lds + xc[kc] + ((4 * j + ky) * WC + kx) * 4), acc);
      drain(acc);
      const f32x4 u4 = *(const f32x4*)((const char*)lds + ubyte_r + 4 * j * RCOLS * 64);
      const float* hl = (const float*)((const char*)lds + pbyte + 4 * j * WC * 4);
      const f32x4 h4 = {hl[0], hl[PLANE], hl[2 * PLANE], hl[3 * PLANE]};
      const f32x4 v = acc + bias_c;
      const f32x4 cnd = {tanh_fast(v.x), tanh_fast(v.y), tanh_fast(v.z), tanh_fast(v.w)};
      out[j] = u4 * h4 + (1.0f - u4) * cnd;
    }

    wait_vmem_all();
    __syncthreads();                   // every wave is done with the window
    if (more) store_tile(sx, sh);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (4 * j + 3 >= TR && wave + 4 * j >= TR) break;
      unsigned oo = ooff == BUF_OOB ? BUF_OOB : ooff + j * ostep;
      if (!full && !(oy0 + wave + 4 * j < a.h && ox0 + p < a.w)) oo = BUF_OOB;
      buf_store4(rout, oo, out[j]);
    }
    if (!more) break;
    __syncthreads();                   // next tile visible
    t = tn; b = bn; tx = txn; ty = tyn;
  }
  }
};

}  // namespace adamvs
