// Level 1 of SliceCostRegNetRED's ConvGRU (reference models/module.py:24-52: gates on cat(x, h), r * h, candidate on
// cat(x, r * h), blend with u; models/adamvs.py:415-417) as ONE fp32 tile loop with the gate convolution in the
// minimal-filtering form F(2x2, 3x3) INSIDE it, and the tiles of a column strip walked top to bottom so that the gate rows two
// vertically adjacent tiles share are computed once.
//
// STATUS (round 5): correct (tests/test_hip_parity.py::test_fused_level_one_with_minimal_filtering_gates), measured, and SLOWER
// than the two kernels it replaces at every size tried -- 180-189 us per hypothesis against 171 at cfg2 / 128 tiles -- so it runs
// only under ADAMVS_GRU_FUSED=1.  DESIGN.md section 4 ("The fused level-1 kernel") has the three mappings that were built and
// their timing builds; this file is the second one.
//
// Why it was built.  As two launches -- gates1 in the F(2x2, 3x3) form, the candidate in the two-row direct form -- level 1 moves
// 370 B per pixel and step: x and h with their halo twice, r * h and u out and back, h a third time for the blend; the
// candidate kernel alone runs at 6.5 TB/s of fabric traffic.  Round 4's fused level (slice_roles_fused.h) moved 173 B but paid for
// it with the direct-form gates on the tile plus a one-pixel ring: 4.6 MFMAs per pixel against 3.75, and lost.  Here
//   * the gates cost 16 instead of 36 products per 2 x 2 outputs: tile rows 1 .. 4 of the 10 x 32 gate region are ONE WAVE each,
//     with all sixteen positions of the transformed filters in registers (64), whole transformed patches formed by the wave and the
//     output transform At M A in registers -- no exchange and no barrier inside the gate phase; tile row 0 is needed by a segment's
//     first tile only and runs in ConvWinoRole's mapping (waves = patch rows, one exchange through LDS);
//   * the ring is paid in ONE direction only: a workgroup takes a segment of `seg` vertically adjacent tiles of a strip; the
//     bottom tile row of a tile's gate region (region rows 8, 9 = the next tile's rows 0, 1) is kept -- r * h of those two rows and
//     u of the second move up inside LDS -- so every tile after the first of its segment computes four tile rows of gates, not
//     five: 160 (+ 16 for a segment's first tile) MFMAs per wave and 8 x 30 tile against 150 for the two kernels; with direct gates and
//     a full ring it was 276;
//   * the candidate's 48 fragment values per lane wait in LDS (they do not fit next to 64 filter and 64 accumulator registers).
// x and h are read once (halo 2; the rows a tile shares with the one above were read by the same workgroup a tile ago: L2),
// r * h and u never leave the chip, h' is written once: ~110 B per pixel and step.
//
// What it costs: 238 registers and 72 KB of LDS per workgroup = two waves per SIMD, where the separate kernels run four or five;
// its time is its matrix time plus everything else with no overlap between the two, and bytes are not what bounds it (a build
// without window loads is 2 % faster).  The arithmetic is that of the two kernels (same transformed filters, same chains; the gate
// bias rides in accumulator position (1, 1)): maps agree to rounding (asserted <= 2e-6 per step).
#pragma once
#include <type_traits>

#include "slice_roles.h"

// Timing builds only (tools/build_variant.py <name> -DFWINO_EXP=<bits>; results are wrong): 1 no MFMAs (the chains are skipped),
// 2 no transcendentals, 4 no output stores, 8 no window loads after the first tile, 16 no exchange barriers
#ifndef FWINO_EXP
#define FWINO_EXP 0
#endif

namespace adamvs {

struct Gru1WArgs {
  const float* x;        // c1 [B][h*w][8]
  const float* hin;      // state in  [B][h*w][8]
  float* hout;           // state out [B][h*w][8] (a different buffer: neighbouring tiles still read the old one)
  const float* wg;       // gates1: transformed filters U = G (-log2e g) Gt, fragments [1][4 i][4 j][4 kc][64] (FuseWeights::gates1_w)
  const float* bg;       // gates1 bias [16], unscaled
  const float* wc;       // cand1 two-row A fragments [12][4][64]
  const float* bc;       // cand1 bias [16] (8 used)
  int h, w;
  int seg;               // tiles per strip segment (>= 1)
};

struct Gru1WinoFusedRole {
  typedef Gru1WArgs Args;
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  static constexpr int TR = 8, TC = 30, WR = TR + 4, WC = TC + 4, NPIX = WR * WC;      // window of x and h: halo 2
  static constexpr int RR = TR + 2, RC = TC + 2;                                      // gate region (tile + ring): five tile rows of 16 tiles
  static constexpr int KC = 4, G = 6, PLANE = plane_pitch16(NPIX), GP = 4 * PLANE + 8;   // groups: x 0-3, x 4-7, h 0-3, h 4-7, r*h 0-3, r*h 4-7
  static_assert((PLANE % 2) == 0 && (GP % 2) == 0 && (WC % 2) == 0, "8-byte aligned patch reads");
  static constexpr int U0 = G * GP;                                                  // floats: u [RR * RC][8]
  static_assert((U0 % 4) == 0, "16-byte aligned u");
  static constexpr int Z0 = U0 + RR * RC * 8;                                        // the exchange of tile row 0: [4 waves][2 b][64 lanes] float4
  static constexpr int ZBUF = 4 * 2 * 64 * 4;
  static constexpr int W0 = Z0 + ZBUF;                                               // the candidate's fragments [12 taps][64 lanes][4 kc]
  static constexpr int NS = (NPIX * 2 + 255) / 256;                                  // float4 items per source and thread
  static constexpr size_t LDS_BYTES = (size_t)(W0 + 12 * 64 * 4) * sizeof(float);
  static constexpr int TILE_W = TC, TILE_H = TR;
  static int rows_of_tiles(const Args& a) { return cdiv(a.h, TR); }
  static int tiles_x(const Args& a) { return cdiv(a.w, TC); }
  static int tiles_y(const Args& a) { return cdiv(rows_of_tiles(a), a.seg); }        // work items per strip: its segments

  // the gate rows as the lanes want them: MFMA row 4 q + e = reset-gate channel 2 q + e (e = 0, 1) | update-gate channel 2 q + e - 2
  // (e = 2, 3), so that every lane ends with two reset and two update values of the same channels -- applied when the fragments are
  // loaded (a row permutation of A is a lane permutation of its fragments), the blob keeps the reference's order
  static __device__ __forceinline__ int gate_row(int m) { return (m & 2) ? 8 + 2 * (m >> 2) + (m & 1) : 2 * (m >> 2) + (m & 1); }

  static __device__ __forceinline__ void run(const Args& a, const TileGrid& tg, TileRange tr, int wg, int nwg, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;

  // ---- weights.  Gates: ALL sixteen positions of the transformed filters in registers (64): tile rows 1 .. 4 of the region are
  // one wave each, every wave forms whole transformed patches and needs no exchange.  Candidate: its 48 fragment values per lane
  // would not fit next to them and the 64 gate accumulators: they wait in LDS as [tap][lane][4 k-chunks] (one 16-byte read per tap).
  float uall[4][4][KC];
  {
    const int src_lane = (lane & 48) | gate_row(lane & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) uall[i][j][kc] = a.wg[(((i * 4 + j) * KC) + kc) * 64 + src_lane];
  }
  float* wcl = lds + W0;
#pragma unroll
  for (int t = 0; t < 12; ++t) wcl[(t * 64 + lane) * 4 + wave] = a.wc[(t * KC + wave) * 64 + lane];      // wave = k-chunk here
  constexpr float PRE = -1.4426950408889634f;            // the transformed gate filters carry -log2 e; the shared bias is scaled here
  // (the bias rides in accumulator position (1, 1): At E A is all ones for the unit element E11, so the sum started from it carries it
  //  into all four outputs of a tile -- costreg2d_wino.hip does the same)
  const f32x4 bias_g = f32x4{a.bg[2 * q], a.bg[2 * q + 1], a.bg[8 + 2 * q], a.bg[8 + 2 * q + 1]} * PRE;
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
  // gates, tile rows 1 .. 4: wave w takes tile row t = w + 1 (region rows 2 t, 2 t + 1; window rows 2 t .. 2 t + 3 feed it)
  const int trow = wave + 1;
  unsigned pw = (unsigned)((q * PLANE + 2 * trow * WC + 2 * p) * 4);                                       // the patch: + k-chunk, patch row
  unsigned hw_ = (unsigned)(((2 + (q >> 1)) * GP + 2 * (q & 1) * PLANE + (2 * trow + 1) * WC + 2 * p + 1) * 4);   // h of pixel (0, 0) of the lane's 2 x 2 tile; r*h: + 2 GP
  unsigned uw = (unsigned)((U0 + (2 * trow * RC + 2 * p) * 8 + 2 * q) * 4);                                // u of that pixel
  pin(pw); pin(hw_); pin(uw);
  // gates, tile row 0 (a segment's first tile only; the others take it from the tile above): the four waves are the four rows i of
  // the transformed patch and meet through LDS (ConvWinoRole's mapping): T = rowA + sgn rowB (i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3)
  const int rowA = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int rowB = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
  const float sgn = wave == 1 ? 1.0f : -1.0f;
  const f32x2v sgn2 = {sgn, sgn};
  const unsigned pa = (unsigned)((q * PLANE + rowA * WC + 2 * p) * 4), pb = (unsigned)((q * PLANE + rowB * WC + 2 * p) * 4);
  const int oa = wave >> 1, ob = wave & 1;               // wave (oa, ob) finishes pixel (oa, ob) of every 2 x 2 tile of that row
  const float os = oa ? -1.0f : 1.0f;
  const unsigned hgate = (unsigned)(((2 + (q >> 1)) * GP + 2 * (q & 1) * PLANE + (oa + 1) * WC + 2 * p + ob + 1) * 4);
  const unsigned ugate = (unsigned)((U0 + (oa * RC + 2 * p + ob) * 8 + 2 * q) * 4);
  f32x4* zb = (f32x4*)(lds + Z0);
  // candidate (two-row form): the wave's pair-runs are (row pair row0 + 2 j, column run cr), j = 0, 1
  const int row0 = wave >> 1, cr = wave & 1;
  unsigned xc0 = (unsigned)((q * PLANE + (2 * row0 + 1) * WC + cr * 16 + 1 + p) * 4);      // cat(x, r*h): + group 0, 1, 4, 5 (immediates)
  pin(xc0);
  // candidate epilogue: lane = inner pixel (2 pair + (q >> 1), cr * 16 + p), channels 4 (q & 1)..: u of region (row + 1, col + 1),
  // h of window (row + 2, col + 2)
  const int orow0 = 2 * row0 + (q >> 1), ocol = cr * 16 + p;
  const unsigned ucand = (unsigned)((U0 + ((orow0 + 1) * RC + ocol + 1) * 8 + 4 * (q & 1)) * 4);
  const unsigned hcand = (unsigned)(((2 + (q & 1)) * GP + (orow0 + 2) * WC + ocol + 2) * 4);
  unsigned ooff = ocol < TC ? (unsigned)(((orow0 * a.w + ocol) * 8 + 4 * (q & 1)) * 4) : BUF_OOB;
  const unsigned ostep = (unsigned)(4 * a.w * 32);                     // bytes between a wave's two row pairs
  pin(ooff);
  // the rows a tile hands to the one below: r * h of region rows 8, 9 (window rows 9, 10 -> 1, 2), 32 columns x 8 channels = two
  // values per thread; u of region row 9 (-> row 1), one value per thread
  unsigned crh[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + 256 * k, c8 = idx >> 6, rr = (idx >> 5) & 1, col = idx & 31;
    crh[k] = (unsigned)(((4 + (c8 >> 2)) * GP + (c8 & 3) * PLANE + (1 + rr) * WC + 1 + col) * 4);      // destination; source: + 8 WC
  }
  const unsigned cu = (unsigned)((U0 + RC * 8 + tid) * 4);                                               // destination; source: + 8 RC * 8

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
  // sigmoid of the lane's four gate values of one pixel -> r * h into its planes of the window, u into the region buffer
  auto gate_pixel = [&](const f32x4& v, unsigned hbyte, unsigned ubyte) {
    const f32x4 sg = (FWINO_EXP & 2) ? v : f32x4{sigmoid_pre(v.x), sigmoid_pre(v.y), sigmoid_pre(v.z), sigmoid_pre(v.w)};
    float* hl = (float*)((char*)lds + hbyte);
    const float h0 = hl[0], h1 = hl[PLANE];
    hl[2 * GP] = sg.x * h0;
    hl[2 * GP + PLANE] = sg.y * h1;
    *(f32x2v*)((char*)lds + ubyte) = f32x2v{sg.z, sg.w};
  };
  // tile row 0 with the waves as patch rows; I = this wave's row (a template argument: the filters are a register array)
  auto gates_row0 = [&](auto ic) {
    constexpr int I = decltype(ic)::value;
    f32x4 m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = (I == 1 && j == 1) ? bias_g : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int off = kc * GP * 4;
      const f32x2v a01 = *(const f32x2v*)((const char*)lds + pa + off), a23 = *(const f32x2v*)((const char*)lds + pa + off + 8);
      const f32x2v b01 = *(const f32x2v*)((const char*)lds + pb + off), b23 = *(const f32x2v*)((const char*)lds + pb + off + 8);
      const f32x2v t01 = __builtin_elementwise_fma(sgn2, b01, a01), t23 = __builtin_elementwise_fma(sgn2, b23, a23);
      const f32x2v v03 = t01 - t23;
      const float v1 = t01.y + t23.x, v2 = t23.x - t01.y;
      if (!(FWINO_EXP & 1)) {
        m[0] = mfma16(uall[I][0][kc], v03.x, m[0]);
        m[1] = mfma16(uall[I][1][kc], v1, m[1]);
        m[2] = mfma16(uall[I][2][kc], v2, m[2]);
        m[3] = mfma16(uall[I][3][kc], v03.y, m[3]);
      } else {
        m[0].x += v03.x; m[1].x += v1; m[2].x += v2; m[3].x += v03.y;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) drain(m[j]);
    zb[(I * 2 + 0) * 64 + lane] = (m[0] + m[1]) + m[2];                // Z_i[b] = sum_j M[i][j] At[b][j]
    zb[(I * 2 + 1) * 64 + lane] = (m[1] - m[2]) - m[3];
  };

  // ---- the walk: work item = (tile b, strip tx, segment sy); inside it the tiles ty = sy * seg ... top to bottom
  const int nrows = (a.h + TR - 1) / TR;
  int it = tr.begin + wg;
  if (it >= tr.end) return;
  int b, tx, sy;
  tile_coords(tg, it, b, tx, sy);
  int ty = sy * a.seg, ty_end = min(ty + a.seg, nrows);
  bool carried = false;                                                // region rows 0, 1 of this tile came from the tile above
  f32x4 sx[NS], sh[NS];
  load_tile(sx, sh, b, tx, ty);
  wait_vmem_all();
  store_tile(sx, sh);
  __syncthreads();
  for (;;) {
    const int oy0 = ty * TR, ox0 = tx * TC;
    const buf_rsrc rout = make_rsrc((char*)a.hout + (((long)b * a.h + oy0) * a.w + ox0) * 32);
    const bool full = oy0 + TR <= a.h && ox0 + TC <= a.w;
    // the next tile: the one below in this segment (it takes this tile's last gate rows), or the first of the next work item
    int nb = b, ntx = tx, nty = ty + 1, nty_end = ty_end, nit = it;
    bool more = true, hand = true;
    if (nty >= ty_end) {
      nit = it + nwg;
      hand = false;
      more = nit < tr.end;
      if (more) {
        int nsy;
        tile_coords(tg, nit, nb, ntx, nsy);
        nty = nsy * a.seg;
        nty_end = min(nty + a.seg, nrows);
      }
    }
    if (more && !((FWINO_EXP & 8))) load_tile(sx, sh, nb, ntx, nty);   // in flight during both chains

    // ---- gates on cat(x, h) (module.py:35-41), F(2x2, 3x3).  Tile row 0 first, where this tile has to form it itself
    if (!carried) {                                                    // uniform
      switch (wave) {
        case 0: gates_row0(std::integral_constant<int, 0>{}); break;
        case 1: gates_row0(std::integral_constant<int, 1>{}); break;
        case 2: gates_row0(std::integral_constant<int, 2>{}); break;
        default: gates_row0(std::integral_constant<int, 3>{}); break;
      }
      if (!(FWINO_EXP & 16)) __syncthreads();
      const f32x4 z0 = zb[((oa + 0) * 2 + ob) * 64 + lane];            // Y[oa][ob] = sum_i At[oa][i] Z_i[ob]
      const f32x4 z1 = zb[((oa + 1) * 2 + ob) * 64 + lane];
      const f32x4 z2 = zb[((oa + 2) * 2 + ob) * 64 + lane];
      gate_pixel(z0 + os * (z1 + z2), hgate, ugate);
    }
    // tile rows 1 .. 4, one per wave: whole transformed patches, sixteen independent sums
    {
      f32x4 m[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = (i == 1 && j == 1) ? bias_g : f32x4{0.f, 0.f, 0.f, 0.f};
      // the raw patch rows of k-chunk kc + 1 are requested before the products of k-chunk kc (two waves per SIMD do not hide an
      // LDS round trip in front of every sixteen MFMAs)
      f32x2v dn[4][2];
      auto read_patch = [&](int kc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dn[r][0] = *(const f32x2v*)((const char*)lds + pw + (kc * GP + r * WC) * 4);
          dn[r][1] = *(const f32x2v*)((const char*)lds + pw + (kc * GP + r * WC) * 4 + 8);
        }
      };
      read_patch(0);
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        f32x2v d[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) { d[r][0] = dn[r][0]; d[r][1] = dn[r][1]; }
        if (kc + 1 < KC) read_patch(kc + 1);
        __builtin_amdgcn_sched_barrier(0);
        // rows of Bt d: d0 - d2 | d1 + d2 | d2 - d1 | d1 - d3; then the same along each row: (c0 - c2, c1 + c2, c2 - c1, c1 - c3)
        const f32x2v T[4][2] = {{d[0][0] - d[2][0], d[0][1] - d[2][1]}, {d[1][0] + d[2][0], d[1][1] + d[2][1]},
                                {d[2][0] - d[1][0], d[2][1] - d[1][1]}, {d[1][0] - d[3][0], d[1][1] - d[3][1]}};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2v v03 = T[i][0] - T[i][1];
          const float v1 = T[i][0].y + T[i][1].x, v2 = T[i][1].x - T[i][0].y;
          if (!(FWINO_EXP & 1)) {
            m[i][0] = mfma16(uall[i][0][kc], v03.x, m[i][0]);
            m[i][1] = mfma16(uall[i][1][kc], v1, m[i][1]);
            m[i][2] = mfma16(uall[i][2][kc], v2, m[i][2]);
            m[i][3] = mfma16(uall[i][3][kc], v03.y, m[i][3]);
          } else {
            m[i][0].x += v03.x; m[i][1].x += v1; m[i][2].x += v2; m[i][3].x += v03.y;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) drain(m[i][j]);
      // Y = At M A in registers: Z_i[b] = sum_j M[i][j] At[b][j], Y[a][b] = sum_i At[a][i] Z_i[b]
      f32x4 Z[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Z[i][0] = (m[i][0] + m[i][1]) + m[i][2];
        Z[i][1] = (m[i][1] - m[i][2]) - m[i][3];
      }
      float hv[4][2];                                                  // the state values of the four pixels, all requested before the first is used
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const float* hl = (const float*)((const char*)lds + hw_ + ((px >> 1) * WC + (px & 1)) * 4);
        hv[px][0] = hl[0]; hv[px][1] = hl[PLANE];
      }
#pragma unroll
      for (int px = 0; px < 4; ++px) {                                 // pixel (px >> 1, px & 1) of the lane's 2 x 2 tile
        const int aa = px >> 1, bb = px & 1;
        const f32x4 v = aa ? (Z[1][bb] - Z[2][bb]) - Z[3][bb] : (Z[0][bb] + Z[1][bb]) + Z[2][bb];
        const f32x4 sg = (FWINO_EXP & 2) ? v : f32x4{sigmoid_pre(v.x), sigmoid_pre(v.y), sigmoid_pre(v.z), sigmoid_pre(v.w)};
        float* rl = (float*)((char*)lds + hw_ + (aa * WC + bb) * 4) + 2 * GP;
        rl[0] = sg.x * hv[px][0];
        rl[PLANE] = sg.y * hv[px][1];
        *(f32x2v*)((char*)lds + uw + (aa * RC + bb) * 32) = f32x2v{sg.z, sg.w};
      }
    }
    __syncthreads();                   // r * h and u visible

    // ---- candidate on cat(x, r * h) (module.py:44-50), the two-row form: the wave's two pair-runs side by side (two independent sums)
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (!(FWINO_EXP & 1)) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const f32x4 wv = *(const f32x4*)(wcl + ((rr * 3 + kx) * 64 + lane) * 4);
          const float wk[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) {
            const int grp = (kc < 2 ? kc : kc + 2) * GP;
            acc[0] = mfma16(wk[kc], *(const float*)((const char*)lds + xc0 + (grp + rr * WC + kx) * 4), acc[0]);
            acc[1] = mfma16(wk[kc], *(const float*)((const char*)lds + xc0 + (grp + (4 + rr) * WC + kx) * 4), acc[1]);
          }
        }
    }
    drain(acc[0]); drain(acc[1]);
    f32x4 out[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 pre_u = *(const f32x4*)((const char*)lds + ucand + j * 4 * RC * 32);
      const float* hl = (const float*)((const char*)lds + hcand + j * 4 * WC * 4);
      const f32x4 pre_h = {hl[0], hl[PLANE], hl[2 * PLANE], hl[3 * PLANE]};
      const f32x4 v = acc[j] + bias_c;
      const f32x4 cnd = (FWINO_EXP & 2) ? v : f32x4{tanh_fast(v.x), tanh_fast(v.y), tanh_fast(v.z), tanh_fast(v.w)};
      out[j] = pre_u * pre_h + (1.0f - pre_u) * cnd;
    }
    // what the tile below takes over (read before the window is given up, written after)
    float keep_rh[2] = {0.f, 0.f}, keep_u = 0.f;
    if (hand) {
      keep_rh[0] = *(const float*)((const char*)lds + crh[0] + 8 * WC * 4);
      keep_rh[1] = *(const float*)((const char*)lds + crh[1] + 8 * WC * 4);
      keep_u = *(const float*)((const char*)lds + cu + 8 * RC * 32);
    }

    wait_vmem_all();                   // the one wait point of the tile: the next window has had both chains to arrive
    __syncthreads();                   // every wave is done with the window
    if (more) store_tile(sx, sh);
    if (hand) {
      *(float*)((char*)lds + crh[0]) = keep_rh[0];
      *(float*)((char*)lds + crh[1]) = keep_rh[1];
      *(float*)((char*)lds + cu) = keep_u;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned oo = ooff == BUF_OOB ? BUF_OOB : ooff + j * ostep;
      if (!full && !(oy0 + orow0 + 4 * j < a.h && ox0 + ocol < a.w)) oo = BUF_OOB;
      if (!(FWINO_EXP & 4)) buf_store4(rout, oo, out[j]);
    }
    if (!more) break;
    __syncthreads();                   // next tile (and the rows handed down) visible
    it = nit; b = nb; tx = ntx; ty = nty; ty_end = nty_end; carried = hand;
  }
  }
};

}  // namespace adamvs
