// Stride-1 layers of CostRegNet2D (reference models/adamvs.py:198-238: conv0, conv2, conv4, conv6, prob; blocks
// models/module.py:254-261) in the minimal-filtering form F(2x4, 3x3): a 2 x 4 output tile from a 4 x 6 input patch,
// 24 products per (cin, cout) pair instead of the direct form's 72 (F(2x2, 3x3), costreg2d_wino.hip: 32 for the same
// eight outputs), fp32 throughout.
//     Y = At2 [ (G2 g G4t) .* (Bt2 d B4) ] A4,       d = the 4 x 6 input patch, g = the 3 x 3 filter
//     rows    (F(2,3)): Bt2 = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G2 = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  At2 = [1 1 1 0; 0 1 -1 -1]
//     columns (F(4,3)): Bt4 = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1],
//                       G4  = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1],
//                       At4 = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// Why 2 x 4 and not 4 x 4 (36 products per sixteen outputs): this kernel family gives a WAVE one row i of the transformed
// patch -- its row of Bt d B costs a handful of packed vector instructions, and every transformed value feeds MT MFMAs --
// and a workgroup one wave per SIMD.  Six patch rows do not divide over four SIMDs; three rows per wave (18 positions)
// leave accumulator registers for one tile row and two channel tiles, i.e. one MFMA per weight fragment fetched from L2
// (32 B per CU and clock).  With four rows of six positions the wave keeps 6 x MT 4 x NT 2 = 48 accumulator tiles -- the
// 192 registers of the F(2x2, 3x3) kernel -- and the same 48 MFMAs per k-step, for 4 x 64 instead of 6 x 32 output pixels.
// Rounding: the column transform has the factors 4, 5, 8 and 1/24: 9e-7 relative L1 against a float64 convolution at
// D = 192 where the direct kernel has 6e-7 and F(2x2, 3x3) 4e-7 (tools/wino_bench.py, tools/experiments/wino24/wino24.py --check-gpu).
//
// OUTCOME (round 4): a third fewer MFMAs, 3.5 % less time.  512 maps of 96 x 192 pixels at D = 192: 23.3 ms against 24.1 for
// F(2x2, 3x3) -- 57 % of the fp32 matrix rate against 73 %; slower on every smaller level of the hourglass (48x96: 7.4 against
// 6.0 ms) and at D = 64.  Per k-step the wave issues the same 48 MFMAs (0.64 us) next to 18 packed transforms, 12 LDS reads,
// 8 fragment loads and 9 fill instructions where F(2x2, 3x3) has 12 / 12 / 4 / 6: 0.33 us against 0.17 on top of the MFMAs,
// and 7.4 against 3.3 us per tile for start + epilogue.  Timing builds (-DWINO24_EXP, of 5.84 ms per 128 maps): no epilogue -0.53,
// no fragment loads -0.68, no transform -0.71, no window fill and barrier -0.97, no raw-row reads -0.57 -- every part a tenth,
// none dominant.  adamvs_cost_reg_net_2d therefore stays on F(2x2, 3x3); this kernel is an op-level entry point with its tests.
//
// Mapping (as costreg2d_wino.hip unless said): workgroup = 4 waves = patch rows i; a tile = NT 2 tile rows x 16 tiles x
// MT 4 channel tiles = 4 x 64 output pixels x 64 channels; LDS window 6 x 66 pixels x 16 input channels, double-buffered,
// one barrier per chunk.  Per tile row and k-step a wave reads two raw rows of six (six ds_read_b64), forms
// T = rowA + sgn rowB (three packed FMAs) and the six column values in six more:
//     (v0, v5) = 4 (t0, t1) - 5 (t2, t3) + (t4, t5)
//     (a, c) = t4 - (4, 1) t2,  (b, e) = t3 - (4, 1) t1,   (v1, v2) = a +- b,   (v3, v4) = c +- 2 e
// -- every operand a register pair as it was loaded (op_sel picks the half), nine packed instructions per 24 MFMAs.
// Weight fragments: per (k-step, row i, channel tile) a lane holds the six positions of its row, fetched as 16 + 8 bytes
// from two arrays.  Epilogue: Z_i[b] = sum_j At4[b][j] M[i][j] (b = 0..3) in registers, through LDS (64 KB: 4 rows x 4 b x MT
// x 1 KB), wave (a, b') forms Y[a][b] = sum_i At2[a][i] Z_i[b] for b = 2 b', 2 b' + 1, adds bias / ReLU / skip and stores.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "persistent.h"

// Timing builds only (tools/build_variant.py <name> -DWINO24_EXP=<bits>; results are wrong): 1 no epilogue, 2 no weight loads in the loop,
// 4 no input transform, 8 no window fill / barrier (32 no barrier only, 64 no LDS stores only, 128 no window loads only), 256 no raw-row reads.
#ifndef WINO24_EXP
#define WINO24_EXP 0
#endif

namespace adamvs {

struct Wino24Args {
  const float* in;     // [N][h*w][D]
  const float* wpk;    // [D/4][4][D/16][64][4] (patch columns 0-3) then [D/4][4][D/16][64][2] (columns 4, 5): k-step, patch row i, channel tile, lane (A-fragment order)
  const float* bias;   // [D]
  const float* skip;   // [N][h*w][D] or null; added after the ReLU
  float* out;          // [N][h*w][D]
  int D, h, w, relu;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MT, int NT>
struct Wino24Geom {
  static constexpr int KC = 16, KS = KC / 4;                 // input channels per LDS chunk, k-steps per chunk
  static constexpr int LR = 2 * NT + 2, LC = 66, NPIX = LR * LC;
  // == 2 (mod 4): a lane reads 8 bytes at pixel 4 p -- banks 4p, 4p+1 -- and the k-rows q, q + 1 of a 32-lane ds_read_b64 group take the other two of every four
  static constexpr int PLANE = (NPIX + 3) / 4 * 4 + 2;
  static constexpr int GP = (4 * PLANE + 63) / 64 * 64 + 16; // channel-group pitch: the 4 groups of 16 neighbouring pixels of a fill on 64 different banks
  static constexpr int CHUNK = KS * GP;                      // floats per buffer
  static constexpr int ZFLOATS = 4 * 4 * MT * 64 * 4;        // epilogue exchange of one tile row
  static constexpr int LDS_FLOATS = (2 * CHUNK > ZFLOATS) ? 2 * CHUNK : ZFLOATS;
  static_assert(PLANE >= NPIX && LDS_FLOATS * 4 <= 65536, "plane pitch / LDS");
};

template <int MT, int NT>
__global__ __launch_bounds__(256) void k_conv_wino24(Wino24Args a, TileGrid tg, int groups, unsigned mgroups) {
  using G = Wino24Geom<MT, NT>;
  constexpr int KC = G::KC, KS = G::KS, LC = G::LC, NPIX = G::NPIX, PLANE = G::PLANE, GP = G::GP, CHUNK = G::CHUNK;
  constexpr int NITEMS = NPIX * (KC / 4), NITA = (NITEMS + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = patch row i
  const int p = lane & 15, q = lane >> 4;
  const int D = a.D, NTILES = D / 16, NC = D / KC;

  // ---- per-lane constants (tile-independent)
  // window fill: item = (pixel of the window, group of 4 channels of the chunk); lane tid holds the items tid + 256 it: the same
  // channel group g for all of them and the pixels pp0 + 64 it -- LDS addresses one pinned base + immediates, global offsets
  // formed per tile (window_of) from (row, column) = pp / LC, pp % LC.  The last round's surplus lanes repeat the last item.
  constexpr int LAST = NITEMS - 1 - 256 * (NITA - 1);        // last item of the last round
  const int fg = tid & 3, pp0 = tid >> 2;
  const int ppl = tid <= LAST ? pp0 + 64 * (NITA - 1) : NPIX - 1;
  unsigned xlds0 = (unsigned)((fg * GP + pp0) * 4), xldsl = (unsigned)(((tid <= LAST ? fg : KC / 4 - 1) * GP + ppl) * 4);
  unsigned gch = (unsigned)((tid <= LAST ? fg : KC / 4 - 1) * 16);   // the last round's channel-group byte offset
  pin(xlds0); pin(xldsl);
  // raw patch rows of the wave: T = rowA + sgn * rowB  (i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3)
  const int rowA = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int rowB = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
  const float sgn = wave == 1 ? 1.0f : -1.0f;
  const f32x2 sgn2 = {sgn, sgn};
  f32x2 k41 = {-4.0f, -1.0f}, kpm = {1.0f, -1.0f}, kpm2 = {2.0f, -2.0f}, k4 = {4.0f, 4.0f}, k5 = {-5.0f, -5.0f};
  // three separately pinned bases per row: merged into ds_read2_b64 the pairs would need a vector add per read for the address
  unsigned pa0 = (unsigned)((q * PLANE + rowA * LC + 4 * p) * 4), pb0 = (unsigned)((q * PLANE + rowB * LC + 4 * p) * 4);
  unsigned pa1 = pa0 + 8, pa2 = pa0 + 16, pb1 = pb0 + 8, pb2 = pb0 + 16;
  pin(pa0); pin(pa1); pin(pa2); pin(pb0); pin(pb1); pin(pb2);
  const buf_rsrc rw = make_rsrc(a.wpk);
  const unsigned wsecond = (unsigned)D * (unsigned)D * 16u * 4u;     // byte offset of the array of patch columns 4, 5
  unsigned woff = (unsigned)(lane * 16), woff2 = wsecond + (unsigned)(lane * 8);
  pin(woff); pin(woff2);
  // epilogue: wave (oa, ob) stores output pixels (2 ty + oa, 4 tx + 2 ob + {0, 1}) of every 2 x 4 tile
  const int oa = wave >> 1, ob = wave & 1;
  const float os = oa ? -1.0f : 1.0f;
  unsigned ooff = (unsigned)(((oa * a.w + 4 * p + 2 * ob) * D + 4 * q) * 4);   // + the tile's origin and the tile row
  pin(ooff);

  struct Tile { int n, r0, c0, cg; };
  auto decode = [&](int t) {
    int n, tx, ty;
    tile_coords(tg, t, n, tx, ty);                           // tx runs over (block column, channel group)
    const int bx = groups == 1 ? tx : (int)__umulhi((unsigned)tx, mgroups);
    return Tile{n, ty * 2 * NT, bx * 64, tx - bx * groups};
  };
  buf_rsrc rx;
  unsigned xoff[NITA];
  auto window_of = [&](const Tile& t) {                      // out-of-image pixels read as zero (BUF_OOB)
    rx = make_rsrc((const char*)a.in + (((long)t.n * a.h + (t.r0 - 1)) * a.w + (t.c0 - 1)) * (long)D * 4);
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
      const int pp = it + 1 < NITA ? pp0 + 64 * it : ppl, r = pp / LC, c = pp - r * LC;
      const bool ok = (unsigned)(t.r0 - 1 + r) < (unsigned)a.h && (unsigned)(t.c0 - 1 + c) < (unsigned)a.w;
      xoff[it] = ok ? (unsigned)((r * a.w + c) * D * 4) + (it + 1 < NITA ? (unsigned)(fg * 16) : gch) : BUF_OOB;
    }
  };

  f32x4 acc[6][MT][NT];
  f32x4 xs[NITA];
  struct Frag { f32x4 lo[MT]; f32x2 hi[MT]; };               // positions 0-3 | 4, 5 of the wave's row
  Frag wf0, wf1, wf2, wf3;                                   // four named fragment sets (below)
  f32x4 bias4[MT];
  int cg = 0;                                                // channel group of the fragments being requested

  auto load_w = [&](Frag& wf, int ks) {                      // ks: global k-step
    if ((WINO24_EXP & 2) && ks > 3) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned frag = (unsigned)((ks * 4 + wave) * NTILES + cg * MT + mt);                          // uniform
      wf.lo[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, frag * 1024u, 0));
      wf.hi[mt] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rw, woff2, frag * 512u, 0));
    }
  };
  auto load_x = [&](int ch) {
#pragma unroll
    for (int it = 0; it < NITA; ++it)
      xs[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[it], (unsigned)ch * 4u, 0));
  };
  auto load_bias = [&]() {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bias4[mt] = *(const f32x4*)(a.bias + (cg * MT + mt) * 16 + 4 * q);
  };
  auto store_items = [&](int buf, int i0, int i1) {          // window items [i0, i1) of the staged chunk -> buffer `buf`
#pragma unroll
    for (int it = i0; it < i1; ++it) {
      float* dl = (float*)((char*)lds + (it + 1 < NITA ? xlds0 : xldsl)) + buf * CHUNK + (it + 1 < NITA ? 64 * it : 0);
      dl[0] = xs[it].x; dl[PLANE] = xs[it].y; dl[2 * PLANE] = xs[it].z; dl[3 * PLANE] = xs[it].w;
    }
  };
  struct Raw { f32x2 a01, a23, a45, b01, b23, b45; };
  auto read_raw = [&](Raw& r, int buf, int ks, int t) {
    const int off = (buf * CHUNK + ks * GP + 2 * t * LC) * 4;
    r.a01 = *(const f32x2*)((const char*)lds + pa0 + off); r.a23 = *(const f32x2*)((const char*)lds + pa1 + off);
    r.a45 = *(const f32x2*)((const char*)lds + pa2 + off);
    r.b01 = *(const f32x2*)((const char*)lds + pb0 + off); r.b23 = *(const f32x2*)((const char*)lds + pb1 + off);
    r.b45 = *(const f32x2*)((const char*)lds + pb2 + off);
  };
  // one tile row of one k-step: 9 packed vector instructions, 6 MT MFMAs.  FIRST: the first k-step of a tile starts the sums
  auto tile_row = [&](const Frag& wf, const Raw& r, int t, auto firstc) {
    constexpr bool FIRST = decltype(firstc)::value;
    const f32x2 t01 = __builtin_elementwise_fma(sgn2, r.b01, r.a01), t23 = __builtin_elementwise_fma(sgn2, r.b23, r.a23),
                t45 = __builtin_elementwise_fma(sgn2, r.b45, r.a45);
    f32x2 v05 = __builtin_elementwise_fma(k4, t01, __builtin_elementwise_fma(k5, t23, t45));
    const f32x2 ac = __builtin_elementwise_fma(__builtin_shufflevector(t23, t23, 0, 0), k41, __builtin_shufflevector(t45, t45, 0, 0));
    const f32x2 be = __builtin_elementwise_fma(__builtin_shufflevector(t01, t01, 1, 1), k41, __builtin_shufflevector(t23, t23, 1, 1));
    f32x2 v12 = __builtin_elementwise_fma(__builtin_shufflevector(be, be, 0, 0), kpm, __builtin_shufflevector(ac, ac, 0, 0));
    f32x2 v34 = __builtin_elementwise_fma(__builtin_shufflevector(be, be, 1, 1), kpm2, __builtin_shufflevector(ac, ac, 1, 1));
    if (WINO24_EXP & 4) { v05 = r.a01; v12 = r.a23; v34 = r.b45; }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[0][mt][t] = mfma16(wf.lo[mt].x, v05.x, FIRST ? zero : acc[0][mt][t]);
      acc[1][mt][t] = mfma16(wf.lo[mt].y, v12.x, FIRST ? zero : acc[1][mt][t]);
      acc[2][mt][t] = mfma16(wf.lo[mt].z, v12.y, FIRST ? zero : acc[2][mt][t]);
      acc[3][mt][t] = mfma16(wf.lo[mt].w, v34.x, FIRST ? zero : acc[3][mt][t]);
      acc[4][mt][t] = mfma16(wf.hi[mt].x, v34.y, FIRST ? zero : acc[4][mt][t]);
      acc[5][mt][t] = mfma16(wf.hi[mt].y, v05.y, FIRST ? zero : acc[5][mt][t]);
    }
  };
  // one k-step (4 input channels): the raw rows of the NEXT tile row (or of the next k-step's first) are requested before the
  // MFMAs of the current one.  FILL: the staged chunk c+1 goes to the other buffer between the tile rows, then chunk c+2 is requested.
  auto kstep = [&](const Frag& wf, Raw& r, int buf, int ks, bool fill, int next_ch, auto firstc) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      Raw nx;
      const bool more = t + 1 < NT || ks + 1 < KS;
      if (more && !(WINO24_EXP & 256)) read_raw(nx, buf, t + 1 < NT ? ks : ks + 1, t + 1 < NT ? t + 1 : 0);
      __builtin_amdgcn_sched_barrier(0);
      tile_row(wf, r, t, firstc);
      if (fill && !(WINO24_EXP & (8 | 64))) store_items(buf ^ 1, t * NITA / NT, (t + 1) * NITA / NT);
      __builtin_amdgcn_sched_barrier(0);
      if (more && !(WINO24_EXP & 256)) r = nx;
    }
    if (fill && !(WINO24_EXP & (8 | 128))) load_x(next_ch);
  };
  const int last_ks = NC * KS - 1;
  static_assert(KS == 4, "four k-steps per chunk, one per fragment set");
  auto chunk = [&](int c, auto curc, auto firstc, bool fill, int next_ch) {
    constexpr int CUR = decltype(curc)::value;
    Raw r;
    read_raw(r, CUR, 0, 0);
    load_w(wf3, min(c * KS + 3, last_ks));
    kstep(wf0, r, CUR, 0, fill, next_ch, firstc);
    load_w(wf0, min(c * KS + 4, last_ks));
    kstep(wf1, r, CUR, 1, false, 0, std::false_type{});
    load_w(wf1, min(c * KS + 5, last_ks));
    kstep(wf2, r, CUR, 2, false, 0, std::false_type{});
    load_w(wf2, min(c * KS + 6, last_ks));
    kstep(wf3, r, CUR, 3, false, 0, std::false_type{});
    if (!(WINO24_EXP & (8 | 32))) __syncthreads();
  };

  // ---- tile loop (persistent grid, as k_conv_wino)
  // Workgroups i, i + 8, ... share an XCD and its L2 (cdna_hip_programming.md T1): give them NEIGHBOURING tiles of every sweep
  // -- the channel groups of a pixel block and the blocks above and below it, which read the same window -- instead of every
  // eighth one.  (Strided, the three groups of a block met in three different L2s: 2 FETCH + WRITE = 9 x the layer's input.)
  const int nwg = (int)gridDim.x, xq = nwg / 8, xr = nwg % 8, xcd = (int)blockIdx.x % 8;
  int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (int)blockIdx.x / 8;
  Tile cur = decode(tile);
  window_of(cur);
  cg = cur.cg;
  load_x(0);
  load_w(wf0, 0);
  load_w(wf1, min(1, last_ks));
  load_w(wf2, min(2, last_ks));
  wait_vmem_all();
  while (true) {
    const int next = tile + (int)gridDim.x;
    const bool more = next < tg.ntiles;
    store_items(0, 0, NITA);
    load_x(KC);
    __syncthreads();
    chunk(0, std::integral_constant<int, 0>{}, std::true_type{}, true, 2 * KC);
    chunk(1, std::integral_constant<int, 1>{}, std::false_type{}, true, 3 * KC);
    for (int c = 2; c < NC - 2; c += 2) {                    // NC is even (>= 4) for every supported D
      chunk(c, std::integral_constant<int, 0>{}, std::false_type{}, true, (c + 2) * KC);
      chunk(c + 1, std::integral_constant<int, 1>{}, std::false_type{}, true, (c + 3) * KC);
    }
    const Tile done = cur;
    if (more) {
      cur = decode(next);
      window_of(cur);
    }
    chunk(NC - 2, std::integral_constant<int, 0>{}, std::false_type{}, true, 0);
    chunk(NC - 1, std::integral_constant<int, 1>{}, std::false_type{}, false, 0);
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) drain(acc[j][mt][0]);   // (tile row 0 is read first; tools/mfma_hazard_lint.py checks the rest)

    cg = done.cg;
    load_bias();                                             // before the next tile's fragment sets: vmcnt retires in order
    cg = cur.cg;
    if (more) {
      load_w(wf0, 0);
      load_w(wf1, min(1, last_ks));
      load_w(wf2, min(2, last_ks));
    }
    // ---- epilogue of `done`: the four rows meet through LDS, one tile row per round
    if (!((WINO24_EXP & 1) && a.relu != 12345)) {
      const long img = (long)done.n * a.h * a.w * (long)D * 4;
      const buf_rsrc ro = make_rsrc((char*)a.out + img);
      const buf_rsrc rk = make_rsrc((const char*)(a.skip ? a.skip : a.out) + img);
      f32x4* zl = (f32x4*)lds;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t) __syncthreads();                                // the previous round's readers are done
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 s12 = acc[1][mt][t] + acc[2][mt][t], d12 = acc[1][mt][t] - acc[2][mt][t];
          const f32x4 s34 = acc[3][mt][t] + acc[4][mt][t], d34 = acc[3][mt][t] - acc[4][mt][t];
          zl[((wave * 4 + 0) * MT + mt) * 64 + lane] = (acc[0][mt][t] + s12) + s34;
          zl[((wave * 4 + 1) * MT + mt) * 64 + lane] = d12 + 2.0f * d34;
          zl[((wave * 4 + 2) * MT + mt) * 64 + lane] = s12 + 4.0f * s34;
          zl[((wave * 4 + 3) * MT + mt) * 64 + lane] = (d12 + 8.0f * d34) + acc[5][mt][t];
        }
        __syncthreads();
        const int oy = done.r0 + 2 * t + oa, ox = done.c0 + 4 * p + 2 * ob;
        const unsigned tbase = ooff + (unsigned)((((done.r0 + 2 * t) * a.w + done.c0) * D + done.cg * MT * 16) * 4);
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const unsigned obase = (oy < a.h && ox + bb < a.w) ? tbase + (unsigned)(bb * D * 4) : BUF_OOB;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 z0 = zl[(((oa + 0) * 4 + 2 * ob + bb) * MT + mt) * 64 + lane];
            const f32x4 z1 = zl[(((oa + 1) * 4 + 2 * ob + bb) * MT + mt) * 64 + lane];
            const f32x4 z2 = zl[(((oa + 2) * 4 + 2 * ob + bb) * MT + mt) * 64 + lane];
            f32x4 v = z0 + os * (z1 + z2) + bias4[mt];
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (a.skip) v += buf_load4(rk, obase == BUF_OOB ? BUF_OOB : obase + mt * 64);
            buf_store4(ro, obase == BUF_OOB ? BUF_OOB : obase + mt * 64, v);
          }
        }
      }
    }
    if (!more) break;
    tile = next;
    // the window chunk and the fragments were requested BEFORE the epilogue's 2 NT MT stores (vmcnt retires in order)
    wait_vmem_but<2 * NT * MT>();
    __syncthreads();                                         // the epilogue's LDS readers are done: the buffers are free
  }
}

template <int MT, int NT>
static int launch_wino24_cfg(const Wino24Args& a, int N, hipStream_t st) {
  const int groups = a.D / (16 * MT);
  auto kern = k_conv_wino24<MT, NT>;
  static const int capacity = resident_blocks(kern, 256, 0);
  TileGrid tg;
  if (int rc = make_tile_grid(tg, groups * cdiv(a.w, 64), cdiv(a.h, 2 * NT), N)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a, tg, groups, (unsigned)(((1ull << 32) + groups - 1) / groups));
  ADAMVS_CHECK_LAUNCH("conv_wino24");
  return 0;
}

int launch_conv_wino24(const float* in, const float* wpk, const float* bias, const float* skip, float* out, int N, int D, int h, int w,
                       int relu, hipStream_t st) {
  const Wino24Args a{in, wpk, bias, skip, out, D, h, w, relu};
  ADAMVS_CHECK_ARG(wino_depth_supported(D), "conv_wino24: D=%d unsupported (a multiple of 64 up to 512)", D);
  ADAMVS_CHECK_ARG((size_t)h * w * D * 4 < 0x7fffffffu, "conv_wino24: a map of %dx%dx%d floats exceeds the 2 GiB a buffer descriptor spans", h, w, D);
  return launch_wino24_cfg<4, 2>(a, N, st);
}

}  // namespace adamvs

using namespace adamvs;

extern "C" int adamvs_conv3x3_dd_wino24(const float* in, const float* wpk, const float* bias, const float* skip, float* out, int N,
                                        int D, int h, int w, int relu, void* stream) {
  ADAMVS_CHECK_ARG(in && wpk && bias && out && N > 0 && h > 0 && w > 0, "conv3x3_dd_wino24: bad arguments");
  return launch_conv_wino24(in, wpk, bias, skip, out, N, D, h, w, relu, (hipStream_t)stream);
}
