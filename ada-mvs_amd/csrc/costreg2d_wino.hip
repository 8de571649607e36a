// Stride-1 layers of CostRegNet2D (reference models/adamvs.py:198-238: conv0, conv2, conv4, conv6, prob; blocks
// models/module.py:254-261) in the minimal-filtering form F(2x2, 3x3) of a 3x3 convolution, fp32 throughout.
//
// The direct kernel (costreg2d.hip) runs at 92 % of the fp32 matrix rate: what is left to gain is the number of
// multiplications.  A 2 x 2 output tile of a 3 x 3 convolution needs 16 products per (cin, cout) pair instead of 36:
//     Y = At [ (G g Gt) .* (Bt d B) ] A,      d = the 4 x 4 input patch, g = the 3 x 3 filter,
//     Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  At = [1 1 1 0; 0 1 -1 -1]
// i.e. 16 independent channel contractions U[i][j][cout][cin] . V[i][j][cin][tile], one per position (i, j) of the
// transformed patch, each an implicit GEMM on v_mfma_f32_16x16x4_f32 with columns = 16 output tiles of one tile row.
// U = G g Gt is formed on the host in double precision (packing.py::pack_reg_layer_wino); Bt d B and At M A are additions.
//
// Mapping.  fp32 MFMA shares the vector lanes (lesson 1), so the input transform must cost few vector
// instructions per MFMA, and each weight fragment (one per MFMA) must be reused over as many tiles as the accumulators
// allow.  A workgroup = 4 waves = the 4 ROWS i of the transformed patch: wave i needs two raw patch rows (d0-d2 | d1+d2 |
// d2-d1 | d1-d3) and the column transform of that row -- four PACKED fp32 instructions in all (v_pk_fma / v_pk_add with
// op_sel / neg modifiers) -- and owns the accumulators of its four positions (i, 0..3) for MT channel tiles x NT tile
// rows: 4 vector instructions per 4 MT MFMAs.  One wave per SIMD; MT 4 x NT 3 = 48 accumulator tiles = 192 AGPRs (more
// than ~200 accumulator registers and the compiler wraps every MFMA in AGPR <-> VGPR copies: DESIGN.md lesson 9); LDS
// double-buffered in chunks of 16 input channels with ONE barrier per chunk, weight fragments L2 -> VGPR as one 16-byte
// load per (k-step, channel tile) carrying the four positions of the wave's row, four sets, three k-steps ahead.
// A workgroup covers 2 NT x 32 output pixels x 16 MT channels; the channel groups of a pixel block are neighbouring tiles
// of the persistent grid (they share the input window in L2).
// The four rows meet in the epilogue: Z_i[b] = sum_j M[i][j] At[b][j] in registers, then through LDS, wave (a, b) forms
// Y[a][b] = sum_i At[a][i] Z_i[b], adds bias / ReLU / skip and stores.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "persistent.h"
#include "planes.h"

// Timing builds only (tools/build_variant.py <name> -DWINO_EXP=<bits>; results are wrong): 1 no epilogue, 2 no weight loads
// in the loop, 4 no input transform, 8 no window fill / barrier (32 barrier only, 64 LDS stores only, 128 window loads only), 16 the chunk loop twice.  DESIGN.md section 4 quotes what each part costs.
#ifndef WINO_EXP
#define WINO_EXP 0
#endif

namespace adamvs {

struct WinoArgs {
  const float* in;     // [N][h*w][D]
  const float* wpk;    // [D/4][4][D/16][64][4]: k-step, patch row i, channel tile, lane (A-fragment order), patch column j
  const float* bias;   // [D]
  const float* skip;   // [N][h*w][D] or null; added after the ReLU
  float* out;          // [N][h*w][D]
  int D, h, w, relu;
  // SM kernels (the `prob` layer under adamvs_depth_stage_forward): no `out`; every lane reduces the 16 scores it holds of a
  // pixel to (max, sum of exp, sum of exp * plane) and stores that partial, [N][h*w][D/16][4]; k_softmax_merge finishes
  float* sm_part;
  PlaneSrc sm_planes;
  int sm_B, sm_D;      // image n belongs to tile n % sm_B; sm_D hypothesis planes (<= D: pad channels score -1e30)
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MT, int NT>
struct WinoGeom {
  static constexpr int KC = 16, KS = KC / 4;                 // input channels per LDS chunk, k-steps per chunk
  static constexpr int LR = 2 * NT + 2, LC = 34, NPIX = LR * LC;
  static constexpr int PLANE = ((NPIX + 31) / 64) * 64 + 32;  // == 32 (mod 64): the k-rows q, q+1 of a 32-lane ds_read_b64 group on disjoint banks
  static constexpr int GP = 4 * PLANE + 8;                   // channel-group pitch: the 4 groups of a pixel on different banks when the tile is filled
  static constexpr int CHUNK = KS * GP;                      // floats per buffer
  static constexpr int ZFLOATS = 4 * 2 * MT * 64 * 4;        // epilogue exchange of one tile row
  static constexpr int LDS_FLOATS = (2 * CHUNK > ZFLOATS) ? 2 * CHUNK : ZFLOATS;
  static_assert(PLANE >= NPIX, "plane pitch");
};

// Persistent: the grid is the resident capacity (one workgroup per CU) and workgroup i walks tiles i, i + grid, ...;
// tile = ((image * tiles_y + ty) * tiles_x + tx) * groups + channel group.  The first window chunk and the first three
// fragment sets of the NEXT tile are requested before the epilogue of the current one, so what a tile start exposes is
// one LDS fill and one barrier instead of a round trip to memory behind the per-lane address arithmetic.
// WPS = waves per SIMD the kernel is built for.  1: the whole register file (192 accumulator registers, four fragment sets three
// k-steps ahead).  2: two workgroups per CU, each wave at most 256 registers -- NT 2 tile rows (128 accumulator registers), two
// fragment sets one k-step ahead.  The second wave was built to fill the matrix-pipe cycles a lone wave loses to its other
// instructions (27 % of them); it fills almost none -- 74.3 against 73.6 % of the matrix rate at 512 maps: what the loads,
// LDS stores and transforms cost the matrix pipe is not issue slots but the register file they share with it -- and earns its
// keep through the XCD-aware tile order below: with three channel groups of every pixel block in one L2 it is the form whose
// extra window traffic (4-row tiles) stops mattering, 76.4 against 73.8 %.
template <int MT, int NT, int WPS, bool SM>
__global__ __launch_bounds__(256, WPS) void k_conv_wino(WinoArgs a, TileGrid tg, int groups, unsigned mgroups) {
  using G = WinoGeom<MT, NT>;
  constexpr int KC = G::KC, KS = G::KS, LC = G::LC, NPIX = G::NPIX, PLANE = G::PLANE, GP = G::GP, CHUNK = G::CHUNK;
  constexpr int NITEMS = NPIX * (KC / 4), NITA = (NITEMS + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS + 512];   // + the layer's bias vector (D <= 512)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = patch row i
  const int p = lane & 15, q = lane >> 4;
  const int D = a.D, NTILES = D / 16, NC = D / KC;

  // ---- per-lane constants (tile-independent)
  // window fill: item = (pixel of the window, group of 4 channels of the chunk); lane tid holds the items tid + 256 it: one channel
  // group for all of them, pixels pp0 + 64 it -- LDS addresses one pinned base + immediates, global offsets formed per tile
  // (window_of) from (row, column) = pp / LC, pp % LC.  The last round's surplus lanes repeat the last item.
  constexpr int LAST = NITEMS - 1 - 256 * (NITA - 1);        // last item of the last round
  const int fg = tid & 3, pp0 = tid >> 2;
  const int ppl = tid <= LAST ? pp0 + 64 * (NITA - 1) : NPIX - 1;
  unsigned xlds0 = (unsigned)((fg * GP + pp0) * 4), xldsl = (unsigned)(((tid <= LAST ? fg : KC / 4 - 1) * GP + ppl) * 4);
  const unsigned gch = (unsigned)((tid <= LAST ? fg : KC / 4 - 1) * 16);   // the last round's channel-group byte offset
  pin(xlds0); pin(xldsl);
  // raw patch rows of the wave: T = rowA + sgn * rowB  (i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3)
  const int rowA = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int rowB = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
  const float sgn = wave == 1 ? 1.0f : -1.0f;
  const f32x2 sgn2 = {sgn, sgn}, pm = {1.0f, -1.0f};
  unsigned pa = (unsigned)((q * PLANE + rowA * LC + 2 * p) * 4), pb = (unsigned)((q * PLANE + rowB * LC + 2 * p) * 4);
  unsigned pa2 = pa + 8, pb2 = pb + 8;
  pin(pa); pin(pb); pin(pa2); pin(pb2);
  // A fragments: 16 bytes per lane = the four positions (i, 0..3) of one (k-step, channel tile)
  const buf_rsrc rw = make_rsrc(a.wpk);
  unsigned woff = (unsigned)(lane * 16);
  pin(woff);
  // epilogue: wave (oa, ob) stores output pixel (2 ty + oa, 2 tx + ob) of every 2 x 2 tile
  const int oa = wave >> 1, ob = wave & 1;
  const float os = oa ? -1.0f : 1.0f;
  unsigned ooff = (unsigned)(((oa * a.w + 2 * p + ob) * D + 4 * q) * 4);        // + the tile's origin and the tile row
  pin(ooff);

  // ---- the tile in flight (scalars) and its window offsets
  struct Tile { int n, r0, c0, cg; };
  auto decode = [&](int t) {
    int n, tx, ty;
    tile_coords(tg, t, n, tx, ty);                           // tx runs over (block column, channel group)
    const int bx = groups == 1 ? tx : (int)__umulhi((unsigned)tx, mgroups);
    return Tile{n, ty * 2 * NT, bx * 32, tx - bx * groups};
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

  f32x4 acc[4][MT][NT];
  f32x4 xs[NITA];
  f32x4 wf0[MT], wf1[MT], wf2[MT], wf3[MT];                  // four named fragment sets (below)
  // The bias rides in the accumulators: At E A is all ones for E = the unit matrix element (1, 1), so position (1, 1) of every tile
  // starts from the bias instead of zero (wave 1, first k-step) and the epilogue adds nothing.  It comes from LDS (filled once per
  // workgroup): a vector-memory load in the epilogue would make its wait drain the next tile's requests, which fly there.
  f32x4 cb[MT];
  for (int i = tid; i < D; i += 256) lds[G::LDS_FLOATS + i] = a.bias[i];
  int cg = 0;                                                // channel group of the fragments being requested

  auto load_w = [&](f32x4 (&wf)[MT], int ks) {               // ks: global k-step
    if ((WINO_EXP & 2) && ks > 3) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned frag = (unsigned)(((ks * 4 + wave) * NTILES + cg * MT + mt) * 1024);                 // uniform
      wf[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, frag, 0));
    }
  };
  auto load_x = [&](int ch) {
#pragma unroll
    for (int it = 0; it < NITA; ++it)
      xs[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[it], (unsigned)ch * 4u, 0));
  };
  auto read_bias = [&]() {                                   // before the first k-step of a tile (cg = its channel group)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 b = *(const f32x4*)(lds + G::LDS_FLOATS + (cg * MT + mt) * 16 + 4 * q);
      cb[mt] = wave == 1 ? b : zero;
    }
  };
  auto store_items = [&](int buf, int i0, int i1) {          // window items [i0, i1) of the staged chunk -> buffer `buf`
#pragma unroll
    for (int it = i0; it < i1; ++it) {
      float* dl = (float*)((char*)lds + (it + 1 < NITA ? xlds0 : xldsl)) + buf * CHUNK + (it + 1 < NITA ? 64 * it : 0);
      dl[0] = xs[it].x; dl[PLANE] = xs[it].y; dl[2 * PLANE] = xs[it].z; dl[3 * PLANE] = xs[it].w;
    }
  };
  // The raw patch rows of tile row t, k-step ks of the chunk in buffer `buf`: four ds_read_b64 at pinned base + immediate
  // (the halves of a row go through two separately pinned bases: merged into one ds_read2_b64, whose offset field is 8 bits,
  // the pair would need a vector add per read for its address).
  struct Raw { f32x2 a01, a23, b01, b23; };
  auto read_raw = [&](Raw& r, int buf, int ks, int t) {
    const int off = (buf * CHUNK + ks * GP + 2 * t * LC) * 4;
    r.a01 = *(const f32x2*)((const char*)lds + pa + off); r.a23 = *(const f32x2*)((const char*)lds + pa2 + off);
    r.b01 = *(const f32x2*)((const char*)lds + pb + off); r.b23 = *(const f32x2*)((const char*)lds + pb2 + off);
  };
  // one tile row of one k-step: 4 packed vector instructions, 4 MT MFMAs.  FIRST: the first k-step of a tile starts the sums
  // (C = 0 as an inline constant: no pass over the 192 accumulator registers to clear them)
  auto tile_row = [&](const f32x4 (&wf)[MT], const Raw& r, int t, auto firstc) {
    constexpr bool FIRST = decltype(firstc)::value;
    // T = A + sgn B; (v0, v3) = (t0 - t2, t1 - t3); (v1, v2) = (t2 + t1, t2 - t1)
    f32x2 t01 = __builtin_elementwise_fma(sgn2, r.b01, r.a01), t23 = __builtin_elementwise_fma(sgn2, r.b23, r.a23);
    f32x2 v03 = t01 - t23;
    f32x2 v12 = __builtin_elementwise_fma(__builtin_shufflevector(t01, t01, 1, 1), pm, __builtin_shufflevector(t23, t23, 0, 0));
    if (WINO_EXP & 4) { v03 = r.a01; v12 = r.b23; }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[0][mt][t] = mfma16(wf[mt].x, v03.x, FIRST ? zero : acc[0][mt][t]);
      acc[1][mt][t] = mfma16(wf[mt].y, v12.x, FIRST ? cb[mt] : acc[1][mt][t]);
      acc[2][mt][t] = mfma16(wf[mt].z, v12.y, FIRST ? zero : acc[2][mt][t]);
      acc[3][mt][t] = mfma16(wf[mt].w, v03.y, FIRST ? zero : acc[3][mt][t]);
    }
  };
  // one k-step (4 input channels): the raw rows of the NEXT tile row (or of the next k-step's first) are requested before the
  // MFMAs of the current one -- with one wave per SIMD nothing else hides the LDS latency.  FILL: the staged chunk c+1 goes to the
  // other buffer between the tile rows (LDS stores issue next to the MFMAs), then chunk c+2 is requested.
  auto kstep = [&](const f32x4 (&wf)[MT], Raw& r, int buf, int ks, bool fill, int next_ch, auto firstc) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      Raw nx;
      const bool more = t + 1 < NT || ks + 1 < KS;
      if (more) read_raw(nx, buf, t + 1 < NT ? ks : ks + 1, t + 1 < NT ? t + 1 : 0);
      __builtin_amdgcn_sched_barrier(0);        // (left alone, the scheduler hoists the next transforms above this tile row's MFMAs
      tile_row(wf, r, t, firstc);               //  and waits for their LDS reads right after issuing them)
      if (fill && !(WINO_EXP & (8 | 64))) store_items(buf ^ 1, t * NITA / NT, (t + 1) * NITA / NT);
      __builtin_amdgcn_sched_barrier(0);
      if (more) r = nx;
    }
    if (fill && !(WINO_EXP & (8 | 128))) load_x(next_ch);
  };
  // Four named fragment sets: k-step ks of a chunk uses set ks, and requests the fragments of three k-steps ahead into the set
  // freed by the previous k-step (the L2 latency under load is longer than one k-step's 48 MFMAs).  Past the last chunk the
  // window requests repeat the last one, so that the number of loads in flight -- what the waits count -- does not change.
  const int last_ks = NC * KS - 1;
  static_assert(KS == 4, "four k-steps per chunk, one per fragment set");
  // chunk c from buffer CUR (a compile-time constant: every LDS address of the loop is a pinned register + an immediate)
  auto chunk = [&](int c, auto curc, auto firstc, bool fill, int next_ch) {
    constexpr int CUR = decltype(curc)::value;
    // every wave is past chunk c-1 (the barrier that ended it): its buffer takes chunk c+1 during k-step 0
    Raw r;
    read_raw(r, CUR, 0, 0);
    if constexpr (WPS == 2) {                                // two sets, one k-step ahead
      kstep(wf0, r, CUR, 0, fill, next_ch, firstc);
      load_w(wf0, min(c * KS + 2, last_ks));
      kstep(wf1, r, CUR, 1, false, 0, std::false_type{});
      load_w(wf1, min(c * KS + 3, last_ks));
      kstep(wf0, r, CUR, 2, false, 0, std::false_type{});
      load_w(wf0, min(c * KS + 4, last_ks));
      kstep(wf1, r, CUR, 3, false, 0, std::false_type{});
      load_w(wf1, min(c * KS + 5, last_ks));
      if (!(WINO_EXP & (8 | 32))) __syncthreads();
      return;
    }
    load_w(wf3, min(c * KS + 3, last_ks));
    kstep(wf0, r, CUR, 0, fill, next_ch, firstc);
    load_w(wf0, min(c * KS + 4, last_ks));                   // (the next tile's first sets are requested below)
    kstep(wf1, r, CUR, 1, false, 0, std::false_type{});
    load_w(wf1, min(c * KS + 5, last_ks));
    kstep(wf2, r, CUR, 2, false, 0, std::false_type{});
    load_w(wf2, min(c * KS + 6, last_ks));
    kstep(wf3, r, CUR, 3, false, 0, std::false_type{});
    if (!(WINO_EXP & (8 | 32))) __syncthreads();
  };

  // ---- tile loop
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
  if (WPS == 1) load_w(wf2, min(2, last_ks));
  wait_vmem_all();
  while (true) {
    const int next = tile + (int)gridDim.x;
    const bool more = next < tg.ntiles;
    // here: chunk 0 of `cur` has arrived in xs, nothing reads LDS
    store_items(0, 0, NITA);
    load_x(KC);
    __syncthreads();
    read_bias();
    // the first k-step of a tile starts the sums (C = 0 as an inline constant: no pass over the 192 accumulator registers)
    chunk(0, std::integral_constant<int, 0>{}, std::true_type{}, true, 2 * KC);
    chunk(1, std::integral_constant<int, 1>{}, std::false_type{}, true, 3 * KC);
    for (int rep = 0; rep < ((WINO_EXP & 16) ? 2 : 1); ++rep)  // (timing build 16: the chunk loop twice)
    for (int c = 2; c < NC - 2; c += 2) {                    // NC is even (>= 4) for every supported D
      chunk(c, std::integral_constant<int, 0>{}, std::false_type{}, true, (c + 2) * KC);
      chunk(c + 1, std::integral_constant<int, 1>{}, std::false_type{}, true, (c + 3) * KC);
    }
    // the last two chunks: the window request of chunk NC-2 is the NEXT tile's first chunk (a chunk and the epilogue ahead of
    // its use; it stays in registers: the buffers are the epilogue's), chunk NC-1 stages and requests nothing
    const Tile done = cur;
    if (more) {
      cur = decode(next);
      window_of(cur);
    }
    chunk(NC - 2, std::integral_constant<int, 0>{}, std::false_type{}, true, 0);
    chunk(NC - 1, std::integral_constant<int, 1>{}, std::false_type{}, false, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) drain(acc[j][mt][0]);   // (tile row 0 is read first; tools/mfma_hazard_lint.py checks the rest)

    cg = cur.cg;
    // the next tile's first fragment sets fly during the epilogue
    if (more) {
      load_w(wf0, 0);
      load_w(wf1, min(1, last_ks));
      if (WPS == 1) load_w(wf2, min(2, last_ks));
    }
    // ---- epilogue of `done`: the four rows meet through LDS, one tile row per round
    if (!((WINO_EXP & 1) && a.relu != 12345)) {
      const long img = (long)done.n * a.h * a.w * (long)D * 4;
      const buf_rsrc ro = make_rsrc((char*)a.out + img);
      const buf_rsrc rk = make_rsrc((const char*)(a.skip ? a.skip : a.out) + img);
      f32x4* zl = (f32x4*)lds;
      // SM: the tile's plane line (uniform: scalar loads) and the partials' descriptor (D/16 partials x 16 bytes = D bytes per pixel)
      const int sm_b = SM ? done.n % a.sm_B : 0, sm_last = a.sm_D - 1, dq = 4 * q;
      const float sm_lo = SM ? a.sm_planes.p[2 * sm_b] : 0.f, sm_hi = SM ? a.sm_planes.p[2 * sm_b + 1] : 0.f;
      const float sm_step = (sm_hi - sm_lo) / (float)sm_last;   // as planes.h::plane_line
      const buf_rsrc rp = make_rsrc((char*)a.sm_part + (size_t)done.n * a.h * a.w * (size_t)(64 * groups));      // 4 partials of 16 bytes per channel group
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t) __syncthreads();                                // the previous round's readers are done
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          zl[((wave * 2 + 0) * MT + mt) * 64 + lane] = (acc[0][mt][t] + acc[1][mt][t]) + acc[2][mt][t];
          zl[((wave * 2 + 1) * MT + mt) * 64 + lane] = (acc[1][mt][t] - acc[2][mt][t]) - acc[3][mt][t];
        }
        __syncthreads();
        const int oy = done.r0 + 2 * t + oa, ox = done.c0 + 2 * p + ob;
        if constexpr (SM) {
          // The softmax partial of the lane's 16 channels (costreg_softmax.h; the merge over a pixel's D/16 lanes is k_softmax_merge's:
          // the channel groups of a pixel are different workgroups), accumulated channel tile by channel tile with the running
          // maximum -- one more exponential per tile instead of sixteen live scores.  Planes: uniform per tile (stage 1,
          // planes.h PLANES_UNIFORM): (lo, hi) through the scalar cache, nothing that would wait on the vector-memory counter and
          // with it on the next tile's window and fragments, which are in flight here.
          float m = -INFINITY, se = 0.f, sd = 0.f;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 z0 = zl[(((oa + 0) * 2 + ob) * MT + mt) * 64 + lane];
            const f32x4 z1 = zl[(((oa + 1) * 2 + ob) * MT + mt) * 64 + lane];
            const f32x4 z2 = zl[(((oa + 2) * 2 + ob) * MT + mt) * 64 + lane];
            const f32x4 v = z0 + os * (z1 + z2);
            const float mn = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            const float sc = __expf(m - mn);                   // (first tile: exp(-inf) = 0 times 0)
            const int d = (done.cg * MT + mt) * 16 + dq;
            const float e0 = __expf(v.x - mn), e1 = __expf(v.y - mn), e2 = __expf(v.z - mn), e3 = __expf(v.w - mn);
            se = __fmaf_rn(se, sc, (e0 + e1) + (e2 + e3));
            sd = sd * sc;
            sd = __fmaf_rn(e3, plane_value(sm_lo, sm_step, min(d + 3, sm_last)), __fmaf_rn(e2, plane_value(sm_lo, sm_step, min(d + 2, sm_last)),
                 __fmaf_rn(e1, plane_value(sm_lo, sm_step, min(d + 1, sm_last)), __fmaf_rn(e0, plane_value(sm_lo, sm_step, min(d, sm_last)), sd))));
            m = mn;
          }
          // one 16-byte store per lane and round, always issued (out-of-image pixels: BUF_OOB) so that the wait below counts right
          const unsigned pbase = (oy < a.h && ox < a.w) ? (unsigned)(((oy * a.w + ox) * (4 * groups) + done.cg * 4 + q) * 16) : BUF_OOB;
          buf_store4(rp, pbase, f32x4{m, se, sd, 0.f});
        } else {
        const unsigned obase = (oy < a.h && ox < a.w)
            ? ooff + (unsigned)((((done.r0 + 2 * t) * a.w + done.c0) * D + done.cg * MT * 16) * 4) : BUF_OOB;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 z0 = zl[(((oa + 0) * 2 + ob) * MT + mt) * 64 + lane];
          const f32x4 z1 = zl[(((oa + 1) * 2 + ob) * MT + mt) * 64 + lane];
          const f32x4 z2 = zl[(((oa + 2) * 2 + ob) * MT + mt) * 64 + lane];
          f32x4 v = z0 + os * (z1 + z2);
          if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (a.skip) v += buf_load4(rk, obase == BUF_OOB ? BUF_OOB : obase + mt * 64);
          buf_store4(ro, obase == BUF_OOB ? BUF_OOB : obase + mt * 64, v);
        }
        }
      }
    }
    if (!more) break;
    tile = next;
    // the window chunk and the fragments were requested BEFORE the epilogue's NT * MT stores (vmcnt retires in order):
    // wait for them, not for the stores.  With a skip operand its loads were waited for already.
    wait_vmem_but<SM ? NT : NT * MT>();
    __syncthreads();                                         // the epilogue's LDS readers are done: the buffers are free
  }
}

template <int MT, int NT, int WPS, bool SM = false>
static int launch_wino_cfg(const WinoArgs& a, int N, hipStream_t st) {
  const int groups = a.D / (16 * MT);
  auto kern = k_conv_wino<MT, NT, WPS, SM>;
  static const int capacity = resident_blocks(kern, 256, 0);       // once per instantiation, thread-safely (magic static)
  TileGrid tg;
  if (int rc = make_tile_grid(tg, groups * cdiv(a.w, 32), cdiv(a.h, 2 * NT), N)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a, tg, groups, (unsigned)(((1ull << 32) + groups - 1) / groups));
  ADAMVS_CHECK_LAUNCH("conv_wino");
  return 0;
}

bool wino_depth_supported(int D) { return D >= 64 && D <= 512 && D % 64 == 0; }       // channel groups of 64; 16-channel LDS chunks

// (Round 6 measured channel groups of 96 -- MT = 6 channel tiles x NT = 2 tile rows, the same 192 accumulator registers as 4 x 3,
// one workgroup per CU, TWO groups per pixel block at D = 192 instead of three: 4 packed vector instructions and 4 LDS reads per 24
// MFMAs instead of 16, a third less window traffic.  It loses to the 4 x 2 form at two workgroups per CU everywhere: 512 maps of
// 96 x 192 / 48 x 96 / 24 x 48: 24.5 / 6.15 / 2.06 ms against 22.0 / 5.79 / 1.89; cfg2's step 393.5 -> 406.3 ms.  What the second
// wave per SIMD hides -- the transforms, the barrier -- is worth more than the vector instructions the wider group saves.
// profiles/r06_wino_mt6_ab.txt; the kernel is generic in MT, the instantiation is not built.)
int launch_conv_wino(const float* in, const float* wpk, const float* bias, const float* skip, float* out, int N, int D, int h, int w,
                     int relu, hipStream_t st) {
  const WinoArgs a{in, wpk, bias, skip, out, D, h, w, relu, nullptr, PlaneSrc{nullptr, 0, 0.f}, 1, D};
  ADAMVS_CHECK_ARG(wino_depth_supported(D), "conv_wino: D=%d unsupported (a multiple of 64 up to 512)", D);
  // 32-bit byte offsets inside one image through a buffer descriptor (advisor, round 3): larger maps would read zeros
  ADAMVS_CHECK_ARG((size_t)h * w * D * 4 < 0x7fffffffu, "conv_wino: a map of %dx%dx%d floats exceeds the 2 GiB a buffer descriptor spans", h, w, D);
  // Two workgroups per CU (4 x 32-pixel tiles) except on the smallest maps of the hourglass, where the 6 x 32 tiles of the
  // one-workgroup form cover a 12-row map without a ragged tile row.  With the XCD-aware tile order, D = 192 (tools/wino_bench.py):
  // 512 maps of 96x192 / 48x96 / 24x48 / 12x24 pixels 23.2 / 5.79 / 1.90 / 0.57 ms against 24.0 / 5.99 / 2.00 / 0.51 - 0.56;
  // 64 maps of 96x192 2.87 against 3.01, 16 maps (cfg4's share of four tiles) 0.74 against 0.76.
  // Option wino_wps = 1 / 2 forces one form (A/B); both give the same bits.
  const int forced = opt(OPT_WINO_WPS);
  const bool two = forced ? forced == 2 : h * w >= 1024;
  return two ? launch_wino_cfg<4, 2, 2>(a, N, st) : launch_wino_cfg<4, 3, 1>(a, N, st);
}

// The merge of a pixel's D/16 partials (m, sum exp, sum exp * plane): the online-softmax merge of costreg_softmax.h.  Four
// lanes per pixel, lane l taking partials l, l + 4, ... -- a load instruction then covers whole 64-byte segments (one thread
// per pixel walked its D bytes 16 at a time, 64 segments per instruction: 0.5 TB/s) -- and a butterfly over the four lanes.
// view weight = max_d softmax = 1 / sum, pair depth = weighted sum / sum (adamvs.py:481-486).
__global__ __launch_bounds__(256) void k_softmax_merge(const f32x4* __restrict__ part, float* __restrict__ vw, float* __restrict__ pd,
                                                       size_t npix, int np) {
  const size_t gp = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2);
  const int l = threadIdx.x & 3;
  const size_t pix = gp < npix ? gp : npix - 1;
  const f32x4* pp = part + pix * np;
  float M = -INFINITY, Z = 0.f, P = 0.f;
  for (int j = l; j < np; j += 4) {                            // np is a multiple of 4
    const f32x4 t = pp[j];
    const float mn = fmaxf(M, t.x), so = __expf(M - mn), sn = __expf(t.x - mn);
    Z = __fmaf_rn(t.y, sn, Z * so);
    P = __fmaf_rn(t.z, sn, P * so);
    M = mn;
  }
#pragma unroll
  for (int o = 1; o < 4; o <<= 1) {
    const float m2 = __shfl_xor(M, o, 64), z2 = __shfl_xor(Z, o, 64), p2 = __shfl_xor(P, o, 64);
    const float mn = fmaxf(M, m2), s1 = __expf(M - mn), s2 = __expf(m2 - mn);
    Z = __fmaf_rn(z2, s2, Z * s1);
    P = __fmaf_rn(p2, s2, P * s1);
    M = mn;
  }
  if (gp < npix && l == 0) {
    vw[pix] = 1.0f / Z;
    pd[pix] = P / Z;
  }
}

// option wino_softmax = 0: the scores of `prob` through the score volume to k_softmax_regress, as in rounds 3-4 (A/B)
bool wino_softmax_fused() { return opt(OPT_WINO_SOFTMAX) != 0; }

size_t wino_softmax_part_floats(int N, int D, int h, int w) { return (size_t)N * h * w * (D / 16) * 4; }

// `prob` in the F(2x2, 3x3) form with softmax / max / depth regression behind it, the score volume never stored:
// part = wino_softmax_part_floats() floats of scratch; planes: uniform per tile ([B][2] = lo, hi: stage 1's plane source),
// n_planes <= D hypothesis planes, image n of tile n % B
int launch_conv_wino_softmax(const float* in, const float* wpk, const float* bias, float* part, const PlaneSrc& planes, float* vw, float* pd,
                             int N, int B, int D, int n_planes, int h, int w, hipStream_t st) {
  const WinoArgs a{in, wpk, bias, nullptr, nullptr, D, h, w, 0, part, planes, B, n_planes > 0 ? n_planes : D};
  ADAMVS_CHECK_ARG(planes.mode == PLANES_UNIFORM && a.sm_D > 1, "conv_wino_softmax: planes uniform per tile, at least two (mode %d, %d planes)", planes.mode, a.sm_D);
  ADAMVS_CHECK_ARG(wino_depth_supported(D), "conv_wino_softmax: D=%d unsupported (a multiple of 64 up to 512)", D);
  ADAMVS_CHECK_ARG((size_t)h * w * D * 4 < 0x7fffffffu, "conv_wino_softmax: a map of %dx%dx%d floats exceeds the 2 GiB a buffer descriptor spans", h, w, D);
  const int forced = opt(OPT_WINO_WPS);
  const bool two = forced ? forced == 2 : h * w >= 1024;
  if (int rc = two ? launch_wino_cfg<4, 2, 2, true>(a, N, st) : launch_wino_cfg<4, 3, 1, true>(a, N, st)) return rc;
  const size_t npix = (size_t)N * h * w;
  // a lane's partial covers the MT = 4 channel tiles of its workgroup: 4 partials per channel group of 64 and pixel
  hipLaunchKernelGGL(k_softmax_merge, dim3((unsigned)((npix + 63) / 64)), dim3(256), 0, st, (const f32x4*)part, vw, pd, npix, D / 16);
  ADAMVS_CHECK_LAUNCH("softmax_merge");
  return 0;
}

}  // namespace adamvs

using namespace adamvs;

extern "C" size_t adamvs_prob_softmax_regress_wino_workspace_bytes(int S, int B, int D, int h, int w) {
  if (!wino_depth_supported(D) || D < 2 || S <= 0 || B <= 0 || h <= 0 || w <= 0) return 0;      // unsupported: no size to satisfy
  return wino_softmax_part_floats(S * B, D, h, w) * sizeof(float);
}

extern "C" int adamvs_prob_softmax_regress_wino(const float* in, const float* wpk, const float* bias, const float* depth_range, float* view_weight,
                                                float* pair_depth, int S, int B, int D, int h, int w, void* workspace,
                                                size_t workspace_bytes, void* stream) {
  ADAMVS_CHECK_ARG(in && wpk && bias && depth_range && view_weight && pair_depth && workspace && S > 0 && B > 0 && h > 0 && w > 0,
                   "prob_softmax_regress_wino: bad arguments");
  // before any size is derived from D (D / 16 partials per pixel would truncate silently)
  ADAMVS_CHECK_ARG(wino_depth_supported(D) && D > 1, "prob_softmax_regress_wino: D=%d unsupported (a multiple of 64 up to 512)", D);
  ADAMVS_CHECK_ARG(workspace_bytes >= adamvs_prob_softmax_regress_wino_workspace_bytes(S, B, D, h, w),
                   "prob_softmax_regress_wino: workspace too small (%zu < %zu bytes)", workspace_bytes,
                   adamvs_prob_softmax_regress_wino_workspace_bytes(S, B, D, h, w));
  return launch_conv_wino_softmax(in, wpk, bias, (float*)workspace, PlaneSrc{depth_range, PLANES_UNIFORM, 0.f}, view_weight, pair_depth, S * B, B,
                                  D, D, h, w, (hipStream_t)stream);
}

extern "C" int adamvs_conv3x3_dd_wino(const float* in, const float* wpk, const float* bias, const float* skip, float* out, int N,
                                      int D, int h, int w, int relu, void* stream) {
  ADAMVS_CHECK_ARG(in && wpk && bias && out && N > 0 && h > 0 && w > 0, "conv3x3_dd_wino: bad arguments");
  return launch_conv_wino(in, wpk, bias, skip, out, N, D, h, w, relu, (hipStream_t)stream);
}
