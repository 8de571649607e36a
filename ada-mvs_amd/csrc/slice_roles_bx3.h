// Device-side roles of the split-bf16 ("bf16x3") recurrent convolutions: see slice_roles.h for the role idea and
// slice_red_bf16x3.hip for the arithmetic.  reference models/adamvs.py:400-424, models/module.py:5-52.
#pragma once
#include "common.h"
#include "kernels.h"
#include "persistent.h"
#include "slice_roles.h"

namespace adamvs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_bx(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// hi = bf16(x) (round to nearest even), lo = bf16(x - hi).  Written on pairs -- one v_cvt_pk_bf16_f32, the two halves widened back
// with a shift and a mask, one packed subtraction, one v_cvt_pk_bf16_f32: 10 vector instructions per four values (the element-wise
// form compiled to 17: the conversions of the first pair once per element and once packed).
__device__ __forceinline__ void split4(f32x4 v, bf16x4& h, bf16x4& l) {
  typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
  u32x2v hp, lp;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const f32x2v x = e ? f32x2v{v.z, v.w} : f32x2v{v.x, v.y};
    const unsigned hh = __builtin_bit_cast(unsigned, bf16x2v{(__bf16)x.x, (__bf16)x.y});
    const f32x2v d = x - f32x2v{__uint_as_float(hh << 16), __uint_as_float(hh & 0xffff0000u)};
    const unsigned ll = __builtin_bit_cast(unsigned, bf16x2v{(__bf16)d.x, (__bf16)d.y});
    if (e) { hp.y = hh; lp.y = ll; } else { hp.x = hh; lp.x = ll; }
  }
  h = __builtin_bit_cast(bf16x4, hp);
  l = __builtin_bit_cast(bf16x4, lp);
}
__device__ __forceinline__ void split_store(__bf16* hi, __bf16* lo, f32x4 v) {
  bf16x4 h, l;
  split4(v, h, l);
  *(bf16x4*)hi = h;
  *(bf16x4*)lo = l;
}

// SPLIT MAPS (round 5).  Every map of the recurrence that a convolution of this mode reads through a window -- c1, the two states,
// c2 -- is split into its bf16 halves ONCE, by the kernel that produces it, instead of by every consumer for every pixel of its
// window (halo included: 1.7 x the tile for the level-1 kernel, whose fill spent 134 of its 435 vector instructions per tile on it):
//     [pixel][hi C bf16 | lo C bf16]      4 C bytes per pixel, like the fp32 map whose place it takes
// hi = bf16(x), lo = bf16(x - hi): the same function the fill applied, so the LDS tiles -- and every result -- keep their bits.
// c1 and c2 exist only in this form; a state is kept twice: fp32 (the blend u h + (1 - u) c of the lane's own pixel and the
// decoder read it; updated in place -- nobody else reads it) and split (what the next step's windows copy; two buffers in turn).
// A window fill is then a copy: 16-byte loads, 16-byte LDS stores, no vector arithmetic.  -DBX3_PRESPLIT=0: the fills split (A/B).
#ifndef BX3_PRESPLIT
#define BX3_PRESPLIT 0
#endif
// Timing builds of Gru1FusedBx3Role only (tools/build_variant.py <name> -DBX3_EXP=<bits>; results are wrong): 1 no MFMAs (the fragment
// reads stay), 2 the B fragments of a tile read once (run 0's serve every run -- NOT a measure of the reads: with identical operands the
// compiler folds the runs' MFMA chains too), 4 no transcendentals, 8 no window (loads and LDS stores).
#ifndef BX3_EXP
#define BX3_EXP 0
#endif
#ifndef BX3_AHEAD
#define BX3_AHEAD 2
#endif
#ifndef BX3_AHEAD2
#define BX3_AHEAD2 1
#endif
#ifndef BX3_GATE_REUSE
#define BX3_GATE_REUSE 1
#endif
#ifndef BX3_CAND_TWO_ROW
#define BX3_CAND_TWO_ROW 1
#endif
__device__ __forceinline__ void keep(const bf16x8& v) { asm volatile("" ::"v"(v)); }
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
// the lane's four channels co4 .. co4 + 3 of one pixel: hi at `off` (= pixel * 4 C + 2 co4), lo `half` (= 2 C) bytes further
__device__ __forceinline__ void buf_store_split4(buf_rsrc r, unsigned off, unsigned half, f32x4 v) {
  bf16x4 h, l;
#ifdef BX3_EXP_NOSPLITSTORE
  return;
#endif
  split4(v, h, l);
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2s, h), r, off, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2s, l), r, off, half, 0);
}

// The fused GRU kernels of this mode take their gate / candidate weights and biases PRE-SCALED on the host
// (packing.pack_slice_reg_net, bf16x3): gates by -log2(e), candidates by 2 log2(e), so that the argument of v_exp_f32 is the
// accumulator itself -- sigmoid(x) = 1 / (1 + 2^(-x log2 e)), tanh(x) = 1 - 2 / (2^(2 x log2 e) + 1) -- and the blend
// u h + (1 - u) c runs as c + u (h - c): one multiply per transcendental and one instruction per blend fewer, of a kernel
// that spends its time on the vector units (3.4 vector instructions per MFMA before, profiles/r04_asm_mix_gru1_bx3.txt).
// (sigmoid_pre, tanh_pre, gru_blend: common.h)

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2b __attribute__((ext_vector_type(2)));

enum { BXE_RELU = 0, BXE_GATES = 1, BXE_CAND = 2, BXE_TWO_ROW = 3 };

// bf16 per pixel of the LDS tile.  ds_read_b128 is serviced in four NON-contiguous 16-lane groups (lanes {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31}, ...: MI355X_MICROARCH.md, LDS), each mixing two k-groups over complementary pixel
// columns: with all k-groups of a lane's fragment inside one pixel (CIN = 32, 16) the group is conflict-free when the
// pitch in 16-byte slots is 2 (mod 4) -- 96 B for 32 channels, 32 B for 16; CIN + 8 (80 / 48 B) is 2-way.
__host__ __device__ constexpr int bx_pixel_pitch(int cin) { return cin == 32 ? 48 : (cin == 16 ? 16 : cin + 8); }

struct SmallConvArgsBx {
  const float* srcA;      // [B][hi*wi][CA]
  const float* srcB;      // [B][hi*wi][CB] (null when CB == 0)
  const bf16x8* wpk;      // A fragments [NT][hi|lo][NKB][64] x 8 bf16
  const float* bias;      // [16*NT] (GATES, CAND)
  float* dst0;            // RELU/TWO_ROW: out [B][ho*wo][cout]; GATES: r*h; CAND: h (in place)
  float* dst1;            // GATES: u out; CAND: u in
  const float* hsrc;      // GATES: h [B][ho*wo][HC] (the centre-pixel state, read in fp32)
  int hi, wi, ho, wo, cout;
  const float* hin;       // CAND: the state that is blended; null = dst0 (update in place)
};

// Persistent and pipelined exactly like k_conv_small in slice_red.hip (uniform buffer descriptors + pinned lane
// offsets, interior tiles without bounds checks, one wait per tile, requests before the MFMA chain, stores after
// it); what differs is the tile in LDS (pixel-major bf16, hi and lo images, written through split_store), the
// chain (flattened k, three MFMAs per k-block) and the GRU state of the lane's own pixel, which the reset gate
// multiplies in full fp32 and therefore comes from global memory with the other epilogue operands.
// Tile = 4 rows x 16 columns, one run per wave; the two-row conv1 takes 8 x 16 (a run = 2 output rows).
// NPOS = 9 taps, or 12 (rr,kx) positions for the two-row conv1.
// SIN: srcA is a split map (the fill copies); SOUT: the output is written as one (RELU / TWO_ROW epilogues).
template <int CA, int CB, int NT, int STRIDE, int EPI, bool SIN = false, bool SOUT = false>
struct ConvSmallBx3Role {
  typedef SmallConvArgsBx Args;
  static constexpr bool TWO = (EPI == BXE_TWO_ROW);
  static_assert(!SIN || (CB == 0 && CA % 8 == 0), "a split source: one input, 16-byte pieces");
  static_assert(!SOUT || EPI == BXE_RELU || EPI == BXE_TWO_ROW, "split output: the plain epilogues");
  static constexpr int CIN = CA + CB, GA = CA / 4, GB = CB / 4, HC = CB;
  static constexpr int NPOS = TWO ? 12 : 9;
  static constexpr int NKB = (NPOS * CIN + 31) / 32;
  static constexpr int TR = TWO ? 8 : 4, TC = 16;
  static constexpr int LR = (STRIDE == 1) ? TR + 2 : 2 * TR + 1;
  static constexpr int LC = (STRIDE == 1) ? TC + 2 : 2 * TC + 1;
  static constexpr int NPIX = LR * LC;
  static constexpr int PP = bx_pixel_pitch(CIN);      // bf16 per pixel
  static constexpr int LO = NPIX * PP * 2;            // byte offset of the lo image
  static constexpr int NA = (NPIX * GA + 255) / 256, NB = (NPIX * GB + 255) / 256, NL = NA + NB;
  static constexpr size_t LDS_BYTES = (size_t)2 * NPIX * PP * sizeof(__bf16);         // [hi|lo][NPIX][PP]
  static constexpr int TILE_W = TC, TILE_H = TR;
  static int tiles_x(const Args& a) { return cdiv(a.wo, TC); }
  static int tiles_y(const Args& a) { return cdiv(a.ho, TR); }

  static __device__ __forceinline__ void run(const Args& a, const TileGrid& tg, TileRange tr, int wg, int nwg, float* lds_) {
  __bf16* ldsb = (__bf16*)lds_;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int row = TWO ? 2 * wave : wave;              // first output row of the wave's run

  // A fragments (hi, lo)
  bf16x8 wh[NT][NKB], wl[NT][NKB];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      wh[nt][kb] = a.wpk[((nt * 2 + 0) * NKB + kb) * 64 + lane];
      wl[nt][kb] = a.wpk[((nt * 2 + 1) * NKB + kb) * 64 + lane];
    }
  f32x4 bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    bias[nt] = (EPI == BXE_GATES || EPI == BXE_CAND) ? *(const f32x4*)(a.bias + nt * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- per-lane constants
  unsigned goff[NL], lbyte[NL];
  int rc[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const bool isA = k < NA;
    const int gs = isA ? GA : GB, cs = isA ? CA : CB;
    int j = tid + (isA ? k : k - NA) * 256;
    j = min(j, NPIX * gs - 1);                        // surplus lanes repeat the last item
    const int g = j % gs, pp = j / gs, r = pp / LC, c = pp % LC;
    goff[k] = (unsigned)(((r * a.wi + c) * cs + 4 * g) * 4);
    lbyte[k] = (unsigned)((pp * PP + (isA ? 0 : CA) + 4 * g) * 2);
    constexpr int GH = GA >= 2 ? GA / 2 : 1;         // 16-byte pieces per half of a split pixel
    if (SIN) lbyte[k] = (unsigned)((pp * PP + 8 * (g % GH)) * 2 + (g / GH) * LO);                 // piece g of the pixel: 8 channels of hi or lo
    rc[k] = r | (c << 16);
    pin(goff[k]); pin(lbyte[k]); pin(rc[k]);
  }
  unsigned xoff[NKB];                                 // B fragment of k-block kb: 8 channels of one position
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    int kk = 32 * kb + 8 * q;
    int pos = kk / CIN, c0 = kk % CIN;
    if (pos >= NPOS) { pos = 0; c0 = 0; }             // zero-weight padding: any valid address
    const int dy = pos / 3, dx = pos % 3;
    xoff[kb] = (unsigned)(((((row * STRIDE) + dy) * LC + p * STRIDE + dx) * PP + c0) * 2);
    pin(xoff[kb]);
  }
  // outputs: lane's pixel (orow, p); two-row: lanes q < 2 own row `row`, q >= 2 row + 1 (channels 4(q&1)..)
  const int orow = TWO ? row + (q >> 1) : row;
  const int CO = (EPI == BXE_RELU) ? a.cout : (TWO ? 8 : HC);
  unsigned ooff[NT], ooff1[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co4 = TWO ? 4 * (q & 1) : nt * 16 + 4 * q;
    unsigned at = (unsigned)(((orow * a.wo + p) * CO + (EPI == BXE_GATES && co4 >= HC ? co4 - HC : co4)) * 4);
    if (SOUT) at -= (unsigned)(2 * co4);             // the hi half of the pixel's split record
    const bool to0 = TWO ? true : (EPI == BXE_RELU ? co4 < a.cout : co4 < HC);
    const bool to1 = EPI == BXE_GATES && co4 >= HC && co4 < 2 * HC;
    ooff[nt] = to0 ? at : BUF_OOB;
    ooff1[nt] = to1 ? at : BUF_OOB;
    pin(ooff[nt]); pin(ooff1[nt]);
  }

  auto load_tile = [&](f32x4 (&stage)[NL], int b, int tx, int ty) {
    const int ix0 = tx * TC * STRIDE - 1, iy0 = ty * TR * STRIDE - 1;
    const long pix0 = ((long)b * a.hi + iy0) * a.wi + ix0;
    const buf_rsrc ra = make_rsrc((const char*)a.srcA + pix0 * (CA * 4));
    const buf_rsrc rb = make_rsrc((const char*)a.srcB + pix0 * (CB * 4));
    if (iy0 >= 0 && ix0 >= 0 && iy0 + LR <= a.hi && ix0 + LC <= a.wi) {
#pragma unroll
      for (int k = 0; k < NL; ++k) stage[k] = buf_load4(k < NA ? ra : rb, goff[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int iy = iy0 + (rc[k] & 0xffff), ix = ix0 + (rc[k] >> 16);
        const bool ok = (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi;
        stage[k] = buf_load4(k < NA ? ra : rb, ok ? goff[k] : BUF_OOB);
      }
    }
  };
  auto store_tile = [&](const f32x4 (&stage)[NL]) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      __bf16* hi = (__bf16*)((char*)ldsb + lbyte[k]);
      if (SIN) *(f32x4*)hi = stage[k];
      else split_store(hi, (__bf16*)((char*)hi + LO), stage[k]);
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
    const bool full = oy0 + TR <= a.ho && ox0 + TC <= a.wo;
    const buf_rsrc r0 = make_rsrc((char*)a.dst0 + opix0 * (CO * 4));
    const buf_rsrc r1 = make_rsrc((char*)a.dst1 + opix0 * (HC * 4));
    const buf_rsrc rh = make_rsrc((const char*)a.hsrc + opix0 * (HC * 4));
    const buf_rsrc rin = make_rsrc((const char*)(a.hin ? a.hin : a.dst0) + opix0 * (CO * 4));     // CAND: state in
    unsigned oo[NT], oo1[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { oo[nt] = ooff[nt]; oo1[nt] = ooff1[nt]; }
    if (!full) {
      const bool valid = oy0 + orow < a.ho && ox0 + p < a.wo;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { oo[nt] = valid ? ooff[nt] : BUF_OOB; oo1[nt] = valid ? ooff1[nt] : BUF_OOB; }
    }

    // requests: epilogue operands first, then the next tile
    f32x4 pre_u[NT], pre_h[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (EPI == BXE_CAND && nt * 16 < HC) {
        pre_u[nt] = buf_load4(r1, oo[nt]);
        pre_h[nt] = buf_load4(rin, oo[nt]);
      }
      if (EPI == BXE_GATES && nt * 16 < HC) pre_h[nt] = buf_load4(rh, oo[nt]);      // h of the lane's pixel, fp32
    }
    const int tn = t + nwg;
    const bool more = tn < tr.end;
    int bn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, bn, txn, tyn);
      load_tile(stage, bn, txn, tyn);
    }

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B fragments two k-blocks ahead of the MFMAs that consume them (LDS latency ~ two k-blocks of a 16-row tile)
    constexpr int AHEAD = 2;
    bf16x8 bh[NKB], bl[NKB];
#pragma unroll
    for (int kb = 0; kb < (AHEAD < NKB ? AHEAD : NKB); ++kb) {
      bh[kb] = *(const bf16x8*)((const char*)ldsb + xoff[kb]);
      bl[kb] = *(const bf16x8*)((const char*)ldsb + xoff[kb] + LO);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb + AHEAD < NKB) {
        bh[kb + AHEAD] = *(const bf16x8*)((const char*)ldsb + xoff[kb + AHEAD]);
        bl[kb + AHEAD] = *(const bf16x8*)((const char*)ldsb + xoff[kb + AHEAD] + LO);
      }
      __builtin_amdgcn_sched_barrier(0);          // the scheduler would sink the reads back to their uses
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[nt] = mfma_bx(wh[nt][kb], bh[kb], acc[nt]);
        acc[nt] = mfma_bx(wh[nt][kb], bl[kb], acc[nt]);
        acc[nt] = mfma_bx(wl[nt][kb], bh[kb], acc[nt]);
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) drain(acc[nt]);

    wait_vmem_all();                   // the one wait point of the tile
    __syncthreads();                   // every wave is done reading the tile
    if (more) store_tile(stage);

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 v = acc[nt] + bias[nt];
      if (EPI == BXE_RELU || EPI == BXE_TWO_ROW) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (SOUT) buf_store_split4(r0, oo[nt], (unsigned)(2 * CO), v);
        else buf_store4(r0, oo[nt], v);
      } else if (EPI == BXE_GATES) {
        f32x4 sg = {sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)};
        if (nt * 16 < HC) buf_store4(r0, oo[nt], sg * pre_h[nt]);              // reset-gate rows -> r * h
        if (nt * 16 + 16 > HC) buf_store4(r1, oo1[nt], sg);                     // update-gate rows -> u
      } else if (nt * 16 < HC) {                                                // BXE_CAND
        f32x4 cnd = {tanh_fast(v.x), tanh_fast(v.y), tanh_fast(v.z), tanh_fast(v.w)};
        f32x4 u4 = pre_u[nt], h4 = pre_h[nt];
        buf_store4(r0, oo[nt], u4 * h4 + (1.0f - u4) * cnd);
      }
    }
    if (!more) break;
    __syncthreads();                   // next tile visible
    t = tn; b = bn; tx = txn; ty = tyn;
  }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Level-1 ConvGRU in one kernel (reference models/module.py:28-52 at full stage resolution): the gate convolution
// on cat(x, h), r*h, the candidate convolution on cat(x, r*h) and the state blend, per 8 x 30 tile.  The level-1
// maps are the widest of the recurrence (8 channels at stage resolution) and the split-bf16 chain is short, so
// the separate gate / candidate kernels were bound by their memory traffic: x twice, h, r*h out and in, u out and
// in, h out.  Fused, a tile reads x and h once (12 x 34 window, halo 2) and writes h once; r*h replaces h inside
// the LDS tile (the candidate convolution only needs it split into bf16 halves anyway) and u stays in LDS.
// The new state goes to a second buffer (neighbouring tiles still read the old one); the caller alternates them.
//
// LDS: window [hi|lo][12*34 pixels][x 8 | h 8] bf16, and u [10*32 region pixels][8] fp32.
//   gates at region pixel (rr, rc) = window (rr+1, rc+1), region = tile grown by 1: 10 rows x 2 runs of 16
//   candidate at inner pixel (ir, ic) = window (ir+2, ic+2): 8 rows x 2 runs (columns 30, 31 of a row are surplus)
// Wave k owns runs k, k+4, ... of both convolutions (its B-fragment offsets differ by compile-time constants).
struct Gru1Args {
  const float* x;        // c1 [B][h*w][8] (a split map under BX3_PRESPLIT)
  const float* hin;      // state in  [B][h*w][8]
  float* hout;           // state out [B][h*w][8] (a different buffer; BX3_PRESPLIT: may be hin -- only the lane's own pixel is read)
  const bf16x8* wg; const float* bg;    // gates1 A fragments [1][hi|lo][5][64], bias [16]
  const bf16x8* wc; const float* bc;    // cand1  A fragments [1][hi|lo][5][64], bias [16]
  int h, w;
  const float* hsin;     // BX3_PRESPLIT: the state as a split map, in / out (two different buffers)
  float* hsout;
};

struct Gru1FusedBx3Role {
  typedef Gru1Args Args;
  static constexpr int TR = 8, TC = 30, WR = TR + 4, WC = TC + 4, NPIXW = WR * WC;
  static constexpr int PB = 32;                        // bytes per window pixel: x 8 | h 8 bf16, no pad -- with ds_read_b128's
                                                       // non-contiguous 16-lane groups a 32-byte pitch is conflict-free here, 48 is 2-way
  static constexpr int LO = NPIXW * PB;                // lo image
  static constexpr int U0 = 2 * LO;                    // u tile
  static constexpr int NKB = 5, NG = 5;                // gates: k-blocks (9 taps x 16 channels), runs per wave
  // The candidate convolution has 8 output channels -- half an MFMA tile -- and runs in the TWO-ROW form (round 5; the form of conv1
  // and of the fp32 cand1): MFMA rows 0-7 are inner row R, rows 8-15 inner row R + 1, k = (window row rr 0..3, kx, channel) = 192 =
  // 6 k-blocks for the PAIR of rows where two one-row runs took 10 -- 18 MFMAs and 12 fragment reads instead of 30 and 20 -- and
  // all four lane groups carry outputs (row R + (q >> 1), channels 4 (q & 1) ..): half the transcendentals per lane.
  // -DBX3_CAND_TWO_ROW=0: one-row runs (cand1 then packed by pack_small_conv_bf16x3; A/B).
  static constexpr bool C2R = BX3_CAND_TWO_ROW != 0;
  static constexpr int NKC = C2R ? 6 : 5, NC = C2R ? 2 : 4;      // candidate: k-blocks, runs per wave
  static constexpr int NITEM = NPIXW * 2, NS = (NITEM + 255) / 256;      // 4-channel groups per source, loads per thread
  static constexpr size_t LDS_BYTES = (size_t)2 * 12 * 34 * 32 + 10 * 32 * 32;
  static constexpr int TILE_W = TC, TILE_H = TR;
  static int tiles_x(const Args& a) { return cdiv(a.w, TC); }
  static int tiles_y(const Args& a) { return cdiv(a.h, TR); }

  static __device__ __forceinline__ void run(const Args& a, const TileGrid& tg, TileRange tr, int wg, int nwg, float* lds_) {
  char* lds = (char*)lds_;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int rr0 = wave >> 1, c0w = (wave & 1) * 16;    // first run of the wave: region row / first column

  bf16x8 gh[NKB], gl[NKB], ch[NKC], cl[NKC];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) { gh[kb] = a.wg[(0 * NKB + kb) * 64 + lane]; gl[kb] = a.wg[(1 * NKB + kb) * 64 + lane]; }
#pragma unroll
  for (int kb = 0; kb < NKC; ++kb) { ch[kb] = a.wc[(0 * NKC + kb) * 64 + lane]; cl[kb] = a.wc[(1 * NKC + kb) * 64 + lane]; }
  const f32x4 bias_g = *(const f32x4*)(a.bg + 4 * q);
  const f32x4 bias_c = *(const f32x4*)(a.bc + 4 * (C2R ? (q & 1) : q));

  unsigned goff[NS], lbyte[NS];
  int rc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    int j = min(tid + k * 256, NITEM - 1);
    const int g = j & 1, pp = j >> 1, r = pp / WC, c = pp % WC;
    goff[k] = (unsigned)(((r * a.w + c) * 8 + 4 * g) * 4);
    lbyte[k] = BX3_PRESPLIT ? (unsigned)(pp * PB + g * LO) : (unsigned)(pp * PB + 8 * g);      // piece g: the hi / the lo half of the pixel
    rc[k] = r | (c << 16);
    pin(goff[k]); pin(lbyte[k]); pin(rc[k]);
  }
  unsigned xoff[NKB];                                  // gate run 0 of the wave; candidate runs add (WC + 1) * PB
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    int kk = 32 * kb + 8 * q;
    int pos = kk / 16, ch0 = kk % 16;
    if (pos >= 9) { pos = 0; ch0 = 0; }
    xoff[kb] = (unsigned)((((rr0 + pos / 3) * WC + c0w + p + pos % 3) * PB) + ch0 * 2);
    pin(xoff[kb]);
  }
  // gate epilogue.  The rows of the gate convolution are PERMUTED on the host (packing.pack_slice_reg_net, bf16x3): MFMA row
  // 4q + e is reset-gate channel 2q + e (e = 0, 1) or update-gate channel 2q + e - 2 (e = 2, 3), so every lane holds two reset
  // and two update values of its pixel -- channels 2q, 2q + 1 -- and does a quarter of the r*h work and a quarter of the u work.
  // (In the reference's order, r = rows 0-7, u = rows 8-15, lanes q < 2 did all of the r*h work -- state halves read back,
  // product, split, store -- while lanes q >= 2 waited behind the exec mask: 33 vector instructions per run and lane, now 24;
  // the gate epilogue was the largest share of the kernel's 3.4 vector instructions per MFMA.)
  const unsigned hbyte = (unsigned)(((rr0 + 1) * WC + c0w + p + 1) * PB + (8 + 2 * q) * 2);
  const unsigned ubyte_w = (unsigned)(U0 + ((rr0 * 32 + c0w + p) * 8 + 2 * q) * 4);
  // candidate epilogue, inner pixel (crow + CSTEP j, c0w + p), channels 4 (q & 1) ..: u of region (ir+1, ic+1).
  // One-row runs: lanes q < 2, rows rr0 + 2 j.  Two-row: every lane, rows 2 (wave >> 1) + (q >> 1) + 4 j.
  constexpr int CSTEP = C2R ? 4 : 2;
  const int crow0 = C2R ? 2 * rr0 : rr0, crow = C2R ? crow0 + (q >> 1) : rr0;
  const unsigned ubyte_r = (unsigned)(U0 + (((crow + 1) * 32 + c0w + p + 1) * 8 + 4 * (q & 1)) * 4);
  const bool lane_out = (C2R || q < 2) && c0w + p < TC;
  unsigned ooff = lane_out ? (unsigned)(((crow * a.w + c0w + p) * 8 + 4 * (q & 1)) * 4) : BUF_OOB;
  const unsigned orow2 = (unsigned)(a.w * 32 * CSTEP); // CSTEP rows of the state maps, bytes
  pin(ooff);
  unsigned xoffc[NKC];                                 // candidate run 0 of the wave
#pragma unroll
  for (int kb = 0; kb < NKC; ++kb) {
    if (C2R) {                                         // k-block = (window row rr, kx) pairs: inner row R reads window rows R + 1 + rr
      const int kk = 32 * kb + 8 * q, pos = kk / 16, ch0 = kk % 16;
      xoffc[kb] = (unsigned)((((crow0 + 1 + pos / 3) * WC + c0w + p + 1 + pos % 3) * PB) + ch0 * 2);
    } else {
      xoffc[kb] = xoff[kb] + (WC + 1) * PB;
    }
    pin(xoffc[kb]);
  }

  auto load_tile = [&](f32x4 (&sx)[NS], f32x4 (&sh)[NS], int b, int tx, int ty) {
    const int ix0 = tx * TC - 2, iy0 = ty * TR - 2;
    const long pix0 = ((long)b * a.h + iy0) * a.w + ix0;
    const buf_rsrc rx = make_rsrc((const char*)a.x + pix0 * 32);
    const buf_rsrc rh = make_rsrc((const char*)(BX3_PRESPLIT ? a.hsin : a.hin) + pix0 * 32);
    if (iy0 >= 0 && ix0 >= 0 && iy0 + WR <= a.h && ix0 + WC <= a.w) {
#pragma unroll
      for (int k = 0; k < NS; ++k) { sx[k] = buf_load4(rx, goff[k]); sh[k] = buf_load4(rh, goff[k]); }
    } else {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const int iy = iy0 + (rc[k] & 0xffff), ix = ix0 + (rc[k] >> 16);
        const bool ok = (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
        const unsigned o = ok ? goff[k] : BUF_OOB;
        sx[k] = buf_load4(rx, o); sh[k] = buf_load4(rh, o);
      }
    }
  };
  auto store_tile = [&](const f32x4 (&sx)[NS], const f32x4 (&sh)[NS]) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      __bf16* hi = (__bf16*)(lds + lbyte[k]);
      if (BX3_PRESPLIT) {
        *(f32x4*)hi = sx[k];
        *(f32x4*)(hi + 8) = sh[k];
      } else {
        split_store(hi, (__bf16*)((char*)hi + LO), sx[k]);
        split_store(hi + 8, (__bf16*)((char*)hi + LO) + 8, sh[k]);
      }
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
    const long opix0 = ((long)b * a.h + oy0) * a.w + ox0;
    const buf_rsrc rin = make_rsrc((const char*)a.hin + opix0 * 32);
    const buf_rsrc rout = make_rsrc((char*)a.hout + opix0 * 32);
    const buf_rsrc rsout = make_rsrc((char*)a.hsout + opix0 * 32);
    const bool full = oy0 + TR <= a.h && ox0 + TC <= a.w;
    unsigned oo[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      oo[j] = ooff + j * orow2;
      if (!full && !(oy0 + crow + CSTEP * j < a.h && ox0 + c0w + p < a.w)) oo[j] = BUF_OOB;
    }
    f32x4 pre_h[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) pre_h[j] = buf_load4(rin, oo[j]);       // exact fp32 state for the blend
    const int tn = t + nwg;
    const bool more = tn < tr.end;
    int bn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, bn, txn, tyn);
      if (!(BX3_EXP & 8)) load_tile(sx, sh, bn, txn, tyn);
    }

    // ---- gates on cat(x, h)
    // A wave's runs are two rows apart, so k-block 3 of run j -- taps (2, 0) and (2, 1) -- IS k-block 0 of run j + 1 -- taps (0, 0)
    // and (0, 1) two rows further down: kept in registers, 42 instead of 50 fragment reads
    // The fragment reads run BX3_AHEAD k-blocks ahead of the MFMAs that consume them, across the runs (the compiler's own schedule
    // waited for every k-block's two reads right after issuing them: with two waves per SIMD the LDS latency was exposed 25 times per
    // tile: 1 - 4 % of the kernel).  One flat sequence of steps s = (run, k-block);
    // everything is unrolled, so the fragment arrays are names for registers, and a reused k-block is the same registers again.
    f32x4 ag[NG];
    {
      constexpr int NSTEP = NG * NKB, AH = BX3_AHEAD;
      bf16x8 fh[NSTEP], fl[NSTEP];
      auto rd = [&](int s) {
        const int j = s / NKB, kb = s % NKB;
        if (BX3_GATE_REUSE && kb == 0 && j > 0) { fh[s] = fh[s - 2]; fl[s] = fl[s - 2]; return; }      // = (run j - 1, k-block 3)
        const char* at = lds + xoff[kb] + ((BX3_EXP & 2) ? 0 : j) * (2 * WC * PB);
        fh[s] = *(const bf16x8*)at;
        fl[s] = *(const bf16x8*)(at + LO);
      };
#pragma unroll
      for (int s = 0; s < AH; ++s) rd(s);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const int j = s / NKB, kb = s % NKB;
        if (s + AH < NSTEP) rd(s + AH);
        if (AH) __builtin_amdgcn_sched_barrier(0);     // (the scheduler would sink the reads back to their uses)
        if (kb == 0) ag[j] = bias_g;
        if (BX3_EXP & 1) { keep(fh[s]); keep(fl[s]); continue; }
        ag[j] = mfma_bx(gh[kb], fh[s], ag[j]);
        ag[j] = mfma_bx(gh[kb], fl[s], ag[j]);
        ag[j] = mfma_bx(gl[kb], fh[s], ag[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) drain(ag[j]);
    __syncthreads();                   // nobody reads the old h halves any more
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const f32x4 v = ag[j];
      const f32x4 sg = (BX3_EXP & 4) ? v : f32x4{sigmoid_pre(v.x), sigmoid_pre(v.y), sigmoid_pre(v.z), sigmoid_pre(v.w)};      // r(2q), r(2q+1), u(2q), u(2q+1)
      __bf16* hi = (__bf16*)(lds + hbyte + j * (2 * WC * PB));
      __bf16* lo = (__bf16*)((char*)hi + LO);
      const bf16x2 h2 = *(const bf16x2*)hi, l2 = *(const bf16x2*)lo;
      const float r0 = sg.x * ((float)h2.x + (float)l2.x), r1 = sg.y * ((float)h2.y + (float)l2.y);
      const bf16x2 nh = {(__bf16)r0, (__bf16)r1};
      const bf16x2 nl = {(__bf16)(r0 - (float)nh.x), (__bf16)(r1 - (float)nh.y)};
      *(bf16x2*)hi = nh;
      *(bf16x2*)lo = nl;
      *(f32x2b*)(lds + ubyte_w + j * (2 * 32 * 32)) = f32x2b{sg.z, sg.w};
    }
    __syncthreads();                   // r*h and u visible

    // ---- candidate on cat(x, r*h)
    f32x4 ac[NC];
    {
      constexpr int NSTEP = NC * NKC, AH = BX3_AHEAD;
      bf16x8 fh[NSTEP], fl[NSTEP];
      auto rd = [&](int s) {
        const int j = s / NKC, kb = s % NKC;
        const char* at = lds + xoffc[kb] + ((BX3_EXP & 2) ? 0 : j) * (CSTEP * WC * PB);
        fh[s] = *(const bf16x8*)at;
        fl[s] = *(const bf16x8*)(at + LO);
      };
#pragma unroll
      for (int s = 0; s < AH; ++s) rd(s);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const int j = s / NKC, kb = s % NKC;
        if (s + AH < NSTEP) rd(s + AH);
        if (AH) __builtin_amdgcn_sched_barrier(0);
        if (kb == 0) ac[j] = bias_c;
        if (BX3_EXP & 1) { keep(fh[s]); keep(fl[s]); continue; }
        ac[j] = mfma_bx(ch[kb], fh[s], ac[j]);
        ac[j] = mfma_bx(ch[kb], fl[s], ac[j]);
        ac[j] = mfma_bx(cl[kb], fh[s], ac[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) drain(ac[j]);
    f32x4 u4[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) u4[j] = *(const f32x4*)(lds + ubyte_r + j * (CSTEP * 32 * 32));

    wait_vmem_all();                   // the one wait point of the tile
    __syncthreads();                   // every wave is done with the tile
    if (more && !(BX3_EXP & 8)) store_tile(sx, sh);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const f32x4 v = ac[j];
      const f32x4 cnd = (BX3_EXP & 4) ? v : f32x4{tanh_pre(v.x), tanh_pre(v.y), tanh_pre(v.z), tanh_pre(v.w)};
      const f32x4 hn = gru_blend(u4[j], pre_h[j], cnd);
      buf_store4(rout, oo[j], hn);
      if (BX3_PRESPLIT) buf_store_split4(rsout, oo[j] == BUF_OOB ? BUF_OOB : oo[j] - 8 * (q & 1), 16, hn);
    }
    if (!more) break;
    __syncthreads();                   // next tile visible
    t = tn; b = bn; tx = txn; ty = tyn;
  }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Level-2 ConvGRU in one kernel (reference models/module.py:28-52 on cat(conv2 output, state2), 16 hidden channels at half
// the stage resolution): the twin of Gru1FusedBx3Role.  The separate gate / candidate kernels exchanged r*h and u through
// memory and each read c2 and the state with their halo (816 B per pixel); fused, a tile reads c2 and the state once
// (12 x 18 window, halo 2, for 8 x 14 pixels: 247 B per pixel) and writes the new state.
// The gate convolution has 32 output rows = two MFMA row tiles, 144 registers of split weights -- too many next to the
// candidate's 72 -- so the waves specialise: waves 0, 1 own the reset-gate rows, waves 2, 3 the update-gate rows, each
// over half of the ten region rows (the 8 x 14 tile grown by one pixel, one 16-pixel run per row); all four then share the
// candidate (two of the eight rows each).  r*h goes, split, to a buffer of its own (the window keeps the state: no
// barrier between a gate run and its epilogue, one accumulator live), where the lanes that carry the state channels of
// the candidate's B fragments read it; u waits in LDS as fp32.
//   LDS: window [hi|lo][12*18 pixels][c2 16 | h 16 | pad 16] bf16; u [10*16 region pixels][16] fp32;
//        r*h [hi|lo][10 rows x 18 (16 + 2 never-written columns that only the two surplus lanes of a run read)][16] bf16.
struct Gru2Args {
  const float* x;        // conv2 output [B][h*w][16]     (h, w: the level-2 size; a split map under BX3_PRESPLIT)
  const float* hin;      // state in  [B][h*w][16]
  float* hout;           // state out [B][h*w][16] (a different buffer; BX3_PRESPLIT: may be hin)
  const bf16x8* wg; const float* bg;    // gates2 A fragments [2][hi|lo][9][64], bias [32]
  const bf16x8* wc; const float* bc;    // cand2  A fragments [1][hi|lo][9][64], bias [16]
  int h, w;
  const float* hsin;     // BX3_PRESPLIT: the state as a split map, in / out (two different buffers)
  float* hsout;
};

struct Gru2FusedBx3Role {
  typedef Gru2Args Args;
  static constexpr int TR = 8, TC = 14, WR = TR + 4, WC = TC + 4, NPIXW = WR * WC;
  static constexpr int PB = 2 * bx_pixel_pitch(32);    // bytes per window pixel
  static constexpr int LO = NPIXW * PB;                // lo image of the window
  static constexpr int U0 = 2 * LO;                    // u tile
  static constexpr int RH0 = U0 + 10 * 16 * 64;        // r*h buffer
  static constexpr int RC = 18, RPB = 2 * bx_pixel_pitch(16), RLO = 10 * RC * RPB;      // its row length, pixel pitch, lo image
  static constexpr int NKB = 9, NG = 5, NC = 2;        // k-blocks (9 taps x 32 channels), gate / candidate runs per wave
  static constexpr int NITEM = NPIXW * 4, NS = (NITEM + 255) / 256;      // 4-channel groups per source, loads per thread
  static constexpr int BIAS0 = RH0 + 2 * RLO;          // gate bias [32], candidate bias [16] (fp32): read where they are used
  static constexpr size_t LDS_BYTES = (size_t)BIAS0 + 48 * 4;
  static_assert(LDS_BYTES <= 64 * 1024, "default dynamic LDS limit");
  static constexpr int TILE_W = TC, TILE_H = TR;
  static int tiles_x(const Args& a) { return cdiv(a.w, TC); }
  static int tiles_y(const Args& a) { return cdiv(a.h, TR); }

  static __device__ __forceinline__ void run(const Args& a, const TileGrid& tg, TileRange tr, int wg, int nwg, float* lds_) {
  char* lds = (char*)lds_;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int half = wave >> 1, rr0 = wave & 1;          // gate rows (0 reset, 1 update); first region row of the wave

  bf16x8 gh[NKB], gl[NKB], ch[NKB], cl[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    gh[kb] = a.wg[((half * 2 + 0) * NKB + kb) * 64 + lane]; gl[kb] = a.wg[((half * 2 + 1) * NKB + kb) * 64 + lane];
    ch[kb] = a.wc[(0 * NKB + kb) * 64 + lane]; cl[kb] = a.wc[(1 * NKB + kb) * 64 + lane];
  }
  if (tid < 48) ((float*)(lds + BIAS0))[tid] = tid < 32 ? a.bg[tid] : a.bc[tid - 32];      // visible after the first barrier
  const unsigned bgbyte = (unsigned)(BIAS0 + (half * 16 + 4 * q) * 4), bcbyte = (unsigned)(BIAS0 + (32 + 4 * q) * 4);

  unsigned goff[NS], lbyte[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    int j = min(tid + k * 256, NITEM - 1);
    const int g = j & 3, pp = j >> 2, r = pp / WC, c = pp % WC;
    goff[k] = (unsigned)(((r * a.w + c) * 16 + 4 * g) * 4);
    lbyte[k] = BX3_PRESPLIT ? (unsigned)(pp * PB + 16 * (g & 1) + (g >> 1) * LO) : (unsigned)(pp * PB + 8 * g);   // piece g: 8 channels of hi (g < 2) or lo
    pin(goff[k]); pin(lbyte[k]);
  }
  // B fragments: k-block = tap, lane q: channels 8q .. 8q+7 of cat(c2, state).  Gate run 0 of the wave reads the window;
  // candidate run 0 (inner row `wave`) reads c2 (q < 2) from the window and r*h (q >= 2) from its buffer.
  // All lanes of a k-block share its tap (32 channels = one k-block per tap), so a fragment address is a lane base plus a
  // compile-time offset; the candidate's lanes differ in geometry (window or r*h buffer), chosen per lane at the read.
  unsigned xg = (unsigned)((rr0 * WC + p) * PB + 16 * q);
  const bool from_window = q < 2;
  unsigned xc = from_window ? (unsigned)(((wave + 1) * WC + p + 1) * PB + 16 * q) : (unsigned)(RH0 + (wave * RC + p) * RPB + 16 * (q - 2));
  pin(xg); pin(xc);
  const unsigned hbyte = (unsigned)(((rr0 + 1) * WC + p + 1) * PB + (16 + 4 * q) * 2);            // the state of the lane's region pixel
  const unsigned rbyte = (unsigned)(RH0 + (rr0 * RC + p) * RPB + 8 * q);                          // its r*h
  const unsigned ubyte_w = (unsigned)(U0 + ((rr0 * 16 + p) * 16 + 4 * q) * 4);
  const unsigned ubyte_r = (unsigned)(U0 + (((wave + 1) * 16 + p + 1) * 16 + 4 * q) * 4);        // inner (ir, p) = region (ir+1, p+1)
  unsigned ooff = p < TC ? (unsigned)(((wave * a.w + p) * 16 + 4 * q) * 4) : BUF_OOB;
  const unsigned orow4 = (unsigned)(a.w * 4 * 64);     // four rows of the state maps, bytes
  pin(ooff);

  auto load_tile = [&](f32x4 (&sx)[NS], f32x4 (&sh)[NS], int b, int tx, int ty) {
    const int ix0 = tx * TC - 2, iy0 = ty * TR - 2;
    const long pix0 = ((long)b * a.h + iy0) * a.w + ix0;
    const buf_rsrc rx = make_rsrc((const char*)a.x + pix0 * 64);
    const buf_rsrc rh = make_rsrc((const char*)(BX3_PRESPLIT ? a.hsin : a.hin) + pix0 * 64);
    if (iy0 >= 0 && ix0 >= 0 && iy0 + WR <= a.h && ix0 + WC <= a.w) {
#pragma unroll
      for (int k = 0; k < NS; ++k) { sx[k] = buf_load4(rx, goff[k]); sh[k] = buf_load4(rh, goff[k]); }
    } else {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const int pp = min(tid + k * 256, NITEM - 1) >> 2;          // (edge tiles only: not worth a register per item)
        const int iy = iy0 + pp / WC, ix = ix0 + pp % WC;
        const bool ok = (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
        const unsigned o = ok ? goff[k] : BUF_OOB;
        sx[k] = buf_load4(rx, o); sh[k] = buf_load4(rh, o);
      }
    }
  };
  auto store_tile = [&](const f32x4 (&sx)[NS], const f32x4 (&sh)[NS]) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      __bf16* hi = (__bf16*)(lds + lbyte[k]);
      if (BX3_PRESPLIT) {
        *(f32x4*)hi = sx[k];
        *(f32x4*)(hi + 16) = sh[k];
      } else {
        split_store(hi, (__bf16*)((char*)hi + LO), sx[k]);
        split_store(hi + 16, (__bf16*)((char*)hi + LO) + 16, sh[k]);
      }
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
    const long opix0 = ((long)b * a.h + oy0) * a.w + ox0;
    const buf_rsrc rin = make_rsrc((const char*)a.hin + opix0 * 64);
    const buf_rsrc rout = make_rsrc((char*)a.hout + opix0 * 64);
    const buf_rsrc rsout = make_rsrc((char*)a.hsout + opix0 * 64);
    const bool full = oy0 + TR <= a.h && ox0 + TC <= a.w;
    unsigned oo[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      oo[j] = ooff == BUF_OOB ? BUF_OOB : ooff + j * orow4;
      if (!full && !(oy0 + wave + 4 * j < a.h && ox0 + p < a.w)) oo[j] = BUF_OOB;
    }

    // ---- the wave's gate rows on cat(c2, h): region rows rr0, rr0 + 2, ...; r*h (from the split state of the window:
    //      what the candidate convolution multiplies is a split value anyway) and u go to their own buffers
#pragma unroll 1                       // (unrolled, the scheduler interleaves the five chains: spills.  Keeping the taps (2, kx) of a
    for (int j = 0; j < NG; ++j) {     //  run in registers as the taps (0, kx) of the next, as level 1 does: 24 registers, spills, 82 -> 91 us)
      f32x4 ag = *(const f32x4*)(lds + bgbyte);
      {                                // fragment reads BX3_AHEAD2 k-blocks ahead of their MFMAs, inside the run (as level 1 does across runs)
        constexpr int AH = BX3_AHEAD2;
        bf16x8 fh[NKB], fl[NKB];
        auto rd = [&](int kb) {
          const char* at = lds + xg + ((kb / 3) * WC + kb % 3) * PB + j * (2 * WC * PB);
          fh[kb] = *(const bf16x8*)at;
          fl[kb] = *(const bf16x8*)(at + LO);
        };
#pragma unroll
        for (int kb = 0; kb < AH; ++kb) rd(kb);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          if (kb + AH < NKB) rd(kb + AH);
          if (AH) __builtin_amdgcn_sched_barrier(0);
          ag = mfma_bx(gh[kb], fh[kb], ag);
          ag = mfma_bx(gh[kb], fl[kb], ag);
          ag = mfma_bx(gl[kb], fh[kb], ag);
        }
      }
      drain(ag);
      const f32x4 sg = {sigmoid_pre(ag.x), sigmoid_pre(ag.y), sigmoid_pre(ag.z), sigmoid_pre(ag.w)};
      if (half == 0) {
        const __bf16* hi = (const __bf16*)(lds + hbyte + j * (2 * WC * PB));
        const bf16x4 h4 = *(const bf16x4*)hi, l4 = *(const bf16x4*)((const char*)hi + LO);
        const f32x4 hv = {(float)h4.x + (float)l4.x, (float)h4.y + (float)l4.y, (float)h4.z + (float)l4.z, (float)h4.w + (float)l4.w};
        __bf16* rh = (__bf16*)(lds + rbyte + j * (2 * RC * RPB));
        split_store(rh, (__bf16*)((char*)rh + RLO), sg * hv);
      } else {
        *(f32x4*)(lds + ubyte_w + j * (2 * 16 * 64)) = sg;
      }
    }
    __syncthreads();                   // r*h and u visible

    // ---- request: the exact fp32 state of the lane's pixels for the blend
    f32x4 pre_h[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) pre_h[j] = buf_load4(rin, oo[j]);

    // ---- candidate on cat(c2, r*h): inner rows wave, wave + 4
    f32x4 ac[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      ac[j] = *(const f32x4*)(lds + bcbyte);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const unsigned off = from_window ? (unsigned)(((kb / 3 + 4 * j) * WC + kb % 3) * PB) : (unsigned)(((kb / 3 + 4 * j) * RC + kb % 3) * RPB);
        const char* at = lds + xc + off;
        const bf16x8 bh = *(const bf16x8*)at;
        const bf16x8 bl = *(const bf16x8*)(at + (from_window ? LO : RLO));
        ac[j] = mfma_bx(ch[kb], bh, ac[j]);
        ac[j] = mfma_bx(ch[kb], bl, ac[j]);
        ac[j] = mfma_bx(cl[kb], bh, ac[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) drain(ac[j]);
    f32x4 u4[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) u4[j] = *(const f32x4*)(lds + ubyte_r + j * (4 * 16 * 64));

    // the next window is requested only now: its 32 staging registers do not fit next to the chains (144 registers of
    // weights); the blend and the other workgroup of the CU cover part of its latency
    const int tn = t + nwg;
    const bool more = tn < tr.end;
    int bn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, bn, txn, tyn);
      load_tile(sx, sh, bn, txn, tyn);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const f32x4 v = ac[j];
      const f32x4 cnd = {tanh_pre(v.x), tanh_pre(v.y), tanh_pre(v.z), tanh_pre(v.w)};
      const f32x4 hn = gru_blend(u4[j], pre_h[j], cnd);
      buf_store4(rout, oo[j], hn);
      if (BX3_PRESPLIT) buf_store_split4(rsout, oo[j] == BUF_OOB ? BUF_OOB : oo[j] - 8 * q, 32, hn);
    }
    if (!more) break;
    __syncthreads();                   // every wave is done with the tile
    wait_vmem_all();
    store_tile(sx, sh);
    __syncthreads();                   // next tile visible
    t = tn; b = bn; tx = txn; ty = tyn;
  }
  }
};

}  // namespace adamvs
