// Device-side "roles" of the recurrent cost regularisation: the persistent tile loops of the small fp32 MFMA
// convolutions (gates / candidate / conv2 of the two ConvGRU levels) and of the decoder, as device functions that
// walk a range of tiles with a given workgroup index and workgroup count.  A role is what used to be one kernel;
// slice_red.hip wraps each into a kernel of its own (the one-step op), recurrence.hip runs several roles of
// different hypothesis steps side by side in one launch (the software-pipelined recurrence).
//
//   SliceCostRegNetRED.forward   reference models/adamvs.py:415-424
//   ConvGRUCell.forward          reference models/module.py:24-52
#pragma once
#include "common.h"
#include "conv_frag.h"
#include "kernels.h"
#include "persistent.h"

namespace adamvs {

enum { EPI_RELU = 0, EPI_GATES = 1, EPI_CAND = 2, EPI_LINEAR = 3 };   // LINEAR: out = conv + bias (MS-REDNet cells)

struct SmallConvArgs {
  const float* srcA;   // [B][hi*wi][CA]
  const float* srcB;   // [B][hi*wi][CB] (null when CB == 0)
  const float* wpk;    // A fragments [NT][9][(CA+CB)/4][64]
  const float* bias;   // [16*NT], zero padded (GATES, CAND)
  float* dst0;         // RELU: out [B][ho*wo][cout];  GATES: r*h [B][..][HC];  CAND: h, updated in place
  float* dst1;         // GATES: u out [B][..][HC];    CAND: u in
  int hi, wi, ho, wo, cout;
  const float* hin;    // CAND: the state that is blended (h of the previous step); null = dst0 (update in place)
  // LINEAR, MS-REDNet: GroupNorm(1 group) statistics of the output in the epilogue (reference models/module.py:62-67: the
  // normalisation that follows gate_conv / output_conv): every (tile, wave) writes the sum and the sum of squares (double) of
  // its real output values of channel group g = channel / gn_hc < gn_groups to
  // gn_part[((b * gn_groups + g) * parts + (ty * tiles_x + tx) * 4 + wave) * 2], parts = 4 * tiles per map; null: none
  double* gn_part;
  int gn_hc, gn_groups;
};

// Range of tiles a role instance walks: tiles [begin, end) of its TileGrid.
struct TileRange { int begin, end; };


// Persistent workgroups.  These launches are short (thousands of tiles of about a microsecond) and sit on the
// sequential critical path of the recurrence, so what costs time is not the matrix work but its packaging.
//  * The grid is exactly the resident capacity (occupancy query); workgroup i walks tiles i, i + grid, ...
//    (A shared atomic tile counter was tried and is slower: one word serves ~88 dequeues/us.)
//  * A tile is TR rows x 16*RW columns = four 16-pixel runs, one per wave.
//  * fp32 MFMA and the vector ALU are the same lanes (the fp32 matrix rate of the chip IS its packed-fp32 vector
//    rate): every VALU instruction is a slot an MFMA does not get, whatever the occupancy.  Measured here: with
//    loads, epilogue and barriers stripped the kernels run at the MFMA bound; each phase put back added its
//    VALU instruction count, nothing else.  So everything per-lane that does not depend on the tile (LDS
//    addresses, offsets inside the tile window, output offsets) is computed once before the loop, the tile
//    enters only through workgroup-uniform base pointers (scalar unit; global accesses are base + 32-bit lane
//    offset), interior tiles take a path with no bounds checks, and the gate non-linearities use v_rcp/v_exp
//    directly.
//  * One wait point per tile.  vmcnt retires in order, so any wait on a late load also waits for everything
//    issued before it.  Per tile: request the epilogue operands of this tile and the input of the NEXT tile; run
//    the MFMA chain out of LDS; only then wait, barrier, refill the LDS tile, epilogue, stores, barrier.
//    Nothing is waited for before it has had a whole MFMA phase to arrive; the stores drain under the next chain.
template <int CA, int CB, int NT, int STRIDE, int EPI, int TR = 4, int RW = 1>
struct ConvSmallRole {
  typedef SmallConvArgs Args;
  static_assert(TR * RW == 4, "one run per wave");
  static constexpr int CIN = CA + CB, KC = CIN / 4, G = CIN / 4, GA = CA / 4, GB = CB / 4, TC = 16 * RW;
  static constexpr int LR = (STRIDE == 1) ? TR + 2 : 2 * TR + 1;
  static constexpr int LC = (STRIDE == 1) ? TC + 2 : 2 * TC + 1;
  static constexpr int NPIX = LR * LC;
  static constexpr int PLANE = (STRIDE == 1) ? plane_pitch16(NPIX) : (NPIX | 1);
  static constexpr int GP = group_pitch(PLANE, G);
  static constexpr int HC = CB;          // hidden width for the GRU epilogues
  // one load instruction covers 256 (pixel, channel group) items of ONE source, so that its base is uniform
  static constexpr int NA = (NPIX * GA + 255) / 256, NB = (NPIX * GB + 255) / 256, NL = NA + NB;
  static constexpr size_t LDS_BYTES = (size_t)G * GP * sizeof(float);         // tile [G][GP]
  static constexpr int TILE_W = TC, TILE_H = TR;                               // output pixels per tile
  static int tiles_x(const Args& a) { return cdiv(a.wo, TC); }
  static int tiles_y(const Args& a) { return cdiv(a.ho, TR); }

  // workgroup `wg` of `nwg` walks tiles tr.begin + wg, + nwg, ... < tr.end
  // `pro` (LINEAR, MS-REDNet; its own kernel argument: the recurrence's launches do not carry it): srcB formed on the fly from
  // the previous kernels' maps (kernels.h, GruPro; srcB itself is the state h) -- GRU_PRO_GATES: r * h; GRU_PRO_OUT:
  // h' = u h + (1 - u) tanh(GN(o)), also stored for the tile's own pixels
  static __device__ __forceinline__ void run(const Args& a, const TileGrid& tg, TileRange tr, int wg, int nwg, float* lds,
                                             const GruPro& pro = GruPro{}) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
  const int p = lane & 15, q = lane >> 4;
  const int row = wave / RW, col = (wave % RW) * 16;      // the wave's run inside a tile

  float wf[NT][9][KC];
  load_wfrag<NT, KC>(wf, a.wpk, lane);
  f32x4 bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    bias[nt] = (EPI == EPI_RELU) ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(a.bias + nt * 16 + 4 * q);

  // ---- per-lane constants of the tile window
  unsigned goff[NL];      // byte offset of the item from the window's first pixel in its source
  unsigned lbyte[NL];     // LDS byte offset of the item's first channel plane
  int rc[NL];             // window row | column << 16 (edge tiles only)
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const bool isA = k < NA;
    const int gs = isA ? GA : GB, cs = isA ? CA : CB;
    int j = tid + (isA ? k : k - NA) * 256;
    j = min(j, NPIX * gs - 1);           // surplus lanes repeat the last item (same value to the same place)
    const int g = j % gs, pp = j / gs, r = pp / LC, c = pp % LC;
    goff[k] = (unsigned)(((r * a.wi + c) * cs + 4 * g) * 4);
    lbyte[k] = (unsigned)((((isA ? 0 : GA) + g) * GP + r * LC + c) * 4);
    rc[k] = r | (c << 16);
    pin(goff[k]); pin(lbyte[k]); pin(rc[k]);
  }
  unsigned xbyte[KC];     // B-fragment origin of the run, per k-chunk
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    xbyte[kc] = (unsigned)((kc * GP + q * PLANE + (row * STRIDE) * LC + (col + p) * STRIDE) * 4);
    pin(xbyte[kc]);
  }
  unsigned hbyte[NT];     // GATES: hidden-state channels co4..co4+3 of the lane's own pixel
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    hbyte[nt] = (unsigned)((((CA + min(nt * 16 + 4 * q, HC - 4)) >> 2) * GP + (row + 1) * LC + col + p + 1) * 4);
    pin(hbyte[nt]);
  }
  // output offsets (bytes) of the lane's pixel inside the tile, per 16-channel slice
  const int CO = (EPI == EPI_RELU || EPI == EPI_LINEAR) ? a.cout : HC;
  // ooff: destination 0 (RELU out, GATES r*h, CAND h); ooff1: destination 1 (GATES u).  Rows that do not
  // belong to a destination carry BUF_OOB, so every store is issued by all lanes with one uniform descriptor.
  unsigned ooff[NT], ooff1[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co4 = nt * 16 + 4 * q;
    const unsigned at = (unsigned)(((row * a.wo + col + p) * CO + (EPI == EPI_GATES && co4 >= HC ? co4 - HC : co4)) * 4);
    const bool to0 = (EPI == EPI_RELU || EPI == EPI_LINEAR) ? co4 < a.cout : co4 < HC;
    const bool to1 = EPI == EPI_GATES && co4 >= HC && co4 < 2 * HC;
    ooff[nt] = to0 ? at : BUF_OOB;
    ooff1[nt] = to1 ? at : BUF_OOB;
    pin(ooff[nt]); pin(ooff1[nt]);
  }

  // MS-REDNet (LINEAR): srcB formed on the fly (GruPro); the f map is 2 CB wide, the o map CB wide like srcB = h
  constexpr int NBP = (EPI == EPI_LINEAR && NB > 0) ? NB : 1;
  constexpr int GBP = GB > 0 ? GB : 1;
  const bool pro_on = EPI == EPI_LINEAR && pro.mode != GRU_PRO_NONE;            // uniform
  unsigned foff[NBP];
  f32x4 gfw[NBP], gfb[NBP], gow[NBP], gob[NBP];
  f32x4 stageF[NBP], stageO[NBP];
  float pst[4] = {0.f, 1.f, 0.f, 1.f};   // mean, 1/sigma of f's group; of o
  int pst_b = -1;                        // the sample they belong to
  if (EPI == EPI_LINEAR && pro_on) {
#pragma unroll
    for (int k = 0; k < NBP; ++k) {
      const int j = min(tid + k * 256, NPIX * GB - 1);
      const int g = j % GBP, pp = j / GBP, r = pp / LC, c = pp % LC;
      foff[k] = (unsigned)(((r * a.wi + c) * 2 * CB + 4 * g) * 4);
      pin(foff[k]);
      gfw[k] = *(const f32x4*)(pro.gn_f + 4 * g); gfb[k] = *(const f32x4*)(pro.gn_f + CB + 4 * g);
      if (pro.mode == GRU_PRO_OUT) { gow[k] = *(const f32x4*)(pro.gn_o + 4 * g); gob[k] = *(const f32x4*)(pro.gn_o + CB + 4 * g); }
    }
  }
  // the GroupNorm statistics of sample b, finished from the partial sums by the whole workgroup (uniform call)
  auto pro_stats = [&](int b) {
    if constexpr (EPI == EPI_LINEAR) {                     // (the scratch exists in the LINEAR instantiations only)
    __shared__ double pro_wsum[4][2];
    __shared__ float pro_st[2][2];
    if (b == pst_b) return;
    pst_b = b;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      if (which == 1 && pro.mode != GRU_PRO_OUT) break;
      const int parts = which ? pro.parts_o : pro.parts_f;
      const double* p = which ? pro.part_o + (size_t)b * parts * 2 : pro.part_f + ((size_t)b * 2 + pro.group_f) * parts * 2;
      double s_ = 0.0, q_ = 0.0;
      for (int k = tid; k < parts; k += 256) { s_ += p[2 * k]; q_ += p[2 * k + 1]; }
      for (int o = 32; o > 0; o >>= 1) { s_ += __shfl_down(s_, o); q_ += __shfl_down(q_, o); }
      __syncthreads();                                     // (the previous use of the scratch is over)
      if (lane == 0) { pro_wsum[wave][0] = s_; pro_wsum[wave][1] = q_; }
      __syncthreads();
      if (tid == 0) {
        double s2 = 0.0, q2 = 0.0;
        for (int w = 0; w < 4; ++w) { s2 += pro_wsum[w][0]; q2 += pro_wsum[w][1]; }
        const double mean = s2 / pro.count, var = fmax(q2 / pro.count - mean * mean, 0.0);
        pro_st[which][0] = (float)mean;
        pro_st[which][1] = (float)(1.0 / sqrt(var + (double)pro.eps));
      }
      __syncthreads();
      pst[2 * which] = pro_st[which][0]; pst[2 * which + 1] = pro_st[which][1];
    }
    }
  };

  auto load_tile = [&](f32x4 (&stage)[NL], int b, int tx, int ty) {
    const int ix0 = tx * TC * STRIDE - 1, iy0 = ty * TR * STRIDE - 1;
    const long pix0 = ((long)b * a.hi + iy0) * a.wi + ix0;                 // may point one row/column outside
    const buf_rsrc ra = make_rsrc((const char*)a.srcA + pix0 * (CA * 4));
    const buf_rsrc rb = make_rsrc((const char*)a.srcB + pix0 * (CB * 4));
    if (EPI == EPI_LINEAR && pro_on) {                     // the operands of the on-the-fly srcB (zero padding is applied when it is formed)
      const buf_rsrc rf = make_rsrc((const char*)pro.f + pix0 * (2 * CB * 4));
      const buf_rsrc ro = make_rsrc((const char*)(pro.mode == GRU_PRO_OUT ? pro.o : a.srcB) + pix0 * (CB * 4));
#pragma unroll
      for (int k = 0; k < NBP; ++k) {
        const int iy = iy0 + (rc[NA + k] & 0xffff), ix = ix0 + (rc[NA + k] >> 16);
        const bool ok = (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi;
        stageF[k] = buf_load4(rf, ok ? foff[k] : BUF_OOB);
        if (pro.mode == GRU_PRO_OUT) stageO[k] = buf_load4(ro, ok ? goff[NA + k] : BUF_OOB);
      }
    }
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + LR <= a.hi && ix0 + LC <= a.wi;
    if (interior) {
#pragma unroll
      for (int k = 0; k < NL; ++k) stage[k] = buf_load4(k < NA ? ra : rb, goff[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int iy = iy0 + (rc[k] & 0xffff), ix = ix0 + (rc[k] >> 16);
        const bool ok = (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi;
        stage[k] = buf_load4(k < NA ? ra : rb, ok ? goff[k] : BUF_OOB);              // zero padding
      }
    }
  };
  auto store_tile = [&](const f32x4 (&stage)[NL], int b, int tx, int ty) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      float* dl = (float*)((char*)lds + lbyte[k]);
      f32x4 v = stage[k];
      if (EPI == EPI_LINEAR && k >= NA && pro_on) {
        const int kb = k - NA < NBP ? k - NA : 0;
        const int wr = rc[k] & 0xffff, wc = rc[k] >> 16;
        const int iy = ty * TR - 1 + wr, ix = tx * TC - 1 + wc;
        const bool ok = (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi;
        const f32x4 fn = (stageF[kb] - pst[0]) * pst[1] * gfw[kb] + gfb[kb];
        const f32x4 sg = {sigmoidf_(fn.x), sigmoidf_(fn.y), sigmoidf_(fn.z), sigmoidf_(fn.w)};
        if (pro.mode == GRU_PRO_GATES) {
          v = sg * v;                                        // r * h
        } else {
          const f32x4 on = (stageO[kb] - pst[2]) * pst[3] * gow[kb] + gob[kb];
          const f32x4 y = {tanhf(on.x), tanhf(on.y), tanhf(on.z), tanhf(on.w)};
          v = sg * v + (1.0f - sg) * y;                      // h' = u h + (1 - u) y
        }
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};              // zero padding
        if (pro.mode == GRU_PRO_OUT && ok && tid + kb * 256 < NPIX * GB && wr >= 1 && wr <= TR && wc >= 1 && wc <= TC) {
          const size_t pix = ((size_t)b * a.hi + iy) * a.wi + ix;      // the tile's own pixels: the new state (and R)
          const int c4 = 4 * ((tid + kb * 256) % GBP);
          *(f32x4*)(pro.state_out + pix * CB + c4) = v;
          if (pro.R) *(f32x4*)(pro.R + pix * pro.RW + c4) = v;
        }
      }
      dl[0] = v.x; dl[PLANE] = v.y; dl[2 * PLANE] = v.z; dl[3 * PLANE] = v.w;
    }
  };

  int t = tr.begin + wg;
  if (t >= tr.end) return;
  int b, tx, ty;
  tile_coords(tg, t, b, tx, ty);
  f32x4 stage[NL];
  load_tile(stage, b, tx, ty);
  wait_vmem_all();                     // fragments, biases and the first tile: nothing is pending inside the loop
  if (EPI == EPI_LINEAR && pro_on) pro_stats(b);
  store_tile(stage, b, tx, ty);
  __syncthreads();
  for (;;) {
    const int oy0 = ty * TR, ox0 = tx * TC;
    const long opix0 = ((long)b * a.ho + oy0) * a.wo + ox0;
    const bool full = oy0 + TR <= a.ho && ox0 + TC <= a.wo;            // uniform: no output of the tile is outside
    const buf_rsrc r0 = make_rsrc((char*)a.dst0 + opix0 * (CO * 4));
    const buf_rsrc r1 = make_rsrc((char*)a.dst1 + opix0 * (HC * 4));
    const buf_rsrc rin = make_rsrc((const char*)(a.hin ? a.hin : a.dst0) + opix0 * (CO * 4));     // CAND: state in
    unsigned oo[NT], oo1[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { oo[nt] = ooff[nt]; oo1[nt] = ooff1[nt]; }
    if (!full) {
      const bool valid = oy0 + row < a.ho && ox0 + col + p < a.wo;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { oo[nt] = valid ? ooff[nt] : BUF_OOB; oo1[nt] = valid ? ooff1[nt] : BUF_OOB; }
    }

    // requests: epilogue operands first, then the next tile
    f32x4 pre_u[NT], pre_h[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (EPI == EPI_CAND && nt * 16 < HC) {
        pre_u[nt] = buf_load4(r1, oo[nt]);
        pre_h[nt] = buf_load4(rin, oo[nt]);
      }
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
    conv3x3_run_at<NT, KC, STRIDE, LC>(acc, wf, lds, xbyte);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) drain(acc[nt]);

    f32x4 hc[NT];      // GATES: the hidden state of the lane's pixel, from the tile (module.py:35-41)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (EPI == EPI_GATES && nt * 16 < HC) {
        const float* hl = (const float*)((const char*)lds + hbyte[nt]);
        hc[nt] = f32x4{hl[0], hl[PLANE], hl[2 * PLANE], hl[3 * PLANE]};
      }
    }

    wait_vmem_all();                   // the wait point (explicit, so that no other wait is scheduled elsewhere)
    __syncthreads();                   // every wave is done reading the tile
    if (EPI == EPI_LINEAR && pro_on && more) pro_stats(bn);
    if (more) store_tile(stage, bn, txn, tyn);

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 v = acc[nt] + bias[nt];
      if (EPI == EPI_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        buf_store4(r0, oo[nt], v);
      } else if (EPI == EPI_LINEAR) {
        buf_store4(r0, oo[nt], v);
      } else if (EPI == EPI_GATES) {
        f32x4 sg = {sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)};
        if (nt * 16 < HC) buf_store4(r0, oo[nt], sg * hc[nt]);              // reset-gate rows -> r * h
        if (nt * 16 + 16 > HC) buf_store4(r1, oo1[nt], sg);                 // update-gate rows -> u
      } else if (nt * 16 < HC) {                        // EPI_CAND   (module.py:44-50)
        f32x4 cnd = {tanh_fast(v.x), tanh_fast(v.y), tanh_fast(v.z), tanh_fast(v.w)};
        f32x4 u4 = pre_u[nt], h4 = pre_h[nt];
        buf_store4(r0, oo[nt], u4 * h4 + (1.0f - u4) * cnd);
      }
    }
    if (EPI == EPI_LINEAR && a.gn_part) {          // uniform
      double gs[2] = {0.0, 0.0}, gq[2] = {0.0, 0.0};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co4 = nt * 16 + 4 * q, g = co4 >= a.gn_hc ? 1 : 0;
        if (oo[nt] != BUF_OOB && co4 < a.gn_groups * a.gn_hc) {
          const f32x4 v = acc[nt] + bias[nt];
          gs[g] += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
          gq[g] += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        }
      }
      const int parts = tg.tiles_x * tg.tiles_y * 4;
      for (int g = 0; g < a.gn_groups; ++g) {
        double s_ = gs[g], q_ = gq[g];
        for (int o = 32; o > 0; o >>= 1) { s_ += __shfl_down(s_, o); q_ += __shfl_down(q_, o); }      // fixed tree: deterministic
        if (lane == 0) {
          double* o = a.gn_part + (((size_t)b * a.gn_groups + g) * parts + (ty * tg.tiles_x + tx) * 4 + wave) * 2;
          o[0] = s_; o[1] = q_;
        }
      }
    }
    if (!more) break;
    __syncthreads();                   // next tile visible
    t = tn; b = bn; tx = txn; ty = tyn;
  }
  }
};

// ---------------------------------------------------------------------------
// conv2 (8 -> 16, stride 2, ReLU; reference adamvs.py:418) and the gate convolution of the level-2 ConvGRU
// (module.py:28-41 on cat(c2, h2)) in one tile loop.  The two are consecutive links of the per-step chain
// conv2 -> gates2 -> cand2; fused, the software pipeline of recurrence.hip needs two dependent launches per hypothesis
// instead of three, which is what a latency-bound stage (few tiles per CU) pays for.  conv2 is cheap (288 MAC per
// half-resolution pixel against 2304 for the gates), so its recomputation on the one-pixel halo of the 4 x 16 tile
// (6 x 18 window: 1.7x) costs little; c2 of the tile proper also goes to memory for the candidate convolution.
//   LDS: h1 window 13 x 37 full-resolution pixels (stride-2 layout of ConvSmallRole), then the 6 x 18 window of
//   cat(c2, h2) in the stride-1 layout the gate chain reads.
struct Conv2Gates2Args {
  const float* st1;     // [B][h*w][8]      level-1 state of the step
  const float* st2;     // [B][h2*w2][16]   level-2 state of the previous step
  const float* wconv2;  // A fragments [1][9][2][64]
  const float* wgates;  // A fragments [2][9][8][64]
  const float* bgates;  // [32]
  float* c2;            // [B][h2*w2][16]   ReLU(conv2(h1)), for the candidate convolution
  float* rh;            // [B][h2*w2][16]   r * h2
  float* u;             // [B][h2*w2][16]
  int h, w, h2, w2;
};

// HALF 0: the reset-gate rows (-> r * h2) and c2 to memory; HALF 1: the update-gate rows (-> u).  One role with both
// halves holds 162 registers of weights and leaves one wave per SIMD to every role of its launch; two half roles
// (72 + 18 registers of weights each, the window loaded and conv2 evaluated twice) keep the launch at two to three.
template <int HALF>
struct Conv2Gates2Role {
  typedef Conv2Gates2Args Args;
  static constexpr int TR = 4, TC = 16, LR = TR + 2, LC = TC + 2, NW = LR * LC;             // c2 / h2 window (half resolution)
  static constexpr int HR = 2 * LR + 1, HC = 2 * LC + 1, NH = HR * HC;                      // h1 window (full resolution)
  static constexpr int PLANE_A = NH | 1, GP_A = group_pitch(PLANE_A, 2);
  static constexpr int PLANE = plane_pitch16(NW), G = 8, GP = group_pitch(PLANE, G);
  static constexpr int LDS_A = 2 * GP_A;                                                    // floats
  static constexpr int NA = (NH * 2 + 255) / 256, NB = (NW * 4 + 255) / 256, NL = NA + NB;
  static constexpr int NRUN = 2;                                                            // conv2 runs per wave (7 runs of 16 window pixels)
  static constexpr size_t LDS_BYTES = (size_t)(LDS_A + G * GP) * sizeof(float);
  static constexpr int TILE_W = TC, TILE_H = TR;
  static int tiles_x(const Args& a) { return cdiv(a.w2, TC); }
  static int tiles_y(const Args& a) { return cdiv(a.h2, TR); }

  static __device__ __forceinline__ void run(const Args& a, const TileGrid& tg, TileRange tr, int wg, int nwg, float* lds) {
  // lds: the h1 window, then (at LDS_A floats) the cat(c2, h2) window
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;

  float wc[1][9][2];
  load_wfrag<1, 2>(wc, a.wconv2, lane);
  float wf[1][9][8];
  load_wfrag<1, 8>(wf, a.wgates + HALF * 9 * 8 * 64, lane);                 // fragment stream [nt][tap][kc][64]: nt = HALF
  const f32x4 bias = *(const f32x4*)(a.bgates + HALF * 16 + 4 * q);

  // ---- per-lane constants: the two windows
  unsigned goff[NL], lbyte[NL];
  int rc[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const bool isA = k < NA;
    if (isA) {
      const int j = min(tid + k * 256, NH * 2 - 1);
      const int g = j % 2, pp = j / 2, r = pp / HC, c = pp % HC;
      goff[k] = (unsigned)(((r * a.w + c) * 8 + 4 * g) * 4);
      lbyte[k] = (unsigned)((g * GP_A + r * HC + c) * 4);
      rc[k] = r | (c << 16);
    } else {
      const int j = min(tid + (k - NA) * 256, NW * 4 - 1);
      const int g = j % 4, pp = j / 4, r = pp / LC, c = pp % LC;
      goff[k] = (unsigned)(((r * a.w2 + c) * 16 + 4 * g) * 4);
      lbyte[k] = (unsigned)((LDS_A + (4 + g) * GP + r * LC + c) * 4);
      rc[k] = r | (c << 16);
    }
    pin(goff[k]); pin(lbyte[k]); pin(rc[k]);
  }
  // conv2: run j of the wave covers window pixels 16 (wave + 4 j) .. + 15 (runs past the 108 pixels repeat the last one)
  unsigned xa[NRUN][2], cbyte[NRUN], coff[NRUN];
  int wrc[NRUN];
#pragma unroll
  for (int j = 0; j < NRUN; ++j) {
    const int i = 16 * (wave + 4 * j) + p, ic = min(i, NW - 1);
    const int wr = ic / LC, wcol = ic % LC;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      xa[j][kc] = (unsigned)((kc * GP_A + q * PLANE_A + (2 * wr) * HC + 2 * wcol) * 4);
      pin(xa[j][kc]);
    }
    cbyte[j] = (unsigned)((LDS_A + q * GP + ic) * 4);                           // c2 channels 4q..4q+3 of the window pixel: planes 0..3 of group q
    const bool inner = i < NW && wr >= 1 && wr <= TR && wcol >= 1 && wcol <= TC;
    coff[j] = inner ? (unsigned)((((wr - 1) * a.w2 + wcol - 1) * 16 + 4 * q) * 4) : BUF_OOB;
    wrc[j] = wr | (wcol << 16) | (i < NW ? 0 : (1 << 30));
    pin(cbyte[j]); pin(coff[j]); pin(wrc[j]);
  }
  // gates: the wave's run is row `wave` of the tile
  unsigned xbyte[8];
#pragma unroll
  for (int kc = 0; kc < 8; ++kc) {
    xbyte[kc] = (unsigned)((LDS_A + kc * GP + q * PLANE + wave * LC + p) * 4);
    pin(xbyte[kc]);
  }
  const unsigned hbyte = (unsigned)((LDS_A + (4 + q) * GP + (wave + 1) * LC + p + 1) * 4);     // h2 channels 4q.. of the lane's pixel (q < 4)
  unsigned ooff = (unsigned)(((wave * a.w2 + p) * 16 + 4 * q) * 4);         // the lane's pixel, channels 4q.. of rh (HALF 0) or u (HALF 1)
  pin(ooff);

  auto load_tile = [&](f32x4 (&stage)[NL], int b, int tx, int ty) {
    const int j0 = tx * TC - 1, i0 = ty * TR - 1;                       // window origin, half resolution
    const int ix0 = 2 * j0 - 1, iy0 = 2 * i0 - 1;                       // h1 window origin
    const buf_rsrc ra = make_rsrc((const char*)a.st1 + (((long)b * a.h + iy0) * a.w + ix0) * 32);
    const buf_rsrc rb = make_rsrc((const char*)a.st2 + (((long)b * a.h2 + i0) * a.w2 + j0) * 64);
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + HR <= a.h && ix0 + HC <= a.w && i0 + LR <= a.h2 && j0 + LC <= a.w2;
    if (interior) {
#pragma unroll
      for (int k = 0; k < NL; ++k) stage[k] = buf_load4(k < NA ? ra : rb, goff[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const bool isA = k < NA;
        const int iy = (isA ? iy0 : i0) + (rc[k] & 0xffff), ix = (isA ? ix0 : j0) + (rc[k] >> 16);
        const bool ok = (unsigned)iy < (unsigned)(isA ? a.h : a.h2) && (unsigned)ix < (unsigned)(isA ? a.w : a.w2);
        stage[k] = buf_load4(isA ? ra : rb, ok ? goff[k] : BUF_OOB);              // zero padding
      }
    }
  };
  auto store_tile = [&](const f32x4 (&stage)[NL]) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      float* dl = (float*)((char*)lds + lbyte[k]);
      const int pl = k < NA ? PLANE_A : PLANE;
      f32x4 v = stage[k];
      dl[0] = v.x; dl[pl] = v.y; dl[2 * pl] = v.z; dl[3 * pl] = v.w;
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
    const long opix0 = ((long)b * a.h2 + oy0) * a.w2 + ox0;
    const bool full = oy0 + TR <= a.h2 && ox0 + TC <= a.w2;
    const buf_rsrc rc2 = make_rsrc((char*)a.c2 + opix0 * 64);
    const buf_rsrc r0 = make_rsrc((char*)a.rh + opix0 * 64);
    const buf_rsrc r1 = make_rsrc((char*)a.u + opix0 * 64);
    const int tn = t + nwg;
    const bool more = tn < tr.end;
    int bn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, bn, txn, tyn);
      load_tile(stage, bn, txn, tyn);
    }

    // ---- conv2 on the 6 x 18 window -> ReLU -> LDS (groups 0..3) and, for the tile proper, memory
    const bool win_in = oy0 >= 1 && ox0 >= 1 && oy0 + TR + 1 <= a.h2 && ox0 + TC + 1 <= a.w2;      // whole window inside the map
#pragma unroll
    for (int j = 0; j < NRUN; ++j) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int kc = 0; kc < 2; ++kc)
            acc = mfma16(wc[0][ky * 3 + kx][kc], *(const float*)((const char*)lds + xa[j][kc] + (ky * HC + kx) * 4), acc);
      drain(acc);
      const int wr = wrc[j] & 0xffff, wcol = (wrc[j] >> 16) & 0x3fff;
      bool inside = true;                                                     // c2 is zero outside the map (the gates' padding)
      if (!win_in) inside = (unsigned)(oy0 - 1 + wr) < (unsigned)a.h2 && (unsigned)(ox0 - 1 + wcol) < (unsigned)a.w2;
      f32x4 v = {fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f)};
      if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};
      float* dl = (float*)((char*)lds + cbyte[j]);
      dl[0] = v.x; dl[PLANE] = v.y; dl[2 * PLANE] = v.z; dl[3 * PLANE] = v.w;   // surplus runs rewrite pixel 107 with its own value
      unsigned co = coff[j];
      if (!full && !(oy0 + wr - 1 < a.h2 && ox0 + wcol - 1 < a.w2)) co = BUF_OOB;
      if (HALF == 0) buf_store4(rc2, co, v);
    }
    __syncthreads();                     // c2 window complete

    // ---- gates on cat(c2, h2)
    unsigned oo = ooff;
    if (!full && !(oy0 + wave < a.h2 && ox0 + p < a.w2)) oo = BUF_OOB;
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    conv3x3_run_at<1, 8, 1, LC>(acc, wf, lds, xbyte);
    drain(acc[0]);
    const float* hl = (const float*)((const char*)lds + hbyte);
    const f32x4 hc = {hl[0], hl[PLANE], hl[2 * PLANE], hl[3 * PLANE]};          // module.py:35-41

    wait_vmem_all();                     // the wait point of the tile
    __syncthreads();                     // every wave is done reading both windows
    if (more) store_tile(stage);

    {
      const f32x4 v = acc[0] + bias;
      const f32x4 sg = {sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)};
      if (HALF == 0) buf_store4(r0, oo, sg * hc);                              // reset-gate rows -> r * h
      else buf_store4(r1, oo, sg);                                             // update-gate rows -> u
    }
    if (!more) break;
    __syncthreads();                     // next tile visible
    t = tn; b = bn; tx = txn; ty = tyn;
  }
  }
};

// ---------------------------------------------------------------------------
// Decoder: s = ReLU(upconv1(h2) + b + h1)   (ConvTranspose2d 16->8, k3 s2 p1 op1)
//          reg = upconv2d(s) + b            (ConvTranspose2d 8->1 k3 s2 p1 op1 when IN_UP,
//                                            Conv2d 8->1 k3 p1 otherwise)
// One block = inner tile 14 x 30 of s (full resolution h x w), s region 16 x 32.
// grid: (ceil(w/30), ceil(h/14), B); block 256.  reg -> vol[b][d][Ho*Wo].
struct DecoderArgs {
  const float* h2;     // [B][(h/2)*(w/2)][16]
  const float* h1;     // [B][h*w][8]
  const float* wup1;   // A fragments [1][9][4][64]: A[cout][cin] of tap (ky,kx)
  const float* bup1;   // [16] zero padded
  const float* wfin;   // [72] index (ky*3 + kx)*8 + c, then bias at [72]
  float* vol;          // [B][D][Ho*Wo]
  int h, w, D, d;
};

// ConvTranspose2d(k3, s2, p1, op1) restricted to one output parity class (PY,PX):
// out[2i+PY][2j+PX] = sum over taps with ky = 2(i-iy)+PY+1, i.e. PY=0 -> (ky=1, iy=i);
// PY=1 -> (ky=2, iy=i), (ky=0, iy=i+1); same along x.  `xbyte` = LDS byte offset of the lane's
// B-fragment origin (k-row q, h2 pixel (li, lj)) in the planar h2 tile.
template <int PY, int PX, int HGP, int HCOLS>
__device__ __forceinline__ f32x4 upconv1_class(const float (&wf)[1][9][4], const float* lds, unsigned xbyte) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ty = 0; ty < (PY ? 2 : 1); ++ty) {
    const int ky = PY ? (ty ? 0 : 2) : 1;
#pragma unroll
    for (int tx = 0; tx < (PX ? 2 : 1); ++tx) {
      const int kx = PX ? (tx ? 0 : 2) : 1;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        acc = mfma16(wf[0][ky * 3 + kx][kc], *(const float*)((const char*)lds + xbyte + (kc * HGP + ty * HCOLS + tx) * 4), acc);
    }
  }
  return acc;
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Persistent, software-pipelined like k_conv_small (same reasoning: uniform descriptors + pinned lane offsets,
// no bounds checks on interior tiles, one wait per tile).  A tile is 8 x 30 pixels of s (full resolution h x w; round 4: 6 x 30
// until then -- phase 2, the larger half of the kernel's vector work, ran 180 of 256 lanes, now 240);
// its s region is 10 x 32 (one pixel of halo each side), fed by a 6 x 17 region of h2.
//   phase 1  upconv1 on the matrix cores: 20 runs = 4 parity classes x 5 rows; wave k takes row k of every
//            class and class k of the fifth row (equal MFMA counts: 9 taps in all for the four classes of its row, and one class of
//            the last row -- 1, 2, 2 or 4 taps).  + bias + h1 (requested during the previous tile) -> ReLU -> s in LDS.
//   phase 2  last layer on the vector units.  Transposed (IN_UP): one thread per s pixel produces its 2 x 2
//            output quad from s[i..i+1][j..j+1] (72 FMAs, weights as scalar operands), two 8-byte stores.
//            Flat: one thread per pixel, 3 x 3 x 8 FMAs.
// Round 5 (A/B: -DDEC_PAIR=0): PAIRED parity classes in phase 1.  upconv1 has 8 output channels: a parity class fills rows 0-7 of an
// MFMA tile.  The classes (py = 0, px) and (py = 1, px) of one h2 base row li read the same h2 pixels, so they share a tile -- rows
// 0-7 = s row 2 li - 1 (tap ky = 1), rows 8-15 = s row 2 li (ky = 2 from row li, ky = 0 from row li + 1): per base row and k-chunk
// 2 MFMAs for px = 0 and 4 for px = 1 instead of 3 and 6, every lane group carries outputs (half the epilogue per lane), and the
// paired fragments are formed per lane at kernel start from the fragments the kernel is packed with (rows 8-15 of a tap's fragment
// are zero: they take rows 0-7 of the partner's, one ds_bpermute each).  Six base rows cover the ten s rows (the outer halves of
// rows 0 and 5 fall outside the region and are dropped): 12 tile runs, three per wave -- (li = wave, px = 1), (li = wave, px = 0)
// and one of (4 | 5, px = 1 | 0) -- 40 / 32 MFMAs per wave instead of 40 - 52.
#ifndef DEC_PAIR
#define DEC_PAIR 1
#endif
template <bool IN_UP>
struct DecoderRole {
  typedef DecoderArgs Args;
  static constexpr int TRI = 8, TCI = 30;                                   // inner tile of s
  static constexpr int SR = TRI + 2, SC = 32, SPX = 12; // s region, channel-last, 8 channels + 4 floats of padding per
                                                  // pixel: 16-byte lane accesses at a 48-byte stride are conflict-free
  static constexpr int HR = TRI / 2 + 2, HCOLS = 17, HPLANE = plane_pitch16(HR * HCOLS);     // h2 region, 16 planes in 4 groups
  static constexpr int NRUN = DEC_PAIR ? 3 : 5;                             // runs per wave.  5: (class j, row wave) j = 0..3; (class wave, row 4)
  static_assert(TRI == 8 && TRI * TCI <= 256, "five s-row pairs, one thread per inner pixel in phase 2");
  static constexpr int HGP = group_pitch(HPLANE, 4);
  static constexpr int NH = (HR * HCOLS * 4 + 255) / 256;                   // h2 load instructions per tile
  static constexpr size_t LDS_BYTES = (size_t)(4 * HGP + SR * SC * SPX) * sizeof(float);
  static constexpr int TILE_W = TCI, TILE_H = TRI;
  static int tiles_x(const Args& a) { return cdiv(a.w, TCI); }
  static int tiles_y(const Args& a) { return cdiv(a.h, TRI); }

  static __device__ __forceinline__ void run(const Args& a, const TileGrid& tg, TileRange tr, int wg, int nwg, float* lds) {
  float* lh2 = lds;
  float* ls = lds + 4 * HGP;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
  const int p = lane & 15, q = lane >> 4;
  const int h = a.h, w = a.w, h2 = h >> 1, w2 = w >> 1;
  float wf[1][9][4];
  load_wfrag<1, 4>(wf, a.wup1, lane);
  const f32x4 bup = *(const f32x4*)(a.bup1 + 4 * (DEC_PAIR ? (q & 1) : q));
  // paired fragments [input offset][kc]: P1 = px 1 classes (01 | 11) at (0,0), (0,1), (1,0), (1,1); P0 = px 0 classes (00 | 10) at (0,0), (1,0)
  float P1[4][4], P0[2][4];
  if (DEC_PAIR) {
    const bool upper = (lane & 15) >= 8;
    const int from = (lane - 8) << 2;                                       // ds_bpermute address of lane - 8: same k-row, output row - 8
    auto pair = [&](float lo, float hi) {
      const float hv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, __builtin_bit_cast(int, hi)));
      return upper ? hv : lo;
    };
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      P1[0][kc] = pair(wf[0][1 * 3 + 2][kc], wf[0][2 * 3 + 2][kc]);
      P1[1][kc] = pair(wf[0][1 * 3 + 0][kc], wf[0][2 * 3 + 0][kc]);
      P1[2][kc] = pair(0.f, wf[0][0 * 3 + 2][kc]);
      P1[3][kc] = pair(0.f, wf[0][0 * 3 + 0][kc]);
      P0[0][kc] = pair(wf[0][1 * 3 + 1][kc], wf[0][2 * 3 + 1][kc]);
      P0[1][kc] = pair(0.f, wf[0][0 * 3 + 1][kc]);
    }
  }

  // ---- per-lane constants
  unsigned h2off[NH], h2lds[NH];
  int h2rc[NH];
#pragma unroll
  for (int k = 0; k < NH; ++k) {
    const int j = min(tid + k * 256, HR * HCOLS * 4 - 1);
    const int g = j & 3, pp = j >> 2, r = pp / HCOLS, c = pp % HCOLS;
    h2off[k] = (unsigned)(((r * w2 + c) * 16 + 4 * g) * 4);
    h2lds[k] = (unsigned)((g * HGP + r * HCOLS + c) * 4);
    h2rc[k] = r | (c << 16);
    pin(h2off[k]); pin(h2lds[k]); pin(h2rc[k]);
  }
  // run j < 4 of this wave = parity class j (py = j >> 1, px = j & 1), row k = wave of that class; run 4 = class `wave`, row 4
  unsigned xbyte[NRUN], xbyte1[NRUN], h1off[NRUN], sbyte[NRUN];
  int src[NRUN];       // s-region row | column << 16 of the lane's pixel
  unsigned rmask = 0;  // DEC_PAIR: bit j = the lane's s row of run j lies inside the region
#pragma unroll
  for (int j = 0; j < NRUN; ++j) {
    if (DEC_PAIR) {
      const int li = j < 2 ? wave : 4 + (wave & 1), px = j == 0 ? 1 : (j == 1 ? 0 : (wave < 2 ? 1 : 0));
      const int lj = px ? p : p + 1, r = 2 * li - 1 + (q >> 1), c = px ? 2 * p : 2 * p + 1;
      const bool rv = (unsigned)r < (unsigned)SR;
      xbyte[j] = (unsigned)((q * HPLANE + li * HCOLS + lj) * 4);
      xbyte1[j] = xbyte[j] + (li + 1 < HR ? HCOLS * 4 : 0);                  // the taps of h2 row li + 1 (base row 5: dropped half, row 5 again)
      h1off[j] = rv ? (unsigned)(((r * w + c) * 8 + 4 * (q & 1)) * 4) : BUF_OOB;
      sbyte[j] = (unsigned)((((rv ? r : 0) * SC + c) * SPX + 4 * (q & 1)) * 4);
      src[j] = (rv ? r : 0) | (c << 16);
      rmask |= rv ? (1u << j) : 0u;
      pin(xbyte1[j]);
    } else {
    const int cls = j < 4 ? j : wave, py = cls >> 1, px = cls & 1, k = j < 4 ? wave : 4;
    const int r = py ? 2 * k : 2 * k + 1, c = px ? 2 * p : 2 * p + 1;
    const int li = py ? k : k + 1, lj = px ? p : p + 1;
    xbyte[j] = (unsigned)((q * HPLANE + li * HCOLS + lj) * 4);
    h1off[j] = q < 2 ? (unsigned)(((r * w + c) * 8 + 4 * q) * 4) : BUF_OOB;       // rows 8-15 of the tile are padding
    sbyte[j] = (unsigned)(((r * SC + c) * SPX + 4 * (q & 1)) * 4);
    src[j] = r | (c << 16);
    }
    pin(xbyte[j]); pin(h1off[j]); pin(sbyte[j]); pin(src[j]);
  }
  pin(rmask);
  // phase 2: thread -> inner pixel (i, j)
  const bool worker = tid < TRI * TCI;
  const int pi = min(tid, TRI * TCI - 1) / TCI, pj = min(tid, TRI * TCI - 1) % TCI;
  unsigned qbyte = (unsigned)(((IN_UP ? (pi + 1) * SC + pj + 1 : pi * SC + pj)) * SPX * 4);   // first s pixel the thread reads
  const int Ho = IN_UP ? 2 * h : h, Wo = IN_UP ? 2 * w : w;
  unsigned ooff = worker ? (unsigned)((IN_UP ? (2 * pi * Wo + 2 * pj) : (pi * Wo + pj)) * 4) : BUF_OOB;
  pin(qbyte); pin(ooff);
  cfloat* wfin0 = as_const(a.wfin);

  auto load_h2 = [&](f32x4 (&stage)[NH], int b, int tx, int ty) {
    const int i0 = ty * (TRI / 2) - 1, j0 = tx * (TCI / 2) - 1;
    const buf_rsrc rh = make_rsrc((const char*)a.h2 + (((long)b * h2 + i0) * w2 + j0) * 64);
    if (i0 >= 0 && j0 >= 0 && i0 + HR <= h2 && j0 + HCOLS <= w2) {
#pragma unroll
      for (int k = 0; k < NH; ++k) stage[k] = buf_load4(rh, h2off[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const int iy = i0 + (h2rc[k] & 0xffff), ix = j0 + (h2rc[k] >> 16);
        stage[k] = buf_load4(rh, ((unsigned)iy < (unsigned)h2 && (unsigned)ix < (unsigned)w2) ? h2off[k] : BUF_OOB);
      }
    }
  };
  // skip operand h1 of the lane's four pixels; returns which of them lie inside the image (bit j; 15 for every
  // lane of an interior tile, which the caller tests with a uniform branch)
  constexpr unsigned ALL_IN = (1u << NRUN) - 1u;
  auto load_h1 = [&](f32x4 (&hreg)[NRUN], int b, int tx, int ty) -> unsigned {
    const int ys0 = ty * TRI - 1, xs0 = tx * TCI - 1;
    const buf_rsrc r1 = make_rsrc((const char*)a.h1 + (((long)b * h + ys0) * w + xs0) * 32);
    if (ys0 >= 0 && xs0 >= 0 && ys0 + SR <= h && xs0 + SC <= w) {
#pragma unroll
      for (int j = 0; j < NRUN; ++j) hreg[j] = buf_load4(r1, h1off[j]);
      return ALL_IN;
    }
    unsigned in = 0;
#pragma unroll
    for (int j = 0; j < NRUN; ++j) {
      const int ys = ys0 + (src[j] & 0xffff), xs = xs0 + (src[j] >> 16);
      const bool ok = ((unsigned)ys < (unsigned)h && (unsigned)xs < (unsigned)w) || (DEC_PAIR && !((rmask >> j) & 1u));     // (a dropped half: nothing to zero)
      hreg[j] = buf_load4(r1, ok ? h1off[j] : BUF_OOB);
      in |= ok ? (1u << j) : 0u;
    }
    return in;
  };
  auto store_h2 = [&](const f32x4 (&stage)[NH]) {
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      float* dl = (float*)((char*)lh2 + h2lds[k]);
      f32x4 v = stage[k];
      dl[0] = v.x; dl[HPLANE] = v.y; dl[2 * HPLANE] = v.z; dl[3 * HPLANE] = v.w;
    }
  };

  int t = tr.begin + wg;
  if (t >= tr.end) return;
  int b, tx, ty;
  tile_coords(tg, t, b, tx, ty);
  f32x4 stage[NH], hreg[NRUN];
  load_h2(stage, b, tx, ty);
  unsigned inside = load_h1(hreg, b, tx, ty);
  wait_vmem_all();
  store_h2(stage);
  __syncthreads();
  for (;;) {
    const int tn = t + nwg;
    const bool more = tn < tr.end;
    int bn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, bn, txn, tyn);
      load_h2(stage, bn, txn, tyn);                 // in flight during phases 1 and 2
    }

    // ---- phase 1
    f32x4 sv[NRUN];
#if DEC_PAIR
    {
      auto chain1 = [&](unsigned x0, unsigned x1) {      // px = 1 tile: h2 pixels (li, lj), (li, lj + 1), (li + 1, lj), (li + 1, lj + 1)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          const char* b0 = (const char*)lh2 + x0 + kc * HGP * 4;
          const char* b1 = (const char*)lh2 + x1 + kc * HGP * 4;
          acc = mfma16(P1[0][kc], *(const float*)b0, acc);
          acc = mfma16(P1[1][kc], *(const float*)(b0 + 4), acc);
          acc = mfma16(P1[2][kc], *(const float*)b1, acc);
          acc = mfma16(P1[3][kc], *(const float*)(b1 + 4), acc);
        }
        return acc;
      };
      auto chain0 = [&](unsigned x0, unsigned x1) {      // px = 0 tile: (li, lj), (li + 1, lj)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          acc = mfma16(P0[0][kc], *(const float*)((const char*)lh2 + x0 + kc * HGP * 4), acc);
          acc = mfma16(P0[1][kc], *(const float*)((const char*)lh2 + x1 + kc * HGP * 4), acc);
        }
        return acc;
      };
      sv[0] = chain1(xbyte[0], xbyte1[0]);
      sv[1] = chain0(xbyte[1], xbyte1[1]);
      if (wave < 2) sv[2] = chain1(xbyte[2], xbyte1[2]);                  // uniform
      else sv[2] = chain0(xbyte[2], xbyte1[2]);
    }
#else
    sv[0] = upconv1_class<0, 0, HGP, HCOLS>(wf, lh2, xbyte[0]);
    sv[1] = upconv1_class<0, 1, HGP, HCOLS>(wf, lh2, xbyte[1]);
    sv[2] = upconv1_class<1, 0, HGP, HCOLS>(wf, lh2, xbyte[2]);
    sv[3] = upconv1_class<1, 1, HGP, HCOLS>(wf, lh2, xbyte[3]);
    switch (wave) {                                                       // uniform: the wave's class of the fifth row
      case 0: sv[4] = upconv1_class<0, 0, HGP, HCOLS>(wf, lh2, xbyte[4]); break;
      case 1: sv[4] = upconv1_class<0, 1, HGP, HCOLS>(wf, lh2, xbyte[4]); break;
      case 2: sv[4] = upconv1_class<1, 0, HGP, HCOLS>(wf, lh2, xbyte[4]); break;
      default: sv[4] = upconv1_class<1, 1, HGP, HCOLS>(wf, lh2, xbyte[4]); break;
    }
#endif
#pragma unroll
    for (int j = 0; j < NRUN; ++j) drain(sv[j]);
    const bool all_in = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(inside != ALL_IN) == 0) != 0;
    if (DEC_PAIR || q < 2) {
#pragma unroll
      for (int j = 0; j < NRUN; ++j) {
        f32x4 v = sv[j] + bup + hreg[j];                                  // adamvs.py:420-421
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        sv[j] = v;
      }
      if (!all_in) {                                                      // s is zero outside the image
#pragma unroll
        for (int j = 0; j < NRUN; ++j)
          if (!((inside >> j) & 1u)) sv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < NRUN; ++j)
        if (!DEC_PAIR || ((rmask >> j) & 1u)) *(f32x4*)((char*)ls + sbyte[j]) = sv[j];
    }
    unsigned inside_n = ALL_IN;
    if (more) inside_n = load_h1(hreg, bn, txn, tyn);                     // hreg is free again
    __syncthreads();                                                      // s complete

    // ---- phase 2
    const int y0 = ty * TRI, x0 = tx * TCI;
    const bool full = y0 + TRI <= h && x0 + TCI <= w;
    unsigned oo = ooff;
    if (!full) oo = (y0 + pi < h && x0 + pj < w) ? ooff : BUF_OOB;
    const char* sp = (const char*)ls + qbyte;
    // 73 scalar weights do not fit next to everything else that is uniform here; hidden from loop-invariant
    // motion, they are fetched per tile through the scalar cache (SMEM) instead of being spilled to VGPR
    // lanes and read back with one v_readlane each.
    cfloat* wfin = wfin0;
    asm volatile("" : "+s"(wfin));
    // s pixel (dy, dx) from the thread's first pixel: 8 channels as four packed pairs
    auto spix = [&](int dy, int dx, f32x2 (&v)[4]) {
      const f32x4 lo = *(const f32x4*)(sp + (dy * SC + dx) * SPX * 4), hi = *(const f32x4*)(sp + (dy * SC + dx) * SPX * 4 + 16);
      v[0] = f32x2{lo.x, lo.y}; v[1] = f32x2{lo.z, lo.w}; v[2] = f32x2{hi.x, hi.y}; v[3] = f32x2{hi.z, hi.w};
    };
    // acc += w[tap][0..7] . v  (packed FMAs, the weight pair is a scalar operand)
    auto tap = [&](f32x2& acc, int t9, const f32x2 (&v)[4]) {
#pragma unroll
      for (int c2 = 0; c2 < 4; ++c2) acc += *(const f32x2 __attribute__((address_space(4)))*)(wfin + t9 * 8 + 2 * c2) * v[c2];
    };
    const float bf = wfin[72];
    if (IN_UP) {
      const buf_rsrc ro = make_rsrc((char*)a.vol + ((((long)b * a.D + a.d) * Ho + 2 * y0) * (long)Wo + 2 * x0) * 4);
      // quad of s pixel (i, j): out[2i][2j] = w11 s00;  out[2i][2j+1] = w12 s00 + w10 s01;
      // out[2i+1][2j] = w21 s00 + w01 s10;  out[2i+1][2j+1] = w22 s00 + w20 s01 + w02 s10 + w00 s11
      f32x2 s00[4], s01[4], s10[4], s11[4];
      spix(0, 0, s00); spix(0, 1, s01); spix(1, 0, s10); spix(1, 1, s11);
      f32x2 o00 = {bf, 0.f}, o01 = {bf, 0.f}, o10 = {bf, 0.f}, o11 = {bf, 0.f};
      tap(o00, 4, s00);
      tap(o01, 5, s00); tap(o01, 3, s01);
      tap(o10, 7, s00); tap(o10, 1, s10);
      tap(o11, 8, s00); tap(o11, 6, s01); tap(o11, 2, s10); tap(o11, 0, s11);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(o00.x + o00.y), __float_as_uint(o01.x + o01.y)}, ro, oo, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(o10.x + o10.y), __float_as_uint(o11.x + o11.y)}, ro,
                                            oo == BUF_OOB ? BUF_OOB : oo + (unsigned)Wo * 4u, 0, 0);
    } else {
      const buf_rsrc ro = make_rsrc((char*)a.vol + ((((long)b * a.D + a.d) * Ho + y0) * (long)Wo + x0) * 4);
      f32x2 o = {bf, 0.f};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          f32x2 v[4];
          spix(ky, kx, v);
          tap(o, ky * 3 + kx, v);
        }
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o.x + o.y), ro, oo, 0, 0);
    }
    if (!more) break;
    wait_vmem_all();                                                      // next tile's h2 and h1 have arrived
    __syncthreads();                                                      // every wave is done with s and h2
    store_h2(stage);
    __syncthreads();
    t = tn; b = bn; tx = txn; ty = tyn; inside = inside_n;
  }
  }
};
// Candidate convolution of the level-1 ConvGRU (reference models/module.py:44-50: tanh(conv(cat(x, r*h))), blend with
// u) in the same two-row form: its 8 output channels fill half an MFMA tile in the one-row form of k_conv_small (36
// MFMAs per 16-pixel run, half of the rows zero); with rows 8-15 = the same channels of the next output row it is
// 24 per run.  Two sources like k_conv_small (x = c1, r*h: 2 + 2 channel groups), its fused epilogue (u and h of the
// lane's own pixel requested before the chain; h updated in place), the tile and pipeline of k_conv1_two_row.
// EPI2 = TR_CAND: that epilogue.  EPI2 = TR_BIAS_RELU: out = ReLU(conv + bias) -- the same two-source, two-row convolution as a
// plain layer (FeatureNet0's deconv2.conv: 3x3 on cat(deconv output, skip), 16 -> 8 channels + folded BatchNorm + ReLU).
enum { TR_CAND = 0, TR_BIAS_RELU = 1 };
template <int EPI2>
struct TwoRowPairRole {
  typedef SmallConvArgs Args;
  static constexpr int CA = 8, CB = 8, C = 16, KC = C / 4, G = C / 4, GA = CA / 4, TR = 8, TC = 16, LR = TR + 2, LC = TC + 2;
  static constexpr int NPIX = LR * LC, PLANE = plane_pitch16(NPIX), GP = group_pitch(PLANE, G);
  static constexpr int NA = (NPIX * GA + 255) / 256, NL = 2 * NA;     // loads per source: a load's base must be uniform
  static constexpr size_t LDS_BYTES = (size_t)G * GP * sizeof(float);         // tile [G][GP]
  static constexpr int TILE_W = TC, TILE_H = TR;
  static int tiles_x(const Args& a) { return cdiv(a.wo, TC); }
  static int tiles_y(const Args& a) { return cdiv(a.ho, TR); }

  static __device__ __forceinline__ void run(const Args& a, const TileGrid& tg, TileRange tr, int wg, int nwg, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int h = a.hi, w = a.wi;
  float wf[12][KC];
#pragma unroll
  for (int t = 0; t < 12; ++t)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) wf[t][kc] = a.wpk[(t * KC + kc) * 64 + lane];
  const f32x4 bias = *(const f32x4*)(a.bias + 4 * (q & 1));

  unsigned goff[NL], lbyte[NL];
  int rc[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const bool isA = k < NA;
    const int j = min(tid + (isA ? k : k - NA) * 256, NPIX * GA - 1);          // surplus lanes repeat the last item
    const int g = j % GA, pp = j / GA, r = pp / LC, c = pp % LC;
    goff[k] = (unsigned)(((r * w + c) * CA + 4 * g) * 4);
    lbyte[k] = (unsigned)((((isA ? 0 : GA) + g) * GP + r * LC + c) * 4);
    rc[k] = r | (c << 16);
    pin(goff[k]); pin(lbyte[k]); pin(rc[k]);
  }
  unsigned xbyte[KC];
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
    const long pix0 = ((long)n * h + iy0) * w + ix0;
    const buf_rsrc ra = make_rsrc((const char*)a.srcA + pix0 * (CA * 4));
    const buf_rsrc rb = make_rsrc((const char*)a.srcB + pix0 * (CB * 4));
    if (iy0 >= 0 && ix0 >= 0 && iy0 + LR <= h && ix0 + LC <= w) {
#pragma unroll
      for (int k = 0; k < NL; ++k) stage[k] = buf_load4(k < NA ? ra : rb, goff[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int iy = iy0 + (rc[k] & 0xffff), ix = ix0 + (rc[k] >> 16);
        stage[k] = buf_load4(k < NA ? ra : rb, ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) ? goff[k] : BUF_OOB);
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

  int t = tr.begin + wg;
  if (t >= tr.end) return;
  int n, tx, ty;
  tile_coords(tg, t, n, tx, ty);
  f32x4 stage[NL];
  load_tile(stage, n, tx, ty);
  wait_vmem_all();
  store_tile(stage);
  __syncthreads();
  for (;;) {
    const int y0 = ty * TR, x0 = tx * TC;
    const long opix0 = ((long)n * h + y0) * w + x0;
    const buf_rsrc rh = make_rsrc((char*)a.dst0 + opix0 * 32);          // new h (the same buffer as the old one when hin is null)
    const buf_rsrc rin = make_rsrc((const char*)(a.hin ? a.hin : a.dst0) + opix0 * 32);     // h of the previous step
    const buf_rsrc ru = make_rsrc((const char*)a.dst1 + opix0 * 32);    // u
    unsigned oo = ooff;
    if (!(y0 + TR <= h && x0 + TC <= w)) oo = (y0 + orow < h && x0 + p < w) ? ooff : BUF_OOB;
    f32x4 pre_u = {0.f, 0.f, 0.f, 0.f}, pre_h = pre_u;
    if (EPI2 == TR_CAND) { pre_u = buf_load4(ru, oo); pre_h = buf_load4(rin, oo); }      // epilogue operands first, then the next tile
    const int tn = t + nwg;
    const bool more = tn < tr.end;
    int nn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, nn, txn, tyn);
      load_tile(stage, nn, txn, tyn);
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

    const f32x4 v = acc + bias;
    if (EPI2 == TR_CAND) {
      const f32x4 cnd = {tanh_fast(v.x), tanh_fast(v.y), tanh_fast(v.z), tanh_fast(v.w)};
      buf_store4(rh, oo, pre_u * pre_h + (1.0f - pre_u) * cnd);
    } else {
      buf_store4(rh, oo, f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)});
    }
    if (!more) break;
    __syncthreads();                                // next tile visible
    t = tn; n = nn; tx = txn; ty = tyn;
  }
  }
};
typedef TwoRowPairRole<TR_CAND> Cand1TwoRowRole;

}  // namespace adamvs
