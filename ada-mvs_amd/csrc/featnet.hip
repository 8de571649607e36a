// FeatureNet0 (reference models/adamvs.py:49-152; blocks models/module.py:164-251, 506-524): the 2D U-Net that
// turns every view into three feature maps (32 / 16 / 8 channels at 1/4, 1/2, 1/1 resolution).  SURVEY.md §8 row f1:
// immediately upstream of the hot path, and — once the hot path runs at 2.4 ms per tile — its largest neighbour
// (1.5 ms per tile on the vendor library, which also needs NCHW maps and a transpose for every stage output).
//
// Everything is channel-last fp32, [N = B*V images][h*w][C], the layout the plane sweep reads.  The eval-mode
// BatchNorm of every block is folded into the A fragments (scale) and a bias (shift) on the host (packing.py).
//   k_fconv    one persistent MFMA kernel for every convolution: 3x3 stride 1, 5x5 stride 2, ConvTranspose2d(k3, s2,
//              p1, op1) (four parity-class accumulators per tile), and the 1x1 output convolutions; up to two
//              concatenated sources; epilogue = bias, ReLU, or "+ two bilinearly upsampled context maps".
//   k_context  the two pooled-context branches of a stage (AvgPool 4 and 8, one read of the feature map):
//              AvgPool(P) -> 1x1 conv -> BN -> ReLU, then multiplied by the branch's
//              slice of the 1x1 output convolution.  (out_k = W . cat(up(b1), up(b2), f); upsampling and the 1x1
//              convolution are both linear, so W_1 is applied to b1 at pooled resolution and the full-resolution
//              kernel only adds up(W_1 b1) + up(W_2 b2) to W_f f: the concatenated 2C-channel map never exists.)
// The kernel structure is that of k_conv_small in slice_red.hip (see the comments there): persistent grid at
// resident capacity, one 16-pixel run per wave, uniform buffer descriptors + pinned lane offsets, one wait per tile.
#include "adamvs_hip.h"
#include "common.h"
#include "conv_frag.h"
#include "kernels.h"
#include "persistent.h"
#include "slice_roles.h"

namespace adamvs {

enum { FM_K3 = 0, FM_K5S2 = 1, FM_K1 = 2, FM_TALL = 3 };      // FM_TALL: ConvTranspose2d(k3, s2, p1, op1), all four parity classes
enum { FE_RELU = 1, FE_CONTEXT = 2, FE_ADD_UP = 4 };       // FE_ADD_UP: + nearest-neighbour 2x upsampling of ctxA [N][(ho/2)(wo/2)][ctot] (FPN top-down path)

struct FConvArgs {
  const float* srcA;     // [N][hi*wi][CA]
  const float* srcB;     // [N][hi*wi][CB] (concatenated after A; null when CB == 0)
  const float* wpk;      // A fragments [NT][NTAPS][(CA+CB)/4][64], BatchNorm scale folded in
  const float* bias;     // [16*NT] (BatchNorm shift; zeros if none)
  float* out;            // [N][Ho*Wo][ctot], this launch writes channels co0 .. co0+cout-1
  const float* ctxA;     // FE_CONTEXT: [N][hA*wA][ctot], added after bilinear upsampling (align_corners=False)
  const float* ctxB;     //             [N][hB*wB][ctot]
  int hi, wi;            // input size
  int ho, wo;            // tile space: output size (input size for the transposed classes, whose output is 2ho x 2wo)
  int cout, ctot, co0;
  int hA, wA, hB, wB;
  float syA, sxA, syB, sxB;   // FE_CONTEXT: input/output size ratios of the two context maps (filled by the launcher)
};

// Transposed convolution: output pixel (2i+py, 2j+px) sums input pixels (i+ty, j+tx), ty <= py, tx <= px, with kernel
// index (py ? (ty ? 0 : 2) : 1, same in x).  One tile = 4 x 16 input pixels = 8 x 32 output pixels; the 2 x 2 input
// neighbourhood is read once and feeds four accumulators (1 + 2 + 2 + 4 = 9 fragments per k-chunk, stored class by class).
template <int MODE> struct FGeom {
  static constexpr bool T = MODE == FM_TALL;
  static constexpr int STR = MODE == FM_K5S2 ? 2 : 1;
  static constexpr int KH = MODE == FM_K3 ? 3 : (MODE == FM_K5S2 ? 5 : (MODE == FM_K1 ? 1 : 2));
  static constexpr int KW = KH;
  static constexpr int ORG = MODE == FM_K3 ? -1 : (MODE == FM_K5S2 ? -2 : 0);      // window origin = tile origin * STR + ORG
  static constexpr int NTAPS = T ? 9 : KH * KW;
  static constexpr int rows(int tr) { return (tr - 1) * STR + KH; }
  static constexpr int cols(int tc) { return (tc - 1) * STR + KW; }
};

// ATen area_pixel_compute_source_index, align_corners=False
__device__ __forceinline__ void up_taps(int dst, int n_in, float scale, int& i0, int& i1, float& l1) {
  float s = ((float)dst + 0.5f) * scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__device__ __forceinline__ f32x4 up_sample4(const float* map, int n, int hm, int wm, int ctot, int c, int y, int x, float sy, float sx) {
  int y0, y1, x0, x1; float ly, lx;
  up_taps(y, hm, sy, y0, y1, ly);
  up_taps(x, wm, sx, x0, x1, lx);
  const float* p = map + (size_t)n * hm * wm * ctot + c;
  const f32x4 a = *(const f32x4*)(p + ((size_t)y0 * wm + x0) * ctot), b = *(const f32x4*)(p + ((size_t)y0 * wm + x1) * ctot);
  const f32x4 cc = *(const f32x4*)(p + ((size_t)y1 * wm + x0) * ctot), d = *(const f32x4*)(p + ((size_t)y1 * wm + x1) * ctot);
  const f32x4 top = a * (1.f - lx) + b * lx, bot = cc * (1.f - lx) + d * lx;
  return top * (1.f - ly) + bot * ly;
}

// PAIR8 (transposed layers with at most 8 output channels, round 5): a parity class fills rows 0-7 of an MFMA tile, so two classes
// that read the same input pixel share one -- tile 1 = (class 00 | class 01), tile 2 = (10 | 11): input offset (0,0) feeds both tiles,
// (0,1) the second halves of both, (1,0) and (1,1) tile 2 -- 6 MFMAs per k-chunk instead of 9, and every lane group carries outputs
// (lanes q < 2 the first class of a tile, q >= 2 the second).  The paired fragments are formed per lane at kernel start from the
// class-by-class fragments k_fconv is packed with: rows 8-15 take rows 0-7 of the partner's fragment (one ds_bpermute each).
template <int CA, int CB, int NT, int MODE, int EPI, bool PAIR8 = false>
__global__ __launch_bounds__(256) void k_fconv(FConvArgs a, TileGrid tg) {
  using GM = FGeom<MODE>;
  static_assert(!PAIR8 || GM::T, "PAIR8: the transposed mode");
  constexpr int CIN = CA + CB, KC = CIN / 4, G = CIN / 4, GA = CA / 4, GB = CB / 4;
  constexpr int TR = 4, TC = 16, STR = GM::STR, NTAPS = GM::NTAPS;
  constexpr int LR = GM::rows(TR), LC = GM::cols(TC), NPIX = LR * LC;
  constexpr int PLANE = (STR == 1) ? plane_pitch16(NPIX) : (NPIX | 1);
  constexpr int GP = group_pitch(PLANE, G);
  constexpr int NA = (NPIX * GA + 255) / 256, NB = (NPIX * GB + 255) / 256, NL = NA + NB;
  extern __shared__ float lds[];         // [G][GP]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int row = wave;                  // the wave's run: row `wave` of the 4 x 16 tile

  float wf[NT][NTAPS][KC];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) wf[nt][t][kc] = a.wpk[((nt * NTAPS + t) * KC + kc) * 64 + lane];
  f32x4 bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bias[nt] = *(const f32x4*)(a.bias + nt * 16 + 4 * (PAIR8 ? (q & 1) : q));
  // PAIR8: pw[tile][input offset ty * 2 + tx][kc]; class taps in wf: 00: 0 | 01: 1 (0,0), 2 (0,1) | 10: 3 (0,0), 4 (1,0) |
  // 11: 5 (0,0), 6 (0,1), 7 (1,0), 8 (1,1)   (index = cbase[cls] + ty * (1 + px) + tx, as in the class-by-class chain below)
  float pw[2][4][KC];
  if constexpr (PAIR8) {
    const bool upper = (lane & 15) >= 8;
    const int from = (lane - 8) << 2;        // ds_bpermute byte address of lane - 8 (same k-row, output row - 8)
    auto pair = [&](float lo, float hi) {   // rows 0-7 from `lo` (its rows 8-15 are zero), rows 8-15 <- rows 0-7 of `hi`
      const float h = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, __builtin_bit_cast(int, hi)));
      return upper ? h : lo;
    };
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      pw[0][0][kc] = pair(wf[0][0][kc], wf[0][1][kc]);      // (0,0): class 00 | class 01
      pw[0][1][kc] = pair(0.f, wf[0][2][kc]);               // (0,1):          | class 01
      pw[0][2][kc] = 0.f; pw[0][3][kc] = 0.f;
      pw[1][0][kc] = pair(wf[0][3][kc], wf[0][5][kc]);      // (0,0): class 10 | class 11
      pw[1][1][kc] = pair(0.f, wf[0][6][kc]);               // (0,1):          | class 11
      pw[1][2][kc] = pair(wf[0][4][kc], wf[0][7][kc]);      // (1,0): class 10 | class 11
      pw[1][3][kc] = pair(0.f, wf[0][8][kc]);               // (1,1):          | class 11
    }
  }

  // ---- per-lane constants
  unsigned goff[NL], lbyte[NL];
  int rc[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const bool isA = k < NA;
    const int gs = isA ? GA : GB, cs = isA ? CA : CB;
    int j = tid + (isA ? k : k - NA) * 256;
    j = min(j, NPIX * gs - 1);           // surplus lanes repeat the last item
    const int g = j % gs, pp = j / gs, r = pp / LC, c = pp % LC;
    goff[k] = (unsigned)(((r * a.wi + c) * cs + 4 * g) * 4);
    lbyte[k] = (unsigned)((((isA ? 0 : GA) + g) * GP + r * LC + c) * 4);
    rc[k] = r | (c << 16);
    pin(goff[k]); pin(lbyte[k]); pin(rc[k]);
  }
  unsigned xbyte[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    xbyte[kc] = (unsigned)((kc * GP + q * PLANE + (row * STR) * LC + p * STR) * 4);
    pin(xbyte[kc]);
  }
  // output: lane's pixel (row, p) of the tile, channels co0 + 16 nt + 4q ..; transposed classes write pixel (2y+PY, 2x+PX)
  static_assert(!GM::T || NT == 1, "transposed layers: at most 16 output channels");
  const int Wo = GM::T ? 2 * a.wo : a.wo;
  unsigned ooff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co4 = PAIR8 ? 4 * (q & 1) : nt * 16 + 4 * q;
    const int py = GM::T ? 2 * row : row, px = GM::T ? 2 * p + (PAIR8 ? (q >> 1) : 0) : p;       // PAIR8: lanes q >= 2 hold the px = 1 class of their tile
    ooff[nt] = co4 < a.cout ? (unsigned)(((py * Wo + px) * a.ctot + a.co0 + co4) * 4) : BUF_OOB;
    pin(ooff[nt]);
  }

  auto load_tile = [&](f32x4 (&stage)[NL], int n, int tx, int ty) {
    const int ix0 = tx * TC * STR + GM::ORG, iy0 = ty * TR * STR + GM::ORG;
    const long pix0 = ((long)n * a.hi + iy0) * a.wi + ix0;
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
        stage[k] = buf_load4(k < NA ? ra : rb, ok ? goff[k] : BUF_OOB);              // zero padding
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
    const int oy0 = ty * TR, ox0 = tx * TC;
    const long opix0 = GM::T ? ((long)n * 2 * a.ho + 2 * oy0) * Wo + 2 * ox0 : ((long)n * a.ho + oy0) * a.wo + ox0;
    const buf_rsrc ro = make_rsrc((char*)a.out + opix0 * ((long)a.ctot * 4));
    const bool valid = oy0 + row < a.ho && ox0 + p < a.wo;

    f32x4 ctx[NT];                       // context terms of this tile's pixels (requested before the chain)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      ctx[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if ((EPI & FE_CONTEXT) && valid && nt * 16 + 4 * q < a.cout) {
        const int c = a.co0 + nt * 16 + 4 * q, y = oy0 + row, x = ox0 + p;
        ctx[nt] = up_sample4(a.ctxA, n, a.hA, a.wA, a.ctot, c, y, x, a.syA, a.sxA) +
                  up_sample4(a.ctxB, n, a.hB, a.wB, a.ctot, c, y, x, a.syB, a.sxB);
      }
      if ((EPI & FE_ADD_UP) && valid && nt * 16 + 4 * q < a.cout) {      // F.interpolate(scale_factor=2, mode="nearest"): source (y/2, x/2)
        const int c = a.co0 + nt * 16 + 4 * q, y = (oy0 + row) >> 1, x = (ox0 + p) >> 1;
        ctx[nt] = *(const f32x4*)(a.ctxA + (((size_t)n * a.hA + y) * a.wA + x) * a.ctot + c);
      }
    }
    const int tn = t + gridDim.x;
    const bool more = tn < tg.ntiles;
    int nn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, nn, txn, tyn);
      load_tile(stage, nn, txn, tyn);
    }

    f32x4 acc[GM::T ? 4 : NT];
#pragma unroll
    for (int i = 0; i < (GM::T ? 4 : NT); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ty9 = 0; ty9 < GM::KH; ++ty9)
#pragma unroll
      for (int tx9 = 0; tx9 < GM::KW; ++tx9)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const float bv = *(const float*)((const char*)lds + xbyte[kc] + (ty9 * LC + tx9) * 4);
          if constexpr (PAIR8) {
            if (ty9 == 0 && tx9 < 2) acc[0] = mfma16(pw[0][tx9][kc], bv, acc[0]);
            acc[1] = mfma16(pw[1][ty9 * 2 + tx9][kc], bv, acc[1]);
          } else if (GM::T) {          // input pixel (i+ty9, j+tx9) feeds the classes with py >= ty9, px >= tx9
#pragma unroll
            for (int cls = 0; cls < 4; ++cls) {
              const int py = cls >> 1, px = cls & 1;
              constexpr int cbase[4] = {0, 1, 3, 5};
              if (py >= ty9 && px >= tx9) acc[cls] = mfma16(wf[0][cbase[cls] + ty9 * (1 + px) + tx9][kc], bv, acc[cls]);
            }
          } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(wf[nt][ty9 * GM::KW + tx9][kc], bv, acc[nt]);
          }
        }

    wait_vmem_all();
    __syncthreads();                   // every wave is done reading the tile
    if (more) store_tile(stage);

    if constexpr (PAIR8) {             // tile t: output row 2 row + t, column 2p + (q >> 1) (in ooff), channels 4 (q & 1) ..
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        f32x4 v = acc[tl] + bias[0];
        if (EPI & FE_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        buf_store4(ro, (valid && ooff[0] != BUF_OOB) ? ooff[0] + (unsigned)(tl * Wo * a.ctot * 4) : BUF_OOB, v);
      }
    } else if (GM::T) {
#pragma unroll
      for (int cls = 0; cls < 4; ++cls) {
        f32x4 v = acc[cls] + bias[0];
        if (EPI & FE_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const unsigned o = ooff[0] + (unsigned)((((cls >> 1) * Wo) + (cls & 1)) * a.ctot * 4);
        buf_store4(ro, (valid && ooff[0] != BUF_OOB) ? o : BUF_OOB, v);
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 v = acc[nt] + bias[nt];
        if (EPI & FE_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (EPI & (FE_CONTEXT | FE_ADD_UP)) v += ctx[nt];
        buf_store4(ro, valid ? ooff[nt] : BUF_OOB, v);
      }
    }
    if (!more) break;
    __syncthreads();                   // next tile visible
    t = tn; n = nn; tx = txn; ty = tyn;
  }
}

// ---------------------------------------------------------------------------
// conv0 of both feature nets in ONE kernel (reference models/adamvs.py:57-59: Conv2d(3, 8, 3, 1) + BN + ReLU, Conv2d(8, 8, 3, 1) +
// BN + ReLU at full resolution), straight from the reference's [N][3][H][W] images.  As separate launches the two layers
// and the RGB repack moved 173 B per pixel (the 8-channel intermediate out and back in with its halo) at 3.3 - 3.5 TB/s;
// fused, a tile reads its 12 x 18 window of the image (23 B per output pixel) and writes its 8 x 14 outputs (32 B).  Fused they
// would be bound by matrix issue instead -- 8 output channels fill half an MFMA tile -- so both layers run in the two-row
// form of conv1 (slice_red.hip): MFMA rows 0-7 = the 8 channels of output row y, rows 8-15 = those of row y+1, fed by the
// same input-row fragment: 12 fragment passes per two rows instead of 18.
//   LDS: image window [4 planes: R, G, B, zero][12 x 18]; intermediate [2 channel groups][4][10 x 16] (the tile grown by one
//   pixel: zero outside the image, which is the second layer's padding).
//   stage A: five row pairs of the 10 x 16 region, one per wave (the fifth goes round the waves with the tile index);
//   stage B: four row pairs of the 8 x 14 tile (columns 14, 15 of a run are surplus), one per wave.
struct Conv0Args {
  const float* imgs;     // [N][3][H][W]
  const float* w0;       // two-row A fragments of conv0.0 [12][1][64] (RGB + a zero channel), BatchNorm scale folded in
  const float* b0;       // [16] BatchNorm shift (8 real)
  const float* w1;       // two-row A fragments of conv0.1 [12][2][64]
  const float* b1;       // [16]
  float* out;            // [N][H*W][8]
  int H, W;
  int B, V, n0;          // V > 0: imgs is [B][V][3][H][W] (the reference's forward() argument) and image n of this launch is view-major
                         // image m = n0 + n = v * B + b, read in place from imgs[b][v]; V == 0: imgs is [N][3][H][W]
};

__global__ __launch_bounds__(256) void k_conv0_fused(Conv0Args a, TileGrid tg) {
  constexpr int TR = 8, TC = 14, WR = TR + 4, WC = TC + 4, RR = TR + 2, RC = TC + 2;
  constexpr int PA = plane_pitch16(WR * WC), PLB = plane_pitch16(RR * RC), GPB = group_pitch(PLB, 2);
  __shared__ float lds[4 * PA + 2 * GPB];
  float* la = lds;               // image window, planar
  float* lb = lds + 4 * PA;      // intermediate, planar in two groups of four channels
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int H = a.H, W = a.W;
  const unsigned plane_bytes = (unsigned)((size_t)H * W * 4);

  float w0[12], w1[12][2];
#pragma unroll
  for (int t = 0; t < 12; ++t) {
    w0[t] = a.w0[t * 64 + lane];
    w1[t][0] = a.w1[(t * 2 + 0) * 64 + lane];
    w1[t][1] = a.w1[(t * 2 + 1) * 64 + lane];
  }
  const f32x4 bias0 = *(const f32x4*)(a.b0 + 4 * (q & 1)), bias1 = *(const f32x4*)(a.b1 + 4 * (q & 1));
  if (tid < PA) la[3 * PA + tid] = 0.f;                  // the fourth input channel (visible after the first barrier)

  // ---- per-lane constants
  const bool loader = tid < WR * WC;                      // one window pixel per thread: three channel planes
  const int lr = min(tid, WR * WC - 1) / WC, lc = min(tid, WR * WC - 1) % WC;
  unsigned goff = (unsigned)((lr * W + lc) * 4);
  pin(goff);
  // stage A: B-fragment origin of row pair 0 (k-row q = channel q); pair rp adds rp * 2 * WC
  unsigned xa = (unsigned)((q * PA + p) * 4);
  // its result: region pixel (2 rp + (q >> 1), p), channels 4 (q & 1) ..: planes of group q & 1
  unsigned sb = (unsigned)((((q & 1) * GPB) + (q >> 1) * RC + p) * 4);
  // stage B: B-fragment origin of the wave's row pair, per k-chunk
  unsigned xb[2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) { xb[kc] = (unsigned)((kc * GPB + q * PLB + (2 * wave) * RC + p) * 4); pin(xb[kc]); }
  pin(xa); pin(sb);
  const int orow = 2 * wave + (q >> 1);
  unsigned ooff = p < TC ? (unsigned)(((orow * W + p) * 8 + 4 * (q & 1)) * 4) : BUF_OOB;
  pin(ooff);

  auto load_tile = [&](float (&st)[3], int n, int tx, int ty) {
    const int ix0 = tx * TC - 2, iy0 = ty * TR - 2;
    const int m = a.n0 + n;
    const long img = a.V ? (long)(m % a.B) * a.V + m / a.B : (long)m;        // uniform
    const buf_rsrc ri = make_rsrc((const char*)a.imgs + ((img * 3 * H + iy0) * W + ix0) * 4);
    const bool ok = loader && (unsigned)(iy0 + lr) < (unsigned)H && (unsigned)(ix0 + lc) < (unsigned)W;
    const unsigned o = ok ? goff : BUF_OOB;               // zero padding of the first layer
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
      st[ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, o, (unsigned)ch * plane_bytes, 0));
  };
  auto store_tile = [&](const float (&st)[3]) {
    if (loader) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) la[ch * PA + tid] = st[ch];
    }
  };

  int t = blockIdx.x;
  if (t >= tg.ntiles) return;
  int n, tx, ty;
  tile_coords(tg, t, n, tx, ty);
  float stage[3];
  load_tile(stage, n, tx, ty);
  wait_vmem_all();
  store_tile(stage);
  __syncthreads();
  for (;;) {
    const int y0 = ty * TR, x0 = tx * TC;
    const int tn = t + gridDim.x;
    const bool more = tn < tg.ntiles;
    int nn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, nn, txn, tyn);
      load_tile(stage, nn, txn, tyn);                     // in flight during both stages
    }
    // ---- stage A: conv0.0 on the region, row pairs wave and (for one wave per tile) 4
    const int extra = t & 3;                               // uniform
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j == 1 && wave != extra) break;                  // wave-uniform
      const int rp = j ? 4 : wave;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const char* at = (const char*)la + xa + rp * (2 * WC * 4);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc = mfma16(w0[rr * 3 + kx], *(const float*)(at + (rr * WC + kx) * 4), acc);
      drain(acc);
      f32x4 v = acc + bias0;
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      const int gy = y0 - 1 + 2 * rp + (q >> 1), gx = x0 - 1 + p;
      if (!((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)) v = f32x4{0.f, 0.f, 0.f, 0.f};      // padding of conv0.1
      float* d = (float*)((char*)lb + sb + rp * (2 * RC * 4));
      d[0] = v.x; d[PLB] = v.y; d[2 * PLB] = v.z; d[3 * PLB] = v.w;
    }
    __syncthreads();                                       // the intermediate is complete; the image window is free
    // ---- stage B: conv0.1 on the tile
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
          acc = mfma16(w1[rr * 3 + kx][kc], *(const float*)((const char*)lb + xb[kc] + (rr * RC + kx) * 4), acc);
    drain(acc);
    wait_vmem_all();                                       // the next window has arrived (nothing else is pending)
    if (more) store_tile(stage);
    {
      f32x4 v = acc + bias1;
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      const buf_rsrc ro = make_rsrc((char*)a.out + (((long)n * H + y0) * W + x0) * 32);
      const bool valid = y0 + orow < H && x0 + p < W;
      buf_store4(ro, valid ? ooff : BUF_OOB, v);
    }
    if (!more) break;
    __syncthreads();                                       // next window visible; every wave is done with the intermediate
    t = tn; n = nn; tx = txn; ty = tyn;
  }
}

struct ViewOrder { int B, V, n0; };      // how the N images of a launch are found in `imgs` (Conv0Args)
static int launch_conv0_fused(const float* imgs, const adamvs_fconv_weights& c00, const adamvs_fconv_weights& c01, float* out, int N,
                              int H, int W, ViewOrder vo, hipStream_t st) {
  static const int capacity = resident_blocks(k_conv0_fused, 256, 0);
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(W, 14), cdiv(H, 8), N)) return rc;
  if ((size_t)H * W * 4 * 3 >= 0x7fffffffu) return set_error(-1, "feature net conv0: image too large for 32-bit plane offsets (%d x %d)", H, W);
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  Conv0Args a{imgs, c00.w, c00.b, c01.w, c01.b, out, H, W, vo.B, vo.V, vo.n0};
  hipLaunchKernelGGL(k_conv0_fused, dim3(grid), dim3(256), 0, st, a, tg);
  ADAMVS_CHECK_LAUNCH("feature net conv0 (fused)");
  return 0;
}

// deconv2.conv (reference models/module.py:506-524, DeConv2dFuse: 3x3 on cat(deconv output, skip), 16 -> 8, BN, ReLU) at full
// resolution on the two-row tile loop of the level-1 candidate convolution (slice_roles.h): 8 output channels fill half an
// MFMA tile in the one-row form of k_fconv (36 MFMAs per 16-pixel run); two rows share the input-row fragments (24 per run),
// and the 8 x 16 tile re-reads less halo than k_fconv's 4 x 16 (1.41 x instead of 1.69 x).
__global__ __launch_bounds__(256) void k_fconv_pair_two_row(SmallConvArgs a, TileGrid tg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  TwoRowPairRole<TR_BIAS_RELU>::run(a, tg, TileRange{0, tg.ntiles}, blockIdx.x, gridDim.x, lds);
}

static int launch_pair_two_row(const float* srcA, const float* srcB, const adamvs_fconv_weights& w, float* out, int N, int h, int wd,
                               hipStream_t st) {
  constexpr size_t lds = TwoRowPairRole<TR_BIAS_RELU>::LDS_BYTES;
  static const int capacity = resident_blocks(k_fconv_pair_two_row, 256, lds);
  SmallConvArgs a{srcA, srcB, w.w, w.b, out, nullptr, h, wd, h, wd, 8, nullptr};
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(wd, 16), cdiv(h, 8), N)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(k_fconv_pair_two_row, dim3(grid), dim3(256), lds, st, a, tg);
  ADAMVS_CHECK_LAUNCH("feature net deconv2.conv (two-row)");
  return 0;
}

// ---------------------------------------------------------------------------
// The stride-1 3 x 3 layers in the minimal-filtering form F(2, 3) ALONG X (round 5; the form of conv1 of the recurrent net,
// slice_red.hip::k_conv1_f23): a lane owns the output pixels (row, 2p) and (row, 2p + 1); per kernel row ky and input channel it
// reads the four pixels 2p .. 2p + 3 (two 8-byte LDS reads), forms (d0 - d2, d1 + d2, d2 - d1, d1 - d3) and feeds four MFMAs with
// U = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2) -- 12 products per two pixels and channel pair instead of 18, two LDS reads
// instead of six, on tiles of 4 x 32 pixels instead of 4 x 16 (half the barriers per pixel).  U is formed per lane from the nine
// tap fragments of the layer as they are packed for k_fconv (every tap fragment holds the same (cout, cin) element in a lane):
// no second weight format.  y(2p) = m0 + m1 + m2, y(2p + 1) = m1 - m2 - m3.  These layers sat at 63 - 76 % of the fp32 matrix rate
// on k_fconv (conv1.1 / 1.2, conv2.1 / 2.2, deconv1.conv: 2.9 of FeatureNet0's 9.8 ms per 160 images).
template <int CA, int CB, int NT, int EPI>
__global__ __launch_bounds__(256, (CA + CB == 16 && NT == 1) ? 4 : 1) void k_fconv_f23(FConvArgs a, TileGrid tg) {      // (16 channels: 132 registers uncapped -- three waves per SIMD; 128: four)
  static_assert(!(EPI & (FE_CONTEXT | FE_ADD_UP)), "plain epilogues");
  constexpr int CIN = CA + CB, KC = CIN / 4, G = CIN / 4, GA = CA / 4, GB = CB / 4;
  constexpr int TR = 4, TC = 32, LR = TR + 2, LC = TC + 2, NPIX = LR * LC;
  constexpr int PLANE = plane_pitch16(NPIX), GP = group_pitch(PLANE, G);
  static_assert((PLANE % 2) == 0 && (GP % 2) == 0 && (LC % 2) == 0, "8-byte aligned pair reads");
  constexpr int NA = (NPIX * GA + 255) / 256, NB = (NPIX * GB + 255) / 256, NL = NA + NB;
  extern __shared__ float lds[];         // [G][GP]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int row = wave;                  // the wave's run: row `wave` of the 4 x 32 tile

  float uf[NT][3][4][KC];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const float g0 = a.wpk[((nt * 9 + ky * 3 + 0) * KC + kc) * 64 + lane], g1 = a.wpk[((nt * 9 + ky * 3 + 1) * KC + kc) * 64 + lane],
                    g2 = a.wpk[((nt * 9 + ky * 3 + 2) * KC + kc) * 64 + lane];
        uf[nt][ky][0][kc] = g0;
        uf[nt][ky][1][kc] = 0.5f * ((g0 + g2) + g1);
        uf[nt][ky][2][kc] = 0.5f * ((g0 + g2) - g1);
        uf[nt][ky][3][kc] = g2;
      }
  f32x4 bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bias[nt] = *(const f32x4*)(a.bias + nt * 16 + 4 * q);

  // ---- per-lane constants
  unsigned goff[NL], lbyte[NL];
  int rc[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const bool isA = k < NA;
    const int gs = isA ? GA : GB, cs = isA ? CA : CB;
    int j = tid + (isA ? k : k - NA) * 256;
    j = min(j, NPIX * gs - 1);           // surplus lanes repeat the last item
    const int g = j % gs, pp = j / gs, r = pp / LC, c = pp % LC;
    goff[k] = (unsigned)(((r * a.wi + c) * cs + 4 * g) * 4);
    lbyte[k] = (unsigned)((((isA ? 0 : GA) + g) * GP + r * LC + c) * 4);
    rc[k] = r | (c << 16);
    pin(goff[k]); pin(lbyte[k]); pin(rc[k]);
  }
  unsigned xbyte[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    xbyte[kc] = (unsigned)((kc * GP + q * PLANE + row * LC + 2 * p) * 4);
    pin(xbyte[kc]);
  }
  unsigned ooff[NT];                     // the lane's pixel (row, 2p), channels co0 + 16 nt + 4q ..; (row, 2p + 1) is ctot floats further
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co4 = nt * 16 + 4 * q;
    ooff[nt] = co4 < a.cout ? (unsigned)(((row * a.wo + 2 * p) * a.ctot + a.co0 + co4) * 4) : BUF_OOB;
    pin(ooff[nt]);
  }

  auto load_tile = [&](f32x4 (&stage)[NL], int n, int tx, int ty) {
    const int ix0 = tx * TC - 1, iy0 = ty * TR - 1;
    const long pix0 = ((long)n * a.hi + iy0) * a.wi + ix0;
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
        stage[k] = buf_load4(k < NA ? ra : rb, ok ? goff[k] : BUF_OOB);              // zero padding
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
  typedef float f32x2f __attribute__((ext_vector_type(2)));
  for (;;) {
    const int oy0 = ty * TR, ox0 = tx * TC;
    const long opix0 = ((long)n * a.ho + oy0) * a.wo + ox0;
    const buf_rsrc ro = make_rsrc((char*)a.out + opix0 * ((long)a.ctot * 4));
    const bool valid0 = oy0 + row < a.ho && ox0 + 2 * p < a.wo, valid1 = oy0 + row < a.ho && ox0 + 2 * p + 1 < a.wo;
    const int tn = t + gridDim.x;
    const bool more = tn < tg.ntiles;
    int nn = 0, txn = 0, tyn = 0;
    if (more) {
      tile_coords(tg, tn, nn, txn, tyn);
      load_tile(stage, nn, txn, tyn);
    }

    f32x4 acc[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[nt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const char* at = (const char*)lds + xbyte[kc] + ky * LC * 4;
        const f32x2f d01 = *(const f32x2f*)at, d23 = *(const f32x2f*)(at + 8);
        // (v0, v3) = d01 - d23 and (v1, v2) = (d2 + d1, d2 - d1) as two packed instructions (written on the vector types: as scalars
        // the compiler emits four)
        const f32x2f v03 = d01 - d23;
        const f32x2f v12 = __builtin_elementwise_fma(__builtin_shufflevector(d01, d01, 1, 1), f32x2f{1.0f, -1.0f}, __builtin_shufflevector(d23, d23, 0, 0));
        const float v0 = v03.x, v1 = v12.x, v2 = v12.y, v3 = v03.y;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[nt][0] = mfma16(uf[nt][ky][0][kc], v0, acc[nt][0]);
          acc[nt][1] = mfma16(uf[nt][ky][1][kc], v1, acc[nt][1]);
          acc[nt][2] = mfma16(uf[nt][ky][2][kc], v2, acc[nt][2]);
          acc[nt][3] = mfma16(uf[nt][ky][3][kc], v3, acc[nt][3]);
        }
      }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) drain(acc[nt][j]);

    wait_vmem_all();
    __syncthreads();                   // every wave is done reading the tile
    if (more) store_tile(stage);

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 y0 = ((acc[nt][0] + acc[nt][1]) + acc[nt][2]) + bias[nt];
      f32x4 y1 = ((acc[nt][1] - acc[nt][2]) - acc[nt][3]) + bias[nt];
      if (EPI & FE_RELU) {
        y0.x = fmaxf(y0.x, 0.f); y0.y = fmaxf(y0.y, 0.f); y0.z = fmaxf(y0.z, 0.f); y0.w = fmaxf(y0.w, 0.f);
        y1.x = fmaxf(y1.x, 0.f); y1.y = fmaxf(y1.y, 0.f); y1.z = fmaxf(y1.z, 0.f); y1.w = fmaxf(y1.w, 0.f);
      }
      buf_store4(ro, valid0 ? ooff[nt] : BUF_OOB, y0);
      buf_store4(ro, (valid1 && ooff[nt] != BUF_OOB) ? ooff[nt] + (unsigned)(a.ctot * 4) : BUF_OOB, y1);
    }
    if (!more) break;
    __syncthreads();                   // next tile visible
    t = tn; n = nn; tx = txn; ty = tyn;
  }
}

// option fconv_f23 = 0: the stride-1 3 x 3 layers on k_fconv, as in rounds 2 - 4 (A/B)
static bool fconv_f23() { return opt(OPT_FCONV_F23) != 0; }

template <int CA, int CB, int NT, int EPI>
static int launch_fconv_f23(const FConvArgs& a, int N, hipStream_t st, const char* name) {
  constexpr int G = (CA + CB) / 4, NPIX = 6 * 34;
  constexpr size_t lds = (size_t)G * group_pitch(plane_pitch16(NPIX), G) * sizeof(float);
  static_assert(lds <= 64 * 1024, "tile exceeds the default dynamic LDS limit");
  auto kern = k_fconv_f23<CA, CB, NT, EPI>;
  static const int capacity = resident_blocks(kern, 256, lds);      // once per instantiation, thread-safely (magic static)
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(a.wo, 32), cdiv(a.ho, 4), N)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, tg);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "feature_net0 %s (F(2,3) along x): %s", name, hipGetErrorString(e));
  return 0;
}

template <int CA, int CB, int NT, int MODE, int EPI>
static int launch_fconv(const FConvArgs& a_in, int N, hipStream_t st, const char* name) {
  using GM = FGeom<MODE>;
  if constexpr (MODE == FM_K3 && !(EPI & (FE_CONTEXT | FE_ADD_UP))) {
    if (fconv_f23()) return launch_fconv_f23<CA, CB, NT, EPI>(a_in, N, st, name);
  }
  FConvArgs a = a_in;
  if (EPI & FE_CONTEXT) {
    a.syA = (float)a.hA / (float)a.ho; a.sxA = (float)a.wA / (float)a.wo;
    a.syB = (float)a.hB / (float)a.ho; a.sxB = (float)a.wB / (float)a.wo;
  }
  constexpr int G = (CA + CB) / 4, NPIX = GM::rows(4) * GM::cols(16);
  constexpr int PLANE = (GM::STR == 1) ? plane_pitch16(NPIX) : (NPIX | 1);
  constexpr size_t lds = (size_t)G * group_pitch(PLANE, G) * sizeof(float);
  static_assert(lds <= 64 * 1024, "tile exceeds the default dynamic LDS limit");
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(a.wo, 16), cdiv(a.ho, 4), N)) return rc;
  if constexpr (GM::T && NT == 1) {
    if (a.cout <= 8 && fconv_f23()) {      // (the same switch: ADAMVS_FCONV_F23=0 = rounds 2 - 4)
      auto kp = k_fconv<CA, CB, NT, MODE, EPI, true>;
      static const int cap8 = resident_blocks(kp, 256, lds);
      hipLaunchKernelGGL(kp, dim3(tg.ntiles < cap8 ? tg.ntiles : cap8), dim3(256), lds, st, a, tg);
      hipError_t e8 = hipGetLastError();
      if (e8 != hipSuccess) return set_error((int)e8, "feature_net0 %s (paired classes): %s", name, hipGetErrorString(e8));
      return 0;
    }
  }
  auto kern = k_fconv<CA, CB, NT, MODE, EPI>;
  static const int capacity = resident_blocks(kern, 256, lds);      // once per instantiation, thread-safely (magic static)
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, tg);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "feature_net0 %s: %s", name, hipGetErrorString(e));
  return 0;
}

// ---------------------------------------------------------------------------
// The full-resolution output convolution (out3: 1x1, 8 -> 8 channels, + the two upsampled context maps) as a streaming kernel:
// one thread per pixel.  On k_fconv's 4 x 16-pixel tiles it was a 2-MFMA chain per wave between a window fill, two barriers and
// 140 vector instructions of bilinear taps (5.4 % MFMA busy, 70 vector instructions per MFMA, 2.9 ms per 160 images: the
// per-tile latency chain of 64 pixels); as 64 FMAs per pixel on the vector units with the weights in scalar registers it
// moves its 64 bytes per pixel and nothing else (FeatureNet0 0.335 -> 0.310 ms per tile; the same for out2 -- 16 channels at half
// resolution, 256 FMAs and 32 tap loads per thread -- was slower than its MFMA kernel: 0.325 -> 0.333).  Same arithmetic as the
// MFMA chain up to the order of the eight products.
__global__ __launch_bounds__(256) void k_out8_context(FConvArgs a, size_t npix) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const int hw = a.ho * a.wo;
  const int n = (int)(i / hw), pp = (int)(i % hw), y = pp / a.wo, x = pp % a.wo;
  const f32x4 f0 = *(const f32x4*)(a.srcA + i * 8), f1 = *(const f32x4*)(a.srcA + i * 8 + 4);
  const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
  f32x4 o[2];                          // all sixteen tap loads are issued here, ahead of the products
#pragma unroll
  for (int h4 = 0; h4 < 2; ++h4) {
    o[h4] = *(const f32x4*)(a.bias + 4 * h4) + up_sample4(a.ctxA, n, a.hA, a.wA, 8, 4 * h4, y, x, a.syA, a.sxA) +
            up_sample4(a.ctxB, n, a.hB, a.wB, 8, 4 * h4, y, x, a.syB, a.sxB);
  }
  float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // A fragments [KC = 2][64]: element lane l of fragment kc = W[cout = l & 15][cin = 4 kc + (l >> 4)] (uniform addresses: scalar loads)
#pragma unroll
  for (int ci = 0; ci < 8; ++ci)
#pragma unroll
    for (int co = 0; co < 8; ++co) r[co] = __fmaf_rn(a.wpk[(ci >> 2) * 64 + (ci & 3) * 16 + co], f[ci], r[co]);
  o[0] += f32x4{r[0], r[1], r[2], r[3]};
  o[1] += f32x4{r[4], r[5], r[6], r[7]};
  *(f32x4*)(a.out + i * 8) = o[0];
  *(f32x4*)(a.out + i * 8 + 4) = o[1];
}

static int launch_out8_context(const FConvArgs& a_in, int N, hipStream_t st) {
  FConvArgs a = a_in;
  a.syA = (float)a.hA / (float)a.ho; a.sxA = (float)a.wA / (float)a.wo;
  a.syB = (float)a.hB / (float)a.ho; a.sxB = (float)a.wB / (float)a.wo;
  const size_t npix = (size_t)N * a.ho * a.wo;
  hipLaunchKernelGGL(k_out8_context, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, a, npix);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "feature_net0 out3: %s", hipGetErrorString(e));
  return 0;
}

// ---------------------------------------------------------------------------
// Pooled-context branches of one stage: AvgPool2d(4) and AvgPool2d(8), each -> 1x1 conv (C -> C/2, BN folded) -> ReLU
// -> 1x1 (C/2 -> C: the branch's columns of the stage's output convolution).  One thread per 8 x 8 block of the
// feature map: it is read once, the four 4 x 4 means feed branch a and their mean feeds branch b.  The pooled maps
// are 16 and 64 times smaller than the feature map, so this is a plain streaming kernel.
//   w1 [C/2][C] (scale folded), b1 [C/2], w2 [C][C/2];  outA [N][(h/4)(w/4)][C], outB [N][(h/8)(w/8)][C]
template <int C>
__global__ __launch_bounds__(256, C == 8 ? 4 : 1) void k_context(const float* __restrict__ feat, adamvs_context_weights wa,
                                                 adamvs_context_weights wb, float* __restrict__ outA,
                                                 float* __restrict__ outB, int h, int w, size_t total) {
  constexpr int C2 = C / 2;
  __shared__ float s_w1[2][C2 * C], s_b1[2][C2], s_w2[2][C * C2];
  for (int i = threadIdx.x; i < C2 * C; i += 256) {
    s_w1[0][i] = wa.w1[i]; s_w2[0][i] = wa.w2[i];
    s_w1[1][i] = wb.w1[i]; s_w2[1][i] = wb.w2[i];
  }
  for (int i = threadIdx.x; i < C2; i += 256) { s_b1[0][i] = wa.b1[i]; s_b1[1][i] = wb.b1[i]; }
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;       // 8 x 8 block
  if (i >= total) return;
  const int w8 = w / 8, h8 = h / 8, w4 = w / 4, h4 = h / 4;
  const int xb = (int)(i % w8), yb = (int)((i / w8) % h8);
  const size_t n = i / ((size_t)w8 * h8);

  auto branch = [&](const float (&pooled)[C], int which, float* o) {
    float mid[C2];
#pragma unroll
    for (int m = 0; m < C2; ++m) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) s += s_w1[which][m * C + c] * pooled[c];
      mid[m] = fmaxf(s + s_b1[which][m], 0.f);
    }
#pragma unroll
    for (int co = 0; co < C; co += 4) {
      f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < C2; ++m) {
        r.x += s_w2[which][(co + 0) * C2 + m] * mid[m]; r.y += s_w2[which][(co + 1) * C2 + m] * mid[m];
        r.z += s_w2[which][(co + 2) * C2 + m] * mid[m]; r.w += s_w2[which][(co + 3) * C2 + m] * mid[m];
      }
      *(f32x4*)(o + co) = r;
    }
  };

  float p8[C];
#pragma unroll
  for (int c = 0; c < C; ++c) p8[c] = 0.f;
  for (int sy = 0; sy < 2; ++sy)
    for (int sx = 0; sx < 2; ++sx) {
      float p4[C];
#pragma unroll
      for (int c = 0; c < C; ++c) p4[c] = 0.f;
      const float* base = feat + (((size_t)n * h + (size_t)yb * 8 + sy * 4) * w + (size_t)xb * 8 + sx * 4) * C;
      for (int dy = 0; dy < 4; ++dy)
        for (int dx = 0; dx < 4; ++dx) {
          const f32x4* px = (const f32x4*)(base + ((size_t)dy * w + dx) * C);
#pragma unroll
          for (int g = 0; g < C / 4; ++g) {
            const f32x4 v = px[g];
            p4[4 * g] += v.x; p4[4 * g + 1] += v.y; p4[4 * g + 2] += v.z; p4[4 * g + 3] += v.w;
          }
        }
#pragma unroll
      for (int c = 0; c < C; ++c) { p8[c] += p4[c]; p4[c] *= (1.0f / 16.0f); }
      branch(p4, 0, outA + (((size_t)n * h4 + (size_t)yb * 2 + sy) * w4 + (size_t)xb * 2 + sx) * C);
    }
#pragma unroll
  for (int c = 0; c < C; ++c) p8[c] *= (1.0f / 64.0f);
  branch(p8, 1, outB + i * C);
}

template <int C>
static int launch_context(const float* feat, const adamvs_context_weights& wa, const adamvs_context_weights& wb, float* outA,
                          float* outB, int N, int h, int w, hipStream_t st) {
  const size_t total = (size_t)N * (h / 8) * (w / 8);
  hipLaunchKernelGGL(k_context<C>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, feat, wa, wb, outA, outB, h, w, total);
  ADAMVS_CHECK_LAUNCH("feature_net0 context branches");
  return 0;
}

static inline size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }

}  // namespace adamvs

using namespace adamvs;

// Workspace layout (floats), N images of H x W (both multiples of 32: three /2 levels, AvgPool 8 at the coarsest):
//   c0 [N][HW][8], c1a, c1b, c1 [N][HW/4][16], c2a, c2b, c2 [N][HW/16][32],
//   d1 [N][HW/4][16] (deconv1.deconv), f1 [N][HW/4][16], d2 [N][HW][8], f2 [N][HW][8], context maps.
extern "C" size_t adamvs_feature_net0_workspace_bytes(int N, int H, int W) {
  const size_t hw = (size_t)H * W, n = (size_t)N;
  size_t f = al64(n * hw * 8) + 3 * al64(n * hw / 4 * 16) + 3 * al64(n * hw / 16 * 32) +
             2 * al64(n * hw / 4 * 16) + 2 * al64(n * hw * 8);
  f += 2 * al64(n * (hw / 16 / 16) * 32) + 2 * al64(n * (hw / 4 / 16) * 16) + 2 * al64(n * (hw / 16) * 8);      // pooled by 4 (x2 for 8: smaller)
  return f * sizeof(float);
}

// conv0 / conv1 / conv2 of both feature nets (reference models/adamvs.py:57-76 = models/msrednet.py:38-54): imgs -> c0 [N][HW][8],
// c1 [N][HW/4][16], c2 [N][HW/16][32]; c1a, c1b, c2a, c2b are scratch.
struct EncoderWeights { adamvs_fconv_weights conv0_0, conv0_1, conv1_0, conv1_1, conv1_2, conv2_0, conv2_1, conv2_2; };
static FConvArgs fconv_args(const float* sa, const float* sb, const adamvs_fconv_weights& w, float* out, int hi, int wi, int ho, int wo,
                            int cout, int ctot, int co0) {
  return FConvArgs{sa, sb, w.w, w.b, out, nullptr, nullptr, hi, wi, ho, wo, cout, ctot, co0, 0, 0, 0, 0, 0.f, 0.f, 0.f, 0.f};
}
static int run_encoder(const float* imgs, const EncoderWeights& fw, float* c0, float* c1a, float* c1b, float* c1,
                       float* c2a, float* c2b, float* c2, int N, int H, int W, hipStream_t st, ViewOrder vo = ViewOrder{0, 0, 0}) {
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
  int rc;
  auto A = fconv_args;
  // conv0: 3 -> 8 -> 8 at full resolution, one kernel straight from the [N][3][H][W] images
  if ((rc = launch_conv0_fused(imgs, fw.conv0_0, fw.conv0_1, c0, N, H, W, vo, st))) return rc;
  // conv1: 5x5 stride 2 (8 -> 16), two 3x3
  if ((rc = launch_fconv<8, 0, 1, FM_K5S2, FE_RELU>(A(c0, nullptr, fw.conv1_0, c1a, H, W, H2, W2, 16, 16, 0), N, st, "conv1.0"))) return rc;
  if ((rc = launch_fconv<16, 0, 1, FM_K3, FE_RELU>(A(c1a, nullptr, fw.conv1_1, c1b, H2, W2, H2, W2, 16, 16, 0), N, st, "conv1.1"))) return rc;
  if ((rc = launch_fconv<16, 0, 1, FM_K3, FE_RELU>(A(c1b, nullptr, fw.conv1_2, c1, H2, W2, H2, W2, 16, 16, 0), N, st, "conv1.2"))) return rc;
  // conv2: 5x5 stride 2 (16 -> 32: two launches of 16 output channels, 100 fragment registers each), two 3x3
  for (int half = 0; half < 2; ++half) {
    adamvs_fconv_weights wh{fw.conv2_0.w + (size_t)half * 25 * 4 * 64, fw.conv2_0.b + half * 16};
    if ((rc = launch_fconv<16, 0, 1, FM_K5S2, FE_RELU>(A(c1, nullptr, wh, c2a, H2, W2, H4, W4, 16, 32, 16 * half), N, st, "conv2.0"))) return rc;
  }
  if ((rc = launch_fconv<32, 0, 2, FM_K3, FE_RELU>(A(c2a, nullptr, fw.conv2_1, c2b, H4, W4, H4, W4, 32, 32, 0), N, st, "conv2.1"))) return rc;
  if ((rc = launch_fconv<32, 0, 2, FM_K3, FE_RELU>(A(c2b, nullptr, fw.conv2_2, c2, H4, W4, H4, W4, 32, 32, 0), N, st, "conv2.2"))) return rc;
  return 0;
}

// ---- the FPN variant of MS-REDNet's FeatureNet (reference models/msrednet.py:74-91, 115-125; arch_mode "fpn", three stages).
// Workspace (floats): the encoder's maps, then t1 [N][HW/4][32] and t2 [N][HW][32] (the top-down maps).
extern "C" size_t adamvs_feature_net_fpn_workspace_bytes(int N, int H, int W) {
  const size_t hw = (size_t)H * W, n = (size_t)N;
  const size_t f = al64(n * hw * 8) + 3 * al64(n * hw / 4 * 16) + 3 * al64(n * hw / 16 * 32) +
                   al64(n * hw / 4 * 32) + al64(n * hw * 32);
  return f * sizeof(float);
}

extern "C" int adamvs_feature_net_fpn(const float* imgs, const adamvs_feature_fpn_weights* wts, float* stage1, float* stage2,
                                      float* stage3, int N, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
  ADAMVS_CHECK_ARG(imgs && wts && stage1 && stage2 && stage3 && workspace, "feature_net_fpn: null pointer");
  ADAMVS_CHECK_ARG(N > 0 && H >= 32 && W >= 32 && (H % 32) == 0 && (W % 32) == 0,
                   "feature_net_fpn: N=%d H=%d W=%d (H, W multiples of 32)", N, H, W);
  ADAMVS_CHECK_ARG(workspace_bytes >= adamvs_feature_net_fpn_workspace_bytes(N, H, W), "feature_net_fpn: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const adamvs_feature_fpn_weights& fw = *wts;
  const size_t hw = (size_t)H * W, n = (size_t)N;
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
  float* p = (float*)workspace;
  auto take = [&](size_t floats) { float* r = p; p += al64(floats); return r; };
  float* c0 = take(n * hw * 8);
  float* c1a = take(n * hw / 4 * 16); float* c1b = take(n * hw / 4 * 16); float* c1 = take(n * hw / 4 * 16);
  float* c2a = take(n * hw / 16 * 32); float* c2b = take(n * hw / 16 * 32); float* c2 = take(n * hw / 16 * 32);
  float* t1 = take(n * hw / 4 * 32); float* t2 = take(n * hw * 32);
  int rc;
  const EncoderWeights enc{fw.conv0_0, fw.conv0_1, fw.conv1_0, fw.conv1_1, fw.conv1_2, fw.conv2_0, fw.conv2_1, fw.conv2_2};
  if ((rc = run_encoder(imgs, enc, c0, c1a, c1b, c1, c2a, c2b, c2, N, H, W, st))) return rc;
  auto A = fconv_args;
  // stage 1: out1 (1x1, 32 -> 32) on conv2
  if ((rc = launch_fconv<32, 0, 2, FM_K1, 0>(A(c2, nullptr, fw.out1, stage1, H4, W4, H4, W4, 32, 32, 0), N, st, "fpn out1"))) return rc;
  // t1 = nearest2x(conv2) + inner1(conv1) (1x1, 16 -> 32, bias); stage 2: out2 (3x3, 32 -> 16) on t1
  {
    FConvArgs a = A(c1, nullptr, fw.inner1, t1, H2, W2, H2, W2, 32, 32, 0);
    a.ctxA = c2; a.hA = H4; a.wA = W4;
    if ((rc = launch_fconv<16, 0, 2, FM_K1, FE_ADD_UP>(a, N, st, "fpn inner1"))) return rc;
  }
  if ((rc = launch_fconv<32, 0, 1, FM_K3, 0>(A(t1, nullptr, fw.out2, stage2, H2, W2, H2, W2, 16, 16, 0), N, st, "fpn out2"))) return rc;
  // t2 = nearest2x(t1) + inner2(conv0) (1x1, 8 -> 32, bias); stage 3: out3 (3x3, 32 -> 8) on t2
  {
    FConvArgs a = A(c0, nullptr, fw.inner2, t2, H, W, H, W, 32, 32, 0);
    a.ctxA = t1; a.hA = H2; a.wA = W2;
    if ((rc = launch_fconv<8, 0, 2, FM_K1, FE_ADD_UP>(a, N, st, "fpn inner2"))) return rc;
  }
  if ((rc = launch_fconv<32, 0, 1, FM_K3, 0>(A(t2, nullptr, fw.out3, stage3, H, W, H, W, 8, 8, 0), N, st, "fpn out3"))) return rc;
  return 0;
}

static int feature_net0_impl(const float* imgs, const adamvs_feature_weights* wts, float* stage1, float* stage2, float* stage3, int N,
                             int H, int W, void* workspace, size_t workspace_bytes, void* stream, ViewOrder vo);

extern "C" int adamvs_feature_net0(const float* imgs, const adamvs_feature_weights* wts, float* stage1, float* stage2,
                                   float* stage3, int N, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
  return feature_net0_impl(imgs, wts, stage1, stage2, stage3, N, H, W, workspace, workspace_bytes, stream, ViewOrder{0, 0, 0});
}

extern "C" int adamvs_feature_net0_views(const float* imgs, const adamvs_feature_weights* wts, float* stage1, float* stage2,
                                         float* stage3, int B, int V, int n0, int n, int H, int W, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  ADAMVS_CHECK_ARG(B > 0 && V > 0 && n0 >= 0 && n > 0 && n0 + n <= B * V, "feature_net0_views: B=%d V=%d n0=%d n=%d", B, V, n0, n);
  return feature_net0_impl(imgs, wts, stage1, stage2, stage3, n, H, W, workspace, workspace_bytes, stream, ViewOrder{B, V, n0});
}

static int feature_net0_impl(const float* imgs, const adamvs_feature_weights* wts, float* stage1, float* stage2, float* stage3, int N,
                             int H, int W, void* workspace, size_t workspace_bytes, void* stream, ViewOrder vo) {
  ADAMVS_CHECK_ARG(imgs && wts && stage1 && stage2 && stage3 && workspace, "feature_net0: null pointer");
  ADAMVS_CHECK_ARG(N > 0 && H >= 32 && W >= 32 && (H % 32) == 0 && (W % 32) == 0,
                   "feature_net0: N=%d H=%d W=%d (H, W multiples of 32)", N, H, W);
  ADAMVS_CHECK_ARG(workspace_bytes >= adamvs_feature_net0_workspace_bytes(N, H, W), "feature_net0: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const adamvs_feature_weights& fw = *wts;
  const size_t hw = (size_t)H * W, n = (size_t)N;
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
  float* p = (float*)workspace;
  auto take = [&](size_t floats) { float* r = p; p += al64(floats); return r; };
  float* c0 = take(n * hw * 8);
  float* c1a = take(n * hw / 4 * 16); float* c1b = take(n * hw / 4 * 16); float* c1 = take(n * hw / 4 * 16);
  float* c2a = take(n * hw / 16 * 32); float* c2b = take(n * hw / 16 * 32); float* c2 = take(n * hw / 16 * 32);
  float* d1 = take(n * hw / 4 * 16); float* f1 = take(n * hw / 4 * 16);
  float* d2 = take(n * hw * 8); float* f2 = take(n * hw * 8);
  float* x1a = take(n * (hw / 16 / 16) * 32); float* x1b = take(n * (hw / 16 / 16) * 32);
  float* x2a = take(n * (hw / 4 / 16) * 16); float* x2b = take(n * (hw / 4 / 16) * 16);
  float* x3a = take(n * (hw / 16) * 8); float* x3b = take(n * (hw / 16) * 8);
  int rc;
  const EncoderWeights enc{fw.conv0_0, fw.conv0_1, fw.conv1_0, fw.conv1_1, fw.conv1_2, fw.conv2_0, fw.conv2_1, fw.conv2_2};
  if ((rc = run_encoder(imgs, enc, c0, c1a, c1b, c1, c2a, c2b, c2, N, H, W, st, vo))) return rc;
  auto A = fconv_args;
  // stage 1 output: out1 . cat(up(branch1_1), up(branch1_2), c2)
  if ((rc = launch_context<32>(c2, fw.br1_1, fw.br1_2, x1a, x1b, N, H4, W4, st))) return rc;
  {
    FConvArgs a{c2, nullptr, fw.out1.w, fw.out1.b, stage1, x1a, x1b, H4, W4, H4, W4, 32, 32, 0, H4 / 4, W4 / 4, H4 / 8, W4 / 8, 0.f, 0.f, 0.f, 0.f};
    if ((rc = launch_fconv<32, 0, 2, FM_K1, FE_CONTEXT>(a, N, st, "out1"))) return rc;
  }
  // deconv1: ConvTranspose2d 32 -> 16 (four parity classes) + BN + ReLU, cat with conv1, 3x3 32 -> 16
  if ((rc = launch_fconv<32, 0, 1, FM_TALL, FE_RELU>(A(c2, nullptr, fw.deconv1_t, d1, H4, W4, H4, W4, 16, 16, 0), N, st, "deconv1.deconv"))) return rc;
  if ((rc = launch_fconv<16, 16, 1, FM_K3, FE_RELU>(A(d1, c1, fw.deconv1_c, f1, H2, W2, H2, W2, 16, 16, 0), N, st, "deconv1.conv"))) return rc;
  // stage 2 output
  if ((rc = launch_context<16>(f1, fw.br2_1, fw.br2_2, x2a, x2b, N, H2, W2, st))) return rc;
  {
    FConvArgs a{f1, nullptr, fw.out2.w, fw.out2.b, stage2, x2a, x2b, H2, W2, H2, W2, 16, 16, 0, H2 / 4, W2 / 4, H2 / 8, W2 / 8, 0.f, 0.f, 0.f, 0.f};
    if ((rc = launch_fconv<16, 0, 1, FM_K1, FE_CONTEXT>(a, N, st, "out2"))) return rc;
  }
  // deconv2: ConvTranspose2d 16 -> 8, cat with conv0, 3x3 16 -> 8
  if ((rc = launch_fconv<16, 0, 1, FM_TALL, FE_RELU>(A(f1, nullptr, fw.deconv2_t, d2, H2, W2, H2, W2, 8, 8, 0), N, st, "deconv2.deconv"))) return rc;
  if ((rc = launch_pair_two_row(d2, c0, fw.deconv2_c, f2, N, H, W, st))) return rc;
  // stage 3 output
  if ((rc = launch_context<8>(f2, fw.br3_1, fw.br3_2, x3a, x3b, N, H, W, st))) return rc;
  {
    FConvArgs a{f2, nullptr, fw.out3.w, fw.out3.b, stage3, x3a, x3b, H, W, H, W, 8, 8, 0, H / 4, W / 4, H / 8, W / 8, 0.f, 0.f, 0.f, 0.f};
    if ((rc = launch_out8_context(a, N, st))) return rc;
  }
  return 0;
}
