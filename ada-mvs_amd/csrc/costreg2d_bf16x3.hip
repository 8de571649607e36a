// CostRegNet2D layers on the bf16 matrix cores with split operands ("bf16x3").
//
// Same layer semantics, tiling and epilogue as costreg2d.hip (reference models/adamvs.py:198-238), but the
// channel contraction runs on v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate) with every fp32 operand
// split into two bf16 halves, x = hi + lo (hi = bf16(x), lo = bf16(x - hi), 16 significant bits together):
//     a.b  ~  a_hi.b_hi + a_hi.b_lo + a_lo.b_hi            (fp32 accumulation in the MFMA)
// Three MFMAs per 32-deep k-step, i.e. 16/3 = 5.3x fewer matrix-pipe cycles than exact fp32.  Activations
// stay fp32 in HBM: they are split on the fly when a tile is written to LDS; weights are split on the host.
// Measured against the fp32 path the maps agree to ~1e-5 relative, 100x inside the 1e-3 bar.
//
// MFMA operand layout (16x16x32): lane l holds A[row l&15][k = 8(l>>4)+j] and B[k = 8(l>>4)+j][col l&15],
// j = 0..7 (one 16-byte register quad each); rows = output channels, cols = 16 pixels, k = 32 input channels.
#include "common.h"
#include "costreg_softmax.h"
#include "kernels.h"

namespace adamvs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct ConvDDArgs16 {
  const float* in;        // [N][hi*wi][D] fp32
  const bf16x8* wpk;      // [2 (hi,lo)][9][D/32][D/16][64] fragments of 8 bf16
  const float* bias;      // [D]
  const float* skip;      // [N][ho*wo][D] or null
  float* out;             // [N][ho*wo][D]
  int D, hi, wi, ho, wo, relu;
  float* sm_vw; float* sm_pd;   // softmax epilogue (stride-1 `prob` layer): see costreg_softmax.h; null = store the scores
  PlaneSrc sm_planes;
  int sm_B;
  int sm_D;               // hypothesis planes of the softmax epilogue when the network runs wider (costreg_width); 0 = D
};

enum { BX_S1 = 0, BX_S2 = 1, BX_T2 = 2 };
constexpr int BX_KB = 32;            // input channels per chunk = one MFMA k-step
constexpr int BX_PIX = 40;           // bf16 per pixel row in LDS: 32 + 8 pad (80 B).  ds_read_b128's four 16-lane groups are
                                     // not contiguous (MI355X_MICROARCH.md, LDS): this pitch is 2-way on the stride-1 layers
                                     // (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE), 96 B is conflict-free -- and not
                                     // faster: the LDS array is 13 % busy either way, the kernel is not bound by it

// Block rows: 8.  (16-row blocks reuse each streamed A fragment twice as often, but their 192 accumulator
// registers leave one wave per SIMD; two waves feeding the matrix pipe from LDS are worth more: 4.0 -> 3.5 ms on
// the full-resolution layers.)  The stride-2 layers stage a 17 x 33 window and keep one wave.
template <int MODE> struct BxGeom;
template <> struct BxGeom<BX_S1> { static constexpr int BR = 8, LR = BR + 2, LC = 18; };
template <> struct BxGeom<BX_S2> { static constexpr int BR = 8, LR = 2 * BR + 1, LC = 33; };
template <> struct BxGeom<BX_T2> { static constexpr int BR = 8, LR = BR + 1, LC = 17; };

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Block = BR rows x 16 columns of output positions x all D channels; waves WM along channels (MT tiles each),
// WN = 4/WM along rows -- the geometry of conv_dd_body in costreg2d.hip with taller blocks.
template <int MT, int WM, int MODE, int PY, int PX, bool SOFTMAX = false>
__device__ __forceinline__ void conv_dd_bx3_body(const ConvDDArgs16& a, __bf16* lds, int n, int by, int bx) {
  using TG = BxGeom<MODE>;
  constexpr int LR = TG::LR, LC = TG::LC, NPIX = LR * LC;
  constexpr int BR = TG::BR, WN = 4 / WM, NTR = BR / WN;
  constexpr int STR = (MODE == BX_S2) ? 2 : 1;
  constexpr int NTY = (MODE == BX_T2) ? 1 + PY : 3;
  constexpr int NTX = (MODE == BX_T2) ? 1 + PX : 3;
  __bf16* lhi = lds;                       // [NPIX][BX_PIX]
  __bf16* llo = lds + NPIX * BX_PIX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int p = lane & 15, q = lane >> 4;
  const int D = a.D, KBT = D / BX_KB, NTILES = D / 16;
  const size_t lo_off = (size_t)9 * KBT * NTILES * 64;       // fragments between the hi and the lo half
  const int r0 = by * BR, c0 = bx * 16;
  const int iy0 = (MODE == BX_T2) ? r0 : r0 * STR - 1;
  const int ix0 = (MODE == BX_T2) ? c0 : c0 * STR - 1;
  const float* inb = a.in + (size_t)n * a.hi * a.wi * D;

  f32x4 acc[MT][NTR];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < NTR; ++r) acc[mt][r] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int NITEMS = NPIX * (BX_KB / 4), NIT = (NITEMS + 255) / 256;     // float4 = 4 channels of a pixel
  auto load_x = [&](f32x4 (&st)[NIT], int ch) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int i = tid + it * 256;
      int g = i % (BX_KB / 4), pp = i / (BX_KB / 4);
      int r = pp / LC, c = pp % LC;
      int iy = iy0 + r, ix = ix0 + c;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < NITEMS && iy >= 0 && iy < a.hi && ix >= 0 && ix < a.wi)
        v = *(const f32x4*)(inb + ((size_t)iy * a.wi + ix) * D + ch + 4 * g);
      st[it] = v;
    }
  };
  auto store_x = [&](const f32x4 (&st)[NIT]) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int i = tid + it * 256;
      if (i < NITEMS) {
        int g = i % (BX_KB / 4), pp = i / (BX_KB / 4);
        f32x4 v = st[it];
        bf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        bf16x4 l = {(__bf16)(v.x - (float)h.x), (__bf16)(v.y - (float)h.y), (__bf16)(v.z - (float)h.z), (__bf16)(v.w - (float)h.w)};
        *(bf16x4*)(lhi + pp * BX_PIX + 4 * g) = h;
        *(bf16x4*)(llo + pp * BX_PIX + 4 * g) = l;
      }
    }
  };
  // A fragments: uniform descriptor + pinned lane offset; the fragment index is the scalar offset operand
  const buf_rsrc rw = make_rsrc(a.wpk);
  unsigned wlane = (unsigned)(lane * 16);
  pin(wlane);
  const unsigned lo_bytes = (unsigned)(lo_off * 16);
  auto load_w = [&](bf16x8 (&wh)[MT], bf16x8 (&wl)[MT], int kb, int t) {
    const int ty = t / NTX, tx = t % NTX;
    const int ky = (MODE == BX_T2) ? (PY ? (ty ? 0 : 2) : 1) : ty;
    const int kx = (MODE == BX_T2) ? (PX ? (tx ? 0 : 2) : 1) : tx;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned f = (unsigned)((((ky * 3 + kx) * KBT + kb) * NTILES + wm * MT + mt) * 1024);       // uniform, bytes
      wh[mt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, f, 0));
      wl[mt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, f + lo_bytes, 0));
    }
  };

  // lane's B-fragment base: pixel column p of the wave's first row, k-group q (8 channels = 16 bytes)
  unsigned boff = (unsigned)((((wn * NTR * STR) * LC + p * STR) * BX_PIX + 8 * q) * 2);     // bytes
  pin(boff);
  constexpr int LOB = NPIX * BX_PIX * 2;              // byte distance hi -> lo image
  constexpr int NTAP = NTY * NTX;
  f32x4 xs[NIT];
  load_x(xs, 0);
  bf16x8 w0h[MT], w0l[MT], w1h[MT], w1l[MT];          // two named fragment sets: tap t uses set t & 1 (no copies)
  auto tap = [&](const bf16x8 (&wh)[MT], const bf16x8 (&wl)[MT], int t) {
    const int ty = t / NTX, tx = t % NTX;
    // B fragments one row ahead of the MFMAs that consume them (LDS latency under the previous row's chain)
    bf16x8 bh[NTR], bl[NTR];
    {
      const char* at = (const char*)lds + boff + (ty * LC + tx) * (BX_PIX * 2);
      bh[0] = *(const bf16x8*)at;
      bl[0] = *(const bf16x8*)(at + LOB);
    }
#pragma unroll
    for (int r = 0; r < NTR; ++r) {
      if (r + 1 < NTR) {
        const char* at = (const char*)lds + boff + (((r + 1) * STR + ty) * LC + tx) * (BX_PIX * 2);
        bh[r + 1] = *(const bf16x8*)at;
        bl[r + 1] = *(const bf16x8*)(at + LOB);
      }
      // the three products of one accumulator are issued MT instructions apart, not back to back
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][r] = mfma_bf16(wh[mt], bh[r], acc[mt][r]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][r] = mfma_bf16(wh[mt], bl[r], acc[mt][r]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][r] = mfma_bf16(wl[mt], bh[r], acc[mt][r]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int kb = 0; kb < KBT; ++kb) {
    __syncthreads();                     // previous chunk's readers are done
    store_x(xs);
    load_w(w0h, w0l, kb, 0);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      // next tap's fragments are requested before this tap's MFMAs; the scheduling barriers keep the taps apart
      // (a free scheduler hoists every LDS fragment of the chunk and spills)
      if (t & 1) {
        if (t + 1 < NTAP) load_w(w0h, w0l, kb, t + 1);
        tap(w1h, w1l, t);
      } else {
        if (t + 1 < NTAP) load_w(w1h, w1l, kb, t + 1);
        // The next chunk's window is requested HERE, behind tap 1's fragments (round 6): vmcnt retires in order, so a window request in
        // front of them made the wait for tap 1's weights a wait for the window too -- one tap after it was issued; now it has two taps
        // to arrive before a wait reaches it (stored at the next chunk's top).  -1.2 % of the kernel (timing builds:
        // profiles/r06_bx3_costreg_timing.txt, which also show what bounds it: matrix time 1.86 ms + everything else 1.64 ms = 3.50).
        if (t == 0 && kb + 1 < KBT) load_x(xs, (kb + 1) * BX_KB);
        tap(w0h, w0l, t);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if (SOFTMAX) {                         // the last layer inside a stage: scores reduced over D here, never stored
    __syncthreads();                     // the last chunk's readers are done: the tile space is reused for the partials
    softmax_epilogue<MT, WM, NTR, BR>(acc, a.bias, a.sm_planes, a.sm_B, n, r0, c0, a.ho, a.wo, a.sm_D ? a.sm_D : D, a.sm_vw, a.sm_pd, (float*)lds);
    return;
  }
  // epilogue (C/D layout of the 16x16 MFMA family is shape-independent: lane owns channels co4..co4+3 of pixel p)
#pragma unroll
  for (int r = 0; r < NTR; ++r) {
    int row = r0 + wn * NTR + r, col = c0 + p;
    int oy = (MODE == BX_T2) ? 2 * row + PY : row;
    int ox = (MODE == BX_T2) ? 2 * col + PX : col;
    bool valid = (MODE == BX_T2) ? (row < a.hi && col < a.wi) : (oy < a.ho && ox < a.wo);
    if (!valid) continue;
    size_t opix = ((size_t)n * a.ho + oy) * a.wo + ox;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int co4 = (wm * MT + mt) * 16 + 4 * q;
      f32x4 v = acc[mt][r] + *(const f32x4*)(a.bias + co4);
      if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (a.skip) v += *(const f32x4*)(a.skip + opix * D + co4);
      *(f32x4*)(a.out + opix * D + co4) = v;
    }
  }
}

template <int MT, int WM, int MODE, bool SM = false>
__global__ __launch_bounds__(256, (MODE == BX_S2 || (MT == 4 && WM == 4)) ? 1 : 2) void k_conv_dd_bx3(ConvDDArgs16 a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];     // [2 (hi,lo)][LR*LC][BX_PIX]
  static_assert(!SM || (MODE == BX_S1 && (size_t)2 * BxGeom<BX_S1>::LR * BxGeom<BX_S1>::LC * BX_PIX * 2 >= (size_t)8 * 16 * 4 * WM * 3 * 4),
                "softmax partials must fit the tile space");
  if (SM) {
    conv_dd_bx3_body<MT, WM, MODE, 0, 0, true>(a, lds, blockIdx.z, blockIdx.y, blockIdx.x);
  } else if (MODE == BX_T2) {
    int n = blockIdx.z >> 2, cls = blockIdx.z & 3;
    switch (cls) {
      case 0: conv_dd_bx3_body<MT, WM, MODE, 0, 0>(a, lds, n, blockIdx.y, blockIdx.x); break;
      case 1: conv_dd_bx3_body<MT, WM, MODE, 0, 1>(a, lds, n, blockIdx.y, blockIdx.x); break;
      case 2: conv_dd_bx3_body<MT, WM, MODE, 1, 0>(a, lds, n, blockIdx.y, blockIdx.x); break;
      default: conv_dd_bx3_body<MT, WM, MODE, 1, 1>(a, lds, n, blockIdx.y, blockIdx.x); break;
    }
  } else {
    conv_dd_bx3_body<MT, WM, MODE, 0, 0>(a, lds, blockIdx.z, blockIdx.y, blockIdx.x);
  }
}

template <int MT, int WM, int MODE, bool SM = false>
static int launch_bx3_mode(const ConvDDArgs16& a, dim3 grid, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * BxGeom<MODE>::LR * BxGeom<MODE>::LC * BX_PIX * sizeof(__bf16);
  auto kern = k_conv_dd_bx3<MT, WM, MODE, SM>;
  static bool attr_set = false;          // idempotent per-kernel attribute (not a stream operation)
  if (lds > 64 * 1024 && !attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_error((int)e, "conv_dd_bf16x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  ADAMVS_CHECK_LAUNCH("conv_dd_bf16x3");
  return 0;
}

template <int MT, int WM>
static int launch_bx3_cfg(const ConvDDArgs16& a, int N, int mode, hipStream_t st) {
  if (mode == BX_S1 && a.sm_vw) return launch_bx3_mode<MT, WM, BX_S1, true>(a, dim3(cdiv(a.wo, 16), cdiv(a.ho, BxGeom<BX_S1>::BR), N), st);
  if (mode == BX_S1) return launch_bx3_mode<MT, WM, BX_S1>(a, dim3(cdiv(a.wo, 16), cdiv(a.ho, BxGeom<BX_S1>::BR), N), st);
  if (mode == BX_S2) return launch_bx3_mode<MT, WM, BX_S2>(a, dim3(cdiv(a.wo, 16), cdiv(a.ho, BxGeom<BX_S2>::BR), N), st);
  return launch_bx3_mode<MT, WM, BX_T2>(a, dim3(cdiv(a.wi, 16), cdiv(a.hi, BxGeom<BX_T2>::BR), N * 4), st);
}

bool costreg_bf16x3_depth_supported(int D) { return D == 32 || D == 64 || D == 96 || D == 128 || D == 192 || D == 256 || D == 384 || D == 512; }

// `wpk` is the layer's packed block reinterpreted: 9*D*D floats worth of bf16 fragments (hi half, then lo half)
int launch_conv_dd_bf16x3(const float* in, const float* wpk, const float* bias, const float* skip, float* out, int N, int D,
                          int hi, int wi, int ho, int wo, int mode, int relu, hipStream_t st, float* sm_vw, float* sm_pd,
                          const PlaneSrc* sm_planes, int sm_B, int n_planes) {
  ConvDDArgs16 a{in, (const bf16x8*)wpk, bias, skip, out, D, hi, wi, ho, wo, relu, sm_vw, sm_pd,
                 sm_planes ? *sm_planes : PlaneSrc{nullptr, 0, 0.f}, sm_B, n_planes};
  constexpr int ZMAX = 65535 / 4;            // blockIdx.z = image (x 4 parity classes in the transposed layers)
  if (N > ZMAX) {                            // sub-batches of whole images (of whole tiles under the softmax epilogue: n % sm_B)
    const int unit = sm_vw ? sm_B : 1;
    ADAMVS_CHECK_ARG(unit <= ZMAX, "conv_dd_bf16x3: softmax epilogue with %d tiles per view (at most %d)", unit, ZMAX);
    const int step = ZMAX / unit * unit;
    const size_t in_img = (size_t)hi * wi * D, out_img = (size_t)ho * wo * D, map = (size_t)ho * wo;
    for (int n0 = 0; n0 < N; n0 += step)
      if (int rc = launch_conv_dd_bf16x3(in + n0 * in_img, wpk, bias, skip ? skip + n0 * out_img : nullptr, out ? out + n0 * out_img : nullptr,
                                         N - n0 < step ? N - n0 : step, D, hi, wi, ho, wo, mode, relu, st, sm_vw ? sm_vw + n0 * map : nullptr,
                                         sm_pd ? sm_pd + n0 * map : nullptr, sm_planes, sm_B, n_planes))
        return rc;
    return 0;
  }
  switch (D) {
    case 384: {                              // the 192-channel tiling twice: each launch contracts all 384 input channels into its
      ADAMVS_CHECK_ARG(!sm_vw, "conv_dd_bf16x3: no softmax epilogue at D=384");      // half of the output channels (fp32 twin: costreg2d.hip)
      for (int half = 0; half < 2; ++half) {
        ConvDDArgs16 h = a;
        h.wpk = a.wpk + (size_t)half * 12 * 64;        // fragment = 64 lanes x 8 bf16; index (... * D/16 + tile)
        h.bias = bias + half * 192;
        h.out = out + half * 192;
        h.skip = skip ? skip + half * 192 : nullptr;
        if (int rc = launch_bx3_cfg<3, 4>(h, N, mode, st)) return rc;
      }
      return 0;
    }
    case 512: {                              // the 256-channel tiling twice
      ADAMVS_CHECK_ARG(!sm_vw, "conv_dd_bf16x3: no softmax epilogue at D=512");
      for (int half = 0; half < 2; ++half) {
        ConvDDArgs16 h = a;
        h.wpk = a.wpk + (size_t)half * 16 * 64;
        h.bias = bias + half * 256;
        h.out = out + half * 256;
        h.skip = skip ? skip + half * 256 : nullptr;
        if (int rc = launch_bx3_cfg<4, 4>(h, N, mode, st)) return rc;
      }
      return 0;
    }
    case 32: return launch_bx3_cfg<2, 1>(a, N, mode, st);
    case 64: return launch_bx3_cfg<4, 1>(a, N, mode, st);
    case 96: return launch_bx3_cfg<3, 2>(a, N, mode, st);
    case 128: return launch_bx3_cfg<4, 2>(a, N, mode, st);
    case 192: return launch_bx3_cfg<3, 4>(a, N, mode, st);
    case 256: return launch_bx3_cfg<4, 4>(a, N, mode, st);      // (the two-launch 128-channel form of the fp32 path does not pay here: 155.0 -> 156.0 ms at cfg5)
  }
  return set_error(-1, "cost_reg_net_2d (bf16x3): D=%d unsupported (32, 64, 96, 128, 192, 256, 384 or 512)", D);
}

}  // namespace adamvs
