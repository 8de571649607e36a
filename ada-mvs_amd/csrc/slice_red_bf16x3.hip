// Split-bf16 ("bf16x3") versions of the small 3x3 convolutions of SliceCostRegNetRED
// (reference models/adamvs.py:400-424, models/module.py:5-52): conv1 (two-row form), the two ConvGRU gate /
// candidate convolutions and conv2.  Same tiles, persistent grids and fused epilogues as slice_red.hip; the
// channel contraction runs on v_mfma_f32_16x16x32_bf16 with every fp32 operand split into hi + lo bf16 halves
// (a.b ~ a_hi.b_hi + a_hi.b_lo + a_lo.b_hi, fp32 accumulation; see costreg2d_bf16x3.hip).
//
// The contraction index is flattened: k = pos * CIN + cin, pos = tap (ky,kx) -- or (input row rr, kx) for the
// two-row conv1 -- padded with zero weights to a multiple of 32.  A lane's k-group (8 consecutive k = 8 channels
// of ONE position, since CIN is a multiple of 8) is one 16-byte LDS read from a pixel-major tile
// [pixel][CIN + 8 pad] of bf16 (hi and lo images); the activations are split when the tile is written.
#include "common.h"
#include "kernels.h"

namespace adamvs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_bx(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void split_store(__bf16* hi, __bf16* lo, f32x4 v) {
  bf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  bf16x4 l = {(__bf16)(v.x - (float)h.x), (__bf16)(v.y - (float)h.y), (__bf16)(v.z - (float)h.z), (__bf16)(v.w - (float)h.w)};
  *(bf16x4*)hi = h;
  *(bf16x4*)lo = l;
}

enum { BXE_RELU = 0, BXE_GATES = 1, BXE_CAND = 2, BXE_TWO_ROW = 3 };

struct SmallConvArgsBx {
  const float* srcA;      // [B][hi*wi][CA]
  const float* srcB;      // [B][hi*wi][CB] (null when CB == 0)
  const bf16x8* wpk;      // A fragments [NT][hi|lo][NKB][64] x 8 bf16
  const float* bias;      // [16*NT] (GATES, CAND)
  float* dst0;            // RELU/TWO_ROW: out [B][ho*wo][cout]; GATES: r*h; CAND: h (in place)
  float* dst1;            // GATES: u out; CAND: u in
  const float* hsrc;      // GATES: h [B][ho*wo][HC] (the centre-pixel state, read in fp32)
  int hi, wi, ho, wo, cout;
};

// grid = resident capacity, workgroup i takes tiles i, i + grid, ...; block 256.
// NPOS = 9 taps, or 12 (rr,kx) positions for the two-row conv1 (then a run is 2 output rows x 16 pixels).
template <int CA, int CB, int NT, int STRIDE, int EPI, int TR>
__global__ __launch_bounds__(256) void k_conv_small_bx3(SmallConvArgsBx a, int tiles_x, int tiles_y, int ntiles) {
  constexpr bool TWO = (EPI == BXE_TWO_ROW);
  constexpr int CIN = CA + CB, G = CIN / 4, HC = CB;
  constexpr int NPOS = TWO ? 12 : 9;
  constexpr int NKB = (NPOS * CIN + 31) / 32;
  constexpr int LR = (STRIDE == 1) ? TR + 2 : 2 * TR + 1;
  constexpr int LC = (STRIDE == 1) ? 34 : 65;
  constexpr int PP = CIN + 8;                         // bf16 per pixel (16-byte pad spreads the banks)
  constexpr int NITEMS = LR * LC * G, NIT = (NITEMS + 255) / 256;
  constexpr int RUNS = TWO ? TR : 2 * TR;             // two-row: (row pair, column half); else (row, column half)
  extern __shared__ __attribute__((aligned(16))) __bf16 ldsb[];     // [hi|lo][LR*LC][PP]
  __bf16* lhi = ldsb;
  __bf16* llo = ldsb + LR * LC * PP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, q = lane >> 4;

  // A fragments (hi, lo) and the lane's LDS offset of every k-block
  bf16x8 wh[NT][NKB], wl[NT][NKB];
  int off[NKB];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      wh[nt][kb] = a.wpk[((nt * 2 + 0) * NKB + kb) * 64 + lane];
      wl[nt][kb] = a.wpk[((nt * 2 + 1) * NKB + kb) * 64 + lane];
    }
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    int kk = 32 * kb + 8 * q;
    int pos = kk / CIN, c0 = kk % CIN;
    if (pos >= NPOS) { pos = 0; c0 = 0; }             // zero-weight padding: any valid address
    int dy = pos / 3, dx = pos % 3;
    off[kb] = (dy * LC + dx) * PP + c0;
  }

  auto load_tile = [&](f32x4 (&stage)[NIT], int t) {
    const int b = t / (tiles_x * tiles_y), ox0 = (t % tiles_x) * 32, oy0 = ((t / tiles_x) % tiles_y) * TR;
    const int ix0 = ox0 * STRIDE - 1, iy0 = oy0 * STRIDE - 1;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int i = tid + it * 256;
      int g = i % G, pp = i / G;
      int r = pp / LC, c = pp % LC;
      int iy = iy0 + r, ix = ix0 + c;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < NITEMS && iy >= 0 && iy < a.hi && ix >= 0 && ix < a.wi) {
        size_t pix = ((size_t)b * a.hi + iy) * a.wi + ix;
        if (4 * g < CA) v = *(const f32x4*)(a.srcA + pix * CA + 4 * g);
        else v = *(const f32x4*)(a.srcB + pix * CB + (4 * g - CA));
      }
      stage[it] = v;
    }
  };

  f32x4 stage[NIT];
  int t = blockIdx.x;
  if (t < ntiles) load_tile(stage, t);
  for (; t < ntiles; t += gridDim.x) {
    __syncthreads();                         // the previous tile's readers are done
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int i = tid + it * 256;
      if (i < NITEMS) {
        int g = i % G, pp = i / G;
        split_store(lhi + pp * PP + 4 * g, llo + pp * PP + 4 * g, stage[it]);
      }
    }
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) load_tile(stage, t + gridDim.x);      // in flight during the MFMAs

    const int b = t / (tiles_x * tiles_y), ox0 = (t % tiles_x) * 32, oy0 = ((t / tiles_x) % tiles_y) * TR;
#pragma unroll 1
    for (int run = wave; run < RUNS; run += 4) {
      const int row = TWO ? (run >> 1) * 2 : (run >> 1), col = (run & 1) * 16;
      // output pixel(s) of this lane: TWO -> lanes q<2 own row `row`, q>=2 own row+1 (channels 4(q&1)..)
      const int oy = oy0 + row + (TWO ? (q >> 1) : 0), ox = ox0 + col + p;
      const bool valid = oy < a.ho && ox < a.wo;
      const size_t opix = ((size_t)b * a.ho + min(oy, a.ho - 1)) * a.wo + min(ox, a.wo - 1);
      f32x4 acc[NT], pre_u[NT], pre_h[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int co4 = nt * 16 + 4 * q;
        if (EPI == BXE_CAND && co4 < HC) {
          pre_u[nt] = *(const f32x4*)(a.dst1 + opix * HC + co4);
          pre_h[nt] = *(const f32x4*)(a.dst0 + opix * HC + co4);
        }
        if (EPI == BXE_GATES && co4 < HC) pre_h[nt] = *(const f32x4*)(a.hsrc + opix * HC + co4);
      }
      const int base = ((row * STRIDE) * LC + (col + p) * STRIDE) * PP;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const bf16x8 bh = *(const bf16x8*)(lhi + base + off[kb]);
        const bf16x8 bl = *(const bf16x8*)(llo + base + off[kb]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[nt] = mfma_bx(wh[nt][kb], bh, acc[nt]);
          acc[nt] = mfma_bx(wh[nt][kb], bl, acc[nt]);
          acc[nt] = mfma_bx(wl[nt][kb], bh, acc[nt]);
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT && valid; ++nt) {
        const int co4 = nt * 16 + 4 * q;
        f32x4 v = acc[nt];
        if (EPI == BXE_RELU) {
          if (co4 < a.cout) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            *(f32x4*)(a.dst0 + opix * a.cout + co4) = v;
          }
        } else if (EPI == BXE_TWO_ROW) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          *(f32x4*)(a.dst0 + opix * 8 + 4 * (q & 1)) = v;
        } else if (EPI == BXE_GATES) {
          v += *(const f32x4*)(a.bias + co4);
          f32x4 sg = {sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)};
          if (co4 < HC) *(f32x4*)(a.dst0 + opix * HC + co4) = sg * pre_h[nt];          // r * h
          else if (co4 < 2 * HC) *(f32x4*)(a.dst1 + opix * HC + (co4 - HC)) = sg;       // u
        } else if (co4 < HC) {                                                          // BXE_CAND
          v += *(const f32x4*)(a.bias + co4);
          f32x4 cnd = {tanh_fast(v.x), tanh_fast(v.y), tanh_fast(v.z), tanh_fast(v.w)};
          f32x4 u4 = pre_u[nt], h4 = pre_h[nt];
          *(f32x4*)(a.dst0 + opix * HC + co4) = u4 * h4 + (1.0f - u4) * cnd;
        }
      }
    }
  }
}

template <typename K>
static int resident_blocks_bx(K kernel, int threads, size_t lds) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) != hipSuccess || n < 1) n = 1;
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  return n * cus;
}

template <int CA, int CB, int NT, int STRIDE, int EPI, int TR>
static int launch_bx(const SmallConvArgsBx& a, int N, hipStream_t st, const char* name) {
  constexpr int LR = (STRIDE == 1) ? TR + 2 : 2 * TR + 1;
  constexpr int LC = (STRIDE == 1) ? 34 : 65;
  constexpr size_t lds = (size_t)2 * LR * LC * (CA + CB + 8) * sizeof(__bf16);
  static_assert(lds <= 64 * 1024, "tile exceeds the default dynamic LDS limit");
  auto kern = k_conv_small_bx3<CA, CB, NT, STRIDE, EPI, TR>;
  static int capacity = 0;              // per instantiation; a pure function of the kernel and the device
  if (!capacity) capacity = resident_blocks_bx(kern, 256, lds);
  const int tiles_x = cdiv(a.wo, 32), tiles_y = cdiv(a.ho, TR);
  const long ntiles = (long)tiles_x * tiles_y * N;
  if (ntiles > 0x7fffffffL) return set_error(-1, "%s: too many tiles", name);
  const int grid = ntiles < capacity ? (int)ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, tiles_x, tiles_y, (int)ntiles);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "%s: %s", name, hipGetErrorString(e));
  return 0;
}

int launch_conv1_bf16x3(const float* cost, const float* w, float* c1, int N, int C, int h, int w_, hipStream_t st) {
  SmallConvArgsBx a{cost, nullptr, (const bf16x8*)w, nullptr, c1, nullptr, nullptr, h, w_, h, w_, 8};
  if (C == 32) return launch_bx<32, 0, 1, 1, BXE_TWO_ROW, 8>(a, N, st, "conv1 (bf16x3)");
  if (C == 16) return launch_bx<16, 0, 1, 1, BXE_TWO_ROW, 8>(a, N, st, "conv1 (bf16x3)");
  if (C == 8) return launch_bx<8, 0, 1, 1, BXE_TWO_ROW, 8>(a, N, st, "conv1 (bf16x3)");
  return set_error(-1, "conv1: C=%d unsupported (8, 16 or 32)", C);
}

// gates1 / cand1 / conv2 / gates2 / cand2 of one recurrent step (the decoder stays on the fp32 path)
int launch_gru_convs_bf16x3(const float* c1, const FuseWeights& fw, const StepBuffers& sb, int B, int h, int w, hipStream_t st) {
  const int h2 = h / 2, w2 = w / 2;
  int rc;
  {
    SmallConvArgsBx g{c1, sb.h1, (const bf16x8*)fw.gates1, fw.gates1_b, sb.rh1, sb.u1, sb.h1, h, w, h, w, 16};
    if ((rc = launch_bx<8, 8, 1, 1, BXE_GATES, 4>(g, B, st, "gates1 (bf16x3)"))) return rc;
    SmallConvArgsBx c{c1, sb.rh1, (const bf16x8*)fw.cand1, fw.cand1_b, sb.h1, sb.u1, nullptr, h, w, h, w, 8};
    if ((rc = launch_bx<8, 8, 1, 1, BXE_CAND, 4>(c, B, st, "cand1 (bf16x3)"))) return rc;
  }
  {
    SmallConvArgsBx a{sb.h1, nullptr, (const bf16x8*)fw.conv2, nullptr, sb.c2, nullptr, nullptr, h, w, h2, w2, 16};
    if ((rc = launch_bx<8, 0, 1, 2, BXE_RELU, 4>(a, B, st, "conv2 (bf16x3)"))) return rc;
  }
  {
    SmallConvArgsBx g{sb.c2, sb.h2, (const bf16x8*)fw.gates2, fw.gates2_b, sb.rh2, sb.u2, sb.h2, h2, w2, h2, w2, 32};
    if ((rc = launch_bx<16, 16, 2, 1, BXE_GATES, 4>(g, B, st, "gates2 (bf16x3)"))) return rc;
    SmallConvArgsBx c{sb.c2, sb.rh2, (const bf16x8*)fw.cand2, fw.cand2_b, sb.h2, sb.u2, nullptr, h2, w2, h2, w2, 16};
    if ((rc = launch_bx<16, 16, 1, 1, BXE_CAND, 4>(c, B, st, "cand2 (bf16x3)"))) return rc;
  }
  return 0;
}

}  // namespace adamvs
