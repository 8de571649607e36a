// Split-bf16 ("bf16x3") versions of the small 3x3 convolutions of SliceCostRegNetRED
// (reference models/adamvs.py:400-424, models/module.py:5-52): conv1 (two-row form), the two ConvGRU gate /
// candidate convolutions and conv2.  Same persistent pipeline and fused epilogues as slice_red.hip; the
// channel contraction runs on v_mfma_f32_16x16x32_bf16 with every fp32 operand split into hi + lo bf16 halves
// (a.b ~ a_hi.b_hi + a_hi.b_lo + a_lo.b_hi, fp32 accumulation; see costreg2d_bf16x3.hip).
//
// The contraction index is flattened: k = pos * CIN + cin, pos = tap (ky,kx) -- or (input row rr, kx) for the
// two-row conv1 -- padded with zero weights to a multiple of 32.  A lane's k-group (8 consecutive k = 8 channels
// of ONE position, since CIN is a multiple of 8) is one 16-byte LDS read from a pixel-major tile
// [pixel][CIN + 8 pad] of bf16 (hi and lo images); the activations are split when the tile is written.
#include "common.h"
#include "kernels.h"
#include "persistent.h"
#include "slice_roles_bx3.h"

namespace adamvs {

template <int CA, int CB, int NT, int STRIDE, int EPI, bool SIN, bool SOUT>
__global__ __launch_bounds__(256) void k_conv_small_bx3(SmallConvArgsBx a, TileGrid tg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  ConvSmallBx3Role<CA, CB, NT, STRIDE, EPI, SIN, SOUT>::run(a, tg, TileRange{0, tg.ntiles}, blockIdx.x, gridDim.x, lds);
}

// fp32 map -> split map (slice_roles_bx3.h): the initial states of adamvs_slice_reg_step (a stage starts from zeros: a memset)
__global__ __launch_bounds__(256) void k_split_map(const float* __restrict__ src, float* __restrict__ dst, long ngroups, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;     // group of 4 channels
  if (i >= ngroups) return;
  const long pix = i / (C / 4);
  const int co4 = (int)(i - pix * (C / 4)) * 4;
  bf16x4 h, l;
  split4(*(const f32x4*)(src + i * 4), h, l);
  char* rec = (char*)dst + pix * (4 * C);
  *(bf16x4*)(rec + 2 * co4) = h;
  *(bf16x4*)(rec + 2 * C + 2 * co4) = l;
}

bool bx3_presplit() { return BX3_PRESPLIT != 0; }

int launch_split_map(const float* src, float* dst, long npix, int C, hipStream_t st) {
  const long ngroups = npix * (C / 4);
  hipLaunchKernelGGL(k_split_map, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, st, src, dst, ngroups, C);
  ADAMVS_CHECK_LAUNCH("split map");
  return 0;
}

__global__ __launch_bounds__(256, 2) void k_gru1_fused_bx3(Gru1Args a, TileGrid tg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  Gru1FusedBx3Role::run(a, tg, TileRange{0, tg.ntiles}, blockIdx.x, gridDim.x, lds);
}

template <int CA, int CB, int NT, int STRIDE, int EPI, bool SIN = false, bool SOUT = false>
static int launch_bx(const SmallConvArgsBx& a, int N, hipStream_t st, const char* name) {
  constexpr int TR = (EPI == BXE_TWO_ROW) ? 8 : 4, TC = 16;
  constexpr int LR = (STRIDE == 1) ? TR + 2 : 2 * TR + 1;
  constexpr int LC = (STRIDE == 1) ? TC + 2 : 2 * TC + 1;
  constexpr size_t lds = (size_t)2 * LR * LC * bx_pixel_pitch(CA + CB) * sizeof(__bf16);
  static_assert(lds <= 64 * 1024, "tile exceeds the default dynamic LDS limit");
  auto kern = k_conv_small_bx3<CA, CB, NT, STRIDE, EPI, SIN, SOUT>;
  static const int capacity = resident_blocks(kern, 256, lds);      // once per instantiation, thread-safely (magic static)
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(a.wo, TC), cdiv(a.ho, TR), N)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, tg);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "%s: %s", name, hipGetErrorString(e));
  return 0;
}

// split_out: c1 as a split map (what the recurrence of this mode reads: slice_roles_bx3.h); otherwise fp32 (adamvs_aggregate_conv1)
int launch_conv1_bf16x3(const float* cost, const float* w, float* c1, int N, int C, int h, int w_, int split_out, hipStream_t st) {
  SmallConvArgsBx a{cost, nullptr, (const bf16x8*)w, nullptr, c1, nullptr, nullptr, h, w_, h, w_, 8};
  if (split_out && BX3_PRESPLIT) {
    if (C == 32) return launch_bx<32, 0, 1, 1, BXE_TWO_ROW, false, true>(a, N, st, "conv1 (bf16x3)");
    if (C == 16) return launch_bx<16, 0, 1, 1, BXE_TWO_ROW, false, true>(a, N, st, "conv1 (bf16x3)");
    if (C == 8) return launch_bx<8, 0, 1, 1, BXE_TWO_ROW, false, true>(a, N, st, "conv1 (bf16x3)");
  }
  if (C == 32) return launch_bx<32, 0, 1, 1, BXE_TWO_ROW>(a, N, st, "conv1 (bf16x3)");
  if (C == 16) return launch_bx<16, 0, 1, 1, BXE_TWO_ROW>(a, N, st, "conv1 (bf16x3)");
  if (C == 8) return launch_bx<8, 0, 1, 1, BXE_TWO_ROW>(a, N, st, "conv1 (bf16x3)");
  return set_error(-1, "conv1: C=%d unsupported (8, 16 or 32)", C);
}

__global__ __launch_bounds__(256, 2) void k_gru2_fused_bx3(Gru2Args a, TileGrid tg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  Gru2FusedBx3Role::run(a, tg, TileRange{0, tg.ntiles}, blockIdx.x, gridDim.x, lds);
}

static int launch_gru2_fused(const Gru2Args& a, int B, hipStream_t st) {
  constexpr size_t lds = Gru2FusedBx3Role::LDS_BYTES;
  static const int capacity = resident_blocks(k_gru2_fused_bx3, 256, lds);      // once per instantiation, thread-safely (magic static)
  TileGrid tg;
  if (int rc = make_tile_grid(tg, Gru2FusedBx3Role::tiles_x(a), Gru2FusedBx3Role::tiles_y(a), B)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(k_gru2_fused_bx3, dim3(grid), dim3(256), lds, st, a, tg);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "gru2 fused (bf16x3): %s", hipGetErrorString(e));
  return 0;
}

static int launch_gru1_fused(const Gru1Args& a, int B, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * 12 * 34 * 32 + 10 * 32 * 32;
  static const int capacity = resident_blocks(k_gru1_fused_bx3, 256, lds);      // once per instantiation, thread-safely (magic static)
  TileGrid tg;
  if (int rc = make_tile_grid(tg, cdiv(a.w, 30), cdiv(a.h, 8), B)) return rc;
  const int grid = tg.ntiles < capacity ? tg.ntiles : capacity;
  hipLaunchKernelGGL(k_gru1_fused_bx3, dim3(grid), dim3(256), lds, st, a, tg);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "gru1 fused (bf16x3): %s", hipGetErrorString(e));
  return 0;
}

// GRU level 1 (fused) / conv2 / GRU level 2 (fused) of one recurrent step (the decoder stays on the fp32 path).
// *h1_now, *h2_now receive the buffers that hold the states after the step.
// BX3_PRESPLIT: c1 is a split map; the fp32 states stay in sb.h1 / sb.h2 (updated in place), their split twins alternate between
// (u1, rh1) / (u2, rh2): step d reads the one step d - 1 wrote -- u for even d, so u1 / u2 hold the split initial states before step 0.
int launch_gru_convs_bf16x3(const float* c1, const FuseWeights& fw, const StepBuffers& sb, int B, int h, int w, int d,
                            float** h1_now, float** h2_now, hipStream_t st) {
  const int h2 = h / 2, w2 = w / 2;
  int rc;
  if (BX3_PRESPLIT) {
    float* s1in = (d & 1) ? sb.rh1 : sb.u1;
    float* s1out = (d & 1) ? sb.u1 : sb.rh1;
    Gru1Args g{c1, sb.h1, sb.h1, (const bf16x8*)fw.gates1, fw.gates1_b, (const bf16x8*)fw.cand1, fw.cand1_b, h, w, s1in, s1out};
    if ((rc = launch_gru1_fused(g, B, st))) return rc;
    *h1_now = sb.h1;
    SmallConvArgsBx a{s1out, nullptr, (const bf16x8*)fw.conv2, nullptr, sb.c2, nullptr, nullptr, h, w, h2, w2, 16};
    if ((rc = launch_bx<8, 0, 1, 2, BXE_RELU, true, true>(a, B, st, "conv2 (bf16x3)"))) return rc;
    float* s2in = (d & 1) ? sb.rh2 : sb.u2;
    float* s2out = (d & 1) ? sb.u2 : sb.rh2;
    Gru2Args g2{sb.c2, sb.h2, sb.h2, (const bf16x8*)fw.gates2, fw.gates2_b, (const bf16x8*)fw.cand2, fw.cand2_b, h2, w2, s2in, s2out};
    if ((rc = launch_gru2_fused(g2, B, st))) return rc;
    *h2_now = sb.h2;
    return 0;
  }
  // level 1 fused; its state alternates between the h1 and rh1 buffers (step d reads the one step d-1 wrote)
  float* hin = (d & 1) ? sb.rh1 : sb.h1;
  float* hout = (d & 1) ? sb.h1 : sb.rh1;
  {
    Gru1Args g{c1, hin, hout, (const bf16x8*)fw.gates1, fw.gates1_b, (const bf16x8*)fw.cand1, fw.cand1_b, h, w, nullptr, nullptr};
    if ((rc = launch_gru1_fused(g, B, st))) return rc;
  }
  *h1_now = hout;
  {
    SmallConvArgsBx a{hout, nullptr, (const bf16x8*)fw.conv2, nullptr, sb.c2, nullptr, nullptr, h, w, h2, w2, 16};
    if ((rc = launch_bx<8, 0, 1, 2, BXE_RELU>(a, B, st, "conv2 (bf16x3)"))) return rc;
  }
  {  // level 2 fused; its state alternates between the h2 and rh2 buffers like level 1's
    float* gin = (d & 1) ? sb.rh2 : sb.h2;
    float* gout = (d & 1) ? sb.h2 : sb.rh2;
    Gru2Args g{sb.c2, gin, gout, (const bf16x8*)fw.gates2, fw.gates2_b, (const bf16x8*)fw.cand2, fw.cand2_b, h2, w2, nullptr, nullptr};
    if ((rc = launch_gru2_fused(g, B, st))) return rc;
    *h2_now = gout;
  }
  return 0;
}

}  // namespace adamvs
