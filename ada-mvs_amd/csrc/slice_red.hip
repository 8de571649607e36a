// Recurrent cost regularisation (one hypothesis slice per step) + soft-argmin.
//
//   SliceCostRegNetRED.forward   reference models/adamvs.py:415-424
//   ConvGRUCell.forward          reference models/module.py:24-52
//   online soft-argmin           reference models/adamvs.py:516-531
//
// Every map is channel-last [B][h*w][ch].  The 3x3 convolutions run on the
// fp32 matrix cores (conv_frag.h); gates / candidate / blend are fused into the
// epilogues, the two decoder layers are one kernel.
#include "common.h"
#include "conv_frag.h"
#include "kernels.h"

namespace adamvs {

enum { EPI_RELU = 0, EPI_GATES = 1, EPI_CAND = 2 };

struct SmallConvArgs {
  const float* srcA;   // [B][hi*wi][CA]
  const float* srcB;   // [B][hi*wi][CB] (null when CB == 0)
  const float* wpk;    // A fragments [NT][9][(CA+CB)/4][64]
  const float* bias;   // [16*NT], zero padded (GATES, CAND)
  float* dst0;         // RELU: out [B][ho*wo][cout];  GATES: r*h [B][..][HC];  CAND: h, updated in place
  float* dst1;         // GATES: u out [B][..][HC];    CAND: u in
  int hi, wi, ho, wo, cout;
};


// Persistent workgroups.  These launches are short (a few thousand tiles of a few microseconds) and sit on the
// sequential critical path of the recurrence, so what costs time is not the matrix work but its packaging: a grid
// that is 1.8x the resident capacity runs as two "rounds" with the second one mostly empty, and every workgroup
// of a round is in its load phase at the same time.  Here the grid is exactly the resident capacity (occupancy
// query); workgroup i walks tiles i, i + grid, ..., requests the input tile of its NEXT tile before the MFMAs of
// the current one (two LDS buffers, one barrier per tile) and fetches the A fragments once.  (A shared atomic
// tile counter was tried and is slower: one word serves ~88 dequeues/us.)  Tile t -> (tile_x, tile_y, b).
template <int CA, int CB, int NT, int STRIDE, int EPI, int TR>
__global__ __launch_bounds__(256) void k_conv_small(SmallConvArgs a, int tiles_x, int tiles_y, int ntiles) {
  constexpr int CIN = CA + CB, KC = CIN / 4, G = CIN / 4;
  constexpr int LR = (STRIDE == 1) ? TR + 2 : 2 * TR + 1;
  constexpr int LC = (STRIDE == 1) ? 34 : 65;
  constexpr int PLANE = (STRIDE == 1) ? plane_pitch16(LR * LC) : ((LR * LC) | 1);
  constexpr int GP = group_pitch(PLANE, G);
  constexpr int HC = CB;                 // hidden width for the GRU epilogues
  constexpr int NITEMS = LR * LC * G, NIT = (NITEMS + 255) / 256;
  extern __shared__ float lds_all[];     // [2][G][GP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, q = lane >> 4;

  float wf[NT][9][KC];
  load_wfrag<NT, KC>(wf, a.wpk, lane);

  auto tile_coords = [&](int t, int& b, int& ox0, int& oy0) {
    int tx = t % tiles_x, r = t / tiles_x;
    int ty = r % tiles_y;
    b = r / tiles_y;
    ox0 = tx * 32; oy0 = ty * TR;
  };
  auto load_tile = [&](f32x4 (&stage)[NIT], int t) {
    int b, ox0, oy0;
    tile_coords(t, b, ox0, oy0);
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
  auto store_tile = [&](const f32x4 (&stage)[NIT], float* lds) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int i = tid + it * 256;
      if (i < NITEMS) {
        int g = i % G, pp = i / G;
        int r = pp / LC, c = pp % LC;
        float* dl = lds + g * GP + r * LC + c;
        f32x4 v = stage[it];
        dl[0] = v.x; dl[PLANE] = v.y; dl[2 * PLANE] = v.z; dl[3 * PLANE] = v.w;
      }
    }
  };

  int t = blockIdx.x;
  f32x4 stage[NIT];
  if (t < ntiles) load_tile(stage, t);
  for (int it = 0; t < ntiles; ++it) {
    float* lds = lds_all + (it & 1) * (G * GP);
    store_tile(stage, lds);
    __syncthreads();        // tile visible; readers of this LDS buffer two iterations ago are done
    const int tn = t + gridDim.x;
    if (tn < ntiles) load_tile(stage, tn);            // in flight during the MFMAs below

    int b, ox0, oy0;
    tile_coords(t, b, ox0, oy0);
    const float* xb = lds + q * PLANE + p * STRIDE;
    // A wave owns (2 TR)/4 runs of 16 pixels, one after the other (rolled: occupancy matters more than unrolling
    // here).  Operands the epilogue needs from global memory are requested before the run's MFMA chain.
#pragma unroll 1
    for (int run = wave; run < TR * 2; run += 4) {
      const int row = run >> 1, col = (run & 1) * 16;
      const int oy = oy0 + row, ox = ox0 + col + p;
      const bool valid = oy < a.ho && ox < a.wo;
      const size_t opix = ((size_t)b * a.ho + min(oy, a.ho - 1)) * a.wo + min(ox, a.wo - 1);
      f32x4 acc[NT], pre_u[NT], pre_h[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int co4 = nt * 16 + 4 * q;
        if (EPI == EPI_CAND && co4 < HC) {
          pre_u[nt] = *(const f32x4*)(a.dst1 + opix * HC + co4);
          pre_h[nt] = *(const f32x4*)(a.dst0 + opix * HC + co4);
        }
      }
      conv3x3_run<NT, KC, STRIDE, GP, LC>(acc, wf, xb, row, col);
#pragma unroll
      for (int nt = 0; nt < NT && valid; ++nt) {
        int co4 = nt * 16 + 4 * q;
        f32x4 v = acc[nt];
        if (EPI == EPI_RELU) {
          if (co4 < a.cout) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            *(f32x4*)(a.dst0 + opix * a.cout + co4) = v;
          }
        } else if (EPI == EPI_GATES) {
          f32x4 bb = *(const f32x4*)(a.bias + co4);
          v += bb;
          f32x4 sg = {sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)};
          if (co4 < HC) {                                   // reset gate -> r * h   (module.py:35-41)
            const float* hl = lds + ((CA + co4) >> 2) * GP + (row + 1) * LC + col + p + 1;
            f32x4 h4 = {hl[0], hl[PLANE], hl[2 * PLANE], hl[3 * PLANE]};
            *(f32x4*)(a.dst0 + opix * HC + co4) = sg * h4;
          } else if (co4 < 2 * HC) {                        // update gate
            *(f32x4*)(a.dst1 + opix * HC + (co4 - HC)) = sg;
          }
        } else {                                            // EPI_CAND   (module.py:44-50)
          if (co4 < HC) {
            f32x4 bb = *(const f32x4*)(a.bias + co4);
            v += bb;
            f32x4 cnd = {tanh_fast(v.x), tanh_fast(v.y), tanh_fast(v.z), tanh_fast(v.w)};
            f32x4 u4 = pre_u[nt], h4 = pre_h[nt];
            *(f32x4*)(a.dst0 + opix * HC + co4) = u4 * h4 + (1.0f - u4) * cnd;
          }
        }
      }
    }
    t = tn;
  }
}

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
  const float* wfin;   // [72] index c*9 + ky*3 + kx, then bias at [72]
  float* vol;          // [B][D][Ho*Wo]
  int h, w, D, d;
};

// ConvTranspose2d(k3, s2, p1, op1) restricted to one output parity class (PY,PX):
// out[2i+PY][2j+PX] = sum over taps with ky = 2(i-iy)+PY+1, i.e. PY=0 -> (ky=1, iy=i);
// PY=1 -> (ky=2, iy=i), (ky=0, iy=i+1); same along x.
template <int PY, int PX, int HGP, int HCOLS>
__device__ __forceinline__ f32x4 upconv1_class(const float (&wf)[1][9][4], const float* xb) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ty = 0; ty < (PY ? 2 : 1); ++ty) {
    const int ky = PY ? (ty ? 0 : 2) : 1;
#pragma unroll
    for (int tx = 0; tx < (PX ? 2 : 1); ++tx) {
      const int kx = PX ? (tx ? 0 : 2) : 1;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        acc = mfma16(wf[0][ky * 3 + kx][kc], xb[kc * HGP + ty * HCOLS + tx], acc);
    }
  }
  return acc;
}

// Persistent like k_conv_small: grid = resident capacity, workgroup i takes tiles i, i + grid, ...
template <bool IN_UP>
__global__ __launch_bounds__(256) void k_decoder(DecoderArgs a, int tiles_x, int tiles_y, int ntiles) {
  constexpr int HR = 10, HCOLS = 18, HPLANE = plane_pitch16(HR * HCOLS);   // h2 region, 16 planes in 4 groups
  constexpr int HGP = group_pitch(HPLANE, 4);
  constexpr int SR = 16, SC = 32, SPLANE = plane_pitch16(SR * SC);        // s region, 8 planes
  __shared__ float lds[4 * HGP + 8 * SPLANE];
  float* lh2 = lds;
  float* ls = lds + 4 * HGP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = a.h, w = a.w, h2 = h >> 1, w2 = w >> 1;
  float wf[1][9][4];
  load_wfrag<1, 4>(wf, a.wup1, lane);

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int b = tile / (tiles_x * tiles_y);
  const int x0 = (tile % tiles_x) * 30, y0 = ((tile / tiles_x) % tiles_y) * 14;     // both even
  const int i0 = (y0 >> 1) - 1, j0 = (x0 >> 1) - 1;         // h2-region origin

  constexpr int NITEMS = HR * HCOLS * 4, NIT = (NITEMS + 255) / 256;
  f32x4 stage[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    int i = tid + it * 256;
    int g = i & 3, pp = i >> 2;
    int r = pp / HCOLS, c = pp % HCOLS;
    int iy = i0 + r, ix = j0 + c;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < NITEMS && iy >= 0 && iy < h2 && ix >= 0 && ix < w2)
      v = *(const f32x4*)(a.h2 + (((size_t)b * h2 + iy) * w2 + ix) * 16 + 4 * g);
    stage[it] = v;
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    int i = tid + it * 256;
    if (i < NITEMS) {
      int g = i & 3, pp = i >> 2;
      int r = pp / HCOLS, c = pp % HCOLS;
      float* dl = lh2 + g * HGP + r * HCOLS + c;
      f32x4 v = stage[it];
      dl[0] = v.x; dl[HPLANE] = v.y; dl[2 * HPLANE] = v.z; dl[3 * HPLANE] = v.w;
    }
  }
  __syncthreads();

  // --- upconv1 on the matrix cores: run = (parity class, row of that class), 32 runs, 8 per wave (rolled).
  const int p = lane & 15, q = lane >> 4;
  const f32x4 bup = *(const f32x4*)(a.bup1 + 4 * q);
#pragma unroll 1
  for (int run = wave; run < 32; run += 4) {
    const int cls = run >> 3, k = run & 7;          // k-th row of this class; wave-uniform
    const int py = cls >> 1, px = cls & 1;
    const int r = py ? 2 * k : 2 * k + 1;            // s-region row / col of the lane's pixel
    const int c = px ? 2 * p : 2 * p + 1;
    const int li = py ? k : k + 1, lj = px ? p : p + 1;
    const int ys = y0 - 1 + r, xs = x0 - 1 + c;
    const bool sin = ys >= 0 && ys < h && xs >= 0 && xs < w;
    f32x4 h1v = {0.f, 0.f, 0.f, 0.f};               // skip operand: in flight during the MFMAs
    if (q < 2 && sin) h1v = *(const f32x4*)(a.h1 + (((size_t)b * h + ys) * w + xs) * 8 + 4 * q);
    const float* xb = lh2 + q * HPLANE + li * HCOLS + lj;
    f32x4 acc;
    switch (cls) {
      case 0: acc = upconv1_class<0, 0, HGP, HCOLS>(wf, xb); break;
      case 1: acc = upconv1_class<0, 1, HGP, HCOLS>(wf, xb); break;
      case 2: acc = upconv1_class<1, 0, HGP, HCOLS>(wf, xb); break;
      default: acc = upconv1_class<1, 1, HGP, HCOLS>(wf, xb); break;
    }
    if (q < 2) {
      f32x4 sv = {0.f, 0.f, 0.f, 0.f};
      if (sin) {
        sv = acc + bup + h1v;                                         // adamvs.py:420-421
        sv.x = fmaxf(sv.x, 0.f); sv.y = fmaxf(sv.y, 0.f); sv.z = fmaxf(sv.z, 0.f); sv.w = fmaxf(sv.w, 0.f);
      }
      float* dl = ls + (4 * q) * SPLANE + r * SC + c;
      dl[0] = sv.x; dl[SPLANE] = sv.y; dl[2 * SPLANE] = sv.z; dl[3 * SPLANE] = sv.w;
    }
  }
  __syncthreads();

  // --- last layer (8 -> 1) on the vector units
  cfloat* wfin = as_const(a.wfin);
  const float bfin = wfin[72];
  if (IN_UP) {
    const int Ho = 2 * h, Wo = 2 * w;
    float* out = a.vol + ((size_t)b * a.D + a.d) * (size_t)Ho * Wo;
    for (int i = tid; i < 28 * 60; i += 256) {
      int ry = i / 60, rx = i % 60;
      int Y = 2 * y0 + ry, X = 2 * x0 + rx;
      if (Y >= Ho || X >= Wo) continue;
      // Y = 2 iy - 1 + ky: Y even -> (ky=1, iy=Y/2); Y odd -> (ky=2, iy=(Y-1)/2), (ky=0, iy=(Y+1)/2)
      int nyt = (Y & 1) ? 2 : 1, nxt = (X & 1) ? 2 : 1;
      float accv = bfin;
      for (int ty = 0; ty < nyt; ++ty) {
        int ky = (Y & 1) ? (ty ? 0 : 2) : 1;
        int iy = (Y + 1 - ky) >> 1;
        int r = iy - (y0 - 1);
        for (int tx = 0; tx < nxt; ++tx) {
          int kx = (X & 1) ? (tx ? 0 : 2) : 1;
          int ix = (X + 1 - kx) >> 1;
          int c = ix - (x0 - 1);
          const float* sp = ls + r * SC + c;
#pragma unroll
          for (int ch = 0; ch < 8; ++ch) accv += sp[ch * SPLANE] * wfin[ch * 9 + ky * 3 + kx];
        }
      }
      out[(size_t)Y * Wo + X] = accv;
    }
  } else {
    float* out = a.vol + ((size_t)b * a.D + a.d) * (size_t)h * w;
    for (int i = tid; i < 14 * 30; i += 256) {
      int ry = i / 30, rx = i % 30;
      int y = y0 + ry, x = x0 + rx;
      if (y >= h || x >= w) continue;
      float accv = bfin;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* sp = ls + (ry + ky) * SC + rx + kx;
#pragma unroll
          for (int ch = 0; ch < 8; ++ch) accv += sp[ch * SPLANE] * wfin[ch * 9 + ky * 3 + kx];
        }
      out[(size_t)y * w + x] = accv;
    }
  }
  __syncthreads();     // the next tile reuses both LDS regions
  }
}

// ---------------------------------------------------------------------------
// Soft-argmin over the regularised cost slices (adamvs.py:516-531):
// p = exp(cost) (no max subtraction), E = sum p, M = max p (initial 0), A = sum depth_d p,
// depth = A / (E + 1e-10), confidence = M / (E + 1e-10).  When IN_UP the hypothesis plane
// is the 2x bilinear upsample (align_corners=False) of planes[b][d] (adamvs.py:521-522).
template <bool IN_UP>
__global__ void k_soft_argmin(const float* __restrict__ vol, const float* __restrict__ planes, float* __restrict__ depth,
                              float* __restrict__ conf, int D, int h, int w, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int Ho = IN_UP ? 2 * h : h, Wo = IN_UP ? 2 * w : w;
  int X = (int)(i % Wo), Y = (int)((i / Wo) % Ho);
  size_t b = i / ((size_t)Wo * Ho);
  size_t hw = (size_t)h * w, HW = (size_t)Ho * Wo;
  int y0 = Y, y1 = Y, x0 = X, x1 = X; float ly = 0.f, lx = 0.f;
  if (IN_UP) {
    float sy = fmaxf(((float)Y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf(((float)X + 0.5f) * 0.5f - 0.5f, 0.f);
    y0 = (int)sy; x0 = (int)sx;
    y1 = y0 + (y0 < h - 1 ? 1 : 0); x1 = x0 + (x0 < w - 1 ? 1 : 0);
    ly = sy - (float)y0; lx = sx - (float)x0;
  }
  const float* v = vol + b * D * HW + (size_t)Y * Wo + X;
  const float* pl = planes + b * D * hw;
  float E = 0.f, M = 0.f, A = 0.f;
  for (int d = 0; d < D; ++d) {
    float pr = __expf(v[(size_t)d * HW]);
    const float* q = pl + (size_t)d * hw;
    float dep;
    if (IN_UP) {
      float top = q[y0 * w + x0] * (1.f - lx) + q[y0 * w + x1] * lx;
      float bot = q[y1 * w + x0] * (1.f - lx) + q[y1 * w + x1] * lx;
      dep = top * (1.f - ly) + bot * ly;
    } else {
      dep = q[y0 * w + x0];
    }
    M = (M < pr) ? pr : M;
    A = dep * pr + A;
    E = E + pr;
  }
  float den = E + 1e-10f;
  depth[i] = A / den;
  conf[i] = M / den;
}

// ---------------------------------------------------------------------------
// host-side launchers shared by the op-level entry point and the stage driver

// workgroups of `kernel` that stay resident per CU (occupancy query, cached per instantiation by the caller)
template <typename K>
static int resident_blocks(K kernel, int threads, size_t lds) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) != hipSuccess || n < 1) n = 1;
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  return n * cus;
}

template <int CA, int CB, int NT, int STRIDE, int EPI, int TR>
static int launch_small_tr(const SmallConvArgs& a, int B, hipStream_t st, const char* name) {
  constexpr int LR = (STRIDE == 1) ? TR + 2 : 2 * TR + 1;
  constexpr int LC = (STRIDE == 1) ? 34 : 65;
  constexpr int PLANE = (STRIDE == 1) ? plane_pitch16(LR * LC) : ((LR * LC) | 1);
  constexpr size_t lds = (size_t)2 * ((CA + CB) / 4) * group_pitch(PLANE, (CA + CB) / 4) * sizeof(float);
  static_assert(lds <= 64 * 1024, "double-buffered tile exceeds the default dynamic LDS limit");
  auto kern = k_conv_small<CA, CB, NT, STRIDE, EPI, TR>;
  static int capacity = 0;              // per instantiation; a pure function of the kernel and the device
  if (!capacity) capacity = resident_blocks(kern, 256, lds);
  const int tiles_x = cdiv(a.wo, 32), tiles_y = cdiv(a.ho, TR);
  const int ntiles = tiles_x * tiles_y * B;
  const int grid = ntiles < capacity ? ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, tiles_x, tiles_y, ntiles);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "%s: %s", name, hipGetErrorString(e));
  return 0;
}

// Tile height 4 (two 16-pixel runs per wave): short tiles keep the end-of-launch imbalance small.
template <int CA, int CB, int NT, int STRIDE, int EPI>
static int launch_small(const SmallConvArgs& a, int B, hipStream_t st, const char* name) {
  if ((long)cdiv(a.ho, 4) * cdiv(a.wo, 32) * B >= 2048) return launch_small_tr<CA, CB, NT, STRIDE, EPI, 4>(a, B, st, name);
  return launch_small_tr<CA, CB, NT, STRIDE, EPI, 2>(a, B, st, name);
}

// conv1 (C -> 8, ReLU; reference adamvs.py:416) in two-row form: the 16 MFMA rows are 8 output channels of
// output row y and the same 8 channels of row y+1, fed by the same input-row fragment (tap ky for the first
// half, ky-1 for the second), so no half of the tile is zero padding: 12 fragment passes per 2 rows instead of 18.
// src [N][hw][C] -> c1 [N][hw][8].  grid (ceil(w/32), ceil(h/8), N); block 256; tile 8 rows x 32 columns.
template <int C>
__global__ __launch_bounds__(256) void k_conv1_two_row(const float* __restrict__ src, const float* __restrict__ wpk,
                                                       float* __restrict__ c1, int h, int w, int tiles_x, int tiles_y,
                                                       int ntiles) {
  constexpr int KC = C / 4, G = C / 4, TR = 8, LR = TR + 2, LC = 34;
  constexpr int PLANE = plane_pitch16(LR * LC), GP = group_pitch(PLANE, G);
  constexpr int NITEMS = LR * LC * G, NIT = (NITEMS + 255) / 256;
  extern __shared__ float lds[];           // [G][GP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, q = lane >> 4;
  float wf[12][KC];
#pragma unroll
  for (int t = 0; t < 12; ++t)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) wf[t][kc] = wpk[(t * KC + kc) * 64 + lane];

  auto load_tile = [&](f32x4 (&stage)[NIT], int t) {
    const int n = t / (tiles_x * tiles_y), x0 = (t % tiles_x) * 32, y0 = ((t / tiles_x) % tiles_y) * TR;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int i = tid + it * 256;
      int g = i % G, pp = i / G;
      int r = pp / LC, c = pp % LC;
      int iy = y0 - 1 + r, ix = x0 - 1 + c;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < NITEMS && iy >= 0 && iy < h && ix >= 0 && ix < w) v = *(const f32x4*)(src + (((size_t)n * h + iy) * w + ix) * C + 4 * g);
      stage[it] = v;
    }
  };

  // persistent: workgroup i takes tiles i, i + grid, ...; the next tile is in flight (registers) during the MFMAs
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
        int r = pp / LC, c = pp % LC;
        float* dl = lds + g * GP + r * LC + c;
        f32x4 v = stage[it];
        dl[0] = v.x; dl[PLANE] = v.y; dl[2 * PLANE] = v.z; dl[3 * PLANE] = v.w;
      }
    }
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) load_tile(stage, t + gridDim.x);

    const int n = t / (tiles_x * tiles_y), x0 = (t % tiles_x) * 32, y0 = ((t / tiles_x) % tiles_y) * TR;
    const float* xb = lds + q * PLANE + p;
    // a wave owns two runs (row pair, column half); their accumulators are independent, so their MFMAs alternate
    const int row0 = (wave >> 1) * 2, col0 = (wave & 1) * 16;        // run = wave
    const int row1 = row0 + 4;                                         // run = wave + 4
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          float b0 = xb[kc * GP + (row0 + rr) * LC + col0 + kx];
          float b1 = xb[kc * GP + (row1 + rr) * LC + col0 + kx];
          acc0 = mfma16(wf[rr * 3 + kx][kc], b0, acc0);
          acc1 = mfma16(wf[rr * 3 + kx][kc], b1, acc1);
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 acc = j ? acc1 : acc0;
      const int y = y0 + (j ? row1 : row0) + (q >> 1), x = x0 + col0 + p;
      if (y < h && x < w) {
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        *(f32x4*)(c1 + (((size_t)n * h + y) * w + x) * 8 + 4 * (q & 1)) = acc;
      }
    }
  }
}

template <int C>
static int launch_conv1_c(const float* cost, const float* w, float* c1, int N, int h, int w_, hipStream_t st) {
  constexpr size_t lds = (size_t)(C / 4) * group_pitch(plane_pitch16(10 * 34), C / 4) * sizeof(float);
  auto kern = k_conv1_two_row<C>;
  static int capacity = 0;
  if (!capacity) capacity = resident_blocks(kern, 256, lds);
  const int tiles_x = cdiv(w_, 32), tiles_y = cdiv(h, 8);
  const long ntiles = (long)tiles_x * tiles_y * N;
  if (ntiles > 0x7fffffffL) return set_error(-1, "conv1: too many tiles");
  const int grid = ntiles < capacity ? (int)ntiles : capacity;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, cost, w, c1, h, w_, tiles_x, tiles_y, (int)ntiles);
  ADAMVS_CHECK_LAUNCH("conv1");
  return 0;
}

int launch_conv1(const float* cost, const float* w, float* c1, int N, int C, int h, int w_, int precision, hipStream_t st) {
  if (precision == PRECISION_BF16X3) return launch_conv1_bf16x3(cost, w, c1, N, C, h, w_, st);
  if (C == 32) return launch_conv1_c<32>(cost, w, c1, N, h, w_, st);
  if (C == 16) return launch_conv1_c<16>(cost, w, c1, N, h, w_, st);
  if (C == 8) return launch_conv1_c<8>(cost, w, c1, N, h, w_, st);
  return set_error(-1, "conv1: C=%d unsupported (8, 16 or 32)", C);
}

// One recurrent step after conv1: c1 -> GRU1 -> conv2 -> GRU2 -> decoder -> vol[:, d].
int launch_slice_step(const float* c1, const FuseWeights& fw, const StepBuffers& sb, float* vol, int B, int h, int w, int D,
                      int d, int in_up, int precision, hipStream_t st) {
  int h2 = h / 2, w2 = w / 2, rc;
  if (precision == PRECISION_BF16X3) {
    if ((rc = launch_gru_convs_bf16x3(c1, fw, sb, B, h, w, st))) return rc;
  } else {
  {  // GRU level 1: gates on cat(c1, h1), candidate on cat(c1, r*h1)
    SmallConvArgs g{c1, sb.h1, fw.gates1, fw.gates1_b, sb.rh1, sb.u1, h, w, h, w, 16};
    if ((rc = launch_small<8, 8, 1, 1, EPI_GATES>(g, B, st, "gates1"))) return rc;
    SmallConvArgs c{c1, sb.rh1, fw.cand1, fw.cand1_b, sb.h1, sb.u1, h, w, h, w, 8};
    if ((rc = launch_small<8, 8, 1, 1, EPI_CAND>(c, B, st, "cand1"))) return rc;
  }
  {  // conv2: 8 -> 16, stride 2, ReLU
    SmallConvArgs a{sb.h1, nullptr, fw.conv2, nullptr, sb.c2, nullptr, h, w, h2, w2, 16};
    if ((rc = launch_small<8, 0, 1, 2, EPI_RELU>(a, B, st, "conv2"))) return rc;
  }
  {  // GRU level 2
    SmallConvArgs g{sb.c2, sb.h2, fw.gates2, fw.gates2_b, sb.rh2, sb.u2, h2, w2, h2, w2, 32};
    if ((rc = launch_small<16, 16, 2, 1, EPI_GATES>(g, B, st, "gates2"))) return rc;
    SmallConvArgs c{sb.c2, sb.rh2, fw.cand2, fw.cand2_b, sb.h2, sb.u2, h2, w2, h2, w2, 16};
    if ((rc = launch_small<16, 16, 1, 1, EPI_CAND>(c, B, st, "cand2"))) return rc;
  }
  }
  DecoderArgs da{sb.h2, sb.h1, fw.upconv1, fw.upconv1_b, fw.final_w, vol, h, w, D, d};
  const int tiles_x = cdiv(w, 30), tiles_y = cdiv(h, 14), ntiles = tiles_x * tiles_y * B;
  static int cap_up = 0, cap_flat = 0;          // resident capacity per instantiation (pure function of kernel + device)
  if (in_up) {
    if (!cap_up) cap_up = resident_blocks(k_decoder<true>, 256, 0);
    hipLaunchKernelGGL((k_decoder<true>), dim3(ntiles < cap_up ? ntiles : cap_up), dim3(256), 0, st, da, tiles_x, tiles_y, ntiles);
  } else {
    if (!cap_flat) cap_flat = resident_blocks(k_decoder<false>, 256, 0);
    hipLaunchKernelGGL((k_decoder<false>), dim3(ntiles < cap_flat ? ntiles : cap_flat), dim3(256), 0, st, da, tiles_x, tiles_y, ntiles);
  }
  ADAMVS_CHECK_LAUNCH("decoder");
  return 0;
}

int launch_soft_argmin(const float* vol, const float* planes, float* depth, float* conf, int B, int D, int h, int w,
                       int in_up, hipStream_t st) {
  size_t total = (size_t)B * (in_up ? 4 : 1) * h * w;
  unsigned nb = (unsigned)((total + 255) / 256);
  if (in_up) hipLaunchKernelGGL((k_soft_argmin<true>), dim3(nb), dim3(256), 0, st, vol, planes, depth, conf, D, h, w, total);
  else hipLaunchKernelGGL((k_soft_argmin<false>), dim3(nb), dim3(256), 0, st, vol, planes, depth, conf, D, h, w, total);
  ADAMVS_CHECK_LAUNCH("soft_argmin");
  return 0;
}

}  // namespace adamvs
