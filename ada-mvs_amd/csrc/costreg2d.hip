// CostRegNet2D: the depth-as-channel 2D hourglass that scores each source view
// in stage 1 (reference models/adamvs.py:198-238, blocks models/module.py:254-261),
// and the per-pixel softmax / max / expectation that turns its scores into a view
// weight and a pair depth (reference models/adamvs.py:481-486, module.py:617-625).
//
// Every layer is a D->D 3x3 convolution: an implicit GEMM with K = 9*D on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).  Activations are channel-last
// [N][h*w][D]; rows of the MFMA tile are output channels, columns are 16
// consecutive output pixels of one row, so a lane ends up with 4 consecutive
// channels of one pixel (one 16-byte store).  Weights are pre-packed on the host in
// A-fragment order with the eval-mode BatchNorm scale folded in and stream
// straight from L2 into VGPRs (each wave owns different output channels, so
// there is nothing to share through LDS); activations go through a planar LDS
// tile shared by the block's four waves.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "conv_frag.h"
#include "costreg_softmax.h"
#include "kernels.h"
#include "persistent.h"

namespace adamvs {

struct ConvDDArgs {
  const float* in;     // [N][hi*wi][D]
  const float* wpk;    // [9][D/4][D/16][64]
  const float* bias;   // [D]
  const float* skip;   // [N][ho*wo][D] or null; added after the ReLU (adamvs.py:234-236)
  float* out;          // [N][ho*wo][D]
  int D, hi, wi, ho, wo, relu;
  // softmax epilogue (the `prob` layer inside a stage, reference adamvs.py:481-486): when sm_vw != null the scores are not
  // stored; the block reduces them over D and writes view_weight = max_d softmax and pair_depth = sum_d softmax * depth_d
  float* sm_vw; float* sm_pd;   // [N][ho*wo]
  PlaneSrc sm_planes;           // image n belongs to tile n % sm_B
  int sm_B;
  const float* in2;    // [N][hi*wi][D] or null: the layer convolves in + in2, summed when the window enters LDS.  The
                       // hourglass's skip additions (x = conv4 + conv7(x), adamvs.py:234-236) run HERE, in the consumer of x:
                       // in the producer's epilogue a skip operand is a second round of loads that nothing overlaps
  // k_conv_dd_resident only (MS-REDNet's deep GRU levels): GroupNorm(1 group) partial sums of the output's channels [0, gn_n)
  // in the epilogue -- wave w of workgroup (x, y) writes (sum, sum of squares) in double to
  // gn_part[((n * gn_ngroups + gn_group) * parts + (y * gridDim.x + x) * 4 + w) * 2], parts = 4 * workgroups per map
  double* gn_part;
  int gn_n, gn_group, gn_ngroups;
  int sm_D;            // softmax epilogue: the number of hypothesis planes when the network runs wider (costreg_width); 0 = D
};

enum { CONV_S1 = 0, CONV_S2 = 1, CONV_T2 = 2 };

// window of a block of BR output rows x 16 output columns (CONV_T2: input positions)
template <int MODE, int BR = 8> struct TileGeom;
template <int BR> struct TileGeom<CONV_S1, BR> { static constexpr int LR = BR + 2, LC = 18, PLANE = plane_pitch16((BR + 2) * 18); };
template <int BR> struct TileGeom<CONV_S2, BR> { static constexpr int LR = 2 * BR + 1, LC = 33, PLANE = ((2 * BR + 1) * 33) | 1; };
template <int BR> struct TileGeom<CONV_T2, BR> { static constexpr int LR = BR + 1, LC = 17, PLANE = plane_pitch16((BR + 1) * 17); };

// KB = input channels per LDS chunk (template parameter: 8 = two MFMA k-steps, 4 = one)

// Block = 8 rows x 16 columns of output positions x all D output channels.
// Waves: WM along output channels (MT tiles of 16 each), WN = 4/WM along rows.
// PY/PX: output parity class, CONV_T2 only (ConvTranspose2d k3 s2 p1 op1).
//
// fp32 MFMA shares the vector lanes (tools/microbench/mfma_issue.hip: every extra VALU instruction per MFMA costs
// 3-6 cycles of matrix time, at any occupancy), and at this register budget there is one wave per SIMD, so the
// chunk loop is written to contain matrix instructions and nothing else that needs the vector ALU:
//  * global accesses are buffer loads: uniform descriptor, per-lane byte offset computed once (bounds folded in
//    as BUF_OOB -> zero fill), the chunk enters as the scalar offset operand;
//  * LDS addresses of the tile fill and of the B-fragment reads are per-lane constants, pinned in registers;
//  * two named register sets for the A fragments and two chunks per loop trip: the fragments and the activation
//    tile of chunk k+1 are requested before the MFMAs of chunk k and waited for once, after them (vmcnt retires
//    in order; a single explicit wait keeps the compiler from scheduling its own in the middle of the chain).
template <int MT, int WM, int MODE, int KB, int PY, int PX, int BR = 8, bool TWO = false, bool SOFTMAX = false>
__device__ __forceinline__ void conv_dd_body(const ConvDDArgs& a, float* lds, int n, int by, int bx) {
  using TG = TileGeom<MODE, BR>;
  constexpr int LR = TG::LR, LC = TG::LC, PLANE = TG::PLANE, GP = group_pitch(PLANE, KB / 4);
  constexpr int WN = 4 / WM, NTR = BR / WN;
  static_assert(NTR >= 1, "rows per wave");
  constexpr int STR = (MODE == CONV_S2) ? 2 : 1;
  constexpr int NTY = (MODE == CONV_T2) ? 1 + PY : 3;
  constexpr int NTX = (MODE == CONV_T2) ? 1 + PX : 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
  const int wm = wave % WM, wn = wave / WM;
  const int p = lane & 15, q = lane >> 4;
  const int D = a.D, KCT = D / 4, NTILES = D / 16;
  const int r0 = by * BR, c0 = bx * 16;                      // block origin (output rows/cols, or input i/j for T2)
  const int iy0 = (MODE == CONV_T2) ? r0 : r0 * STR - 1;
  const int ix0 = (MODE == CONV_T2) ? c0 : c0 * STR - 1;
  constexpr int NTAP = NTY * NTX;
  constexpr int NITEMS = LR * LC * (KB / 4), NITA = (NITEMS + 255) / 256;

  // ---- per-lane constants
  // activation tile: item = (pixel of the window, group of 4 channels); out-of-image pixels read as zero
  const buf_rsrc rx = make_rsrc((const char*)a.in + (((long)n * a.hi + iy0) * a.wi + ix0) * (long)D * 4);
  unsigned xoff[NITA], xlds[NITA];
#pragma unroll
  for (int it = 0; it < NITA; ++it) {
    const int i = min(tid + it * 256, NITEMS - 1);           // surplus lanes repeat the last item
    const int g = i % (KB / 4), pp = i / (KB / 4), r = pp / LC, c = pp % LC;
    const bool ok = (unsigned)(iy0 + r) < (unsigned)a.hi && (unsigned)(ix0 + c) < (unsigned)a.wi;
    xoff[it] = ok ? (unsigned)(((r * a.wi + c) * D + 4 * g) * 4) : BUF_OOB;
    xlds[it] = (unsigned)((g * GP + r * LC + c) * 4);
    pin(xoff[it]); pin(xlds[it]);
  }
  // A fragments: [tap][D/4][D/16][64]; lane offset only, the rest is uniform
  const buf_rsrc rw = make_rsrc(a.wpk);
  unsigned woff = (unsigned)(lane * 4);
  pin(woff);
  // B fragments: lane's k-row and pixel, per k-chunk of the LDS tile
  unsigned xb[KB / 4];
#pragma unroll
  for (int kc = 0; kc < KB / 4; ++kc) {
    xb[kc] = (unsigned)((kc * GP + q * PLANE + (wn * NTR * STR) * LC + p * STR) * 4);
    pin(xb[kc]);
  }

  f32x4 acc[MT][NTR];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < NTR; ++r) acc[mt][r] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_w = [&](float (&wf)[NTAP][KB / 4][MT], int ch) {
#pragma unroll
    for (int ty = 0; ty < NTY; ++ty)
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) {
        const int ky = (MODE == CONV_T2) ? (PY ? (ty ? 0 : 2) : 1) : ty;
        const int kx = (MODE == CONV_T2) ? (PX ? (tx ? 0 : 2) : 1) : tx;
#pragma unroll
        for (int kc = 0; kc < KB / 4; ++kc)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const unsigned frag = (unsigned)(((ky * 3 + kx) * KCT + ch / 4 + kc) * NTILES + wm * MT + mt) * 256u;   // uniform
            wf[ty * NTX + tx][kc][mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, woff, frag, 0));
          }
      }
  };
  constexpr bool two = TWO;          // the layer convolves in + in2 (a separate instantiation: the plain chunk loop stays as it was)
  const buf_rsrc rx2 = make_rsrc((const char*)(two ? a.in2 : a.in) + (((long)n * a.hi + iy0) * a.wi + ix0) * (long)D * 4);
  f32x4 xs2[two ? NITA : 1];
  auto load_x = [&](f32x4 (&st)[NITA], int ch) {
#pragma unroll
    for (int it = 0; it < NITA; ++it)
      st[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[it], (unsigned)ch * 4u, 0));
    if (two) {
#pragma unroll
      for (int it = 0; it < NITA; ++it)
        xs2[two ? it : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx2, xoff[it], (unsigned)ch * 4u, 0));
    }
  };
  auto store_x = [&](const f32x4 (&st)[NITA]) {
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
      float* dl = (float*)((char*)lds + xlds[it]);
      const f32x4 v = two ? st[it] + xs2[two ? it : 0] : st[it];
      dl[0] = v.x; dl[PLANE] = v.y; dl[2 * PLANE] = v.z; dl[3 * PLANE] = v.w;
    }
  };
  auto mfma_chunk = [&](const float (&wf)[NTAP][KB / 4][MT]) {
#pragma unroll
    for (int ty = 0; ty < NTY; ++ty)
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx)
#pragma unroll
        for (int kc = 0; kc < KB / 4; ++kc) {
          float bv[NTR];
#pragma unroll
          for (int r = 0; r < NTR; ++r) bv[r] = *(const float*)((const char*)lds + xb[kc] + ((r * STR + ty) * LC + tx) * 4);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < NTR; ++r) acc[mt][r] = mfma16(wf[ty * NTX + tx][kc][mt], bv[r], acc[mt][r]);
        }
  };

  float wfA[NTAP][KB / 4][MT], wfB[NTAP][KB / 4][MT];
  f32x4 xs[NITA];
  load_w(wfA, 0);
  load_x(xs, 0);
  for (int ch = 0; ch < D; ch += 2 * KB) {       // D / KB is even for every supported D
    wait_vmem_all();
    __syncthreads();                    // previous chunk's readers are done
    store_x(xs);
    __syncthreads();
    load_w(wfB, ch + KB);
    load_x(xs, ch + KB);
    mfma_chunk(wfA);

    wait_vmem_all();
    __syncthreads();
    store_x(xs);
    __syncthreads();
    if (ch + 2 * KB < D) {
      load_w(wfA, ch + 2 * KB);
      load_x(xs, ch + 2 * KB);
    }
    mfma_chunk(wfB);
  }

  if (SOFTMAX) {
    __syncthreads();                                   // the last chunk's readers are done: the tile space is reused
    softmax_epilogue<MT, WM, NTR, BR>(acc, a.bias, a.sm_planes, a.sm_B, n, r0, c0, a.ho, a.wo, a.sm_D ? a.sm_D : D, a.sm_vw, a.sm_pd, lds);
    return;
  }
  // epilogue: lane owns channels co4..co4+3 of the pixel in column p
#pragma unroll
  for (int r = 0; r < NTR; ++r) {
    int row = r0 + wn * NTR + r, col = c0 + p;
    int oy = (MODE == CONV_T2) ? 2 * row + PY : row;
    int ox = (MODE == CONV_T2) ? 2 * col + PX : col;
    bool valid = (MODE == CONV_T2) ? (row < a.hi && col < a.wi) : (oy < a.ho && ox < a.wo);
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

// ConvTranspose2d(k3, s2, p1, op1), all four output parity classes of one 8 x 16 block of input positions in one
// workgroup, one after the other.  A class on its own is a short block (1, 2, 2 or 4 taps against the 9 of a
// stride-1 block) whose prologue -- per-lane constants, the latency of the first fragment and window loads -- and
// epilogue weigh as much as a third of it; here the constants are shared and the first chunk of the next class is
// requested during the last chunk of the current one, so its loads and the current class's epilogue overlap.
template <int MT, int WM, int KB>
__device__ __forceinline__ void conv_dd_t2_all(const ConvDDArgs& a, float* lds, int n, int by, int bx) {
  using TG = TileGeom<CONV_T2>;
  constexpr int LR = TG::LR, LC = TG::LC, PLANE = TG::PLANE, GP = group_pitch(PLANE, KB / 4);
  constexpr int WN = 4 / WM, NTR = 8 / WN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
  const int wm = wave % WM, wn = wave / WM;
  const int p = lane & 15, q = lane >> 4;
  const int D = a.D, KCT = D / 4, NTILES = D / 16;
  const int r0 = by * 8, c0 = bx * 16;                       // block origin: input positions (i, j)
  constexpr int NITEMS = LR * LC * (KB / 4), NITA = (NITEMS + 255) / 256;

  const buf_rsrc rx = make_rsrc((const char*)a.in + (((long)n * a.hi + r0) * a.wi + c0) * (long)D * 4);
  const bool two = a.in2 != nullptr;                         // uniform: the layer convolves in + in2
  const buf_rsrc rx2 = make_rsrc((const char*)(two ? a.in2 : a.in) + (((long)n * a.hi + r0) * a.wi + c0) * (long)D * 4);
  unsigned xoff[NITA], xlds[NITA];
#pragma unroll
  for (int it = 0; it < NITA; ++it) {
    const int i = min(tid + it * 256, NITEMS - 1);           // surplus lanes repeat the last item
    const int g = i % (KB / 4), pp = i / (KB / 4), r = pp / LC, c = pp % LC;
    const bool ok = r0 + r < a.hi && c0 + c < a.wi;
    xoff[it] = ok ? (unsigned)(((r * a.wi + c) * D + 4 * g) * 4) : BUF_OOB;
    xlds[it] = (unsigned)((g * GP + r * LC + c) * 4);
    pin(xoff[it]); pin(xlds[it]);
  }
  const buf_rsrc rw = make_rsrc(a.wpk);
  unsigned woff = (unsigned)(lane * 4);
  pin(woff);
  unsigned xb[KB / 4];
#pragma unroll
  for (int kc = 0; kc < KB / 4; ++kc) {
    xb[kc] = (unsigned)((kc * GP + q * PLANE + (wn * NTR) * LC + p) * 4);
    pin(xb[kc]);
  }
  // epilogue: the lane's output pixel of class (0, 0) in row r of its wave, as a byte offset from the image (the class and
  // the channel tile are added as uniform terms); BUF_OOB outside the map, so that EVERY lane issues every store and
  // the number of stores in flight after an epilogue is known: the next class does not wait for them
  const buf_rsrc ro = make_rsrc((char*)a.out + (long)n * a.ho * a.wo * (long)D * 4);
  const buf_rsrc rk = make_rsrc((const char*)(a.skip ? a.skip : a.out) + (long)n * a.ho * a.wo * (long)D * 4);
  unsigned ooff[NTR];
#pragma unroll
  for (int r = 0; r < NTR; ++r) {
    const int row = r0 + wn * NTR + r, col = c0 + p;
    ooff[r] = (row < a.hi && col < a.wi) ? (unsigned)((((2 * row) * a.wo + 2 * col) * D + wm * MT * 16 + 4 * q) * 4) : BUF_OOB;
    pin(ooff[r]);
  }

  float wfA[4][KB / 4][MT], wfB[4][KB / 4][MT];              // up to 4 taps
  f32x4 xs[NITA], xs2[NITA];
  f32x4 acc[MT][NTR];

  auto load_x = [&](int ch) {
#pragma unroll
    for (int it = 0; it < NITA; ++it)
      xs[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[it], (unsigned)ch * 4u, 0));
    if (two) {
#pragma unroll
      for (int it = 0; it < NITA; ++it)
        xs2[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx2, xoff[it], (unsigned)ch * 4u, 0));
    }
  };
  auto store_x = [&]() {
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
      float* dl = (float*)((char*)lds + xlds[it]);
      const f32x4 v = two ? xs[it] + xs2[it] : xs[it];
      dl[0] = v.x; dl[PLANE] = v.y; dl[2 * PLANE] = v.z; dl[3 * PLANE] = v.w;
    }
  };
  auto load_w = [&](auto pyc, auto pxc, float (&wf)[4][KB / 4][MT], int ch) {
    constexpr int PY = decltype(pyc)::value, PX = decltype(pxc)::value;
#pragma unroll
    for (int ty = 0; ty <= PY; ++ty)
#pragma unroll
      for (int tx = 0; tx <= PX; ++tx) {
        const int ky = PY ? (ty ? 0 : 2) : 1, kx = PX ? (tx ? 0 : 2) : 1;
#pragma unroll
        for (int kc = 0; kc < KB / 4; ++kc)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const unsigned frag = (unsigned)(((ky * 3 + kx) * KCT + ch / 4 + kc) * NTILES + wm * MT + mt) * 256u;   // uniform
            wf[ty * (1 + PX) + tx][kc][mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, woff, frag, 0));
          }
      }
  };
  auto mfma_chunk = [&](auto pyc, auto pxc, const float (&wf)[4][KB / 4][MT]) {
    constexpr int PY = decltype(pyc)::value, PX = decltype(pxc)::value;
#pragma unroll
    for (int ty = 0; ty <= PY; ++ty)
#pragma unroll
      for (int tx = 0; tx <= PX; ++tx)
#pragma unroll
        for (int kc = 0; kc < KB / 4; ++kc) {
          float bv[NTR];
#pragma unroll
          for (int r = 0; r < NTR; ++r) bv[r] = *(const float*)((const char*)lds + xb[kc] + ((r + ty) * LC + tx) * 4);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < NTR; ++r) acc[mt][r] = mfma16(wf[ty * (1 + PX) + tx][kc][mt], bv[r], acc[mt][r]);
        }
  };
  // one class; wfA and xs hold its first chunk on entry; on exit they hold the first chunk of class (nyc, nxc), if any.
  // `first`: nothing but that first chunk is in flight (block entry); otherwise the previous class's stores are, and the
  // wait for the chunk -- requested BEFORE them, vmcnt retires in order -- is left to the compiler, which counts them.
  auto run_class = [&](auto pyc, auto pxc, auto nyc, auto nxc, auto has_next, auto first) {
    constexpr int PY = decltype(pyc)::value, PX = decltype(pxc)::value;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < NTR; ++r) acc[mt][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < D; ch += 2 * KB) {
      if (decltype(first)::value || ch > 0) wait_vmem_all();
      __syncthreads();
      store_x();
      __syncthreads();
      load_w(pyc, pxc, wfB, ch + KB);
      load_x(ch + KB);
      mfma_chunk(pyc, pxc, wfA);

      wait_vmem_all();
      __syncthreads();
      store_x();
      __syncthreads();
      if (ch + 2 * KB < D) {
        load_w(pyc, pxc, wfA, ch + 2 * KB);
        load_x(ch + 2 * KB);
      } else if (decltype(has_next)::value) {
        load_w(nyc, nxc, wfA, 0);
        load_x(0);
      }
      mfma_chunk(pyc, pxc, wfB);
    }
    const unsigned cls = (unsigned)((PY * a.wo + PX) * D * 4);           // uniform: the class's pixel offset
#pragma unroll
    for (int r = 0; r < NTR; ++r)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4 v = acc[mt][r] + *(const f32x4*)(a.bias + (wm * MT + mt) * 16 + 4 * q);
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const unsigned o = ooff[r] == BUF_OOB ? BUF_OOB : ooff[r] + cls + (unsigned)(mt * 64);
        if (a.skip) v += buf_load4(rk, o);
        buf_store4(ro, o, v);
      }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  load_w(I1{}, I1{}, wfA, 0);
  load_x(0);
  run_class(I1{}, I1{}, I1{}, I0{}, std::true_type{}, std::true_type{});
  run_class(I1{}, I0{}, I0{}, I1{}, std::true_type{}, std::false_type{});
  run_class(I0{}, I1{}, I0{}, I0{}, std::true_type{}, std::false_type{});
  run_class(I0{}, I0{}, I0{}, I0{}, std::false_type{}, std::false_type{});
}

// ConvTranspose2d(k3, s2, p1, op1) with the four output parity classes of a block of input positions advanced TOGETHER,
// chunk by chunk of input channels.  conv_dd_t2_all above runs the classes one after the other: each loads the whole
// input window again and spends a barrier pair per channel chunk on 1, 2, 2 or 4 taps (24 ... 96 MFMAs per wave), which is
// why the transposed layers sat at 40-71 % of the matrix rate.  Here a chunk is loaded once and feeds all nine
// (input offset, class) pairs -- exactly the nine taps of the 3 x 3 kernel, each used once:
//     offset (0,0): class 00 tap (1,1) | 01 (1,2) | 10 (2,1) | 11 (2,2);  offset (0,1): 01 (1,0) | 11 (2,0);
//     offset (1,0): 10 (0,1) | 11 (0,2);  offset (1,1): 11 (0,0)          (class = output parity (py, px))
// The price is four accumulator sets: at two waves per SIMD they leave room for BR = 2 rows x 16 columns of input
// positions per block (4 x 32 outputs; 4 rows need one wave per SIMD and were 40 % slower, 8 rows spill), i.e. 54 MFMAs
// per wave and barrier pair -- the average of the class-by-class kernel, but uniformly, with the window loaded once and
// four times the workgroups: it wins on SMALL grids (the deep levels of the hourglass, few tiles per launch) and is
// chosen there (launch_conv_dd_cfg).
template <int MT, int WM, int KB, int BR>
__device__ __forceinline__ void conv_dd_t2_fused(const ConvDDArgs& a, float* lds, int n, int by, int bx) {
  constexpr int LR = BR + 1, LC = 17, PLANE = plane_pitch16(LR * LC), GP = group_pitch(PLANE, KB / 4);
  constexpr int WN = 4 / WM, NTR = BR / WN;
  static_assert(NTR >= 1, "rows per wave");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
  const int wm = wave % WM, wn = wave / WM;
  const int p = lane & 15, q = lane >> 4;
  const int D = a.D, KCT = D / 4, NTILES = D / 16;
  const int r0 = by * BR, c0 = bx * 16;                      // block origin: input positions (i, j)
  constexpr int NITEMS = LR * LC * (KB / 4), NITA = (NITEMS + 255) / 256;

  const buf_rsrc rx = make_rsrc((const char*)a.in + (((long)n * a.hi + r0) * a.wi + c0) * (long)D * 4);
  unsigned xoff[NITA], xlds[NITA];
#pragma unroll
  for (int it = 0; it < NITA; ++it) {
    const int i = min(tid + it * 256, NITEMS - 1);           // surplus lanes repeat the last item
    const int g = i % (KB / 4), pp = i / (KB / 4), r = pp / LC, c = pp % LC;
    const bool ok = r0 + r < a.hi && c0 + c < a.wi;
    xoff[it] = ok ? (unsigned)(((r * a.wi + c) * D + 4 * g) * 4) : BUF_OOB;
    xlds[it] = (unsigned)((g * GP + r * LC + c) * 4);
    pin(xoff[it]); pin(xlds[it]);
  }
  const buf_rsrc rw = make_rsrc(a.wpk);
  unsigned woff = (unsigned)(lane * 4);
  pin(woff);
  unsigned xb[KB / 4];
#pragma unroll
  for (int kc = 0; kc < KB / 4; ++kc) {
    xb[kc] = (unsigned)((kc * GP + q * PLANE + (wn * NTR) * LC + p) * 4);
    pin(xb[kc]);
  }

  f32x4 acc[4][MT][NTR];                                     // [class = 2 py + px]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < NTR; ++r) acc[c][mt][r] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_w = [&](float (&wf)[9][KB / 4][MT], int ch) {    // all nine taps of the chunk, tap = ky * 3 + kx
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kc = 0; kc < KB / 4; ++kc)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const unsigned frag = (unsigned)((t * KCT + ch / 4 + kc) * NTILES + wm * MT + mt) * 256u;   // uniform
          wf[t][kc][mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, woff, frag, 0));
        }
  };
  const bool two = a.in2 != nullptr;                         // uniform: the layer convolves in + in2
  const buf_rsrc rx2 = make_rsrc((const char*)(two ? a.in2 : a.in) + (((long)n * a.hi + r0) * a.wi + c0) * (long)D * 4);
  f32x4 xs2[NITA];
  auto load_x = [&](f32x4 (&st)[NITA], int ch) {
#pragma unroll
    for (int it = 0; it < NITA; ++it)
      st[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[it], (unsigned)ch * 4u, 0));
    if (two) {
#pragma unroll
      for (int it = 0; it < NITA; ++it)
        xs2[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx2, xoff[it], (unsigned)ch * 4u, 0));
    }
  };
  auto store_x = [&](const f32x4 (&st)[NITA]) {
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
      float* dl = (float*)((char*)lds + xlds[it]);
      const f32x4 v = two ? st[it] + xs2[it] : st[it];
      dl[0] = v.x; dl[PLANE] = v.y; dl[2 * PLANE] = v.z; dl[3 * PLANE] = v.w;
    }
  };
  auto mfma_chunk = [&](const float (&wf)[9][KB / 4][MT]) {
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int tx = 0; tx < 2; ++tx)
#pragma unroll
        for (int kc = 0; kc < KB / 4; ++kc) {
          float bv[NTR];
#pragma unroll
          for (int r = 0; r < NTR; ++r) bv[r] = *(const float*)((const char*)lds + xb[kc] + ((r + ty) * LC + tx) * 4);
#pragma unroll
          for (int py = ty; py < 2; ++py)                    // classes that reach input offset (ty, tx)
#pragma unroll
            for (int px = tx; px < 2; ++px) {
              const int ky = py ? (ty ? 0 : 2) : 1, kx = px ? (tx ? 0 : 2) : 1;
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < NTR; ++r)
                  acc[2 * py + px][mt][r] = mfma16(wf[ky * 3 + kx][kc][mt], bv[r], acc[2 * py + px][mt][r]);
            }
        }
  };

  float wfA[9][KB / 4][MT], wfB[9][KB / 4][MT];
  f32x4 xs[NITA];
  load_w(wfA, 0);
  load_x(xs, 0);
  for (int ch = 0; ch < D; ch += 2 * KB) {       // D / KB is even for every supported D
    wait_vmem_all();
    __syncthreads();                    // previous chunk's readers are done
    store_x(xs);
    __syncthreads();
    load_w(wfB, ch + KB);
    load_x(xs, ch + KB);
    mfma_chunk(wfA);

    wait_vmem_all();
    __syncthreads();
    store_x(xs);
    __syncthreads();
    if (ch + 2 * KB < D) {
      load_w(wfA, ch + 2 * KB);
      load_x(xs, ch + 2 * KB);
    }
    mfma_chunk(wfB);
  }

#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < NTR; ++r) {
      const int row = r0 + wn * NTR + r, col = c0 + p;
      if (!(row < a.hi && col < a.wi)) continue;
      const size_t opix = ((size_t)n * a.ho + 2 * row + (c >> 1)) * a.wo + 2 * col + (c & 1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int co4 = (wm * MT + mt) * 16 + 4 * q;
        f32x4 v = acc[c][mt][r] + *(const f32x4*)(a.bias + co4);
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (a.skip) v += *(const f32x4*)(a.skip + opix * D + co4);
        *(f32x4*)(a.out + opix * D + co4) = v;
      }
    }
}

// Stride-2 layer (conv1, conv3, conv5 of the hourglass: reference models/adamvs.py:206-211, 229-232) with the minimal-filtering
// form of its polyphase decomposition along x: two neighbouring outputs of a row share the input column between them,
//     out[2m]     = W0 c[4m]   + W1 c[4m+1] + W2 c[4m+2]
//     out[2m + 1] = W0 c[4m+2] + W1 c[4m+3] + W2 c[4m+4]            (c = the window row, W = the three taps of one kernel row)
// and the two-tap filter (W0, W2) over the even columns costs three products per two outputs instead of four:
//     A0 = W1 c[4m+1] + W0 (c[4m] - c[4m+2]);  A1 = W1 c[4m+3] + W2 (c[4m+4] - c[4m+2]);  A2 = (W0 + W2) c[4m+2]
//     out[2m] = A0 + A2,  out[2m + 1] = A1 + A2
// -- five MFMAs per (kernel row, 4 input channels, channel tile, output row) for 32 output pixels where the direct form issues six
// (15 / 18 of the layer's products; the same along y would give 25 / 36 but needs nine accumulator sets).  The columns of an MFMA are
// 16 output PAIRS; a lane reads its five window values as one 16-byte and one 4-byte LDS read (row pitch 68: aligned), forms the
// two differences (2 vector instructions per 15 MFMAs) and W0 + W2 once per chunk from the fragments it streams anyway (9 adds):
// the blob keeps its nine taps.  Block = 3 output rows x 32 columns x all channels: three accumulator sets of 3 x 3 tiles = 108
// registers, two waves per SIMD as in conv_dd_body; one k-step per chunk, chunks double-buffered the same way.
constexpr int S2P_ROWS = 3;      // output rows per block (4: 144 accumulator registers next to 63 of fragments: spills; the hourglass's 48 / 24 / 12 rows divide by 3)
template <int MT, int WM>
__device__ __forceinline__ void conv_dd_s2_pairs(const ConvDDArgs& a, float* lds, int n, int by, int bx) {
  static_assert(WM == 4, "every wave takes all rows of the block");
  constexpr int BR = S2P_ROWS, LR = 2 * BR + 1, WCOLS = 65, LC = 68, PLANE = ((LR * LC + 31) / 32) * 32 + 16;
  constexpr int NITEMS = LR * WCOLS, NITA = (NITEMS + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63, wm = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave = channel slice
  const int p = lane & 15, q = lane >> 4;
  const int D = a.D, KCT = D / 4, NTILES = D / 16;
  const int r0 = by * BR, c0 = bx * 32;
  const int iy0 = 2 * r0 - 1, ix0 = 2 * c0 - 1;

  const buf_rsrc rx = make_rsrc((const char*)a.in + (((long)n * a.hi + iy0) * a.wi + ix0) * (long)D * 4);
  unsigned xoff[NITA], xlds[NITA];
#pragma unroll
  for (int it = 0; it < NITA; ++it) {
    const int i = min(tid + it * 256, NITEMS - 1);           // surplus lanes repeat the last item
    const int r = i / WCOLS, c = i % WCOLS;
    const bool ok = (unsigned)(iy0 + r) < (unsigned)a.hi && (unsigned)(ix0 + c) < (unsigned)a.wi;
    xoff[it] = ok ? (unsigned)(((r * a.wi + c) * D) * 4) : BUF_OOB;
    xlds[it] = (unsigned)((r * LC + c) * 4);
    pin(xoff[it]); pin(xlds[it]);
  }
  const buf_rsrc rw = make_rsrc(a.wpk);
  unsigned woff = (unsigned)(lane * 4);
  pin(woff);
  unsigned xb = (unsigned)((q * PLANE + 4 * p) * 4);           // the lane's k-row and first window column of its pair
  pin(xb);

  f32x4 acc[3][MT][BR];                                        // A0 | A1 | A2
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < BR; ++r) acc[t][mt][r] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_w = [&](float (&wf)[9][MT], int ch) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const unsigned frag = (unsigned)((t * KCT + ch / 4) * NTILES + wm * MT + mt) * 256u;   // uniform
        wf[t][mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, woff, frag, 0));
      }
  };
  auto load_x = [&](f32x4 (&st)[NITA], int ch) {
#pragma unroll
    for (int it = 0; it < NITA; ++it)
      st[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[it], (unsigned)ch * 4u, 0));
  };
  auto store_x = [&](const f32x4 (&st)[NITA]) {
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
      float* dl = (float*)((char*)lds + xlds[it]);
      dl[0] = st[it].x; dl[PLANE] = st[it].y; dl[2 * PLANE] = st[it].z; dl[3 * PLANE] = st[it].w;
    }
  };
  auto mfma_chunk = [&](const float (&wf)[9][MT]) {
    float w02[3][MT];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) w02[ky][mt] = wf[ky * 3 + 0][mt] + wf[ky * 3 + 2][mt];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int r = 0; r < BR; ++r) {
        const char* row = (const char*)lds + xb + ((2 * r + ky) * LC) * 4;
        const f32x4 c = *(const f32x4*)row;
        const float c4 = *(const float*)(row + 16);
        const float d0 = c.x - c.z, d1 = c4 - c.z;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[0][mt][r] = mfma16(wf[ky * 3 + 1][mt], c.y, acc[0][mt][r]);
          acc[0][mt][r] = mfma16(wf[ky * 3 + 0][mt], d0, acc[0][mt][r]);
          acc[1][mt][r] = mfma16(wf[ky * 3 + 1][mt], c.w, acc[1][mt][r]);
          acc[1][mt][r] = mfma16(wf[ky * 3 + 2][mt], d1, acc[1][mt][r]);
          acc[2][mt][r] = mfma16(w02[ky][mt], c.z, acc[2][mt][r]);
        }
      }
  };

  float wfA[9][MT], wfB[9][MT];
  f32x4 xs[NITA];
  load_w(wfA, 0);
  load_x(xs, 0);
  for (int ch = 0; ch < D; ch += 8) {            // two chunks of 4 input channels per trip (D / 4 is even for every supported D)
    wait_vmem_all();
    __syncthreads();                    // previous chunk's readers are done
    store_x(xs);
    __syncthreads();
    load_w(wfB, ch + 4);
    load_x(xs, ch + 4);
    mfma_chunk(wfA);

    wait_vmem_all();
    __syncthreads();
    store_x(xs);
    __syncthreads();
    if (ch + 8 < D) {
      load_w(wfA, ch + 8);
      load_x(xs, ch + 8);
    }
    mfma_chunk(wfB);
  }
  // epilogue: lane owns channels co4 .. co4 + 3 of the output pair (c0 + 2 p, c0 + 2 p + 1) of every row
#pragma unroll
  for (int r = 0; r < BR; ++r) {
    const int oy = r0 + r, ox = c0 + 2 * p;
    if (oy >= a.ho || ox >= a.wo) continue;
    const size_t opix = ((size_t)n * a.ho + oy) * a.wo + ox;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int co4 = (wm * MT + mt) * 16 + 4 * q;
      const f32x4 b = *(const f32x4*)(a.bias + co4);
      f32x4 v0 = (acc[0][mt][r] + acc[2][mt][r]) + b, v1 = (acc[1][mt][r] + acc[2][mt][r]) + b;
      if (a.relu) {
        v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
        v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
      }
      *(f32x4*)(a.out + opix * D + co4) = v0;
      if (ox + 1 < a.wo) *(f32x4*)(a.out + (opix + 1) * D + co4) = v1;
    }
  }
}

// grid: (ceil(wo/32), ceil(ho/S2P_ROWS), N); block 256
template <int MT, int WM>
__global__ __launch_bounds__(256, 2) void k_conv_dd_s2p(ConvDDArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[4 * ((((2 * S2P_ROWS + 1) * 68 + 31) / 32) * 32 + 16)];
  conv_dd_s2_pairs<MT, WM>(a, lds, blockIdx.z, blockIdx.y, blockIdx.x);
}

// option s2_pairs = 0: the stride-2 layers on the direct kernel, as in rounds 1-4 (A/B)
static bool s2_pairs() { return opt(OPT_S2_PAIRS) != 0; }

// grid: (ceil(cols/16), ceil(rows/BR), N); block 256; OCC = waves per SIMD the register budget is held to
template <int MT, int WM, int KB, int BR, int OCC>
__global__ __launch_bounds__(256, OCC) void k_conv_dd_t2_fused(ConvDDArgs a) {
  __shared__ float lds[(KB / 4) * group_pitch(plane_pitch16((BR + 1) * 17), KB / 4)];
  conv_dd_t2_fused<MT, WM, KB, BR>(a, lds, blockIdx.z, blockIdx.y, blockIdx.x);
}

// grid: (ceil(cols/16), ceil(rows/8), N); block 256
// TWO (stride-1 layers only): the layer convolves in + in2; the transposed kernel takes in2 as a uniform run-time switch
// SM (stride-1 layers only): the softmax epilogue of the `prob` layer (conv_dd_body)
constexpr int imax(int a, int b) { return a > b ? a : b; }
template <int MT, int WM, int MODE, int KB, bool TWO = false, bool SM = false>
__global__ __launch_bounds__(256, (MT == 4 && WM == 4) ? 1 : 2) void k_conv_dd(ConvDDArgs a) {
  __shared__ float lds[imax((KB / 4) * group_pitch(TileGeom<MODE>::PLANE, KB / 4), SM ? 8 * 16 * 4 * WM * 3 : 0)];
  if (MODE == CONV_T2) {
    conv_dd_t2_all<MT, WM, KB>(a, lds, blockIdx.z, blockIdx.y, blockIdx.x);
  } else {
    conv_dd_body<MT, WM, MODE, KB, 0, 0, 8, TWO, SM>(a, lds, blockIdx.z, blockIdx.y, blockIdx.x);
  }
}

// Stride-1 / stride-2 layers on a SMALL grid (the deep levels of the hourglass when few maps are in flight: 12 x 24 maps
// of 16 tiles are 64 blocks of 8 x 16 on 256 CUs, a third of each block padding): blocks of 2 output rows, four times the
// workgroups, no padded rows.  54 MFMAs per wave and barrier pair instead of 216 -- it pays only while the chip is not full.
template <int MT, int WM, int MODE, int KB, bool TWO = false>
__global__ __launch_bounds__(256, 2) void k_conv_dd_rows2(ConvDDArgs a) {
  __shared__ float lds[(KB / 4) * group_pitch(TileGeom<MODE, 2>::PLANE, KB / 4)];
  conv_dd_body<MT, WM, MODE, KB, 0, 0, 2, TWO>(a, lds, blockIdx.z, blockIdx.y, blockIdx.x);
}

// Measured (cfg3, 4 / 32 tiles per step): 64 blocks of 8 rows 0.185 -> 0.073 ms, 144: 0.193 -> 0.139, 576: 0.49 -> 0.38, 512: 0.33 -> 0.25,
// 1152: 0.80 -> 0.72; at 4608 blocks it is a wash and beyond it loses (10.9 -> 11.2 ms).  Option conv_rows2 = 0 / 1 forces.
static bool small_grid_rows2(long blocks8) {
  const int forced = opt(OPT_CONV_ROWS2);
  if (forced >= 0) return forced != 0;
  return blocks8 <= 2048;
}

// Two waves per SIMD (a single wave feeding the matrix pipe from LDS reaches ~83 % of it, two reach ~92 %:
// tools/microbench/mfma_issue.hip), i.e. at most 256 registers: the widest tilings take one k-step per chunk
// (D = 256 does not fit even so and keeps one wave).
// Which transposed kernel: measured (tools/r02_t2_ab.sh), the fused form wins while the class-by-class grid is small --
// 64 ... 2048 blocks: 0.246 -> 0.080 ms, 0.267 -> 0.169, 0.646 -> 0.509, 1.57 -> 1.31 -- and loses on the large layers
// (4608 blocks: 3.61 -> 3.92 ms; 18432: 14.1 -> 15.4), where the 4-tap class alone already gives the class-by-class
// kernel 96 MFMAs per barrier pair.  Option t2_fused = 0 / 1 forces one of them (A/B timing).
static bool t2_fused(long blocks_class_by_class) {
  const int forced = opt(OPT_T2_FUSED);
  if (forced >= 0) return forced != 0;
  return blocks_class_by_class <= 2048;
}

// option t2_kb8 = 0: one k-step per chunk in the transposed kernel at D = 192, as in rounds 1-2 (A/B).  Two k-steps halve
// the barrier pairs per class (a class has at most four taps, so the fragments still fit two waves per SIMD):
// measured conv11 of cfg2 14.2 -> 13.8 ms, with the epilogue freed of the skip operand 14.5 -> 12.3.
static bool t2_kb8() { return opt(OPT_T2_KB8) != 0; }
// option costreg_defer_skips = 0: the skip additions in the producing layer's epilogue, as in rounds 1-2 (A/B)
static bool costreg_deferred_skips() { return opt(OPT_COSTREG_DEFER_SKIPS) != 0; }

template <int MT, int WM>
static int launch_conv_dd_cfg(const ConvDDArgs& a_, int N, int mode, hipStream_t st) {
  const ConvDDArgs& a = a_;
  constexpr int KB = (WM == 4 || (MT == 4 && WM == 2)) ? 4 : 8;
  const bool rows2 = WM >= 2 && mode != CONV_T2 && small_grid_rows2((long)cdiv(a.wo, 16) * cdiv(a.ho, 8) * N);
  if (a.sm_vw) {                // the caller has checked cost_reg_softmax_fusable(): stride 1, no second input
    hipLaunchKernelGGL((k_conv_dd<MT, WM, CONV_S1, KB, false, true>), dim3(cdiv(a.wo, 16), cdiv(a.ho, 8), N), dim3(256), 0, st, a);
    ADAMVS_CHECK_LAUNCH("conv_dd (softmax epilogue)");
    return 0;
  }
  if (mode == CONV_S1 && rows2 && a.in2)
    hipLaunchKernelGGL((k_conv_dd_rows2<MT, (WM >= 2 ? WM : 2), CONV_S1, KB, true>), dim3(cdiv(a.wo, 16), cdiv(a.ho, 2), N), dim3(256), 0, st, a);
  else if (mode == CONV_S1 && a.in2)
    hipLaunchKernelGGL((k_conv_dd<MT, WM, CONV_S1, KB, true>), dim3(cdiv(a.wo, 16), cdiv(a.ho, 8), N), dim3(256), 0, st, a);
  else if (mode == CONV_S1 && rows2)
    hipLaunchKernelGGL((k_conv_dd_rows2<MT, (WM >= 2 ? WM : 2), CONV_S1, KB>), dim3(cdiv(a.wo, 16), cdiv(a.ho, 2), N), dim3(256), 0, st, a);
  else if (mode == CONV_S2 && rows2)
    hipLaunchKernelGGL((k_conv_dd_rows2<MT, (WM >= 2 ? WM : 2), CONV_S2, KB>), dim3(cdiv(a.wo, 16), cdiv(a.ho, 2), N), dim3(256), 0, st, a);
  else if (mode == CONV_S1)
    hipLaunchKernelGGL((k_conv_dd<MT, WM, CONV_S1, KB>), dim3(cdiv(a.wo, 16), cdiv(a.ho, 8), N), dim3(256), 0, st, a);
  // D = 192 (and 384 as two launches): 15 / 18 of the products.  Blocks are 32 output columns wide: taken where a row divides into
  // them (cfg2 at 128 tiles: conv1, 48 x 96 outputs, 11.47 -> 10.10 ms; conv3, 24 x 48 -- a block and a half per row -- 2.88 -> 3.31)
  else if (mode == CONV_S2 && MT == 3 && WM == 4 && !a.skip && !a.in2 && (a.wo % 32 == 0 || a.wo >= 256) && s2_pairs())
    hipLaunchKernelGGL((k_conv_dd_s2p<MT, (WM == 4 ? 4 : 4)>), dim3(cdiv(a.wo, 32), cdiv(a.ho, S2P_ROWS), N), dim3(256), 0, st, a);
  else if (mode == CONV_S2)
    hipLaunchKernelGGL((k_conv_dd<MT, WM, CONV_S2, KB>), dim3(cdiv(a.wo, 16), cdiv(a.ho, 8), N), dim3(256), 0, st, a);
  else if (WM >= 2 && t2_fused((long)cdiv(a.wi, 16) * cdiv(a.hi, 8) * N))      // small grids: 2-row blocks, all classes per chunk
    hipLaunchKernelGGL((k_conv_dd_t2_fused<MT, (WM >= 2 ? WM : 2), 4, 2, 2>), dim3(cdiv(a.wi, 16), cdiv(a.hi, 2), N), dim3(256), 0, st, a);
  else if (WM == 4 && MT == 3 && t2_kb8())     // at most four taps per class: two k-steps per chunk fit the register budget
    hipLaunchKernelGGL((k_conv_dd<MT, WM, CONV_T2, 8>), dim3(cdiv(a.wi, 16), cdiv(a.hi, 8), N), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((k_conv_dd<MT, WM, CONV_T2, KB>), dim3(cdiv(a.wi, 16), cdiv(a.hi, 8), N), dim3(256), 0, st, a);
  ADAMVS_CHECK_LAUNCH("conv_dd");
  return 0;
}

bool costreg_depth_supported(int D) {
  return D == 16 || D == 32 || D == 48 || D == 64 || D == 96 || D == 128 || D == 192 || D == 256 || D == 384 || D == 512;
}

// The width CostRegNet2D runs at for D hypotheses: the next supported one.  The reference builds the network for any D
// (models/adamvs.py:198-228); here the extra channels are zero weights (packing.py::pack_reg pads to this width: a pad
// channel's activations are ReLU(0) = 0 in every layer and feed nothing), the pair-similarity volume carries zeros in them,
// and the pad channels of `prob` get a bias of -1e30: exp(-1e30 - max) = 0 exactly, so softmax, its maximum and the depth
// expectation see D hypotheses.  0 = more than 512.
int costreg_width(int D) {
  static const int widths[] = {16, 32, 48, 64, 96, 128, 192, 256, 384, 512};
  for (int w : widths)
    if (D <= w) return w;
  return 0;
}
int costreg_width_bf16x3(int D) {
  static const int widths[] = {32, 64, 96, 128, 192, 256, 384, 512};
  for (int w : widths)
    if (D <= w) return w;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-1 layer on a SMALL grid (the deep GRU levels of MS-REDNet: 64 channels on 12x24 ... 48x96 maps, one
// convolution at a time in a recurrence).  An 8 x 16 output tile of k_conv_dd at D = 64 is 9.4 MFLOP = 37k cycles
// of fp32 MFMA on one CU (measured 27 us per launch whatever the map size) while a 12 x 24 map has three such tiles:
// 250 CUs idle.  Here a workgroup takes TR = 1..8 output rows x 16 columns, TR chosen at launch so that the grid
// covers the chip, and everything is requested once: each wave owns one tile of 16 output channels and keeps all its
// A fragments in registers (9 * D/4 <= 144 VGPRs), the (TR+2) x 18 window of all D input channels goes to LDS in one
// round of loads, and what follows is MFMAs fed from LDS.  Same fragment layout, planar LDS layout and epilogue as
// k_conv_dd.  D = 32: two waves per channel tile (NTR rows each, TR = 2 NTR); D = 64: one (TR = NTR).
// What the elementwise kernels between the convolutions of a ConvGRUCell2 did (reference models/module.py:72-106), folded
// into the window fill of the convolution that FOLLOWS them (MS-REDNet's deep levels at one or a few tiles are bound by the
// number of dependent launches per plane: four -> two).  Every workgroup finishes the GroupNorm reductions it needs from the
// partial sums the producing convolutions' epilogues wrote, then forms its input window on the fly:
//   GRU_PRO_GATES (the candidate convolution):  in = sigmoid(GN_r(f)) * h                       f = gate_conv's reset half
//   GRU_PRO_OUT   (the NEXT plane's gate convolutions):  in = h' = u * h + (1 - u) * tanh(GN_o(o)),   u = sigmoid(GN_u(f)),
//                 f = the previous plane's update half, o = its output_conv; the workgroup also stores h' of its own pixels
//                 (the new state, a second buffer: neighbours still read the old one) and its channels [0, hc) into R
// (halo pixels are recomputed by the neighbours: the same operands in the same order, the same bits).
// mean and 1 / sqrt(biased variance + eps) from `parts` partial sums (doubles), by the whole block, deterministic
template <int NTHR>
__device__ __forceinline__ void gru_pro_stats(const double* __restrict__ p, int parts, int count, float eps, double (*wsum)[2], float* st) {
  const int tid = threadIdx.x;
  double s_ = 0.0, q_ = 0.0;
  for (int k = tid; k < parts; k += NTHR) { s_ += p[2 * k]; q_ += p[2 * k + 1]; }
  for (int o = 32; o > 0; o >>= 1) { s_ += __shfl_down(s_, o); q_ += __shfl_down(q_, o); }
  if ((tid & 63) == 0) { wsum[tid >> 6][0] = s_; wsum[tid >> 6][1] = q_; }
  __syncthreads();
  if (tid == 0) {
    double s2 = 0.0, q2 = 0.0;
    for (int w = 0; w < NTHR / 64; ++w) { s2 += wsum[w][0]; q2 += wsum[w][1]; }
    const double mean = s2 / count, var = fmax(q2 / count - mean * mean, 0.0);
    st[0] = (float)mean;
    st[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
}

// DUAL: two layers on the same input in one launch (the reset- and the update-gate convolution of a ConvGRUCell2,
// reference models/module.py:72-92): 8 waves, waves 4-7 take the second layer's weights / bias / skip / output (`b`) over
// the window the first four staged with them; their GroupNorm partials go to group 1.
template <int D, int NTR, bool DUAL>
__global__ __launch_bounds__(DUAL ? 512 : 256) void k_conv_dd_resident(ConvDDArgs a0, ConvDDArgs a1, GruPro pro) {
  constexpr int WM = D / 16, WN = 4 / WM, TR = NTR * WN, KCT = D / 4, NTILES = D / 16;
  constexpr int LR = TR + 2, LC = 18, PLANE = plane_pitch16(LR * LC);
  constexpr int NTHR = DUAL ? 512 : 256;
  constexpr int NITEMS = LR * LC * KCT, NIT = (NITEMS + NTHR - 1) / NTHR;
  __shared__ float lds[D * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = DUAL ? wave8 >> 2 : 0, wave = wave8 & 3;
  const ConvDDArgs& a = half ? a1 : a0;
  const int wm = wave % WM, wn = wave / WM;
  const int p = lane & 15, q = lane >> 4;
  const int n = blockIdx.z, r0 = blockIdx.y * TR, c0 = blockIdx.x * 16;
  const int iy0 = r0 - 1, ix0 = c0 - 1;

  // all loads of the block, back to back: A fragments of the wave's channel tile, then the input window
  const buf_rsrc rw = make_rsrc(a.wpk);
  float wf[9][KCT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kc = 0; kc < KCT; ++kc)
      wf[t][kc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
          rw, (unsigned)(lane * 4), (unsigned)(((t * KCT + kc) * NTILES + wm) * 256), 0));
  const long wbase = (((long)n * a.hi + iy0) * a.wi + ix0) * (long)D * 4;
  const buf_rsrc rx = make_rsrc((const char*)a0.in + wbase);
  f32x4 xs[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = min(tid + it * NTHR, NITEMS - 1);
    const int g = i % KCT, pp = i / KCT, r = pp / LC, c = pp % LC;
    const bool ok = (unsigned)(iy0 + r) < (unsigned)a.hi && (unsigned)(ix0 + c) < (unsigned)a.wi;
    xs[it] = buf_load4(rx, ok ? (unsigned)(((r * a.wi + c) * D + 4 * g) * 4) : BUF_OOB);
  }
  if (pro.mode != GRU_PRO_NONE) {                            // uniform: the window is formed from the previous kernels' maps
    __shared__ double wsum[NTHR / 64][2];
    __shared__ float st[2][2];
    gru_pro_stats<NTHR>(pro.part_f + ((size_t)n * 2 + pro.group_f) * pro.parts_f * 2, pro.parts_f, pro.count, pro.eps, wsum, st[0]);
    if (pro.mode == GRU_PRO_OUT) gru_pro_stats<NTHR>(pro.part_o + (size_t)n * pro.parts_o * 2, pro.parts_o, pro.count, pro.eps, wsum, st[1]);
    const buf_rsrc rf = make_rsrc((const char*)pro.f + wbase);
    const buf_rsrc ro_ = make_rsrc((const char*)(pro.mode == GRU_PRO_OUT ? pro.o : pro.f) + wbase);
    const float mf = st[0][0], rf_ = st[0][1], mo = st[1][0], ro2 = st[1][1];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = min(tid + it * NTHR, NITEMS - 1);
      const int g = i % KCT, pp = i / KCT, r = pp / LC, c = pp % LC;
      const bool ok = (unsigned)(iy0 + r) < (unsigned)a.hi && (unsigned)(ix0 + c) < (unsigned)a.wi && 4 * g < pro.hc;
      const unsigned off = ok ? (unsigned)(((r * a.wi + c) * D + 4 * g) * 4) : BUF_OOB;
      const int c4 = min(4 * g, pro.hc - 4);
      const f32x4 fv = buf_load4(rf, off);
      const f32x4 gw = *(const f32x4*)(pro.gn_f + c4), gb = *(const f32x4*)(pro.gn_f + pro.hc + c4);
      const f32x4 fn = (fv - mf) * rf_ * gw + gb;
      const f32x4 sg = {sigmoidf_(fn.x), sigmoidf_(fn.y), sigmoidf_(fn.z), sigmoidf_(fn.w)};
      f32x4 v;
      if (pro.mode == GRU_PRO_GATES) {
        v = sg * xs[it];                                     // r * h
      } else {
        const f32x4 ov = buf_load4(ro_, off);
        const f32x4 ow = *(const f32x4*)(pro.gn_o + c4), ob = *(const f32x4*)(pro.gn_o + pro.hc + c4);
        const f32x4 on = (ov - mo) * ro2 * ow + ob;
        const f32x4 y = {tanhf(on.x), tanhf(on.y), tanhf(on.z), tanhf(on.w)};
        v = sg * xs[it] + (1.0f - sg) * y;                   // h' = u h + (1 - u) y
      }
      xs[it] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};           // zero padding / padding channels
      // the block's own pixels: the new state (and R)
      if (pro.mode == GRU_PRO_OUT && ok && tid + it * NTHR < NITEMS && r >= 1 && r <= TR && c >= 1 && c <= 16) {
        const size_t pix = ((size_t)n * a.hi + (iy0 + r)) * a.wi + (ix0 + c);
        *(f32x4*)(pro.state_out + pix * D + 4 * g) = xs[it];
        if (pro.R) *(f32x4*)(pro.R + pix * pro.RW + 4 * g) = xs[it];
      }
    }
  }
  if (a0.in2) {                                              // uniform: the layer convolves in + in2
    const buf_rsrc rx2 = make_rsrc((const char*)a0.in2 + (((long)n * a.hi + iy0) * a.wi + ix0) * (long)D * 4);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = min(tid + it * NTHR, NITEMS - 1);
      const int g = i % KCT, pp = i / KCT, r = pp / LC, c = pp % LC;
      const bool ok = (unsigned)(iy0 + r) < (unsigned)a.hi && (unsigned)(ix0 + c) < (unsigned)a.wi;
      xs[it] += buf_load4(rx2, ok ? (unsigned)(((r * a.wi + c) * D + 4 * g) * 4) : BUF_OOB);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = min(tid + it * NTHR, NITEMS - 1);
    const int g = i % KCT, pp = i / KCT;
    float* d = lds + 4 * g * PLANE + pp;
    d[0] = xs[it].x; d[PLANE] = xs[it].y; d[2 * PLANE] = xs[it].z; d[3 * PLANE] = xs[it].w;
  }
  __syncthreads();

  f32x4 acc[NTR];
#pragma unroll
  for (int r = 0; r < NTR; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* xb = lds + q * PLANE + (wn * NTR) * LC + p;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kc = 0; kc < KCT; ++kc)
#pragma unroll
      for (int r = 0; r < NTR; ++r)
        acc[r] = mfma16(wf[t][kc], xb[4 * kc * PLANE + (r + t / 3) * LC + t % 3], acc[r]);

  const int co4 = wm * 16 + 4 * q;
  const f32x4 bias = *(const f32x4*)(a.bias + co4);
#pragma unroll
  for (int r = 0; r < NTR; ++r) {
    const int oy = r0 + wn * NTR + r, ox = c0 + p;
    if (oy >= a.ho || ox >= a.wo) continue;
    const size_t opix = ((size_t)n * a.ho + oy) * a.wo + ox;
    f32x4 v = acc[r] + bias;
    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (a.skip) v += *(const f32x4*)(a.skip + opix * D + co4);
    *(f32x4*)(a.out + opix * D + co4) = v;
  }
  if (a.gn_part) {                                           // uniform
    double gs = 0.0, gq = 0.0;
#pragma unroll
    for (int r = 0; r < NTR; ++r) {
      const int oy = r0 + wn * NTR + r, ox = c0 + p;
      if (oy < a.ho && ox < a.wo && co4 < a.gn_n) {
        f32x4 v = acc[r] + bias;
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (a.skip) v += *(const f32x4*)(a.skip + (((size_t)n * a.ho + oy) * a.wo + ox) * D + co4);
        gs += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        gq += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
      }
    }
    for (int o = 32; o > 0; o >>= 1) { gs += __shfl_down(gs, o); gq += __shfl_down(gq, o); }             // fixed tree: deterministic
    if (lane == 0) {
      const int parts = gridDim.x * gridDim.y * 4;
      double* o = a.gn_part + (((size_t)n * a.gn_ngroups + (DUAL ? half : a.gn_group)) * parts + (blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 2;
      o[0] = gs; o[1] = gq;
    }
  }
}

static int small_grid_limit() { return opt(OPT_CONV_SMALL_GRID); }      // workgroups up to which the resident form is used (0 disables it)

template <int D, int NTR>
static int launch_conv_dd_resident_rows(const ConvDDArgs& a, int N, hipStream_t st, const GruPro& pro = GruPro{}) {
  constexpr int TR = NTR * (4 / (D / 16));
  hipLaunchKernelGGL((k_conv_dd_resident<D, NTR, false>), dim3(cdiv(a.wo, 16), cdiv(a.ho, TR), N), dim3(256), 0, st, a, a, pro);
  ADAMVS_CHECK_LAUNCH("conv_dd (resident)");
  return 0;
}
template <int D, int NTR>
static int launch_conv_dd_resident_dual(const ConvDDArgs& a, const ConvDDArgs& b, int N, hipStream_t st, const GruPro& pro = GruPro{}) {
  constexpr int TR = NTR * (4 / (D / 16));
  hipLaunchKernelGGL((k_conv_dd_resident<D, NTR, true>), dim3(cdiv(a.wo, 16), cdiv(a.ho, TR), N), dim3(512), 0, st, a, b, pro);
  ADAMVS_CHECK_LAUNCH("conv_dd (resident, two layers)");
  return 0;
}

// the fewest rows per workgroup (most workgroups) that stays within the limit; 0: grid too large for this form
template <int D>
static int resident_rows(const ConvDDArgs& a, int N) {
  constexpr int WN = 4 / (D / 16);
  const long cols = (long)cdiv(a.wo, 16) * N, limit = small_grid_limit();
  for (int ntr = 1; ntr <= 4; ntr *= 2)
    if (cols * cdiv(a.ho, ntr * WN) <= limit) return ntr;
  return 0;
}
template <int D>
static int launch_conv_dd_resident(const ConvDDArgs& a, int N, hipStream_t st) {
  switch (resident_rows<D>(a, N)) {
    case 1: return launch_conv_dd_resident_rows<D, 1>(a, N, st);
    case 2: return launch_conv_dd_resident_rows<D, 2>(a, N, st);
    case 4: return launch_conv_dd_resident_rows<D, 4>(a, N, st);
  }
  return -1;
}

static bool conv256_split() { return opt(OPT_CONV256_SPLIT) != 0; }

static int launch_conv_dd_z(const ConvDDArgs& a, int N, int mode, hipStream_t st);

// The kernels below take the image index from blockIdx.z (at most 65535): more images run as several launches over
// sub-batches, every pointer moved by whole images (the softmax epilogue's tile index is n % sm_B: sub-batches of whole tiles).
static int launch_conv_dd(const ConvDDArgs& a, int N, int mode, hipStream_t st) {
  constexpr int ZMAX = 65535;
  if (N <= ZMAX) return launch_conv_dd_z(a, N, mode, st);
  ADAMVS_CHECK_ARG(!a.gn_part, "conv_dd: GroupNorm partials with N=%d images (at most %d)", N, ZMAX);
  const int unit = a.sm_vw ? a.sm_B : 1;
  ADAMVS_CHECK_ARG(unit <= ZMAX, "conv_dd: softmax epilogue with %d tiles per view (at most %d)", unit, ZMAX);
  const int step = ZMAX / unit * unit;
  const size_t in_img = (size_t)a.hi * a.wi * a.D, out_img = (size_t)a.ho * a.wo * a.D, map = (size_t)a.ho * a.wo;
  for (int n0 = 0; n0 < N; n0 += step) {
    ConvDDArgs b = a;
    b.in = a.in + n0 * in_img;
    if (a.in2) b.in2 = a.in2 + n0 * in_img;
    if (a.out) b.out = a.out + n0 * out_img;
    if (a.skip) b.skip = a.skip + n0 * out_img;
    if (a.sm_vw) { b.sm_vw = a.sm_vw + n0 * map; b.sm_pd = a.sm_pd + n0 * map; }
    if (int rc = launch_conv_dd_z(b, N - n0 < step ? N - n0 : step, mode, st)) return rc;
  }
  return 0;
}

static int launch_conv_dd_z(const ConvDDArgs& a, int N, int mode, hipStream_t st) {
  if (mode == CONV_S1 && (a.D == 32 || a.D == 64) && !a.sm_vw) {
    const int rc = a.D == 32 ? launch_conv_dd_resident<32>(a, N, st) : launch_conv_dd_resident<64>(a, N, st);
    if (rc >= 0) return rc;
  }
  switch (a.D) {
    case 16: return launch_conv_dd_cfg<1, 1>(a, N, mode, st);
    case 32: return launch_conv_dd_cfg<2, 1>(a, N, mode, st);
    case 48: return launch_conv_dd_cfg<3, 1>(a, N, mode, st);
    case 64: return launch_conv_dd_cfg<4, 1>(a, N, mode, st);
    case 96: return launch_conv_dd_cfg<3, 2>(a, N, mode, st);
    case 128: return launch_conv_dd_cfg<4, 2>(a, N, mode, st);
    case 192: return launch_conv_dd_cfg<3, 4>(a, N, mode, st);
    case 256:
      // 16 output tiles per block need 256 accumulator registers: one wave per SIMD, which feeds the matrix pipe at 83 % (DESIGN
      // section 4, lesson 2; measured 74.5 % of the fp32 MFMA peak for the whole network at cfg5 against 88.5 % at D = 192).  The
      // 128-channel tiling (two waves per SIMD) run twice instead: the second launch sees the weights, bias, skip and output
      // moved by 8 channel tiles and contracts over all 256 input channels like the first.  The softmax epilogue needs all the
      // channels of a pixel in one workgroup and keeps the wide tiling.  ADAMVS_CONV256_SPLIT=0: the wide tiling everywhere.
      // Measured at cfg5 (8 tiles): 291.9 -> 279.5 ms per step.
      if (!a.sm_vw && conv256_split()) {
        ConvDDArgs h = a;
        for (int half = 0; half < 2; ++half) {
          h.wpk = a.wpk + (size_t)half * 8 * 64;       // fragment index = (... * D/16 + tile) * 64 floats
          h.bias = a.bias + half * 128;
          h.out = a.out + half * 128;
          h.skip = a.skip ? a.skip + half * 128 : nullptr;
          if (int rc = launch_conv_dd_cfg<4, 2>(h, N, mode, st)) return rc;
        }
        return 0;
      }
      return launch_conv_dd_cfg<4, 4>(a, N, mode, st);
    case 384: {
      // the 192-channel tiling twice (as 256 = 2 x 128 above): each launch contracts over all 384 input channels into its half
      // of the output channels.  No softmax epilogue at this width (it needs a pixel's channels in one workgroup): the scores
      // go through the score volume to k_softmax_regress.
      ADAMVS_CHECK_ARG(!a.sm_vw, "conv_dd: no softmax epilogue at D=384");
      ConvDDArgs h = a;
      for (int half = 0; half < 2; ++half) {
        h.wpk = a.wpk + (size_t)half * 12 * 64;
        h.bias = a.bias + half * 192;
        h.out = a.out + half * 192;
        h.skip = a.skip ? a.skip + half * 192 : nullptr;
        if (int rc = launch_conv_dd_cfg<3, 4>(h, N, mode, st)) return rc;
      }
      return 0;
    }
    case 512: {
      // the 128-channel tiling four times (round 5; the class default of the reference is num_depth = 384, adamvs.py:538, and
      // ndepths[0] is unconstrained): each launch contracts over all 512 input channels into its quarter of the output channels
      ADAMVS_CHECK_ARG(!a.sm_vw, "conv_dd: no softmax epilogue at D=512");
      ConvDDArgs h = a;
      for (int part = 0; part < 4; ++part) {
        h.wpk = a.wpk + (size_t)part * 8 * 64;
        h.bias = a.bias + part * 128;
        h.out = a.out + part * 128;
        h.skip = a.skip ? a.skip + part * 128 : nullptr;
        if (int rc = launch_conv_dd_cfg<4, 2>(h, N, mode, st)) return rc;
      }
      return 0;
    }
  }
  return set_error(-1, "cost_reg_net_2d: D=%d unsupported (16, 32, 48, 64, 96, 128, 192, 256, 384 or 512)", a.D);
}

int launch_conv_dd_gn(const float* in, const float* wpk, const float* bias, const float* skip, float* out, int N, int D, int h, int w,
                      hipStream_t st, double* gn_part, int gn_n, int gn_group, int gn_ngroups, int* gn_parts, const GruPro* prop) {
  ConvDDArgs a{in, wpk, bias, skip, out, D, h, w, h, w, 0, nullptr, nullptr, PlaneSrc{nullptr, 0, 0.f}, 1, nullptr};
  *gn_parts = 0;
  if (gn_part && (D == 32 || D == 64)) {
    const int ntr = D == 32 ? resident_rows<32>(a, N) : resident_rows<64>(a, N);
    const long parts = ntr ? (long)cdiv(w, 16) * cdiv(h, ntr * (4 / (D / 16))) * 4 : 0;
    if (gn_epilogue_partials(parts, N)) {
      a.gn_part = gn_part; a.gn_n = gn_n; a.gn_group = gn_group; a.gn_ngroups = gn_ngroups;
      *gn_parts = (int)parts;
      if (prop) {                                            // the caller has asked can_fold_gru_applies(): this branch is taken
        if (D == 32) {
          switch (ntr) { case 1: return launch_conv_dd_resident_rows<32, 1>(a, N, st, *prop);
                         case 2: return launch_conv_dd_resident_rows<32, 2>(a, N, st, *prop);
                         default: return launch_conv_dd_resident_rows<32, 4>(a, N, st, *prop); }
        }
        switch (ntr) { case 1: return launch_conv_dd_resident_rows<64, 1>(a, N, st, *prop);
                       case 2: return launch_conv_dd_resident_rows<64, 2>(a, N, st, *prop);
                       default: return launch_conv_dd_resident_rows<64, 4>(a, N, st, *prop); }
      }
    }
  }
  if (prop) return set_error(-1, "conv_dd: the folded GRU prologue needs the small-grid kernel (D=%d, %dx%d, N=%d)", D, h, w, N);
  return launch_conv_dd(a, N, CONV_S1, st);
}

// true when launch_conv_dd_gn / _gates_gn run a level's convolutions on the small-grid kernel with epilogue partials: the
// condition for folding the elementwise kernels of a ConvGRUCell2 into the convolutions' window fills (msred.hip)
bool can_fold_gru_applies(int N, int D, int h, int w) {
  if (D != 32 && D != 64) return false;
  ConvDDArgs a{nullptr, nullptr, nullptr, nullptr, nullptr, D, h, w, h, w, 0, nullptr, nullptr, PlaneSrc{nullptr, 0, 0.f}, 1, nullptr};
  const int ntr = D == 32 ? resident_rows<32>(a, N) : resident_rows<64>(a, N);
  const long parts = ntr ? (long)cdiv(w, 16) * cdiv(h, ntr * (4 / (D / 16))) * 4 : 0;
  return gru_fold_enabled(N) && gn_epilogue_partials(parts, N);
}

// Two stride-1 layers on the same input (the gate convolutions of a ConvGRUCell2) with their GroupNorm partial sums (groups
// 0 and 1 of 2): ONE launch when the small-grid kernel takes the map (*gn_parts > 0), otherwise two plain launches (*gn_parts = 0).
int launch_conv_dd_gates_gn(const float* in, const float* wpk_r, const float* bias_r, const float* skip_r, float* out_r,
                            const float* wpk_u, const float* bias_u, const float* skip_u, float* out_u, int N, int D, int h, int w,
                            hipStream_t st, double* gn_part, int gn_n, int* gn_parts, const GruPro* prop) {
  const GruPro pro = prop ? *prop : GruPro{};
  ConvDDArgs a{in, wpk_r, bias_r, skip_r, out_r, D, h, w, h, w, 0, nullptr, nullptr, PlaneSrc{nullptr, 0, 0.f}, 1, nullptr};
  ConvDDArgs b = a;
  b.wpk = wpk_u; b.bias = bias_u; b.skip = skip_u; b.out = out_u;
  *gn_parts = 0;
  if (gn_part && (D == 32 || D == 64)) {
    const int ntr = D == 32 ? resident_rows<32>(a, N) : resident_rows<64>(a, N);
    const long parts = ntr ? (long)cdiv(w, 16) * cdiv(h, ntr * (4 / (D / 16))) * 4 : 0;
    if (gn_epilogue_partials(parts, N)) {
      a.gn_part = b.gn_part = gn_part; a.gn_n = b.gn_n = gn_n; a.gn_ngroups = b.gn_ngroups = 2;
      *gn_parts = (int)parts;
      if (D == 32) {
        switch (ntr) { case 1: return launch_conv_dd_resident_dual<32, 1>(a, b, N, st, pro);
                       case 2: return launch_conv_dd_resident_dual<32, 2>(a, b, N, st, pro);
                       default: return launch_conv_dd_resident_dual<32, 4>(a, b, N, st, pro); }
      }
      switch (ntr) { case 1: return launch_conv_dd_resident_dual<64, 1>(a, b, N, st, pro);
                     case 2: return launch_conv_dd_resident_dual<64, 2>(a, b, N, st, pro);
                     default: return launch_conv_dd_resident_dual<64, 4>(a, b, N, st, pro); }
    }
  }
  if (prop) return set_error(-1, "conv_dd: the folded GRU prologue needs the small-grid kernel (D=%d, %dx%d, N=%d)", D, h, w, N);
  if (int rc = launch_conv_dd(a, N, CONV_S1, st)) return rc;
  return launch_conv_dd(b, N, CONV_S1, st);
}

// ---------------------------------------------------------------------------
// softmax over D, max probability, expectation of depth.  16 lanes per pixel.
// score [N=S*B][hw][D] (n = s*B + b), planes [B][D][hw] -> vw, pd [S*B][hw].
// NQ = D/64 rounded up: the lane's scores stay in registers between the max pass and the exp pass, so the
// score volume is read once (NQ = 0: any D, two passes over global memory).
// Dp = hypothesis planes (<= D, the channels of `score`: pad channels carry -1e30 and weigh exactly 0).
template <int NQ>
__global__ __launch_bounds__(256) void k_softmax_regress(const float* __restrict__ score, PlaneSrc planes,
                                                         float* __restrict__ vw, float* __restrict__ pd, int B, int D, int hw,
                                                         size_t npix, int Dp) {
  // persistent: the grid is the resident capacity and a workgroup walks the 16-pixel groups blk, blk + grid, ... -- as one
  // workgroup per group (590 000 of them at cfg2) the launch ran at the rate workgroups are dispatched, 2.2 TB/s
  constexpr int DMAX = NQ > 0 ? 64 * NQ : 1;
  __shared__ float lp[NQ > 0 ? DMAX * 17 : 1];
  const int l = threadIdx.x & 15;
  const size_t nblk = (npix + 15) / 16;
  for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
  size_t gp = blk * 16 + (threadIdx.x >> 4);
  const bool live = gp < npix;
  size_t pix = live ? gp : npix - 1;
  size_t n = pix / hw, pp = pix % hw;
  size_t b = n % B;
  const float* sc = score + pix * D;
  const PlaneLine pl = plane_line(planes, b, pp, Dp, hw);
  // The 16 pixels of a block are consecutive; read per lane, a plane value costs a whole 64-byte sector for 16 bytes
  // (lanes of a wave hold 16 different planes of 4 pixels: 4x the plane volume through L2).  When the block lies inside
  // one map the [D][16] patch is staged through LDS with full sectors instead (row pitch 17: 2-way at worst).
  const size_t gp0 = blk * 16;
  // (generated planes need no staging: a plane value is one multiply and one add)
  const bool staged = NQ > 0 && planes.mode == PLANES_EXPLICIT && gp0 + 16 <= npix && (gp0 % hw) + 16 <= (size_t)hw;     // uniform
  if (staged) {
    const float* src = planes.p + ((gp0 / hw) % B) * Dp * hw + gp0 % hw;
    for (int i = threadIdx.x; i < D * 16; i += 256) lp[(i >> 4) * 17 + (i & 15)] = src[(size_t)min(i >> 4, Dp - 1) * hw + (i & 15)];
    __syncthreads();
  }
  const float* lpp = lp + (threadIdx.x >> 4);
  float m = -INFINITY;
  f32x4 keep[NQ > 0 ? NQ : 1];
  if (NQ > 0) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int d = 4 * l + 64 * i;
      keep[i] = d < D ? *(const f32x4*)(sc + d) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      m = fmaxf(m, fmaxf(fmaxf(keep[i].x, keep[i].y), fmaxf(keep[i].z, keep[i].w)));
    }
  } else {
    for (int d = 4 * l; d < D; d += 64) {
      f32x4 v = *(const f32x4*)(sc + d);
      m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float se = 0.f, sd = 0.f;
  // explicit fused multiply-adds in a fixed order: the sum must not depend on where the plane values come from (staged
  // through LDS, read from the tensor or generated), which a contraction left to the compiler would not guarantee
  auto term = [&](f32x4 v, int d) {
    float e0 = __expf(v.x - m), e1 = __expf(v.y - m), e2 = __expf(v.z - m), e3 = __expf(v.w - m);
    se += (e0 + e1) + (e2 + e3);
    float p0, p1, p2, p3;
    if (staged) { p0 = lpp[d * 17]; p1 = lpp[(d + 1) * 17]; p2 = lpp[(d + 2) * 17]; p3 = lpp[(d + 3) * 17]; }
    else { p0 = plane_at_pad(planes, pl, d, hw); p1 = plane_at_pad(planes, pl, d + 1, hw); p2 = plane_at_pad(planes, pl, d + 2, hw); p3 = plane_at_pad(planes, pl, d + 3, hw); }
    sd = __fmaf_rn(e3, p3, __fmaf_rn(e2, p2, __fmaf_rn(e1, p1, __fmaf_rn(e0, p0, sd))));
  };
  if (NQ > 0) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int d = 4 * l + 64 * i;
      if (d < D) term(keep[i], d);
    }
  } else {
    for (int d = 4 * l; d < D; d += 64) term(*(const f32x4*)(sc + d), d);
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) { se += __shfl_xor(se, o, 64); sd += __shfl_xor(sd, o, 64); }
  if (live && l == 0) {
    vw[pix] = 1.0f / se;          // max_d softmax = exp(max - max) / sum
    pd[pix] = sd / se;
  }
  if (staged) __syncthreads();    // (uniform) the patch is refilled by the next group
  }
}

// ---------------------------------------------------------------------------
// Layer plan.  Packed weights: 11 layers x (9*D*D fragment floats + D bias floats), in the order
// conv0..conv6, conv7, conv9, conv11, prob; fp32 with D in {64, 128, 192, 256}: followed by the five stride-1 layers
// (conv0, conv2, conv4, conv6, prob) in the F(2x2, 3x3) form, 16*D*D floats each (costreg2d_wino.hip).
// Workspace: 3 * N*h*w*D floats.
// sm_vw != null: the `prob` layer reduces its scores over D in its epilogue (costreg_softmax.h) and writes the view
// weights / pair depths of softmax_max_regress directly (score is not written); the caller skips launch_softmax_regress.
bool cost_reg_softmax_fusable(int D, int precision, const PlaneSrc& planes) {
  (void)planes; (void)precision;      // any plane source, both precisions: one code path, so generated == materialised bit for bit
  return opt(OPT_FUSE_SOFTMAX) != 0 && D >= 16;
}

// option winograd = 0: the stride-1 layers on the direct kernel, as in rounds 1-2 (A/B)
bool cost_reg_winograd(int D, int precision) {
  return opt(OPT_WINOGRAD) != 0 && precision == PRECISION_FP32 && wino_depth_supported(D);
}

int launch_cost_reg_net_2d(const float* x, const float* wpk, float* ws, float* score, int N, int D, int h, int w,
                           int precision, hipStream_t st, float* sm_vw, float* sm_pd, const PlaneSrc* sm_planes, int sm_B, int n_planes) {
  // D = the width the network runs at; n_planes <= D hypothesis planes for the softmax (0: D)
  const bool fuse_softmax = sm_vw != nullptr;
  // the softmax epilogue of the direct kernels needs a pixel's channels in one workgroup: not at D = 384 (two launches)
  const bool epilogue_sm = fuse_softmax && D <= 256;
  const bool wino = cost_reg_winograd(D, precision);
  const size_t F = (size_t)N * h * w * D;
  const size_t LW = (size_t)9 * D * D + D;
  float* conv0 = ws;                  // F
  float* t1 = conv0 + F;              // F/4
  float* conv2 = t1 + F / 4;          // F/4
  float* t3 = conv2 + F / 4;          // F/16
  float* conv4 = t3 + F / 16;         // F/16
  float* t5 = conv4 + F / 16;         // F/64
  float* t6 = t5 + F / 64;            // F/64
  float* x7 = t6 + F / 64;            // F/16
  float* x9 = x7 + F / 16;            // F/4
  float* x11 = x9 + F / 4;            // F
  const int h2 = h / 2, w2 = w / 2, h4 = h / 4, w4 = w / 4, h8 = h / 8, w8 = w / 8;
  // The three skip additions of the hourglass (x = conv4 + conv7(x); x = conv2 + conv9(x); x = conv0 + conv11(x),
  // adamvs.py:233-236).  bf16x3: in the epilogue of the transposed layer that produces x (`skip`).  fp32: in the layer
  // that CONSUMES x (`in2`: the sum is formed when the window enters LDS, one chunk ahead of its use) -- the transposed
  // kernel's epilogue is then bias + ReLU + stores and nothing waits for it.  The same two addends, one fp32 add: same bits.
  struct L { const float* in; float* out; const float* skip; int hi, wi, ho, wo, mode, relu; };
  const L plan[11] = {
      {x, conv0, nullptr, h, w, h, w, CONV_S1, 1},
      {conv0, t1, nullptr, h, w, h2, w2, CONV_S2, 1},
      {t1, conv2, nullptr, h2, w2, h2, w2, CONV_S1, 1},
      {conv2, t3, nullptr, h2, w2, h4, w4, CONV_S2, 1},
      {t3, conv4, nullptr, h4, w4, h4, w4, CONV_S1, 1},
      {conv4, t5, nullptr, h4, w4, h8, w8, CONV_S2, 1},
      {t5, t6, nullptr, h8, w8, h8, w8, CONV_S1, 1},
      {t6, x7, conv4, h8, w8, h4, w4, CONV_T2, 1},
      {x7, x9, conv2, h4, w4, h2, w2, CONV_T2, 1},
      {x9, x11, conv0, h2, w2, h, w, CONV_T2, 1},
      {x11, score, nullptr, h, w, h, w, CONV_S1, 0},
  };
  // deferred where the CONSUMER is a transposed layer (conv7 -> conv9 -> conv11: its window is small and re-read per
  // class anyway); conv11 keeps its own skip: in the stride-1 kernel of `prob` a second input costs more than the epilogue
  // of conv11 gains (measured at cfg2: prob 42.9 -> 45.4 ms against conv11 13.6 -> 12.3)
  const bool defer = precision == PRECISION_FP32 && costreg_deferred_skips();
  for (int i = 0; i < 11; ++i) {
    const float* wl = wpk + (size_t)i * LW;
    int rc;
    if (precision == PRECISION_BF16X3) {
      const bool sm = i == 10 && epilogue_sm;
      rc = launch_conv_dd_bf16x3(plan[i].in, wl, wl + (size_t)9 * D * D, plan[i].skip, plan[i].out, N, D, plan[i].hi,
                                 plan[i].wi, plan[i].ho, plan[i].wo, plan[i].mode, plan[i].relu, st, sm ? sm_vw : nullptr,
                                 sm ? sm_pd : nullptr, sm ? sm_planes : nullptr, sm_B, n_planes);
      if (!rc && i == 10 && fuse_softmax && !epilogue_sm)
        rc = launch_softmax_regress(score, *sm_planes, sm_vw, sm_pd, N / sm_B, sm_B, D, h, w, st, n_planes);
    } else if (wino && plan[i].mode == CONV_S1) {
      // stride-1 layers in the minimal-filtering form (none of them carries a skip or takes a second input).  The channel groups of
      // a pixel are different workgroups there, so the softmax behind the last layer is every lane's partial + a merge kernel
      // (uniform planes: stage 1); with per-pixel planes, or ADAMVS_WINO_SOFTMAX=0, the scores go through the score volume to
      // k_softmax_regress
      const float* ww = wpk + (size_t)11 * LW + (size_t)(i == 10 ? 4 : i / 2) * 16 * D * D;
      if (i == 10 && fuse_softmax && wino_softmax_fused() && sm_planes->mode == PLANES_UNIFORM && (n_planes > 0 ? n_planes : D) > 1) {
        // every lane's softmax partial instead of the scores (D bytes per pixel instead of 4 D), in the score volume's place
        rc = launch_conv_wino_softmax(plan[i].in, ww, wl + (size_t)9 * D * D, score, *sm_planes, sm_vw, sm_pd, N, sm_B, D, n_planes, h, w, st);
      } else {
        rc = launch_conv_wino(plan[i].in, ww, wl + (size_t)9 * D * D, nullptr, plan[i].out, N, D, plan[i].hi, plan[i].wi, plan[i].relu, st);
        if (!rc && i == 10 && fuse_softmax) rc = launch_softmax_regress(score, *sm_planes, sm_vw, sm_pd, N / sm_B, sm_B, D, h, w, st, n_planes);
      }
    } else {
      const bool give = defer && i + 1 < 11 && plan[i + 1].mode == CONV_T2;        // layer i + 1 adds this layer's skip to its input
      const bool take = defer && i > 0 && plan[i].mode == CONV_T2 && plan[i - 1].skip;
      ConvDDArgs a{plan[i].in, wl, wl + (size_t)9 * D * D, give ? nullptr : plan[i].skip, plan[i].out, D,
                   plan[i].hi, plan[i].wi, plan[i].ho, plan[i].wo, plan[i].relu, nullptr, nullptr, PlaneSrc{nullptr, 0, 0.f}, 1,
                   take ? plan[i - 1].skip : nullptr};
      if (i == 10 && epilogue_sm) { a.sm_vw = sm_vw; a.sm_pd = sm_pd; a.sm_planes = *sm_planes; a.sm_B = sm_B; a.sm_D = n_planes; }
      rc = launch_conv_dd(a, N, plan[i].mode, st);
      if (!rc && i == 10 && fuse_softmax && !epilogue_sm)
        rc = launch_softmax_regress(score, *sm_planes, sm_vw, sm_pd, N / sm_B, sm_B, D, h, w, st, n_planes);
    }
    if (rc) return rc;
  }
  return 0;
}

template <int NQ>
static void launch_softmax_nq(const float* score, const PlaneSrc& planes, float* vw, float* pd, int B, int D, int hw, size_t npix, int Dp,
                              hipStream_t st) {
  static const int capacity = resident_blocks(k_softmax_regress<NQ>, 256, 0);     // once per instantiation
  const size_t nblk = (npix + 15) / 16;
  hipLaunchKernelGGL(k_softmax_regress<NQ>, dim3((unsigned)(nblk < (size_t)capacity ? nblk : (size_t)capacity)), dim3(256), 0, st, score,
                     planes, vw, pd, B, D, hw, npix, Dp);
}

int launch_softmax_regress(const float* score, PlaneSrc planes, float* vw, float* pd, int S, int B, int D, int h, int w,
                           hipStream_t st, int n_planes) {
  size_t npix = (size_t)S * B * h * w;
  const int Dp = n_planes > 0 ? n_planes : D;
  ADAMVS_CHECK_ARG(Dp <= D, "softmax_regress: %d planes for %d score channels", Dp, D);
  switch ((D + 63) / 64) {
    case 1: launch_softmax_nq<1>(score, planes, vw, pd, B, D, h * w, npix, Dp, st); break;
    case 2: launch_softmax_nq<2>(score, planes, vw, pd, B, D, h * w, npix, Dp, st); break;
    case 3: launch_softmax_nq<3>(score, planes, vw, pd, B, D, h * w, npix, Dp, st); break;
    case 4: launch_softmax_nq<4>(score, planes, vw, pd, B, D, h * w, npix, Dp, st); break;
    case 6: launch_softmax_nq<6>(score, planes, vw, pd, B, D, h * w, npix, Dp, st); break;
    case 8: launch_softmax_nq<8>(score, planes, vw, pd, B, D, h * w, npix, Dp, st); break;
    default: launch_softmax_nq<0>(score, planes, vw, pd, B, D, h * w, npix, Dp, st); break;
  }
  ADAMVS_CHECK_LAUNCH("softmax_regress");
  return 0;
}

}  // namespace adamvs

using namespace adamvs;

extern "C" size_t adamvs_cost_reg_net_2d_workspace_bytes(int N, int D, int h, int w) {
  return (size_t)3 * N * h * w * D * sizeof(float);
}

static int check_precision(int precision, int D, const char* who) {
  ADAMVS_CHECK_ARG(precision == PRECISION_FP32 || precision == PRECISION_BF16X3, "%s: precision=%d (0 fp32, 1 bf16x3)", who, precision);
  ADAMVS_CHECK_ARG(precision == PRECISION_FP32 || costreg_bf16x3_depth_supported(D),
                   "%s: bf16x3 needs D in {32,64,96,128,192,256,384,512}, got %d", who, D);
  return 0;
}

extern "C" int adamvs_cost_reg_width(int D, int precision) {
  return D < 1 ? 0 : (precision == PRECISION_BF16X3 ? costreg_width_bf16x3(D) : costreg_width(D));
}

size_t adamvs::cost_reg_weight_floats(int D, int precision) {
  if (!(precision == PRECISION_BF16X3 ? costreg_bf16x3_depth_supported(D) : costreg_depth_supported(D))) return 0;
  return (size_t)11 * ((size_t)9 * D * D + D) + (precision == PRECISION_FP32 && wino_depth_supported(D) ? (size_t)5 * 16 * D * D : 0);
}
extern "C" size_t adamvs_cost_reg_net_2d_weight_floats(int D, int precision) { return cost_reg_weight_floats(D, precision); }

extern "C" int adamvs_cost_reg_net_2d(const float* x, const float* wpk, size_t wpk_floats, float* score, int N, int D, int h, int w,
                                      int precision, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_precision(precision, D, "cost_reg_net_2d")) return rc;
  ADAMVS_CHECK_ARG(x && wpk && score && workspace && N > 0 && h > 0 && w > 0, "cost_reg_net_2d: bad arguments");
  ADAMVS_CHECK_ARG(costreg_depth_supported(D), "cost_reg_net_2d: D=%d unsupported (16, 32, 48, 64, 96, 128, 192, 256, 384 or 512)", D);
  ADAMVS_CHECK_ARG(wpk_floats == cost_reg_weight_floats(D, precision),
                   "cost_reg_net_2d: wpk holds %zu floats, the layout for D=%d precision=%d has %zu (include/adamvs_hip.h)", wpk_floats, D,
                   precision, cost_reg_weight_floats(D, precision));
  ADAMVS_CHECK_ARG((h % 8) == 0 && (w % 8) == 0, "cost_reg_net_2d: h=%d w=%d must be multiples of 8 (three stride-2 levels)", h, w);
  ADAMVS_CHECK_ARG(workspace_bytes >= adamvs_cost_reg_net_2d_workspace_bytes(N, D, h, w),
                   "cost_reg_net_2d: workspace too small (%zu < %zu bytes)", workspace_bytes,
                   adamvs_cost_reg_net_2d_workspace_bytes(N, D, h, w));
  return launch_cost_reg_net_2d(x, wpk, (float*)workspace, score, N, D, h, w, precision, (hipStream_t)stream);
}

extern "C" int adamvs_conv3x3_dd(const float* in, const float* in2, const float* wpk, const float* bias, const float* skip,
                                 float* out, int N, int D, int hi, int wi, int mode, int relu, int precision, void* stream) {
  if (int rc = check_precision(precision, D, "conv3x3_dd")) return rc;
  ADAMVS_CHECK_ARG(in && wpk && bias && out && N > 0 && hi > 0 && wi > 0, "conv3x3_dd: bad arguments");
  ADAMVS_CHECK_ARG(costreg_depth_supported(D), "conv3x3_dd: D=%d unsupported (16, 32, 48, 64, 96, 128, 192, 256, 384 or 512)", D);
  ADAMVS_CHECK_ARG(mode >= 0 && mode <= 2, "conv3x3_dd: mode=%d (0 stride 1, 1 stride 2, 2 transposed stride 2)", mode);
  ADAMVS_CHECK_ARG(mode != CONV_S2 || ((hi % 2) == 0 && (wi % 2) == 0), "conv3x3_dd: stride 2 needs even hi, wi");
  int ho = mode == CONV_S2 ? hi / 2 : (mode == CONV_T2 ? 2 * hi : hi);
  int wo = mode == CONV_S2 ? wi / 2 : (mode == CONV_T2 ? 2 * wi : wi);
  ADAMVS_CHECK_ARG(!(in2 && (precision == PRECISION_BF16X3 || mode == CONV_S2)),
                   "conv3x3_dd: in2 is implemented for fp32, modes 0 and 2 (otherwise pass the addition as the producer's skip)");
  if (precision == PRECISION_BF16X3)
    return launch_conv_dd_bf16x3(in, wpk, bias, skip, out, N, D, hi, wi, ho, wo, mode, relu, (hipStream_t)stream);
  ConvDDArgs a{in, wpk, bias, skip, out, D, hi, wi, ho, wo, relu, nullptr, nullptr, PlaneSrc{nullptr, 0, 0.f}, 1, in2};
  return launch_conv_dd(a, N, mode, (hipStream_t)stream);
}

extern "C" int adamvs_prob_softmax_regress(const float* in, const float* wpk, const float* bias, const float* planes,
                                           float* view_weight, float* pair_depth, int S, int B, int D, int h, int w, int precision,
                                           void* stream) {
  if (int rc = check_precision(precision, D, "prob_softmax_regress")) return rc;
  ADAMVS_CHECK_ARG(in && wpk && bias && planes && view_weight && pair_depth && S > 0 && B > 0 && h > 0 && w > 0,
                   "prob_softmax_regress: bad arguments");
  ADAMVS_CHECK_ARG(costreg_depth_supported(D) && D <= 256, "prob_softmax_regress: D=%d unsupported (16, 32, 48, 64, 96, 128, 192 or 256)", D);
  const PlaneSrc ps = explicit_planes(planes);
  if (precision == PRECISION_BF16X3)
    return launch_conv_dd_bf16x3(in, wpk, bias, nullptr, nullptr, S * B, D, h, w, h, w, CONV_S1, 0, (hipStream_t)stream, view_weight,
                                 pair_depth, &ps, B);
  ConvDDArgs a{in, wpk, bias, nullptr, nullptr, D, h, w, h, w, 0, view_weight, pair_depth, ps, B, nullptr};
  return launch_conv_dd(a, S * B, CONV_S1, (hipStream_t)stream);
}

extern "C" int adamvs_softmax_max_regress(const float* score, const float* planes, float* view_weight, float* pair_depth,
                                          int S, int B, int D, int h, int w, void* stream) {
  ADAMVS_CHECK_ARG(score && planes && view_weight && pair_depth && S > 0 && B > 0 && D > 0 && (D % 4) == 0 && h > 0 && w > 0,
                   "softmax_max_regress: bad arguments (D=%d must be a multiple of 4)", D);
  return launch_softmax_regress(score, explicit_planes(planes), view_weight, pair_depth, S, B, D, h, w, (hipStream_t)stream);
}
