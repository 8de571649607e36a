// MS-REDNet inference pieces that are not 3x3 convolutions (reference models/msrednet.py:373-436,
// models/module.py:54-106): the variance cost of one hypothesis plane, the
// GroupNorm(1 group) statistics and the two fused gate / candidate epilogues of ConvGRUCell2, a strided
// channel-range copy, and the online soft-argmin over the stored log-probabilities.
// All maps are channel-last [N][pixels][D] with the channel count padded to what adamvs_conv3x3_dd takes;
// the convolutions themselves run on k_conv_dd (costreg2d.hip).  These kernels stream: HBM-bound by design.
#include "../../include/adamvs_hip.h"
#include "common.h"
#include "kernels.h"
#include "warp_math.h"

namespace adamvs {

// One thread = one plane x one reference pixel x one group of 4 channels.  mean and mean of squares over the
// reference feature and the S warped source features (bilinear, zero padding per tap: module.py:563-564), then
// E[x^2] - E[x]^2, negated when `negate` (both consumers of the cost take -cost, msrednet.py:351,362).
// Output map index = d * B + b (plane-major: the planes of one step are contiguous).
__global__ void k_red_variance(const float* __restrict__ feat, const float* __restrict__ rt, const float* __restrict__ planes,
                               float* __restrict__ out_a, float* __restrict__ out_b, int B, int S, int C, int D, int h, int w,
                               int Da, int Db, float sign, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = C >> 2;
  const int g = (int)(i % G);
  const size_t hw = (size_t)h * w;
  const size_t dbp = i / G;                      // (d * B + b) * hw + pixel
  const int pix = (int)(dbp % hw);
  const size_t db = dbp / hw;
  const size_t b = db % B, dd = db / B;
  const size_t bp = b * hw + pix;
  const int x = pix % w, y = pix / w;
  const size_t vstride = (size_t)B * hw * C;
  const f32x4 r4 = *(const f32x4*)(feat + bp * C + 4 * g);
  f32x4 sum = r4, sq = r4 * r4;
  const float d = planes[(b * D + dd) * hw + pix];
  for (int s = 0; s < S; ++s) {
    const WarpTaps tp = warp_taps(rt + (b * S + s) * 12, (float)x, (float)y, d, h, w);
    const float* src = feat + (size_t)(s + 1) * vstride + b * hw * C + 4 * g;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (tp.w00 != 0.f) v += tp.w00 * *(const f32x4*)(src + (size_t)tp.o00 * C);
    if (tp.w01 != 0.f) v += tp.w01 * *(const f32x4*)(src + (size_t)tp.o01 * C);
    if (tp.w10 != 0.f) v += tp.w10 * *(const f32x4*)(src + (size_t)tp.o10 * C);
    if (tp.w11 != 0.f) v += tp.w11 * *(const f32x4*)(src + (size_t)tp.o11 * C);
    sum += v;
    sq += v * v;
  }
  const float inv = 1.0f / (float)(S + 1);
  const f32x4 m = sum * inv;
  const f32x4 var = (sq * inv - m * m) * sign;
  *(f32x4*)(out_a + dbp * Da + 4 * g) = var;
  if (out_b) *(f32x4*)(out_b + dbp * Db + 4 * g) = var;
}

template <typename T>       // float, or f32x4 when the channel count, both offsets and both pixel strides are multiples of 4
__global__ void k_channel_copy(const float* __restrict__ src, float* __restrict__ dst, int npix, int n, long sbs, int sps, int os,
                               long dbs, int dps, int od, size_t total) {
  constexpr int V = sizeof(T) / sizeof(float);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int nv = n / V;
  const int c = (int)(i % nv) * V;
  const size_t bp = i / nv;
  const int p = (int)(bp % npix);
  const size_t b = bp / npix;
  *(T*)(dst + b * dbs + (size_t)p * dps + od + c) = *(const T*)(src + b * sbs + (size_t)p * sps + os + c);
}

// ---- GroupNorm(1 group) statistics, deterministic: fixed partial ranges, double accumulation.
// k_gn_partial: grid (parts, ngroups, N); group g = map x0 / x1, channels [0, n) of a D-wide map; parts = gn_parts(npix)
// pixel ranges per (sample, group): 64 for the small maps of the deep levels, up to 1024 so that a full-resolution
// map (5 M pixels at the reference's predict size) is not reduced by 64 workgroups.  The consumers (the two epilogue
// kernels below) finish the reduction themselves: one fewer dependent launch per normalisation.
constexpr int GN_PARTS_MAX = GN_PARTS_LIMIT;
__host__ __device__ inline int gn_parts(int npix) {
  const int want = (npix + 2047) / 2048;
  return want < 64 ? 64 : (want > GN_PARTS_MAX ? GN_PARTS_MAX : want);
}

__global__ __launch_bounds__(256) void k_gn_partial(const float* __restrict__ x0, const float* __restrict__ x1,
                                                     double* __restrict__ part, int npix, int D, int n) {
  const int g = blockIdx.y, b = blockIdx.z, ngroups = gridDim.y, parts = gridDim.x;
  const float* x = g ? x1 : x0;
  const int per = (npix + parts - 1) / parts;
  const int p0 = min(npix, (int)blockIdx.x * per), p1 = min(npix, p0 + per);
  const int n4 = n >> 2;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < (p1 - p0) * n4; i += 256) {
    const int p = p0 + i / n4, c = 4 * (i % n4);
    const f32x4 v = *(const f32x4*)(x + ((size_t)b * npix + p) * D + c);
    s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  __shared__ double sh[2][256];
  sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = q;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) { sh[0][threadIdx.x] += sh[0][threadIdx.x + k]; sh[1][threadIdx.x] += sh[1][threadIdx.x + k]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* o = part + (((size_t)b * ngroups + g) * parts + blockIdx.x) * 2;
    o[0] = sh[0][0]; o[1] = sh[1][0];
  }
}

// mean and 1/sqrt(biased var + eps) of the groups of sample b from the partial sums, into shared memory.  Every thread of
// the block sums partials tid, tid + blockDim, ... in order, a fixed shuffle tree joins the lanes of a wave and thread 0 the
// waves in order -- the same association every time.  (A convolution epilogue writes up to 2048 partials per group: summed by
// one wave alone they cost every block of the consumer 18 dependent trips to L2.)
__device__ __forceinline__ void gn_finish(const double* __restrict__ part, int b, int ngroups, int parts, int count, float eps,
                                          float (*st)[2]) {
  __shared__ double wsum[2][4][2];
  const int tid = threadIdx.x, nw = (blockDim.x + 63) >> 6;
  for (int g = 0; g < ngroups; ++g) {
    const double* p = part + ((size_t)b * ngroups + g) * parts * 2;
    double s = 0.0, q = 0.0;
    for (int k = tid; k < parts; k += blockDim.x) { s += p[2 * k]; q += p[2 * k + 1]; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); q += __shfl_down(q, o); }
    if ((tid & 63) == 0) { wsum[g][tid >> 6][0] = s; wsum[g][tid >> 6][1] = q; }
  }
  __syncthreads();
  if (tid < ngroups) {
    double s = 0.0, q = 0.0;
    for (int w = 0; w < nw; ++w) { s += wsum[tid][w][0]; q += wsum[tid][w][1]; }
    const double mean = s / count;
    const double var = fmax(q / count - mean * mean, 0.0);    // biased, as torch.nn.GroupNorm
    st[tid][0] = (float)mean;
    st[tid][1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
}

__global__ void k_gn_final(const double* __restrict__ part, float* __restrict__ stats, int ngroups, int parts, int count,
                           float eps) {
  __shared__ float st[2][2];
  gn_finish(part, blockIdx.x, ngroups, parts, count, eps, st);
  if ((int)threadIdx.x < ngroups) {
    stats[(blockIdx.x * ngroups + threadIdx.x) * 2] = st[threadIdx.x][0];
    stats[(blockIdx.x * ngroups + threadIdx.x) * 2 + 1] = st[threadIdx.x][1];
  }
}

// gates (module.py:72-92): fr / fu = the reset / update halves of gate_conv(cat(x, h)), Wf-wide maps with HC real
// channels (two maps, or the two halves of one); r = sigmoid(GN(fr)), u = sigmoid(GN(fu)); rh = r * h (h, rh W-wide,
// HC real), u out [N][npix][HC].
// grid (blocks over npix * HC/4, N).
__global__ void k_gru2_gates_apply(const float* __restrict__ fr, const float* __restrict__ fu, const double* __restrict__ part,
                                   const float* __restrict__ gn, const float* __restrict__ h, float* __restrict__ rh,
                                   float* __restrict__ u, int npix, int Wf, int W, int HC, float eps, int parts) {
  __shared__ float st[2][2];
  const int b = blockIdx.y;
  gn_finish(part, b, 2, parts, npix * HC, eps, st);
  const int G = HC >> 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * G) return;
  const int c = 4 * (i % G);
  const size_t bp = (size_t)b * npix + i / G;
  const f32x4 a = *(const f32x4*)(fr + bp * Wf + c), v = *(const f32x4*)(fu + bp * Wf + c);
  const f32x4 gr = *(const f32x4*)(gn + c), br = *(const f32x4*)(gn + HC + c);
  const f32x4 gu = *(const f32x4*)(gn + 2 * HC + c), bu = *(const f32x4*)(gn + 3 * HC + c);
  const f32x4 rn = (a - st[0][0]) * st[0][1] * gr + br, un = (v - st[1][0]) * st[1][1] * gu + bu;
  const f32x4 r = {sigmoidf_(rn.x), sigmoidf_(rn.y), sigmoidf_(rn.z), sigmoidf_(rn.w)};
  const f32x4 uu = {sigmoidf_(un.x), sigmoidf_(un.y), sigmoidf_(un.z), sigmoidf_(un.w)};
  *(f32x4*)(rh + bp * W + c) = r * *(const f32x4*)(h + bp * W + c);
  *(f32x4*)(u + bp * HC + c) = uu;
}

// candidate + blend (module.py:91-106): y = tanh(GN(o)); h' = u*h + (1-u)*y, in place on the W-wide state map and
// into channels [0, HC) of out [N][npix][Wo] (the decoder's input).  grid (blocks over npix * HC/4, N).
__global__ void k_gru2_out_apply(const float* __restrict__ o, const double* __restrict__ part, const float* __restrict__ gn,
                                 const float* __restrict__ u, float* __restrict__ h, float* __restrict__ out, int npix, int W,
                                 int HC, int Wo, float eps, int parts) {
  __shared__ float st[2][2];
  const int b = blockIdx.y;
  gn_finish(part, b, 1, parts, npix * HC, eps, st);
  const int G = HC >> 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * G) return;
  const int c = 4 * (i % G);
  const size_t bp = (size_t)b * npix + i / G;
  const f32x4 ov = *(const f32x4*)(o + bp * W + c);
  const f32x4 ga = *(const f32x4*)(gn + c), be = *(const f32x4*)(gn + HC + c);
  const f32x4 on = (ov - st[0][0]) * st[0][1] * ga + be;
  const f32x4 y = {tanhf(on.x), tanhf(on.y), tanhf(on.z), tanhf(on.w)};
  const f32x4 u4 = *(const f32x4*)(u + bp * HC + c);
  float* hp = h + bp * W + c;
  const f32x4 hn = u4 * *(const f32x4*)hp + (1.0f - u4) * y;
  *(f32x4*)hp = hn;
  if (out) *(f32x4*)(out + bp * Wo + c) = hn;
}

// The last plane of a level whose elementwise kernels are folded into the convolutions (adamvs_red_recur_split below): there
// is no next gate convolution to form h' in its window fill.  u = sigmoid(GN_u(fu)), y = tanh(GN_o(o)), h' = u h + (1 - u) y
// -> channels [0, HC) of out.  gn [6][HC] as adamvs_red_recur_*.  grid (blocks over npix * HC/4, N).
__global__ void k_gru2_last_apply(const float* __restrict__ o, const float* __restrict__ fu, const double* __restrict__ part_o, int parts_o,
                                  const double* __restrict__ part_f, int parts_f, const float* __restrict__ gn,
                                  const float* __restrict__ h, float* __restrict__ out, int npix, int Wfu, int W, int HC, int Wo,
                                  float eps) {        // fu: Wfu floats per pixel; o, h: W; out: Wo
  __shared__ float sf[2][2], so[2][2];
  const int b = blockIdx.y;
  gn_finish(part_f, b, 2, parts_f, npix * HC, eps, sf);
  gn_finish(part_o, b, 1, parts_o, npix * HC, eps, so);
  const int G = HC >> 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * G) return;
  const int c = 4 * (i % G);
  const size_t bp = (size_t)b * npix + i / G;
  const f32x4 fv = *(const f32x4*)(fu + bp * Wfu + c), ov = *(const f32x4*)(o + bp * W + c);
  const f32x4 gu = *(const f32x4*)(gn + 2 * HC + c), bu = *(const f32x4*)(gn + 3 * HC + c);
  const f32x4 ga = *(const f32x4*)(gn + 4 * HC + c), be = *(const f32x4*)(gn + 5 * HC + c);
  const f32x4 un = (fv - sf[1][0]) * sf[1][1] * gu + bu, on = (ov - so[0][0]) * so[0][1] * ga + be;
  const f32x4 u4 = {sigmoidf_(un.x), sigmoidf_(un.y), sigmoidf_(un.z), sigmoidf_(un.w)};
  const f32x4 y = {tanhf(on.x), tanhf(on.y), tanhf(on.z), tanhf(on.w)};
  *(f32x4*)(out + bp * Wo + c) = u4 * *(const f32x4*)(h + bp * W + c) + (1.0f - u4) * y;
}

// `parts` partial sums per (sample, group): gn_parts(npix) from k_gn_partial, or what the producing convolution's epilogue wrote
static int launch_gates_apply(const float* fr, const float* fu, int Wf, const double* partials, int parts, const float* gn,
                              const float* h, float* rh, float* u, int N, int npix, int W, int HC, float eps, hipStream_t st) {
  hipLaunchKernelGGL(k_gru2_gates_apply, dim3(cdiv(npix * (HC / 4), 256), N), dim3(256), 0, st, fr, fu, partials, gn, h, rh, u, npix,
                     Wf, W, HC, eps, parts);
  ADAMVS_CHECK_LAUNCH("gru2_gates_apply");
  return 0;
}
static int launch_out_apply(const float* o, const double* partials, int parts, const float* gn, const float* u, float* h, float* out,
                            int Wo, int N, int npix, int W, int HC, float eps, hipStream_t st) {
  hipLaunchKernelGGL(k_gru2_out_apply, dim3(cdiv(npix * (HC / 4), 256), N), dim3(256), 0, st, o, partials, gn, u, h, out, npix, W, HC,
                     Wo, eps, parts);
  ADAMVS_CHECK_LAUNCH("gru2_out_apply");
  return 0;
}

}  // namespace adamvs

// =====================================================================================================================
using namespace adamvs;

extern "C" int adamvs_red_variance_cost(const float* feat, const float* rt, const float* planes, float* out_a, int Da,
                                        float* out_b, int Db, int B, int S, int C, int D, int h, int w, int negate,
                                        void* stream) {
  ADAMVS_CHECK_ARG(feat && rt && planes && out_a && B > 0 && S > 0 && C > 0 && (C % 4) == 0 && D > 0 && h > 0 && w > 0 &&
                   Da >= C && (Da % 4) == 0 && (!out_b || (Db >= C && (Db % 4) == 0)),
                   "red_variance_cost: bad arguments (B=%d S=%d C=%d D=%d h=%d w=%d Da=%d Db=%d)", B, S, C, D, h, w, Da, Db);
  // the register-resident-tap sweep of the Ada-MVS aggregation in its variance mode (sweep.hip): the feature maps are read
  // about once per pixel instead of once per pixel and plane (this kernel below gathered 116 GB through L2 for 16
  // tiles of stage 1).  It writes the negated variance; other view counts / widths / sign take the plain kernel.
  // Its plane loop is sequential inside a thread: worth it once the pixels alone fill the chip (one small tile: 9.3 vs 8.7 ms).
  if (negate && S <= 8 && (C == 8 || C == 16 || C == 32) && (size_t)B * h * w >= 65536)
    return launch_sweep_variance(feat, rt, planes, out_a, Da, out_b, Db, B, S, C, D, h, w, (hipStream_t)stream);
  const size_t total = (size_t)D * B * h * w * (C / 4);
  ADAMVS_CHECK_ARG(total / 256 < 0x7fffffffu, "red_variance_cost: too many planes x pixels for one launch");
  hipLaunchKernelGGL(k_red_variance, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feat, rt, planes,
                     out_a, out_b, B, S, C, D, h, w, Da, Db, negate ? -1.0f : 1.0f, total);
  ADAMVS_CHECK_LAUNCH("red_variance_cost");
  return 0;
}

extern "C" int adamvs_channel_copy(const float* src, float* dst, int nbatch, int npix, int n, long src_batch_stride,
                                   int src_pix_stride, int src_c0, long dst_batch_stride, int dst_pix_stride, int dst_c0,
                                   void* stream) {
  ADAMVS_CHECK_ARG(src && dst && nbatch > 0 && npix > 0 && n > 0 && src_c0 >= 0 && dst_c0 >= 0, "channel_copy: bad arguments");
  const bool vec = (n % 4) == 0 && (src_c0 % 4) == 0 && (dst_c0 % 4) == 0 && (src_pix_stride % 4) == 0 && (dst_pix_stride % 4) == 0 &&
                   (src_batch_stride % 4) == 0 && (dst_batch_stride % 4) == 0 && ((size_t)src % 16) == 0 && ((size_t)dst % 16) == 0;
  const size_t total = (size_t)nbatch * npix * (vec ? n / 4 : n);
  if (vec)
    hipLaunchKernelGGL(k_channel_copy<f32x4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, npix,
                       n, src_batch_stride, src_pix_stride, src_c0, dst_batch_stride, dst_pix_stride, dst_c0, total);
  else
    hipLaunchKernelGGL(k_channel_copy<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, npix,
                       n, src_batch_stride, src_pix_stride, src_c0, dst_batch_stride, dst_pix_stride, dst_c0, total);
  ADAMVS_CHECK_LAUNCH("channel_copy");
  return 0;
}

extern "C" size_t adamvs_group_stats_workspace_bytes(int N, int ngroups) {
  return (size_t)N * ngroups * GN_PARTS_MAX * 2 * sizeof(double);
}

extern "C" int adamvs_group_stats_partial(const float* x0, const float* x1, int N, int npix, int D, int n, void* partials,
                                          size_t partials_bytes, void* stream) {
  const int ngroups = x1 ? 2 : 1;
  ADAMVS_CHECK_ARG(x0 && partials && N > 0 && npix > 0 && n > 0 && (n % 4) == 0 && (D % 4) == 0 && n <= D,
                   "group_stats_partial: bad arguments (N=%d npix=%d D=%d n=%d)", N, npix, D, n);
  ADAMVS_CHECK_ARG(partials_bytes >= adamvs_group_stats_workspace_bytes(N, ngroups), "group_stats_partial: buffer too small");
  hipLaunchKernelGGL(k_gn_partial, dim3(gn_parts(npix), ngroups, N), dim3(256), 0, (hipStream_t)stream, x0, x1,
                     (double*)partials, npix, D, n);
  ADAMVS_CHECK_LAUNCH("group_stats_partial");
  return 0;
}

extern "C" int adamvs_group_stats_finish(const void* partials, float* stats, int N, int ngroups, int npix, int n, float eps,
                                         void* stream) {
  ADAMVS_CHECK_ARG(partials && stats && N > 0 && (ngroups == 1 || ngroups == 2) && npix > 0 && n > 0,
                   "group_stats_finish: bad arguments");
  hipLaunchKernelGGL(k_gn_final, dim3(N), dim3(64), 0, (hipStream_t)stream, (const double*)partials, stats, ngroups,
                     gn_parts(npix), npix * n, eps);
  ADAMVS_CHECK_LAUNCH("group_stats_finish");
  return 0;
}

extern "C" int adamvs_gru2_gates_apply(const float* fr, const float* fu, int Wf, const void* partials, const float* gn,
                                       const float* h, float* rh, float* u, int N, int npix, int W, int HC, float eps,
                                       void* stream) {
  ADAMVS_CHECK_ARG(fr && fu && partials && gn && h && rh && u && N > 0 && npix > 0 && (HC % 4) == 0 && (W % 4) == 0 && HC <= W &&
                   (Wf % 4) == 0 && HC <= Wf, "gru2_gates_apply: bad arguments (Wf=%d W=%d HC=%d)", Wf, W, HC);
  return launch_gates_apply(fr, fu, Wf, (const double*)partials, gn_parts(npix), gn, h, rh, u, N, npix, W, HC, eps, (hipStream_t)stream);
}

extern "C" int adamvs_gru2_out_apply(const float* o, const void* partials, const float* gn, const float* u, float* h, float* out,
                                     int Wo, int N, int npix, int W, int HC, float eps, void* stream) {
  ADAMVS_CHECK_ARG(o && partials && gn && u && h && N > 0 && npix > 0 && (HC % 4) == 0 && (W % 4) == 0 && HC <= W &&
                   (!out || ((Wo % 4) == 0 && HC <= Wo)), "gru2_out_apply: bad arguments (W=%d HC=%d Wo=%d)", W, HC, Wo);
  return launch_out_apply(o, (const double*)partials, gn_parts(npix), gn, u, h, out, Wo, N, npix, W, HC, eps, (hipStream_t)stream);
}

extern "C" int adamvs_conv3x3_pair(const float* srcA, int CA, const float* srcB, int CB, const float* wpk, const float* bias,
                                   float* out, int cout, int B, int h, int w, void* stream) {
  ADAMVS_CHECK_ARG(srcA && srcB && wpk && bias && out && B > 0 && h > 0 && w > 0 && cout > 0 && (cout % 4) == 0,
                   "conv3x3_pair: bad arguments (B=%d h=%d w=%d cout=%d)", B, h, w, cout);
  return launch_conv_pair(srcA, CA, srcB, CB, wpk, bias, out, cout, B, h, w, (hipStream_t)stream);
}

// ---- one level's recurrence over the planes of a stage, launched from native code (a Python loop of 6-7 ctypes calls
// per plane and level costs more than the kernels it launches)
namespace {
struct RecurBuffers { float *state, *rh, *f, *fu, *f2, *fu2, *o, *u; double* part; };

// state, rh, o: W wide; f, fu and their second buffers (the folded path of _split keeps two planes of them): Wf wide; u: HC
size_t recur_floats(int B, int npix, int W, int Wf, int HC) {
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  return 3 * al((size_t)B * npix * W) + 4 * al((size_t)B * npix * Wf) + al((size_t)B * npix * HC);
}
size_t recur_partial_bytes(int B) { return 3 * adamvs_group_stats_workspace_bytes(B, 2); }     // f of two planes, o

int carve_recur(RecurBuffers& r, void* workspace, size_t bytes, int B, int npix, int W, int Wf, int HC, hipStream_t st) {
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  const size_t need = recur_floats(B, npix, W, Wf, HC) * sizeof(float) + recur_partial_bytes(B);
  if (!workspace || bytes < need) return set_error(-1, "red_recur: workspace too small (%zu < %zu bytes)", bytes, need);
  float* p = (float*)workspace;
  r.state = p; p += al((size_t)B * npix * W);
  r.rh = p;    p += al((size_t)B * npix * W);
  r.o = p;     p += al((size_t)B * npix * W);
  r.f = p;     p += al((size_t)B * npix * Wf);
  r.fu = p;    p += al((size_t)B * npix * Wf);
  r.f2 = p;    p += al((size_t)B * npix * Wf);
  r.fu2 = p;   p += al((size_t)B * npix * Wf);
  r.u = p;     p += al((size_t)B * npix * HC);
  r.part = (double*)p;
  // the state starts at zero; rh keeps zeros in its padding channels (only HC channels are ever written)
  hipError_t e = zero_floats(r.state, 2 * al((size_t)B * npix * W), st);
  if (e != hipSuccess) return set_error((int)e, "red_recur: zero state: %s", hipGetErrorString(e));
  return 0;
}
}  // namespace

extern "C" size_t adamvs_red_recur_workspace_bytes(int B, int h, int w, int W, int Wf, int HC) {
  if (B <= 0 || h <= 0 || w <= 0 || W <= 0 || Wf <= 0 || HC <= 0) return 0;
  return recur_floats(B, h * w, W, Wf, HC) * sizeof(float) + recur_partial_bytes(B);
}

extern "C" int adamvs_red_recur_pair(const float* x, int Cx, const float* wg, const float* bg, const float* wc, const float* bc,
                                     const float* gn, float* R, int RW, int B, int D, int h, int w, int HC, float eps,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  ADAMVS_CHECK_ARG(x && wg && bg && wc && bc && gn && R && B > 0 && D > 0 && h > 0 && w > 0 && HC > 0 && (HC % 4) == 0 && RW >= HC,
                   "red_recur_pair: bad arguments (B=%d D=%d h=%d w=%d Cx=%d HC=%d RW=%d)", B, D, h, w, Cx, HC, RW);
  hipStream_t st = (hipStream_t)stream;
  const int npix = h * w;
  RecurBuffers r;
  if (int rc = carve_recur(r, workspace, workspace_bytes, B, npix, HC, 2 * HC, HC, st)) return rc;
  const size_t pbytes = adamvs_group_stats_workspace_bytes(B, 2);
  if (gru_fold_enabled(B) && conv_pair_epilogue_partials(B, h, w)) {
    // two dependent launches per plane, as adamvs_red_recur_split below: gate_conv of plane d forms h(d-1) in its window fill
    // and stores it, output_conv forms r * h(d-1); compact maps (f: 2 HC wide, the reset half first)
    float* S[2] = {r.state, r.rh};                           // both zeroed by carve_recur
    float* F[2] = {r.f, r.f2};
    double* PF[2] = {r.part, r.part + pbytes / sizeof(double)};
    double* PO = r.part + 2 * (pbytes / sizeof(double));
    int pf_parts[2] = {0, 0}, po_parts = 0;
    for (int d = 0; d < D; ++d) {
      const float* xd = x + (size_t)d * B * npix * Cx;
      int rc;
      GruPro out{GRU_PRO_OUT, F[(d + 1) & 1] + HC, r.o, PF[(d + 1) & 1], PO, pf_parts[(d + 1) & 1], 1, po_parts, gn + 2 * HC, gn + 4 * HC,
                 S[(d + 1) & 1], d > 0 ? R + (size_t)(d - 1) * B * npix * RW : nullptr, RW, HC, npix * HC, eps};
      if ((rc = launch_conv_pair(xd, Cx, S[d & 1], HC, wg, bg, F[d & 1], 2 * HC, B, h, w, st, PF[d & 1], HC, 2, &pf_parts[d & 1],
                                 d > 0 ? &out : nullptr)))
        return rc;
      GruPro gates{GRU_PRO_GATES, F[d & 1], nullptr, PF[d & 1], nullptr, pf_parts[d & 1], 0, 0, gn, nullptr,
                   nullptr, nullptr, 0, HC, npix * HC, eps};
      if ((rc = launch_conv_pair(xd, Cx, S[(d + 1) & 1], HC, wc, bc, r.o, HC, B, h, w, st, PO, HC, 1, &po_parts, &gates))) return rc;
    }
    hipLaunchKernelGGL(k_gru2_last_apply, dim3(cdiv(npix * (HC / 4), 256), B), dim3(256), 0, st, r.o, F[(D - 1) & 1] + HC, PO, po_parts,
                       PF[(D - 1) & 1], pf_parts[(D - 1) & 1], gn, S[D & 1], R + (size_t)(D - 1) * B * npix * RW, npix, 2 * HC, HC, HC, RW,
                       eps);
    ADAMVS_CHECK_LAUNCH("gru2_last_apply");
    return 0;
  }
  for (int d = 0; d < D; ++d) {
    const float* xd = x + (size_t)d * B * npix * Cx;
    int rc;
    // the GroupNorm statistics of a convolution's output come out of its epilogue as partial sums (one dependent launch fewer
    // per normalisation); maps with more tiles than the partial buffer holds are reduced by k_gn_partial as before
    int parts = 0;
    if ((rc = launch_conv_pair(xd, Cx, r.state, HC, wg, bg, r.f, 2 * HC, B, h, w, st, r.part, HC, 2, &parts))) return rc;
    if (!parts) {
      if ((rc = adamvs_group_stats_partial(r.f, r.f + HC, B, npix, 2 * HC, HC, r.part, pbytes, stream))) return rc;
      parts = gn_parts(npix);
    }
    if ((rc = launch_gates_apply(r.f, r.f + HC, 2 * HC, r.part, parts, gn, r.state, r.rh, r.u, B, npix, HC, HC, eps, st))) return rc;
    if ((rc = launch_conv_pair(xd, Cx, r.rh, HC, wc, bc, r.o, HC, B, h, w, st, r.part, HC, 1, &parts))) return rc;
    if (!parts) {
      if ((rc = adamvs_group_stats_partial(r.o, nullptr, B, npix, HC, HC, r.part, pbytes, stream))) return rc;
      parts = gn_parts(npix);
    }
    if ((rc = launch_out_apply(r.o, r.part, parts, gn + 4 * HC, r.u, r.state, R + (size_t)d * B * npix * RW, RW, B, npix, HC, HC, eps, st)))
      return rc;
  }
  return 0;
}

extern "C" int adamvs_red_recur_split(const float* gxr, const float* gxu, const float* cx, const float* w_ghr, const float* w_ghu,
                                      const float* w_ch, const float* gn, float* R, int RW, int B, int D, int h, int w, int W,
                                      int HC, float eps, void* workspace, size_t workspace_bytes, void* stream) {
  ADAMVS_CHECK_ARG(gxr && gxu && cx && w_ghr && w_ghu && w_ch && gn && R && B > 0 && D > 0 && h > 0 && w > 0 && HC > 0 &&
                   (HC % 4) == 0 && W >= HC && RW >= HC, "red_recur_split: bad arguments (B=%d D=%d h=%d w=%d W=%d HC=%d RW=%d)",
                   B, D, h, w, W, HC, RW);
  hipStream_t st = (hipStream_t)stream;
  const int npix = h * w;
  RecurBuffers r;
  if (int rc = carve_recur(r, workspace, workspace_bytes, B, npix, W, W, HC, st)) return rc;
  const size_t pbytes = adamvs_group_stats_workspace_bytes(B, 2);
  const size_t wsz = (size_t)9 * W * W, plane = (size_t)B * npix * W;
  if (can_fold_gru_applies(B, W, h, w)) {
    // Two dependent launches per plane: the gate convolutions of plane d form h(d-1) = u h(d-2) + (1 - u) tanh(GN(o)) in their
    // window fill (GRU_PRO_OUT) and store it; the candidate convolution forms r * h(d-1) in its own (GRU_PRO_GATES).  The maps a
    // launch reads while it writes their successors are double-buffered: states S[k & 1] = h(k), f(d) in F[d & 1], and the
    // partial sums with them (rh's space of the unfolded path is the second state).
    float* S[2] = {r.state, r.rh};                           // both zeroed by carve_recur: h(-1) = h(-2) = 0
    float* FR[2] = {r.f, r.f2};
    float* FU[2] = {r.fu, r.fu2};
    double* PF[2] = {r.part, r.part + pbytes / sizeof(double)};
    double* PO = r.part + 2 * (pbytes / sizeof(double));
    int pf_parts[2] = {0, 0}, po_parts = 0;
    for (int d = 0; d < D; ++d) {
      int rc;
      GruPro out{GRU_PRO_OUT, FU[(d + 1) & 1], r.o, PF[(d + 1) & 1], PO, pf_parts[(d + 1) & 1], 1, po_parts, gn + 2 * HC, gn + 4 * HC,
                 S[(d + 1) & 1], d > 0 ? R + (size_t)(d - 1) * B * npix * RW : nullptr, RW, HC, npix * HC, eps};
      if ((rc = launch_conv_dd_gates_gn(S[d & 1], w_ghr, w_ghr + wsz, gxr + d * plane, FR[d & 1], w_ghu, w_ghu + wsz, gxu + d * plane,
                                        FU[d & 1], B, W, h, w, st, PF[d & 1], HC, &pf_parts[d & 1], d > 0 ? &out : nullptr)))
        return rc;
      GruPro gates{GRU_PRO_GATES, FR[d & 1], nullptr, PF[d & 1], nullptr, pf_parts[d & 1], 0, 0, gn, nullptr,
                   nullptr, nullptr, 0, HC, npix * HC, eps};
      if ((rc = launch_conv_dd_gn(S[(d + 1) & 1], w_ch, w_ch + wsz, cx + d * plane, r.o, B, W, h, w, st, PO, HC, 0, 1, &po_parts, &gates)))
        return rc;
    }
    hipLaunchKernelGGL(k_gru2_last_apply, dim3(cdiv(npix * (HC / 4), 256), B), dim3(256), 0, st, r.o, FU[(D - 1) & 1], PO, po_parts,
                       PF[(D - 1) & 1], pf_parts[(D - 1) & 1], gn, S[D & 1], R + (size_t)(D - 1) * B * npix * RW, npix, W, W, HC, RW, eps);
    ADAMVS_CHECK_LAUNCH("gru2_last_apply");
    return 0;
  }
  for (int d = 0; d < D; ++d) {
    int rc;
    // Wh.h + (Wx.x + b): the x halves of all planes were computed before the recurrence and enter as `skip`
    int pr = 0, po = 0;
    // the reset- and the update-gate convolution read the same state: one launch (8 waves) on the small maps
    if ((rc = launch_conv_dd_gates_gn(r.state, w_ghr, w_ghr + wsz, gxr + d * plane, r.f, w_ghu, w_ghu + wsz, gxu + d * plane, r.fu, B, W,
                                      h, w, st, r.part, HC, &pr)))
      return rc;
    if (!pr) {
      if ((rc = adamvs_group_stats_partial(r.f, r.fu, B, npix, W, HC, r.part, pbytes, stream))) return rc;
      pr = gn_parts(npix);
    }
    if ((rc = launch_gates_apply(r.f, r.fu, W, r.part, pr, gn, r.state, r.rh, r.u, B, npix, W, HC, eps, st))) return rc;
    if ((rc = launch_conv_dd_gn(r.rh, w_ch, w_ch + wsz, cx + d * plane, r.o, B, W, h, w, st, r.part, HC, 0, 1, &po))) return rc;
    if (!po) {
      if ((rc = adamvs_group_stats_partial(r.o, nullptr, B, npix, W, HC, r.part, pbytes, stream))) return rc;
      po = gn_parts(npix);
    }
    if ((rc = launch_out_apply(r.o, r.part, po, gn + 4 * HC, r.u, r.state, R + (size_t)d * B * npix * RW, RW, B, npix, W, HC, eps, st)))
      return rc;
  }
  return 0;
}

extern "C" int adamvs_soft_argmin(const float* vol, const float* planes, float* depth, float* confidence, int B, int D, int h,
                                  int w, void* stream) {
  ADAMVS_CHECK_ARG(vol && planes && depth && confidence && B > 0 && D > 0 && h > 0 && w > 0, "soft_argmin: bad arguments");
  return launch_soft_argmin(vol, planes, depth, confidence, B, D, h, w, 0, (hipStream_t)stream);
}
