// C-ABI glue: error plumbing, the one-step op and the whole-stage driver
// (InferDepthNet0.forward, reference models/adamvs.py:433-533).
#include <stdarg.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "../../include/adamvs_hip.h"
#include "common.h"
#include "kernels.h"

namespace adamvs {

thread_local char g_last_error[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}

static_assert(sizeof(FuseWeights) == sizeof(adamvs_fuse_weights), "adamvs_fuse_weights layout");

// ---- run-time options (options.h; documented in include/adamvs_hip.h "OPTIONS")
struct OptionEntry { const char* name; int def; };
static const OptionEntry g_option_table[OPT_COUNT] = {
#define ADAMVS_OPTION_ROW(e, n, d) {n, d},
    ADAMVS_OPTION_LIST(ADAMVS_OPTION_ROW)
#undef ADAMVS_OPTION_ROW
};
static std::atomic<int> g_option_value[OPT_COUNT];
static std::once_flag g_option_once;
// defaults, then ADAMVS_<NAME> from the environment: the library's one read of the process environment besides the two cost
// tables of recurrence.hip (tuning), made once
static void option_init() {
  std::call_once(g_option_once, [] {
    for (int i = 0; i < OPT_COUNT; ++i) {
      int v = g_option_table[i].def;
      char key[64] = "ADAMVS_";
      size_t n = strlen(key);
      for (const char* c = g_option_table[i].name; *c && n + 1 < sizeof(key); ++c) key[n++] = (*c >= 'a' && *c <= 'z') ? (char)(*c - 32) : *c;
      key[n] = 0;
      if (const char* e = getenv(key))
        if (*e) v = atoi(e);
      g_option_value[i].store(v, std::memory_order_relaxed);
    }
  });
}
int opt(Option o) {
  option_init();
  return g_option_value[o].load(std::memory_order_relaxed);
}
static int option_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (!strcmp(name, g_option_table[i].name)) return i;
  return -1;
}

static size_t align_up(size_t n) { return (n + 63) & ~(size_t)63; }   // in floats: 256-byte slots

// Zero fills and device-to-device copies are KERNELS here, not the runtime's asynchronous memset / memcpy calls.  Inside a captured
// hipGraph those become memset / memcpy nodes, and on this stack (ROCm 7.x, gfx950) a memset node between kernel nodes was observed to
// run out of order with them: a replayed stage started its recurrence from the previous replay's states as soon as the kernels around
// the node were short (profiles/r06_graph_memset_node.txt: the same graph is bit-exact over 28 replays with the fill as a kernel and
// wrong from the second replay on with the memset call, whatever the kernel-argument placement).
__global__ __launch_bounds__(256) void k_fill_zero(float* __restrict__ p, size_t n) {
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (((size_t)p & 15) == 0)
    for (size_t i = i0; i < n4; i += stride) ((f32x4*)p)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  else
    for (size_t i = i0; i < n4 * 4; i += stride) p[i] = 0.f;
  if (i0 < n - n4 * 4) p[n4 * 4 + i0] = 0.f;
}
__global__ __launch_bounds__(256) void k_copy_floats(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
static unsigned fill_grid(size_t n) { return (unsigned)(n / 1024 + 1 < 2048 ? n / 1024 + 1 : 2048); }
hipError_t zero_floats(float* p, size_t n, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_fill_zero, dim3(fill_grid(n)), dim3(256), 0, st, p, n);
  return hipGetLastError();
}
hipError_t copy_floats(const float* src, float* dst, size_t n, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_copy_floats, dim3(fill_grid(n)), dim3(256), 0, st, src, dst, n);
  return hipGetLastError();
}

// Workspace of one stage.  The hypothesis axis is processed in chunks of DC = sweep_chunk_planes(D) planes (aggregation
// -> conv1 -> recurrence -> soft-argmin accumulation per chunk, reference models/adamvs.py:495-531 runs the same
// chain per plane), so nothing below grows with D except the stage-1 similarity / score volumes, whose hypothesis axis
// is CostRegNet2D's channel axis (models/adamvs.py:464-486).
struct StageCarve {
  size_t c1, h1[4], rh1, u1, c2[2], h2[2], rh2, u2, vol[2], acc, agg, sim, score, creg, total;   // offsets in floats
  int dc;
};

// the width CostRegNet2D runs at for the stage's D hypotheses (kernels.h::costreg_width; 0: none)
static int stage_reg_width(const adamvs_stage_desc& s) {
  return s.precision == PRECISION_BF16X3 ? costreg_width_bf16x3(s.D) : costreg_width(s.D);
}

static StageCarve carve(const adamvs_stage_desc& s) {
  StageCarve c;
  size_t hw = (size_t)s.h * s.w, hw4 = (size_t)(s.h / 2) * (s.w / 2);
  size_t HW = s.in_up ? 4 * hw : hw;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += align_up(n); return r; };
  c.dc = sweep_chunk_planes(s.D);
  c.c1 = take((size_t)c.dc * s.B * hw * 8);
  for (int i = 0; i < 4; ++i) c.h1[i] = take((size_t)s.B * hw * 8);
  c.rh1 = take((size_t)s.B * hw * 8);
  c.u1 = take((size_t)s.B * hw * 8);
  for (int i = 0; i < 2; ++i) c.c2[i] = take((size_t)s.B * hw4 * 16);
  for (int i = 0; i < 2; ++i) c.h2[i] = take((size_t)s.B * hw4 * 16);
  c.rh2 = take((size_t)s.B * hw4 * 16);
  c.u2 = take((size_t)s.B * hw4 * 16);
  for (int i = 0; i < 2; ++i) c.vol[i] = take((size_t)s.B * c.dc * HW);      // the decoder runs two or three hypotheses behind level 1
  c.acc = take(3 * (size_t)s.B * HW);
  c.agg = take(sweep_workspace_floats(s.B, s.C, s.D, s.h, s.w));
  c.sim = c.score = c.creg = o;
  if (s.first_stage) {
    // the similarity / score volumes and CostRegNet2D's activations at the width the network runs at (>= D)
    size_t F = (size_t)s.S * s.B * hw * stage_reg_width(s);
    c.sim = take(F);
    c.score = take(F);
    c.creg = take(3 * F);
  }
  c.total = o;
  return c;
}

static int check_desc(const adamvs_stage_desc* d) {
  ADAMVS_CHECK_ARG(d, "stage: null descriptor");
  ADAMVS_CHECK_ARG(d->B > 0 && d->S > 0 && d->D > 1 && d->h > 1 && d->w > 1, "stage: bad shape (B=%d S=%d D=%d h=%d w=%d)",
                   d->B, d->S, d->D, d->h, d->w);
  ADAMVS_CHECK_ARG(d->C == 8 || d->C == 16 || d->C == 32, "stage: C=%d unsupported (8, 16 or 32)", d->C);
  ADAMVS_CHECK_ARG((d->h % 2) == 0 && (d->w % 2) == 0, "stage: h=%d w=%d must be even", d->h, d->w);
  ADAMVS_CHECK_ARG(d->B <= 65535, "stage: B=%d tiles per call (at most 65535)", d->B);
  ADAMVS_CHECK_ARG(d->precision == PRECISION_FP32 || d->precision == PRECISION_BF16X3, "stage: precision=%d (0 fp32, 1 bf16x3)", d->precision);
  ADAMVS_CHECK_ARG(d->eps_in_numerator == 0 || d->eps_in_numerator == 1, "stage: eps_in_numerator=%d (0 or 1)", d->eps_in_numerator);
  ADAMVS_CHECK_ARG(d->plane_mode >= PLANES_EXPLICIT && d->plane_mode <= PLANES_WINDOW, "stage: plane_mode=%d (0 explicit, 1 uniform, 2 window)", d->plane_mode);
  ADAMVS_CHECK_ARG(d->precision_fuse == PRECISION_FP32 || d->precision_fuse == PRECISION_BF16X3,
                   "stage: precision_fuse=%d (0 fp32, 1 bf16x3)", d->precision_fuse);
  if (d->first_stage) {
    ADAMVS_CHECK_ARG(stage_reg_width(*d) > 0, "stage: D=%d hypotheses: CostRegNet2D runs at widths up to 512", d->D);
    ADAMVS_CHECK_ARG((d->h % 8) == 0 && (d->w % 8) == 0, "stage: first stage needs h=%d w=%d multiples of 8", d->h, d->w);
  } else {
    ADAMVS_CHECK_ARG(d->prev_h > 0 && d->prev_w > 0, "stage: prev_h/prev_w missing");
  }
  return 0;
}

}  // namespace adamvs

using namespace adamvs;

extern "C" int adamvs_version(void) { return ADAMVS_ABI_VERSION; }

extern "C" int adamvs_option_count(void) { return OPT_COUNT; }
extern "C" const char* adamvs_option_name(int index) { return index >= 0 && index < OPT_COUNT ? g_option_table[index].name : nullptr; }
extern "C" int adamvs_option_default(const char* name, int* value) {
  const int i = option_index(name);
  ADAMVS_CHECK_ARG(i >= 0 && value, "option_default: unknown option '%s' (include/adamvs_hip.h, OPTIONS)", name ? name : "(null)");
  *value = g_option_table[i].def;
  return 0;
}
extern "C" int adamvs_get_option(const char* name, int* value) {
  const int i = option_index(name);
  ADAMVS_CHECK_ARG(i >= 0 && value, "get_option: unknown option '%s' (include/adamvs_hip.h, OPTIONS)", name ? name : "(null)");
  *value = opt((Option)i);
  return 0;
}
extern "C" int adamvs_set_option(const char* name, int value) {
  const int i = option_index(name);
  ADAMVS_CHECK_ARG(i >= 0, "set_option: unknown option '%s' (include/adamvs_hip.h, OPTIONS)", name ? name : "(null)");
  option_init();
  g_option_value[i].store(value, std::memory_order_relaxed);
  return 0;
}
extern "C" const char* adamvs_last_error_string(void) { return g_last_error; }

extern "C" size_t adamvs_slice_reg_step_scratch_bytes(int B, int h, int w) {
  size_t hw = (size_t)h * w, hw4 = (size_t)(h / 2) * (w / 2);
  return (3 * align_up((size_t)B * hw * 8) + 3 * align_up((size_t)B * hw4 * 16)) * sizeof(float);
}

extern "C" int adamvs_slice_reg_step(const float* cost, float* state1, float* state2, const adamvs_fuse_weights* weights,
                                     float* reg_cost, int B, int C, int h, int w, int in_up, int precision, void* scratch,
                                     size_t scratch_bytes, void* stream) {
  ADAMVS_CHECK_ARG(precision == PRECISION_FP32 || precision == PRECISION_BF16X3, "slice_reg_step: precision=%d (0 fp32, 1 bf16x3)", precision);
  ADAMVS_CHECK_ARG(cost && state1 && state2 && weights && reg_cost && scratch, "slice_reg_step: null pointer");
  ADAMVS_CHECK_ARG(B > 0 && h > 1 && w > 1 && (h % 2) == 0 && (w % 2) == 0, "slice_reg_step: bad shape (B=%d h=%d w=%d, even sizes)", B, h, w);
  ADAMVS_CHECK_ARG(C == 8 || C == 16 || C == 32, "slice_reg_step: C=%d unsupported (8, 16 or 32)", C);
  ADAMVS_CHECK_ARG(scratch_bytes >= adamvs_slice_reg_step_scratch_bytes(B, h, w), "slice_reg_step: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  size_t n1 = align_up((size_t)B * h * w * 8), n2 = align_up((size_t)B * (h / 2) * (w / 2) * 16);
  float* s = (float*)scratch;
  float* c1 = s;
  StepBuffers sb{state1, s + n1, s + 2 * n1, s + 3 * n1, state2, s + 3 * n1 + n2, s + 3 * n1 + 2 * n2};
  FuseWeights fw;
  memcpy(&fw, weights, sizeof(fw));
  int rc = launch_conv1(cost, fw.conv1, c1, B, C, h, w, precision, st);
  if (rc) return rc;
  float* h1_now = state1;
  float* h2_now = state2;
  if ((rc = launch_slice_step(c1, fw, sb, reg_cost, B, h, w, 1, 0, in_up, precision, st, &h1_now, &h2_now))) return rc;
  if (h1_now != state1) {              // the split-bf16 GRU kernels write the new state to the other buffer
    hipError_t e = copy_floats(h1_now, state1, (size_t)B * h * w * 8, st);
    if (e != hipSuccess) return set_error((int)e, "slice_reg_step: state copy: %s", hipGetErrorString(e));
  }
  if (h2_now != state2) {
    hipError_t e = copy_floats(h2_now, state2, (size_t)B * (h / 2) * (w / 2) * 16, st);
    if (e != hipSuccess) return set_error((int)e, "slice_reg_step: state copy: %s", hipGetErrorString(e));
  }
  return 0;
}

extern "C" int adamvs_recurrence_schedule(int precision_fuse, long long pixels) { return recurrence_mode(precision_fuse, (long)pixels); }
extern "C" int adamvs_gru_wino_mask(void) { return gru_wino_mask(); }

extern "C" size_t adamvs_depth_stage_workspace_bytes(const adamvs_stage_desc* desc) {
  if (check_desc(desc)) return 0;
  return carve(*desc).total * sizeof(float);
}

static int stage_forward(const adamvs_stage_desc* desc, const float* feat, const float* rt, const float* planes,
                         const float* prev_conf, const float* w_reg, size_t w_reg_floats, const adamvs_fuse_weights* w_fuse,
                         float* view_weight, float* pair_depth, float* depth, float* confidence, int phases, void* workspace,
                         size_t workspace_bytes, void* stream, bool timing_only) {
  int rc = check_desc(desc);
  if (rc) return rc;
  ADAMVS_CHECK_ARG(phases >= 0 && phases <= ADAMVS_PHASE_ALL, "stage: phases=%d (a subset of ADAMVS_PHASE_ALL = 15)", phases);
  const adamvs_stage_desc& s = *desc;
  ADAMVS_CHECK_ARG(feat && rt && planes && w_fuse && view_weight && depth && confidence && workspace, "stage: null pointer");
  ADAMVS_CHECK_ARG(!s.first_stage || (w_reg && pair_depth), "stage: first stage needs w_reg and pair_depth");
  ADAMVS_CHECK_ARG(!s.first_stage || w_reg_floats == cost_reg_weight_floats(stage_reg_width(s), s.precision),
                   "stage: w_reg holds %zu floats; D=%d hypotheses run at width %d, whose layout has %zu (include/adamvs_hip.h)",
                   w_reg_floats, s.D, stage_reg_width(s), cost_reg_weight_floats(stage_reg_width(s), s.precision));
  ADAMVS_CHECK_ARG(s.first_stage || prev_conf, "stage: later stages need prev_conf");
  StageCarve c = carve(s);
  ADAMVS_CHECK_ARG(workspace_bytes >= c.total * sizeof(float), "stage: workspace too small (%zu < %zu bytes)", workspace_bytes,
                   c.total * sizeof(float));
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  FuseWeights fw;
  memcpy(&fw, w_fuse, sizeof(fw));
  const PlaneSrc ps{planes, s.plane_mode, s.half_span, s.plane_mode == PLANES_WINDOW ? s.half_span_dev : nullptr};

  // -- view weights: scored by CostRegNet2D (stage 1) or resampled from the previous stage
  if (!(phases & ADAMVS_PHASE_VIEW_WEIGHTS)) {
  } else if (s.first_stage) {
    // Dr >= D: the width CostRegNet2D runs at (w_reg is packed for it: adamvs_cost_reg_width); the pad channels of the
    // similarity volume are zeros, their scores -1e30, so the softmax sees D hypotheses
    const int Dr = stage_reg_width(s);
    if ((rc = launch_pair_similarity(feat, rt, ps, ws + c.sim, s.B, s.S, s.C, s.D, s.h, s.w, st, Dr))) return rc;
    if (cost_reg_softmax_fusable(Dr, s.precision, ps)) {       // softmax / max / regression in the epilogue of the last layer
      if ((rc = launch_cost_reg_net_2d(ws + c.sim, w_reg, ws + c.creg, ws + c.score, s.S * s.B, Dr, s.h, s.w, s.precision, st,
                                       view_weight, pair_depth, &ps, s.B, s.D)))
        return rc;
    } else {
      if ((rc = launch_cost_reg_net_2d(ws + c.sim, w_reg, ws + c.creg, ws + c.score, s.S * s.B, Dr, s.h, s.w, s.precision, st))) return rc;
      if ((rc = launch_softmax_regress(ws + c.score, ps, view_weight, pair_depth, s.S, s.B, Dr, s.h, s.w, st, s.D))) return rc;
    }
  } else {
    if ((rc = adamvs_resize_bilinear(prev_conf, view_weight, s.S * s.B, s.prev_h, s.prev_w, s.h, s.w, stream))) return rc;
  }

  // -- per chunk of hypotheses: weighted aggregation + conv1 (state-independent), the recurrence, soft-argmin accumulation.
  // With the three bits set they are interleaved chunk by chunk and the recurrence runs as a pipeline across chunk
  // boundaries.  The workspace holds ONE chunk of conv1 outputs and two of cost slices, so with more than one chunk a
  // proper subset of the three cannot hand its results to a later call: refused by adamvs_depth_stage_forward.  The measurement
  // entry point adamvs_bench_stage_phase (bench.py's phase-by-phase timing) runs the selected phase alone over all chunks on
  // whatever the buffers hold: its duration is the phase's, it promises no maps.
  const bool do_agg = phases & ADAMVS_PHASE_AGGREGATE, do_rec = phases & ADAMVS_PHASE_RECURRENCE, do_arg = phases & ADAMVS_PHASE_SOFT_ARGMIN;
  if (!do_agg && !do_rec && !do_arg) return 0;
  const size_t hw = (size_t)s.h * s.w, hw4 = (size_t)(s.h / 2) * (s.w / 2);
  const int dc = c.dc, nchunks = (s.D + dc - 1) / dc;
  ADAMVS_CHECK_ARG(nchunks == 1 || (do_agg && do_rec && do_arg) || timing_only,
                   "stage: phases=%d selects a proper subset of AGGREGATE|RECURRENCE|SOFT_ARGMIN, but D=%d runs in %d chunks of %d "
                   "hypotheses and the workspace keeps one: pass all three in one call (adamvs_bench_stage_phase times one "
                   "phase alone, no maps)", phases, s.D, nchunks, dc);
  const size_t c1_stride = (size_t)s.B * hw * 8;
  GruStateRing rb{{ws + c.h1[0], ws + c.h1[1], ws + c.h1[2], ws + c.h1[3]}, ws + c.rh1, ws + c.u1, {ws + c.c2[0], ws + c.c2[1]},
                  {ws + c.h2[0], ws + c.h2[1]}, ws + c.rh2, ws + c.u2};
  const int mode = recurrence_mode(s.precision_fuse, (long)s.B * s.h * s.w);
  const bool pipelined = mode != 0;
  const int lag = recurrence_lags(mode, s.precision_fuse).dec;       // the decoder runs `lag` hypotheses behind level 1
  if (do_rec) {      // zero initial states (adamvs.py:448-449): h1[-1] = ring slot 3, h2[-1] = ring slot 1 (slot 0 when sequential)
    hipError_t e = zero_floats(pipelined ? rb.h1[3] : rb.h1[0], (size_t)s.B * hw * 8, st);
    if (e == hipSuccess) e = zero_floats(pipelined ? rb.h2[1] : rb.h2[0], (size_t)s.B * hw4 * 16, st);
    if (e != hipSuccess) return set_error((int)e, "stage: zero initial states: %s", hipGetErrorString(e));
  }
  auto vol_of = [&](int d) { return ws + c.vol[(d / dc) & 1]; };
  auto argmin_chunk = [&](int k) {
    const int d0 = k * dc, nd = (d0 + dc < s.D ? d0 + dc : s.D) - d0;
    return launch_soft_argmin_chunk(ws + c.vol[k & 1], dc, ps, s.D, d0, nd, ws + c.acc, k == 0, k == nchunks - 1, depth, confidence,
                                    s.B, s.h, s.w, s.in_up, st);
  };
  StepBuffers sb{rb.h1[0], rb.rh1, rb.u1, rb.c2[0], rb.h2[0], rb.rh2, rb.u2};     // sequential mode: states updated in place
  for (int k = 0; k < nchunks; ++k) {
    const int d0 = k * dc, d1 = d0 + dc < s.D ? d0 + dc : s.D;
    if (do_agg && (rc = launch_sweep_conv1_chunk(feat, rt, ps, view_weight, fw.conv1, ws + c.c1, ws + c.agg, s.B, s.S, s.C, s.D, d0,
                                                 d1, s.h, s.w, s.precision_fuse, s.eps_in_numerator, st)))
      return rc;
    if (do_rec) {
      for (int t = d0; t < d1; ++t) {
        const float* c1_t = ws + c.c1 + (size_t)(t - d0) * c1_stride;
        if (pipelined) {
          // level 1 of hypothesis t, level 2 of t-1 (t-2), decoder of t-lag (which may belong to the previous chunk)
          const int sd = t - lag;
          if ((rc = launch_recur_pipeline_step(rb, fw, s.B, s.h, s.w, s.D, t, c1_t, sd >= 0 ? vol_of(sd) : nullptr, dc, sd >= 0 ? sd % dc : 0,
                                               s.in_up, s.precision_fuse, mode, st)))
            return rc;
          if (do_arg && sd >= 0 && sd % dc == dc - 1 && (rc = argmin_chunk(sd / dc))) return rc;   // a chunk of the volume is complete
        } else if ((rc = launch_slice_step(c1_t, fw, sb, vol_of(t), s.B, s.h, s.w, dc, t % dc, s.in_up, s.precision_fuse, st))) {
          return rc;
        }
      }
      if (!pipelined && do_arg && (rc = argmin_chunk(k))) return rc;
    } else if (do_arg && (rc = argmin_chunk(k))) {
      return rc;
    }
  }
  if (do_rec && pipelined) {
    for (int t = s.D; t < s.D + lag; ++t) {     // drain: the levels and decoders still behind
      const int sd = t - lag;
      if ((rc = launch_recur_pipeline_step(rb, fw, s.B, s.h, s.w, s.D, t, nullptr, sd >= 0 ? vol_of(sd) : nullptr, dc, sd >= 0 ? sd % dc : 0,
                                           s.in_up, s.precision_fuse, mode, st)))
        return rc;
      if (do_arg && sd >= 0 && sd % dc == dc - 1 && sd / dc < nchunks - 1 && (rc = argmin_chunk(sd / dc))) return rc;
    }
    if (do_arg && (rc = argmin_chunk(nchunks - 1))) return rc;
  }
  return 0;
}

extern "C" int adamvs_depth_stage_forward(const adamvs_stage_desc* desc, const float* feat, const float* rt,
                                          const float* planes, const float* prev_conf, const float* w_reg, size_t w_reg_floats,
                                          const adamvs_fuse_weights* w_fuse, float* view_weight, float* pair_depth,
                                          float* depth, float* confidence, int phases, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  return stage_forward(desc, feat, rt, planes, prev_conf, w_reg, w_reg_floats, w_fuse, view_weight, pair_depth, depth, confidence,
                       phases, workspace, workspace_bytes, stream, false);
}

// MEASUREMENT ONLY (bench.py's phase table): the selected phases of a stage for their duration; depth / confidence are not valid
extern "C" int adamvs_bench_stage_phase(const adamvs_stage_desc* desc, const float* feat, const float* rt,
                                        const float* planes, const float* prev_conf, const float* w_reg, size_t w_reg_floats,
                                        const adamvs_fuse_weights* w_fuse, float* view_weight, float* pair_depth,
                                        float* depth, float* confidence, int phases, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  return stage_forward(desc, feat, rt, planes, prev_conf, w_reg, w_reg_floats, w_fuse, view_weight, pair_depth, depth, confidence,
                       phases, workspace, workspace_bytes, stream, true);
}
