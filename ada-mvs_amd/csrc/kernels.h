// Launchers shared between the kernel translation units and the C-ABI glue.
#pragma once
#include <stdlib.h>

#include "common.h"
#include "options.h"
#include "planes.h"

namespace adamvs {

enum { PRECISION_FP32 = 0, PRECISION_BF16X3 = 1 };

// zero fill / copy as kernels (api.hip: a captured hipGraph must not contain memset / memcpy nodes)
hipError_t zero_floats(float* p, size_t n, hipStream_t st);
hipError_t copy_floats(const float* src, float* dst, size_t n, hipStream_t st);

struct FuseWeights {          // mirrors adamvs_fuse_weights in include/adamvs_hip.h
  const float* conv1;         // [1][9][C/4][64]
  const float* gates1; const float* gates1_b;   // [1][9][4][64], [16]
  const float* cand1;  const float* cand1_b;    // [12][4][64] (two-row form, fp32), [16]
  const float* conv2;                           // [1][9][2][64]
  const float* gates2; const float* gates2_b;   // [2][9][8][64], [32]
  const float* cand2;  const float* cand2_b;    // [1][9][8][64], [16]
  const float* upconv1; const float* upconv1_b; // [1][9][4][64], [16]
  const float* final_w;                         // [73]
  // fp32 only: the GRU convolutions in the F(2x2, 3x3) form (slice_roles_wino.h), U = G g Gt as fragments [NT][4][4][cin/4][64]
  const float* gates1_w; const float* gates2_w; const float* cand2_w; const float* cand1_w;
};
int gru_wino_mask();          // which GRU convolutions run in the F(2x2, 3x3) form in one-role launches (slice_red.hip)

struct StepBuffers {          // all channel-last
  float* h1; float* rh1; float* u1;              // [B][hw][8]
  float* c2; float* h2; float* rh2; float* u2;   // [B][hw/4][16]
};

// State of the software-pipelined recurrence (recurrence.hip): h1 of step t in h1[t % 4]; h2 and conv2's output of step t
// in h2[t % 2], c2[t % 2]
struct GruStateRing {
  float* h1[4]; float* rh1; float* u1;              // [B][hw][8]
  float* c2[2]; float* h2[2]; float* rh2; float* u2;   // [B][hw/4][16]
};
struct RecurLags { int c2, dec; };       // hypotheses by which cand2 / the decoder run behind level 1
int recurrence_mode(int precision, long pixels);
RecurLags recurrence_lags(int schedule, int precision);
int launch_recur_pipeline_step(const GruStateRing& rb, const FuseWeights& fw, int B, int h, int w, int D, int t, const float* c1_t,
                               float* vol_dec, int D_vol, int d_dec, int in_up, int precision, int schedule, hipStream_t st);
int launch_soft_argmin_chunk(const float* vol, int vol_D, PlaneSrc planes, int D, int d0, int nd, float* acc, int first, int last,
                             float* depth, float* conf, int B, int h, int w, int in_up, hipStream_t st);
int sweep_chunk_planes(int D);
int launch_sweep_conv1_chunk(const float* feat, const float* rt, PlaneSrc planes, const float* vw, const float* w1pk,
                             float* c1_chunk, float* sim_ws, int B, int S, int C, int D, int d0, int d1, int h, int w, int precision,
                             int eps_in_numerator, hipStream_t st);

int launch_conv1(const float* cost, const float* w, float* c1, int B, int C, int h, int w_, int precision, hipStream_t st);
int launch_conv1_bf16x3(const float* cost, const float* w, float* c1, int N, int C, int h, int w_, hipStream_t st);
int launch_gru_convs_bf16x3(const float* c1, const FuseWeights& fw, const StepBuffers& sb, int B, int h, int w, int d,
                            float** h1_now, float** h2_now, hipStream_t st);
// step d of a stage; *h1_now / *h2_now (optional) receive the buffers holding the states afterwards (sb.h1 / sb.h2, or
// sb.rh1 / sb.rh2 after an even step of the split-bf16 path, whose fused GRU kernels alternate the two)
int launch_slice_step(const float* c1, const FuseWeights& fw, const StepBuffers& sb, float* vol, int B, int h, int w, int D,
                      int d, int in_up, int precision, hipStream_t st, float** h1_now = nullptr, float** h2_now = nullptr);
// (csrc/costreg2d.hip, k_conv_dd_resident)
enum { GRU_PRO_NONE = 0, GRU_PRO_GATES = 1, GRU_PRO_OUT = 2 };
struct GruPro {
  int mode;
  const float* f;            // [N][npix][D] (k_conv_dd_resident); k_conv_small: [N][npix][2 hc] + the half's channel offset
  const float* o;            // [N][npix][D] (GRU_PRO_OUT); k_conv_small: [N][npix][hc]
  const double* part_f;      // partial sums of f's group: [(n * 2 + group_f) * parts_f + k][2]
  const double* part_o;      // [(n * parts_o + k)][2]
  int parts_f, group_f, parts_o;
  const float* gn_f;         // [2][hc]: weight, bias of f's norm
  const float* gn_o;         // [2][hc]: output_norm
  float* state_out;          // [N][npix][D]
  float* R; int RW;          // [N][npix][RW] or null
  int hc, count;             // real channels; pixels x real channels of a sample (the norm's population)
  float eps;
};

constexpr int GN_PARTS_LIMIT = 2048;     // partial sums per (sample, group) the GroupNorm buffers of msred.hip hold
// A convolution writes its GroupNorm partials itself (one dependent launch fewer) while the stage is launch-bound: every block
// of the consumer finishes the reduction on its own, which is free for 64 partials and not for 1152 x 16 samples (measured at
// 16 tiles per step: 136 -> 147 ms with epilogue partials everywhere).
inline bool gn_epilogue_partials(long parts, int samples) { return parts > 0 && parts <= GN_PARTS_LIMIT && parts * samples <= 4096; }
int launch_conv_pair(const float* srcA, int CA, const float* srcB, int CB, const float* wpk, const float* bias, float* out,
                     int cout, int B, int h, int w, hipStream_t st, double* gn_part = nullptr, int gn_hc = 0, int gn_groups = 0,
                     int* gn_parts = nullptr, const GruPro* pro = nullptr);
bool conv_pair_epilogue_partials(int B, int h, int w);       // whether launch_conv_pair writes the GroupNorm partials itself
// one stride-1 layer with the GroupNorm partial sums of its output in the epilogue when the small-grid kernel takes it
// (*gn_parts > 0), plain otherwise (*gn_parts = 0): MS-REDNet's deep levels
// Folding pays while a stage is bound by its dependent launches: one or two tiles per step (measured at cfg3's shape: 51.3 ->
// 53.8 maps/s at one tile; 92.2 -> 88.7 at four and 115.4 -> 111.6 at sixteen, where the window halo's recomputed sigmoid /
// tanh and the per-workgroup reductions cost more than the launches they replace).  Option red_fold_applies = 0 / 1 forces.
inline bool gru_fold_enabled(int samples) {
  const int forced = opt(OPT_RED_FOLD_APPLIES);
  return forced >= 0 ? forced != 0 : samples <= 2;
}
bool can_fold_gru_applies(int N, int D, int h, int w);
int launch_conv_dd_gates_gn(const float* in, const float* wpk_r, const float* bias_r, const float* skip_r, float* out_r,
                            const float* wpk_u, const float* bias_u, const float* skip_u, float* out_u, int N, int D, int h, int w,
                            hipStream_t st, double* gn_part, int gn_n, int* gn_parts, const GruPro* pro = nullptr);
int launch_conv_dd_gn(const float* in, const float* wpk, const float* bias, const float* skip, float* out, int N, int D, int h, int w,
                      hipStream_t st, double* gn_part, int gn_n, int gn_group, int gn_ngroups, int* gn_parts, const GruPro* pro = nullptr);
int launch_soft_argmin(const float* vol, const float* planes, float* depth, float* conf, int B, int D, int h, int w,
                       int in_up, hipStream_t st);
int launch_sweep_conv1(const float* feat, const float* rt, PlaneSrc planes, const float* vw, const float* w1pk,
                       float* c1, float* sim_ws, int B, int S, int C, int D, int h, int w, int precision, int eps_in_numerator,
                       hipStream_t st);
int launch_sweep_variance(const float* feat, const float* rt, const float* planes, float* out_a, int Da, float* out_b, int Db,
                          int B, int S, int C, int D, int h, int w, hipStream_t st);
size_t sweep_workspace_floats(int B, int C, int D, int h, int w);
int launch_cost_reg_net_2d(const float* x, const float* wpk, float* ws, float* score, int N, int D, int h, int w,
                           int precision, hipStream_t st, float* sm_vw = nullptr, float* sm_pd = nullptr,
                           const PlaneSrc* sm_planes = nullptr, int sm_B = 1, int n_planes = 0);
size_t cost_reg_weight_floats(int D, int precision);        // floats of the packed weight blob at width D (0: unsupported)
int costreg_width(int D);            // the width CostRegNet2D runs at for D hypotheses (next supported; 0: none)
int costreg_width_bf16x3(int D);
bool cost_reg_softmax_fusable(int D, int precision, const PlaneSrc& planes);
int launch_conv_dd_bf16x3(const float* in, const float* wpk, const float* bias, const float* skip, float* out, int N, int D,
                          int hi, int wi, int ho, int wo, int mode, int relu, hipStream_t st, float* sm_vw = nullptr,
                          float* sm_pd = nullptr, const PlaneSrc* sm_planes = nullptr, int sm_B = 1, int n_planes = 0);
bool costreg_bf16x3_depth_supported(int D);
bool wino_depth_supported(int D);
bool cost_reg_winograd(int D, int precision);
bool wino_softmax_fused();
size_t wino_softmax_part_floats(int N, int D, int h, int w);
int launch_conv_wino_softmax(const float* in, const float* wpk, const float* bias, float* part, const PlaneSrc& planes, float* vw, float* pd,
                             int N, int B, int D, int n_planes, int h, int w, hipStream_t st);
int launch_conv_wino(const float* in, const float* wpk, const float* bias, const float* skip, float* out, int N, int D, int h, int w,
                     int relu, hipStream_t st);

// Dw >= D: channels per pixel of sim (the width CostRegNet2D runs at); channels [D, Dw) are written as zeros
int launch_pair_similarity(const float* feat, const float* rt, PlaneSrc planes, float* sim, int B, int S, int C, int D, int h,
                           int w, hipStream_t st, int Dw = 0);
// n_planes <= D hypothesis planes for D score channels (0: D)
int launch_softmax_regress(const float* score, PlaneSrc planes, float* vw, float* pd, int S, int B, int D, int h, int w,
                           hipStream_t st, int n_planes = 0);
bool costreg_depth_supported(int D);

}  // namespace adamvs
