/* libadamvs_hip.so -- C ABI of the MI355X (gfx950) Ada-MVS depth-inference hot path.
 *
 * The reference (gpcv-liujin/Ada-MVS) is pure PyTorch and has no native seam;
 * this header IS the seam a maintainer binds (ctypes stub: INTEGRATION.md).
 * Each entry point names the reference code it replaces (paths relative to the
 * reference repository root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless stated otherwise;
 *     the caller (PyTorch) owns all memory, including the workspace; nothing
 *     is allocated, freed or retained by the library;
 *   - all work is enqueued on `stream` (a hipStream_t); no call synchronises,
 *     so every call can be captured into a hipGraph;
 *   - return 0 on success, <0 for an argument/shape error, >0 a hipError_t;
 *     adamvs_last_error_string() describes the last failure on this thread;
 *   - re-entrant; the only global state is the thread-local error string and the option table below (process-wide
 *     integers that select between equivalent kernel forms; read on every call, written only by adamvs_set_option).
 *
 * Layouts ("channel-last"): feature maps [view][B][h*w][C], GRU states
 * [B][h*w][ch], cost-regularisation activations [N][h*w][D]; hypothesis planes
 * and output maps are [B][D][h*w] / [B][h*w] as in the reference.
 */
#ifndef ADAMVS_HIP_H
#define ADAMVS_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADAMVS_ABI_VERSION 16

int adamvs_version(void);
const char* adamvs_last_error_string(void);

/* ---- OPTIONS ---------------------------------------------------------------
 * Integers that choose between kernel forms of the SAME layer (every form is held to the same oracle; forms differ in the
 * order of their fp32 sums at most) and a few tuning limits.  adamvs_set_option takes effect at the next call of any entry
 * point, for every caller of the process (two callers that need different forms set the option before their calls; the
 * table is atomic integers, not locked state).  When the table is first touched each option is seeded from the environment
 * variable ADAMVS_<NAME IN CAPITALS> if that is set (A/B timing of an unmodified caller); nothing else in the compute path
 * reads the environment (two tuning tables of the recurrence's grid split: ADAMVS_RECUR_COSTS, ADAMVS_RECUR_COSTS_FUSED).
 *
 *   name                  default  meaning
 *   winograd                 1     CostRegNet2D (models/adamvs.py:229-238), fp32, widths that are multiples of 64: the five stride-1
 *                                  layers in the minimal-filtering form F(2x2, 3x3) (16 of 36 products); 0: direct kernels
 *   wino_softmax             1     ... its `prob` layer carries the softmax partials, no score volume (stage path); 0: score volume
 *   wino_wps                 0     ... 1 / 2: one / two workgroups per CU for every map size (same bits); 0: by map size
 *   fuse_softmax             1     direct `prob` kernel: softmax / max / depth regression in its epilogue; 0: k_softmax_regress
 *   s2_pairs                 1     CostRegNet2D: large stride-2 layers in the pair form along x (15 of 18 products); 0: direct
 *   conv_rows2              -1     CostRegNet2D: 2-row blocks for small grids: 0 never, 1 always, -1 by grid size
 *   t2_fused                -1     transposed layers, the four parity classes in one launch: 0 / 1 / -1 by grid size
 *   t2_kb8                   1     transposed layers at D = 192: two k-steps per chunk; 0: one
 *   costreg_defer_skips      1     the hourglass's skip additions formed by the consuming layer; 0: in the producer's epilogue
 *   conv256_split            1     D = 256 direct layers as two launches of 128 output channels; 0: one
 *   conv_small_grid       1024     MS-REDNet (models/msrednet.py): workgroups up to which a layer takes the resident form; 0: never
 *   red_fold_applies        -1     MS-REDNet: the GRU's element-wise applies folded into the next layer's prologue: 0 / 1 / -1 by batch
 *   conv1_f23                3     conv1 of SliceCostRegNetRED (adamvs.py:417) in the F(2, 3)-along-x form: bit 1 C = 32, bit 2 C = 16 / 8
 *   fconv_f23                1     FeatureNet0's stride-1 3x3 layers in the F(2, 3)-along-x form; 0: k_fconv
 *   gru_wino                 7     fp32 ConvGRU convolutions (module.py:24-52) in the F(2x2, 3x3) form where a role has a launch of
 *                                  its own and in the three-launch schedule: 1 gates1, 2 gates2, 4 cand2, 8 cand1; 0: direct kernels
 *                                  (the software-pipelined schedules 3 and 5 always use the direct / fused roles)
 *   recur_mode              -1     launches per hypothesis of the recurrence: 0 one role per launch (six; bf16x3: four), 1 three,
 *                                  3 two, 5 one (both levels one kernel each); -1: by stage size (adamvs_recurrence_schedule)
 */
int adamvs_option_count(void);
const char* adamvs_option_name(int index);                   /* 0 <= index < adamvs_option_count(); NULL outside */
int adamvs_option_default(const char* name, int* value);     /* -1: unknown name */
int adamvs_get_option(const char* name, int* value);
int adamvs_set_option(const char* name, int value);

/* ---- geometry ----------------------------------------------------------- */

/* T = P_src . P_ref^-1 for every (batch, source view); models/module.py:539-541.
 * proj [B][V][4][4] (view 0 = reference) -> rt [B][V-1][12] = {R row-major, t}. */
int adamvs_relative_transforms(const float* proj, float* rt, int B, int V, void* stream);

/* NCHW [B][C][h][w] -> channel-last [B][h*w][C] (one view slot) and back. C % 4 == 0. */
int adamvs_pack_features(const float* nchw, float* nhwc, int B, int C, int h, int w, void* stream);
int adamvs_unpack_features(const float* nhwc, float* nchw, int B, int C, int h, int w, void* stream);

/* get_depth_range_samples, models/module.py:646-663: depth_values [B][2]={min,max}
 * -> out [B][D][h][w], D uniform samples (the interval argument is ignored there). */
int adamvs_depth_range_samples_uniform(const float* depth_values, float* out, int B, int D, int h, int w, void* stream);
/* get_cur_depth_range_samples, models/module.py:628-643: window cur -+ D/2*interval,
 * D samples, no clamping.  cur_depth [B][h][w] -> out [B][D][h][w].  The interval is a double: the reference forms
 * ndepth / 2 * depth_inteval_pixel in Python floats and only the product is rounded to fp32 (module.py:632). */
int adamvs_depth_range_samples_window(const float* cur_depth, double depth_interval_pixel, float* out, int B, int D,
                                      int h, int w, void* stream);

/* F.interpolate(x, [ho,wo], mode='bilinear', align_corners=False) on [N][hi][wi];
 * models/adamvs.py:505 (view weights) and :522 (hypothesis plane). */
int adamvs_resize_bilinear(const float* in, float* out, int N, int hi, int wi, int ho, int wo, void* stream);

/* depth_regression, models/module.py:617-625: prob [B][D][h][w]; depth_values
 * [B][D] (hd = wd = 0) or [B][D][hd][wd] (bilinearly resized to [h][w]) -> out [B][h][w]. */
int adamvs_depth_regression(const float* prob, const float* depth_values, float* out, int B, int D, int h, int w,
                            int hd, int wd, void* stream);

/* homo_warping_float, models/module.py:527-568, on the reference's own layouts:
 * src_fea [B][C][h][w], rt [B][12] (adamvs_relative_transforms of this view),
 * depth_values [B][Nd][h][w] -> out [B][C][Nd][h][w]. */
int adamvs_homo_warp(const float* src_fea, const float* rt, const float* depth_values, float* out, int B, int C,
                     int Nd, int h, int w, void* stream);

/* ---- stage 1, per-view weighting (InferDepthNet0.forward pass A) --------- */

/* models/adamvs.py:464-478: sim[s][b][pix][d] = mean_c(ref[c] * warp_d(src_s)[c]).
 * feat [V][B][h*w][C], rt [B][S][12], planes [B][D][h*w] -> sim [S][B][h*w][D]. C in {8,16,32}. */
int adamvs_pair_similarity(const float* feat, const float* rt, const float* planes, float* sim, int B, int S, int C,
                           int D, int h, int w, void* stream);

/* CostRegNet2D.forward, models/adamvs.py:229-238, on x [N][h*w][D] -> score [N][h*w][D].
 * wpk: 11 layers (conv0..conv6, conv7, conv9, conv11, prob), each 9*D*D floats in
 * A-fragment order [tap][cin/4][cout/16][lane] with value
 * W[cout = 16*tile + (lane&15)][cin = 4*kc + (lane>>4)][tap] * bn_scale[cout]
 * (ConvTranspose2d layers: W[cin][cout][tap]) followed by D bias floats (folded BN shift,
 * or the conv bias for `prob`).  D in {16,32,48,64,96,128,192,256,384,512}; h, w multiples of 8.
 *
 * The reference builds the network for any number of hypotheses (adamvs.py:198-228); here a network of D hypotheses runs at
 * the next width of that list, adamvs_cost_reg_width(D, precision), with zero filters for the extra channels and -1e30 as
 * the bias of `prob`'s extra channels (ada-mvs_amd/packing.py::pack_cost_reg_net_2d): the extra channels stay 0 through the
 * hourglass and weigh exactly 0 in the softmax.  The op-level entry points below take the WIDTH as D (x and score carry
 * that many channels); adamvs_depth_stage_forward takes the number of hypotheses and handles the rest.
 *
 * fp32 with D in {64,128,192,256,384,512}: the 11 blocks are followed by the five stride-1 layers (conv0, conv2, conv4, conv6, prob)
 * in the minimal-filtering form F(2x2, 3x3), 16*D*D floats each, laid out as adamvs_conv3x3_dd_wino takes them; those
 * layers run on that kernel (fp32 throughout, 16 products instead of 36 per 2x2 outputs and channel pair; environment
 * ADAMVS_WINOGRAD=0: on the direct kernel).  ada-mvs_amd/packing.py::pack_cost_reg_net_2d produces exactly this.
 * wpk_floats = the length of the blob, adamvs_cost_reg_net_2d_weight_floats(D, precision): a blob of another layout
 * (e.g. the 11-block blob of ABI <= 10 at a width that now carries the F(2x2, 3x3) blocks) is refused, not read past its end.
 *
 * precision ADAMVS_PRECISION_FP32 (0): fp32 MFMA.  ADAMVS_PRECISION_BF16X3 (1): bf16 MFMA with every
 * operand split into two bf16 halves, a.b ~ a_hi.b_hi + a_hi.b_lo + a_lo.b_hi, fp32 accumulation (maps agree
 * with the fp32 path to ~1e-5); D must then be a multiple of 32 and each layer's 9*D*D-float block of wpk holds
 * bf16 fragments instead: [hi|lo][tap][cin/32][cout/16][lane][8] with element j of lane l =
 * W[cout = 16*tile + (l&15)][cin = 32*kb + 8*(l>>4) + j][tap] * bn_scale[cout] (same total size). */
#define ADAMVS_PRECISION_FP32 0
#define ADAMVS_PRECISION_BF16X3 1
size_t adamvs_cost_reg_net_2d_workspace_bytes(int N, int D, int h, int w);
int adamvs_cost_reg_width(int D, int precision);                       /* the width the network runs at for D hypotheses; 0: none (D > 512) */
size_t adamvs_cost_reg_net_2d_weight_floats(int D, int precision);     /* floats of wpk at width D; 0: not a supported width */
int adamvs_cost_reg_net_2d(const float* x, const float* wpk, size_t wpk_floats, float* score, int N, int D, int h, int w,
                           int precision, void* workspace, size_t workspace_bytes, void* stream);

/* One layer of CostRegNet2D: ConvBnReLU.forward (models/module.py:254-261) or the
 * ConvTranspose2d-BN-ReLU blocks of models/adamvs.py:212-225, BN folded into wpk/bias.
 * mode 0: 3x3 stride 1; 1: stride 2; 2: transposed stride 2 (k3 p1 op1).
 * in [N][hi*wi][D] -> out [N][ho*wo][D]; wpk as one layer above.  The hourglass's additions
 * (x = conv4 + conv7(x), adamvs.py:233-236) can sit on either side of a layer:
 *   skip [N][ho*wo][D] or NULL: added to this layer's output after the ReLU (producer side);
 *   in2  [N][hi*wi][D] or NULL: the layer convolves in + in2 (consumer side: the sum is formed while the
 *        window is staged, so a transposed layer's epilogue carries no second round of loads).
 * adamvs_cost_reg_net_2d uses in2 in fp32 and skip in bf16x3; the result is the same fp32 sum either way. */
int adamvs_conv3x3_dd(const float* in, const float* in2, const float* wpk, const float* bias, const float* skip, float* out,
                      int N, int D, int hi, int wi, int mode, int relu, int precision, void* stream);

/* A stride-1 layer of CostRegNet2D (conv0, conv2, conv4, conv6, prob: models/adamvs.py:205-227, ConvBnReLU.forward
 * models/module.py:254-261) in the minimal-filtering form F(2x2, 3x3): 16 products per 2x2 output tile and channel pair
 * instead of 36, fp32 throughout (results agree with adamvs_conv3x3_dd mode 0 to a few ulp of the accumulated sums).
 * wpk [D/4][4][D/16][64][4] = the transformed filters U = G w G^T (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], BN scale folded
 * in) as MFMA A fragments: element j of lane l of fragment (k-step kc, patch row i, channel tile) =
 * U[i][j][cout = 16*tile + (l&15)][cin = 4*kc + (l>>4)].  D a multiple of 64 up to 512; a map at most 2 GiB.  in, out, skip,
 * bias, relu as above. */
int adamvs_conv3x3_dd_wino(const float* in, const float* wpk, const float* bias, const float* skip, float* out, int N, int D,
                           int h, int w, int relu, void* stream);

/* models/adamvs.py:481-486 + module.py:617-625: softmax over D, its maximum (view
 * weight) and the expectation of the hypothesis planes (pair depth).
 * score [S*B][h*w][D], planes [B][D][h*w] -> view_weight, pair_depth [S][B][h*w]. */
int adamvs_softmax_max_regress(const float* score, const float* planes, float* view_weight, float* pair_depth,
                               int S, int B, int D, int h, int w, void* stream);

/* The last layer of CostRegNet2D (`prob`, adamvs.py:227, 238) with the view weighting above fused into its epilogue, the
 * way adamvs_depth_stage_forward runs it: score = conv3x3(in) + bias is reduced over D by the workgroup that
 * computes it and never stored.  in [S*B][h*w][D]; wpk / bias: the `prob` block of the packed weights (9*D*D + D floats);
 * planes [B][D][h*w] -> view_weight, pair_depth [S][B][h*w].  precision as adamvs_cost_reg_net_2d (wpk packed accordingly). */
int adamvs_prob_softmax_regress(const float* in, const float* wpk, const float* bias, const float* planes,
                                float* view_weight, float* pair_depth, int S, int B, int D, int h, int w, int precision,
                                void* stream);

/* The same with `prob` in the form F(2x2, 3x3) (what the fp32 stage runs at D a multiple of 64): the channel groups of a pixel
 * are different workgroups, so every lane stores the softmax partial (max, sum of exp, sum of exp * plane) of the 16 scores it
 * holds -- D bytes per pixel instead of the 4 D of the scores -- and a second kernel merges a pixel's D/16 partials.
 * The hypothesis planes are those of stage 1, uniform per tile: depth_range [B][2] = (first, last) plane of tile b, plane d =
 * first + d * ((last - first) / (D - 1)) (get_depth_range_samples, module.py:628-640); with per-pixel planes the stage takes
 * the score volume and adamvs_softmax_max_regress.  wpk as adamvs_conv3x3_dd_wino; workspace:
 * adamvs_prob_softmax_regress_wino_workspace_bytes (S*B*h*w*D bytes). */
size_t adamvs_prob_softmax_regress_wino_workspace_bytes(int S, int B, int D, int h, int w);
int adamvs_prob_softmax_regress_wino(const float* in, const float* wpk, const float* bias, const float* depth_range, float* view_weight,
                                     float* pair_depth, int S, int B, int D, int h, int w, void* workspace, size_t workspace_bytes,
                                     void* stream);

/* ---- aggregation + recurrent regularisation (pass B) --------------------- */

/* SliceCostRegNetRED weights (models/adamvs.py:400-413), packed by the host:
 * conv weights as MFMA A fragments [cout tile][tap][cin/4][64 lanes] holding
 * W[cout = 16*tile + (lane&15)][cin = 4*kc + (lane>>4)][tap] (0 beyond cout),
 * biases zero-padded to 16 per tile.  conv1 and cand1 (8 output channels each) use the two-row form: fragment
 * (rr, kx, kc), rr = 0..3, holds for lanes with (lane&15) < 8 the weights of output row y,
 * W[lane&15][cin][ky = rr][kx] (0 if rr = 3), and for the other lanes those of output row y+1,
 * W[(lane&15)-8][cin][ky = rr-1][kx] (0 if rr = 0), so one MFMA feeds two output rows.
 *
 * With precision ADAMVS_PRECISION_BF16X3 the conv1 / gates / cand / conv2 fields point to split-bf16 fragments
 * instead: the contraction index is flattened, k = pos*cin_total + cin (pos = tap ky*3+kx, or rr*3+kx for the
 * two-row conv1 and -- since ABI 15 -- the two-row cand1: 12 positions x 16 channels = 6 k-blocks, rows as in the fp32 two-row
 * form above), zero-padded to a multiple of 32; layout [cout tile][hi|lo][k/32][lane][8 bf16], element j of
 * lane l = W[16*tile + (l&15)][k = 32*kb + 8*(l>>4) + j].  upconv1 / final_w / the biases keep the fp32 form.
 *
 * PRE-SCALED FIELDS (since ABI 13; nothing in the struct's size or layout shows it, so a caller that packs its own blob must
 * do this or gets silently wrong GRU states).  The kernels that consume these fields feed the convolution's accumulator to
 * v_exp_f32 (2^x) directly -- sigmoid(x) = 1 / (1 + 2^(-x log2 e)), tanh(x) = 1 - 2 / (2^(2 x log2 e) + 1) -- so the factor is
 * folded into weights AND bias by the host, in double precision, before the fragments are formed (rounded once):
 *   ADAMVS_PRECISION_BF16X3:  gates1, gates1_b, gates2, gates2_b  = (-log2 e) x the reference tensors;
 *                             cand1, cand1_b, cand2, cand2_b       = (2 log2 e) x the reference tensors;
 *                             and the 16 output rows of gates1 / gates1_b are INTERLEAVED so that every lane of the fused
 *                             level-1 kernel holds two reset- and two update-gate values: MFMA row m = 4 q + e carries
 *                             reset-gate channel 2 q + e for e = 0, 1 and update-gate channel 2 q + e - 2 for e = 2, 3
 *                             (q = 0..3), i.e. rows = conv_gates rows [0, 1, 8, 9, 2, 3, 10, 11, 4, 5, 12, 13, 6, 7, 14, 15]
 *                             (conv_gates rows 0-7 = reset gate, 8-15 = update gate, module.py:35-41).
 *                             gates2 keeps the reference's row order.  conv1, conv2: unscaled.
 *   ADAMVS_PRECISION_FP32:    gates1 / gates2 / cand1 / cand2 and their biases are the UNSCALED reference tensors (the direct
 *                             kernels apply exp themselves); the *_w fields below are U = G (s g) G^T with s = -log2 e for
 *                             gates1_w, gates2_w and s = 2 log2 e for cand1_w, cand2_w, reference row order; the kernels that
 *                             read a *_w field scale the shared, unscaled bias by the same s once per launch.
 * ada-mvs_amd/packing.py::pack_slice_reg_net is the reference implementation of this format
 * (tests/test_host_logic.py::test_gru_prescaled_fields_follow_the_header recomputes it from this text). */
typedef struct adamvs_fuse_weights {
  const float* conv1;                           /* [12][C/4][64] two-row form reg_fuse.conv1.conv.weight */
  const float* gates1; const float* gates1_b;   /* [1][9][4][64], [16]      conv_gru1.conv_gates.0 */
  const float* cand1;  const float* cand1_b;    /* [12][4][64] two-row form like conv1 (fp32), [16]   conv_gru1.convc.0 */
  const float* conv2;                           /* [1][9][2][64]            conv2.conv.weight */
  const float* gates2; const float* gates2_b;   /* [2][9][8][64], [32]      conv_gru2.conv_gates.0 */
  const float* cand2;  const float* cand2_b;    /* [1][9][8][64], [16]      conv_gru2.convc.0 */
  const float* upconv1; const float* upconv1_b; /* [1][9][4][64], [16]      upconv1 (transposed: W[cin][cout][tap]) */
  const float* final_w;                         /* [73]: w[tap*8+c], bias   upconv2d */
  /* fp32 (NULL in bf16x3): the same gate / candidate convolutions as transformed filters U = G (s g) G^T of the minimal-filtering
   * form F(2x2, 3x3) (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]; s = -log2 e for the gates, 2 log2 e for the candidates: "PRE-SCALED
   * FIELDS" above), fragments [cout tile][patch row i][patch column j][cin/4][64]:
   * lane l = U[i][j][cout = 16*tile + (l&15)][cin = 4*kc + (l>>4)] */
  const float* gates1_w;                        /* [1][4][4][4][64]         conv_gru1.conv_gates.0 */
  const float* gates2_w;                        /* [2][4][4][8][64]         conv_gru2.conv_gates.0 */
  const float* cand2_w;                         /* [1][4][4][8][64]         conv_gru2.convc.0 */
  const float* cand1_w;                         /* [1][4][4][4][64]         conv_gru1.convc.0 (8 of 16 rows) */
} adamvs_fuse_weights;

/* ---- FeatureNet0.forward, reference models/adamvs.py:49-152 (blocks models/module.py:164-251,
 * 506-524), for N = B*V images at once.  imgs [N][3][H][W] (the reference's layout); outputs channel-last, the layout
 * every entry point above takes: stage1 [N][(H/4)(W/4)][32], stage2 [N][(H/2)(W/2)][16], stage3 [N][H*W][8].
 * H and W must be multiples of 32.  base_channels = 8 (the only configuration the reference uses).
 *
 * Weights (ada-mvs_amd/packing.py::pack_feature_net): every convolution as fp32 MFMA A fragments with the eval-mode
 * BatchNorm scale folded in, w = [cout tile][tap][cin/4][64 lanes], lane l = W[16*tile + (l&15)][4*kc + (l>>4)][tap]
 * (rows beyond cout zero), and b = the BatchNorm shift padded to 16 per tile (zeros for the plain output convs).
 * conv0_0 (RGB + a zero channel) and conv0_1 run fused in one kernel and are stored as TWO-ROW fragments [12][cin/4][64], the
 * layout of adamvs_fuse_weights.conv1 (rows 0-7: output row y, tap ky = rr; rows 8-15: row y+1, tap ky = rr-1; rr = 0..3).
 * The 5x5 stride-2 convolutions hold 25 taps; conv2_0 is stored as two
 * 16-channel halves.  deconv*_t hold the ConvTranspose2d(k3,s2,p1,op1) weights per output parity class (py,px):
 * 1 + 2 + 2 + 4 taps in the order 00, 01, 10, 11, tap (ty,tx) = kernel index (py ? (ty ? 0 : 2) : 1, same in x) applied
 * to input pixel (i+ty, j+tx).  deconv*_c convolve cat(deconv output, skip); deconv2_c (8 output channels) is stored as
 * two-row fragments [12][4][64] like conv0.  out_k multiply the feature map only;
 * the pooled-context branches br_k_j = {w1 [C/2][C] (BN folded), b1 [C/2], w2 [C][C/2] = columns of out_k for that
 * branch} are applied at pooled resolution and their bilinear upsampling (align_corners=False) is added in the
 * epilogue of out_k (the 1x1 convolution and the upsampling are both linear). */
typedef struct adamvs_fconv_weights { const float* w; const float* b; } adamvs_fconv_weights;
typedef struct adamvs_context_weights { const float* w1; const float* b1; const float* w2; } adamvs_context_weights;
typedef struct adamvs_feature_weights {
  adamvs_fconv_weights conv0_0, conv0_1, conv1_0, conv1_1, conv1_2, conv2_0, conv2_1, conv2_2;
  adamvs_fconv_weights out1, deconv1_t, deconv1_c, out2, deconv2_t, deconv2_c, out3;
  adamvs_context_weights br1_1, br1_2, br2_1, br2_2, br3_1, br3_2;
} adamvs_feature_weights;
size_t adamvs_feature_net0_workspace_bytes(int N, int H, int W);
int adamvs_feature_net0(const float* imgs, const adamvs_feature_weights* weights, float* stage1, float* stage2, float* stage3,
                        int N, int H, int W, void* workspace, size_t workspace_bytes, void* stream);
/* The same network on the images as reference Infer_AdaMVSNet.forward receives them, imgs [B][V][3][H][W] (adamvs.py:574-577
 * runs the net view by view), without the view-major copy: the launch computes images m = n0 .. n0+n-1 of the V*B images in
 * view-major order (m = v * B + b, the order of the feature maps every entry point above takes), reading imgs[b][v] in place;
 * stage1..3 receive those n images.  n0 / n let a caller bound the workspace (adamvs_feature_net0_workspace_bytes(n, H, W)). */
int adamvs_feature_net0_views(const float* imgs, const adamvs_feature_weights* weights, float* stage1, float* stage2, float* stage3,
                              int B, int V, int n0, int n, int H, int W, void* workspace, size_t workspace_bytes, void* stream);

/* The FPN variant of MS-REDNet's FeatureNet, reference models/msrednet.py:74-91 (constructor), 115-125 (forward), arch_mode
 * "fpn" with three stages: stage1 = out1(conv2); t1 = nearest2x(conv2) + inner1(conv1); stage2 = out2(t1);
 * t2 = nearest2x(t1) + inner2(conv0); stage3 = out3(t2).  Same images, outputs and size rule as adamvs_feature_net0.
 * conv* as in adamvs_feature_weights; out1 [32][32] 1x1 (no bias: b = zeros); inner1 [32][16], inner2 [32][8] 1x1 with
 * b = their bias; out2 [16][32], out3 [8][32] 3x3, b = zeros (fragment layout of adamvs_fconv_weights). */
typedef struct adamvs_feature_fpn_weights {
  adamvs_fconv_weights conv0_0, conv0_1, conv1_0, conv1_1, conv1_2, conv2_0, conv2_1, conv2_2;
  adamvs_fconv_weights out1, inner1, out2, inner2, out3;
} adamvs_feature_fpn_weights;
size_t adamvs_feature_net_fpn_workspace_bytes(int N, int H, int W);
int adamvs_feature_net_fpn(const float* imgs, const adamvs_feature_fpn_weights* weights, float* stage1, float* stage2,
                           float* stage3, int N, int H, int W, void* workspace, size_t workspace_bytes, void* stream);

/* models/adamvs.py:495-512 fused with conv1 of SliceCostRegNetRED (adamvs.py:416), for all
 * D hypotheses at once: c1[d][b][pix][8] = ReLU(conv1(sum_v w_v warp_v ref / (1e-5 + sum_v w_v))).
 * view_weight [S][B][h*w].  Hypothesis loop inside the thread, bilinear taps cached in registers
 * across planes; the similarity of a chunk of planes (at most 32) goes through the workspace and conv1 runs
 * over it as a tiled MFMA convolution.  On return the workspace holds the aggregated similarity of the LAST
 * chunk, [planes of the chunk][B][h*w][C] (with D <= 32: of all planes; the parity tests read it there). */
size_t adamvs_aggregate_conv1_workspace_bytes(int B, int C, int D, int h, int w);
int adamvs_aggregate_conv1(const float* feat, const float* rt, const float* planes, const float* view_weight,
                           const float* w1pk, float* c1, int B, int S, int C, int D, int h, int w, int precision,
                           void* workspace, size_t workspace_bytes, void* stream);

/* SliceCostRegNetRED.forward, models/adamvs.py:415-424 (one recurrent step).
 * cost [B][h*w][C]; state1 [B][h*w][8], state2 [B][(h/2)*(w/2)][16] updated in place;
 * reg_cost [B][Ho*Wo] with Ho x Wo = 2h x 2w (in_up) or h x w.
 * scratch: adamvs_slice_reg_step_scratch_bytes(B,h,w) bytes. */
size_t adamvs_slice_reg_step_scratch_bytes(int B, int h, int w);
int adamvs_slice_reg_step(const float* cost, float* state1, float* state2, const adamvs_fuse_weights* weights,
                          float* reg_cost, int B, int C, int h, int w, int in_up, int precision, void* scratch,
                          size_t scratch_bytes, void* stream);

/* ---- whole stage: InferDepthNet0.forward, models/adamvs.py:433-533 -------- */
typedef struct adamvs_stage_desc {
  int B, S, C, h, w, D;   /* batch, source views (any number), feature channels, feature rows/cols, hypotheses (first stage:
                             at most 512; CostRegNet2D runs at adamvs_cost_reg_width(D, precision) channels) */
  int in_up;              /* 1: maps come out at 2h x 2w (stages 1, 2); 0: h x w (stage 3) */
  int first_stage;        /* 1: confidence_map is None -> pass A scores the views (stage 1) */
  int prev_h, prev_w;     /* size of prev_conf maps when !first_stage */
  int precision;          /* ADAMVS_PRECISION_* for CostRegNet2D (w_reg must be packed accordingly) */
  int precision_fuse;     /* ADAMVS_PRECISION_* for conv1 and the ConvGRU convolutions (w_fuse packed accordingly) */
  int eps_in_numerator;   /* where the 1e-5 of the weighted aggregation sits: 0 = InferDepthNet0, sum_v w_v x_v / (1e-5 +
                             sum_v w_v) (adamvs.py:497-512); 1 = the train/test twin DepthNet0, (1e-5 + sum_v w_v x_v) /
                             sum_v w_v (adamvs.py:262-300) */
  int plane_mode;         /* what `planes` points to: ADAMVS_PLANES_EXPLICIT [B][D][h*w] (caller-made depth_values, as
                             InferDepthNet0.forward receives them); ADAMVS_PLANES_UNIFORM [B][2] = (min, max) -> plane d =
                             min + d (max - min)/(D - 1) (module.py:650-658); ADAMVS_PLANES_WINDOW cur_depth [B][h*w] ->
                             lo = cur - half_span, hi = cur + half_span, plane d = lo + d (hi - lo)/(D - 1) (module.py:628-643).
                             Generated planes equal the materialised ones bit for bit and never cross HBM. */
  float half_span;        /* ADAMVS_PLANES_WINDOW: ndepth / 2 * depth_interval_pixel (module.py:632) */
  const float* half_span_dev; /* ADAMVS_PLANES_WINDOW, optional (NULL: half_span above): DEVICE pointer to that one float.  The value
                             is then read by the kernels when they run, not baked into the launch: a captured hipGraph of the
                             stage serves tiles of any depth range (ada_mvs_amd/graphed.py; predict_whu.py's loop) */
} adamvs_stage_desc;
#define ADAMVS_PLANES_EXPLICIT 0
#define ADAMVS_PLANES_UNIFORM  1
#define ADAMVS_PLANES_WINDOW   2

size_t adamvs_depth_stage_workspace_bytes(const adamvs_stage_desc* desc);

/* How a stage of B*h*w pixels runs its recurrence (for accounting: bench.py prices executed flops): the schedule
 * (0: one role per launch, sequential; 1, 3, 5: software-pipelined, see option recur_mode) and, for
 * schedule 0 in fp32, the bit mask of the GRU convolutions that run in the minimal-filtering form F(2x2, 3x3)
 * (1 gates1, 2 gates2, 4 cand2, 8 cand1: 16 of the 36 products of the direct form; option gru_wino, default 7). */
int adamvs_recurrence_schedule(int precision_fuse, long long pixels);
int adamvs_gru_wino_mask(void);

/* phases of a stage.  VIEW_WEIGHTS may run in a call of its own: its results are the view_weight / pair_depth OUTPUT
 * tensors, which a later call reads back.  AGGREGATE, RECURRENCE and SOFT_ARGMIN form one chain over chunks of 32
 * hypotheses (the workspace holds one chunk of conv1 outputs and two of cost slices, nothing of it grows with D), so
 * for D > 32 they produce maps only when all three are in ONE call: a call with a proper subset of them returns an
 * argument error (-1).  For D <= 32 (one chunk) every subset is valid, in order. */
#define ADAMVS_PHASE_VIEW_WEIGHTS 1  /* pass A (pair similarity, CostRegNet2D, softmax) or resample of prev_conf */
#define ADAMVS_PHASE_AGGREGATE    2  /* weighted aggregation + conv1 */
#define ADAMVS_PHASE_RECURRENCE   4  /* D sequential ConvGRU encoder-decoder steps */
#define ADAMVS_PHASE_SOFT_ARGMIN  8  /* depth / confidence from the regularised slices */
#define ADAMVS_PHASE_ALL         15

/* feat [V=S+1][B][h*w][C]; rt [B][S][12]; planes: see adamvs_stage_desc.plane_mode;
 * prev_conf [S][B][prev_h*prev_w] (previous stage's view weights; ignored when first_stage);
 * w_reg: packed CostRegNet2D weights at width adamvs_cost_reg_width(D, precision), w_reg_floats of them
 *        (= adamvs_cost_reg_net_2d_weight_floats(width, precision); first_stage only);
 * outputs: view_weight [S][B][h*w] (what the next stage consumes as prev_conf),
 *          pair_depth [S][B][h*w] (first_stage only), depth / confidence [B][Ho*Wo].
 * phases: ADAMVS_PHASE_ALL, or VIEW_WEIGHTS alone followed by the other three together (see above). */
int adamvs_depth_stage_forward(const adamvs_stage_desc* desc, const float* feat, const float* rt, const float* planes,
                               const float* prev_conf, const float* w_reg, size_t w_reg_floats, const adamvs_fuse_weights* w_fuse,
                               float* view_weight, float* pair_depth, float* depth, float* confidence,
                               int phases, void* workspace, size_t workspace_bytes, void* stream);

/* MEASUREMENT ONLY -- not part of the inference path.  The same arguments; runs the selected phases of a stage alone over all
 * chunks on whatever the workspace holds, also where adamvs_depth_stage_forward refuses (a proper subset of
 * AGGREGATE|RECURRENCE|SOFT_ARGMIN at D > 32): the call's duration is the phase's, depth / confidence are NOT valid
 * afterwards.  bench.py's phase-by-phase table uses it, after the timed region. */
int adamvs_bench_stage_phase(const adamvs_stage_desc* desc, const float* feat, const float* rt, const float* planes,
                             const float* prev_conf, const float* w_reg, size_t w_reg_floats, const adamvs_fuse_weights* w_fuse,
                             float* view_weight, float* pair_depth, float* depth, float* confidence,
                             int phases, void* workspace, size_t workspace_bytes, void* stream);

/* ---- MS-REDNet inference (models/msrednet.py:330-436) ------------------------
 * The sibling model of predict_whu.py --model msrednet.  Its 3x3 convolutions run on adamvs_conv3x3_dd with the
 * channel counts zero-padded to a supported width; what follows are the pieces around them.  Maps are
 * channel-last [N][pixels][D]. */

/* Variance cost of the D hypothesis planes of a stage, msrednet.py:396-412: over the reference feature and the S source
 * features warped onto the plane (homo_warping_float, module.py:527-568): E[x^2] - E[x]^2 per channel, negated when
 * `negate` (both consumers take -cost, msrednet.py:351,362).  feat [V=S+1][B][h*w][C], rt [B][S][12],
 * planes [B][D][h*w] -> channels [0,C) of out_a [D][B][h*w][Da] (plane-major) and, when out_b != NULL, of
 * out_b [D][B][h*w][Db]; other channels untouched. */
int adamvs_red_variance_cost(const float* feat, const float* rt, const float* planes, float* out_a, int Da, float* out_b,
                             int Db, int B, int S, int C, int D, int h, int w, int negate, void* stream);

/* dst[b][p][dst_c0 + c] = src[b][p][src_c0 + c], c < n (strides in floats): narrows / widens / concatenates maps. */
int adamvs_channel_copy(const float* src, float* dst, int nbatch, int npix, int n, long src_batch_stride,
                        int src_pix_stride, int src_c0, long dst_batch_stride, int dst_pix_stride, int dst_c0, void* stream);

/* nn.GroupNorm(1, HC) statistics (module.py:63-68), in two deterministic halves.  _partial: for map g in {x0, x1}
 * (x1 may be NULL), over channels [0, n) and fixed pixel ranges of sample b: double-precision partial sums into
 * `partials` (adamvs_group_stats_workspace_bytes(N, ngroups)).  The two epilogues below finish the reduction
 * themselves; _finish does it standalone (same npix and n as _partial): stats[b][g] = {mean, 1/sqrt(biased var + eps)}. */
size_t adamvs_group_stats_workspace_bytes(int N, int ngroups);
int adamvs_group_stats_partial(const float* x0, const float* x1, int N, int npix, int D, int n, void* partials,
                               size_t partials_bytes, void* stream);
int adamvs_group_stats_finish(const void* partials, float* stats, int N, int ngroups, int npix, int n, float eps,
                              void* stream);

/* ConvGRUCell2.gates + the reset product, module.py:72-92.  The gate convolution is linear in cat(x, h):
 * gate_conv(cat(x, h)) = Wx.x + Wh.h + b, and so is the output convolution.  The x halves do not depend on the state
 * and are computed for all planes at once; per plane only the h halves remain, with the x halves added through the
 * `skip` operand of adamvs_conv3x3_dd.  fr, fu [N][npix][Wf]: reset / update halves (HC real channels each; two maps,
 * or fu = fr + HC inside one 2HC-wide map as adamvs_conv3x3_pair writes it);
 * partials from adamvs_group_stats_partial(fr, fu); gn [4][HC] = reset_gate_norm weight, bias, update_gate_norm
 * weight, bias; h [N][npix][W] the state.  -> rh = sigmoid(GN(fr)) * h [N][npix][W], u = sigmoid(GN(fu)) [N][npix][HC]. */
int adamvs_gru2_gates_apply(const float* fr, const float* fu, int Wf, const void* partials, const float* gn, const float* h,
                            float* rh, float* u, int N, int npix, int W, int HC, float eps, void* stream);

/* ConvGRUCell2.output + forward, module.py:91-106.  o [N][npix][W] = output_conv(cat(x, r*h)) (HC real channels);
 * partials from adamvs_group_stats_partial(o, NULL); gn [2][HC] = output_norm weight, bias.
 * h' = u*h + (1-u)*tanh(GN(o)) replaces h [N][npix][W] and goes to channels [0, HC) of out [N][npix][Wo] (may be NULL). */
int adamvs_gru2_out_apply(const float* o, const void* partials, const float* gn, const float* u, float* h, float* out,
                          int Wo, int N, int npix, int W, int HC, float eps, void* stream);

/* out = conv3x3(cat(srcA, srcB)) + bias, stride 1, zero padding, on COMPACT channel-last maps: srcA [B][h*w][CA],
 * srcB [B][h*w][CB] -> out [B][h*w][cout].  gate_conv / output_conv of ConvGRUCell2 (module.py:62-67) for the two shallow
 * levels of MS-REDNet, whose 8/16-channel states would waste a 16-wide k_conv_dd tile: weights stay in registers as
 * A fragments wpk [ceil(cout/16)][9][(CA+CB)/4][64] (value W[cout = 16*tile + (lane&15)][cin = 4*kc + (lane>>4)][tap]),
 * bias [16*ceil(cout/16)] zero padded.  (CA, CB, cout) in (32|16|8, 8, <=16) or (16, 16, <=32). */
int adamvs_conv3x3_pair(const float* srcA, int CA, const float* srcB, int CB, const float* wpk, const float* bias,
                        float* out, int cout, int B, int h, int w, void* stream);

/* One level's whole recurrence over the D planes of a stage (the loop of slice_RED_Regularization.forward restricted to
 * one ConvGRUCell2, msrednet.py:349-366), launched from native code: per plane the convolutions, the two GroupNorm
 * reductions and the two epilogues above (small maps: the partial sums come out of the convolutions' own epilogues, the
 * two gate convolutions of _split are one launch and, at one or two samples, the elementwise kernels are folded into the
 * window fill of the convolution that follows them -- two dependent launches per plane; ADAMVS_RED_FOLD_APPLIES=0 / 1
 * forces); the state starts at zero; h' of plane d goes to channels [0, HC) of
 * R [D][B][h*w][RW] (plane-major).  gn [6][HC] = reset / update / output norm weight, bias.
 * _pair (levels 1, 2): x [D][B][h*w][Cx] compact, wg / wc + bg / bc as for adamvs_conv3x3_pair (gate_conv with 2 HC
 * rows, output_conv).  _split (levels 3, 4): gxr, gxu, cx [D][B][h*w][W] = the x halves (+ bias) of the reset / update /
 * candidate convolutions, w_ghr / w_ghu / w_ch = the h halves as adamvs_conv3x3_dd blocks (9 W W fragment floats + W
 * zero bias floats each).  workspace: adamvs_red_recur_workspace_bytes(B, h, w, W, Wf, HC) with (W, Wf) = (HC, 2 HC)
 * for _pair and (W, W) for _split. */
size_t adamvs_red_recur_workspace_bytes(int B, int h, int w, int W, int Wf, int HC);
int adamvs_red_recur_pair(const float* x, int Cx, const float* wg, const float* bg, const float* wc, const float* bc,
                          const float* gn, float* R, int RW, int B, int D, int h, int w, int HC, float eps, void* workspace,
                          size_t workspace_bytes, void* stream);
int adamvs_red_recur_split(const float* gxr, const float* gxu, const float* cx, const float* w_ghr, const float* w_ghu,
                           const float* w_ch, const float* gn, float* R, int RW, int B, int D, int h, int w, int W, int HC,
                           float eps, void* workspace, size_t workspace_bytes, void* stream);

/* The running exp-sum / max / weighted-depth update of msrednet.py:415-436 (same as adamvs.py:512-531) in one pass over
 * the stored slices: vol [B][D][h*w] = reg_cost of every plane, planes [B][D][h*w] -> depth, confidence [B][h*w]. */
int adamvs_soft_argmin(const float* vol, const float* planes, float* depth, float* confidence, int B, int D, int h, int w,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADAMVS_HIP_H */
