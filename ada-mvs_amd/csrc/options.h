// Run-time options of the library: which of two equivalent kernel forms a layer takes, and a few tuning limits.
//
// One table instead of environment reads scattered over the launchers: every option has a name, a default and a line in
// include/adamvs_hip.h ("OPTIONS"); `adamvs_set_option(name, value)` changes one at any time (the launchers read the table on
// every call, nothing is cached), `adamvs_get_option` reads one.  The environment is consulted ONCE, when the table is first
// touched: `ADAMVS_<NAME>=<int>` seeds option <name> (A/B timing without touching the caller).  No option changes WHAT is
// computed -- every form of a layer is held to the same oracle; they differ in rounding order at most (each entry says).
#pragma once

namespace adamvs {

// X(enumerator, "name", default)
#define ADAMVS_OPTION_LIST(X)                                                                                                     \
  X(OPT_WINOGRAD, "winograd", 1)                     /* CostRegNet2D: stride-1 layers in the F(2x2, 3x3) form (fp32, D % 64 == 0) */ \
  X(OPT_WINO_SOFTMAX, "wino_softmax", 1)             /* ... `prob` carries the softmax partials (no score volume) */                 \
  X(OPT_WINO_WPS, "wino_wps", 0)                     /* ... 1 / 2 workgroups per CU; 0 = by map size */                              \
  X(OPT_FUSE_SOFTMAX, "fuse_softmax", 1)             /* direct `prob` kernel: softmax / max / regression in its epilogue */           \
  X(OPT_S2_PAIRS, "s2_pairs", 1)                     /* CostRegNet2D: large stride-2 layers in the pair form along x */               \
  X(OPT_CONV_ROWS2, "conv_rows2", -1)                /* CostRegNet2D: 2-row blocks on small grids; -1 = by grid size */               \
  X(OPT_T2_FUSED, "t2_fused", -1)                    /* transposed layers: four classes per launch; -1 = by grid size */              \
  X(OPT_T2_KB8, "t2_kb8", 1)                         /* transposed layers at D = 192: two k-steps per chunk */                        \
  X(OPT_COSTREG_DEFER_SKIPS, "costreg_defer_skips", 1) /* skip additions formed by the consuming layer */                           \
  X(OPT_CONV256_SPLIT, "conv256_split", 1)           /* D = 256: two launches of 128 output channels */                               \
  X(OPT_CONV_SMALL_GRID, "conv_small_grid", 1024)    /* MS-REDNet: workgroups up to which the resident form is used (0: never) */     \
  X(OPT_RED_FOLD_APPLIES, "red_fold_applies", -1)    /* MS-REDNet: GRU applies folded into the next layer; -1 = by batch */           \
  X(OPT_CONV1_F23, "conv1_f23", 3)                   /* conv1 in the F(2, 3)-along-x form: bit 1 C = 32, bit 2 C = 16 / 8 */          \
  X(OPT_FCONV_F23, "fconv_f23", 1)                   /* FeatureNet0: stride-1 3 x 3 layers in the F(2, 3)-along-x form */             \
  X(OPT_GRU_WINO, "gru_wino", 7)                     /* fp32 GRU convolutions in the F(2x2, 3x3) form: 1 gates1, 2 gates2, 4 cand2 */ \
  X(OPT_RECUR_MODE, "recur_mode", -1)                /* launches per hypothesis of the recurrence: 0 / 1 / 3 / 5; -1 = by size */

enum Option {
#define ADAMVS_OPTION_ENUM(e, n, d) e,
  ADAMVS_OPTION_LIST(ADAMVS_OPTION_ENUM)
#undef ADAMVS_OPTION_ENUM
  OPT_COUNT
};

int opt(Option o);            // the current value (api.hip)

}  // namespace adamvs
