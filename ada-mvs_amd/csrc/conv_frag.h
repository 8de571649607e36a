// 3x3 convolution of small-channel maps on the fp32 matrix cores.
//
// All the small convolutions of SliceCostRegNetRED (reference
// models/adamvs.py:400-424, models/module.py:5-52) are evaluated as
//     D[cout 16][pixel 16] += W_tap[cout 16][cin 4] . X_tap[cin 4][pixel 16]
// with v_mfma_f32_16x16x4_f32 (exact fp32).  A wave owns a run of 16
// consecutive output pixels of one row; weights sit in VGPRs as A fragments
// (packed on the host in fragment order), inputs are read from a planar LDS
// tile [cin][row][col] as B fragments.
#pragma once
#include "common.h"

namespace adamvs {

// Load NT x 9 x KC A-fragments; host layout [nt][tap][kc][64 lanes].
template <int NT, int KC>
__device__ __forceinline__ void load_wfrag(float (&wf)[NT][9][KC], const float* __restrict__ wpk, int lane) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) wf[nt][t][kc] = wpk[((nt * 9 + t) * KC + kc) * 64 + lane];
}

// LDS tile layout: channel c of pixel `pix` lives at (c >> 2) * GP + (c & 3) * PLANE + pix, with
// PLANE % 32 == 16 (the two k-rows of a 32-lane group of a B-fragment read hit disjoint banks) and
// GP = 4 * PLANE + 32 / (channel groups): when a tile is filled, the lanes of one ds_write cover a few pixels x
// all channel groups, and the extra 32/G floats per group spread those groups over the 32 banks (without
// them every group of a pixel lands on the same bank: an 8-way conflict for 32 input channels).
__host__ __device__ constexpr int group_pitch(int plane, int groups) { return 4 * plane + (groups >= 2 ? 32 / groups : 0); }

// One run of 16 output pixels, the lane's B-fragment origin (its own k-row and pixel) given as one pinned LDS byte
// offset per k-chunk: every read is base register + immediate (tap offsets fit the ds_read2 immediates), no address
// arithmetic in the chain.
template <int NT, int KC, int STRIDE, int PITCH>
__device__ __forceinline__ void conv3x3_run_at(f32x4 (&acc)[NT], const float (&wf)[NT][9][KC], const float* lds,
                                               const unsigned (&xbyte)[KC]) {
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        float bv = *(const float*)((const char*)lds + xbyte[kc] + (ky * PITCH + kx) * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(wf[nt][ky * 3 + kx][kc], bv, acc[nt]);
      }
}

}  // namespace adamvs
