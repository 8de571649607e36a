// Microbenchmark: what does v_mfma_f32_16x16x32_bf16 sustain on gfx950 in the instruction mixes of the bf16x3 kernels?
//   build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_bf16_issue.hip -o /tmp/mfma_bf16 && /tmp/mfma_bf16
// Prints cycles per MFMA per SIMD (16 = the 2.5 PFLOP/s peak at 2.4 GHz on 1024 SIMDs) for
//   reg      operands in registers, CH independent accumulators, W waves per SIMD
//   x3       the split-product order of the kernels: acc[m] += Ah.Bh (m = 0..2), += Ah.Bl, += Al.Bh per B fragment pair
//   lds      ... with the B fragment pair read from LDS (two ds_read_b128 per 9 MFMAs), one row ahead
//   ldsw     ... and the A fragments reloaded from global memory (L2-resident, 6 x 16 bytes per lane per 72 MFMAs), one tap ahead
//   bar      ... and two workgroup barriers per 648 MFMAs (the chunk structure of k_conv_dd_bx3)
//   valu     n independent vector instructions per MFMA in the same wave (the fused GRU kernels issue 1.7 - 2.5 per MFMA):
//            hidden under the matrix instruction, or added to it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

template <int CH>
__global__ __launch_bounds__(256) void k_reg(float* out, int iters, const bf16x8* src) {
  bf16x8 a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
  f32x4 acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 72; ++u) acc[u % CH] = mfma(a, b, acc[u % CH]);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < CH; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

// MODE 0: x3 order, operands in registers; 1: B from LDS; 2: + A from global; 3: + barriers per chunk
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_x3(float* out, int iters, const bf16x8* src) {
  __shared__ __attribute__((aligned(16))) __bf16 tile[2 * 180 * 40];
  for (int i = threadIdx.x; i < 2 * 180 * 40; i += 256) tile[i] = (__bf16)(float)(i & 7);
  __syncthreads();
  const int lane = threadIdx.x & 63, p = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
  bf16x8 wh[2][3], wl[2][3];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int m = 0; m < 3; ++m) { wh[s][m] = src[(s * 6 + m) * 64 + lane]; wl[s][m] = src[(s * 6 + 3 + m) * 64 + lane]; }
  f32x4 acc[3][8];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[m][r] = f32x4{0, 0, 0, 0};
  const char* lb = (const char*)tile + (p * 40 + 8 * q) * 2;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(src), 0, 0x7fffffff, 0x00020000);
  const unsigned wlane = lane * 16;
  for (int it = 0; it < iters; ++it) {                 // one "chunk": 9 taps x 8 rows x 9 MFMAs
    if (MODE >= 3) { __syncthreads(); __syncthreads(); }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int cur = t & 1, nxt = cur ^ 1;
      if (MODE >= 2) {                                  // next tap's fragments requested before this tap's MFMAs
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const unsigned f = (unsigned)((((it * 9 + t) * 4 + wave) * 6 + m) & 1023) * 1024;
          wh[nxt][m] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, f, 0));
          wl[nxt][m] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, f + 3072, 0));
        }
      }
      bf16x8 bh[8], bl[8];
      if (MODE >= 1) { bh[0] = *(const bf16x8*)(lb + (t % 3) * 80); bl[0] = *(const bf16x8*)(lb + (t % 3) * 80 + 180 * 80); }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (MODE >= 1) {
          if (r + 1 < 8) {
            bh[r + 1] = *(const bf16x8*)(lb + ((r + 1) * 18 + t % 3) * 80);
            bl[r + 1] = *(const bf16x8*)(lb + ((r + 1) * 18 + t % 3) * 80 + 180 * 80);
          }
        } else {
          bh[r] = wh[cur][r % 3]; bl[r] = wl[cur][(r + 1) % 3];
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m][r] = mfma(wh[cur][m], bh[r], acc[m][r]);
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m][r] = mfma(wh[cur][m], bl[r], acc[m][r]);
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m][r] = mfma(wl[cur][m], bh[r], acc[m][r]);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  f32x4 s = {0, 0, 0, 0};
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int r = 0; r < 8; ++r) s += acc[m][r];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

// NV independent vector instructions per MFMA in the SAME wave (v_fma_f32 on private registers; QUARTER: v_exp_f32, the quarter-rate
// transcendental of the gate epilogues): do they hide under the 16 cycles of the matrix instruction?
template <int NV, bool QUARTER>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters, const bf16x8* src) {
  bf16x8 a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + 0.001f * (float)(threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 72; ++u) {
      acc[u & 3] = mfma(a, b, acc[u & 3]);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        float& x = v[(u * NV + k) & 7];
        if (QUARTER) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(v[(u + k + 1) & 7]));
      }
    }
  }
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  float t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) t += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w + t;
}

// the vector instructions alone (no MFMA): their own cost per "MFMA slot"
template <int NV, bool QUARTER>
__global__ __launch_bounds__(256) void k_valu_only(float* out, int iters, const bf16x8* src) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + 0.001f * (float)(threadIdx.x + i) + (float)((const unsigned short*)src)[threadIdx.x & 7];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 72; ++u)
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        float& x = v[(u * NV + k) & 7];
        if (QUARTER) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(v[(u + k + 1) & 7]));
      }
  }
  float t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) t += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <class K>
static double run(K kern, int wps, int iters, int mfma_per_iter, float* out, const bf16x8* src, const char* name) {
  const int grid = 256 * wps;                            // workgroups of 4 waves: `wps` waves per SIMD
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 4, src);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, src);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * mfma_per_iter * wps);      // per MFMA per SIMD
  printf("%-44s %d wave(s)/SIMD: %6.2f cycles per MFMA per SIMD  (%5.1f %% of the 16-cycle peak)\n", name, wps, cyc, 100.0 * 16.0 / cyc);
  return cyc;
}

int main() {
  float* out;
  bf16x8* src;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  hipMalloc(&src, 4 << 20);
  std::vector<unsigned short> h((4 << 20) / 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3f80 + (unsigned short)(i & 3);      // bf16 values near 1
  hipMemcpy(src, h.data(), 4 << 20, hipMemcpyHostToDevice);
  for (int wps = 1; wps <= 2; ++wps) {
    run(k_reg<1>, wps, 2000, 72, out, src, "reg, 1 accumulator (dependent chain)");
    run(k_reg<2>, wps, 2000, 72, out, src, "reg, 2 accumulators");
    run(k_reg<4>, wps, 2000, 72, out, src, "reg, 4 accumulators");
    run(k_reg<24>, wps, 2000, 72, out, src, "reg, 24 accumulators");
    run(k_x3<0>, wps, 300, 648, out, src, "x3 order, operands in registers");
    run(k_x3<1>, wps, 300, 648, out, src, "x3 order, B from LDS one row ahead");
    run(k_x3<2>, wps, 300, 648, out, src, "x3 order, B from LDS, A from L2 one tap ahead");
    run(k_x3<3>, wps, 300, 648, out, src, "... and two barriers per 648 MFMAs");
    run(k_valu<1, false>, wps, 2000, 72, out, src, "reg, 4 acc + 1 v_fma_f32 per MFMA");
    run(k_valu<2, false>, wps, 2000, 72, out, src, "reg, 4 acc + 2 v_fma_f32 per MFMA");
    run(k_valu<3, false>, wps, 2000, 72, out, src, "reg, 4 acc + 3 v_fma_f32 per MFMA");
    run(k_valu<4, false>, wps, 2000, 72, out, src, "reg, 4 acc + 4 v_fma_f32 per MFMA");
    run(k_valu<1, true>, wps, 2000, 72, out, src, "reg, 4 acc + 1 v_exp_f32 per MFMA");
    run(k_valu<2, true>, wps, 2000, 72, out, src, "reg, 4 acc + 2 v_exp_f32 per MFMA");
    run(k_valu_only<3, false>, wps, 2000, 72, out, src, "(3 v_fma_f32 per slot alone, no MFMA)");
    run(k_valu_only<2, true>, wps, 2000, 72, out, src, "(2 v_exp_f32 per slot alone, no MFMA)");
  }
  return 0;
}
