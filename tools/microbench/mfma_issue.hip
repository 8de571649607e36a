// Microbenchmark: issue behaviour of v_mfma_f32_16x16x4_f32 on gfx950.
//   chains   1 / 2 / 4 independent accumulators per wave
//   file     accumulators in AGPRs or in arch VGPRs
//   waves    1..4 waves per SIMD (grid = 256 CUs x waves workgroups of 256 threads)
//   valu     n extra independent VALU instructions (v_fma_f32) per MFMA, to see whether they hide under the MFMA
//   lds      B operand read from LDS: one ds_read_b32 per MFMA, or one ds_read_b96 per 3 MFMAs (LDS < 0)
// Build: hipcc -O3 --offload-arch=gfx950 mfma_issue.hip -o mfma_issue ; prints cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int LDSMODE>
__global__ __launch_bounds__(256) void kl(float* out, int iters, float a, float b) {
  __shared__ float tile[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) tile[i] = (float)i;
  __syncthreads();
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const int lane = threadIdx.x & 63;
  const float* xb = tile + (lane >> 4) * 208 + (lane & 15) + (threadIdx.x >> 6) * 36;
  for (int i = 0; i < iters; ++i) {
    if (LDSMODE % 10 == 1) {            // one ds_read_b32 (or ds_read2) per MFMA, compiler-scheduled
#pragma unroll
      for (int u = 0; u < 48; ++u) acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[(u % 3) + (u / 3) * 18 + (i & 1) * 836], acc[u % CHAINS], 0, 0, 0);
    } else if (LDSMODE % 10 == 2) {     // like the two-row conv1 chain: 4 row pairs share rows
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int rp = 0; rp < 4; ++rp) acc[rp % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[(2 * rp + rr) * 18 + kx + (i & 1) * 836], acc[rp % CHAINS], 0, 0, 0);
    } else {                       // no LDS
#pragma unroll
      for (int u = 0; u < 48; ++u) acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % CHAINS], 0, 0, 0);
    }
    if (LDSMODE >= 10) __syncthreads();
  }
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

// operands from memory: DATA 0 = all lanes 1.0 / 0.5, 1 = random normal-ish values (register toggling -> power)
template <int DATA>
__global__ __launch_bounds__(256) void kd(float* out, const float* vals, int iters) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = DATA ? vals[(threadIdx.x * 16 + i) & 4095] : 1.0f;
    b[i] = DATA ? vals[(threadIdx.x * 16 + 8 + i) & 4095] : 0.5f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u)
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[u & 3]) : "v"(a[u & 7]), "v"(b[(u * 3) & 7]));
  }
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

template <int DATA>
void rund(const char* name, float* out, const float* vals, int waves, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kd<DATA>), dim3(256 * waves), dim3(256), 0, 0, out, vals, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((kd<DATA>), dim3(256 * waves), dim3(256), 0, 0, out, vals, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 32 * waves;
  const double ns_per = ms * 1e6 / mfma_per_simd;
  printf("%-34s waves/SIMD %d  %7.3f ms  %6.2f ns per MFMA per SIMD  (%5.1f cycles @2.4GHz)  %6.1f TFLOP/s\n", name, waves, ms,
         ns_per, ns_per * 2.4, 2048.0 * mfma_per_simd * 1024 / (ms * 1e-3) / 1e12);
}

template <int CHAINS, int LDSMODE>
void runl(const char* name, float* out, int waves) {
  const int iters = 1000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kl<CHAINS, LDSMODE>), dim3(256 * waves), dim3(256), 0, 0, out, 10, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((kl<CHAINS, LDSMODE>), dim3(256 * waves), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 48 * waves;
  const double ns_per = ms * 1e6 / mfma_per_simd;
  printf("%-34s waves/SIMD %d  %7.3f ms  %6.2f ns per MFMA per SIMD  (%5.1f cycles @2.4GHz)  %6.1f TFLOP/s\n", name, waves, ms,
         ns_per, ns_per * 2.4, 2048.0 * mfma_per_simd * 1024 / (ms * 1e-3) / 1e12);
}

template <int CHAINS, bool AGPR, int VALU>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float x0 = a, x1 = b, x2 = a + 1, x3 = b + 1;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c = u % CHAINS;
      if (AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
#pragma unroll
      for (int v = 0; v < VALU; ++v) {
        if ((v & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
        if ((v & 3) == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
        if ((v & 3) == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
        if ((v & 3) == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
      }
    }
  }
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w + x0 + x1 + x2 + x3;
}

template <int CHAINS, bool AGPR, int VALU>
void run(const char* name, float* out, int waves) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, AGPR, VALU>), dim3(256 * waves), dim3(256), 0, 0, out, 10, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS, AGPR, VALU>), dim3(256 * waves), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 16 * waves;            // each workgroup puts one wave on every SIMD
  const double ns_per = ms * 1e6 / mfma_per_simd;
  printf("%-34s waves/SIMD %d  %7.3f ms  %6.2f ns per MFMA per SIMD  (%5.1f cycles @2.4GHz)  %6.1f TFLOP/s\n", name, waves, ms,
         ns_per, ns_per * 2.4, 2048.0 * mfma_per_simd * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  {
    std::vector<float> hv(4096);
    unsigned x = 12345;
    for (auto& v : hv) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 32768.0f - 1.0f; v *= 1.7f; }
    float* vals;
    hipMalloc(&vals, 4096 * sizeof(float));
    hipMemcpy(vals, hv.data(), 4096 * sizeof(float), hipMemcpyHostToDevice);
    for (int iters : {1000, 20000, 200000}) {
      rund<0>("constant operands", out, vals, 2, iters);
      rund<1>("random operands", out, vals, 2, iters);
    }
  }
  for (int waves : {1, 2, 3, 4, 5}) {
    runl<1, 0>("builtin, 1 chain, no LDS", out, waves);
    runl<4, 0>("builtin, 4 chains, no LDS", out, waves);
    runl<1, 1>("1 chain, B from LDS (1 read/MFMA)", out, waves);
    runl<4, 1>("4 chains, B from LDS (1 read/MFMA)", out, waves);
    runl<4, 2>("4 chains, conv1 pattern", out, waves);
    runl<4, 12>("conv1 pattern + barrier / 48 MFMA", out, waves);
    runl<1, 11>("1 chain LDS + barrier / 48 MFMA", out, waves);
  }
  for (int waves : {2}) {
    run<1, true, 0>("1 chain, AGPR", out, waves);
    run<4, true, 0>("4 chains, AGPR", out, waves);
    run<1, false, 0>("1 chain, VGPR", out, waves);
    run<4, false, 0>("4 chains, VGPR", out, waves);
    run<4, true, 1>("4 chains, AGPR, +1 VALU/MFMA", out, waves);
    run<4, true, 4>("4 chains, AGPR, +4 VALU/MFMA", out, waves);
    run<4, true, 7>("4 chains, AGPR, +7 VALU/MFMA", out, waves);
    run<1, true, 4>("1 chain, AGPR, +4 VALU/MFMA", out, waves);
  }
  return 0;
}
