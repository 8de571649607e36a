// What a dependent kernel launch costs inside a replayed hipGraph on this machine (tools/microbench: not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
// Chains of N dependent kernels on one stream, captured once, replayed: microseconds per kernel for
//   empty        <<<1, 64>>>, <<<256, 256>>>, <<<1024, 256>>> with no memory access
//   touch        every workgroup loads one cache line written by the previous kernel and stores one (the shape of a recurrence step)
//   big args     the same with a 1 KB kernel-argument struct (the slot kernels pass 300-600 bytes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

struct Big { float pad[250]; float* p; };
__global__ void k_empty() {}
__global__ void k_touch(float* a, float* b) { const int i = blockIdx.x * 64; if (threadIdx.x == 0) b[i] = a[(i + 64) % (gridDim.x * 64)] + 1.f; }
__global__ void k_big(Big g, float* b) { const int i = blockIdx.x * 64; if (threadIdx.x == 0) b[i] = g.p[(i + 64) % (gridDim.x * 64)] + g.pad[blockIdx.x % 250]; }

template <typename F> static double chain(hipStream_t st, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; ++i) launch(i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return us / reps / n;
}

int main() {
  hipStream_t st; hipStreamCreate(&st);
  float *a, *b; hipMalloc(&a, 1 << 22); hipMalloc(&b, 1 << 22); hipMemset(a, 0, 1 << 22); hipMemset(b, 0, 1 << 22);
  const int N = 2000;
  const int grids[3] = {1, 256, 1024};
  for (int g : grids) {
    printf("empty  <<<%4d, 256>>>  %.2f us per kernel\n", g, chain(st, N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, st); }));
    printf("touch  <<<%4d, 256>>>  %.2f us per kernel\n", g, chain(st, N, [&](int i) { hipLaunchKernelGGL(k_touch, dim3(g), dim3(256), 0, st, (i & 1) ? a : b, (i & 1) ? b : a); }));
    Big big; big.p = a;
    for (int i = 0; i < 250; ++i) big.pad[i] = 0.f;
    printf("bigarg <<<%4d, 256>>>  %.2f us per kernel\n", g, chain(st, N, [&](int i) { big.p = (i & 1) ? a : b; hipLaunchKernelGGL(k_big, dim3(g), dim3(256), 0, st, big, (i & 1) ? b : a); }));
  }
  return 0;
}
