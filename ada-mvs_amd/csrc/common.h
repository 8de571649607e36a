// Shared helpers for the gfx950 kernels of the Ada-MVS depth-inference path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace adamvs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing (thread-local: DataParallel calls forward from one thread per device)
extern thread_local char g_last_error[512];
int set_error(int code, const char* fmt, ...);

#define ADAMVS_CHECK_ARG(cond, ...) \
  do { if (!(cond)) return ::adamvs::set_error(-1, __VA_ARGS__); } while (0)

#define ADAMVS_CHECK_LAUNCH(name) \
  do { hipError_t e_ = hipGetLastError(); \
       if (e_ != hipSuccess) return ::adamvs::set_error((int)e_, "%s: %s", name, hipGetErrorString(e_)); } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device helpers -------------------------------------------------------
// D[16 x 16] += A[16 x 4] . B[4 x 16], exact fp32 (v_mfma_f32_16x16x4_f32).
// lane l: a = A[l&15][l>>4], b = B[l>>4][l&15]; result reg r = D[4*(l>>4)+r][l&15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence: ~10 fewer VALU instructions per value in the
// gate epilogues, far inside the parity tolerance.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
  float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// The same with the argument of v_exp_f32 (2^x) prepared by the producer: xs = -x log2(e) for the sigmoid, 2 x log2(e) for tanh
// (the host folds the factor into the weights and the bias of the convolution whose output this is: one multiply per value less),
// and the GRU blend u h + (1 - u) c as c + u (h - c).  Used by the kernels whose weights are packed that way (packing.py).
__device__ __forceinline__ float sigmoid_pre(float xs) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xs)); }
__device__ __forceinline__ float tanh_pre(float xs) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(xs) + 1.0f); }
__device__ __forceinline__ f32x4 gru_blend(f32x4 u, f32x4 h, f32x4 c) { return c + u * (h - c); }

// s_waitcnt vmcnt(0) as a real instruction (not inline asm), so the compiler's own wait insertion knows that no
// vector-memory result is pending after it.  gfx9 encoding: vmcnt [3:0]+[15:14], expcnt [6:4], lgkmcnt [11:8].
__device__ __forceinline__ void wait_vmem_all() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// vmcnt(N): wait until at most the N youngest vector-memory operations are outstanding (they retire in order)
template <int N> __device__ __forceinline__ void wait_vmem_but() {
  static_assert(N >= 0 && N < 64, "vmcnt is six bits");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// ---- buffer addressing: workgroup-uniform base (128-bit descriptor in SGPRs) + 32-bit lane offset.
// No per-lane 64-bit address arithmetic, and the range check does the masking: a lane offset >= num_records
// (BUF_OOB) makes a load return zeros and a store disappear, so edge handling is one v_cndmask, not a branch.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t buf_rsrc;
constexpr unsigned BUF_OOB = 0x80000000u;
__device__ __forceinline__ buf_rsrc make_rsrc(const void* base) {      // base must be workgroup-uniform
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(buf_rsrc r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ void buf_store4(buf_rsrc r, unsigned byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 0);
}
// An MFMA result may be read only after the matrix pipe has written it back (8 passes for the 16x16 shapes used here: 11
// wait states); the hardware does not interlock, the compiler inserts s_nop.  Its hazard recognizer (ROCm 7.2, clang 22)
// counts wait states along ONE predecessor path when the first read sits behind a branch: for
//     chain; barrier; if (more) refill_lds(); epilogue(acc)
// it emitted `s_nop 0` on the path that skips the refill, and the last tile of every workgroup lost the final MFMA in
// accumulator rows 8-15.  drain() makes the accumulator a VGPR value in the basic block of the chain itself, where the
// count is linear and right; tools/mfma_hazard_lint.py checks the built library for the pattern.
__device__ __forceinline__ void drain(f32x4& acc) { asm volatile("" : "+v"(acc)); }

// Pin a per-lane constant in a register: computed once, never rematerialised or sunk into a branch.
template <typename T> __device__ __forceinline__ void pin(T& x) { asm volatile("" : "+v"(x)); }

// Uniform (scalar-cache) view of read-only parameters.
typedef const float __attribute__((address_space(4))) cfloat;
__device__ __forceinline__ cfloat* as_const(const float* p) { return (cfloat*)p; }

// LDS plane pitch (in floats) >= n with plane % 32 == 16: with channel planes
// that far apart the two 16-lane halves of a 32-lane LDS group (k = 0/1 of an
// MFMA B fragment) hit disjoint banks.
__host__ __device__ constexpr int plane_pitch16(int n) { return ((n + 15) / 32) * 32 + 16; }

}  // namespace adamvs
