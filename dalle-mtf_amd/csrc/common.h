// common.h -- shared device helpers for libdalle_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dalle_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WAVE 64

// s_setprio(1) around an MFMA cluster (cdna_hip_programming.md T5): with several blocks per CU in different phases the
// scheduler then prefers the wave that is feeding the matrix pipe over co-resident waves issuing DMA / LDS traffic.
// Measured in the dalle_example step (profiles/r03_ab_setprio_groupm.log, same-call A/B): NT GEMMs 16.84 -> 16.74 ms/step
// (the BK = 64 kernel +3 % on the K >= 1536 shapes, the 256x128 kernel neutral); the same around the attention kernels'
// S / PV clusters: neutral to -0.5 % (not applied).
#define MFMA_PRIO(x) __builtin_amdgcn_s_setprio(x)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // RNE (v_cvt_pk_bf16_f32 on gfx950)
  return __builtin_bit_cast(unsigned short, b);
}
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  const pk_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pk_bf16x2));  // one v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
  return v;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}

// ------------------------------------------------------------------ host-side error plumbing
void dmi_set_error(const char* fmt, ...);
#define DMI_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      dmi_set_error(__VA_ARGS__);         \
      return DMI_ERR_INVALID;             \
    }                                     \
  } while (0)
#define DMI_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      dmi_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return DMI_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
