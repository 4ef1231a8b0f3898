// MFMA energy probe for gfx950: the SAME flops issued as v_mfma_f32_32x32x16_bf16 or as v_mfma_f32_16x16x32_bf16, from
// registers only (no LDS, no memory in the loop), on random and on zero-filled operands, for several seconds each so that the
// package settles at its power cap.  Reports sustained TFLOP/s: under the cap that is a measure of energy per flop.
// Standalone: hipcc --offload-arch=gfx950 -O3 tools/mfmabench.hip -o tools/_build/mfmabench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// MODE 0: 8 accumulators of 32x32 (128 regs), 8 MFMAs 32x32x16 per inner step = 8 * 32768 flops
// MODE 1: 32 accumulators of 16x16 (128 regs), 16 MFMAs 16x16x32 per inner step = 16 * 16384 flops (same flops)
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * 256;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, src[(tid * 8 + i) & 0xffff]);
    b[i] = __builtin_bit_cast(bf16x8, src[(tid * 8 + 4 + i) & 0xffff]);
  }
  float s = 0.f;
  if (MODE == 0) {
    f32x16 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k & 3], b[k >> 1], acc[k], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][7] + acc[k][15];
  } else {
    f32x4 acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[k][e] = 0.f;
    for (int it = 0; it < iters; it += 2) {   // two inner steps per trip: all 32 accumulators live, static indices
#pragma unroll
      for (int q = 0; q < 32; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q & 3], b[(q >> 2) & 3], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) s += acc[k][0] + acc[k][3];
  }
  if (s == 12345.678f) out[tid] = s;   // keep the accumulators alive
}

template <int MODE>
static void run(const char* name, const u32x4* src, float* out, double secs) {
  const int iters = 4096, blocks = 256 * 2 * 4;   // 2 blocks / CU x 4 rounds
  probe<MODE><<<blocks, 256>>>(src, out, 64);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  double el = 0;
  while (el < secs) {
    for (int i = 0; i < 20; ++i) probe<MODE><<<blocks, 256>>>(src, out, iters);
    hipDeviceSynchronize();
    launches += 20;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double flops = (double)launches * blocks * 4 /*waves*/ * iters * 8.0 * 32768.0;
  printf("%-34s %8.1f TFLOP/s  (%.1f s)\n", name, flops / el / 1e12, el);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 4.0;
  const size_t n = 65536;
  u32x4* h = (u32x4*)malloc(n * sizeof(u32x4));
  u32x4 *dr, *dz;
  float* out;
  hipMalloc(&dr, n * sizeof(u32x4)); hipMalloc(&dz, n * sizeof(u32x4)); hipMalloc(&out, 4 << 20);
  srand(1);
  for (size_t i = 0; i < n; ++i)
    for (int j = 0; j < 4; ++j) {   // two random bf16 in [-1, 1): sign, exponent 0x3c..0x3f, random mantissa
      auto r = [&]() { unsigned m = rand() & 0x7f, e = 0x3c + (rand() & 3), s = rand() & 1; return (s << 15) | (e << 7) | m; };
      h[i][j] = r() | (r() << 16);
    }
  hipMemcpy(dr, h, n * sizeof(u32x4), hipMemcpyHostToDevice);
  hipMemset(dz, 0, n * sizeof(u32x4));
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("32x32x16  random operands", dr, out, secs);
    run<1>("16x16x32  random operands", dr, out, secs);
    run<0>("32x32x16  zero operands", dz, out, secs);
    run<1>("16x16x32  zero operands", dz, out, secs);
  }
  return 0;
}
