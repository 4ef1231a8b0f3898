// LDS-DMA issue-cost probe (gfx950): cycles a wave spends per `buffer_load_dwordx4 ... lds` (1 KiB per wave-instruction) when it
// issues them back to back from an L2-resident source, with 1 / 2 waves per SIMD.  The NT / TN kernels issue 6 - 8 of them per
// K-step per wave next to 16 - 64 MFMAs.   hipcc --offload-arch=gfx950 -O3 tools/dmabench.hip -o tools/_build/dmabench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

template <int PER_WAIT>
__global__ __launch_bounds__(512) void probe(const char* src, int iters, unsigned long long* out, int span) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * span), 0, span, 0x00020000);
  char* base = smem + wid * 8192;
  int voff = lane * 16 + wid * 8192;   // every wave streams its own 128-KiB slice (no sharing through the vector L1)
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < PER_WAIT; ++i) glds16(r, base + (i & 7) * 1024, voff + ((it * PER_WAIT + i) * 1024) % 8192, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
}

// the register path: global_load_dwordx4 (16 B per lane) into VGPRs, then ds_write_b128; same bytes, same source
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int PER_WAIT>
__global__ __launch_bounds__(512) void probe_reg(const char* src, int iters, unsigned long long* out, int span) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const char* s = src + (size_t)blockIdx.x * span;
  char* base = smem + wid * 8192;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[PER_WAIT];
#pragma unroll
    for (int i = 0; i < PER_WAIT; ++i) v[i] = *(const u32x4*)(s + (lane * 16 + wid * 8192 + ((it * PER_WAIT + i) * 1024) % 8192));
#pragma unroll
    for (int i = 0; i < PER_WAIT; ++i) *(u32x4*)(base + (i & 7) * 1024 + lane * 16) = v[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
}
// both at once: half of the bytes by LDS-DMA, half through registers
template <int PER_WAIT>
__global__ __launch_bounds__(512) void probe_mix(const char* src, int iters, unsigned long long* out, int span) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const char* s = src + (size_t)blockIdx.x * span;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)s, 0, span, 0x00020000);
  char* base = smem + wid * 8192;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[PER_WAIT / 2];
#pragma unroll
    for (int i = 0; i < PER_WAIT / 2; ++i) v[i] = *(const u32x4*)(s + (lane * 16 + wid * 8192 + ((it * PER_WAIT + i) * 1024) % 8192));
#pragma unroll
    for (int i = 0; i < PER_WAIT / 2; ++i) glds16(r, base + (i & 3) * 1024, lane * 16 + wid * 8192 + ((it * PER_WAIT + PER_WAIT / 2 + i) * 1024) % 8192, 0);
#pragma unroll
    for (int i = 0; i < PER_WAIT / 2; ++i) *(u32x4*)(base + 4096 + (i & 3) * 1024 + lane * 16) = v[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
}
template <int MODE, int PER_WAIT>
static void run2(int waves, const char* src, unsigned long long* d) {
  const int iters = 2000, span = 1 << 16;
  auto k = MODE == 1 ? probe_reg<PER_WAIT> : probe_mix<PER_WAIT>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  k<<<256, waves * 64, 65536>>>(src, 50, d, span);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<<<256, waves * 64, 65536>>>(src, iters, d, span);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * PER_WAIT;
  printf("%d waves/CU, %2d KiB per wait, %s: %7.1f cycles per KiB per wave (wave 0), %6.2f TB/s chip-wide\n", waves, PER_WAIT,
         MODE == 1 ? "global_load + ds_write_b128" : "half LDS-DMA, half registers ", h[0] / n, n * waves * 1024.0 * 256 / (ms * 1e-3) / 1e12);
}

template <int PER_WAIT>
static void run(int waves, const char* src, unsigned long long* d) {
  const int iters = 2000, span = 1 << 16;   // 1 MiB per block: L2-resident after the warm-up launch
  hipFuncSetAttribute((const void*)probe<PER_WAIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  probe<PER_WAIT><<<256, waves * 64, 65536>>>(src, 50, d, span);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<PER_WAIT><<<256, waves * 64, 65536>>>(src, iters, d, span);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * PER_WAIT;
  printf("%d waves/CU, %2d DMAs per vmcnt(0): %7.1f cycles per DMA per wave (wave 0), %6.2f TB/s chip-wide\n", waves, PER_WAIT, h[0] / n,
         n * waves * 1024.0 * 256 / (ms * 1e-3) / 1e12);
}

int main() {
  char* src;
  unsigned long long* d;
  hipMalloc(&src, (size_t)256 << 20);
  hipMemset(src, 1, (size_t)256 << 20);
  hipMalloc(&d, 256 * 8 * 8);
  for (int w : {4, 8}) {
    run<1>(w, src, d);
    run<6>(w, src, d);
    run<24>(w, src, d);
    run2<1, 6>(w, src, d);
    run2<1, 16>(w, src, d);
    run2<2, 6>(w, src, d);
    run2<2, 16>(w, src, d);
  }
  return 0;
}
