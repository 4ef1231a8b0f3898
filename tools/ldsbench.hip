// LDS read-rate probe for gfx950: how many clocks per wave-instruction do ds_read_b64_tr_b16 / ds_read_b64 / ds_read_b128 cost
// when 8 waves of a CU issue them back to back (the weight-gradient kernels read every MFMA operand through the transposing
// form).  Standalone: hipcc --offload-arch=gfx950 tools/ldsbench.hip -o tools/_build/ldsbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int ROWB>
__global__ __launch_bounds__(512) void probe(int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < 65536 / 4; i += 512) ((int*)smem)[i] = i;
  __syncthreads();
  const int h = lane >> 5, g4 = lane >> 4, l16 = lane & 15, rr = l16 >> 2;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned addr;
  if (MODE == 2) addr = lds0 + (lane & 31) * ROWB / 8 * 0 + lane * 16 + wid * 1024;   // linear 16-B reads
  else addr = lds0 + (8 * h + rr) * ROWB + (((wid & 1) * 128 + 32 * (g4 & 1) + 8 * (l16 & 3)) ^ (64 * rr));
  u32x4 s = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      u32x2 a, b, c, d;
      asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:2048\n\t"
                   "ds_read_b64_tr_b16 %2, %4 offset:64\n\tds_read_b64_tr_b16 %3, %4 offset:2112\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr) : "memory");
      s[0] ^= a[0] ^ b[0]; s[1] ^= c[1] ^ d[1];
    } else if (MODE == 1) {
      u32x2 a, b, c, d;
      asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:2048\n\t"
                   "ds_read_b64 %2, %4 offset:64\n\tds_read_b64 %3, %4 offset:2112\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr) : "memory");
      s[0] ^= a[0] ^ b[0]; s[1] ^= c[1] ^ d[1];
    } else {
      u32x4 a, b;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
      s[0] ^= a[0] ^ b[0]; s[1] ^= a[3] ^ b[3];
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
  if (s[0] == 0x12345 && s[1] == 0x54321) out[0] = 0;
}

template <int MODE, int ROWB>
static void run(const char* name, int nwaves) {
  unsigned long long* d;
  hipMalloc(&d, 256 * 8 * 8);
  const int iters = 20000;
  hipFuncSetAttribute((const void*)probe<MODE, ROWB>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, ROWB><<<256, nwaves * 64, 65536>>>(100, d);
  hipEventRecord(e0);
  probe<MODE, ROWB><<<256, nwaves * 64, 65536>>>(iters, d);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double bytes_per_iter_wave = 2048.0;   // 4 x 512 B or 2 x 1024 B
  // shader clocks unknown exactly: report ns per wave-iteration and B/ns per CU
  printf("%-28s waves/CU %d: %.3f ms, %.2f ns per iteration per wave, %.1f B/ns/CU (cycle ctr %llu per iter*100)\n", name, nwaves, ms,
         ms * 1e6 / iters, bytes_per_iter_wave * nwaves * iters / (ms * 1e6), h[0] * 100 / iters);
  hipFree(d);
}

int main() {
  for (int w : {4, 8}) {
    if (w == 4) { run<0, 512>("tr_b16 (512-B rows)", 4); run<0, 256>("tr_b16 (256-B rows)", 4); run<1, 512>("b64 (512-B rows)", 4); run<2, 512>("b128 linear", 4); }
    else { run<0, 512>("tr_b16 (512-B rows)", 8); run<0, 256>("tr_b16 (256-B rows)", 8); run<1, 512>("b64 (512-B rows)", 8); run<2, 512>("b128 linear", 8); }
  }
  return 0;
}
