// HBM write-rate probe for GEMM epilogue patterns (gfx950): is a K = 512 GEMM that writes 1.65 TB/s of bf16 output bound by the
// write pattern?  Every block stores one TM x TN bf16 tile of a [M][ldc] matrix as 16-B lane stores forming full row segments
// (what the NT kernels' LDS-staged epilogue does), tiles visited in the GEMM's order (groups of 8 row tiles sweep the columns).
// Compared with a plain linear stream of the same bytes.   hipcc --offload-arch=gfx950 -O3 tools/wrbench.hip -o tools/_build/wrbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int TM, int TN>
__global__ __launch_bounds__(256) void tile_store(uint16_t* C, int64_t ldc, int tiles_m, int tiles_n, int group_m, int delay) {
  // tile order: groups of group_m row tiles, column-major inside a group (consecutive blocks share a column tile)
  const int t = blockIdx.x;
  const int per_group = group_m * tiles_n;
  const int g = t / per_group, r = t - g * per_group;
  const int gm = (tiles_m - g * group_m < group_m) ? tiles_m - g * group_m : group_m;
  const int tm = g * group_m + r % gm, tn = r / gm;
  if (tm >= tiles_m || tn >= tiles_n) return;
  // emulate compute time before the epilogue
  if (delay > 0) __builtin_amdgcn_s_sleep(127);
  const int tid = threadIdx.x;
  constexpr int CPR = TN / 8;                // 16-B chunks per tile row
  constexpr int ROWS_PER_PASS = 256 / CPR;
  const u32x4 v = {(uint32_t)t, (uint32_t)tid, 0x3f803f80u, 0x3f803f80u};
#pragma unroll 4
  for (int r0 = 0; r0 < TM; r0 += ROWS_PER_PASS) {
    const int row = r0 + tid / CPR, ch = tid % CPR;
    *(u32x4*)(C + (int64_t)(tm * TM + row) * ldc + tn * TN + ch * 8) = v;
  }
}
__global__ __launch_bounds__(256) void linear_store(u32x4* C, int64_t n16) {
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) C[i] = v;
}

int main() {
  const int M = 40960, N = 50816;
  const int64_t ldc = 51200;   // padded pitch: 256- and 512-wide tiles overhang N
  uint16_t* C;
  hipMalloc(&C, (int64_t)M * ldc * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = (double)M * N * 2 / 1e9;
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %7.3f ms  %6.2f TB/s\n", name, ms / 5, gb / (ms / 5) );
  };
  time("linear 16-B stores, 4096 blocks", [&] { linear_store<<<4096, 256>>>((u32x4*)C, (int64_t)M * N * 2 / 16); });
  time("linear 16-B stores, 1024 blocks", [&] { linear_store<<<1024, 256>>>((u32x4*)C, (int64_t)M * N * 2 / 16); });
  for (int gm : {8, 1, 160}) {
    char nm[128];
    snprintf(nm, sizeof nm, "256x128 tiles (256-B row segments), group_m %d", gm);
    time(nm, [&] { tile_store<256, 128><<<160 * 397, 256>>>(C, ldc, 160, 397, gm, 0); });
    snprintf(nm, sizeof nm, "128x128 tiles (256-B row segments), group_m %d", gm);
    time(nm, [&] { tile_store<128, 128><<<320 * 397, 256>>>(C, ldc, 320, 397, gm, 0); });
    snprintf(nm, sizeof nm, "256x256 tiles (512-B row segments), group_m %d", gm);
    time(nm, [&] { tile_store<256, 256><<<160 * 199, 256>>>(C, ldc, 160, 199, gm, 0); });
  }
  time("128x512 tiles (1-KiB row segments), group_m 8", [&] { tile_store<128, 512><<<320 * 100, 256>>>(C, ldc, 320, 100, 8, 0); });
  hipFree(C);
  return 0;
}
