// gemm.hip -- MFMA GEMMs for the dense layers of the DALL-E step (SURVEY.md §2.2 K3,K5,K6,K7).
//
// NT kernel:  C[M,N] = A[M,K] . Bt[N,K]^T, bf16 operands (both K-contiguous), fp32 accumulate on
//   v_mfma_f32_32x32x16_bf16.  128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles.
//   LDS: two stages x (A 16 KiB + B 16 KiB) = 64 KiB -> 2 blocks / CU.  Rows are 128 B (8 x 16-B chunks);
//   chunk index is XOR-swizzled with (row>>1)&7 so that the ds_read_b128 fragment reads of a 16-lane group
//   hit 16 distinct 16-B slots of the 256-B bank row (conflict-free).  Staging is either
//     GLDS=1: global_load_lds_dwordx4 (LDS-DMA; destination lane-linear, swizzle applied on the SOURCE
//             address -- cdna_hip_programming.md §5.4 rule 21), or
//     GLDS=0: global_load_dwordx4 -> registers -> ds_write_b128 (issue before compute, write after).
//   The MFMA operand roles are swapped (weights as the A-operand) so each lane ends up with ONE output row
//   and 4 consecutive columns per accumulator quad -> 8-byte bf16 / 16-byte fp32 stores.
//   blockIdx -> tile: bijective XCD remap (block b runs on XCD b%8) then GROUP_M-grouped ordering so the
//   8 A-tiles x all-B-tiles of a group stay in that XCD's 4 MiB L2.
//   Split-K (gridDim.y) writes fp32 partial slabs reduced deterministically by reduce_slabs_kernel.
//
// TN kernel (weight gradients): dW[I,J] = sum_m X[m,I] dY[m,J].  The contraction index is the ROW of both
//   operands, so fragments are fetched with ds_read_b64_tr_b16 (hardware 4x16 transpose read) from
//   natural-layout LDS tiles (pitch 320 B => conflict-free for the 2x32-lane service groups).
#include "common.h"
#include <string.h>

static int g_opt_glds = 1;
static int g_opt_tn_trread = 1;
extern "C" int dmi_get_option(const char* name) {
  if (!strcmp(name, "glds")) return g_opt_glds;
  if (!strcmp(name, "tn_trread")) return g_opt_tn_trread;
  return -1;
}
extern "C" int dmi_set_option(const char* name, int value) {
  if (!strcmp(name, "glds")) { g_opt_glds = value; return 0; }
  if (!strcmp(name, "tn_trread")) { g_opt_tn_trread = value; return 0; }
  return -1;
}

#define BM 128
#define BN 128
#define BK 64
#define GROUP_M 8

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const bf16_t* bias;
  const bf16_t* residual;
  const bf16_t* relu_src;
  int M, N, K, lda, ldb, ldc;
  int tiles_m, tiles_n;
  int k_per_split;     // multiple of BK
  int64_t slab_stride;  // elements between split-K slabs of C (fp32)
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ void tile_of_block(int L, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int per_group = GROUP_M * tiles_n;
  const int gid = L / per_group, first_m = gid * GROUP_M;
  const int gsz = (tiles_m - first_m < GROUP_M) ? tiles_m - first_m : GROUP_M;
  const int in_g = L - gid * per_group;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}

// byte offset inside a [128][64] bf16 LDS tile of the 16-B chunk `ch` (0..7) of row `row`
__device__ __forceinline__ int lds_chunk_off(int row, int ch) { return (row * 8 + (ch ^ ((row >> 1) & 7))) * 16; }

template <bool GLDS>
struct Stager {
  u32x4 r[4];
  // tile rows [row0, row0+128) of a [rows_total, ld] bf16 matrix, k range [k0, k0+64)
  __device__ __forceinline__ void issue(const bf16_t* __restrict__ base, int ld, int row0, int rows_total, int k0,
                                        char* lds_tile, int tid) {
    const int chp = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + 32 * i;
      int gr = row0 + row;
      gr = gr < rows_total ? gr : rows_total - 1;
      const int src_ch = chp ^ ((row >> 1) & 7);
      const bf16_t* gp = base + (int64_t)gr * ld + k0 + 8 * src_ch;
      if constexpr (GLDS) {
        // wave-uniform LDS base; each lane lands at base + lane*16  (chunk c = tid + 256 i)
        char* lbase = lds_tile + ((tid & ~63) + 256 * i) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)lbase, 16, 0, 0);
      } else {
        r[i] = *(const u32x4*)gp;
      }
    }
  }
  __device__ __forceinline__ void commit(char* lds_tile, int tid) {
    if constexpr (!GLDS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *(u32x4*)(lds_tile + (tid + 256 * i) * 16) = r[i];
    }
  }
};

template <int FLAGS, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 16K | B 16K]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int r = lane & 31, h = lane >> 5;

  int tm, tn;
  tile_of_block(xcd_remap(blockIdx.x, gridDim.x), a.tiles_m, a.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int kb = blockIdx.y * a.k_per_split;
  const int ke = (kb + a.k_per_split < a.K) ? kb + a.k_per_split : a.K;
  const int nt = (ke - kb) / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  Stager<GLDS> sa, sb;
  sa.issue(a.A, a.lda, m0, a.M, kb, smem, tid);
  sb.issue(a.B, a.ldb, n0, a.N, kb, smem + 16384, tid);
  sa.commit(smem, tid);
  sb.commit(smem + 16384, tid);
  if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * 32768;
    char* nxt = smem + ((t + 1) & 1) * 32768;
    const bool more = (t + 1 < nt);
    if (more) {
      sa.issue(a.A, a.lda, m0, a.M, kb + (t + 1) * BK, nxt, tid);
      sb.issue(a.B, a.ldb, n0, a.N, kb + (t + 1) * BK, nxt + 16384, tid);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 fa[2], fb[2];
      const int ch = kk * 2 + h;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *(const bf16x8*)(cur + lds_chunk_off(wm * 64 + i * 32 + r, ch));
        fb[i] = *(const bf16x8*)(cur + 16384 + lds_chunk_off(wn * 64 + i * 32 + r, ch));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);  // D[n][m]
    }
    if (more) {
      sa.commit(nxt, tid);
      sb.commit(nxt + 16384, tid);
      if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }

  // epilogue: lane owns row m = ..+r ; acc quad q holds columns n = ..+8q+4h+{0..3}
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + r;
    if (m >= a.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * h;
        if (n >= a.N) continue;
        float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        const int64_t off = (int64_t)m * a.ldc + n;
        if constexpr (FLAGS & DMI_GEMM_BIAS) {
          const u32x2 bb = *(const u32x2*)(a.bias + n);
          v[0] += __uint_as_float(bb[0] << 16);
          v[1] += __uint_as_float(bb[0] & 0xffff0000u);
          v[2] += __uint_as_float(bb[1] << 16);
          v[3] += __uint_as_float(bb[1] & 0xffff0000u);
        }
        if constexpr (FLAGS & DMI_GEMM_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (FLAGS & DMI_GEMM_RESIDUAL) {
          const u32x2 rr = *(const u32x2*)(a.residual + off);
          v[0] += __uint_as_float(rr[0] << 16);
          v[1] += __uint_as_float(rr[0] & 0xffff0000u);
          v[2] += __uint_as_float(rr[1] << 16);
          v[3] += __uint_as_float(rr[1] & 0xffff0000u);
        }
        if constexpr (FLAGS & DMI_GEMM_RELU_MASK) {
          const u32x2 hh = *(const u32x2*)(a.relu_src + off);
          // relu'(x) = (x > 0): post-relu activations are >= 0, so "> 0" <=> nonzero and sign clear
          v[0] = (__uint_as_float(hh[0] << 16) > 0.f) ? v[0] : 0.f;
          v[1] = (__uint_as_float(hh[0] & 0xffff0000u) > 0.f) ? v[1] : 0.f;
          v[2] = (__uint_as_float(hh[1] << 16) > 0.f) ? v[2] : 0.f;
          v[3] = (__uint_as_float(hh[1] & 0xffff0000u) > 0.f) ? v[3] : 0.f;
        }
        if constexpr (FLAGS & DMI_GEMM_OUT_F32) {
          float* cp = (float*)a.C + (int64_t)blockIdx.y * a.slab_stride + off;
          *(f32x4*)cp = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          bf16_t* cp = (bf16_t*)a.C + off;
          *(u32x2*)cp = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        }
      }
    }
  }
}

// out[i] = sum_s slabs[s*stride + i]   (float4 lanes, deterministic order)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                           int nsplit, int64_t n4, int64_t stride4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 acc = ((const f32x4*)slabs)[i];
    for (int s = 1; s < nsplit; ++s) {
      const f32x4 v = ((const f32x4*)slabs)[s * stride4 + i];
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    ((f32x4*)out)[i] = acc;
  }
}

template <int FLAGS>
static int launch_nt(const GemmArgs& a, int nsplit, hipStream_t st) {
  const dim3 grid(a.tiles_m * a.tiles_n, nsplit), blk(256);
  const size_t shm = 65536;
  if (g_opt_glds) {
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<FLAGS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr_done = true; }
    gemm_nt_kernel<FLAGS, true><<<grid, blk, shm, st>>>(a);
  } else {
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<FLAGS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr_done = true; }
    gemm_nt_kernel<FLAGS, false><<<grid, blk, shm, st>>>(a);
  }
  DMI_CHECK_LAUNCH("gemm_nt");
  return DMI_OK;
}

static int check_nt(const void* A, int lda, const void* B, int ldb, const void* C, int ldc, int M, int N, int K) {
  DMI_REQUIRE(A && B && C, "gemm_nt: null pointer");
  DMI_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0 && N % 8 == 0, "gemm_nt: need K%%64==0 and N%%8==0 (M=%d N=%d K=%d)", M, N, K);
  DMI_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && lda >= K && ldb >= K && ldc >= N, "gemm_nt: bad leading dimensions");
  DMI_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0 && ((uintptr_t)C & 7) == 0, "gemm_nt: operands must be 16-byte aligned");
  return DMI_OK;
}

extern "C" int dmi_gemm_nt(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, void* C, int ldc, int M, int N,
                           int K, int flags, const uint16_t* bias, const uint16_t* residual, const uint16_t* relu_src,
                           void* stream) {
  int rc = check_nt(A, lda, Bt, ldb, C, ldc, M, N, K);
  if (rc) return rc;
  DMI_REQUIRE(!(flags & DMI_GEMM_BIAS) || bias, "gemm_nt: bias flag without pointer");
  DMI_REQUIRE(!(flags & DMI_GEMM_RESIDUAL) || residual, "gemm_nt: residual flag without pointer");
  DMI_REQUIRE(!(flags & DMI_GEMM_RELU_MASK) || relu_src, "gemm_nt: relu-mask flag without pointer");
  GemmArgs a;
  a.A = A; a.B = Bt; a.C = C; a.bias = bias; a.residual = residual; a.relu_src = relu_src;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
  a.k_per_split = K; a.slab_stride = 0;
  hipStream_t st = (hipStream_t)stream;
  switch (flags) {
    case 0: return launch_nt<0>(a, 1, st);
    case DMI_GEMM_BIAS: return launch_nt<DMI_GEMM_BIAS>(a, 1, st);
    case DMI_GEMM_BIAS | DMI_GEMM_RELU: return launch_nt<DMI_GEMM_BIAS | DMI_GEMM_RELU>(a, 1, st);
    case DMI_GEMM_BIAS | DMI_GEMM_RESIDUAL: return launch_nt<DMI_GEMM_BIAS | DMI_GEMM_RESIDUAL>(a, 1, st);
    case DMI_GEMM_RELU_MASK: return launch_nt<DMI_GEMM_RELU_MASK>(a, 1, st);
    case DMI_GEMM_OUT_F32: return launch_nt<DMI_GEMM_OUT_F32>(a, 1, st);
    default:
      dmi_set_error("gemm_nt: unsupported flag combination %d", flags);
      return DMI_ERR_UNSUPPORTED;
  }
}

// =====================================================================================
// TN weight-gradient GEMM
// =====================================================================================
#define TN_BKM 64      // rows of m per step
#define TN_PITCH 160   // elements per LDS row (128 + 32 pad) = 320 B

static int tn_splits(int M, int I, int J) {
  const int tiles = ((I + 127) / 128) * ((J + 127) / 128);
  int s = (768 + tiles - 1) / tiles;           // aim for >= 3 blocks per CU
  const int max_s = (M + 4 * TN_BKM - 1) / (4 * TN_BKM);  // at least 4 k-steps per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s;
}
static int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

extern "C" int64_t dmi_gemm_tn_workspace_bytes(int M, int I, int J) {
  const int s = tn_splits(M, I, J);
  const int64_t slabs = (s > 1) ? (int64_t)s * I * J * 4 : 0;
  const int64_t Mp = round_up64(M, 64);
  const int64_t tr = ((int64_t)I * Mp + (int64_t)J * Mp) * 2;  // fallback: transposed operand copies
  return round_up64(slabs, 256) + round_up64(tr, 256) + 256;
}

struct TnArgs {
  const bf16_t* X;
  const bf16_t* Y;
  float* C;
  int M, I, J, ldx, ldy;
  int tiles_i, tiles_j;
  int m_per_split;  // multiple of TN_BKM
  int64_t slab_stride;
};

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t sx[TN_BKM * TN_PITCH];
  __shared__ __attribute__((aligned(16))) bf16_t sy[TN_BKM * TN_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wi = wid >> 1, wj = wid & 1;
  const int h = lane >> 5, g4 = lane >> 4, l16 = lane & 15;

  int ti, tj;
  tile_of_block(xcd_remap(blockIdx.x, gridDim.x), a.tiles_i, a.tiles_j, ti, tj);
  const int i0 = ti * 128, j0 = tj * 128;
  const int mb = blockIdx.y * a.m_per_split;
  const int me = (mb + a.m_per_split < a.M) ? mb + a.m_per_split : a.M;
  const int nt = (me - mb + TN_BKM - 1) / TN_BKM;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  u32x4 rx[4], ry[4];
  const int lrow = tid >> 4, lch = tid & 15;  // 16 chunks (of 8 cols) per 128-col row; rows lrow + 16 i
  auto load_tiles = [&](int m_base) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m_base + lrow + 16 * i;
      const bool ok = m < me;
      const int ci = i0 + 8 * lch, cj = j0 + 8 * lch;
      rx[i] = (ok && ci < a.I) ? *(const u32x4*)(a.X + (int64_t)m * a.ldx + ci) : u32x4{0, 0, 0, 0};
      ry[i] = (ok && cj < a.J) ? *(const u32x4*)(a.Y + (int64_t)m * a.ldy + cj) : u32x4{0, 0, 0, 0};
    }
  };
  load_tiles(mb);
  for (int t = 0; t < nt; ++t) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(u32x4*)(sx + (lrow + 16 * i) * TN_PITCH + 8 * lch) = rx[i];
      *(u32x4*)(sy + (lrow + 16 * i) * TN_PITCH + 8 * lch) = ry[i];
    }
    __syncthreads();
    if (t + 1 < nt) load_tiles(mb + (t + 1) * TN_BKM);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 fx[2], fy[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // 16-lane group g4 fetches the 4(k) x 16(col) block: rows 16kk+8h+4q.., cols tile + 16*(g4&1)..
        const int colx = wi * 64 + i * 32 + 16 * (g4 & 1) + 4 * (l16 & 3);
        const int coly = wj * 64 + i * 32 + 16 * (g4 & 1) + 4 * (l16 & 3);
        const int krow = 16 * kk + 8 * h + (l16 >> 2);
        const bf16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sx + krow * TN_PITCH + colx));
        const bf16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sx + (krow + 4) * TN_PITCH + colx));
        const bf16x4 y0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sy + krow * TN_PITCH + coly));
        const bf16x4 y1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sy + (krow + 4) * TN_PITCH + coly));
        fx[i] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
        fy[i] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx[i], fy[j], acc[i][j], 0, 0, 0);  // D[i][j]
    }
  }
  // store: reg e -> row i = (e&3) + 8*(e>>2) + 4h ; col j = lane&31
  float* C = a.C + (int64_t)blockIdx.y * a.slab_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = j0 + wj * 64 + j * 32 + (lane & 31);
      if (col >= a.J) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = i0 + wi * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < a.I) C[(int64_t)row * a.J + col] = acc[i][j][e];
      }
    }
}

int dmi_transpose_padded(const uint16_t* in, uint16_t* out, int nb, int nh, int Rv, int Rp, int C,
                         int64_t in_b_stride, int64_t in_h_stride, int64_t in_r_stride, void* stream);

extern "C" int dmi_gemm_tn(const uint16_t* X, int ldx, const uint16_t* dY, int ldy, float* dW, int M, int I, int J,
                           void* workspace, void* stream) {
  DMI_REQUIRE(X && dY && dW && workspace, "gemm_tn: null pointer");
  DMI_REQUIRE(M > 0 && I % 8 == 0 && J % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= I && ldy >= J,
              "gemm_tn: I, J, ldx, ldy must be multiples of 8 (M=%d I=%d J=%d)", M, I, J);
  DMI_REQUIRE((((uintptr_t)X | (uintptr_t)dY | (uintptr_t)dW | (uintptr_t)workspace) & 15) == 0, "gemm_tn: 16-byte alignment required");
  hipStream_t st = (hipStream_t)stream;
  const int nsplit = tn_splits(M, I, J);
  float* slabs = (float*)workspace;
  const int64_t slab_bytes = (nsplit > 1) ? round_up64((int64_t)nsplit * I * J * 4, 256) : 0;
  if (g_opt_tn_trread) {
    TnArgs a;
    a.X = X; a.Y = dY; a.M = M; a.I = I; a.J = J; a.ldx = ldx; a.ldy = ldy;
    a.tiles_i = (I + 127) / 128; a.tiles_j = (J + 127) / 128;
    a.m_per_split = (int)round_up64((M + nsplit - 1) / nsplit, TN_BKM);
    a.C = (nsplit > 1) ? slabs : dW;
    a.slab_stride = (nsplit > 1) ? (int64_t)I * J : 0;
    gemm_tn_kernel<<<dim3(a.tiles_i * a.tiles_j, nsplit), dim3(256), 0, st>>>(a);
    DMI_CHECK_LAUNCH("gemm_tn");
  } else {
    // fallback: explicit transposes + split-K NT GEMM:  dW[I,J] = Xt[I,Mp] . dYt[J,Mp]^T
    const int Mp = (int)round_up64(M, 64);
    bf16_t* Xt = (bf16_t*)((char*)workspace + slab_bytes);
    bf16_t* Yt = Xt + (int64_t)I * Mp;
    int rc = dmi_transpose_padded(X, Xt, 1, 1, M, Mp, I, 0, 0, ldx, stream);  // pad rows [M, Mp) = 0
    if (rc) return rc;
    rc = dmi_transpose_padded(dY, Yt, 1, 1, M, Mp, J, 0, 0, ldy, stream);
    if (rc) return rc;
    GemmArgs g;
    g.A = Xt; g.B = Yt; g.bias = nullptr; g.residual = nullptr; g.relu_src = nullptr;
    g.M = I; g.N = J; g.K = Mp; g.lda = Mp; g.ldb = Mp; g.ldc = J;
    g.tiles_m = (I + BM - 1) / BM; g.tiles_n = (J + BN - 1) / BN;
    g.k_per_split = (int)round_up64((Mp + nsplit - 1) / nsplit, BK);
    g.C = (nsplit > 1) ? (void*)slabs : (void*)dW;
    g.slab_stride = (nsplit > 1) ? (int64_t)I * J : 0;
    const int ns = (Mp + g.k_per_split - 1) / g.k_per_split;
    rc = launch_nt<DMI_GEMM_OUT_F32>(g, ns, st);
    if (rc) return rc;
    if (ns != nsplit && nsplit > 1) {
      // fewer effective splits than planned: zero the unused slabs so the reduce stays exact
      hipError_t e = hipMemsetAsync(slabs + (int64_t)ns * I * J, 0, (int64_t)(nsplit - ns) * I * J * 4, st);
      DMI_REQUIRE(e == hipSuccess, "gemm_tn: memset failed");
    }
  }
  if (nsplit > 1) {
    const int64_t n4 = (int64_t)I * J / 4;
    int64_t blocks = cdiv64(n4, 256);
    if (blocks > 2048) blocks = 2048;
    reduce_slabs_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(slabs, dW, nsplit, n4, n4);
    DMI_CHECK_LAUNCH("gemm_tn_reduce");
  }
  return DMI_OK;
}
