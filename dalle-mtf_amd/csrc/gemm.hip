// gemm.hip -- MFMA GEMMs for the dense layers of the DALL-E step (SURVEY.md §2.2 K3,K5,K6,K7) and the VAE convolutions.
//
// NT kernels:  C[M,N] = A[M,K] . Bt[N,K]^T, bf16 operands (both K-contiguous), fp32 accumulate on
//   v_mfma_f32_32x32x16_bf16.  Two tilings share one epilogue:
//     gemm_nt2_kernel  128x128x64 block tile, 4 waves (2x2) of 64x64, 2 LDS stages x 32 KiB -> 2 blocks / CU;
//     gemm_nt4_kernel  256x128x32 block tile, 4 waves of 128x64, 2 stages x 24 KiB (K <= 1024 and >= 3 residencies).
//   Staging is LDS-DMA (buffer_load_dwordx4 ... lds) through a per-block buffer descriptor: the destination is
//   lane-linear, so the XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is applied on the SOURCE
//   address (cdna_hip_programming.md §5.4 rule 21).  The MFMA operand roles are swapped (weights as the A-operand) so
//   each lane ends up with ONE output row.  blockIdx -> tile: bijective XCD remap (block b runs on XCD b%8) then
//   GROUP_M-grouped ordering so the 8 A-tiles x all-B-tiles of a group stay in that XCD's 4 MiB L2.
//   Split-K (gridDim.y) writes fp32 partial slabs reduced deterministically.
//   Epilogues (compile-time flags): bias, ReLU, residual, ReLU-mask, per-row scale, and the softmax-numerator epilogue
//   of the vocabulary projection (see gemm_nt_softmax below).
//
// TN kernel (weight gradients): dW[I,J] = sum_m X[m,I] dY[m,J].  The contraction index is the ROW of both
//   operands, so fragments are fetched with ds_read_b64_tr_b16 (hardware 4x16 transpose read) from
//   natural-layout LDS tiles filled by LDS-DMA (XOR-swizzled on the source side => conflict-free).
//
// The measured-slower generations of these kernels (register-staged v1, persistent v3, hand-scheduled 3-stage v5,
// grouped / stream-K weight gradients) live on the git branch `r01-kernel-variants`, not in the product library.
#include "common.h"
#include <string.h>
#include <type_traits>

static int g_opt_nt4 = 1;        // 256x128 NT tile: 0 never, 1 auto (>= 3 residencies and K <= 1024), 2 always (tests)
static int g_opt_nt8 = 1;        // 256x256 NT tile (8 waves): 0 never, 1 auto (K >= nt8_min_k, whole residencies), 2 always (tests)
int g_opt_reserve_cus = 0;       // CUs the persistent kernels (256x256 NT, attention) leave free, and which switch the one-tile-per-CU
                                 // full-row kernel off: a block of those kernels whose CU is held by a concurrent kernel (the RCCL channels
                                 // of the gradient exchange) starts late and stretches the whole launch -- measured with the token sort as
                                 // the intruder (layer 0's forward 687 us instead of 410 us).  The engine sets 16 when world_size > 1.
int g_opt_attn_fwd = 1;          // attention forward kernel: 0 round-2 form, 1 software-pipelined (round 6)
int g_opt_attn_bwd = 1;          // attention backward: bit 0 = whole-row epilogue stores through LDS (round 6)
int g_opt_attn_xcd = 8;          // attention block order: 0 plain grid; G >= 1: per-XCD ranges, groups of G (batch, head) pairs tile-major
static int g_opt_tn8 = 0;        // 256x256 weight-gradient tile (8 waves, still on 32x32x16 MFMAs): 0 never (default since the 128x128 kernel
                                 // moved to 16x16x32: 45 / 76 us vs 60 / 86 us on the 512x512 / 512x1536 gradients), 1 auto (few tiles), 2 always (tests)
static int g_opt_tn8_max_tiles = 16;   // auto mode: use the 256x256 weight-gradient kernel below this many tiles (A/B hook)
static int g_opt_nt8_min_k = 2048;   // auto mode of the 256x256 NT tile: minimum K.  2048 (was 4096): the 1.3B dimensions' K = 2048 products
                                     // run 354 -> 339 ms/step (profiles/r03_ab_nt8_min_k.log); K = 1024 measured equal, K = 512 slower
static int g_opt_skinny = 1;     // M <= 32 products (the decode step) on the weight-streaming kernel: 0 never, 1 auto
static int g_opt_tn_wide = 1;    // [r06] unsplit weight gradients with many column stripes (the head) as a gang stream-K on 128 x 256 tiles: 0 never, 1 auto
static int g_opt_tn_tail = 1;    // weight gradients: row-split the tiles of a ragged last residency (see gemm_tn_tail_kernel)
static int g_opt_cstream = 1;    // bf16 outputs stored write-through (sc1; + non-temporal from cstream_nt_min_mb MiB): 0 never, 1 auto (outputs >=
                                 // cstream_min_mb MiB: they cannot be re-read from L2 anyway, and as plain stores they evict the operands the concurrent
                                 // tiles share), 2 always sc1, 3 always sc1 nt, 4 auto with sc1 only (A/B)
static int g_opt_cstream_min_mb = 256;
static int g_opt_cstream_nt_min_mb = 1024;
static int g_opt_res16 = 1;      // register epilogue: the residual as 16-byte pieces through the row swap (1) or 8-byte pieces in the accumulator layout (0)
static int g_opt_ntr = 1;        // full-row 160x512 tiles for N = 512 products (gemm_ntr_kernel): 0 never, 1 auto, 2 whenever the shape allows
static int g_opt_nt8p = 1;       // persistent 256x256 NT kernel with the register epilogue (gemm_nt8p_kernel): 0 never, 1 auto (short K, >= 2
                                 // tiles per CU), 2 whenever the shape allows it (tests, tools/kbench.py)
static int g_opt_relu_bits = 1;   // FFN ReLU mask as one bit per element (dmi_gemm_nt_relu_bits / _mask_bits) where the persistent kernel runs: 0 never, 1 auto
static int g_opt_nt8p_max_k = 1024;   // auto mode: K above this keeps the one-tile-per-block kernels (the main loop then dominates a tile)
static int g_opt_nt4_lds = 49152;   // dynamic LDS requested by the 256x128 NT kernel: 49152 = what it uses (3 blocks / CU); 65536 / 98304
                                    // cap the residency at 2 / 1 blocks per CU (tools/phases.py: a block's phases without co-resident blocks)
unsigned long long* g_dbg_buf = nullptr;   // (also read by the ATTN_STAMP experiment build of attention.hip)
extern "C" int dmi_set_debug_buffer(void* p) { g_dbg_buf = (unsigned long long*)p; return 0; }
extern "C" int dmi_get_option(const char* name) {
  if (!strcmp(name, "nt4")) return g_opt_nt4;
  if (!strcmp(name, "nt8")) return g_opt_nt8;
  if (!strcmp(name, "tn_tail")) return g_opt_tn_tail;
  if (!strcmp(name, "tn_wide")) return g_opt_tn_wide;
  if (!strcmp(name, "nt4_lds")) return g_opt_nt4_lds;
  if (!strcmp(name, "nt8p")) return g_opt_nt8p;
  if (!strcmp(name, "ntr")) return g_opt_ntr;
  if (!strcmp(name, "cstream")) return g_opt_cstream;
  if (!strcmp(name, "cstream_min_mb")) return g_opt_cstream_min_mb;
  if (!strcmp(name, "cstream_nt_min_mb")) return g_opt_cstream_nt_min_mb;
  if (!strcmp(name, "nt8p_max_k")) return g_opt_nt8p_max_k;
  if (!strcmp(name, "relu_bits")) return g_opt_relu_bits;
  if (!strcmp(name, "res16")) return g_opt_res16;
  if (!strcmp(name, "skinny")) return g_opt_skinny;
  if (!strcmp(name, "nt8_min_k")) return g_opt_nt8_min_k;
  if (!strcmp(name, "tn8")) return g_opt_tn8;
  if (!strcmp(name, "tn8_max_tiles")) return g_opt_tn8_max_tiles;
  if (!strcmp(name, "attn_xcd")) return g_opt_attn_xcd;
  if (!strcmp(name, "attn_fwd")) return g_opt_attn_fwd;
  if (!strcmp(name, "attn_bwd")) return g_opt_attn_bwd;
  if (!strcmp(name, "reserve_cus")) return g_opt_reserve_cus;
  return -1;
}
extern "C" int dmi_set_option(const char* name, int value) {
  if (!strcmp(name, "nt4")) { g_opt_nt4 = value; return 0; }
  if (!strcmp(name, "nt8")) { g_opt_nt8 = value; return 0; }
  if (!strcmp(name, "tn_tail")) { g_opt_tn_tail = value; return 0; }
  if (!strcmp(name, "tn_wide")) { g_opt_tn_wide = value; return 0; }
  if (!strcmp(name, "nt4_lds")) { if (value < 49152 || value > 163840) return -1; g_opt_nt4_lds = value; return 0; }
  if (!strcmp(name, "nt8p")) { g_opt_nt8p = value; return 0; }
  if (!strcmp(name, "ntr")) { g_opt_ntr = value; return 0; }
  if (!strcmp(name, "cstream")) { g_opt_cstream = value; return 0; }
  if (!strcmp(name, "cstream_min_mb")) { g_opt_cstream_min_mb = value; return 0; }
  if (!strcmp(name, "cstream_nt_min_mb")) { g_opt_cstream_nt_min_mb = value; return 0; }
  if (!strcmp(name, "nt8p_max_k")) { g_opt_nt8p_max_k = value; return 0; }
  if (!strcmp(name, "relu_bits")) { g_opt_relu_bits = value; return 0; }
  if (!strcmp(name, "res16")) { g_opt_res16 = value; return 0; }
  if (!strcmp(name, "skinny")) { g_opt_skinny = value; return 0; }
  if (!strcmp(name, "nt8_min_k")) { g_opt_nt8_min_k = value; return 0; }
  if (!strcmp(name, "tn8")) { g_opt_tn8 = value; return 0; }
  if (!strcmp(name, "tn8_max_tiles")) { g_opt_tn8_max_tiles = value; return 0; }
  if (!strcmp(name, "attn_xcd")) { g_opt_attn_xcd = value; return 0; }
  if (!strcmp(name, "attn_fwd")) { g_opt_attn_fwd = value; return 0; }
  if (!strcmp(name, "attn_bwd")) { g_opt_attn_bwd = value; return 0; }
  if (!strcmp(name, "reserve_cus")) { if (value < 0) return -1; g_opt_reserve_cus = value; return 0; }
  return -1;
}

#define BM 128
#define BN 128
#define BK 64
#ifndef GROUP_M
#define GROUP_M 8
#endif

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const bf16_t* bias;
  const bf16_t* residual;
  const bf16_t* relu_src;
  unsigned short* relu_bits;   // GEMM_RELU_BITS: written; GEMM_MASK_BITS: read -- [N / 64][M][4] 16-bit words, see epilogue_regs
  const float* rowscale;   // DMI_GEMM_ROWSCALE: fp32 [M], C[m, :] *= rowscale[m]
  const float* rowshift;   // GEMM_SOFTMAX: nullable fp32 [M], C[m, n] = exp(acc + bias - rowshift[m])  (NULL: no shift)
  float* rowsum_part;      // GEMM_SOFTMAX: fp32 [ceil(N/64)][M] partial row sums of the fp32 exponentials
  int M, N, K, lda, ldb, ldc;
  int tiles_m, tiles_n;
  int k_per_split;     // multiple of BK
  int64_t slab_stride;  // elements between split-K slabs of C (fp32)
  unsigned long long* dbg;  // optional per-block phase timestamps (tools/phases.py); nullptr in production
  int pf;                   // bit 1: register epilogue fetches the residual as 16-byte pieces through the row swap (option res16)
  int cpol;                 // cache policy of the bf16 output stores: 0 plain, 1 sc1 (write-through, the line is not kept in the XCD's L2), 2 sc1 nt
  // fused LayerNorm of the output rows (dmi_gemm_nt_ln, full-row tiles only): Y = LN(C) * gamma + beta, row statistics
  const bf16_t* ln_gamma;
  const bf16_t* ln_beta;
  bf16_t* ln_y;
  float* ln_mean;
  float* ln_rstd;
  float ln_eps;
  int ln_ldy;
  // fused LayerNorm BACKWARD of the product (dmi_gemm_nt_lnbwd, full-row tiles only): the product is dy; ln_x = the LayerNorm's input,
  // ln_gamma / ln_mean / ln_rstd as above, residual = the gradient arriving over the residual connection (nullable), ln_y = dx,
  // ln_part = [2 * blocks][2 * N] fp32 partial gain | bias gradients (one row per 80-row half tile)
  const bf16_t* ln_x;
  float* ln_part;
  // ... and, chained behind it in the same launch, the product that consumes dx (dmi_gemm_nt_lnbwd with B2): C2[M, 512] = dx . B2^T --
  // a block owns whole rows of dx, i.e. the whole contraction range of its rows (the out-projection's input gradient d_o = dx . Wo^T)
  const bf16_t* B2;
  bf16_t* C2;
  int ldb2;
};
#define GEMM_SOFTMAX 64   // internal epilogue flag of dmi_gemm_nt_softmax (not part of the public flag set)
#define GEMM_RELU_BITS 128   // internal: with DMI_GEMM_RELU, also emit one bit per output (> 0) -- dmi_gemm_nt_relu_bits
#define GEMM_MASK_BITS 256   // internal: C *= bit, the bits written by GEMM_RELU_BITS -- dmi_gemm_nt_mask_bits

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
// 64-B LDS rows (32-wide k-steps, 4 chunks of 16 B): see the 256x128 kernel below
__device__ __forceinline__ int lds4_swz(int row) { return (-(row >> 2)) & 3; }
__device__ __forceinline__ int lds4_off(int row, int ch) { return row * 64 + ((ch ^ lds4_swz(row)) << 4); }

// =====================================================================================
// NT kernel v2: same tiling/swizzle as gemm_nt_kernel, with
//  * LDS-DMA through a per-block buffer descriptor (buffer_load_dwordx4 ... lds): the 8 per-lane byte
//    offsets are loop-invariant VGPRs, the K advance is one SGPR add -> no address VALU in the K-loop
//    (v1 issued 3.6 VALU per MFMA, rocprofv3 PMC profiles/r01);
//  * swizzled fragment offsets hoisted (4 VGPRs per operand) + immediates, stages unrolled x2;
//  * bf16 epilogue staged through LDS (fp32, 32 rows at a time per wave) so every lane stores 16 B of one
//    output row: full 128-B row segments instead of scattered 8-B pieces (the vocabulary projection was
//    62 % store-wait with the direct epilogue).
// =====================================================================================
// bf16 epilogue shared by the v2 / v4 / v5 kernels: wave-private fp32 staging (32 rows x 64 cols, pitch 272 B) so that
// every lane stores 16 B of one output row (full 128-B segments), with fused bias / ReLU / residual / ReLU-mask.
// A per-block timeline (tools/phases.py) showed the epilogue taking 39 % of a block's life on the K = 512 shapes: the
// bias vector was re-loaded (a dependent ~600-cycle L2 round trip) in each of the 16 store iterations.  Here the bias is
// loaded once per tile and the residual / ReLU-source rows of a 32-row group are requested before that group's LDS
// round trip, so no store iteration waits on a global load it has just issued.
// (Measured and dropped: replacing the bf16 relu_src read of the FFN-2 input gradient by 1 sign bit per element -- 8 ballots in
// the FFN-1 epilogue, 8 broadcast 64-bit loads + shifts in the mask epilogue -- saves 158 MB of reads per layer but ran the two
// kernels 125 / 149 us instead of 110 / 134 us: the epilogue is issue-bound, not bandwidth-bound.)
// accumulators -> wave-private fp32 staging block of 32 rows x 64 columns (pitch 68 floats), for the two MFMA shapes:
//   32x32x16: acc[MI][2] (f32x16): lane (r = lane & 31, h = lane >> 5) holds row r, columns j*32 + 8q + 4h + {0..3}
//   16x16x32: acc[2 MI][4] (f32x4): lane (c = lane & 15, g = lane >> 4) holds row ii*16 + c, columns j*16 + 4g + {0..3}
// 16-B store of 8 bf16 outputs at element offset `off` of C.  cpol 1: through a buffer descriptor with the sc1 bit -- written
// through to the memory side, the line is not kept in the XCD's L2; cpol 2 (what the auto mode picks): sc1 + nt, non-temporal as
// well -- the stream does not displace the re-read operands from the Infinity Cache either: vocabulary projection 2505 -> 2125 us
// (1000 TFLOP/s) on top of the 2500 <- 2535 of sc1 alone, step -0.27 ms (profiles/r04ac_kbench_nt_store.log, r04ad_kbench_nt.log).  The 4 GB of softmax numerators the vocabulary projection
// writes would otherwise pass through the 4-MiB L2s as dirty lines and evict the A / B panels that the XCD's 32 concurrent tiles
// share (round 3 measured 2.5 GB fetched per launch against 94 MB of operands).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t c_rsrc(const GemmArgs& a) {
  return __builtin_amdgcn_make_buffer_rsrc(a.C, 0, -1 /* 2^32 - 1 bytes: the host checks the matrix is smaller */, 0x00020000);
}
__device__ __forceinline__ void store_c16(const GemmArgs& a, __amdgpu_buffer_rsrc_t rc, int64_t off, u32x4 v) {
  if (a.cpol == 1) __builtin_amdgcn_raw_buffer_store_b128(v, rc, (int)(unsigned)(off * 2), 0, 16);
  else if (a.cpol == 2) __builtin_amdgcn_raw_buffer_store_b128(v, rc, (int)(unsigned)(off * 2), 0, 18);   // sc1 nt
  else *(u32x4*)((bf16_t*)a.C + off) = v;
}

template <int MI>
__device__ __forceinline__ void stage_rows32(const f32x16 (&acc)[MI][2], int i, float* stg, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(f32x4*)(stg + r * 68 + j * 32 + 8 * q + 4 * h) =
          f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
}
template <int MI>
__device__ __forceinline__ void stage_rows32(const f32x4 (&acc)[2 * MI][4], int i, float* stg, int lane) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(stg + (ii * 16 + c) * 68 + j * 16 + 4 * g) = acc[2 * i + ii][j];
}

// ---- register epilogue (round 4) ----------------------------------------------------------------------------------
// The LDS-staged epilogue above needs 8.5 KB of LDS per wave, i.e. the stage buffers: nothing can be prefetched into them while
// it runs, and its write -> wait -> read -> wait -> VALU -> store chains take 13 k cycles of a lone block's life with the
// softmax-numerator epilogue (tools/phases.py, profiles/r04a_phases_*.log).  This form keeps everything in registers: the element-wise part (bias, ReLU, row scale,
// exp2) runs on the accumulators where they are -- lane (c = lane & 15, g = lane >> 4) holds C[row 16 t + c][16 j + 4 g + {0..3}]
// of the wave tile -- the results are packed to bf16 (P[j][0..1]: the 4-column chunk 4 j + g of the 16 chunks of a row) and ONE
// round of row swaps, v_permlane16_swap (odd 16-lane rows of the first operand <-> even rows of the second) on (P[0], P[1]) and
// (P[2], P[3]), leaves lane group g with chunks {2p, 2p+1} and {8+2p, 8+2p+1}, p = 0, 2, 1, 3 for g = 0..3: two 16-B pieces of
// its row, and the four lane groups of a row write 64 contiguous bytes per store instruction.  (The full 4 x 4 transpose --
// a v_permlane32_swap round first, 32 contiguous bytes per lane -- stores alternating 16-B pieces per instruction and measured
// 10 % SLOWER than the LDS form: the co-resident blocks' DMA loads share the memory path with those stores, r04b_kbench_k512.log.)
// No LDS, no waits, the 2 MI row tiles of a wave are independent chains.  The ReLU mask of the FFN-2 input gradient is applied
// to the ROUNDED values after the transpose (a select commutes with rounding: bit-identical) from two 16-B loads of the
// lane's own 16 columns; a residual is added in fp32 BEFORE rounding from 8-B pieces in the accumulator layout.  bf16 outputs are
// bit-identical to the LDS form; the softmax row-sum partials add the same 64 fp32 exponentials in a different (fixed) order.
// Where it is used: the persistent 256x256 kernel (gemm_nt8p_kernel), whose stage buffers must stay free for the next tile's
// prefetch.  In the one-tile-per-block kernels it measured equal (vocabulary projection) to 6-18 % SLOWER (QKV, FFN-1, FFN-2
// input gradient: profiles/r04c_kbench_k512.log, step 16.35 -> 16.50 ms): a lone block's plain epilogue takes 4.8 k cycles in
// both forms -- it is bound by the 64 KB of stores (~14 B / clk / CU), not by the LDS round trips -- and with three blocks per CU
// a shorter epilogue only moves the wait into the DMA-bound k-steps (2008 -> 2415 cycles per k-step).  Those kernels keep the
// LDS form, whose stores are whole 128-B lines.
template <int FLAGS>
constexpr bool epi_regs_ok = !(FLAGS & DMI_GEMM_OUT_F32);

typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {   // a: rows 1, 3 <-> b: rows 0, 2 (16-lane rows)
  const u32x2v r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
// keep the bf16 halves of `v` whose counterpart in `src` (packed bf16) is > 0  (DMI_GEMM_RELU_MASK)
__device__ __forceinline__ unsigned relu_mask2(unsigned v, unsigned src) {
  const unsigned lo = (__uint_as_float(src << 16) > 0.f) ? 0x0000ffffu : 0u;
  const unsigned hi = (__uint_as_float(src & 0xffff0000u) > 0.f) ? 0xffff0000u : 0u;
  return v & (lo | hi);
}

// ---- the ReLU mask as one BIT per element (round 5) ----------------------------------------------------------------
// The FFN-2 input gradient dh = (dx . W2^T) * (h > 0) read the whole bf16 h (168 MB per launch at dalle_example, as much as it
// writes) only for its sign: 129 us against 87 us for the same product without the mask (FFN-1 forward), and the difference is
// that second stream (168 MB / 43 us = 3.9 TB/s).  The forward product now emits the bits from its register epilogue and the
// gradient product reads 2 bytes per 16 outputs.  Layout = the register epilogue's own: after the row swap lane (c, g) of a wave
// holds columns nst + {0..7} ("lo") and nst + 32 + {0..7} ("hi") of row m of the wave's 64-column group; its 16-bit word
// (bit k: lo element k, bit 8 + k: hi element k) lives at bits[((col_group * M) + m) * 4 + g] -- a store / load instruction of a
// wave moves 16 rows x 8 bytes = 128 contiguous bytes.  Producer and consumer are both epilogue_regs, so the layout is private
// to it (N % 64 == 0).  (Round 2 built the ballot form in the LDS epilogue: issue-bound there, slower.)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// the 8 outputs of a 16-B piece (packed bf16, all >= +0 after the ReLU) -> 8 bits, bit k = element k > 0
__device__ __forceinline__ unsigned relu_bits8(const u32x4 w) {
  unsigned x = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    // h in [0x0001, 0x7fff] <=> bit 15 of h + 0x7fff (a packed 16-bit add: no carry between the halves)
    const unsigned wd = w[d];     // (through a scalar: __builtin_bit_cast applied to the subscript itself reads element 0 for every d)
    const u16x2 t = __builtin_bit_cast(u16x2, wd) + u16x2{0x7fff, 0x7fff};
    const unsigned m = __builtin_bit_cast(unsigned, t) & 0x80008000u;
    x |= m >> (15 - 2 * d);      // low half -> bit 2 d, high half -> bit 16 + 2 d
  }
  return (x & 0x55u) | ((x >> 15) & 0xaau);
}
// keep the bf16 halves of dword d of a piece whose bits (2 d, 2 d + 1) of `mk` (shifted so that the piece starts at bit 0) are set
template <int B0>
__device__ __forceinline__ unsigned keep_bits2(unsigned v, unsigned mk) {
  const unsigned lo = (unsigned)__builtin_amdgcn_sbfe((int)mk, B0, 1), hi = (unsigned)__builtin_amdgcn_sbfe((int)mk, B0 + 1, 1);
  return v & ((lo & 0x0000ffffu) | (hi & 0xffff0000u));
}

template <int FLAGS, int NT>     // NT = 16-row tiles of the wave tile (its width is 64 columns)
__device__ __forceinline__ void epilogue_regs(const GemmArgs& a, const f32x4 (&acc)[NT][4], int lane, int mrow0, int ncol0) {
  const int c16 = lane & 15, g16 = lane >> 4;
  constexpr float LOG2E = 1.4426950408889634f;
  float bias[4][4];
  bool jok[4];   // column chunk 16 j + 4 g of the wave tile lies inside the matrix (N % 8 == 0: a chunk is whole or absent)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = ncol0 + 16 * j + 4 * g16;
    jok[j] = n < a.N;
    if constexpr (FLAGS & DMI_GEMM_BIAS) {
      u32x2 raw = {0u, 0u};
      if (jok[j]) raw = *(const u32x2*)(a.bias + n);
      bias[j][0] = __uint_as_float(raw[0] << 16);
      bias[j][1] = __uint_as_float(raw[0] & 0xffff0000u);
      bias[j][2] = __uint_as_float(raw[1] << 16);
      bias[j][3] = __uint_as_float(raw[1] & 0xffff0000u);
      if constexpr (FLAGS & GEMM_SOFTMAX) {   // the bias rides in the exponent's fma
#pragma unroll
        for (int e = 0; e < 4; ++e) bias[j][e] *= LOG2E;
      }
    }
  }
  const bool shifted = (FLAGS & GEMM_SOFTMAX) && a.rowshift != nullptr;   // wave-uniform
  const __amdgpu_buffer_rsrc_t rc = c_rsrc(a);
  // this lane's two 16-B pieces of its row: columns nst + {0..7} and nst + 32 + {0..7}; the four lane groups of a row cover
  // pieces 0, 2, 1, 3 (+4): each store instruction writes 64 contiguous bytes per row
  const int nst = ncol0 + 8 * (((g16 & 1) << 1) | (g16 >> 1));
  const bool ok0 = nst < a.N, ok1 = nst + 32 < a.N;
  unsigned mbits[NT];
  if constexpr (FLAGS & GEMM_MASK_BITS) {   // all row tiles' words up front: NT independent 2-byte loads
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int m = mrow0 + 16 * t + c16;
      mbits[t] = (m < a.M && ok0) ? (unsigned)a.relu_bits[((int64_t)(ncol0 >> 6) * a.M + m) * 4 + g16] : 0u;
    }
  }
  u32x4 nsrc0 = {0u, 0u, 0u, 0u}, nsrc1 = {0u, 0u, 0u, 0u};
  if constexpr (FLAGS & DMI_GEMM_RELU_MASK) {
    const int m = mrow0 + c16;
    if (m < a.M && ok0) nsrc0 = *(const u32x4*)(a.relu_src + (int64_t)m * a.ldc + nst);
    if (m < a.M && ok1) nsrc1 = *(const u32x4*)(a.relu_src + (int64_t)m * a.ldc + nst + 32);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int m = mrow0 + 16 * t + c16;
    const bool mok = m < a.M;
    const int64_t off = (int64_t)m * a.ldc + nst;
    u32x2 rres[4];
    if constexpr (FLAGS & DMI_GEMM_RESIDUAL) {   // fp32 add before rounding, i.e. in the accumulator layout
      if (a.pf & 2) {
        // [r05] fetched as the two 16-byte pieces of the STORE layout (64 contiguous bytes per row and instruction) and brought into
        // the accumulator layout by the row swap, which is its own inverse: same values, half the load instructions
        u32x4 q0 = {0u, 0u, 0u, 0u}, q1 = {0u, 0u, 0u, 0u};
        if (mok && ok0) q0 = *(const u32x4*)(a.residual + off);
        if (mok && ok1) q1 = *(const u32x4*)(a.residual + off + 32);
        unsigned R[4][2] = {{q0[0], q0[1]}, {q0[2], q0[3]}, {q1[0], q1[1]}, {q1[2], q1[3]}};
#pragma unroll
        for (int d = 0; d < 2; ++d) { swap16(R[0][d], R[1][d]); swap16(R[2][d], R[3][d]); }
#pragma unroll
        for (int j = 0; j < 4; ++j) rres[j] = u32x2{R[j][0], R[j][1]};
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          rres[j] = (mok && jok[j]) ? *(const u32x2*)(a.residual + (int64_t)m * a.ldc + ncol0 + 16 * j + 4 * g16) : u32x2{0u, 0u};
      }
    }
    float rsc = 0.f;
    if constexpr (FLAGS & DMI_GEMM_ROWSCALE) rsc = mok ? a.rowscale[m] : 0.f;
    if constexpr (FLAGS & GEMM_SOFTMAX) rsc = (shifted && mok) ? a.rowshift[m] * LOG2E : 0.f;
    u32x4 src0 = {0u, 0u, 0u, 0u}, src1 = {0u, 0u, 0u, 0u};
    if constexpr (FLAGS & DMI_GEMM_RELU_MASK) {   // mask source of THIS row tile: requested one row tile ahead (below), so that its
      src0 = nsrc0; src1 = nsrc1;                 // latency sits behind the previous tile's arithmetic and stores
      const int m2 = m + 16;
      nsrc0 = u32x4{0u, 0u, 0u, 0u}; nsrc1 = u32x4{0u, 0u, 0u, 0u};
      if (t + 1 < NT && m2 < a.M) {
        if (ok0) nsrc0 = *(const u32x4*)(a.relu_src + off + (int64_t)16 * a.ldc);
        if (ok1) nsrc1 = *(const u32x4*)(a.relu_src + off + (int64_t)16 * a.ldc + 32);
      }
    }
    unsigned P[4][2];
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4] = {acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]};
      if constexpr ((FLAGS & DMI_GEMM_BIAS) && !(FLAGS & GEMM_SOFTMAX)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bias[j][e];
      }
      if constexpr (FLAGS & DMI_GEMM_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if constexpr (FLAGS & DMI_GEMM_RESIDUAL) {
        v[0] += __uint_as_float(rres[j][0] << 16);
        v[1] += __uint_as_float(rres[j][0] & 0xffff0000u);
        v[2] += __uint_as_float(rres[j][1] << 16);
        v[3] += __uint_as_float(rres[j][1] & 0xffff0000u);
      }
      if constexpr (FLAGS & DMI_GEMM_ROWSCALE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= rsc;
      }
      if constexpr (FLAGS & GEMM_SOFTMAX) {   // exp(v + bias - shift) as one fma + v_exp_f32 per element; the fp32 values feed the row sum
        float pj = 0.f;
        if (shifted) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[e], LOG2E, bias[j][e] - rsc)); pj += v[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[e], LOG2E, bias[j][e])); pj += v[e]; }
        }
        ps += jok[j] ? pj : 0.f;
      }
      P[j][0] = pack2bf(v[0], v[1]);
      P[j][1] = pack2bf(v[2], v[3]);
    }
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      swap16(P[0][d], P[1][d]);
      swap16(P[2][d], P[3][d]);
    }
    u32x4 lo = {P[0][0], P[0][1], P[1][0], P[1][1]}, hi = {P[2][0], P[2][1], P[3][0], P[3][1]};
    if constexpr (FLAGS & DMI_GEMM_RELU_MASK) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo[e] = relu_mask2(lo[e], src0[e]); hi[e] = relu_mask2(hi[e], src1[e]); }
    }
    if constexpr (FLAGS & GEMM_MASK_BITS) {
      const unsigned mk = mbits[t];
      lo[0] = keep_bits2<0>(lo[0], mk); lo[1] = keep_bits2<2>(lo[1], mk); lo[2] = keep_bits2<4>(lo[2], mk); lo[3] = keep_bits2<6>(lo[3], mk);
      hi[0] = keep_bits2<8>(hi[0], mk); hi[1] = keep_bits2<10>(hi[1], mk); hi[2] = keep_bits2<12>(hi[2], mk); hi[3] = keep_bits2<14>(hi[3], mk);
    }
    if (mok && ok0) store_c16(a, rc, off, lo);
    if (mok && ok1) store_c16(a, rc, off + 32, hi);
    if constexpr (FLAGS & GEMM_RELU_BITS) {
      if (mok && ok0) a.relu_bits[((int64_t)(ncol0 >> 6) * a.M + m) * 4 + g16] = (unsigned short)(relu_bits8(lo) | (relu_bits8(hi) << 8));
    }
    if constexpr (FLAGS & GEMM_SOFTMAX) {
      // the four lane groups of a row hold its 64 columns: two fixed-order exchanges (deterministic), group 0 writes the
      // row's partial to slot ncol0 / 64 of the [slots][M] table
      ps += __shfl_xor(ps, 16, 64);
      ps += __shfl_xor(ps, 32, 64);
      if (g16 == 0 && mok && ncol0 < a.N) a.rowsum_part[(int64_t)(ncol0 >> 6) * a.M + m] = ps;
    }
  }
}

template <int FLAGS, int MI, class ACC>
__device__ __forceinline__ void epilogue_bf16(const GemmArgs& a, ACC& acc, float* stg, int lane, int mrow0, int ncol0) {
  const int orow = lane >> 3, ocol = (lane & 7) * 8;
  const int n = ncol0 + ocol;
  const bool nok = n < a.N;
  constexpr float LOG2E = 1.4426950408889634f;
  float bias[8];
  if constexpr (FLAGS & DMI_GEMM_BIAS) {
    u32x4 braw = {0, 0, 0, 0};
    if (nok) braw = *(const u32x4*)(a.bias + n);
    unpack8(braw, bias);
    if constexpr (FLAGS & GEMM_SOFTMAX) {   // the bias rides in the exponent's fma: exp2(acc * log2e + bias * log2e [- shift * log2e])
#pragma unroll
      for (int e = 0; e < 8; ++e) bias[e] *= LOG2E;
    }
  }
  const bool shifted = (FLAGS & GEMM_SOFTMAX) && a.rowshift != nullptr;   // wave-uniform
  const __amdgpu_buffer_rsrc_t rc = c_rsrc(a);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    u32x4 rres[4], rsrc[4];
    float rsc[4];   // per-row fp32 scalars (row scale / softmax shift) of the 4 store iterations
    float psum[4];  // softmax: fp32 sum of this lane's 8 exponentials per store iteration
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int m = mrow0 + i * 32 + it * 8 + orow;
      const int64_t off = (int64_t)m * a.ldc + n;
      const bool ok = nok && m < a.M;
      if constexpr (FLAGS & DMI_GEMM_RESIDUAL) rres[it] = ok ? *(const u32x4*)(a.residual + off) : u32x4{0, 0, 0, 0};
      if constexpr (FLAGS & DMI_GEMM_RELU_MASK) rsrc[it] = ok ? *(const u32x4*)(a.relu_src + off) : u32x4{0, 0, 0, 0};
      if constexpr (FLAGS & DMI_GEMM_ROWSCALE) rsc[it] = (m < a.M) ? a.rowscale[m] : 0.f;
      if constexpr (FLAGS & GEMM_SOFTMAX) rsc[it] = (shifted && m < a.M) ? a.rowshift[m] * LOG2E : 0.f;
    }
    stage_rows32<MI>(acc, i, stg, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: in-order LDS, no barrier needed
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + orow;
      const int m = mrow0 + i * 32 + row;
      const f32x4 lo = *(const f32x4*)(stg + row * 68 + ocol);
      const f32x4 hi = *(const f32x4*)(stg + row * 68 + ocol + 4);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if constexpr ((FLAGS & DMI_GEMM_BIAS) && !(FLAGS & GEMM_SOFTMAX)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bias[e];
      }
      if constexpr (FLAGS & DMI_GEMM_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if constexpr (FLAGS & DMI_GEMM_RESIDUAL) {
        float b[8];
        unpack8(rres[it], b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b[e];
      }
      if constexpr (FLAGS & DMI_GEMM_RELU_MASK) {
        float b[8];
        unpack8(rsrc[it], b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (b[e] > 0.f) ? v[e] : 0.f;
      }
      if constexpr (FLAGS & DMI_GEMM_ROWSCALE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= rsc[it];
      }
      if constexpr (FLAGS & GEMM_SOFTMAX) {  // exp(v + bias - shift) as one fma + v_exp_f32 per element; the fp32 values feed the row sum
        float ps = 0.f;
        if (shifted) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[e], LOG2E, bias[e] - rsc[it]));
            ps += v[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[e], LOG2E, bias[e]));
            ps += v[e];
          }
        }
        psum[it] = nok ? ps : 0.f;
      }
      if (nok && m < a.M) store_c16(a, rc, (int64_t)m * a.ldc + n, pack8(v));
    }
    if constexpr (FLAGS & GEMM_SOFTMAX) {
      // the 8 lanes of a row (lane & 7) hold its 64 columns: butterfly over xor 1, 2, 4 (fixed order -> deterministic),
      // then lane (orow, 0) writes the row's partial to slot ncol0 / 64 of the [slots][M] table.
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float ps = psum[it];
        ps += __shfl_xor(ps, 1, 64);
        ps += __shfl_xor(ps, 2, 64);
        ps += __shfl_xor(ps, 4, 64);
        const int m = mrow0 + i * 32 + it * 8 + orow;
        if ((lane & 7) == 0 && m < a.M && ncol0 < a.N) a.rowsum_part[(int64_t)(ncol0 >> 6) * a.M + m] = ps;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
// (A non-temporal hint on the LOADS of the streamed operand of the head's two gradient products -- the 4-GB softmax numerators -- was
// measured and dropped: weight gradient 1984 -> 2080 us, input gradient unchanged, step +0.07 ms, profiles/r04ad_kbench_nt.log.)

template <int FLAGS>
__global__ __launch_bounds__(256, 2) void gemm_nt2_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 16K | B 16K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;

  int tm, tn;
  tile_of_block(xcd_remap(blockIdx.x, gridDim.x), a.tiles_m, a.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int kb = blockIdx.y * a.k_per_split;
  const int ke = (kb + a.k_per_split < a.K) ? kb + a.k_per_split : a.K;
  const int nt = (ke - kb) / BK;

  // per-block descriptors based at the tile origin (offsets stay < 2^31 even for the 4 GiB logits matrix)
  const bf16_t* Ab = a.A + (int64_t)m0 * a.lda + kb;
  const bf16_t* Bb = a.B + (int64_t)n0 * a.ldb + kb;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, 0x7fffffff, 0x00020000);
  int voa[4], vob[4];
  {
    const int chp = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const int src_ch = chp ^ ((row >> 1) & 7);
      int ra_ = m0 + row < a.M ? row : a.M - 1 - m0;   // clamp inside the matrix (results discarded)
      int rb_ = n0 + row < a.N ? row : a.N - 1 - n0;
      voa[i] = (ra_ * a.lda + 8 * src_ch) * 2;
      vob[i] = (rb_ * a.ldb + 8 * src_ch) * 2;
    }
  }
  // hoisted fragment byte offsets: chunk (2kk+h) ^ ((row>>1)&7); (row>>1)&7 == (r>>1)&7 for every tile row used
  const int c16 = lane & 15, g16 = lane >> 4;
  int offa[2], offb[2];    // per 32-wide k-substep: chunk 4 ks + g of row (tile base + c); 16-row tiles are 2048 B apart
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    offa[ks] = lds_chunk_off(wm * 64 + c16, ks * 4 + g16);
    offb[ks] = 16384 + lds_chunk_off(wn * 64 + c16, ks * 4 + g16);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int st, int soff) {
    char* base = smem + st * 32768 + wid * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(ra, base + i * 4096, voa[i], soff);
      glds16(rb, base + 16384 + i * 4096, vob[i], soff);
    }
  };
  // [r04] The k-step in buffer st as 8 groups of 4 MFMAs (group g: A fragment g & 3 of k-substep g >> 2 against the four B
  // fragments).  The NEXT k-step's load into the other buffer goes out behind groups 0..3 (one A and one B piece each) instead
  // of as a burst behind the barrier -- the four waves leave it together, and 32 LDS-DMA instructions in a row hold them in the
  // address unit's queue with the matrix pipe idle (see gemm_nt8p_kernel).  Order pinned by sched_barrier(0); fragments
  // requested two groups ahead of their first use.  Same k order as before: bit-identical results.
  auto compute = [&](int st, int soff) {
    const char* cur = smem + st * 32768;
    char* dst = smem + (st ^ 1) * 32768 + wid * 1024;
    bf16x8 fb[2][4], fa[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[0][j] = *(const bf16x8*)(cur + offb[0] + j * 2048);
    fa[0] = *(const bf16x8*)(cur + offa[0]);
    fa[1] = *(const bf16x8*)(cur + offa[0] + 2048);
    MFMA_PRIO(1);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int ks = g >> 2, i = g & 3;
      if (g + 2 < 8) fa[g + 2] = *(const bf16x8*)(cur + offa[(g + 2) >> 2] + ((g + 2) & 3) * 2048);
      if (g == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[1][j] = *(const bf16x8*)(cur + offb[1] + j * 2048);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[g], acc[i][j], 0, 0, 0);  // D[n][m]: lane (c, g) holds C[m = c][n = 4g ..]
      if (g < 4) {
        glds16(ra, dst + g * 4096, voa[g], soff);
        glds16(rb, dst + 16384 + g * 4096, vob[g], soff);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    MFMA_PRIO(0);
  };

  // every k-step loads the next one; the LAST k-step of the tile re-loads k-step 0 into the idle buffer (read by nobody, landed
  // before the epilogue re-uses the buffers: one code path for every k-step keeps the accumulators out of branch merges)
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; t += 2) {
    compute(0, (t + 1 < nt ? t + 1 : 0) * BK * 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nt) {
      compute(1, (t + 2 < nt ? t + 2 : 0) * BK * 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  if constexpr (FLAGS & DMI_GEMM_OUT_F32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + c16;
      if (m >= a.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + 4 * g16;
        if (n >= a.N) continue;
        *(f32x4*)((float*)a.C + (int64_t)blockIdx.y * a.slab_stride + (int64_t)m * a.ldc + n) = acc[i][j];
      }
    }
  } else {
    epilogue_bf16<FLAGS, 2>(a, acc, (float*)(smem + wid * 8704), lane, m0 + wm * 64, n0 + wn * 64);
  }
}

// =====================================================================================
// NT kernel v4: 256x128x32 block tile, 4 waves (2x2), each wave 128x64 = 4x2 MFMA 32x32x16 tiles (128 accumulator
// VGPRs).  Two stages x (A 16 KiB + B 8 KiB) = 48 KiB LDS -> still 2 blocks / CU and the same 16 MFMAs per wave per
// barrier as v2, but per output element 25 % fewer LDS-read / LDS-DMA / L2 bytes and half the per-tile prologue +
// epilogue overhead (the K = 512 shapes of this model ran the MFMA pipe at ~30 % with 128x128 tiles vs 61 % at large K).
// LDS rows are 64 B (4 chunks); chunk ^= (row>>2)&3 on the DMA source side keeps ds_read_b128 conflict-free.
// Used when the grid still covers >= 3 full residencies (N >= 1024 at M = 40960).
// =====================================================================================
#define BM4 256
#define BK4 32
// 64-B LDS rows (4 chunks of 16 B): chunk ^= (-(row >> 2)) & 3.  A ds_read_b128 of the 16x16x32 fragments is served in lane groups
// {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32): rows 0-3 and 12-15 with chunk g, rows 4-11 with chunk g + 1 -- the row-quad map
// 0, 3, 2, 1 puts those four (row quad, chunk) pairs on four different 64-B columns of the 256-B bank row: conflict-free.

template <int FLAGS>
__global__ __launch_bounds__(256, 2) void gemm_nt4_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 16K | B 8K]
  constexpr int STG = 24576;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;

  int tm, tn;
  tile_of_block(xcd_remap(blockIdx.x, gridDim.x), a.tiles_m, a.tiles_n, tm, tn);
  const int m0 = tm * BM4, n0 = tn * BN;
  const int nt = a.K / BK4;  // even (K % 64 == 0)

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (int64_t)m0 * a.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.B + (int64_t)n0 * a.ldb), 0, 0x7fffffff, 0x00020000);
  int voa[4], vob[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 2, pc = c & 3;
    const int rr = m0 + row < a.M ? row : a.M - 1 - m0;
    voa[i] = (rr * a.lda + 8 * (pc ^ lds4_swz(row))) * 2;
    if (i < 2) {
      const int rn = n0 + row < a.N ? row : a.N - 1 - n0;
      vob[i] = (rn * a.ldb + 8 * (pc ^ lds4_swz(row))) * 2;
    }
  }
  const int c16 = lane & 15, g16 = lane >> 4;
  // one 32-wide k-step per stage: chunk g of row (tile base + c); 16-row tiles are 1024 B apart (the swizzle term repeats every 16 rows)
  const int offa = lds4_off(wm * 128 + c16, g16);
  const int offb = 16384 + lds4_off(wn * 64 + c16, g16);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int st, int soff) {
    char* base = smem + st * STG + wid * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(ra, base + i * 4096, voa[i], soff);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(rb, base + 16384 + i * 4096, vob[i], soff);
  };
  auto compute = [&](int st) {
    const char* cur = smem + st * STG;
    MFMA_PRIO(1);
    {
      bf16x8 fa[8], fb[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = *(const bf16x8*)(cur + offa + i * 1024);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8*)(cur + offb + j * 1024);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);  // D[n][m]: lane (c, g) holds C[m = c][n = 4g ..]
    }
    MFMA_PRIO(0);
  };

  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  if (a.dbg) t0 = __builtin_readcyclecounter();
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (a.dbg) t1 = __builtin_readcyclecounter();
  for (int t = 0; t < nt; t += 2) {
    stage(1, (t + 1) * BK4 * 2);
    compute(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 2 < nt) stage(0, (t + 2) * BK4 * 2);
    compute(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (a.dbg) t2 = __builtin_readcyclecounter();

  epilogue_bf16<FLAGS, 4>(a, acc, (float*)(smem + wid * 8704), lane, m0 + wm * 128, n0 + wn * 64);
  if (a.dbg) {  // {start, after prologue, after main loop, stores issued, hw id, stores retired} of this block (wave 0's clock)
    const unsigned long long t3 = __builtin_readcyclecounter();  // stores issued, not necessarily retired
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t4 = __builtin_readcyclecounter();
    if (tid == 0) {
      unsigned hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      unsigned long long* d = a.dbg + (size_t)blockIdx.x * 6;
      d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3; d[4] = hwid; d[5] = t4;
    }
  }
}

// =====================================================================================
// NT kernel v8: 256x256x64 block tile, 8 waves (2 x 4), each wave 128x64 = 4x2 MFMA tiles (128 accumulator VGPRs);
// two stages x (A 32 KiB + B 32 KiB) = 128 KiB LDS -> ONE block of 512 threads per CU, two waves per SIMD.
// Per output element half the L2->LDS traffic of the 128x128 tile and a quarter fewer LDS reads per MFMA; 64 MFMAs per
// wave between barriers, so the stage issued at the top of a K-step has a whole step (~2 us) to land.  For the main-loop
// -bound GEMMs (long K: the head's input gradient); short-K shapes keep the 2-blocks-per-CU kernels whose co-resident
// blocks overlap prologue / epilogue with the other block's main loop.  Same k order as v2/v4 -> bit-identical results.
// =====================================================================================
#define BM8 256
#define BN8 256
template <int FLAGS>
__global__ __launch_bounds__(512, 2) void gemm_nt8_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 32K | B 32K]
  constexpr int STG = 65536;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;

  int tm, tn;
  tile_of_block(xcd_remap(blockIdx.x, gridDim.x), a.tiles_m, a.tiles_n, tm, tn);
  const int m0 = tm * BM8, n0 = tn * BN8;
  const int kb = blockIdx.y * a.k_per_split;
  const int ke = (kb + a.k_per_split < a.K) ? kb + a.k_per_split : a.K;
  const int nt = (ke - kb) / BK;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (int64_t)m0 * a.lda + kb), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.B + (int64_t)n0 * a.ldb + kb), 0, 0x7fffffff, 0x00020000);
  int voa[4], vob[4];
  {
    const int chp = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + 64 * i;
      const int src_ch = chp ^ ((row >> 1) & 7);
      const int ra_ = m0 + row < a.M ? row : a.M - 1 - m0;   // clamp inside the matrix (results discarded)
      const int rb_ = n0 + row < a.N ? row : a.N - 1 - n0;
      voa[i] = (ra_ * a.lda + 8 * src_ch) * 2;
      vob[i] = (rb_ * a.ldb + 8 * src_ch) * 2;
    }
  }
  const int c16 = lane & 15, g16 = lane >> 4;
  int offa[2], offb[2];    // per 32-wide k-substep: chunk 4 ks + g of row (tile base + c); tiles are 16 rows = 2048 B apart
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    offa[ks] = lds_chunk_off(wm * 128 + c16, ks * 4 + g16);
    offb[ks] = 32768 + lds_chunk_off(wn * 64 + c16, ks * 4 + g16);
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int st, int soff) {
    char* base = smem + st * STG + wid * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(ra, base + i * 8192, voa[i], soff);
      glds16(rb, base + 32768 + i * 8192, vob[i], soff);
    }
  };
  // [r04] The k-step in buffer st as 16 groups of 4 MFMAs (group g: A fragment g & 7 of k-substep g >> 3 against the four B
  // fragments).  DMA: behind group p = 0..7 goes piece p (0..3: A rows 64 p.., 4..7: B rows) of the NEXT k-step's load into the
  // other buffer, instead of all eight as a burst behind the barrier (the eight waves leave it together and 64 LDS-DMA
  // instructions in a row hold every wave in the address unit's queue with the matrix pipe idle -- see gemm_nt8p_kernel).
  // Order pinned by sched_barrier(0); fragments requested two groups ahead of their first use.  Same k order as before.
  auto compute = [&](int st, int soff) {
    const char* cur = smem + st * STG;
    char* dst = smem + (st ^ 1) * STG + wid * 1024;
    bf16x8 fb[2][4], fa[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[0][j] = *(const bf16x8*)(cur + offb[0] + j * 2048);
    fa[0] = *(const bf16x8*)(cur + offa[0]);
    fa[1] = *(const bf16x8*)(cur + offa[0] + 2048);
    MFMA_PRIO(1);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int ks = g >> 3, i = g & 7;
      if (g + 2 < 16) fa[g + 2] = *(const bf16x8*)(cur + offa[(g + 2) >> 3] + ((g + 2) & 7) * 2048);
      if (g == 5) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[1][j] = *(const bf16x8*)(cur + offb[1] + j * 2048);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[g], acc[i][j], 0, 0, 0);  // D[n][m]: lane (c, g) holds C[m = c][n = 4g ..]
      if (g < 4) glds16(ra, dst + g * 8192, voa[g], soff);
      else if (g < 8) glds16(rb, dst + 32768 + (g - 4) * 8192, vob[g - 4], soff);
      __builtin_amdgcn_sched_barrier(0);
    }
    MFMA_PRIO(0);
  };
  // every k-step loads the next one; the LAST k-step of the tile re-loads k-step 0 into the idle buffer (read by nobody, landed
  // before the epilogue re-uses the buffers: one code path for every k-step keeps the accumulators out of branch merges)
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; t += 2) {
    compute(0, (t + 1 < nt ? t + 1 : 0) * BK * 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nt) {
      compute(1, (t + 2 < nt ? t + 2 : 0) * BK * 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  if constexpr (FLAGS & DMI_GEMM_OUT_F32) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wm * 128 + i * 16 + c16;
      if (m >= a.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + 4 * g16;
        if (n >= a.N) continue;
        *(f32x4*)((float*)a.C + (int64_t)blockIdx.y * a.slab_stride + (int64_t)m * a.ldc + n) = acc[i][j];
      }
    }
  } else {
    epilogue_bf16<FLAGS, 4>(a, acc, (float*)(smem + wid * 8704), lane, m0 + wm * 128, n0 + wn * 64);
  }
}

// =====================================================================================
// NT kernel v8p (round 4): the 256x256x64 / 8-wave tile of v8 as a PERSISTENT kernel for the short-K products (K = 512:
// QKV, FFN-1, FFN-2 input gradient, vocabulary projection), one block per CU looping over its tiles.
// Why: the LDS-DMA path delivers ~36 B / clk / CU from L2 (three co-resident 256x128 blocks move 72 KB per ~2000-cycle k-step,
// tools/phases.py), so the 256x128 tile (48 B per MFMA-issue clock) cannot run the matrix pipe above ~75 % in its main loop and
// the 128x128 tile (64 B) not above ~55 %; only this tile (32 B) fits.  What it lacked at K = 512 was overlap around its 8
// k-steps (one block per CU: nothing hides the first tile load, the epilogue, the 128 KB of output stores) and inside them:
//   * the register epilogue (epilogue_regs) needs no LDS, so the next tile's k-step 0 is loaded under this tile's LAST k-step
//     and the next main loop starts right behind the epilogue; the epilogue's stores are not waited for -- the first wait that
//     covers them is the vmcnt(0) at the end of the next tile's first k-step;
//   * the eight 1-KB LDS-DMA pieces a wave issues per k-step go out BETWEEN the MFMA groups of the k-step that runs meanwhile,
//     not as a burst behind the barrier: all eight waves leave the barrier together, and 64 LDS-DMA instructions in a row keep
//     every wave in the queue of the CU's one address unit for ~1200 cycles with the matrix pipe idle (the one-tile kernels run
//     4.1 k cycles per k-step against 2 k of MFMA issue per SIMD).
// One buffer descriptor per operand for the whole kernel (based at the matrix origin, exact size: rows past M / N read as
// zeros) -- [r05] one descriptor per operand and TILE, based at the tile's first row: with one whole-matrix descriptor and the tile
// origin in the instruction's scalar offset (round 4) the rows of a ragged last tile past M / N were fetched from beyond the operand
// (the range check does not cover the scalar offset; masked in the epilogue, but an out-of-allocation read).  The per-lane offsets
// are tile-independent and "which tile does this load belong to" is a scalar select of the descriptor -- every k-step of every tile
// runs the same code.
// Tiles: virtual block id v = blockIdx + i * gridDim (gridDim a multiple of 8, so v % 8 is this block's XCD) through the same
// XCD remap + GROUP_M order as the one-tile-per-block kernels: the 32 CUs of an XCD work on 32 consecutive tiles of its list.
// Same k order as every other NT kernel -> bit-identical results.  Needs K % 128 == 0 (a tile's k-steps alternate between the
// two buffers and every tile must start in buffer 0), operands below 2 GiB, and an epilogue that has a register form.
// =====================================================================================
template <int FLAGS>
__global__ __launch_bounds__(512, 2) void gemm_nt8p_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 32K | B 32K]
  constexpr int STG = 65536;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int ntiles = a.tiles_m * a.tiles_n;
  const int nt = a.K / BK;   // even

  const int c16 = lane & 15, g16 = lane >> 4;
  int offa[2], offb[2];    // per 32-wide k-substep: chunk 4 ks + g of row (tile base + c); tiles are 16 rows = 2048 B apart
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    offa[ks] = lds_chunk_off(wm * 128 + c16, ks * 4 + g16);
    offb[ks] = 32768 + lds_chunk_off(wn * 64 + c16, ks * 4 + g16);
  }
  // descriptors based at a tile's first row, sized to the end of the operand: a row past M / N has a vector offset >= num_records
  // and reads as zeros (the range check covers the vector offset only, so the k position -- always inside a valid row -- may ride in
  // the scalar offset, but the tile origin may not)
  auto rsrc_a = [&](int m0_) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (int64_t)m0_ * a.lda), 0, (int)(((int64_t)(a.M - 1 - m0_) * a.lda + a.K) * 2), 0x00020000);
  };
  auto rsrc_b = [&](int n0_) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(a.B + (int64_t)n0_ * a.ldb), 0, (int)(((int64_t)(a.N - 1 - n0_) * a.ldb + a.K) * 2), 0x00020000);
  };
  int voa[4], vob[4];      // per-lane source offsets inside a tile (16-B chunk XOR-swizzled on the source side), tile-independent
  {
    const int chp = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + 64 * i;
      const int src_ch = chp ^ ((row >> 1) & 7);
      voa[i] = (row * a.lda + 8 * src_ch) * 2;
      vob[i] = (row * a.ldb + 8 * src_ch) * 2;
    }
  }

  f32x4 acc[8][4];
  // The k-step in buffer st as 16 groups of 4 MFMAs (group g: A fragment g & 7 of k-substep g >> 3 against the four B
  // fragments); behind group p = 0..7 goes piece p (0..3: A rows 64 p.., 4..7: B rows) of the load into the OTHER buffer -- the
  // second substep's groups cover the landing of the last pieces.  The order is pinned with sched_barrier(0) between the groups
  // (left alone, hipcc clusters the DMAs again) and the fragment reads are software-pipelined by hand: the A fragment of group
  // g + 2 and, in group 5, the second substep's B fragments are requested two groups ahead of their first use.
  auto compute = [&](int st, const __amdgpu_buffer_rsrc_t ra, const __amdgpu_buffer_rsrc_t rb, int sk) {   // ra / rb: the tile being loaded, sk: its k position (bytes)
    const char* cur = smem + st * STG;
    char* dst = smem + (st ^ 1) * STG + wid * 1024;
    bf16x8 fb[2][4], fa[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[0][j] = *(const bf16x8*)(cur + offb[0] + j * 2048);
    fa[0] = *(const bf16x8*)(cur + offa[0]);
    fa[1] = *(const bf16x8*)(cur + offa[0] + 2048);
    MFMA_PRIO(1);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int ks = g >> 3, i = g & 7;
      if (g + 2 < 16) fa[g + 2] = *(const bf16x8*)(cur + offa[(g + 2) >> 3] + ((g + 2) & 7) * 2048);
      if (g == 5) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[1][j] = *(const bf16x8*)(cur + offb[1] + j * 2048);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[g], acc[i][j], 0, 0, 0);  // D[n][m]: lane (c, g) holds C[m = c][n = 4g ..]
      if (g < 4) glds16(ra, dst + g * 8192, voa[g], sk);
      else if (g < 8) glds16(rb, dst + 32768 + (g - 4) * 8192, vob[g - 4], sk);
      __builtin_amdgcn_sched_barrier(0);
    }
    MFMA_PRIO(0);
  };

  int v = blockIdx.x;
  if (v >= ntiles) return;
  int tm, tn;
  tile_of_block(xcd_remap(v, ntiles), a.tiles_m, a.tiles_n, tm, tn);
  int m0 = tm * BM8, n0 = tn * BN8;
  {   // first tile only: nothing to hide its first k-step behind
    char* base = smem + wid * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(rsrc_a(m0), base + i * 8192, voa[i], 0);
      glds16(rsrc_b(n0), base + 32768 + i * 8192, vob[i], 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  unsigned long long c_main = 0, c_epi = 0, c_first = 0, c_start = 0;   // tools/phases.py: per-block phase totals (a.dbg only)
  int ntl = 0;
  if (a.dbg) c_start = __builtin_readcyclecounter();
  while (true) {
    // on entry: k-step 0 of this tile has landed in buffer 0 (every wave waited for its pieces, then the barrier)
    unsigned long long s0 = 0, s1 = 0, s2 = 0;
    if (a.dbg) s0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int vn = v + gridDim.x;
    const bool has_next = vn < ntiles;       // block-uniform
    int nm0 = m0, nn0 = n0;                  // the last tile of a block re-loads its own k-step 0 into the idle buffer: harmless
    if (has_next) {
      int tm2, tn2;
      tile_of_block(xcd_remap(vn, ntiles), a.tiles_m, a.tiles_n, tm2, tn2);
      nm0 = tm2 * BM8; nn0 = tn2 * BN8;
    }
    const __amdgpu_buffer_rsrc_t ra_c = rsrc_a(m0), rb_c = rsrc_b(n0);       // this tile's rows of A / B
    const __amdgpu_buffer_rsrc_t ra_n = rsrc_a(nm0), rb_n = rsrc_b(nn0);     // the next tile's (built at the use instead, the softmax form spills VGPRs)
    for (int t = 0; t < nt; t += 2) {
      // k-step t in buffer 0 while k-step t+1 loads into buffer 1
      compute(0, ra_c, rb_c, (t + 1) * BK * 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // t = 0 of a later tile: also the previous tile's output stores
      __syncthreads();
      if (a.dbg && t == 0) c_first += __builtin_readcyclecounter() - s0;
      // k-step t+1 in buffer 1 while k-step t+2 -- or the NEXT tile's k-step 0 -- loads into buffer 0
      const bool last = t + 2 >= nt;
      compute(1, last ? ra_n : ra_c, last ? rb_n : rb_c, last ? 0 : (t + 2) * BK * 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (a.dbg) s1 = __builtin_readcyclecounter();
    epilogue_regs<FLAGS, 8>(a, acc, lane, m0 + wm * 128, n0 + wn * 64);
    if (a.dbg) { s2 = __builtin_readcyclecounter(); c_main += s1 - s0; c_epi += s2 - s1; ++ntl; }
    if (!has_next) break;
    v = vn; m0 = nm0; n0 = nn0;
  }
  if (a.dbg && tid == 0) {   // {life, main loops, epilogues (issue), first k-steps, tiles} of this block
    unsigned long long* d = a.dbg + (size_t)blockIdx.x * 6;
    d[0] = __builtin_readcyclecounter() - c_start; d[1] = c_main; d[2] = c_epi; d[3] = c_first; d[4] = (unsigned long long)ntl; d[5] = 0;
  }
}

// (A form of this kernel with a deeper load pipeline -- 32-wide k-steps in four buffers, loads 2 or 3 k-steps ahead with a counted
// vmcnt: gemm_nt8q_kernel on the branch r04-kernel-variants -- measured the same cycles per k-step on the head, 3915 / 4145 / 4144
// at depths 1 / 2 / 3, and 2-10 % slower overall (twice the barriers): the load RATE bounds these k-steps, not the load latency.)

// =====================================================================================
// NT kernel "row" (round 4): FULL-ROW tiles for the products with N = 512 outputs (out-projection, FFN-2, the three input
// gradients that end in the residual stream).  Tile = (32 RT) rows x 512 columns, 8 waves as 2 (rows) x 4 (columns), wave tile
// (16 RT) x 128 = RT x 8 MFMA tiles (RT = 5: 160 accumulator registers).  With M = 40960 rows and 256 CUs a tile of 160 rows
// is exactly one tile per CU: no ragged residency, and 672 operand rows are loaded per 160 x 512 outputs -- 1 / 122 B per MAC
// against 1 / 64 for the 128x128 tile these products ran on (the LDS-DMA path from L2 is what bounds the NT kernels, see
// gemm_nt8p_kernel) and 1 / 128 for the 256x256 tile, which N = 512 cannot fill the chip with.  A block owning whole rows is
// also what a fused LayerNorm needs (dmi_gemm_nt_ln).
// k-steps of 32 in three 42-KiB buffers (A (32 RT) rows x 64 B | B 512 rows x 64 B, the 64-B-row swizzle), loads two k-steps
// ahead with a counted vmcnt; the 42 1-KB pieces of a k-step are dealt round-robin to the 8 waves and go out between the
// MFMA groups.  Same k order as every NT kernel -> bit-identical results.
// =====================================================================================
// Epilogue of the full-row kernel with a fused LayerNorm (reference src/dalle_mtf/models.py:330,333,373-389 + layers.py:30-33,
// applied to the output of the product that feeds it: out-projection + residual -> norm_2, FFN-2 + residual -> the next
// block's norm_1 / to_logits' norm):
//   C = bf16(acc + bias + residual)                    (what the unfused path stores, then reads back)
//   mean = sum(C) / 512;  rstd = rsqrt(sum((C - mean)^2) / 512 + eps)          (two passes over the ROUNDED row, as ln_fwd_kernel)
//   Y = bf16((C - mean) * rstd * gamma + beta)
// A row's 512 columns live in the four waves of a row group (128 each): per pass the lanes sum their 32 values, the four lane
// groups combine by two cross-lane exchanges, the four waves through 2.5 KB of LDS (the stage buffers are free by now) in
// fixed order -- deterministic.  The centred values replace the accumulators in place.
template <int RT>
__device__ __forceinline__ void epilogue_ln(const GemmArgs& a, f32x4 (&acc)[RT][8], char* smem, int lane, int wm, int wn, int m0) {
  constexpr int RM = 32 * RT;
  const int c16 = lane & 15, g16 = lane >> 4;
  float* red = (float*)smem;                       // [2 passes][RM rows][4 waves]
  const __amdgpu_buffer_rsrc_t rc = c_rsrc(a);
  const int ncolw = wn * 128;
  const int pcol = 8 * (((g16 & 1) << 1) | (g16 >> 1));   // this lane's 16-B pieces after the row swap: columns pcol + {0..7} and + 32 of a 64-column half
  // ---- C = bf16(acc + bias + residual), kept as fp32 in place; row sums; store C
  float s[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) s[t] = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    unsigned P[RT][4][2];
    unsigned R[RT][4][2];     // [r05] the residual as 16-byte pieces of the store layout, through the row swap (its own inverse)
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int m = m0 + wm * (RM / 2) + 16 * t + c16;
      u32x4 q0 = {0u, 0u, 0u, 0u}, q1 = {0u, 0u, 0u, 0u};
      if (a.residual && m < a.M) {
        const bf16_t* rp = a.residual + (int64_t)m * a.ldc + ncolw + 64 * h + pcol;
        q0 = *(const u32x4*)rp; q1 = *(const u32x4*)(rp + 32);
      }
      R[t][0][0] = q0[0]; R[t][0][1] = q0[1]; R[t][1][0] = q0[2]; R[t][1][1] = q0[3];
      R[t][2][0] = q1[0]; R[t][2][1] = q1[1]; R[t][3][0] = q1[2]; R[t][3][1] = q1[3];
#pragma unroll
      for (int d = 0; d < 2; ++d) { swap16(R[t][0][d], R[t][1][d]); swap16(R[t][2][d], R[t][3][d]); }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 4 * h + jj;
      const int n = ncolw + 16 * j + 4 * g16;
      const u32x2 braw = a.bias ? *(const u32x2*)(a.bias + n) : u32x2{0u, 0u};
      const float b0 = __uint_as_float(braw[0] << 16), b1 = __uint_as_float(braw[0] & 0xffff0000u);
      const float b2 = __uint_as_float(braw[1] << 16), b3 = __uint_as_float(braw[1] & 0xffff0000u);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const u32x2 r = {R[t][jj][0], R[t][jj][1]};
        const float v0 = acc[t][j][0] + b0 + __uint_as_float(r[0] << 16);
        const float v1 = acc[t][j][1] + b1 + __uint_as_float(r[0] & 0xffff0000u);
        const float v2 = acc[t][j][2] + b2 + __uint_as_float(r[1] << 16);
        const float v3 = acc[t][j][3] + b3 + __uint_as_float(r[1] & 0xffff0000u);
        const unsigned p0 = pack2bf(v0, v1), p1 = pack2bf(v2, v3);
        P[t][jj][0] = p0; P[t][jj][1] = p1;
        const f32x4 vr = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
        acc[t][j] = vr;
        s[t] += (vr[0] + vr[1]) + (vr[2] + vr[3]);
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int m = m0 + wm * (RM / 2) + 16 * t + c16;
#pragma unroll
      for (int d = 0; d < 2; ++d) { swap16(P[t][0][d], P[t][1][d]); swap16(P[t][2][d], P[t][3][d]); }
      if (m < a.M) {
        const int64_t off = (int64_t)m * a.ldc + ncolw + 64 * h + pcol;
        store_c16(a, rc, off, u32x4{P[t][0][0], P[t][0][1], P[t][1][0], P[t][1][1]});
        store_c16(a, rc, off + 32, u32x4{P[t][2][0], P[t][2][1], P[t][3][0], P[t][3][1]});
      }
    }
  }
  // ---- mean
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    s[t] += __shfl_xor(s[t], 16, 64);
    s[t] += __shfl_xor(s[t], 32, 64);
    if (g16 == 0) red[(wm * (RM / 2) + 16 * t + c16) * 4 + wn] = s[t];
  }
  __syncthreads();
  float mu[RT], q[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const f32x4 r = *(const f32x4*)(red + (wm * (RM / 2) + 16 * t + c16) * 4);
    mu[t] = ((r[0] + r[1]) + (r[2] + r[3])) * (1.0f / 512.0f);
    q[t] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[t][j][e] -= mu[t]; q[t] += acc[t][j][e] * acc[t][j][e]; }
    }
    q[t] += __shfl_xor(q[t], 16, 64);
    q[t] += __shfl_xor(q[t], 32, 64);
    if (g16 == 0) red[RM * 4 + (wm * (RM / 2) + 16 * t + c16) * 4 + wn] = q[t];
  }
  __syncthreads();
  float rs[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const f32x4 r = *(const f32x4*)(red + RM * 4 + (wm * (RM / 2) + 16 * t + c16) * 4);
    rs[t] = rsqrtf(((r[0] + r[1]) + (r[2] + r[3])) * (1.0f / 512.0f) + a.ln_eps);
    const int m = m0 + wm * (RM / 2) + 16 * t + c16;
    if (wn == 0 && g16 == 0 && m < a.M) { a.ln_mean[m] = mu[t]; a.ln_rstd[m] = rs[t]; }
  }
  // ---- Y = (C - mean) * rstd * gamma + beta
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    unsigned P[RT][4][2];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 4 * h + jj;
      const int n = ncolw + 16 * j + 4 * g16;
      const u32x2 graw = *(const u32x2*)(a.ln_gamma + n), braw = *(const u32x2*)(a.ln_beta + n);
      const float g0 = __uint_as_float(graw[0] << 16), g1 = __uint_as_float(graw[0] & 0xffff0000u);
      const float g2 = __uint_as_float(graw[1] << 16), g3 = __uint_as_float(graw[1] & 0xffff0000u);
      const float b0 = __uint_as_float(braw[0] << 16), b1 = __uint_as_float(braw[0] & 0xffff0000u);
      const float b2 = __uint_as_float(braw[1] << 16), b3 = __uint_as_float(braw[1] & 0xffff0000u);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        P[t][jj][0] = pack2bf(acc[t][j][0] * rs[t] * g0 + b0, acc[t][j][1] * rs[t] * g1 + b1);
        P[t][jj][1] = pack2bf(acc[t][j][2] * rs[t] * g2 + b2, acc[t][j][3] * rs[t] * g3 + b3);
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int m = m0 + wm * (RM / 2) + 16 * t + c16;
#pragma unroll
      for (int d = 0; d < 2; ++d) { swap16(P[t][0][d], P[t][1][d]); swap16(P[t][2][d], P[t][3][d]); }
      if (m < a.M) {
        bf16_t* yp = a.ln_y + (int64_t)m * a.ln_ldy + ncolw + 64 * h + pcol;
        *(u32x4*)yp = u32x4{P[t][0][0], P[t][0][1], P[t][1][0], P[t][1][1]};
        *(u32x4*)(yp + 32) = u32x4{P[t][2][0], P[t][2][1], P[t][3][0], P[t][3][1]};
      }
    }
  }
}

// Epilogue of the full-row kernel with a fused LayerNorm BACKWARD (round 5).  The two input-gradient products of a block that end
// in a LayerNorm -- dxn = dh . W1^T -> norm_2, dxn = dqkv . Wqkv^T -> norm_1 (reference: the backward of src/dalle_mtf/models.py:330,
// 333 through layers.py:30-33 and models.py:387-388) -- wrote dxn (42 MB) for ln_bwd_kernel to read back next to x and the residual
// gradient.  A block of this kernel owns whole rows, so it does the row reductions itself:
//   dy = bf16(acc)                                 (what the unfused path stores and reads back: same rounding)
//   xh = (x - mean) * rstd;  gy = dy * gamma;  s1 = sum_n gy / N;  s2 = sum_n gy * xh / N
//   dx = bf16(rstd * (gy - s1 - xh * s2) + dres);  dgamma += dy * xh;  dbeta += dy   (column partials per 80-row half tile)
// Pass 1 walks the accumulators column tile by column tile (x in 8-byte pieces in the accumulator layout), leaves gy in place,
// reduces the tile's 16 rows of the column partials inside the 16-lane rows with DPP adds (fixed order) and writes one partial row per
// (block, row half); the row sums combine over the four lane groups by two exchanges and over the four waves of a row through 5 KB
// of LDS.  Pass 2 re-reads x (L2-hot) and the residual gradient, forms dx and stores it through the row swap as the other register
// epilogues.  Deterministic; differs from ln_bwd_kernel only in the summation order of the reductions.
__device__ __forceinline__ float dpp_row_sum16(float v) {     // sum over the 16 lanes of a DPP row, result in every lane
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));   // row_mirror
  return v;
}
template <int RT>
__device__ __forceinline__ void epilogue_lnbwd(const GemmArgs& a, f32x4 (&acc)[RT][8], char* smem, int lane, int wm, int wn, int m0) {
  constexpr int RM = 32 * RT;
  const int c16 = lane & 15, g16 = lane >> 4;
  float* red = (float*)smem;                       // [RM rows][4 waves][2]
  const int ncolw = wn * 128;
  const int pcol = 8 * (((g16 & 1) << 1) | (g16 >> 1));
  const float invn = 1.0f / 512.0f;
  // x / dres / dx / mean / rstd through buffer descriptors based at the tile's first row: one 32-bit offset per row tile, the column
  // tile in the instruction's offset field (64-bit pointers per (row, column tile) cost 80 registers); rows past M read as zeros and
  // their stores are dropped
  const int nbytes = (int)(((int64_t)(a.M - m0) * 512) * 2);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ln_x + (int64_t)m0 * 512), 0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)((a.residual ? a.residual : a.ln_x) + (int64_t)m0 * 512), 0,
                                                                       a.residual ? nbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ln_y + (int64_t)m0 * 512), 0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rmu = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ln_mean + m0), 0, (a.M - m0) * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ln_rstd + m0), 0, (a.M - m0) * 4, 0x00020000);
  const int r0 = wm * (RM / 2) + c16;                      // this lane's row of row tile 0 (row tile t: + 16 t)
  float* prow = a.ln_part + (int64_t)(blockIdx.x * 2 + wm) * 1024;
  auto lo = [](unsigned w) { return __uint_as_float(w << 16); };
  auto hi = [](unsigned w) { return __uint_as_float(w & 0xffff0000u); };
  // dy = bf16(acc) and x are kept PACKED (80 + 80 registers; the three passes unpack what they touch).  x and dres are fetched as the
  // 16-byte pieces of the store layout (64 contiguous bytes per row and instruction) and brought into the accumulator layout by the
  // row swap, which is its own inverse -- as 8-byte pieces in the accumulator layout, re-read by every pass, the epilogue cost 50 us.
  // The per-row statistics and gamma are re-loaded by the passes that use them (L1-hot) instead of living through all three.
  unsigned dyp[RT][8][2], xp[RT][8][2];
  const int vst = (r0 * 512 + ncolw + pcol) * 2;      // this lane's 16-byte piece: row of tile 0, column ncolw + pcol (+ 32; + 64 h); row tile t: + 16 KB t
  auto ld16 = [&](const __amdgpu_buffer_rsrc_t rsrc, int voff, int imm) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, imm, 0));
  };
  // [r06] the row-tile term belongs in the VECTOR offset: the descriptor's range check covers voffset + the instruction's immediate, NOT the
  // scalar offset -- with 16384 t (64 t) in soffset a lane whose tile-0 row is below M read and STORED its rows of the tiles t >= 1 past M
  // unchecked (advisor finding, round 5; ragged last tile only, M % 160 != 0).  The compiler folds the small column term into the immediate.
  auto ldmu = [&](int t) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmu, r0 * 4 + 64 * t, 0, 0)); };
  auto ldrs = [&](int t) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, r0 * 4 + 64 * t, 0, 0)); };
  const bf16_t* gp = a.ln_gamma + ncolw + 4 * g16;     // this lane's gamma pieces: + 16 j
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u32x4 l0 = ld16(rx, vst + 16384 * t + 128 * h, 0), l1 = ld16(rx, vst + 16384 * t + 128 * h + 64, 0);
      xp[t][4 * h + 0][0] = l0[0]; xp[t][4 * h + 0][1] = l0[1]; xp[t][4 * h + 1][0] = l0[2]; xp[t][4 * h + 1][1] = l0[3];
      xp[t][4 * h + 2][0] = l1[0]; xp[t][4 * h + 2][1] = l1[1]; xp[t][4 * h + 3][0] = l1[2]; xp[t][4 * h + 3][1] = l1[3];
    }
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dyp[t][j][0] = pack2bf(acc[t][j][0], acc[t][j][1]); dyp[t][j][1] = pack2bf(acc[t][j][2], acc[t][j][3]);
      asm volatile("" : "+v"(dyp[t][j][0]), "+v"(dyp[t][j][1]));   // materialised HERE (left alone, hipcc sinks the conversions to their uses and keeps the accumulators alive)
    }
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) { swap16(xp[t][4 * h][dd], xp[t][4 * h + 1][dd]); swap16(xp[t][4 * h + 2][dd], xp[t][4 * h + 3][dd]); }
  // (between the passes the packed values are made opaque again: otherwise the passes' unpackings are merged into ONE set of 320
  // live floats, which is what spilled)
  auto opaque = [&]() {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(dyp[t][j][0]), "+v"(dyp[t][j][1]), "+v"(xp[t][j][0]), "+v"(xp[t][j][1]));
  };
  opaque();
  // ---- pass 1a: row sums s1 = sum gy, s2 = sum gy * xh, one row tile at a time
  float s1[RT], s2[RT];
  {
    unsigned gmp[8][2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32x2 graw = *(const u32x2*)(gp + 16 * j);
      gmp[j][0] = graw[0]; gmp[j][1] = graw[1];
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const float mu = ldmu(t), rs = ldrs(t);
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv[4] = {lo(xp[t][j][0]), hi(xp[t][j][0]), lo(xp[t][j][1]), hi(xp[t][j][1])};
        const float dy[4] = {lo(dyp[t][j][0]), hi(dyp[t][j][0]), lo(dyp[t][j][1]), hi(dyp[t][j][1])};
        const float gm[4] = {lo(gmp[j][0]), hi(gmp[j][0]), lo(gmp[j][1]), hi(gmp[j][1])};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xv[e] - mu) * rs;
          const float gy = dy[e] * gm[e];
          a1 += gy;
          a2 += gy * xh;
        }
      }
      s1[t] = a1; s2[t] = a2;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the four lane groups of a row, then its four waves through LDS (fixed order)
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    s1[t] += __shfl_xor(s1[t], 16, 64); s1[t] += __shfl_xor(s1[t], 32, 64);
    s2[t] += __shfl_xor(s2[t], 16, 64); s2[t] += __shfl_xor(s2[t], 32, 64);
    if (g16 == 0) {
      float* q = red + ((r0 + 16 * t) * 4 + wn) * 2;
      q[0] = s1[t]; q[1] = s2[t];
    }
  }
  __syncthreads();
  opaque();
  // ---- pass 1b: column partials dgamma += dy * xh, dbeta += dy over this wave's 80 rows, one column tile at a time: the 16 rows of a
  // row tile are the 16 lanes of a DPP row (4 adds, fixed order), lane c16 == 0 of each lane group writes its four columns
  {
    float mu[RT], rs[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) { mu[t] = ldmu(t); rs[t] = ldrs(t); }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const float xv[4] = {lo(xp[t][j][0]), hi(xp[t][j][0]), lo(xp[t][j][1]), hi(xp[t][j][1])};
        const float dy[4] = {lo(dyp[t][j][0]), hi(dyp[t][j][0]), lo(dyp[t][j][1]), hi(dyp[t][j][1])};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ab[e] += dy[e];
          ag[e] += dy[e] * ((xv[e] - mu[t]) * rs[t]);
        }
      }
      f32x4 pg, pb;
#pragma unroll
      for (int e = 0; e < 4; ++e) { pg[e] = dpp_row_sum16(ag[e]); pb[e] = dpp_row_sum16(ab[e]); }
      if (c16 == 0) {
        const int n = ncolw + 16 * j + 4 * g16;
        *(f32x4*)(prow + n) = pg;
        *(f32x4*)(prow + 512 + n) = pb;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  opaque();
  // ---- pass 2: dx = rstd * (dy * gamma - s1 - xh * s2) + dres, one (row tile, 64-column half) at a time; dres arrives as 16-byte pieces
  // and goes through the row swap into the accumulator layout, dx leaves through it
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const f32x4 q0 = *(const f32x4*)(red + (r0 + 16 * t) * 8);
    const f32x4 q1 = *(const f32x4*)(red + (r0 + 16 * t) * 8 + 4);
    const float m1 = ((q0[0] + q0[2]) + (q1[0] + q1[2])) * invn;
    const float m2 = ((q0[1] + q0[3]) + (q1[1] + q1[3])) * invn;
    const float mu = ldmu(t), rs = ldrs(t);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u32x4 l0 = ld16(rr, vst + 16384 * t + 128 * h, 0), l1 = ld16(rr, vst + 16384 * t + 128 * h + 64, 0);
      unsigned R[4][2] = {{l0[0], l0[1]}, {l0[2], l0[3]}, {l1[0], l1[1]}, {l1[2], l1[3]}};
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) { swap16(R[0][dd], R[1][dd]); swap16(R[2][dd], R[3][dd]); }
      unsigned P[4][2];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * h + jj;
        const u32x2 graw = *(const u32x2*)(gp + 16 * j);
        const float xv[4] = {lo(xp[t][j][0]), hi(xp[t][j][0]), lo(xp[t][j][1]), hi(xp[t][j][1])};
        const float rv[4] = {lo(R[jj][0]), hi(R[jj][0]), lo(R[jj][1]), hi(R[jj][1])};
        const float gm[4] = {lo(graw[0]), hi(graw[0]), lo(graw[1]), hi(graw[1])};
        const float dy[4] = {lo(dyp[t][j][0]), hi(dyp[t][j][0]), lo(dyp[t][j][1]), hi(dyp[t][j][1])};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xv[e] - mu) * rs;
          o[e] = rs * (dy[e] * gm[e] - m1 - xh * m2) + rv[e];
        }
        P[jj][0] = pack2bf(o[0], o[1]);
        P[jj][1] = pack2bf(o[2], o[3]);
      }
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) { swap16(P[0][dd], P[1][dd]); swap16(P[2][dd], P[3][dd]); }
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{P[0][0], P[0][1], P[1][0], P[1][1]}, ry, vst + 16384 * t + 128 * h, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{P[2][0], P[2][1], P[3][0], P[3][1]}, ry, vst + 16384 * t + 128 * h + 64, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int FLAGS, int RT, int LN = 0>      // LN: 0 plain register epilogue, 1 fused LayerNorm of the output, 2 fused LayerNorm backward
__global__ __launch_bounds__(512, 2) void gemm_ntr_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RM = 32 * RT;                 // rows per tile
  constexpr int ABYTES = RM * 64;             // A part of a buffer (RT = 5: 10240)
  constexpr int BUF = ABYTES + 32768;
  constexpr int NPA = RM / 16;                // 1-KB pieces of A per k-step (16 rows each)
  constexpr int NP = NPA + 32;                // + B
  constexpr int PW = (NP + 7) / 8;            // pieces per wave (the last round is partial)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int m0 = blockIdx.x * RM;
  int ns = a.K / 32;

  const int c16 = lane & 15, g16 = lane >> 4;
  int offa = lds4_off(wm * (RM / 2) + c16, g16);                // 16-row tiles are 1024 B apart
  int offb = ABYTES + lds4_off(wn * 128 + c16, g16);
  // (operands, offsets and the k-step count are variables: the chained form (LN == 3) runs the main loop a second time on other operands)
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (int64_t)m0 * a.lda), 0,
                                                                 (int)(((int64_t)(a.M - 1 - m0) * a.lda + a.K) * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, 0, (int)(((int64_t)(a.N - 1) * a.ldb + a.K) * 2), 0x00020000);
  // piece p = wid + 8 q of a k-step: p < NPA -> A rows 16 p .., else B rows 16 (p - NPA) ..; this lane's chunk = row 16 p' + lane / 4, 16-B piece lane % 4
  int vo[PW];
  auto set_offsets = [&](int lda_, int ldb_, int ln) {
#pragma unroll
    for (int q = 0; q < PW; ++q) {
      const int p = wid + 8 * q;
      const bool isa = p < NPA;
      const int row = 16 * (isa ? p : p - NPA) + (ln >> 2), pc = ln & 3;
      vo[q] = (row * (isa ? lda_ : ldb_) + 8 * (pc ^ lds4_swz(row))) * 2;
    }
  };
  set_offsets(a.lda, a.ldb, lane);
  auto piece = [&](int buf, int q, int soff) {
    const int p = wid + 8 * q;             // wave-uniform
    if (p >= NP) return;
    char* dst = smem + buf * BUF + (p < NPA ? p * 1024 : ABYTES + (p - NPA) * 1024);
    if (p < NPA) glds16(ra, dst, vo[q], soff);
    else glds16(rb, dst, vo[q], soff);
  };

  f32x4 acc[RT][8];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (A residual L2 prefetch during the last k-steps -- two rows per wave and k-step as LDS-DMA pieces into a scratch block -- was
  // prepared in round 4 and measured neutral in round 5, alone with cold caches and in the step: the whole A operand is cold there,
  // not just the residual.  Removed; profiles/r05a_ab_prefetch.log, r05a_kbench_n512_cold*.log.)
  // k-step in buffer `buf`: RT groups of 8 MFMAs (A fragment i against the eight B fragments); behind the groups go the pieces
  // of the load into buffer `nbuf`
  auto compute = [&](int buf, int nbuf, int soff) {
    const char* cur = smem + buf * BUF;
    bf16x8 fb[8], fa[RT];
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[j] = *(const bf16x8*)(cur + offb + j * 1024);
    fa[0] = *(const bf16x8*)(cur + offa);
    if (RT > 1) fa[1] = *(const bf16x8*)(cur + offa + 1024);
    MFMA_PRIO(1);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      if (i + 2 < RT) fa[i + 2] = *(const bf16x8*)(cur + offa + (i + 2) * 1024);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);  // D[n][m]: lane (c, g) holds C[m = c][n = 4g ..]
#pragma unroll
      for (int q = 2 * i; q < 2 * i + 2; ++q)
        if (q < PW) piece(nbuf, q, soff);
      __builtin_amdgcn_sched_barrier(0);
    }
    MFMA_PRIO(0);
  };

  auto mainloop = [&]() {
    // prologue: k-steps 0 and 1
#pragma unroll
    for (int q = 0; q < PW; ++q) piece(0, q, 0);
#pragma unroll
    for (int q = 0; q < PW; ++q) piece(1, q, ns > 1 ? 64 : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // k-step s in buffer s % 3 while k-step s + 2 loads into buffer (s + 2) % 3 (past the end: k-step 0 again, read by nobody);
    // the pieces of k-step s + 2 may stay in flight across the barrier, those of s + 1 must have landed: every wave issued PW
    // pieces (or PW - 1) per k-step -> vmcnt(PW - 1) is conservative for both
    for (int s = 0; s < ns; s += 3) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (s + u < ns) {
          const int q = s + u + 2;
          compute(u, (u + 2) % 3, (q < ns ? q : 0) * 64);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW - 1) : "memory");
          __builtin_amdgcn_s_barrier();
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  mainloop();
  if constexpr (LN == 3) {
    // LayerNorm backward, then the product that consumes dx, chained in the same launch: dx has just been stored by THIS block (whole
    // rows = the whole contraction range of its rows), so it is read back from L2 as the A operand of a second main loop
    __syncthreads();
    {
      int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(lane_e));
      epilogue_lnbwd<RT>(a, acc, smem, lane_e, wm, wn, m0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave's dx stores have been acknowledged ...
    __syncthreads();                                      // ... before anybody's loads of them (and `red` in the stage buffers is dead)
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ln_y + (int64_t)m0 * a.ln_ldy), 0, (int)(((int64_t)(a.M - 1 - m0) * a.ln_ldy + a.N) * 2), 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc((void*)a.B2, 0, (int)(((int64_t)(a.N - 1) * a.ldb2 + a.N) * 2), 0x00020000);
    // everything per-lane of the second pass is recomputed from here -- the lane index from the exec mask (v_mbcnt), not from the
    // work-item id -- so that nothing of the first pass lives through the epilogue (254 registers)
    int lane2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane2));
    offa = lds4_off(wm * (RM / 2) + (lane2 & 15), lane2 >> 4);
    offb = ABYTES + lds4_off(wn * 128 + (lane2 & 15), lane2 >> 4);
    set_offsets(a.ln_ldy, a.ldb2, lane2);
    ns = a.N / 32;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    mainloop();
    GemmArgs b2 = a;
    b2.C = a.C2; b2.ldc = a.N; b2.cpol = 0;
    f32x4 lo[RT][4], hi[RT][4];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { lo[i][j] = acc[i][j]; hi[i][j] = acc[i][j + 4]; }
    epilogue_regs<0, RT>(b2, lo, lane2, m0 + wm * (RM / 2), wn * 128);
    epilogue_regs<0, RT>(b2, hi, lane2, m0 + wm * (RM / 2), wn * 128 + 64);
  } else if constexpr (LN != 0) {
    // every wave's in-flight pieces have landed before anybody's partial sums overwrite the stage buffers (when (K / 32) % 3 == 2 the
    // last k-step's redundant reload targets buffer 0, where `red` lives: K = 256, 1024)
    __syncthreads();
    if constexpr (LN == 1) epilogue_ln<RT>(a, acc, smem, lane, wm, wn, m0);     // (the last k-step's barrier: every wave is done with the stage buffers)
    else {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));     // opaque: keeps the epilogue's address arithmetic and loads out of the main loop (they spilled 140 registers there)
      epilogue_lnbwd<RT>(a, acc, smem, lane_e, wm, wn, m0);
    }
  } else {   // the register epilogue works on 64-column halves of the wave tile
    f32x4 lo[RT][4], hi[RT][4];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { lo[i][j] = acc[i][j]; hi[i][j] = acc[i][j + 4]; }
    epilogue_regs<FLAGS, RT>(a, lo, lane, m0 + wm * (RM / 2), wn * 128);
    epilogue_regs<FLAGS, RT>(a, hi, lane, m0 + wm * (RM / 2), wn * 128 + 64);
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

// =====================================================================================
// skinny NT GEMM: M <= 32 rows (the incremental decode step: one new position per sequence)
// =====================================================================================
// C[M <= 32, N] = A[M, K] . Bt[N, K]^T with the bias / ReLU / residual epilogues of the decode step.  The 128-row tiles above
// put such a product on N / 128 CUs (4 blocks for N = 512: 19 us at K = 2048, all of it latency); the work is a stream over
// the WEIGHTS (N * K * 2 bytes, HBM / L2 bound), so here a block owns 16 output columns and its eight waves split K in
// 64-element steps: every lane loads 32 contiguous bytes of one weight row and of two activation rows per step (whole 128-B
// lines per 4 lanes), straight into v_mfma_f32_16x16x32_bf16 operands -- the contraction order inside a step is permuted the
// same way for both operands, which a dot product does not see -- and the eight partial tiles are summed through LDS in a
// fixed order.  No LDS staging of operands, no barriers in the K loop.
#define SK_WAVES 8
template <int FLAGS>
__global__ __launch_bounds__(64 * SK_WAVES) void gemm_nt_skinny_kernel(const GemmArgs a) {
  __shared__ f32x4 red[SK_WAVES][2][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int r0 = c < a.M ? c : a.M - 1, r1 = 16 + c < a.M ? 16 + c : a.M - 1;   // rows past M: clamped, never stored
  const bf16_t* pa0 = a.A + (int64_t)r0 * a.lda + 16 * g;
  const bf16_t* pa1 = a.A + (int64_t)r1 * a.lda + 16 * g;
  const bf16_t* pb = a.B + (int64_t)(n0 + c) * a.ldb + 16 * g;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int k0 = 64 * wid; k0 < a.K; k0 += 64 * SK_WAVES) {
    const bf16x8 b0 = *(const bf16x8*)(pb + k0), b1 = *(const bf16x8*)(pb + k0 + 8);
    const bf16x8 x0 = *(const bf16x8*)(pa0 + k0), x1 = *(const bf16x8*)(pa0 + k0 + 8);
    const bf16x8 y0 = *(const bf16x8*)(pa1 + k0), y1 = *(const bf16x8*)(pa1 + k0 + 8);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0, b0, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, b1, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y1, b1, acc1, 0, 0, 0);
  }
  red[wid][0][lane] = acc0;
  red[wid][1][lane] = acc1;
  __syncthreads();
  if (wid >= 2) return;
  // waves 0 / 1 finish rows 0..15 / 16..31: lane (c, g) holds column n0 + c of rows 4g + e
  f32x4 v = red[0][wid][lane];
#pragma unroll
  for (int w = 1; w < SK_WAVES; ++w) {
    const f32x4 t = red[w][wid][lane];
    v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
  }
  const int n = n0 + c;
  float bias = 0.f;
  if constexpr (FLAGS & DMI_GEMM_BIAS) bias = bf2f(a.bias[n]);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = 16 * wid + 4 * g + e;
    if (m >= a.M) continue;
    float x = v[e] + bias;
    if constexpr (FLAGS & DMI_GEMM_RELU) x = fmaxf(x, 0.f);
    if constexpr (FLAGS & DMI_GEMM_RESIDUAL) x += bf2f(a.residual[(int64_t)m * a.ldc + n]);
    ((bf16_t*)a.C)[(int64_t)m * a.ldc + n] = f2bf(x);
  }
}

// LayerNorm fused into the skinny product (decode step: layer_norm + dense, reference src/dalle_mtf/models.py:373-389 feeding
// :242-244 / :317-324 / :391-395): C[M <= 32, N] = LN(X)[M, K] . Bt[N, K]^T (+ bias, ReLU), K = the normalised width.  Every block
// normalises the <= 32 rows itself (32 x K bf16 from L2 -- nothing next to the weight stream): the raw rows sit in the MFMA
// operand registers, mean and centred variance are reduced across the lane groups and the eight waves (two LDS round trips, the
// two-pass form of ln_fwd_kernel), then each operand is normalised in fp32 and rounded to bf16 exactly where the standalone
// kernel rounds its output.  Saves one of the ~7 us dependent launches per LayerNorm of the decode step.
#define SKLN_MAXSTEPS 4   // K <= 64 * SK_WAVES * 4 = 2048
template <int FLAGS>
__global__ __launch_bounds__(64 * SK_WAVES) void gemm_ln_nt_skinny_kernel(const GemmArgs a, const bf16_t* __restrict__ gamma,
                                                                          const bf16_t* __restrict__ beta, float eps) {
  __shared__ f32x4 red[SK_WAVES][2][64];
  __shared__ float part[SK_WAVES][32];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int r0 = c < a.M ? c : a.M - 1, r1 = 16 + c < a.M ? 16 + c : a.M - 1;
  const bf16_t* pa0 = a.A + (int64_t)r0 * a.lda + 16 * g;
  const bf16_t* pa1 = a.A + (int64_t)r1 * a.lda + 16 * g;
  const bf16_t* pb = a.B + (int64_t)(n0 + c) * a.ldb + 16 * g;
  float x[SKLN_MAXSTEPS][2][16];   // [step][row c | row 16 + c][k = k0 + 16 g + 0..15]
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int t = 0; t < SKLN_MAXSTEPS; ++t) {
    const int k0 = 64 * (wid + SK_WAVES * t);
    if (k0 < a.K) {
      unpack8(*(const u32x4*)(pa0 + k0), x[t][0]);
      unpack8(*(const u32x4*)(pa0 + k0 + 8), x[t][0] + 8);
      unpack8(*(const u32x4*)(pa1 + k0), x[t][1]);
      unpack8(*(const u32x4*)(pa1 + k0 + 8), x[t][1] + 8);
#pragma unroll
      for (int j = 0; j < 16; ++j) { s0 += x[t][0][j]; s1 += x[t][1][j]; }
    }
  }
  auto rows_total = [&](float v0, float v1, float& t0, float& t1) {   // sum over the 4 lane groups and the 8 waves, fixed order
    v0 += __shfl_xor(v0, 16, 64); v0 += __shfl_xor(v0, 32, 64);
    v1 += __shfl_xor(v1, 16, 64); v1 += __shfl_xor(v1, 32, 64);
    if (g == 0) { part[wid][c] = v0; part[wid][16 + c] = v1; }
    __syncthreads();
    t0 = 0.f; t1 = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) { t0 += part[w][c]; t1 += part[w][16 + c]; }
    __syncthreads();
  };
  float mu0, mu1, q0 = 0.f, q1 = 0.f;
  rows_total(s0, s1, mu0, mu1);
  mu0 /= (float)a.K; mu1 /= (float)a.K;
#pragma unroll
  for (int t = 0; t < SKLN_MAXSTEPS; ++t) {
    if (64 * (wid + SK_WAVES * t) < a.K) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        x[t][0][j] -= mu0; q0 += x[t][0][j] * x[t][0][j];
        x[t][1][j] -= mu1; q1 += x[t][1][j] * x[t][1][j];
      }
    }
  }
  float rs0, rs1;
  rows_total(q0, q1, rs0, rs1);
  rs0 = rsqrtf(rs0 / (float)a.K + eps);
  rs1 = rsqrtf(rs1 / (float)a.K + eps);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < SKLN_MAXSTEPS; ++t) {
    const int k0 = 64 * (wid + SK_WAVES * t);
    if (k0 < a.K) {
      float fg[16], fb[16], o0[16], o1[16];
      unpack8(*(const u32x4*)(gamma + k0 + 16 * g), fg);
      unpack8(*(const u32x4*)(gamma + k0 + 16 * g + 8), fg + 8);
      unpack8(*(const u32x4*)(beta + k0 + 16 * g), fb);
      unpack8(*(const u32x4*)(beta + k0 + 16 * g + 8), fb + 8);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        o0[j] = x[t][0][j] * rs0 * fg[j] + fb[j];
        o1[j] = x[t][1][j] * rs1 * fg[j] + fb[j];
      }
      const bf16x8 b0 = *(const bf16x8*)(pb + k0), b1 = *(const bf16x8*)(pb + k0 + 8);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pack8(o0)), b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pack8(o1)), b0, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pack8(o0 + 8)), b1, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pack8(o1 + 8)), b1, acc1, 0, 0, 0);
    }
  }
  red[wid][0][lane] = acc0;
  red[wid][1][lane] = acc1;
  __syncthreads();
  if (wid >= 2) return;
  f32x4 v = red[0][wid][lane];
#pragma unroll
  for (int w = 1; w < SK_WAVES; ++w) {
    const f32x4 t = red[w][wid][lane];
    v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
  }
  const int n = n0 + c;
  float bias = 0.f;
  if constexpr (FLAGS & DMI_GEMM_BIAS) bias = bf2f(a.bias[n]);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = 16 * wid + 4 * g + e;
    if (m >= a.M) continue;
    float y = v[e] + bias;
    if constexpr (FLAGS & DMI_GEMM_RELU) y = fmaxf(y, 0.f);
    ((bf16_t*)a.C)[(int64_t)m * a.ldc + n] = f2bf(y);
  }
}

static int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n < 8) n = 256;
  }
  return n;
}

static int persistent_grid() {    // blocks of a one-per-CU persistent kernel: a multiple of 8 (block id % 8 = XCD)
  int n = (num_cus() - g_opt_reserve_cus) & ~7;
  return n < 8 ? 8 : n;
}

// persistent 256x256 tiles (gemm_nt8p_kernel): what a launch needs, and whether the automatic choice takes it (short K, at
// least two tiles per CU)
static bool nt8p_can(const GemmArgs& a) {
  return a.k_per_split == a.K && a.K % 128 == 0 && (int64_t)a.M * a.lda < (1 << 30) && (int64_t)a.N * a.ldb < (1 << 30);
}
static bool nt8p_auto(int M, int N, int K) {
  return K <= g_opt_nt8p_max_k && ((M + BM8 - 1) / BM8) * ((N + BN8 - 1) / BN8) >= 2 * num_cus();
}
template <int FLAGS>
static int launch_nt8p(const GemmArgs& a, hipStream_t st) {
  static bool attr8p = false;
  if (!attr8p) { (void)hipFuncSetAttribute((const void*)gemm_nt8p_kernel<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); attr8p = true; }
  GemmArgs b = a;
  b.tiles_m = (a.M + BM8 - 1) / BM8; b.tiles_n = (a.N + BN8 - 1) / BN8;
  gemm_nt8p_kernel<FLAGS><<<dim3(persistent_grid()), dim3(512), 131072, st>>>(b);
  DMI_CHECK_LAUNCH("gemm_nt8p");
  return DMI_OK;
}

template <int FLAGS>
static int launch_nt(const GemmArgs& a, int nsplit, hipStream_t st) {
  const dim3 grid(a.tiles_m * a.tiles_n, nsplit), blk(256);
  if constexpr (FLAGS == 0 || FLAGS == DMI_GEMM_BIAS || FLAGS == (DMI_GEMM_BIAS | DMI_GEMM_RELU) || FLAGS == (DMI_GEMM_BIAS | DMI_GEMM_RESIDUAL)) {
    if (g_opt_skinny && a.M <= 32 && nsplit == 1 && a.N % 16 == 0) {
      gemm_nt_skinny_kernel<FLAGS><<<dim3(a.N / 16), dim3(64 * SK_WAVES), 0, st>>>(a);
      DMI_CHECK_LAUNCH("gemm_nt_skinny");
      return DMI_OK;
    }
  }
  if constexpr (epi_regs_ok<FLAGS> && !(FLAGS & GEMM_SOFTMAX)) {
    // full-row tiles: N = 512 exactly, one 160-row tile per block
    if (g_opt_ntr && a.N == 512 && nsplit == 1 && a.k_per_split == a.K && a.K % 32 == 0 && (int64_t)a.N * a.ldb < (1 << 30) &&
        (g_opt_ntr == 2 || (a.M >= 160 * (num_cus() / 2) && a.K <= 4096 && g_opt_reserve_cus == 0))) {   // (the head's input gradient, K = 50816: 1.83 ms here vs 1.32 ms on 256x256 tiles)
      constexpr int RT = 5;
      constexpr int LDSB = 3 * (32 * RT * 64 + 32768);
      static bool attrr = false;
      if (!attrr) { (void)hipFuncSetAttribute((const void*)gemm_ntr_kernel<FLAGS, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); attrr = true; }
      GemmArgs b = a;
      b.pf = g_opt_res16 ? 2 : 0;     // residual rows as 16-byte pieces (A/B switch)
      gemm_ntr_kernel<FLAGS, RT><<<dim3((a.M + 32 * RT - 1) / (32 * RT)), dim3(512), LDSB, st>>>(b);
      DMI_CHECK_LAUNCH("gemm_ntr");
      return DMI_OK;
    }
  }
  if constexpr (epi_regs_ok<FLAGS>) {
    // persistent 256x256 tiles: short-K products with at least two tiles per CU (see gemm_nt8p_kernel)
    const bool can = nsplit == 1 && nt8p_can(a);
    const bool auto_ok = nt8p_auto(a.M, a.N, a.K);
    // 3 = auto for the softmax head only (tests: its register epilogue adds the row-sum partials in its own fixed order, so a
    // step that is to be compared bit for bit with the 128x128 kernels keeps the head where it is)
    if (can && ((g_opt_nt8p == 1 && auto_ok) || g_opt_nt8p == 2 || (g_opt_nt8p == 3 && (FLAGS & GEMM_SOFTMAX) && auto_ok)))
      return launch_nt8p<FLAGS>(a, st);
  }
  {
    const int t8m = (a.M + BM8 - 1) / BM8, t8n = (a.N + BN8 - 1) / BN8;
    // 256x256 tiles (one 8-wave block per CU): main-loop-bound shapes only -- long K and whole residencies of 256 blocks
    const bool auto8 = g_opt_nt8 == 1 && a.k_per_split >= g_opt_nt8_min_k && (t8m * t8n * nsplit) % 256 == 0;
    if (auto8 || g_opt_nt8 == 2) {
      static bool attr8 = false;
      if (!attr8) { (void)hipFuncSetAttribute((const void*)gemm_nt8_kernel<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); attr8 = true; }
      GemmArgs b = a;
      b.tiles_m = t8m; b.tiles_n = t8n;
      gemm_nt8_kernel<FLAGS><<<dim3(t8m * t8n, nsplit), dim3(512), 131072, st>>>(b);
      DMI_CHECK_LAUNCH("gemm_nt8");
      return DMI_OK;
    }
  }
  if constexpr (!(FLAGS & DMI_GEMM_OUT_F32)) {
    const int tiles4 = ((a.M + BM4 - 1) / BM4) * a.tiles_n;
    // 256x128 tiles when the grid still covers >= 3 residencies; long-K shapes prefer the BK = 64 kernel.  2 = force (tests)
    if (g_opt_nt4 && nsplit == 1 && a.k_per_split == a.K && ((tiles4 >= 1536 && a.K <= 1024) || g_opt_nt4 == 2)) {
      static int attr4 = 0;
      if (attr4 != g_opt_nt4_lds) { (void)hipFuncSetAttribute((const void*)gemm_nt4_kernel<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, g_opt_nt4_lds); attr4 = g_opt_nt4_lds; }
      GemmArgs b = a;
      b.tiles_m = (a.M + BM4 - 1) / BM4;
      gemm_nt4_kernel<FLAGS><<<dim3(tiles4), blk, g_opt_nt4_lds, st>>>(b);
      DMI_CHECK_LAUNCH("gemm_nt4");
      return DMI_OK;
    }
  }
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_nt2_kernel<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); attr_done = true; }
  gemm_nt2_kernel<FLAGS><<<grid, blk, 65536, st>>>(a);
  DMI_CHECK_LAUNCH("gemm_nt");
  return DMI_OK;
}

static int check_nt(const void* A, int lda, const void* B, int ldb, const void* C, int ldc, int M, int N, int K) {
  DMI_REQUIRE(A && B && C, "gemm_nt: null pointer");
  DMI_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0 && N % 8 == 0, "gemm_nt: need K%%64==0 and N%%8==0 (M=%d N=%d K=%d)", M, N, K);
  DMI_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && lda >= K && ldb >= K && ldc >= N, "gemm_nt: bad leading dimensions");
  DMI_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0 && ((uintptr_t)C & 15) == 0, "gemm_nt: operands must be 16-byte aligned");
  return DMI_OK;
}

static void fill_nt_args(GemmArgs& a, const uint16_t* A, int lda, const uint16_t* Bt, int ldb, void* C, int ldc, int M, int N, int K) {
  a.A = A; a.B = Bt; a.C = C; a.bias = nullptr; a.residual = nullptr; a.relu_src = nullptr; a.relu_bits = nullptr;
  a.rowscale = nullptr; a.rowshift = nullptr; a.rowsum_part = nullptr;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
  a.k_per_split = K; a.slab_stride = 0; a.dbg = nullptr; a.pf = 0;
  a.ln_gamma = nullptr; a.ln_beta = nullptr; a.ln_y = nullptr; a.ln_mean = nullptr; a.ln_rstd = nullptr; a.ln_eps = 0.f; a.ln_ldy = 0;
  a.ln_x = nullptr; a.ln_part = nullptr; a.B2 = nullptr; a.C2 = nullptr; a.ldb2 = 0;
  const int64_t cbytes = ((int64_t)(M - 1) * ldc + N) * 2;     // bf16 outputs (the fp32 forms do not use the policy)
  // 1 (default): auto by size -- sc1 from cstream_min_mb, sc1 nt from cstream_nt_min_mb (a stream far larger than the 256-MB
  // Infinity Cache: the 4-GB softmax numerators; outputs of a few hundred MB that the next kernel re-reads were measured with
  // sc1 only and keep it); 2: always sc1; 3: always sc1 nt; 4: auto by size, sc1 only (A/B)
  const bool big = cbytes >= ((int64_t)g_opt_cstream_min_mb << 20), huge = cbytes >= ((int64_t)g_opt_cstream_nt_min_mb << 20);
  const int pol = (g_opt_cstream == 1 && big) ? (huge ? 2 : 1) : (g_opt_cstream == 2) ? 1 : (g_opt_cstream == 3) ? 2 : (g_opt_cstream == 4 && big) ? 1 : 0;
  a.cpol = cbytes < (int64_t)0xffffffff ? pol : 0;
}

extern "C" int dmi_gemm_nt(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, void* C, int ldc, int M, int N,
                           int K, int flags, const uint16_t* bias, const uint16_t* residual, const uint16_t* relu_src,
                           const float* rowscale, void* stream) {
  int rc = check_nt(A, lda, Bt, ldb, C, ldc, M, N, K);
  if (rc) return rc;
  DMI_REQUIRE(!(flags & DMI_GEMM_BIAS) || bias, "gemm_nt: bias flag without pointer");
  DMI_REQUIRE(!(flags & DMI_GEMM_RESIDUAL) || residual, "gemm_nt: residual flag without pointer");
  DMI_REQUIRE((((uintptr_t)residual | (uintptr_t)relu_src) & 15) == 0, "gemm_nt: residual / relu_src must be 16-byte aligned");
  DMI_REQUIRE(!(flags & DMI_GEMM_RELU_MASK) || relu_src, "gemm_nt: relu-mask flag without pointer");
  DMI_REQUIRE(!(flags & DMI_GEMM_ROWSCALE) || rowscale, "gemm_nt: row-scale flag without pointer");
  GemmArgs a;
  fill_nt_args(a, A, lda, Bt, ldb, C, ldc, M, N, K);
  a.bias = bias; a.residual = residual; a.relu_src = relu_src; a.rowscale = rowscale; a.dbg = g_dbg_buf;
  hipStream_t st = (hipStream_t)stream;
  switch (flags) {
    case 0: return launch_nt<0>(a, 1, st);
    case DMI_GEMM_BIAS: return launch_nt<DMI_GEMM_BIAS>(a, 1, st);
    case DMI_GEMM_BIAS | DMI_GEMM_RELU: return launch_nt<DMI_GEMM_BIAS | DMI_GEMM_RELU>(a, 1, st);
    case DMI_GEMM_BIAS | DMI_GEMM_RESIDUAL: return launch_nt<DMI_GEMM_BIAS | DMI_GEMM_RESIDUAL>(a, 1, st);
    case DMI_GEMM_RELU_MASK: return launch_nt<DMI_GEMM_RELU_MASK>(a, 1, st);
    case DMI_GEMM_RESIDUAL: return launch_nt<DMI_GEMM_RESIDUAL>(a, 1, st);
    case DMI_GEMM_ROWSCALE: return launch_nt<DMI_GEMM_ROWSCALE>(a, 1, st);
    case DMI_GEMM_OUT_F32: return launch_nt<DMI_GEMM_OUT_F32>(a, 1, st);
    default:
      dmi_set_error("gemm_nt: unsupported flag combination %d", flags);
      return DMI_ERR_UNSUPPORTED;
  }
}

// ---- the ReLU mask as bits (see relu_bits8 above): FFN-1 forward emits them, the FFN-2 input gradient applies them ----
// reference: h = mtf.relu(dense(x)) (src/dalle_mtf/models.py:320-321) and its backward dh = dh * (h > 0).
extern "C" int64_t dmi_relu_bits_bytes(int M, int N) { return (int64_t)((N + 63) / 64) * M * 8; }
// 1 if the automatic dispatch runs these shapes on the kernel that has the bit forms (the engine then uses them; otherwise it keeps
// dmi_gemm_nt with DMI_GEMM_RELU / DMI_GEMM_RELU_MASK, which every tile size serves)
extern "C" int dmi_relu_bits_auto(int M, int N, int K) {
  return (g_opt_relu_bits && g_opt_nt8p == 1 && N % 64 == 0 && K % 128 == 0 && nt8p_auto(M, N, K)) ? 1 : 0;
}
static int check_bits(const char* who, const GemmArgs& a, const void* bits) {
  DMI_REQUIRE(bits && ((uintptr_t)bits & 7) == 0, "%s: null / unaligned bit buffer", who);
  if (a.N % 64 != 0 || !nt8p_can(a)) {
    dmi_set_error("%s: needs N %% 64 == 0, K %% 128 == 0 and operands below 2 GiB (M=%d N=%d K=%d)", who, a.M, a.N, a.K);
    return DMI_ERR_UNSUPPORTED;
  }
  return DMI_OK;
}
extern "C" int dmi_gemm_nt_relu_bits(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, uint16_t* C, int ldc, int M, int N, int K,
                                     const uint16_t* bias, void* bits, void* stream) {
  int rc = check_nt(A, lda, Bt, ldb, C, ldc, M, N, K);
  if (rc) return rc;
  DMI_REQUIRE(bias, "gemm_nt_relu_bits: null bias");
  GemmArgs a;
  fill_nt_args(a, A, lda, Bt, ldb, C, ldc, M, N, K);
  a.bias = bias; a.relu_bits = (unsigned short*)bits; a.dbg = g_dbg_buf;
  if ((rc = check_bits("gemm_nt_relu_bits", a, bits))) return rc;
  return launch_nt8p<DMI_GEMM_BIAS | DMI_GEMM_RELU | GEMM_RELU_BITS>(a, (hipStream_t)stream);
}
extern "C" int dmi_gemm_nt_mask_bits(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, uint16_t* C, int ldc, int M, int N, int K,
                                     const void* bits, void* stream) {
  int rc = check_nt(A, lda, Bt, ldb, C, ldc, M, N, K);
  if (rc) return rc;
  GemmArgs a;
  fill_nt_args(a, A, lda, Bt, ldb, C, ldc, M, N, K);
  a.relu_bits = (unsigned short*)bits; a.dbg = g_dbg_buf;
  if ((rc = check_bits("gemm_nt_mask_bits", a, bits))) return rc;
  return launch_nt8p<GEMM_MASK_BITS>(a, (hipStream_t)stream);
}

// 1 where dmi_gemm_nt_ln / dmi_gemm_nt_lnbwd accept a product [M, K] x [N, K]^T with K-contiguous operands (lda = ldb = K) AND the
// library would run it on full-row tiles itself: N = 512 (one block owns whole rows), every operand inside the 32-bit buffer offsets
// of gemm_ntr_kernel, no CUs reserved for a concurrent exchange (launch_nt keeps the 128x128 kernel then).  Callers gate the fused
// forms on it when they size their buffers and keep the two-kernel path otherwise (advisor finding, round 5).
extern "C" int dmi_gemm_nt_ln_auto(int M, int N, int K) {
  if (M <= 0 || N != 512 || K <= 0 || K % BK != 0) return 0;
  if ((int64_t)N * K >= (1 << 30) || (int64_t)M * K >= ((int64_t)1 << 31) || (int64_t)M * N * 2 >= ((int64_t)1 << 31)) return 0;
  return g_opt_reserve_cus == 0 ? 1 : 0;
}

// Product with N = 512 outputs + bias + residual, and the LayerNorm of the result in the same pass (full-row tiles, see
// gemm_ntr_kernel / epilogue_ln): C = bf16(A . Bt^T + bias + residual) [M, 512], Y = bf16(LN(C) * gamma + beta), mean / rstd fp32 [M].
extern "C" int dmi_gemm_nt_ln(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, uint16_t* C, int ldc, int M, int N, int K,
                              const uint16_t* bias, const uint16_t* residual, const uint16_t* gamma, const uint16_t* beta, float eps,
                              uint16_t* Y, int ldy, float* mean, float* rstd, void* stream) {
  int rc = check_nt(A, lda, Bt, ldb, C, ldc, M, N, K);
  if (rc) return rc;
  DMI_REQUIRE(gamma && beta && Y && mean && rstd, "gemm_nt_ln: null pointer");
  DMI_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)Y | (uintptr_t)bias | (uintptr_t)residual) & 15) == 0 && ldy % 8 == 0 && ldy >= N,
              "gemm_nt_ln: operands must be 16-byte aligned");
  if (N != 512 || (int64_t)N * ldb >= (1 << 30) || (int64_t)M * lda >= ((int64_t)1 << 31)) {
    dmi_set_error("gemm_nt_ln: the fused form needs N = 512 (one block owns whole rows); got N=%d", N);
    return DMI_ERR_UNSUPPORTED;
  }
  GemmArgs a;
  fill_nt_args(a, A, lda, Bt, ldb, C, ldc, M, N, K);
  a.bias = bias; a.residual = residual;
  a.ln_gamma = gamma; a.ln_beta = beta; a.ln_y = Y; a.ln_ldy = ldy; a.ln_mean = mean; a.ln_rstd = rstd; a.ln_eps = eps;
  constexpr int RT = 5;
  constexpr int LDSB = 3 * (32 * RT * 64 + 32768);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_ntr_kernel<0, RT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); attr = true; }
  gemm_ntr_kernel<0, RT, 1><<<dim3((M + 32 * RT - 1) / (32 * RT)), dim3(512), LDSB, (hipStream_t)stream>>>(a);
  DMI_CHECK_LAUNCH("gemm_nt_ln");
  return DMI_OK;
}

// Product with N = 512 outputs whose result is the gradient arriving at a LayerNorm, and that LayerNorm's backward in the same pass
// (see epilogue_lnbwd): dx = LN'(x; gamma, mean, rstd)(A . Bt^T) + dres.  part: dmi_gemm_nt_lnbwd_parts(M) rows of [2 N] fp32 partial
// gain | bias gradients, to be summed by dmi_layernorm_bwd_finish_parts.
extern "C" int dmi_gemm_nt_lnbwd_parts(int M) { return 2 * ((M + 159) / 160); }
extern "C" int dmi_gemm_nt_lnbwd(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, int M, int N, int K, const uint16_t* x,
                                 const uint16_t* gamma, const float* mean, const float* rstd, const uint16_t* dres, uint16_t* dx,
                                 float* part, const uint16_t* B2, int ldb2, uint16_t* C2, void* stream) {
  int rc = check_nt(A, lda, Bt, ldb, dx, N, M, N, K);
  if (rc) return rc;
  DMI_REQUIRE(x && gamma && mean && rstd && part, "gemm_nt_lnbwd: null pointer");
  DMI_REQUIRE((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dres | (uintptr_t)part) & 15) == 0, "gemm_nt_lnbwd: operands must be 16-byte aligned");
  if (N != 512 || (int64_t)N * ldb >= (1 << 30) || (int64_t)M * lda >= ((int64_t)1 << 31) || (int64_t)M * N * 2 >= ((int64_t)1 << 31)) {
    dmi_set_error("gemm_nt_lnbwd: the fused form needs N = 512 (one block owns whole rows) and operands below 2 GiB; got M=%d N=%d", M, N);
    return DMI_ERR_UNSUPPORTED;
  }
  GemmArgs a;
  fill_nt_args(a, A, lda, Bt, ldb, dx, N, M, N, K);
  a.residual = dres; a.ln_x = x; a.ln_gamma = gamma; a.ln_mean = (float*)mean; a.ln_rstd = (float*)rstd; a.ln_y = dx; a.ln_ldy = N; a.ln_part = part;
  constexpr int RT = 5;
  constexpr int LDSB = 3 * (32 * RT * 64 + 32768);
  if (B2 || C2) {     // chained: C2[M, N] = dx . B2^T in the same launch (B2 [N, ldb2] bf16, K-contiguous like Bt; C2 row pitch N)
    DMI_REQUIRE(B2 && C2 && ldb2 >= N && ldb2 % 8 == 0 && (((uintptr_t)B2 | (uintptr_t)C2) & 15) == 0 && (const void*)C2 != (const void*)dx,
                "gemm_nt_lnbwd: bad chained product (B2 / ldb2 / C2)");
    a.B2 = B2; a.ldb2 = ldb2; a.C2 = C2;
    static bool attr3 = false;
    if (!attr3) { (void)hipFuncSetAttribute((const void*)gemm_ntr_kernel<0, RT, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); attr3 = true; }
    gemm_ntr_kernel<0, RT, 3><<<dim3((M + 32 * RT - 1) / (32 * RT)), dim3(512), LDSB, (hipStream_t)stream>>>(a);
    DMI_CHECK_LAUNCH("gemm_nt_lnbwd (chained)");
    return DMI_OK;
  }
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_ntr_kernel<0, RT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); attr = true; }
  gemm_ntr_kernel<0, RT, 2><<<dim3((M + 32 * RT - 1) / (32 * RT)), dim3(512), LDSB, (hipStream_t)stream>>>(a);
  DMI_CHECK_LAUNCH("gemm_nt_lnbwd");
  return DMI_OK;
}

extern "C" int dmi_ln_gemm_nt(const uint16_t* X, int ldx, const uint16_t* gamma, const uint16_t* beta, float eps, const uint16_t* Bt, int ldb,
                              uint16_t* C, int ldc, int M, int N, int K, int flags, const uint16_t* bias, void* stream) {
  int rc = check_nt(X, ldx, Bt, ldb, C, ldc, M, N, K);
  if (rc) return rc;
  DMI_REQUIRE(gamma && beta, "ln_gemm_nt: null LayerNorm parameters");
  DMI_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "ln_gemm_nt: gamma / beta must be 16-byte aligned");
  DMI_REQUIRE(!(flags & DMI_GEMM_BIAS) || bias, "ln_gemm_nt: bias flag without pointer");
  if (M > 32 || N % 16 != 0 || K > 64 * SK_WAVES * SKLN_MAXSTEPS) {
    dmi_set_error("ln_gemm_nt: only the decode-step shapes (M <= 32, N %% 16 == 0, K <= %d); got M=%d N=%d K=%d", 64 * SK_WAVES * SKLN_MAXSTEPS, M, N, K);
    return DMI_ERR_UNSUPPORTED;
  }
  GemmArgs a;
  fill_nt_args(a, X, ldx, Bt, ldb, C, ldc, M, N, K);
  a.bias = bias;
  const dim3 grid(N / 16), blk(64 * SK_WAVES);
  hipStream_t st = (hipStream_t)stream;
  switch (flags) {
    case 0: gemm_ln_nt_skinny_kernel<0><<<grid, blk, 0, st>>>(a, gamma, beta, eps); break;
    case DMI_GEMM_BIAS: gemm_ln_nt_skinny_kernel<DMI_GEMM_BIAS><<<grid, blk, 0, st>>>(a, gamma, beta, eps); break;
    case DMI_GEMM_BIAS | DMI_GEMM_RELU: gemm_ln_nt_skinny_kernel<DMI_GEMM_BIAS | DMI_GEMM_RELU><<<grid, blk, 0, st>>>(a, gamma, beta, eps); break;
    default:
      dmi_set_error("ln_gemm_nt: unsupported flag combination %d", flags);
      return DMI_ERR_UNSUPPORTED;
  }
  DMI_CHECK_LAUNCH("ln_gemm_nt");
  return DMI_OK;
}

// Vocabulary projection with the softmax numerator fused into the epilogue (reference src/dalle_mtf/models.py:391-395 feeding
// :348-359): E[m, n] = bf16(exp(A[m,:] . Bt[n,:] + bias[n] - rowshift[m])), rowsum_part[n / 64][m] = the fp32 sum of those
// exponentials over columns [64 (n/64), +64).  The logits are never written: floating point keeps its relative precision over
// 2^+-127, so the exponent needs no row maximum -- no shift at all (rowshift NULL: logits within +-87, one VALU op per element
// less) or any per-row shift within ~+-60 of the row maximum (e.g. the label logit) is exact to rounding; rows that do overflow
// or vanish are detected from their sum and redone with the row maximum (dmi_softmax_finish).  The row's normaliser is the sum
// of the partials and the softmax' per-row 1/sum factor moves into the consumers of dlogits (DMI_GEMM_ROWSCALE of the
// input-gradient GEMM, the pre-scaled X operand of dmi_gemm_tn).
extern "C" int dmi_gemm_nt_softmax(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, const uint16_t* bias,
                                   const float* rowshift, uint16_t* E, int lde, float* rowsum_part, int M, int N, int K,
                                   void* stream) {
  int rc = check_nt(A, lda, Bt, ldb, E, lde, M, N, K);
  if (rc) return rc;
  DMI_REQUIRE(bias && rowsum_part, "gemm_nt_softmax: null pointer");
  GemmArgs a;
  fill_nt_args(a, A, lda, Bt, ldb, E, lde, M, N, K);
  a.bias = bias; a.rowshift = rowshift; a.rowsum_part = rowsum_part; a.dbg = g_dbg_buf;
  return launch_nt<DMI_GEMM_BIAS | GEMM_SOFTMAX>(a, 1, (hipStream_t)stream);
}
extern "C" int64_t dmi_gemm_nt_softmax_partials(int N) { return (N + 63) / 64; }

// out_bf16[i] = (sum_s slabs[s*stride + i]) * rowscale[row]   (split-K epilogue for bf16 outputs; deterministic order)
__global__ __launch_bounds__(256) void reduce_slabs_bf16_kernel(const float* __restrict__ slabs, bf16_t* __restrict__ out,
                                                                int nsplit, int64_t n8, int64_t stride, const float* __restrict__ rowscale, int N) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float v[8];
    const f32x4 a0 = ((const f32x4*)slabs)[2 * i], a1 = ((const f32x4*)slabs)[2 * i + 1];
    v[0] = a0[0]; v[1] = a0[1]; v[2] = a0[2]; v[3] = a0[3]; v[4] = a1[0]; v[5] = a1[1]; v[6] = a1[2]; v[7] = a1[3];
    for (int s = 1; s < nsplit; ++s) {
      const f32x4 b0 = ((const f32x4*)(slabs + s * stride))[2 * i], b1 = ((const f32x4*)(slabs + s * stride))[2 * i + 1];
      v[0] += b0[0]; v[1] += b0[1]; v[2] += b0[2]; v[3] += b0[3]; v[4] += b1[0]; v[5] += b1[1]; v[6] += b1[2]; v[7] += b1[3];
    }
    if (rowscale) {
      const float sc = rowscale[(i * 8) / N];   // N % 8 == 0: the 8 elements share a row
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= sc;
    }
    ((u32x4*)out)[i] = pack8(v);
  }
}

extern "C" int64_t dmi_gemm_nt_splitk_workspace_bytes(int M, int N, int nsplit) { return (int64_t)nsplit * M * N * 4 + 256; }

// C[M,N] (bf16, ldc == N) = (A . Bt^T) [* rowscale[m]] with the K range split over `nsplit` block groups (fp32 slabs in
// `workspace`, reduced deterministically).  For long-K GEMMs whose tile count does not fill whole residencies of the chip.
extern "C" int dmi_gemm_nt_splitk(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, uint16_t* C, int M, int N, int K,
                                  int nsplit, const float* rowscale, void* workspace, void* stream) {
  int rc = check_nt(A, lda, Bt, ldb, C, N, M, N, K);
  if (rc) return rc;
  DMI_REQUIRE(workspace && nsplit >= 1 && nsplit <= 16 && ((int64_t)M * N) % 8 == 0, "gemm_nt_splitk: bad arguments");
  GemmArgs a;
  fill_nt_args(a, A, lda, Bt, ldb, workspace, N, M, N, K);
  a.k_per_split = (int)((((K + nsplit - 1) / nsplit) + BK - 1) / BK * BK);
  a.slab_stride = (int64_t)M * N;
  const int ns = (K + a.k_per_split - 1) / a.k_per_split;
  hipStream_t st = (hipStream_t)stream;
  rc = launch_nt<DMI_GEMM_OUT_F32>(a, ns, st);
  if (rc) return rc;
  const int64_t n8 = (int64_t)M * N / 8;
  int64_t blocks = (n8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  reduce_slabs_bf16_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((const float*)workspace, (bf16_t*)C, ns, n8, (int64_t)M * N, rowscale, N);
  DMI_CHECK_LAUNCH("gemm_nt_splitk_reduce");
  return DMI_OK;
}

// =====================================================================================
// TN weight-gradient GEMM
// =====================================================================================
#define TN_BKM 64      // rows of m per step

static int tn_splits(int M, int I, int J) {
  // 2 blocks of 256 threads are resident per CU (64 KiB LDS each): 512 slots.  Pick the split count that fills
  // whole multiples of the residency (a 1.5x grid leaves a quarter of the chip idle in the tail) and keeps at
  // least 4 k-steps per block; fewer splits also means fewer fp32 slabs to write and reduce.
  const int tiles = ((I + 127) / 128) * ((J + 127) / 128);
  const int max_s = (M + 4 * TN_BKM - 1) / (4 * TN_BKM);
  if (tiles >= 384) return 1;
  int s = 512 / tiles;
  if (s < 1) s = 1;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s;
}
static int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
// 256x256-tile weight-gradient kernel (8 waves, one block per CU): used when both dimensions fill 256-wide tiles
// (measured in the dalle_example step, profiles/r02c_*: it wins where the 128x128 plan needs many row splits -- 512x512: 64 -> 57 us,
// 512x1536: 88 -> 79 us incl. the slab reduce -- and ties or loses from 16 tiles up, where its 64-row steps give the LDS-DMA
// no more cover than the two co-resident blocks of the 128x128 kernel have)
static bool tn8_eligible(int M, int I, int J) {
  if (g_opt_tn8 == 2) return true;
  const int tiles = ((I + 255) / 256) * ((J + 255) / 256);
  return g_opt_tn8 == 1 && I >= 256 && J >= 256 && M >= 2048 && tiles < g_opt_tn8_max_tiles;
}
static int tn8_splits(int M, int I, int J) {
  const int tiles = ((I + 255) / 256) * ((J + 255) / 256);
  const int max_s = (M + 8 * TN_BKM - 1) / (8 * TN_BKM);   // at least 8 k-steps per block
  int best = 1;
  double best_cost = 1e30;
  for (int s2 = 1; s2 <= 64 && s2 <= max_s; ++s2) {
    // makespan in units of one unsplit block: rounds of 256 resident blocks, each 1/s long, + the fp32 slab round trip
    // (write + read of s x I x J x 4 bytes at ~4 TB/s vs the block's 2 x 256 x 256 x M flops at ~4.5 TFLOP/s per CU)
    const double rounds = (double)(((int64_t)tiles * s2 + 255) / 256) / s2;
    const double slab = (s2 > 1) ? (double)s2 * I * J * 8.0 / 4.0e12 / (2.0 * 256 * 256 * (double)M / 4.5e12) : 0.0;
    const double cost = rounds + slab;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = s2; }
  }
  return best;
}

extern "C" int64_t dmi_gemm_tn_workspace_bytes(int M, int I, int J) {
  const int s = tn_splits(M, I, J);
  const int64_t slabs = (s > 1) ? (int64_t)s * I * J * 4 : 0;
  const int64_t cs = (int64_t)s * J * 4;  // bias partials
  // unsplit shapes with a ragged last residency: row-split slabs of the tail stripe, < 512 tiles x 64 KiB (+ bias rows)
  const int64_t tiles = (int64_t)((I + 127) / 128) * ((J + 127) / 128);
  const int64_t tail = (s == 1 && tiles > 512) ? (512 * 65536 + 512 * 128 * 4 * (int64_t)((I + 127) / 128)) : 0;
  int64_t base = round_up64(slabs, 256) + round_up64(cs, 256) + 256;
  if (I >= 256 && J >= 256) {   // the 256x256-tile kernel's split plan (whatever the option says now)
    const int s8 = tn8_splits(M, I, J);
    const int64_t b8 = round_up64((s8 > 1) ? (int64_t)s8 * I * J * 4 : 0, 256) + round_up64((int64_t)s8 * J * 4, 256) + 256;
    if (b8 > base) base = b8;
  }
  return base > tail + 256 ? base : tail + 256;
}

#define CONV_MAX_TAPS 16
struct ConvGeom {   // convolution geometry of the implicit-im2col kernels (dmi_conv_gemm_nt, dmi_conv_wgrad_tn)
  int H, W, C, Ho, Wo, stride, ntaps;
  int lw, lh;       // log2(Wo), log2(Ho) (weight-gradient kernel only: output dims are powers of two there)
  int dy[CONV_MAX_TAPS], dx[CONV_MAX_TAPS];
};

struct TnArgs {
  const bf16_t* X;
  const bf16_t* Y;
  float* C;
  float* bias_part;  // [nsplit][J] column sums of Y (nullable)
  const bf16_t* bias_w;  // nullable bf16 [M]: bias_part = sum_m bias_w[m] * Y[m, :] instead of the plain column sums
  int M, I, J, ldx, ldy;
  int tiles_i, tiles_j;
  int m_per_split;  // multiple of TN_BKM
  int64_t slab_stride;
  unsigned long long* dbg;  // optional per-block phase timestamps (tools/phases.py); nullptr in production
};

// [r06] A row-streamed operand of the weight-gradient kernels.  The buffer descriptor is REBASED every step (scalar adds) instead of
// accumulating the step in the per-lane 32-bit offsets: those wrap at 4 GiB and the range check (num_records, clipped to 2^31 - 1) made
// every row whose byte offset inside the operand passed 2 GiB read as ZEROS -- with 50 816 logit columns (101 632 bytes per row of dY)
// that is row 21 130: rounds 1-5 computed the vocabulary projection's weight / bias gradient at the benchmark batch (40 960 rows) from
// the first 52 % of the rows.  Invisible to every parity test, which compare with the oracle at <= 2 sequences; found in round 6 by
// asking where 32-bit offsets can end (tests/test_kernels_gpu.py::test_gemm_tn_rows_beyond_2GiB).  The per-lane offsets now stay inside
// one step (checked on the host: step bytes < 2^31).
struct TnStream {
  const char* p;      // first row of the current step (+ the tile's column origin)
  int64_t rem;        // valid bytes from p to the end of the piece's last row; <= 0: everything reads as zeros
  int64_t step;       // bytes per step
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc() const {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(rem <= 0 ? 0 : (rem > 0x7fffffff ? 0x7fffffff : rem)), 0x00020000);
  }
  __device__ __forceinline__ void next() { p += step; rem -= step; }
};
// Transposed fragment fetch through inline asm: hipcc orders the ds_read_tr16_b64 INTRINSIC behind every in-flight
// LDS-DMA (it cannot prove they do not alias) and drains vmcnt(0) before it, which serialises load and compute.
// An asm read is invisible to that bookkeeping; completion is waited for explicitly (lgkmcnt) by a statement that
// names every destination "+v", so no consumer can be scheduled above it (cdna_hip_programming.md §5.7 form ii).
struct TrFrag {
  u32x2 x0a, x0b, x1a, x1b, y0a, y0b, y1a, y1b;  // {x|y}{i}{a: rows +0..3 | b: rows +4..7}
};
__device__ __forceinline__ void tr_issue(TrFrag& f, unsigned ax0, unsigned ax1, unsigned ay0, unsigned ay1) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
      "ds_read_b64_tr_b16 %2, %9\n\t"
      "ds_read_b64_tr_b16 %3, %9 offset:1024\n\t"
      "ds_read_b64_tr_b16 %4, %10\n\t"
      "ds_read_b64_tr_b16 %5, %10 offset:1024\n\t"
      "ds_read_b64_tr_b16 %6, %11\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:1024"
      : "=&v"(f.x0a), "=&v"(f.x0b), "=&v"(f.x1a), "=&v"(f.x1b), "=&v"(f.y0a), "=&v"(f.y0b), "=&v"(f.y1a), "=&v"(f.y1b)
      : "v"(ax0), "v"(ax1), "v"(ay0), "v"(ay1)
      : "memory");
}
// wait until at most 8 LDS operations (= one younger fragment set) are outstanding: LDS data returns in order
__device__ __forceinline__ void tr_wait8(TrFrag& f) {
  asm volatile("s_waitcnt lgkmcnt(8)"
               : "+v"(f.x0a), "+v"(f.x0b), "+v"(f.x1a), "+v"(f.x1b), "+v"(f.y0a), "+v"(f.y0b), "+v"(f.y1a), "+v"(f.y1b)
               :
               : "memory");
}
__device__ __forceinline__ void tr_wait(TrFrag& f) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(f.x0a), "+v"(f.x0b), "+v"(f.x1a), "+v"(f.x1b), "+v"(f.y0a), "+v"(f.y0b), "+v"(f.y1a), "+v"(f.y1b)
               :
               : "memory");
}
__device__ __forceinline__ bf16x8 tr_cat(u32x2 a, u32x2 b) {
  u32x4 v = {a[0], a[1], b[0], b[1]};
  return __builtin_bit_cast(bf16x8, v);
}
// per-row weights of the bias-gradient MFMA (A operand rows all equal): 8 consecutive bf16 per k-substep from the stage's
// 128-B weight strip; issued ahead of the transposed fragment reads, completed by the same lgkmcnt(0)
struct WFrag {
  u32x4 w[4];
};
__device__ __forceinline__ void w_issue(WFrag& f, unsigned addr) {
  asm volatile(
      "ds_read_b128 %0, %4\n\t"
      "ds_read_b128 %1, %4 offset:32\n\t"
      "ds_read_b128 %2, %4 offset:64\n\t"
      "ds_read_b128 %3, %4 offset:96"
      : "=&v"(f.w[0]), "=&v"(f.w[1]), "=&v"(f.w[2]), "=&v"(f.w[3])
      : "v"(addr)
      : "memory");
}
__device__ __forceinline__ void w_wait_nodrain(WFrag& f) {   // the weight reads are older than every fragment read: already in
  asm volatile("" : "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3]) : : "memory");
}
__device__ __forceinline__ void w_wait(WFrag& f) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3]) : : "memory");
}
// ---- 16x16x32 fragments of the 128x128 weight-gradient tile: per 32-row k-substep a wave reads 4 x-tiles and 4 y-tiles of 16
// columns; each fragment is two transposing 8-byte reads (rows +0..3 | +4..7 of the lane group's 8 rows).  Issued in halves of
// 8 reads so that counted waits stay inside the 4-bit lgkmcnt.
struct TrHalf16 {
  u32x2 t[4][2];   // tile {0..3}{rows +0..3 | +4..7}
};
template <int OFF>
__device__ __forceinline__ void tr16_issue(TrHalf16& f, const unsigned (&ad)[4]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %2, %9 offset:%12\n\t"
      "ds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
      "ds_read_b64_tr_b16 %4, %10 offset:%12\n\t"
      "ds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
      "ds_read_b64_tr_b16 %6, %11 offset:%12\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:%13"
      : "=&v"(f.t[0][0]), "=&v"(f.t[0][1]), "=&v"(f.t[1][0]), "=&v"(f.t[1][1]), "=&v"(f.t[2][0]), "=&v"(f.t[2][1]),
        "=&v"(f.t[3][0]), "=&v"(f.t[3][1])
      : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "i"(OFF), "i"(OFF + 1024)
      : "memory");
}
// wait until at most N LDS operations are outstanding; names both halves so that no consumer is scheduled above it
template <int N>
__device__ __forceinline__ void tr16_wait(TrHalf16& x, TrHalf16& y) {
  asm volatile("s_waitcnt lgkmcnt(%16)"
               : "+v"(x.t[0][0]), "+v"(x.t[0][1]), "+v"(x.t[1][0]), "+v"(x.t[1][1]), "+v"(x.t[2][0]), "+v"(x.t[2][1]),
                 "+v"(x.t[3][0]), "+v"(x.t[3][1]), "+v"(y.t[0][0]), "+v"(y.t[0][1]), "+v"(y.t[1][0]), "+v"(y.t[1][1]),
                 "+v"(y.t[2][0]), "+v"(y.t[2][1]), "+v"(y.t[3][0]), "+v"(y.t[3][1])
               : "n"(N)
               : "memory");
}
// bias-gradient weights of one stage for the 16x16x32 form: 8 consecutive bf16 per 32-row substep at byte 64 s + 16 g
struct WFrag16 {
  u32x4 w[2];
};
__device__ __forceinline__ void w16_issue(WFrag16& f, unsigned addr) {
  asm volatile(
      "ds_read_b128 %0, %2\n\t"
      "ds_read_b128 %1, %2 offset:64"
      : "=&v"(f.w[0]), "=&v"(f.w[1])
      : "v"(addr)
      : "memory");
}
__device__ __forceinline__ void w16_keep(WFrag16& f) {   // the weight reads are older than every fragment read: already in
  asm volatile("" : "+v"(f.w[0]), "+v"(f.w[1]) : : "memory");
}
#define TN_LDS_BYTES (65536 + 512)   // two 32-KiB operand stages + two 256-B bias-weight strips

// v3: LDS-DMA staging (buffer_load_dwordx4 ... lds; rows past the split / matrix end read as 0 through the
// descriptor's num_records), two stages x (X 64x128 | Y 64x128) bf16 = 64 KiB, ONE barrier per 64-row step.
// LDS rows are 256 B (linear, as the DMA requires); the 16-B chunk index is XOR-ed with 4*(row&3) on the SOURCE
// side so that the 2x32-lane service groups of ds_read_b64_tr_b16 (4 rows x 2 x 32 B) cover all 64 banks once.
// row&3 is a per-lane constant of the fragment read ((l16>>2)), so the swizzle folds into 2 hoisted offsets.
// Bias gradients (column sums of dY): blocks of the first row-tile issue one extra MFMA per k-step with an
// all-ones A operand (D[i][j] = sum_k Y[k][j]).
// One (tile, row-range) unit of work: C[i0.., j0..] (row pitch ldc) = X[mb..mb+rows, i0..]^T dY[mb..mb+rows, j0..];
// bias_out[j0..] = column sums of that dY range (blocks of the first row-tile only).
// CONV: X is the virtual im2col matrix of an NHWC activation tensor (a.X = tensor base, a.ldx = its size in bytes, cg = the
// geometry): a lane's column chunk belongs to one tap for the whole tile, its rows are output pixels decoded with shifts.
template <bool CONV>
__device__ __forceinline__ void tn_tile(const TnArgs& a, const ConvGeom* cg, char* smem_tn, int ti, int tj, int mb, int rows,
                                        float* C, int64_t ldc, float* bias_out) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wid >> 1, wj = wid & 1;
  const int g4 = lane >> 4, l16 = lane & 15;

  const int i0 = ti * 128, j0 = tj * 128;
  const int nt = (rows + TN_BKM - 1) / TN_BKM;

  const int wx = (a.I - i0 < 128) ? a.I - i0 : 128, wy = (a.J - j0 < 128) ? a.J - j0 : 128;
  const int64_t nbx = rows > 0 ? ((int64_t)(rows - 1) * a.ldx + wx) * 2 : 0;
  const int64_t nby = rows > 0 ? ((int64_t)(rows - 1) * a.ldy + wy) * 2 : 0;
  const __amdgpu_buffer_rsrc_t rx_conv = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, CONV ? a.ldx : 0, 0x00020000);   // CONV: the whole tensor (< 2 GiB, host check), offsets computed per row
  TnStream sx = {(const char*)(a.X + (CONV ? 0 : (int64_t)mb * a.ldx + i0)), nbx, (int64_t)TN_BKM * a.ldx * 2};
  TnStream sy = {(const char*)(a.Y + (int64_t)mb * a.ldy + j0), nby, (int64_t)TN_BKM * a.ldy * 2};
  // DMA: linear LDS chunk c = tid + 256 i -> row c>>4, physical chunk c&15 holds source chunk (c&15) ^ 4*(row&3) ^ 2*((row>>3)&1)
  int vox[4], voy[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 4) + 16 * i;
    const int sch = (tid & 15) ^ (4 * (row & 3)) ^ (2 * ((row >> 3) & 1));
    vox[i] = (8 * sch < wx) ? (row * a.ldx + 8 * sch) * 2 : 0x7ffffff0;  // columns past the width read as 0
    voy[i] = (8 * sch < wy) ? (row * a.ldy + 8 * sch) * 2 : 0x7ffffff0;
  }
  // CONV: per-lane constants (the source chunk, hence the tap and channel, do not depend on i: 16 i keeps row & 3 and row bit 3)
  int cv_dy = 0, cv_dx = 0, cv_coff = 0, cv_m = 0;
  bool cv_colok = false;
  if constexpr (CONV) {
    const int row0 = tid >> 4;
    const int sch = (tid & 15) ^ (4 * (row0 & 3)) ^ (2 * ((row0 >> 3) & 1));
    const int kcol = i0 + 8 * sch;
    cv_colok = 8 * sch < wx;
    const int tap = cv_colok ? kcol / cg->C : 0;
    cv_dy = cg->dy[tap];
    cv_dx = cg->dx[tap];
    cv_coff = (kcol - tap * cg->C) * 2;
    cv_m = mb + row0;   // output pixel of this lane's first row in the next stage
  }
  // fragment read offsets (bytes) for v_mfma_f32_16x16x32_bf16: lane group g4 owns the substep's rows 8 g4 + (l16>>2) [+4]; byte in
  // row = (wave base + tile*32 + 8*(l16&3)) ^ 64*(row&3) ^ 32*((row>>3)&1) -- the two lane groups of a 32-lane service group read
  // rows 8 apart (same bank row), so the second XOR term sends them to different 32-B halves: all 64 banks once per service group
  const int rr = l16 >> 2;
  const int rowb = (8 * g4 + rr) * 256;
  const int cb = 8 * (l16 & 3);
  const int xm = (64 * rr) ^ (32 * (g4 & 1));
  int ofx[4], ofy[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    ofx[t] = rowb + ((wi * 128 + t * 32 + cb) ^ xm);
    ofy[t] = 16384 + rowb + ((wj * 128 + t * 32 + cb) ^ xm);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  // Bias gradient = (weighted) column sums of dY: blocks of the first row-tile issue two extra MFMAs per k-substep whose A operand
  // is all ones (plain sums) or, with a.bias_w, the per-row weights broadcast over the 16 output rows: D[i][j] = sum_m w[m] Y[m][j].
  // (Spreading that work over all row-tile blocks was measured: mixed, slower in the step.)
  const bool do_bias = (bias_out != nullptr) && (ti == 0);
  const bool use_w = do_bias && (a.bias_w != nullptr);
  const int64_t nbw = rows > 0 ? (int64_t)rows * 2 : 0;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(use_w ? a.bias_w + mb : a.Y), 0, (int)(use_w ? nbw : 0), 0x00020000);
  int vow = lane * 4;   // 2 weights per lane: 64 lanes cover 128 rows, the step uses the first 64
  const u32x4 ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);

  auto stage = [&](int st) {
    char* base = smem_tn + st * 32768 + wid * 1024;
    const __amdgpu_buffer_rsrc_t rx = CONV ? rx_conv : sx.rsrc(), ry = sy.rsrc();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (CONV) {
        const int m = cv_m + 16 * i;
        const int ox = m & (cg->Wo - 1), oy = (m >> cg->lw) & (cg->Ho - 1), b = m >> (cg->lw + cg->lh);
        const int iy = oy * cg->stride + cv_dy, ix = ox * cg->stride + cv_dx;
        const bool ok = cv_colok && (m < mb + rows) && ((unsigned)iy < (unsigned)cg->H) && ((unsigned)ix < (unsigned)cg->W);
        const int vo = ok ? ((b * cg->H + iy) * cg->W + ix) * cg->C * 2 + cv_coff : 0x7ffffff0;
        glds16(rx, base + i * 4096, vo, 0);
      } else {
        glds16(rx, base + i * 4096, vox[i], 0);
      }
      glds16(ry, base + 16384 + i * 4096, voy[i], 0);
    }
    if constexpr (!CONV) sx.next();
    sy.next();
    if constexpr (CONV) cv_m += TN_BKM;
    if (use_w) {   // block-uniform; every wave writes the same 256 bytes (uniform DMA count per wave)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(smem_tn + 65536 + st * 256), 4, vow, 0, 0, 0);
      vow += TN_BKM * 2;
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_tn;
  auto bf = [](const u32x2 (&p)[2]) { return tr_cat(p[0], p[1]); };
  auto mfmas = [&](TrHalf16& X, TrHalf16& Y, const bf16x8 wa) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(X.t[i]), bf(Y.t[j]), acc[i][j], 0, 0, 0);   // D[i][j]: lane (c, g) holds rows 4g.., column c
    if (do_bias) {  // wave (wi, wj) sums the columns of y tiles 2 wi, 2 wi + 1  (wave-uniform branches)
      if (wi == 0) {
        bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bf(Y.t[0]), bacc[0], 0, 0, 0);
        bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bf(Y.t[1]), bacc[1], 0, 0, 0);
      } else {
        bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bf(Y.t[2]), bacc[0], 0, 0, 0);
        bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bf(Y.t[3]), bacc[1], 0, 0, 0);
      }
    }
  };
  auto compute = [&](int st) {
    const unsigned base = lds0 + st * 32768;
    unsigned ax[4], ay[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { ax[t] = base + ofx[t]; ay[t] = base + ofy[t]; }
    WFrag16 wf;
    if (use_w) w16_issue(wf, lds0 + 65536 + st * 256 + 16 * g4);
    // two 32-row substeps; the second substep's fragments are read under the first substep's MFMAs (counted lgkmcnt: the reads
    // return in order, at most 8 of the younger half may still be in flight when the older set is consumed)
    TrHalf16 X0, Y0, X1, Y1;
    MFMA_PRIO(1);
    tr16_issue<0>(X0, ax);
    tr16_issue<0>(Y0, ay);
    tr16_issue<8192>(X1, ax);
    tr16_wait<8>(X0, Y0);
    if (use_w) w16_keep(wf);
    tr16_issue<8192>(Y1, ay);
    mfmas(X0, Y0, use_w ? __builtin_bit_cast(bf16x8, wf.w[0]) : ones);
    tr16_wait<0>(X1, Y1);
    mfmas(X1, Y1, use_w ? __builtin_bit_cast(bf16x8, wf.w[1]) : ones);
    MFMA_PRIO(0);
  };

  unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, rq0 = 0;
  if (a.dbg) { tq0 = __builtin_readcyclecounter(); rq0 = __builtin_amdgcn_s_memrealtime(); }
  if (nt > 0) {
    // stages addressed with compile-time constants (x2 unroll): otherwise the compiler cannot prove the in-flight
    // LDS-DMA does not alias the fragment reads and drains vmcnt(0) before them.
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (a.dbg) tq1 = __builtin_readcyclecounter();
    int t = 0;
    for (; t + 2 <= nt; t += 2) {
      stage(1);
      compute(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + 2 < nt) stage(0);
      compute(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (t < nt) {
      compute(0);
      __syncthreads();
    }
  }
  if (a.dbg) tq2 = __builtin_readcyclecounter();
  // store: tile (i, j), reg e -> row 16 i + 4 g + e ; column 16 j + c
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = j0 + wj * 64 + j * 16 + l16;
      if (col >= a.J) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = i0 + wi * 64 + i * 16 + 4 * g4 + e;
        if (row < a.I) C[(int64_t)row * ldc + col] = acc[i][j][e];
      }
    }
  if (do_bias && g4 == 0) {  // row 0 of D (reg 0 of lane group 0) holds the column sums
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int col = j0 + wj * 64 + (2 * wi + q) * 16 + l16;
      if (col < a.J) bias_out[col] = bacc[q][0];
    }
  }
  if (a.dbg && tid == 0) {
    unsigned long long* dq = a.dbg + (size_t)blockIdx.x * 8;   // 100 MHz realtime stamps calibrate the cycle counter
    dq[0] = tq0; dq[1] = tq1; dq[2] = tq2; dq[3] = __builtin_readcyclecounter(); dq[4] = (unsigned long long)nt;
    dq[5] = rq0; dq[6] = __builtin_amdgcn_s_memrealtime();
  }
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_tn[];  // [2 stages][X 16K | Y 16K]
  // 1-D grid of tiles x splits, split-major, cut into one contiguous range per XCD: an XCD then works on whole row
  // ranges (splits), whose X / dY slabs are fetched over the fabric once and shared by all of that split's tiles through
  // the XCD's L2.  (With the split on gridDim.y every XCD held 1/8 of the tiles of EVERY split: each X tile crossed the
  // fabric once per column tile -- 756 MB instead of 210 MB for the 2048 x 512 gradient.)
  const int tiles = a.tiles_i * a.tiles_j;
  const int P = xcd_remap(blockIdx.x, gridDim.x);
  const int split = P / tiles;
  int ti, tj;
  tile_of_block(P - split * tiles, a.tiles_i, a.tiles_j, ti, tj);
  const int mb = split * a.m_per_split;
  const int me = (mb + a.m_per_split < a.M) ? mb + a.m_per_split : a.M;
  const int rows = me > mb ? me - mb : 0;
  tn_tile<false>(a, nullptr, smem_tn, ti, tj, mb, rows, a.C + (int64_t)split * a.slab_stride, a.J,
          a.bias_part ? a.bias_part + (int64_t)split * a.J : nullptr);
}

// ---- several weight gradients in ONE launch (round 5) ------------------------------------------------------------------------
// The 512 x 512 out-projection gradient has 16 tiles: alone it needs 32 row splits to fill the 512 block slots (20 k-steps per
// block: prologue / epilogue-bound, 458 TFLOP/s) and the 512 x 1536 QKV gradient (48 tiles) fills 480 of them.  Together they are
// 64 tiles x 8 splits = 512 blocks of 80 k-steps -- the shape of an FFN gradient (960 TFLOP/s).  Problems share M and the split
// plan; a block picks its problem from the tile index (scalar selects), everything else is tn_tile.  Split-major numbering over
// the UNION of the tiles, one contiguous range per XCD, as gemm_tn_kernel.
#define TN_GROUP_MAX 4
struct TnGroup {
  TnArgs p[TN_GROUP_MAX];
  int first_tile[TN_GROUP_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256, 2) void gemm_tn_group_kernel(TnGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem_tn[];
  const int T = g.first_tile[g.n];
  const int P = xcd_remap(blockIdx.x, gridDim.x);
  const int split = P / T, t = P - split * T;
  TnArgs a = g.p[0];
  int t0 = 0;
  if (g.n > 1 && t >= g.first_tile[1]) { a = g.p[1]; t0 = g.first_tile[1]; }
  if (g.n > 2 && t >= g.first_tile[2]) { a = g.p[2]; t0 = g.first_tile[2]; }
  if (g.n > 3 && t >= g.first_tile[3]) { a = g.p[3]; t0 = g.first_tile[3]; }
  int ti, tj;
  tile_of_block(t - t0, a.tiles_i, a.tiles_j, ti, tj);
  const int mb = split * a.m_per_split;
  const int me = (mb + a.m_per_split < a.M) ? mb + a.m_per_split : a.M;
  const int rows = me > mb ? me - mb : 0;
  tn_tile<false>(a, nullptr, smem_tn, ti, tj, mb, rows, a.C + (int64_t)split * a.slab_stride, a.J,
                 a.bias_part ? a.bias_part + (int64_t)split * a.J : nullptr);
}

// ---- [r06] wide weight-gradient tile (128 (I) x 256 (J), 4 waves of 64 x 128, 32-row steps) as a gang stream-K --------------------
// The 128x128 tile moves 32 KB through the LDS-DMA path per 2.1-MFLOP block step and two co-resident blocks run that path at the
// ~32-36 B/clk/CU it delivers (the head's gradient: 2047 clk per block step for 515 clk of MFMA issue, twice per CU).  This tile
// moves 24 KB for the same 2.1 MFLOP (32 rows x (128 + 256) columns), reads 24 transposed fragments per 32 MFMAs instead of 32, and
// keeps everything else: two blocks per CU (2 x 48 KB LDS, 128 accumulator registers), one barrier per 32 MFMAs per wave, the same
// source-side swizzle (per 256-byte segment of a row), the same bias path and the same k order (32-row chunks in row order: a whole
// stripe has the 128x128 kernel's bits).  Measured where whole tiles fill whole residencies (M = 40960, I = 512, J = 65536): 2880 ->
// 2397 us = 955 -> 1147 TFLOP/s (with every row read: see TnStream); the transposed shape (256 x 128) 5 % behind it; on the row-split gradients of the layers (32 tiles x 16
// splits instead of 64 x 8) slower (88 -> 97 us: twice the slabs, half as long blocks) -- those stay on the 128x128 tile
// (profiles/r06_tn_wide.log).
#define TNW_BKM 32
#define TNW_TI 128
#define TNW_TJ 256
#define TNW_XB (TNW_BKM * TNW_TI * 2)     // bytes of a stage's X image (row pitch 256 B)
#define TNW_YB (TNW_BKM * TNW_TJ * 2)     // ... Y image (row pitch 512 B)
#define TNW_STB (TNW_XB + TNW_YB)
#define TNW_LDS_BYTES (2 * TNW_STB + 512)  // two stages + two 256-B bias-weight strips
template <int OFFA, int OFFB>
__device__ __forceinline__ void trw_issue(TrHalf16& f, const unsigned (&ad)[4]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %2, %9 offset:%12\n\t"
      "ds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
      "ds_read_b64_tr_b16 %4, %10 offset:%12\n\t"
      "ds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
      "ds_read_b64_tr_b16 %6, %11 offset:%12\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:%13"
      : "=&v"(f.t[0][0]), "=&v"(f.t[0][1]), "=&v"(f.t[1][0]), "=&v"(f.t[1][1]), "=&v"(f.t[2][0]), "=&v"(f.t[2][1]),
        "=&v"(f.t[3][0]), "=&v"(f.t[3][1])
      : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "i"(OFFA), "i"(OFFB)
      : "memory");
}
template <int N>
__device__ __forceinline__ void trw_wait(TrHalf16& x, TrHalf16& y, TrHalf16& z) {
  asm volatile("s_waitcnt lgkmcnt(%24)"
               : "+v"(x.t[0][0]), "+v"(x.t[0][1]), "+v"(x.t[1][0]), "+v"(x.t[1][1]), "+v"(x.t[2][0]), "+v"(x.t[2][1]),
                 "+v"(x.t[3][0]), "+v"(x.t[3][1]), "+v"(y.t[0][0]), "+v"(y.t[0][1]), "+v"(y.t[1][0]), "+v"(y.t[1][1]),
                 "+v"(y.t[2][0]), "+v"(y.t[2][1]), "+v"(y.t[3][0]), "+v"(y.t[3][1]), "+v"(z.t[0][0]), "+v"(z.t[0][1]),
                 "+v"(z.t[1][0]), "+v"(z.t[1][1]), "+v"(z.t[2][0]), "+v"(z.t[2][1]), "+v"(z.t[3][0]), "+v"(z.t[3][1])
               : "n"(N)
               : "memory");
}
// One (row tile ti, column stripe tj, row range [mb, mb + rows)) piece: C[128 ti.., 256 tj..] (=|+=) X[rows, 128 ti..]^T dY[rows, 256 tj..],
// bias_out[256 tj..] (=|+=) the (weighted) column sums of that dY range (blocks of the first row tile only).  atomic: the piece is one of
// exactly TWO addends onto a zeroed element -- a two-addend fp32 sum does not depend on the order.
__device__ __forceinline__ void tn_tile_wide(const TnArgs& a, char* smem_tn, int ti, int tj, int mb, int rows, float* C, int64_t ldc,
                                             float* bias_out, bool atomic) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));   // opaque per call: the caller's piece loop must not keep this function's per-lane constants alive across pieces
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wid >> 1, wj = wid & 1;
  const int g4 = lane >> 4, l16 = lane & 15;
  const int i0 = ti * TNW_TI, j0 = tj * TNW_TJ;
  const int nt = (rows + TNW_BKM - 1) / TNW_BKM;
  const int wx = (a.I - i0 < TNW_TI) ? a.I - i0 : TNW_TI, wy = (a.J - j0 < TNW_TJ) ? a.J - j0 : TNW_TJ;
  const int64_t nbx = rows > 0 ? ((int64_t)(rows - 1) * a.ldx + wx) * 2 : 0;
  const int64_t nby = rows > 0 ? ((int64_t)(rows - 1) * a.ldy + wy) * 2 : 0;
  TnStream sx = {(const char*)(a.X + (int64_t)mb * a.ldx + i0), nbx, (int64_t)TNW_BKM * a.ldx * 2};   // rebased every step: see TnStream
  TnStream sy = {(const char*)(a.Y + (int64_t)mb * a.ldy + j0), nby, (int64_t)TNW_BKM * a.ldy * 2};
  // DMA: linear LDS chunk c = tid + 256 i -> row c / (pitch / 16), physical chunk pc; its 256-byte segment pc >> 4 holds source chunk
  // (pc & 15) ^ 4 (row & 3) ^ 2 ((row >> 3) & 1) of that segment (tn_tile's swizzle)
  int vox[2], voy[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i, row = c >> 4, pc = c & 15;
    const int col = 8 * (pc ^ (4 * (row & 3)) ^ (2 * ((row >> 3) & 1)));
    vox[i] = (col < wx) ? (row * a.ldx + col) * 2 : 0x7ffffff0;   // columns past the width read as 0
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 5, pc = c & 31;
    const int col = (pc >> 4) * 128 + 8 * ((pc & 15) ^ (4 * (row & 3)) ^ (2 * ((row >> 3) & 1)));
    voy[i] = (col < wy) ? (row * a.ldy + col) * 2 : 0x7ffffff0;
  }
  // fragment offsets: lane group g4 owns rows 8 g4 + (l16 >> 2) [+ 4]; byte in row as in tn_tile (the XOR stays inside a 256-byte segment)
  const int rr = l16 >> 2;
  const int cb = 8 * (l16 & 3);
  const int xm = (64 * rr) ^ (32 * (g4 & 1));
  int ofx[4], ofy[8];
#pragma unroll
  for (int t = 0; t < 4; ++t) ofx[t] = (8 * g4 + rr) * 256 + ((wi * 128 + t * 32 + cb) ^ xm);
#pragma unroll
  for (int t = 0; t < 8; ++t) ofy[t] = TNW_XB + (8 * g4 + rr) * 512 + ((wj * 256 + t * 32 + cb) ^ xm);

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bacc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bacc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient: wave (wi, wj) of a first-row-tile block sums the columns of its y tiles 4 wi .. 4 wi + 3 (A operand all ones, or the
  // per-row weights broadcast over the 16 output rows), as in tn_tile
  const bool do_bias = (bias_out != nullptr) && (ti == 0);
  const bool use_w = do_bias && (a.bias_w != nullptr);
  const int64_t nbw = rows > 0 ? (int64_t)rows * 2 : 0;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(use_w ? a.bias_w + mb : a.Y), 0, (int)(use_w ? nbw : 0), 0x00020000);
  int vow = lane * 4;   // 2 weights per lane: 64 lanes cover 128 rows, the step uses the first 32
  const u32x4 ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

  auto stage = [&](int st) {
    char* base = smem_tn + st * TNW_STB + wid * 1024;
    const __amdgpu_buffer_rsrc_t rx = sx.rsrc(), ry = sy.rsrc();
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(rx, base + i * 4096, vox[i], 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(ry, base + TNW_XB + i * 4096, voy[i], 0);
    sx.next(); sy.next();
    if (use_w) {   // block-uniform; every wave writes the same 256 bytes (uniform DMA count per wave)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(smem_tn + 2 * TNW_STB + st * 256), 4, vow, 0, 0, 0);
      vow += TNW_BKM * 2;
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_tn;
  auto bf = [](const u32x2 (&p)[2]) { return tr_cat(p[0], p[1]); };
  auto compute = [&](int st) {   // X once, Y in two halves of four tiles: the second half's reads land under the first half's MFMAs
    const unsigned base = lds0 + st * TNW_STB;
    u32x4 wv = ones_u;
    if (use_w) asm volatile("ds_read_b128 %0, %1" : "=&v"(wv) : "v"(lds0 + 2 * TNW_STB + st * 256 + 16 * g4) : "memory");
    MFMA_PRIO(1);
    unsigned ax[4], ay0[4], ay1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { ax[t] = base + ofx[t]; ay0[t] = base + ofy[t]; ay1[t] = base + ofy[4 + t]; }
    TrHalf16 X, Y0, Y1;
    trw_issue<0, 1024>(X, ax);
    trw_issue<0, 2048>(Y0, ay0);
    trw_issue<0, 2048>(Y1, ay1);
    trw_wait<8>(X, Y0, Y1);
    if (use_w) asm volatile("" : "+v"(wv) : : "memory");   // older than every fragment read: already in
    const bf16x8 wa = __builtin_bit_cast(bf16x8, wv);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(X.t[i]), bf(Y0.t[j]), acc[i][j], 0, 0, 0);
    if (do_bias && wi == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bacc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bf(Y0.t[q]), bacc[q], 0, 0, 0);
    }
    trw_wait<0>(X, Y0, Y1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(X.t[i]), bf(Y1.t[j]), acc[i][4 + j], 0, 0, 0);
    if (do_bias && wi == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bacc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bf(Y1.t[q]), bacc[q], 0, 0, 0);
    }
    MFMA_PRIO(0);
  };

  if (nt > 0) {   // stages addressed with compile-time constants (x2 unroll), as in tn_tile
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int t = 0;
    for (; t + 2 <= nt; t += 2) {
      stage(1);
      compute(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + 2 < nt) stage(0);
      compute(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (t < nt) {
      compute(0);
      __syncthreads();
    }
  }
  // store: tile (i, j), reg e -> row 16 i + 4 g + e ; column 16 j + c
  // (Whole pieces staged through wave-private LDS strips and stored as 16-byte pieces of 512-byte row segments -- 32 store instructions per
  // wave instead of 128 -- measured NEUTRAL: 1727 vs 1728 us on the head (before the TnStream fix), profiles/r06_tn_wide.log.  Removed.)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = j0 + wj * 128 + j * 16 + l16;
      if (col >= a.J) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = i0 + wi * 64 + i * 16 + 4 * g4 + e;
        if (row < a.I) {
          if (atomic) unsafeAtomicAdd(&C[(int64_t)row * ldc + col], acc[i][j][e]);
          else C[(int64_t)row * ldc + col] = acc[i][j][e];
        }
      }
    }
  if (do_bias && g4 == 0) {  // row 0 of D (reg 0 of lane group 0) holds the column sums
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = j0 + wj * 128 + (4 * wi + q) * 16 + l16;
      if (col < a.J) {
        if (atomic) unsafeAtomicAdd(&bias_out[col], bacc[q][0]);
        else bias_out[col] = bacc[q][0];
      }
    }
  }
}
// Unsplit gradients with many column stripes (the head: I = 512, J = 50816 -> 4 x 199 wide tiles on 512 block slots = 1.55 residencies,
// which whole tiles can only run as 2): the block slots form gangs of tiles_i blocks, a gang walks its share of the flattened
// (stripe, 32-row step) space -- every gang the same number of steps -- and its blocks do the same (stripe, row range) piece for their own
// row tile AT THE SAME TIME, so a dY stripe still crosses the fabric once and is shared through the XCD's L2 (what a per-block stream-K
// cut loses, see gemm_tn_tail_kernel).  A gang's share is at least one stripe long, so a stripe is cut at most ONCE: its two pieces are
// added onto zeroed elements with fp32 atomics, and a two-addend sum does not depend on the order -- deterministic without slabs
// (whole stripes are stored, with the 128x128 kernel's bits).
// stripe_is_cut: does a gang boundary fall strictly inside stripe s?  (device and host use this one expression)
__host__ __device__ __forceinline__ int tnw_gang_begin(int64_t U, int gang, int gangs) { return (int)(U * gang / gangs); }
__global__ __launch_bounds__(256) void tn_wide_zero_kernel(float* __restrict__ dW, float* __restrict__ dbias, int I, int J, int tiles_j,
                                                           int gangs, int nsteps) {
  const int s = blockIdx.x;
  const int64_t U = (int64_t)tiles_j * nsteps;
  const int lo = s * nsteps, hi = lo + nsteps;
  int g = (int)((int64_t)lo * gangs / U);   // tnw_gang_begin(g) <= lo; the first boundary above lo is at g + 1 or g + 2
  bool cut = false;
  for (int k = g; k <= g + 2 && k < gangs; ++k) {
    const int b = tnw_gang_begin(U, k, gangs);
    cut = cut || (b > lo && b < hi);
  }
  if (!cut) return;
  const int c0 = s * TNW_TJ, w4 = ((J - c0 < TNW_TJ) ? J - c0 : TNW_TJ) / 4;   // J % 4 == 0 (host check)
  for (int i = threadIdx.x; i < I * w4; i += 256) {
    const int r = i / w4, c = i - r * w4;
    *(f32x4*)(dW + (int64_t)r * J + c0 + 4 * c) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (dbias) for (int i = threadIdx.x; i < w4; i += 256) *(f32x4*)(dbias + c0 + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
}
__global__ __launch_bounds__(256, 2) void gemm_tn_wide_sk_kernel(TnArgs a, int gangs, int nsteps) {
  extern __shared__ __attribute__((aligned(16))) char smem_tn[];
  const int P = xcd_remap(blockIdx.x, gridDim.x);   // gangs % 8 == 0: a gang's blocks are neighbours on one XCD
  const int gang = P / a.tiles_i, ti = P - gang * a.tiles_i;
  const int64_t U = (int64_t)a.tiles_j * nsteps;     // < 2^31 (host check)
  const int g0 = tnw_gang_begin(U, gang, gangs), g1 = tnw_gang_begin(U, gang + 1, gangs);
  // The gang walks its range ROTATED: from the first stripe boundary inside it to its end, then the leading partial stripe.  A range is
  // (tail of a stripe | whole stripes | head of a stripe) and all ranges are equally long, so every gang works on row step t of some
  // stripe at time t (phase 0) until its leading tail, which all gangs reach at the same time and walk with ONE common phase: at any
  // moment the XCD's gangs read two row ranges of X, not sixteen (in range order each gang has its own phase: 1844 vs 1730 us, measured before the TnStream fix).
  int ub = ((g0 + nsteps - 1) / nsteps) * nsteps;
  if (ub > g1) ub = g1;
  for (int part = 0; part < 2; ++part) {
    int u0 = part == 0 ? ub : g0;
    const int u1 = part == 0 ? g1 : ub;
    while (u0 < u1) {
      const int stripe = u0 / nsteps, st0 = u0 - stripe * nsteps;
      int ue = (stripe + 1) * nsteps;
      if (ue > u1) ue = u1;
      const int nst = ue - u0;
      const int mb = st0 * TNW_BKM;
      const int rows = (mb + nst * TNW_BKM < a.M) ? nst * TNW_BKM : a.M - mb;
      tn_tile_wide(a, smem_tn, ti, stripe, mb, rows, a.C, a.J, a.bias_part, nst < nsteps);
      u0 = ue;
    }
  }
}

// several row-split gradients in one launch (gemm_tn_group_kernel) on the wide tile: where the union of the problems' 128 x 256 tiles
// fills the block slots with no more row splits than each problem's own workspace holds slabs for (the two FFN gradients of a block:
// 32 + 32 tiles x 8 splits -- alone, either needs 16 splits, twice the slabs and half as long blocks, and loses to the 128x128 tile)
__global__ __launch_bounds__(256, 2) void gemm_tn_group_wide_kernel(TnGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem_tn[];
  const int T = g.first_tile[g.n];
  const int P = xcd_remap(blockIdx.x, gridDim.x);
  const int split = P / T, t = P - split * T;
  TnArgs a = g.p[0];
  int t0 = 0;
  if (g.n > 1 && t >= g.first_tile[1]) { a = g.p[1]; t0 = g.first_tile[1]; }
  if (g.n > 2 && t >= g.first_tile[2]) { a = g.p[2]; t0 = g.first_tile[2]; }
  if (g.n > 3 && t >= g.first_tile[3]) { a = g.p[3]; t0 = g.first_tile[3]; }
  int ti, tj;
  tile_of_block(t - t0, a.tiles_i, a.tiles_j, ti, tj);
  const int mb = split * a.m_per_split;
  const int me = (mb + a.m_per_split < a.M) ? mb + a.m_per_split : a.M;
  const int rows = me > mb ? me - mb : 0;
  tn_tile_wide(a, smem_tn, ti, tj, mb, rows, a.C + (int64_t)split * a.slab_stride, a.J,
               a.bias_part ? a.bias_part + (int64_t)split * a.J : nullptr, false);
}

// ---- 256x256-tile weight gradient (8 waves) ----------------------------------------------------------------------------
// Same data path as tn_tile (natural-layout LDS tiles by LDS-DMA, hardware transpose reads), block tile 256 (I) x 256 (J),
// 8 waves as 2 (I) x 4 (J) of 128 x 64 = 4 x 2 MFMA tiles; two stages x (X 64 x 256 | Y 64 x 256) bf16 = 128 KiB -> one
// block per CU, two waves per SIMD; 0.75 transpose reads per MFMA instead of 1 and half the L2->LDS bytes per flop of the
// 128x128 tile (measured on the NT twin: +20 % on main-loop-bound shapes).  LDS rows are 512 B; 16-B chunk index XOR
// 4*(row&3) on the source side as in tn_tile.  Fragment sets are double-buffered across the four 16-row k-substeps of a
// stage (counted lgkmcnt), not pre-read for all four (they would need 96 VGPRs).  A 4-stage ring of 32-row stages with
// counted vmcnt (loads three stages ahead) was measured equal to this 2-stage loop on every shape of the model.
struct TrFrag8 {
  u32x2 x[4][2], y[2][2];   // {x i | y j}{rows +0..3 | +4..7}
};
template <int OFF>
__device__ __forceinline__ void tr8_issue(TrFrag8& f, const unsigned (&ax)[4], const unsigned (&ay)[2]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %12 offset:%18\n\t"
      "ds_read_b64_tr_b16 %1, %12 offset:%19\n\t"
      "ds_read_b64_tr_b16 %8, %16 offset:%18\n\t"
      "ds_read_b64_tr_b16 %9, %16 offset:%19\n\t"
      "ds_read_b64_tr_b16 %10, %17 offset:%18\n\t"
      "ds_read_b64_tr_b16 %11, %17 offset:%19\n\t"
      "ds_read_b64_tr_b16 %2, %13 offset:%18\n\t"
      "ds_read_b64_tr_b16 %3, %13 offset:%19\n\t"
      "ds_read_b64_tr_b16 %4, %14 offset:%18\n\t"
      "ds_read_b64_tr_b16 %5, %14 offset:%19\n\t"
      "ds_read_b64_tr_b16 %6, %15 offset:%18\n\t"
      "ds_read_b64_tr_b16 %7, %15 offset:%19"
      : "=&v"(f.x[0][0]), "=&v"(f.x[0][1]), "=&v"(f.x[1][0]), "=&v"(f.x[1][1]), "=&v"(f.x[2][0]), "=&v"(f.x[2][1]),
        "=&v"(f.x[3][0]), "=&v"(f.x[3][1]), "=&v"(f.y[0][0]), "=&v"(f.y[0][1]), "=&v"(f.y[1][0]), "=&v"(f.y[1][1])
      : "v"(ax[0]), "v"(ax[1]), "v"(ax[2]), "v"(ax[3]), "v"(ay[0]), "v"(ay[1]), "i"(OFF), "i"(OFF + 2048)
      : "memory");
}
template <int N>
__device__ __forceinline__ void tr8_wait(TrFrag8& f) {
  asm volatile("s_waitcnt lgkmcnt(%12)"
               : "+v"(f.x[0][0]), "+v"(f.x[0][1]), "+v"(f.x[1][0]), "+v"(f.x[1][1]), "+v"(f.x[2][0]), "+v"(f.x[2][1]),
                 "+v"(f.x[3][0]), "+v"(f.x[3][1]), "+v"(f.y[0][0]), "+v"(f.y[0][1]), "+v"(f.y[1][0]), "+v"(f.y[1][1])
               : "n"(N)
               : "memory");
}
#define TN8_LDS_BYTES (131072 + 512)
__global__ __launch_bounds__(512, 2) void gemm_tn8_kernel(TnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_tn[];  // [2 stages][X 32K | Y 32K] | 2 x 256 B weight strips
  const int tiles = a.tiles_i * a.tiles_j;
  const int P = xcd_remap(blockIdx.x, gridDim.x);   // split-major: an XCD works on whole row ranges (see gemm_tn_kernel)
  const int split = P / tiles;
  int ti, tj;
  tile_of_block(P - split * tiles, a.tiles_i, a.tiles_j, ti, tj);
  const int mb = split * a.m_per_split;
  const int me = (mb + a.m_per_split < a.M) ? mb + a.m_per_split : a.M;
  const int rows = me > mb ? me - mb : 0;
  float* C = a.C + (int64_t)split * a.slab_stride;
  float* bias_out = a.bias_part ? a.bias_part + (int64_t)split * a.J : nullptr;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wid >> 2, wj = wid & 3;
  const int h = lane >> 5, g4 = lane >> 4, l16 = lane & 15;
  const int i0 = ti * 256, j0 = tj * 256;
  const int nt = (rows + TN_BKM - 1) / TN_BKM;
  const int wx = (a.I - i0 < 256) ? a.I - i0 : 256, wy = (a.J - j0 < 256) ? a.J - j0 : 256;
  const int64_t nbx = rows > 0 ? ((int64_t)(rows - 1) * a.ldx + wx) * 2 : 0;
  const int64_t nby = rows > 0 ? ((int64_t)(rows - 1) * a.ldy + wy) * 2 : 0;
  TnStream sx = {(const char*)(a.X + (int64_t)mb * a.ldx + i0), nbx, (int64_t)TN_BKM * a.ldx * 2};   // rebased every step: see TnStream
  TnStream sy = {(const char*)(a.Y + (int64_t)mb * a.ldy + j0), nby, (int64_t)TN_BKM * a.ldy * 2};
  // DMA: linear LDS chunk c = tid + 512 i -> row c>>5, physical chunk c&31 holds source chunk (c&31) ^ 4*(row&3)
  int vox[4], voy[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 5) + 16 * i;
    const int sch = (tid & 31) ^ (4 * (row & 3));
    vox[i] = (8 * sch < wx) ? (row * a.ldx + 8 * sch) * 2 : 0x7ffffff0;  // columns past the width read as 0
    voy[i] = (8 * sch < wy) ? (row * a.ldy + 8 * sch) * 2 : 0x7ffffff0;
  }
  const bool do_bias = (bias_out != nullptr) && (ti == 0);
  const bool use_w = do_bias && (a.bias_w != nullptr);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(use_w ? a.bias_w + mb : a.Y), 0, (int)(use_w ? (int64_t)rows * 2 : 0), 0x00020000);
  int vow = lane * 4;
  // fragment read offsets (bytes): row 8h + (l16>>2) [+16kk, +4], byte in row = (wave col base + f*64 + 32*(g4&1) + 8*(l16&3)) ^ 64*(row&3)
  const int rr = l16 >> 2;
  const int rowb = (8 * h + rr) * 512;
  const int cb = 32 * (g4 & 1) + 8 * (l16 & 3);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_tn;
  unsigned ofx[4], ofy[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) ofx[i] = lds0 + rowb + ((wi * 256 + i * 64 + cb) ^ (64 * rr));
#pragma unroll
  for (int j = 0; j < 2; ++j) ofy[j] = lds0 + 32768 + rowb + ((wj * 128 + j * 64 + cb) ^ (64 * rr));

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x16 bacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) bacc[e] = 0.f;
  const u32x4 ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);

  auto stage = [&](int st) {
    char* base = smem_tn + st * 65536 + wid * 1024;
    const __amdgpu_buffer_rsrc_t rx = sx.rsrc(), ry = sy.rsrc();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(rx, base + i * 8192, vox[i], 0);
      glds16(ry, base + 32768 + i * 8192, voy[i], 0);
    }
    sx.next(); sy.next();
    if (use_w) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(smem_tn + 131072 + st * 256), 4, vow, 0, 0, 0);
      vow += TN_BKM * 2;
    }
  };
  auto mfmas = [&](TrFrag8& f, const bf16x8 wa) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_cat(f.x[i][0], f.x[i][1]), tr_cat(f.y[j][0], f.y[j][1]), acc[i][j], 0, 0, 0);  // D[i][j]
    if (do_bias) {  // waves (0, wj) / (1, wj) sum columns wj*64 + {0..31} / {32..63}
      if (wi == 0) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, tr_cat(f.y[0][0], f.y[0][1]), bacc, 0, 0, 0);
      else bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, tr_cat(f.y[1][0], f.y[1][1]), bacc, 0, 0, 0);
    }
  };
  auto compute = [&](auto stc) {
    constexpr int st = decltype(stc)::value;
    unsigned ax[4], ay[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) ax[i] = ofx[i] + st * 65536;
#pragma unroll
    for (int j = 0; j < 2; ++j) ay[j] = ofy[j] + st * 65536;
    WFrag wf;
    if (use_w) w_issue(wf, lds0 + 131072 + st * 256 + 16 * h);
    TrFrag8 F0, F1;
    tr8_issue<0>(F0, ax, ay);
    tr8_issue<8192>(F1, ax, ay);
    tr8_wait<12>(F0);
    if (use_w) w_wait(wf);   // (drains F1 too; bias blocks only)
    mfmas(F0, use_w ? __builtin_bit_cast(bf16x8, wf.w[0]) : ones);
    tr8_issue<16384>(F0, ax, ay);
    tr8_wait<12>(F1);
    mfmas(F1, use_w ? __builtin_bit_cast(bf16x8, wf.w[1]) : ones);
    tr8_issue<24576>(F1, ax, ay);
    tr8_wait<12>(F0);
    mfmas(F0, use_w ? __builtin_bit_cast(bf16x8, wf.w[2]) : ones);
    tr8_wait<0>(F1);
    mfmas(F1, use_w ? __builtin_bit_cast(bf16x8, wf.w[3]) : ones);
  };

  if (nt > 0) {
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int t = 0;
    for (; t + 2 <= nt; t += 2) {
      stage(1);
      compute(std::integral_constant<int, 0>{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + 2 < nt) stage(0);
      compute(std::integral_constant<int, 1>{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (t < nt) {
      compute(std::integral_constant<int, 0>{});
      __syncthreads();
    }
  }
  // store: reg e -> row i = (e&3) + 8*(e>>2) + 4h ; col j = lane&31
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = j0 + wj * 64 + j * 32 + (lane & 31);
      if (col >= a.J) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = i0 + wi * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < a.I) C[(int64_t)row * a.J + col] = acc[i][j][e];
      }
    }
  if (do_bias && h == 0) {  // row 0 of D (reg 0 of the lower half-wave) holds the column sums
    const int col = j0 + wj * 64 + wi * 32 + (lane & 31);
    if (col < a.J) bias_out[col] = bacc[0];
  }
}

// Unsplit shapes whose tile count is not a multiple of the 512 resident blocks (the head: 4 x 397 = 1588 tiles = 3.1
// residencies of ~0.85 ms blocks -> the last 52 blocks ran alone for ~20 % of the kernel, MFMA pipe 37 % busy, PMC
// profiles/pmc/r01g_tn_*).  The first n_whole = 512 * floor(tiles / 512) tiles run as before; each remaining tile is cut
// into S row ranges that write fp32 slabs of the tail's column range [c0, c0 + W) (tiles are numbered ti-fastest, so
// the tail is a column stripe of dW), reduced afterwards in fixed order -- deterministic, no atomics.  (A stream-K cut
// of the whole (tile, step) space was measured 35 % SLOWER: a block then walks the four row-tiles of one dY column
// stripe one after another and re-streams the stripe from HBM each time instead of sharing it through L2.)
__global__ __launch_bounds__(256, 2) void gemm_tn_tail_kernel(TnArgs a, int n_whole, int S, int m_per_split, float* tail_slabs,
                                                              float* tail_bias, int c0, int W) {
  extern __shared__ __attribute__((aligned(16))) char smem_tn[];
  int ti, tj;
  if ((int)blockIdx.x < n_whole) {
    tile_of_block(xcd_remap(blockIdx.x, n_whole), a.tiles_i, a.tiles_j, ti, tj);
    tn_tile<false>(a, nullptr, smem_tn, ti, tj, 0, a.M, a.C, a.J, a.bias_part);
    return;
  }
  const int r = blockIdx.x - n_whole, sp = r % S;
  tile_of_block(n_whole + r / S, a.tiles_i, a.tiles_j, ti, tj);
  const int mb = sp * m_per_split;
  const int me = (mb + m_per_split < a.M) ? mb + m_per_split : a.M;
  tn_tile<false>(a, nullptr, smem_tn, ti, tj, mb, me > mb ? me - mb : 0, tail_slabs + (int64_t)sp * a.I * W - c0, W,
          a.bias_part ? tail_bias + (int64_t)sp * W - c0 : nullptr);
}
// out[r * ldo + c] = sum_s slabs[(s * R + r) * W + c]   (W % 4 == 0, fixed order)
__global__ __launch_bounds__(256) void reduce_slabs_2d_kernel(const float* __restrict__ slabs, float* __restrict__ out, int S,
                                                              int R, int W, int64_t ldo) {
  const int64_t n4 = (int64_t)R * W / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / (W / 4)), c = (int)(i % (W / 4)) * 4;
    f32x4 acc = *(const f32x4*)(slabs + (int64_t)r * W + c);
    for (int s = 1; s < S; ++s) {
      const f32x4 v = *(const f32x4*)(slabs + ((int64_t)s * R + r) * W + c);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    *(f32x4*)(out + (int64_t)r * ldo + c) = acc;
  }
}

// ---- deferred slab reduces ---------------------------------------------------------------------------------------------------
// A split weight gradient ends with two tiny launches (slabs -> dW, bias partials -> dbias: 5-15 us each, mostly launch cost);
// a transformer block has seven of them.  With `deferred` the GEMM only records them and the caller runs all reduces of the
// block in ONE launch (dmi_reduce_slabs_batch) -- the slabs then have to stay alive, i.e. each deferred GEMM needs its own
// workspace.  Same summation order as the per-GEMM reduce: bit-identical results.
#define REDUCE_BATCH_MAX 16
struct ReduceBatch {
  const float* slabs[REDUCE_BATCH_MAX];
  float* out[REDUCE_BATCH_MAX];
  int nsplit[REDUCE_BATCH_MAX];
  int64_t n4[REDUCE_BATCH_MAX];      // float4 count of one slab
  int first_block[REDUCE_BATCH_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void reduce_slabs_batch_kernel(ReduceBatch g) {
  int k = 0;
  while (k + 1 < g.n && (int)blockIdx.x >= g.first_block[k + 1]) ++k;
  const int nb = g.first_block[k + 1] - g.first_block[k];
  const f32x4* sl = (const f32x4*)g.slabs[k];
  const int64_t n4 = g.n4[k];
  for (int64_t i = (int64_t)(blockIdx.x - g.first_block[k]) * 256 + threadIdx.x; i < n4; i += (int64_t)nb * 256) {
    // loads in independent batches of eight, additions in slab order (same sum as the one-at-a-time loop, which the compiler
    // could not pipeline over a run-time split count: 2.4 TB/s)
    const int ns = g.nsplit[k];
    f32x4 acc = sl[i];
    int s2 = 1;
    for (; s2 + 8 <= ns; s2 += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = sl[(int64_t)(s2 + u) * n4 + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc[0] += v[u][0]; acc[1] += v[u][1]; acc[2] += v[u][2]; acc[3] += v[u][3]; }
    }
    if (s2 + 4 <= ns) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = sl[(int64_t)(s2 + u) * n4 + i];
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc[0] += v[u][0]; acc[1] += v[u][1]; acc[2] += v[u][2]; acc[3] += v[u][3]; }
      s2 += 4;
    }
    for (; s2 < ns; ++s2) {
      const f32x4 v = sl[(int64_t)s2 * n4 + i];
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    ((f32x4*)g.out[k])[i] = acc;
  }
}
extern "C" int dmi_reduce_slabs_batch(const dmi_reduce_item* items, int n, void* stream) {
  if (n == 0) return DMI_OK;
  DMI_REQUIRE(items && n > 0 && n <= REDUCE_BATCH_MAX, "reduce_slabs_batch: 1..%d items", REDUCE_BATCH_MAX);
  ReduceBatch g;
  g.n = n;
  int nblk = 0;
  for (int k = 0; k < n; ++k) {
    DMI_REQUIRE(items[k].slabs && items[k].out && items[k].nsplit >= 1 && items[k].n4 > 0, "reduce_slabs_batch: bad item %d", k);
    g.slabs[k] = items[k].slabs; g.out[k] = items[k].out; g.nsplit[k] = items[k].nsplit; g.n4[k] = items[k].n4;
    g.first_block[k] = nblk;
    int64_t b = cdiv64(items[k].n4, 256);
    if (b > 1024) b = 1024;
    nblk += (int)b;
  }
  g.first_block[n] = nblk;
  reduce_slabs_batch_kernel<<<dim3(nblk), dim3(256), 0, (hipStream_t)stream>>>(g);
  DMI_CHECK_LAUNCH("reduce_slabs_batch");
  return DMI_OK;
}

// Weight gradients that share M in one launch (see gemm_tn_group_kernel).  Every problem's workspace holds its own slabs
// (dmi_gemm_tn_workspace_bytes(M, I, J) bytes suffice: the group never splits finer than the single launch would); the slab
// reduces are appended to `deferred` (2 per problem at most) or, with deferred == NULL, run here as one batched launch.
// row splits of the group's plan on 128 x 256 tiles, or 0 where the group stays on 128 x 128 tiles: the union of the wide tiles must
// fill (>= 448 of) the 512 block slots with a split count >= 2 that every problem's own workspace has slabs for
static int tn_group_wide_splits(const int* I, const int* J, int n, int M) {
  if (!g_opt_tn_wide) return 0;
  int Tw = 0, smin = 1 << 30;
  for (int k = 0; k < n; ++k) {
    Tw += ((I[k] + TNW_TI - 1) / TNW_TI) * ((J[k] + TNW_TJ - 1) / TNW_TJ);
    const int sk = tn_splits(M, I[k], J[k]);
    if (sk < smin) smin = sk;
  }
  const int max_s = (M + 4 * TN_BKM - 1) / (4 * TN_BKM);
  int Sw = Tw >= 384 ? 1 : 512 / Tw;
  if (Sw > max_s) Sw = max_s;
  return (Sw >= 2 && Sw <= smin && Tw * Sw >= 448) ? Sw : 0;
}
extern "C" int dmi_gemm_tn_group_plan(const int* I, const int* J, int n, int M) {
  if (!I || !J || n < 1 || n > TN_GROUP_MAX || M <= 0) return 0;
  return tn_group_wide_splits(I, J, n, M);
}
extern "C" int dmi_gemm_tn_group(const dmi_tn_problem* probs, int n, int M, dmi_reduce_item* deferred, int* n_deferred, void* stream) {
  DMI_REQUIRE(probs && n >= 1 && n <= TN_GROUP_MAX && M > 0, "gemm_tn_group: 1..%d problems", TN_GROUP_MAX);
  TnGroup g;
  g.n = n;
  int T = 0;
  for (int k = 0; k < n; ++k) {
    const dmi_tn_problem& q = probs[k];
    DMI_REQUIRE(q.X && q.dY && q.dW && q.workspace, "gemm_tn_group: null pointer in problem %d", k);
    DMI_REQUIRE(q.I % 8 == 0 && q.J % 8 == 0 && q.ldx % 8 == 0 && q.ldy % 8 == 0 && q.ldx >= q.I && q.ldy >= q.J,
                "gemm_tn_group: I, J, ldx, ldy must be multiples of 8 (problem %d: I=%d J=%d)", k, q.I, q.J);
    DMI_REQUIRE((((uintptr_t)q.X | (uintptr_t)q.dY | (uintptr_t)q.dW | (uintptr_t)q.workspace) & 15) == 0, "gemm_tn_group: 16-byte alignment required");
    DMI_REQUIRE(!q.bias_weights || (q.dbias && ((uintptr_t)q.bias_weights & 3) == 0 && M % 2 == 0), "gemm_tn_group: bias_weights needs dbias, 4-byte alignment and an even M");
    DMI_REQUIRE((int64_t)TN_BKM * (q.ldx > q.ldy ? q.ldx : q.ldy) * 2 < 0x7fffffff, "gemm_tn_group: leading dimension too large");
    g.first_tile[k] = T;
    T += ((q.I + 127) / 128) * ((q.J + 127) / 128);
  }
  g.first_tile[n] = T;
  for (int k = n + 1; k <= TN_GROUP_MAX; ++k) g.first_tile[k] = T;
  const int max_s = (M + 4 * TN_BKM - 1) / (4 * TN_BKM);
  int S = T >= 384 ? 1 : 512 / T;
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  // [r06] the wide tile, if its plan fills the slots with splits every problem's workspace has slabs for
  bool wide = false;
  {
    int Is[TN_GROUP_MAX], Js[TN_GROUP_MAX];
    for (int k = 0; k < n; ++k) { Is[k] = probs[k].I; Js[k] = probs[k].J; }
    const int Sw = tn_group_wide_splits(Is, Js, n, M);
    if (Sw) {
      wide = true; S = Sw; T = 0;
      for (int k = 0; k < n; ++k) {
        g.first_tile[k] = T;
        T += ((probs[k].I + TNW_TI - 1) / TNW_TI) * ((probs[k].J + TNW_TJ - 1) / TNW_TJ);
      }
      for (int k = n; k <= TN_GROUP_MAX; ++k) g.first_tile[k] = T;
    }
  }
  dmi_reduce_item items[2 * TN_GROUP_MAX];
  int ni = 0;
  for (int k = 0; k < n; ++k) {
    const dmi_tn_problem& q = probs[k];
    TnArgs& a = g.p[k];
    a.X = q.X; a.Y = q.dY; a.M = M; a.I = q.I; a.J = q.J; a.ldx = q.ldx; a.ldy = q.ldy;
    a.tiles_i = (q.I + 127) / 128; a.tiles_j = wide ? (q.J + TNW_TJ - 1) / TNW_TJ : (q.J + 127) / 128;
    a.m_per_split = (int)round_up64((M + S - 1) / S, TN_BKM);
    float* slabs = (float*)q.workspace;
    const int64_t slab_bytes = (S > 1) ? round_up64((int64_t)S * q.I * q.J * 4, 256) : 0;
    float* bpart = (float*)((char*)q.workspace + slab_bytes);
    a.C = (S > 1) ? slabs : q.dW;
    a.slab_stride = (S > 1) ? (int64_t)q.I * q.J : 0;
    a.bias_w = q.bias_weights;
    a.bias_part = q.dbias ? ((S > 1) ? bpart : q.dbias) : nullptr;
    a.dbg = nullptr;
    if (S > 1) {
      if (q.dbias) { items[ni].slabs = bpart; items[ni].out = q.dbias; items[ni].nsplit = S; items[ni].n4 = q.J / 4; ++ni; }
      items[ni].slabs = slabs; items[ni].out = q.dW; items[ni].nsplit = S; items[ni].n4 = (int64_t)q.I * q.J / 4; ++ni;
    }
  }
  for (int k = n; k < TN_GROUP_MAX; ++k) g.p[k] = g.p[0];
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_tn_group_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TNW_LDS_BYTES);
    attr_done = true;
  }
  if (wide) gemm_tn_group_wide_kernel<<<dim3(T * S), dim3(256), TNW_LDS_BYTES, (hipStream_t)stream>>>(g);
  else gemm_tn_group_kernel<<<dim3(T * S), dim3(256), TN_LDS_BYTES, (hipStream_t)stream>>>(g);
  DMI_CHECK_LAUNCH("gemm_tn_group");
  if (n_deferred) *n_deferred = 0;
  if (deferred && n_deferred) {
    for (int i = 0; i < ni; ++i) deferred[i] = items[i];
    *n_deferred = ni;
    return DMI_OK;
  }
  return dmi_reduce_slabs_batch(items, ni, stream);
}

extern "C" int dmi_gemm_tn(const uint16_t* X, int ldx, const uint16_t* dY, int ldy, float* dW, float* dbias,
                           const uint16_t* bias_weights, int M, int I, int J, void* workspace, dmi_reduce_item* deferred,
                           int* n_deferred, void* stream) {
  DMI_REQUIRE(X && dY && dW && workspace, "gemm_tn: null pointer");
  DMI_REQUIRE(M > 0 && I % 8 == 0 && J % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= I && ldy >= J,
              "gemm_tn: I, J, ldx, ldy must be multiples of 8 (M=%d I=%d J=%d)", M, I, J);
  DMI_REQUIRE((((uintptr_t)X | (uintptr_t)dY | (uintptr_t)dW | (uintptr_t)workspace) & 15) == 0, "gemm_tn: 16-byte alignment required");
  DMI_REQUIRE(!bias_weights || (dbias && ((uintptr_t)bias_weights & 3) == 0 && M % 2 == 0),
              "gemm_tn: bias_weights needs dbias, 4-byte alignment and an even M (the weights travel as dwords: the last weight of an odd M would be range-checked away)");
  DMI_REQUIRE((int64_t)TN_BKM * (ldx > ldy ? ldx : ldy) * 2 < 0x7fffffff, "gemm_tn: leading dimension too large");
  hipStream_t st = (hipStream_t)stream;
  const bool use8 = tn8_eligible(M, I, J);
  const int nsplit = use8 ? tn8_splits(M, I, J) : tn_splits(M, I, J);
  float* slabs = (float*)workspace;
  const int64_t slab_bytes = (nsplit > 1) ? round_up64((int64_t)nsplit * I * J * 4, 256) : 0;
  TnArgs a;
  a.X = X; a.Y = dY; a.M = M; a.I = I; a.J = J; a.ldx = ldx; a.ldy = ldy;
  a.tiles_i = use8 ? (I + 255) / 256 : (I + 127) / 128; a.tiles_j = use8 ? (J + 255) / 256 : (J + 127) / 128;
  a.m_per_split = (int)round_up64((M + nsplit - 1) / nsplit, TN_BKM);
  a.C = (nsplit > 1) ? slabs : dW;
  a.slab_stride = (nsplit > 1) ? (int64_t)I * J : 0;
  a.dbg = g_dbg_buf;
  a.bias_w = bias_weights;
  float* bpart = (float*)((char*)workspace + slab_bytes);   // [nsplit][J] bias partials behind the slabs
  a.bias_part = dbias ? ((nsplit > 1) ? bpart : dbias) : nullptr;
  static bool attr_done = false;
  const int shm = TN_LDS_BYTES;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    (void)hipFuncSetAttribute((const void*)gemm_tn_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    attr_done = true;
  }
  const int tiles = a.tiles_i * a.tiles_j;
  const int tail_tiles = tiles % 512, n_whole = tiles - tail_tiles;
  const int S = tail_tiles ? 512 / tail_tiles : 1;
  if (!use8 && g_opt_tn_wide && nsplit == 1) {   // [r06] gang stream-K on the 128 x 256 tile (gemm_tn_wide_sk_kernel): unsplit gradients with many stripes
    const int wti = (I + TNW_TI - 1) / TNW_TI, wtj = (J + TNW_TJ - 1) / TNW_TJ, nsteps = (M + TNW_BKM - 1) / TNW_BKM;
    const int gangs = (2 * persistent_grid() / wti) & ~7;   // two blocks per CU, the reserved CUs left free; whole gangs per XCD
    if (gangs >= 8 && wtj >= gangs && (int64_t)(wtj + 1) * nsteps < 0x7fffffff) {
      a.tiles_i = wti; a.tiles_j = wtj;
      a.C = dW; a.slab_stride = 0; a.bias_part = dbias;
      static bool attrs = false;
      if (!attrs) { (void)hipFuncSetAttribute((const void*)gemm_tn_wide_sk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TNW_LDS_BYTES); attrs = true; }
      tn_wide_zero_kernel<<<dim3(wtj), dim3(256), 0, st>>>(dW, dbias, I, J, wtj, gangs, nsteps);
      DMI_CHECK_LAUNCH("gemm_tn_wide_zero");
      gemm_tn_wide_sk_kernel<<<dim3(gangs * wti), dim3(256), TNW_LDS_BYTES, st>>>(a, gangs, nsteps);
      DMI_CHECK_LAUNCH("gemm_tn_wide_sk");
      if (n_deferred) *n_deferred = 0;
      return DMI_OK;
    }
  }
  if (use8) {
    static bool attr8 = false;
    if (!attr8) { (void)hipFuncSetAttribute((const void*)gemm_tn8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN8_LDS_BYTES); attr8 = true; }
    gemm_tn8_kernel<<<dim3(tiles * nsplit), dim3(512), TN8_LDS_BYTES, st>>>(a);
    DMI_CHECK_LAUNCH("gemm_tn8");
  } else
  // tiles are numbered ti-fastest: with tiles_i | n_whole the tail is the column stripe [c0, J) of dW
  if (g_opt_tn_tail && nsplit == 1 && tiles > 512 && S >= 2 && a.tiles_i <= GROUP_M && n_whole % a.tiles_i == 0 && J % 4 == 0) {
    const int c0 = (n_whole / a.tiles_i) * 128, W = J - c0;
    const int mps = (int)round_up64((M + S - 1) / S, TN_BKM);
    float* tslab = (float*)workspace;                       // [S][I][W]
    float* tbias = tslab + (int64_t)S * I * W;              // [S][W]
    gemm_tn_tail_kernel<<<dim3(n_whole + tail_tiles * S), dim3(256), shm, st>>>(a, n_whole, S, mps, tslab, tbias, c0, W);
    DMI_CHECK_LAUNCH("gemm_tn_tail");
    reduce_slabs_2d_kernel<<<dim3(512), dim3(256), 0, st>>>(tslab, dW + c0, S, I, W, J);
    DMI_CHECK_LAUNCH("gemm_tn_tail_reduce");
    if (dbias) {
      reduce_slabs_2d_kernel<<<dim3(8), dim3(256), 0, st>>>(tbias, dbias + c0, S, 1, W, J);
      DMI_CHECK_LAUNCH("gemm_tn_tail_bias_reduce");
    }
  } else {
    gemm_tn_kernel<<<dim3(tiles * nsplit), dim3(256), shm, st>>>(a);
    DMI_CHECK_LAUNCH("gemm_tn");
  }
  if (n_deferred) *n_deferred = 0;
  if (nsplit > 1) {
    const int64_t n4 = (int64_t)I * J / 4;
    if (deferred && n_deferred) {   // the caller reduces later (dmi_reduce_slabs_batch); the workspace must stay untouched until then
      int k = 0;
      if (dbias) { deferred[k].slabs = bpart; deferred[k].out = dbias; deferred[k].nsplit = nsplit; deferred[k].n4 = J / 4; ++k; }
      deferred[k].slabs = slabs; deferred[k].out = dW; deferred[k].nsplit = nsplit; deferred[k].n4 = n4; ++k;
      *n_deferred = k;
      return DMI_OK;
    }
    if (dbias) {
      reduce_slabs_kernel<<<dim3((unsigned)cdiv64(J / 4, 256)), dim3(256), 0, st>>>(bpart, dbias, nsplit, J / 4, J / 4);
      DMI_CHECK_LAUNCH("gemm_tn_bias_reduce");
    }
    int64_t blocks = cdiv64(n4, 256);
    if (blocks > 2048) blocks = 2048;
    reduce_slabs_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(slabs, dW, nsplit, n4, n4);
    DMI_CHECK_LAUNCH("gemm_tn_reduce");
  }
  return DMI_OK;
}

// =====================================================================================
// Implicit-im2col convolution GEMM (VAE, src/vae_tf/models.py:59-78 conv_block / conv2d_transpose lowering):
//   out[(b,oy,ox)][n] = sum_t sum_c x[b, oy*s + dy_t, ox*s + dx_t, c] * Wt[n][t*C + c]   (+ bias / ReLU / residual / mask)
// i.e. dmi_im2col + dmi_gemm_nt without the K^2-times-larger column matrix in HBM (im2col was 39 % of the vae_coco step
// in the rocprofv3 profile).  Same tiling / swizzle / epilogue as gemm_nt2_kernel; only the A stager differs: a 64-wide
// k-step lies inside ONE tap (C % 64 == 0), so a lane's source address is its output pixel's base offset plus a
// wave-uniform tap offset, and taps that fall outside the image are steered past the buffer descriptor's end (reads as 0
// = the SAME zero padding).  The k order (tap-major, channel) and all arithmetic equal the materialised path: results
// are bit-identical to im2col + gemm_nt.
// =====================================================================================
template <int FLAGS>
__global__ __launch_bounds__(256, 2) void conv_gemm_nt_kernel(GemmArgs a, ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 16K | B 16K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;

  int tm, tn;
  tile_of_block(xcd_remap(blockIdx.x, gridDim.x), a.tiles_m, a.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nt = a.K / BK;

  // A: the whole activation tensor behind one descriptor (num_records = its size: anything past it reads as zero)
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, a.lda /* bytes of x */, 0x00020000);
  const bf16_t* Bb = a.B + (int64_t)n0 * a.ldb;
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, 0x7fffffff, 0x00020000);
  int pix[4], vob[4];
  unsigned mask[4];
  {
    const int chp = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const int src_ch = chp ^ ((row >> 1) & 7);
      const int m = m0 + row;
      const int ox = m % g.Wo, oy = (m / g.Wo) % g.Ho, b = m / (g.Wo * g.Ho);
      pix[i] = (((b * g.H + oy * g.stride) * g.W + ox * g.stride) * g.C + 8 * src_ch) * 2;
      unsigned mk = 0;
      if (m < a.M) {
        for (int t = 0; t < g.ntaps; ++t) {
          const int iy = oy * g.stride + g.dy[t], ix = ox * g.stride + g.dx[t];
          if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) mk |= 1u << t;
        }
      }
      mask[i] = mk;
      const int rb_ = n0 + row < a.N ? row : a.N - 1 - n0;
      vob[i] = (rb_ * a.ldb + 8 * src_ch) * 2;
    }
  }
  const int c16 = lane & 15, g16 = lane >> 4;
  int offa[2], offb[2];    // per 32-wide k-substep: chunk 4 ks + g of row (tile base + c); 16-row tiles are 2048 B apart
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    offa[ks] = lds_chunk_off(wm * 64 + c16, ks * 4 + g16);
    offb[ks] = 16384 + lds_chunk_off(wn * 64 + c16, ks * 4 + g16);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int st, int t) {   // k-step t: tap = t*64 / C, channel offset (t*64) % C  (wave-uniform scalars)
    char* base = smem + st * 32768 + wid * 1024;
    const int k0 = t * BK;
    const int tap = k0 / g.C, c0 = k0 - tap * g.C;
    const int tapoff = ((g.dy[tap] * g.W + g.dx[tap]) * g.C + c0) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int voa = ((mask[i] >> tap) & 1u) ? pix[i] + tapoff : 0x7ffffff0;
      glds16(ra, base + i * 4096, voa, 0);
      glds16(rb, base + 16384 + i * 4096, vob[i], k0 * 2);
    }
  };
  auto compute = [&](int st) {
    const char* cur = smem + st * 32768;
    MFMA_PRIO(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *(const bf16x8*)(cur + offa[ks] + i * 2048);
        fb[i] = *(const bf16x8*)(cur + offb[ks] + i * 2048);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);  // D[n][m]: lane (c, g) holds C[m = c][n = 4g ..]
    }
    MFMA_PRIO(0);
  };

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int t = 0;
  for (; t + 2 <= nt; t += 2) {
    if (t + 1 < nt) stage(1, t + 1);
    compute(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 2 < nt) stage(0, t + 2);
    compute(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (t < nt) {  // odd tail (stage 0 holds it)
    compute(0);
    __syncthreads();
  }
  epilogue_bf16<FLAGS, 2>(a, acc, (float*)(smem + wid * 8704), lane, m0 + wm * 64, n0 + wn * 64);
}

template <int FLAGS>
static int launch_conv(const GemmArgs& a, const ConvGeom& g, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)conv_gemm_nt_kernel<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); attr_done = true; }
  conv_gemm_nt_kernel<FLAGS><<<dim3(a.tiles_m * a.tiles_n), dim3(256), 65536, st>>>(a, g);
  DMI_CHECK_LAUNCH("conv_gemm_nt");
  return DMI_OK;
}

extern "C" int dmi_conv_gemm_nt(const uint16_t* x, int B, int H, int W, int C, int Ho, int Wo, int stride, int ntaps,
                                const int* dy, const int* dx, const uint16_t* Wt, int ldw, uint16_t* out, int ldc, int N,
                                int flags, const uint16_t* bias, const uint16_t* residual, const uint16_t* relu_src,
                                void* stream) {
  DMI_REQUIRE(x && Wt && out && dy && dx, "conv_gemm_nt: null pointer");
  DMI_REQUIRE(C % 64 == 0 && ntaps >= 1 && ntaps <= CONV_MAX_TAPS && N % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && ldw >= ntaps * C && ldc >= N,
              "conv_gemm_nt: need C%%64==0, 1..16 taps, N/ldw/ldc multiples of 8 (C=%d ntaps=%d N=%d)", C, ntaps, N);
  DMI_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && stride >= 1, "conv_gemm_nt: bad geometry");
  const int64_t xbytes = (int64_t)B * H * W * C * 2, Ml = (int64_t)B * Ho * Wo;
  DMI_REQUIRE(xbytes < 0x7ffffff0 && Ml < 0x7fffffff, "conv_gemm_nt: activation tensor too large for 32-bit buffer offsets");
  DMI_REQUIRE((((uintptr_t)x | (uintptr_t)Wt | (uintptr_t)out) & 15) == 0, "conv_gemm_nt: operands must be 16-byte aligned");
  DMI_REQUIRE(!(flags & DMI_GEMM_BIAS) || bias, "conv_gemm_nt: bias flag without pointer");
  DMI_REQUIRE(!(flags & DMI_GEMM_RESIDUAL) || residual, "conv_gemm_nt: residual flag without pointer");
  DMI_REQUIRE(!(flags & DMI_GEMM_RELU_MASK) || relu_src, "conv_gemm_nt: relu-mask flag without pointer");
  GemmArgs a;
  a.A = x; a.B = Wt; a.C = out; a.bias = bias; a.residual = residual; a.relu_src = relu_src;
  a.M = (int)Ml; a.N = N; a.K = ntaps * C; a.lda = (int)xbytes /* descriptor size */; a.ldb = ldw; a.ldc = ldc;
  a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
  a.rowscale = nullptr; a.rowshift = nullptr; a.rowsum_part = nullptr; a.relu_bits = nullptr;
  a.k_per_split = a.K; a.slab_stride = 0; a.dbg = nullptr; a.cpol = 0;
  a.ln_gamma = nullptr; a.ln_beta = nullptr; a.ln_y = nullptr; a.ln_mean = nullptr; a.ln_rstd = nullptr; a.ln_eps = 0.f; a.ln_ldy = 0;
  a.ln_x = nullptr; a.ln_part = nullptr; a.B2 = nullptr; a.C2 = nullptr; a.ldb2 = 0;
  ConvGeom g;
  g.H = H; g.W = W; g.C = C; g.Ho = Ho; g.Wo = Wo; g.stride = stride; g.ntaps = ntaps; g.lw = 0; g.lh = 0;
  for (int i = 0; i < CONV_MAX_TAPS; ++i) { g.dy[i] = i < ntaps ? dy[i] : 0; g.dx[i] = i < ntaps ? dx[i] : 0; }
  hipStream_t st = (hipStream_t)stream;
  switch (flags) {
    case 0: return launch_conv<0>(a, g, st);
    case DMI_GEMM_BIAS: return launch_conv<DMI_GEMM_BIAS>(a, g, st);
    case DMI_GEMM_BIAS | DMI_GEMM_RELU: return launch_conv<DMI_GEMM_BIAS | DMI_GEMM_RELU>(a, g, st);
    case DMI_GEMM_BIAS | DMI_GEMM_RESIDUAL: return launch_conv<DMI_GEMM_BIAS | DMI_GEMM_RESIDUAL>(a, g, st);
    case DMI_GEMM_RESIDUAL: return launch_conv<DMI_GEMM_RESIDUAL>(a, g, st);
    case DMI_GEMM_RELU_MASK: return launch_conv<DMI_GEMM_RELU_MASK>(a, g, st);
    default: DMI_REQUIRE(false, "conv_gemm_nt: unsupported epilogue flags %d", flags);
  }
  return DMI_OK;
}

// ---- implicit-im2col weight gradient: dW[(t,c)][n] = sum_pixels x[pixel + tap t][c] * dy[pixel][n] -------------------
__global__ __launch_bounds__(256, 2) void conv_wgrad_tn_kernel(TnArgs a, ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem_tn[];
  const int tiles = a.tiles_i * a.tiles_j;
  const int P = xcd_remap(blockIdx.x, gridDim.x);
  const int split = P / tiles;
  int ti, tj;
  tile_of_block(P - split * tiles, a.tiles_i, a.tiles_j, ti, tj);
  const int mb = split * a.m_per_split;
  const int me = (mb + a.m_per_split < a.M) ? mb + a.m_per_split : a.M;
  tn_tile<true>(a, &g, smem_tn, ti, tj, mb, me > mb ? me - mb : 0, a.C + (int64_t)split * a.slab_stride, a.J,
                a.bias_part ? a.bias_part + (int64_t)split * a.J : nullptr);
}

static int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return ((1 << l) == v) ? l : -1;
}
extern "C" int64_t dmi_conv_wgrad_tn_workspace_bytes(int M, int K, int N) { return dmi_gemm_tn_workspace_bytes(M, K, N); }
extern "C" int dmi_conv_wgrad_tn(const uint16_t* x, int B, int H, int W, int C, int Ho, int Wo, int stride, int ntaps,
                                 const int* dy, const int* dx, const uint16_t* dY, int ldy, int N, float* dW, float* dbias,
                                 void* workspace, dmi_reduce_item* deferred, int* n_deferred, void* stream) {
  DMI_REQUIRE(x && dY && dW && dy && dx && workspace, "conv_wgrad_tn: null pointer");
  if (n_deferred) *n_deferred = 0;
  DMI_REQUIRE(C % 64 == 0 && ntaps >= 1 && ntaps <= CONV_MAX_TAPS && N % 8 == 0 && ldy % 8 == 0 && ldy >= N,
              "conv_wgrad_tn: need C%%64==0, 1..16 taps, N/ldy multiples of 8 (C=%d ntaps=%d N=%d)", C, ntaps, N);
  const int lw = ilog2_exact(Wo), lh = ilog2_exact(Ho);
  DMI_REQUIRE(lw >= 0 && lh >= 0 && B > 0 && H > 0 && W > 0 && stride >= 1, "conv_wgrad_tn: output dims must be powers of two (Ho=%d Wo=%d)", Ho, Wo);
  const int64_t xbytes = (int64_t)B * H * W * C * 2, Ml = (int64_t)B * Ho * Wo;
  DMI_REQUIRE(xbytes < 0x7ffffff0 && Ml < 0x7fffffff, "conv_wgrad_tn: activation tensor too large for 32-bit buffer offsets");
  DMI_REQUIRE((((uintptr_t)x | (uintptr_t)dY | (uintptr_t)dW | (uintptr_t)workspace) & 15) == 0, "conv_wgrad_tn: 16-byte alignment required");
  DMI_REQUIRE((int64_t)TN_BKM * ldy * 2 < 0x7fffffff, "conv_wgrad_tn: leading dimension too large");
  hipStream_t st = (hipStream_t)stream;
  const int M = (int)Ml, I = ntaps * C, J = N;
  const int nsplit = tn_splits(M, I, J);
  float* slabs = (float*)workspace;
  const int64_t slab_bytes = (nsplit > 1) ? round_up64((int64_t)nsplit * I * J * 4, 256) : 0;
  TnArgs a;
  a.X = x; a.Y = dY; a.M = M; a.I = I; a.J = J; a.ldx = (int)xbytes /* descriptor size */; a.ldy = ldy;
  a.tiles_i = (I + 127) / 128; a.tiles_j = (J + 127) / 128;
  a.m_per_split = (int)round_up64((M + nsplit - 1) / nsplit, TN_BKM);
  a.C = (nsplit > 1) ? slabs : dW;
  a.slab_stride = (nsplit > 1) ? (int64_t)I * J : 0;
  a.dbg = nullptr; a.bias_w = nullptr;
  float* bpart = (float*)((char*)workspace + slab_bytes);
  a.bias_part = dbias ? ((nsplit > 1) ? bpart : dbias) : nullptr;
  ConvGeom g;
  g.H = H; g.W = W; g.C = C; g.Ho = Ho; g.Wo = Wo; g.stride = stride; g.ntaps = ntaps; g.lw = lw; g.lh = lh;
  for (int i = 0; i < CONV_MAX_TAPS; ++i) { g.dy[i] = i < ntaps ? dy[i] : 0; g.dx[i] = i < ntaps ? dx[i] : 0; }
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)conv_wgrad_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES); attr_done = true; }
  conv_wgrad_tn_kernel<<<dim3(a.tiles_i * a.tiles_j * nsplit), dim3(256), TN_LDS_BYTES, st>>>(a, g);
  DMI_CHECK_LAUNCH("conv_wgrad_tn");
  if (nsplit > 1) {
    const int64_t n4 = (int64_t)I * J / 4;
    if (deferred && n_deferred) {   // the caller reduces later (dmi_reduce_slabs_batch); the workspace must stay untouched until then
      int k = 0;
      if (dbias) { deferred[k].slabs = bpart; deferred[k].out = dbias; deferred[k].nsplit = nsplit; deferred[k].n4 = J / 4; ++k; }
      deferred[k].slabs = slabs; deferred[k].out = dW; deferred[k].nsplit = nsplit; deferred[k].n4 = n4; ++k;
      *n_deferred = k;
      return DMI_OK;
    }
    if (dbias) {
      reduce_slabs_kernel<<<dim3((unsigned)cdiv64(J / 4, 256)), dim3(256), 0, st>>>(bpart, dbias, nsplit, J / 4, J / 4);
      DMI_CHECK_LAUNCH("conv_wgrad_tn_bias_reduce");
    }
    int64_t blocks = cdiv64(n4, 256);
    if (blocks > 2048) blocks = 2048;
    reduce_slabs_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(slabs, dW, nsplit, n4, n4);
    DMI_CHECK_LAUNCH("conv_wgrad_tn_reduce");
  }
  return DMI_OK;
}
