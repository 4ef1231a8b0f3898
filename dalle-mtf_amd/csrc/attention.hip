// attention.hip -- causal attention for the DALL-E block (reference call site
// src/dalle_mtf/models.py:221-227,292-299; semantics SURVEY Appendix A.2/A.3):
//   logits = q.k^T  UNSCALED (the 1/sqrt(k) factor is folded into Wq's initialiser), fp32,
//   + (-1e10 where key > query), softmax over keys, . v          head dim fixed at 128.
//
// Design (gfx950, v_mfma_f32_32x32x16_bf16): never materialise [S,S].  Every kernel arranges the MFMA
// operand roles so that the softmax row index is the LANE (accumulator column) and the reduction index
// lives in the accumulator REGISTERS -> row max / row sum are 15 in-register ops + one half-wave
// exchange, and P / dS feed the next MFMA as B-operands straight from registers (the contraction-slot
// -> key mapping of the packed registers is matched by the way the V^T / K^T / Q^T / dO^T operand
// fragments are fetched: two 8-byte LDS reads at key offsets 4h and 8+4h).
//   fwd      : S^T = K Q^T ;  O^T += V^T P^T        (block = 128 queries, 64-key tiles)
//   bwd dQ   : S^T = K Q^T ; dP^T = V dO^T ; dQ^T += K^T dS^T
//   bwd dKdV : S = Q K^T ; dP = dO V^T ; dV^T += dO^T P ; dK^T += Q^T dS   (block = 128 keys, 32-query tiles)
// No transposed operand copies: K / V / Q / dO tiles land in LDS in their natural layout by LDS-DMA (swizzled on the
// source side) and every transposed fragment (V^T, K^T, Q^T, dO^T) is a hardware transpose read (ds_read_b64_tr_b16).
#include "common.h"
#include <type_traits>

#define HD 128
#define LOG2E_F 1.4426950408889634f
#define KP 136  // pitch (elements) of a natural [rows][128] tile  : 272 B
#define TP 68   // pitch of a transposed [128][64] tile             : 136 B
#define TP32 36 // pitch of a transposed [128][32] tile            :  72 B

__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ bf16x8 pack_bf8(const float* p) {
  u32x4 v = {pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
  return __builtin_bit_cast(bf16x8, v);
}

// natural tile: ROWS x 128 bf16 from a row-major matrix (row stride ld), rows clamped to [0, nrows-1]
template <int ROWS>
__device__ __forceinline__ void load_nat_regs(u32x4* r, const bf16_t* __restrict__ g, int64_t ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS / 16; ++i) {
    const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
    int gr = row0 + row;
    gr = gr < nrows ? gr : nrows - 1;
    r[i] = *(const u32x4*)(g + (int64_t)gr * ld + 8 * ch);
  }
}
template <int ROWS>
__device__ __forceinline__ void store_nat_lds(const u32x4* r, bf16_t* lds, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS / 16; ++i) {
    const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
    *(u32x4*)(lds + row * KP + 8 * ch) = r[i];
  }
}
// transposed tile: 128 rows (d) x COLS (positions) from T[128][S] (row stride S), cols >= S zero-filled
template <int COLS>
__device__ __forceinline__ void load_tr_regs(u32x4* r, const bf16_t* __restrict__ g, int S, int col0, int tid) {
  constexpr int CPR = COLS / 8;  // chunks per row
#pragma unroll
  for (int i = 0; i < 128 * CPR / 256; ++i) {
    const int c = tid + 256 * i, row = c / CPR, ch = c % CPR;
    const int col = col0 + 8 * ch;
    r[i] = (col < S) ? *(const u32x4*)(g + (int64_t)row * S + col) : u32x4{0, 0, 0, 0};
  }
}
template <int COLS, int PITCH>
__device__ __forceinline__ void store_tr_lds(const u32x4* r, bf16_t* lds, int tid) {
  constexpr int CPR = COLS / 8;
#pragma unroll
  for (int i = 0; i < 128 * CPR / 256; ++i) {
    const int c = tid + 256 * i, row = c / CPR, ch = c % CPR;
    u32x2* p = (u32x2*)(lds + row * PITCH + 8 * ch);  // rows are only 8-byte aligned
    p[0] = u32x2{r[i][0], r[i][1]};
    p[1] = u32x2{r[i][2], r[i][3]};
  }
}

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

struct Tr4 {
  u32x2 a0, a1, b0, b1, c0, c1, d0, d1;  // 4 fragments (dt = 0..3) x {keys/queries +0..3, +8..11}
};
__device__ __forceinline__ void tr4_issue(Tr4& f, unsigned lo0, unsigned hi0, unsigned lo1, unsigned hi1, unsigned lo2,
                                          unsigned hi2, unsigned lo3, unsigned hi3) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\t"
      "ds_read_b64_tr_b16 %1, %9\n\t"
      "ds_read_b64_tr_b16 %2, %10\n\t"
      "ds_read_b64_tr_b16 %3, %11\n\t"
      "ds_read_b64_tr_b16 %4, %12\n\t"
      "ds_read_b64_tr_b16 %5, %13\n\t"
      "ds_read_b64_tr_b16 %6, %14\n\t"
      "ds_read_b64_tr_b16 %7, %15"
      : "=&v"(f.a0), "=&v"(f.a1), "=&v"(f.b0), "=&v"(f.b1), "=&v"(f.c0), "=&v"(f.c1), "=&v"(f.d0), "=&v"(f.d1)
      : "v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1), "v"(lo2), "v"(hi2), "v"(lo3), "v"(hi3)
      : "memory");
}
// the same eight reads at a compile-time byte offset from eight per-step base addresses (no address VALU per fragment set)
template <int OFF>
__device__ __forceinline__ void tr4_issue_off(Tr4& f, const unsigned (&a)[8]) {
#ifdef ATTN_DBG_NOLDS
  asm volatile("" : "=v"(f.a0), "=v"(f.a1), "=v"(f.b0), "=v"(f.b1), "=v"(f.c0), "=v"(f.c1), "=v"(f.d0), "=v"(f.d1) : "v"(a[0]), "v"(a[1]));
  return;
#endif
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%16\n\t"
      "ds_read_b64_tr_b16 %1, %9 offset:%16\n\t"
      "ds_read_b64_tr_b16 %2, %10 offset:%16\n\t"
      "ds_read_b64_tr_b16 %3, %11 offset:%16\n\t"
      "ds_read_b64_tr_b16 %4, %12 offset:%16\n\t"
      "ds_read_b64_tr_b16 %5, %13 offset:%16\n\t"
      "ds_read_b64_tr_b16 %6, %14 offset:%16\n\t"
      "ds_read_b64_tr_b16 %7, %15 offset:%16"
      : "=&v"(f.a0), "=&v"(f.a1), "=&v"(f.b0), "=&v"(f.b1), "=&v"(f.c0), "=&v"(f.c1), "=&v"(f.d0), "=&v"(f.d1)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "i"(OFF)
      : "memory");
}
__device__ __forceinline__ void tr4_wait(Tr4& f) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1), "+v"(f.c0), "+v"(f.c1), "+v"(f.d0), "+v"(f.d1)
               :
               : "memory");
}
// wait until at most 8 younger LDS operations are outstanding; "prev" names a set whose consumers must stay above
__device__ __forceinline__ void tr4_wait8(Tr4& f, Tr4& prev) {
  asm volatile("s_waitcnt lgkmcnt(8)"
               : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1), "+v"(f.c0), "+v"(f.c1), "+v"(f.d0), "+v"(f.d1),
                 "+v"(prev.a0), "+v"(prev.a1), "+v"(prev.b0), "+v"(prev.b1), "+v"(prev.c0), "+v"(prev.c1), "+v"(prev.d0), "+v"(prev.d1)
               :
               : "memory");
}
// wait until at most N younger LDS operations are outstanding; "other" names the set whose readers / whose issue must not move across
template <int N>
__device__ __forceinline__ void tr4_waitn(Tr4& f, Tr4& other) {
  asm volatile("s_waitcnt lgkmcnt(%16)"
               : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1), "+v"(f.c0), "+v"(f.c1), "+v"(f.d0), "+v"(f.d1),
                 "+v"(other.a0), "+v"(other.a1), "+v"(other.b0), "+v"(other.b1), "+v"(other.c0), "+v"(other.c1), "+v"(other.d0), "+v"(other.d1)
               : "n"(N)
               : "memory");
}
struct Tr2 {
  u32x2 a0, a1, b0, b1;  // 2 fragments x {+0..3, +8..11}
};
__device__ __forceinline__ void tr2_issue(Tr2& f, unsigned lo0, unsigned hi0, unsigned lo1, unsigned hi1) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %4\n\t"
      "ds_read_b64_tr_b16 %1, %5\n\t"
      "ds_read_b64_tr_b16 %2, %6\n\t"
      "ds_read_b64_tr_b16 %3, %7"
      : "=&v"(f.a0), "=&v"(f.a1), "=&v"(f.b0), "=&v"(f.b1)
      : "v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1)
      : "memory");
}
__device__ __forceinline__ void tr2_wait(Tr2& f) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1) : : "memory");
}
__device__ __forceinline__ bf16x8 cat2(u32x2 a, u32x2 b) {
  u32x4 v = {a[0], a[1], b[0], b[1]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, int voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ void dma4(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, int voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 4, voff, 0, 0, 0);
}

// Natural-layout operand triple of one k-step of the dK/dV kernel's S / dP products (Q row, dO row, V row fragments),
// read by inline asm so that three sets can be in flight ahead of the MFMAs with counted lgkmcnt waits; hipcc's own
// schedule of the equivalent C++ loads waits ~26 times per 32 MFMAs, which a kernel at ONE wave per SIMD cannot hide.
struct Fr3 {
  u32x4 q, d, v;
};
template <int OFFQ>
__device__ __forceinline__ void fr3_issue(Fr3& f, unsigned aq, unsigned av) {
  asm volatile(
      "ds_read_b128 %0, %3 offset:%5\n\t"
      "ds_read_b128 %1, %3 offset:%6\n\t"
      "ds_read_b128 %2, %4"
      : "=&v"(f.q), "=&v"(f.d), "=&v"(f.v)
      : "v"(aq), "v"(av), "i"(OFFQ), "i"(OFFQ + 8192)
      : "memory");
}
// wait until at most N younger LDS operations are outstanding (LDS returns in order).  Naming the awaited set "+v" keeps
// its consumers below this point; naming the previously consumed set "+v" keeps the MFMAs that read it above (naming
// the accumulators instead would drag them out of the AGPRs on every k-step).
template <int N>
__device__ __forceinline__ void fr3_wait(Fr3& f, Fr3& prev) {
  asm volatile("s_waitcnt lgkmcnt(%6)"
               : "+v"(f.q), "+v"(f.d), "+v"(f.v), "+v"(prev.q), "+v"(prev.d), "+v"(prev.v)
               : "n"(N)
               : "memory");
}
template <int N>
__device__ __forceinline__ void fr3_wait(Fr3& f) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f.q), "+v"(f.d), "+v"(f.v) : "n"(N) : "memory");
}

extern int g_opt_attn_xcd;
#ifdef ATTN_STAMP
extern unsigned long long* g_dbg_buf;
#endif
extern int g_opt_attn_fwd;      // 0: the round-2 forward kernel, 1 (default): the software-pipelined round-6 kernel (A/B switch)
extern int g_opt_attn_bwd;      // backward kernels: bit 0 = whole-row epilogue stores through LDS (A/B switch)
extern int g_opt_reserve_cus;   // CUs the persistent grids leave to concurrent kernels (csrc/gemm.hip)
// Persistent-block schedule: the grid is one (dK/dV) or two (forward, dQ) blocks per CU; hardware block L (dispatched to XCD L % 8) owns
// bin k = L / 8 of its XCD.  The XCD's work items -- (tile, batch*head) for its contiguous eighth of the (batch, head) pairs,
// sorted heaviest tile first -- are dealt to the bins in serpentine order (round 0: bins 0..P-1, round 1: P-1..0, ...): with
// causal tiles of 40, 36, ..., 4 steps and 16 pairs per XCD every bin gets 108 or 112 steps, there are no block launch gaps
// (measured: a CU's five blocks summed to 114 us inside a 160 us kernel) and no atomics.  perxcd = 0 (option attn_xcd = 0, or
// shapes that do not divide by 8): one serpentine over all blocks.
struct AttnSched {
  int nbh, bh_lo, P, k, nitems;
};
__device__ __forceinline__ AttnSched attn_sched(int T, int BH, int perxcd) {
  AttnSched s;
  const int L = blockIdx.x, G = gridDim.x;
  if (perxcd) { s.nbh = BH >> 3; s.bh_lo = (L & 7) * s.nbh; s.P = G >> 3; s.k = L >> 3; }
  else { s.nbh = BH; s.bh_lo = 0; s.P = G; s.k = L; }
  s.nitems = T * s.nbh;
  return s;
}
__device__ __forceinline__ bool attn_item(const AttnSched& s, int r, int& tile, int& bh) {
  const int j = (r & 1) ? (r + 1) * s.P - 1 - s.k : r * s.P + s.k;   // increasing in r: the first miss ends the block
  if (j >= s.nitems) return false;
  tile = j / s.nbh;
  bh = s.bh_lo + j - tile * s.nbh;
  return true;
}
// [r06] Epilogue of a wave's 32 x 128 result tile held as 32x32 MFMA accumulators (lane (r, h) owns row r, acc[dt][e] is column
// 32 dt + (e & 3) + 8 (e >> 2) + 4 h): staged through a wave-private LDS strip (32 rows, pitch 272 B) and stored as WHOLE 256-byte rows --
// 16 B per lane, four rows per store instruction -- instead of 16 eight-byte pieces per lane that touch 32 rows per instruction (the
// store tail of an attention kernel is issue-bound on this part: cdna_hip_programming.md T21 / MI355X_MICROARCH.md "attention epilogue
// store tail").  The caller guarantees that nobody else uses the strip (a barrier behind the last tile's LDS reads).
#define ROWS_PITCH 272
__device__ __forceinline__ void store_rows_via_lds(char* strip, const f32x16 (&acc)[4], float scale, bf16_t* gbase, int64_t pitch,
                                                   int rows_valid, int lane) {
  const int r = lane & 31, h = lane >> 5, g4 = lane >> 4, l16 = lane & 15;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int dd = dt * 32 + 8 * q4 + 4 * h;
      *(u32x2*)(strip + r * ROWS_PITCH + dd * 2) = u32x2{pack2bf(acc[dt][4 * q4] * scale, acc[dt][4 * q4 + 1] * scale),
                                                          pack2bf(acc[dt][4 * q4 + 2] * scale, acc[dt][4 * q4 + 3] * scale)};
    }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private strip: in-order LDS, no block barrier needed
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = 4 * it + g4;
    const u32x4 v = *(const u32x4*)(strip + row * ROWS_PITCH + l16 * 16);
    if (row < rows_valid) *(u32x4*)(gbase + (int64_t)row * pitch + l16 * 8) = v;
  }
  __builtin_amdgcn_wave_barrier();
}
// (A coalesced form of the prologue fetches -- Q / dO / O / K as 128-byte half rows, 8 rows per load instruction, turned into fragments
// through wave-private 4-KB LDS strips -- was built, bit-identical and measured NEUTRAL in all three kernels (backward 242.2 vs 242.3 us,
// profiles/r06_attn_ab.log): the fragments' 32-byte pieces of 32 rows per instruction are absorbed by the vector L1.  Removed.)
#define QK_STAGE 32768  // K 16384 | V 16384
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                          float* __restrict__ lse, int B, int H, int S, int perxcd) {
  extern __shared__ __attribute__((aligned(16))) char sm[];  // 2 x QK_STAGE
  const int d = H * HD, ld3 = 3 * d;
  const int T = (S + 127) / 128;
  const AttnSched sched = attn_sched(T, B * H, perxcd);
  int tile_, bh;
  for (int round = 0; attn_item(sched, round, tile_, bh); ++round) {   // persistent: two blocks per CU walk their item lists
  const int qt = T - 1 - tile_;  // heaviest (latest) query tiles first
  const int b = bh / H, hh = bh % H;
  const int q0 = qt * 128;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));   // opaque per item (see attn_bwd_dkv_kernel)
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, g4 = lane >> 4, l16 = lane & 15;
  const int qrow = q0 + wid * 32 + r;
  const int qrow_c = qrow < S ? qrow : S - 1;
  const bf16_t* qb = qkv + (int64_t)b * S * ld3 + hh * HD;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sm;
  const int nbytes = (int)(((int64_t)(S - 1) * ld3 + HD) * 2);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(qb + d), 0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(qb + 2 * d), 0, nbytes, 0x00020000);

  bf16x8 qf[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) qf[kk] = *(const bf16x8*)(qb + (int64_t)qrow_c * ld3 + 16 * kk + 8 * h);

  int vo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 4, pc = c & 15;
    vo[i] = (row * ld3 + 8 * (pc ^ swz(row))) * 2;
  }
  auto stage = [&](int st, int key0) {
    char* base = sm + st * QK_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma16(rk, base + (wid * 64 + 256 * i) * 16, vo[i] + key0 * ld3 * 2);
      dma16(rv, base + 16384 + (wid * 64 + 256 * i) * 16, vo[i] + key0 * ld3 * 2);
    }
  };
  int ofa[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) ofa[kk] = r * 256 + (((2 * kk + h) ^ swz(r)) << 4);
  const int rr = l16 >> 2;
  unsigned oft[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int w2 = 0; w2 < 2; ++w2) {
      const int row = 4 * h + rr + 8 * w2;
      const int chunk = dt * 4 + 2 * (g4 & 1) + ((l16 & 3) >> 1);
      oft[dt][w2] = row * 256 + ((chunk ^ swz(row)) << 4) + 8 * (l16 & 1);
    }

  f32x16 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  float m = -1e30f, l = 0.f;

  const int qlast = (q0 + 127 < S - 1) ? q0 + 127 : S - 1;
  const int nsteps = qlast / 64 + 1;
  const int wave_qmin = q0 + wid * 32, wave_qmax = wave_qmin + 31;

  auto compute = [&](int st, int j) {
    if (64 * j > wave_qmax) return;  // wave-uniform: tile entirely above the diagonal for this wave
    const char* base = sm + st * QK_STAGE;
    const unsigned vb = lds0 + st * QK_STAGE + 16384;
    f32x16 s0, s1;
#pragma unroll
    for (int e = 0; e < 16; ++e) s0[e] = s1[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const bf16x8 a0 = *(const bf16x8*)(base + ofa[kk]);
      const bf16x8 a1 = *(const bf16x8*)(base + 8192 + ofa[kk]);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qf[kk], s0, 0, 0, 0);  // S^T[key][q]
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qf[kk], s1, 0, 0, 0);
    }
    Tr4 tv[2];
    tr4_issue(tv[0], vb + oft[0][0], vb + oft[0][1], vb + oft[1][0], vb + oft[1][1], vb + oft[2][0], vb + oft[2][1], vb + oft[3][0], vb + oft[3][1]);
    if (64 * j + 63 > wave_qmin) {  // diagonal tile: additive -1e10 mask == probability exactly 0
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = 64 * j + (e & 3) + 8 * (e >> 2) + 4 * h;
        s0[e] = (key > qrow) ? -1e30f : s0[e];
        s1[e] = (key + 32 > qrow) ? -1e30f : s1[e];
      }
    }
    float mx = -1e30f;
#pragma unroll
    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fmaxf(s0[e], s1[e]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // online softmax in base 2: p = exp2(s * log2(e) - m2), m2 = running max * log2(e): one fma + one v_exp_f32 per element.
    // The accumulator rescale (64 multiplies per lane) runs only when some row's maximum actually grew (wave-uniform
    // branch; with alpha == 1 for every lane the skipped work is the identity, so results are unchanged) -- after the
    // first few key tiles of a row that is rare.
    if (__any(mx > m)) {
      const float mn = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f((m - mn) * LOG2E_F);
      m = mn;
      l *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
    }
    const float m2 = m * LOG2E_F;
    float rs = 0.f;
    bf16x8 pb[4];
    {
      float pe[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        pe[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[e], LOG2E_F, -m2));
        rs += pe[e];
      }
      pb[0] = pack_bf8(pe);
      pb[1] = pack_bf8(pe + 8);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        pe[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[e], LOG2E_F, -m2));
        rs += pe[e];
      }
      pb[2] = pack_bf8(pe);
      pb[3] = pack_bf8(pe + 8);
    }
    l += rs;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // 16-key step: keys 16*ks + {4h..4h+3, 8+4h..8+4h+3}
      Tr4& c = tv[ks & 1];
      tr4_wait(c);
      if (ks < 3) {
        const unsigned o2 = vb + (ks + 1) * 4096;
        tr4_issue(tv[(ks + 1) & 1], o2 + oft[0][0], o2 + oft[0][1], o2 + oft[1][0], o2 + oft[1][1], o2 + oft[2][0], o2 + oft[2][1], o2 + oft[3][0], o2 + oft[3][1]);
      }
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(c.a0, c.a1), pb[ks], oacc[0], 0, 0, 0);  // O^T[d][q]
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(c.b0, c.b1), pb[ks], oacc[1], 0, 0, 0);
      oacc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(c.c0, c.c1), pb[ks], oacc[2], 0, 0, 0);
      oacc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(c.d0, c.d1), pb[ks], oacc[3], 0, 0, 0);
    }
  };

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int j = 0;
  for (; j + 2 <= nsteps; j += 2) {
    stage(1, 64 * (j + 1));
    compute(0, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 2 < nsteps) stage(0, 64 * (j + 2));
    compute(1, j + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (j < nsteps) compute(0, j);

  l += __shfl_xor(l, 32, 64);
  if (qrow < S) {
    const float inv = 1.f / l;
    bf16_t* op = o + ((int64_t)b * S + qrow) * d + hh * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int dd = dt * 32 + 8 * q4 + 4 * h;
        *(u32x2*)(op + dd) = u32x2{pack2bf(oacc[dt][4 * q4] * inv, oacc[dt][4 * q4 + 1] * inv),
                                   pack2bf(oacc[dt][4 * q4 + 2] * inv, oacc[dt][4 * q4 + 3] * inv)};
      }
    if (h == 0) lse[(int64_t)bh * S + qrow] = m + __logf(l);
  }
  __syncthreads();   // the last step's LDS reads are done before the next item's first DMA
  }   // items
}


// =====================================================================================
// forward, round 6: software-pipelined over the key tiles
// =====================================================================================
// The round-2 kernel above computes a key tile in program order -- S = K Q^T (hipcc waits for every pair of ds_read_b128 before the
// two MFMAs that use it), softmax, O += V^T P -- so a wave's LDS latencies, its ~200 VALU instructions per tile and its 32 MFMAs ADD,
// and only the second wave of the SIMD (the other block of the CU) hides anything: normalised MFMA-busy 0.29, 6.7 k cycles per tile
// step of a block for 1 k cycles of matrix work per wave (profiles/r05_pmc_normalised.txt).  This form keeps the tile shapes, the
// LDS layout, the operand roles (softmax row = lane) and the persistent schedule, and changes the instruction stream:
//   * two S accumulators: step j runs  phase A  S(j+1) = K(j+1) Q^T  (16 MFMAs, K fragments by inline-asm ds_read_b128 one k-step
//     ahead with counted waits)  interleaved with  exp2 / row sum / bf16 packing of S(j)  (one 4-element slice per two MFMAs,
//     pinned with sched_group_barrier);  phase B  O += V(j)^T P(j)  (16 MFMAs, transposed V fragments one 16-key step ahead)  with
//     the causal mask and the row maximum of S(j+1) riding between them.  K is staged two tiles ahead, V one (same 64 KB of LDS);
//   * the running maximum lives in base-2 units as an INTEGER (m2 = ceil(max * log2 e)): every rescale factor is an exact power of
//     two, so rescaling or not rescaling gives the same bits as long as nothing overflows -- and the O / l rescale (64 multiplies
//     per lane) runs only when some row's maximum grew by more than 2^16 since its last rescale (on random scores the round-2
//     kernel's "some row grew at all" fired on nearly every tile).  P <= 2^17 in bf16 (8 exponent bits), l and O in fp32;
//   * O is normalised by the sum of the bf16-ROUNDED probabilities (v_dot2c_f32_bf16 against (1, 1) in phase B, where the VALU idles), so
//     the weights the MFMA applies sum to exactly 1; lse keeps the unrounded sum (the backward needs the true one).  Without this the
//     integer maximum costs accuracy: the dominant probability is no longer exactly 1, its rounding error stops cancelling and peaked
//     rows carry one extra bf16 rounding (O at 1.47 x pure rounding against 1.02 for the round-2 kernel; 1.05 with the rounded sum);
//   * O leaves through LDS as whole 256-byte rows (16 B per lane, 4 rows per store instruction) instead of 8-byte pieces of 32 rows
//     per instruction; the half-row exchange of the row maximum is a v_permlane32_swap (inline asm: the builtin aliases its two results
//     when both inputs are one value) instead of a ds_bpermute; the next item's query rows are touched two steps before an item ends.
// Semantics unchanged (reference src/dalle_mtf/models.py:275-299): unscaled fp32 logits, key > query contributes exactly 0.
struct Fk2 {
  u32x4 a, b;   // K rows r and r + 32 of a 64-key tile, 16-byte chunk of k-step kk
};
template <int OFF>
__device__ __forceinline__ void fk2_issue(Fk2& f, unsigned addr) {
#ifdef ATTN_DBG_NOLDS
  asm volatile("" : "=v"(f.a), "=v"(f.b) : "v"(addr));
  return;
#endif
  asm volatile(
      "ds_read_b128 %0, %2 offset:%3\n\t"
      "ds_read_b128 %1, %2 offset:%4"
      : "=&v"(f.a), "=&v"(f.b)
      : "v"(addr), "i"(OFF), "i"(OFF + 8192)
      : "memory");
}
template <int N>
__device__ __forceinline__ void fk2_wait(Fk2& f, Fk2& prev) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.a), "+v"(f.b), "+v"(prev.a), "+v"(prev.b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void fk2_wait(Fk2& f) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.a), "+v"(f.b) : "n"(N) : "memory");
}
__device__ __forceinline__ float half_swap_max(float v) {   // max over the two lanes (l, l ^ 32) that share a softmax row
  // v_permlane32_swap exchanges lanes 32..63 of its first operand with lanes 0..31 of the second: afterwards one register holds the lower
  // half-row's value in every lane, the other the upper half-row's.  Inline asm with the two wait states the instruction needs behind a
  // VALU write of its operands: through __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2) maps BOTH results to the first register when
  // the two inputs are copies of one value (a three-line kernel shows r[0] + 2 r[1] compiled to v_fmac v1, 2.0, v1), i.e. the maximum
  // came out as the LOWER half's only -- which ordinary scores hide (the running maximum is only a scale) and a late large score in
  // an upper-half lane turned into exp2 overflow.
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
#ifndef F2_THR
#define F2_THR 16.0f      // base-2 exponent growth of a row maximum that forces a rescale
#endif
#ifndef F2_KDEPTH
#define F2_KDEPTH 2       // K fragment pairs in flight ahead of their MFMAs
#endif
#ifndef F2_VEARLY
#define F2_VEARLY 1       // first two transposed V fragment sets requested under the last two k-steps of S
#endif
#ifndef F2_LROUNDED
#define F2_LROUNDED 1     // normalise O by the sum of the bf16-ROUNDED probabilities (the weights the MFMA actually applies sum to 1)
#endif
#ifndef F2_QPREFETCH
#define F2_QPREFETCH 1    // touch the next item's query rows (L2 prefetch) two steps before the end of an item
#endif
#ifdef ATTN_STAMP      // experiment build: per-block sums of shader-clock intervals (wave 0), tools/experiments/r06_fwd_stamps.py
#define STAMP_ARG , unsigned long long* __restrict__ stamp
#define STAMP_DECL unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = __builtin_readcyclecounter(); const unsigned long long st_t0 = st_t;
#define STAMP(i) { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += t_ - st_t; st_t = t_; }
#define STAMP_COUNT(i) { st_acc[i] += 1; }
#else
#define STAMP_ARG
#define STAMP_DECL
#define STAMP(i)
#define STAMP_COUNT(i)
#endif
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                           float* __restrict__ lse, int B, int H, int S, int perxcd STAMP_ARG) {
  extern __shared__ __attribute__((aligned(16))) char sm[];  // K slot 0 | K slot 1 | V slot 0 | V slot 1, 16 KB each
  STAMP_DECL
  const int d = H * HD, ld3 = 3 * d;
  const int T = (S + 127) / 128;
  const AttnSched sched = attn_sched(T, B * H, perxcd);
  int tile_, bh;
#if F2_QPREFETCH
  unsigned pf_reg = 0;    // landing register of the next item's Q prefetch (kept alive until that item has waited for its loads)
#endif
  for (int round = 0; attn_item(sched, round, tile_, bh); ++round) {
  const int qt = T - 1 - tile_;  // heaviest (latest) query tiles first
  const int b = bh / H, hh = bh % H;
  const int q0 = qt * 128;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));   // opaque per item (see attn_bwd_dkv_kernel)
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, g4 = lane >> 4, l16 = lane & 15;
  const int qrow = q0 + wid * 32 + r;
  const int qrow_c = qrow < S ? qrow : S - 1;
  const bf16_t* qb = qkv + (int64_t)b * S * ld3 + hh * HD;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sm;
  const int nbytes = (int)(((int64_t)(S - 1) * ld3 + HD) * 2);
#ifdef ATTN_DBG_SAMEKV   // experiment: every item streams the K / V of pair (0, 0) -> always L2-resident (prices perfect K / V locality)
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + d), 0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + 2 * d), 0, nbytes, 0x00020000);
#else
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(qb + d), 0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(qb + 2 * d), 0, nbytes, 0x00020000);
#endif

  bf16x8 qf[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) qf[kk] = *(const bf16x8*)(qb + (int64_t)qrow_c * ld3 + 16 * kk + 8 * h);

  int vo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 4, pc = c & 15;
    vo[i] = (row * ld3 + 8 * (pc ^ swz(row))) * 2;
  }
  auto stage_k = [&](int slot, int key0) {
    char* base = sm + slot * 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(rk, base + (wid * 64 + 256 * i) * 16, vo[i] + key0 * ld3 * 2);
  };
  auto stage_v = [&](int slot, int key0) {
    char* base = sm + 32768 + slot * 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(rv, base + (wid * 64 + 256 * i) * 16, vo[i] + key0 * ld3 * 2);
  };
  // K fragment address of k-step kk in slot 0: row r, chunk (2 kk + h) ^ swz(r) = ((h ^ swz(r)) ^ 2 kk) -> ONE base register, kk enters
  // as an XOR of bits 5..7 and the slot base rides in the add (v_xad_u32); the second 32 keys are an immediate offset.  Transposed V
  // fragments {keys +0..3, +8..11}: d-tile dt is an XOR of bits 6..7 of two bases.  (Eight + eight address registers hoisted out of
  // the tile loop were what spilled first; the bases are made opaque per step so that they are recomputed, one VALU each.)
  unsigned ak0 = r * 256 + ((h ^ swz(r)) << 4);
  const int rr = l16 >> 2;
  unsigned av0[2];
#pragma unroll
  for (int w2 = 0; w2 < 2; ++w2) {
    const int row = 4 * h + rr + 8 * w2;
    const int chunk = 2 * (g4 & 1) + ((l16 & 3) >> 1);
    av0[w2] = row * 256 + ((chunk ^ swz(row)) << 4) + 8 * (l16 & 1);
  }

  f32x16 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  float m2 = -1e30f, l = 0.f, mx = -1e30f;
#if F2_LROUNDED
  float lr_ = 0.f;      // row sum of the bf16-ROUNDED probabilities: what O is normalised by (l, from the unrounded ones, gives lse)
#endif

  const int qlast = (q0 + 127 < S - 1) ? q0 + 127 : S - 1;
  const int nsteps = qlast / 64 + 1;
  int jmax = (q0 + wid * 32 + 31) / 64;      // this wave's last (= its only diagonal) key tile
  jmax = jmax < nsteps ? jmax : nsteps - 1;  // (a wave whose rows all lie past S: garbage in, nothing stored)

  // causal mask of key tile jt (-1e10 additive mask == probability exactly 0) and the row maximum of the masked scores
  auto mask_tile = [&](int jt, f32x16& s0, f32x16& s1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = 64 * jt + (e & 3) + 8 * (e >> 2) + 4 * h;
      s0[e] = (key > qrow) ? -1e30f : s0[e];
      s1[e] = (key + 32 > qrow) ? -1e30f : s1[e];
    }
  };
  // (compiler-visible v_max / v_max3: an inline-asm v_max3_f32 reading MFMA results gets NO hazard protection from hipcc -- placed right behind
  // the last S MFMA it read accumulators that were still in flight; with ordinary scores a slightly wrong maximum is invisible, a late
  // spike then overflowed exp2.  The canonicalising v_max per operand that fmaxf on MFMA outputs costs rides in phase B, which has VALU slack.)
  auto row_max = [&](const f32x16& s0, const f32x16& s1) {
    float v = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int e = 1; e < 16; ++e) v = fmaxf(fmaxf(v, s0[e]), s1[e]);
    return half_swap_max(v);
  };
  // K(j + 2) and V(j + 1) on their way at the top of step j: their slots' last readers are behind the previous step's barrier
#if F2_QPREFETCH
  // The NEXT item's query rows, touched two steps before this item ends: one dword per 128-byte line (lane l: row l / 2, half l % 2 of
  // this wave's 32 rows), so that the item's real loads find them in L2.  Issued BEHIND the step's DMA and left in flight across the
  // step's barrier (vmcnt(1): loads return in order); the landing register stays reserved until the next item has waited for everything.
  int ntile, nbh;
  const bool has_next_item = attn_item(sched, round + 1, ntile, nbh);
  const int jpf = nsteps > 2 ? nsteps - 3 : 0;
  const bf16_t* pf_ptr = qkv;
  if (has_next_item) {
    int prow = (T - 1 - ntile) * 128 + wid * 32 + (lane >> 1);
    prow = prow < S ? prow : S - 1;
    pf_ptr = qkv + ((int64_t)(nbh / H) * S + prow) * ld3 + (nbh % H) * HD + 64 * (lane & 1);
  }
#endif
  auto step_dma = [&](int j) {
#ifndef ATTN_DBG_NODMA
    if (j + 2 < nsteps) stage_k(j & 1, 64 * (j + 2));
    if (j + 1 < nsteps) stage_v((j + 1) & 1, 64 * (j + 1));
#endif
#if F2_QPREFETCH
    if (j == jpf && has_next_item) asm volatile("global_load_dword %0, %1, off" : "=v"(pf_reg) : "v"(pf_ptr) : "memory");
#endif
  };
  auto step_end = [&](int j) {
#if F2_QPREFETCH
    if (j == jpf && has_next_item) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifndef ATTN_DBG_NOBARRIER
    __builtin_amdgcn_s_barrier();
#endif
  };
  // one step: softmax + P V of the tile whose scores are in (c0, c1) (key tile j, V in slot j & 1), S of tile j + 1 into (n0, n1).
  // ONE variant (two inlined copies for the two buffer roles): the wave's last step computes the scores of a tile it does not need
  // from whatever its slot holds (16 MFMAs per item, under the softmax) instead of a third code path -- with three variants x two roles
  // the register allocator shuffled 52 registers and spilled 170 around every transition, 10 us per item.
  auto body = [&](int j, f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1) {
    STAMP(8)
    step_dma(j);
    asm volatile("" : "+v"(ak0), "+v"(av0[0]), "+v"(av0[1]));
    const unsigned kb = lds0 + ((j + 1) & 1) * 16384, vb = lds0 + 32768 + (j & 1) * 16384;
#ifndef ATTN_DBG_NOCOMPUTE
    // rescale only when some row's maximum outgrew its scale by more than 2^F2_THR (exact power-of-two factors)
    const float mx2 = mx * LOG2E_F;
    if (__any(mx2 - m2 > F2_THR)) {
      const float m2n = fmaxf(m2, __builtin_ceilf(mx2));
      const float alpha = __builtin_ldexpf(1.0f, (int)fmaxf(m2 - m2n, -200.0f));
      m2 = m2n;
      l *= alpha;
#if F2_LROUNDED
      lr_ *= alpha;
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
    }
    STAMP(0)
    unsigned pw[16];
    float rs = 0.f;
#if F2_LROUNDED
    float rr = 0.f;
#endif
    // K fragments F2_KDEPTH k-steps ahead of their MFMAs, the first two transposed V fragment sets under the last two k-steps (F2_VEARLY)
    Fk2 F[F2_KDEPTH + 1];
#pragma unroll
    for (int e = 0; e < 16; ++e) n0[e] = n1[e] = 0.f;
#pragma unroll
    for (int q = 0; q < F2_KDEPTH; ++q) fk2_issue<0>(F[q], (ak0 ^ (unsigned)(q << 5)) + kb);
    Tr4 tv[2];
    unsigned avt[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      Fk2& c = F[kk % (F2_KDEPTH + 1)];
      Fk2& pv = F[(kk + F2_KDEPTH) % (F2_KDEPTH + 1)];    // the set consumed one k-step ago (its MFMAs stay above this wait)
      // LDS operations issued behind this k-step's pair: the younger K pairs and, at the end, the first V set
      constexpr int KD = F2_KDEPTH;
      const int younger_k = (kk + KD - 1 < 8 ? KD - 1 : 7 - kk) * 2;
      const int younger_v = (F2_VEARLY && kk == 7) ? 8 : 0;
      if (kk == 0) { if (younger_k + younger_v == 0) fk2_wait<0>(c); else if (younger_k == 2) fk2_wait<2>(c); }
      else {
        if (younger_k + younger_v == 0) fk2_wait<0>(c, pv);
        else if (younger_k + younger_v == 2) fk2_wait<2>(c, pv);
        else fk2_wait<8>(c, pv);
      }
      if (kk + KD < 8) fk2_issue<0>(pv, (ak0 ^ (unsigned)((kk + KD) << 5)) + kb);
      if (kk == (F2_VEARLY ? 6 : 7)) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { avt[2 * dt] = (av0[0] ^ (unsigned)(dt << 6)) + vb; avt[2 * dt + 1] = (av0[1] ^ (unsigned)(dt << 6)) + vb; }
        tr4_issue_off<0>(tv[0], avt);
      }
      if (F2_VEARLY && kk == 7) tr4_issue_off<4096>(tv[1], avt);
#ifndef ATTN_DBG_NOSMFMA
      n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c.a), qf[kk], n0, 0, 0, 0);  // S^T[key][q]
      n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c.b), qf[kk], n1, 0, 0, 0);
#else
      n0[kk] += __builtin_bit_cast(f32x4, c.a)[0]; n1[kk] += __builtin_bit_cast(f32x4, c.b)[0];
#endif
      const f32x16& cs = kk < 4 ? c0 : c1;
      float pe[4];
#ifndef ATTN_DBG_NOEXP
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pe[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(cs[4 * (kk & 3) + e], LOG2E_F, -m2));
        rs += pe[e];
      }
#else
#pragma unroll
      for (int e = 0; e < 4; ++e) pe[e] = cs[4 * (kk & 3) + e];
#endif
      pw[2 * kk] = pack2bf(pe[0], pe[1]);
      pw[2 * kk + 1] = pack2bf(pe[2], pe[3]);
      asm volatile("" : "+v"(pw[2 * kk]), "+v"(pw[2 * kk + 1]), "+v"(rs));   // pin here: otherwise the exponentials / the row-sum chain sink below the last MFMA
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    l += rs;
    STAMP(1)
    bf16x8 pb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) pb[ks] = __builtin_bit_cast(bf16x8, u32x4{pw[4 * ks], pw[4 * ks + 1], pw[4 * ks + 2], pw[4 * ks + 3]});
    // phase B: O^T[d][q] += V^T[d][key] P^T[key][q], 16-key steps: keys 16 ks + {4h..4h+3, 8+4h..8+4h+3}
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Tr4& c = tv[ks & 1];
#if F2_VEARLY
      // two sets in flight: the set of step ks + 2 is requested into this step's registers right behind the MFMAs that read them
      // (the MFMAs take their A operand when they issue; the LDS data arrive >= 64 cycles later -- as the dK/dV kernel's tdo reuse)
      if (ks < 3) tr4_waitn<8>(c, tv[(ks + 1) & 1]); else tr4_waitn<0>(c, tv[(ks + 1) & 1]);
#else
      tr4_wait(c);
      if (ks == 0) tr4_issue_off<4096>(tv[1], avt);
      if (ks == 1) tr4_issue_off<8192>(tv[0], avt);
      if (ks == 2) tr4_issue_off<12288>(tv[1], avt);
#endif
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(c.a0, c.a1), pb[ks], oacc[0], 0, 0, 0);
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(c.b0, c.b1), pb[ks], oacc[1], 0, 0, 0);
      oacc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(c.c0, c.c1), pb[ks], oacc[2], 0, 0, 0);
      oacc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(c.d0, c.d1), pb[ks], oacc[3], 0, 0, 0);
#if F2_LROUNDED      // the row sum of the ROUNDED probabilities rides here, where the VALU is idle: v_dot2c_f32_bf16 against (1, 1), one per packed word
      {
        typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
        const bf2_t one2 = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
        for (int q = 0; q < 4; ++q) rr = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, pw[4 * ks + q]), one2, rr, false);
      }
#endif
#if F2_VEARLY
      __builtin_amdgcn_sched_barrier(0);
      if (ks == 0) tr4_issue_off<8192>(tv[0], avt);
      if (ks == 1) tr4_issue_off<12288>(tv[1], avt);
#endif
    }
#if F2_LROUNDED
    lr_ += rr;
#endif
    if (j + 1 == jmax) mask_tile(j + 1, n0, n1);     // (wave-uniform)
    mx = row_max(n0, n1);
#endif
#ifdef ATTN_STAMP
    asm volatile("" : "+v"(mx));
    __builtin_amdgcn_sched_barrier(0);
#endif
    STAMP(2)
    step_end(j);
    STAMP(3)
    STAMP_COUNT(6)
  };

  f32x16 sA0, sA1, sB0, sB1;
  STAMP(8)
  // prologue: K(0), K(1), V(0); S(0) in program order
  stage_k(0, 0);
  stage_v(0, 0);
  if (nsteps > 1) stage_k(1, 64);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if F2_QPREFETCH
  asm volatile("" : "+v"(pf_reg));      // (the previous item's prefetch has landed by now: the register may be reused)
#endif
  __syncthreads();
  {
#pragma unroll
    for (int e = 0; e < 16; ++e) sA0[e] = sA1[e] = 0.f;
    Fk2 F[2];
    fk2_issue<0>(F[0], ak0 + lds0);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      Fk2& c = F[kk & 1];
      if (kk == 0) fk2_wait<0>(c); else fk2_wait<0>(c, F[(kk + 1) & 1]);
      if (kk + 1 < 8) fk2_issue<0>(F[(kk + 1) & 1], (ak0 ^ (unsigned)((kk + 1) << 5)) + lds0);
      sA0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c.a), qf[kk], sA0, 0, 0, 0);
      sA1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c.b), qf[kk], sA1, 0, 0, 0);
    }
    if (jmax == 0) mask_tile(0, sA0, sA1);
    mx = row_max(sA0, sA1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();     // every wave has read K(0): step 0 refills its slot
  STAMP(4)

  // This wave's steps 0 .. jmax (the diagonal tile's scores are masked when step jmax - 1 produces them); afterwards it only keeps
  // issuing its share of the DMA and meeting the barriers until the block's last step (the barrier counts arrivals, not program counters).
  int j = 0;
  for (;;) {
    body(j, sA0, sA1, sB0, sB1);
    if (j == jmax) break;
    ++j;
    body(j, sB0, sB1, sA0, sA1);
    if (j == jmax) break;
    ++j;
  }
  STAMP(8)
  for (j = jmax + 1; j < nsteps; ++j) {
    step_dma(j);
    step_end(j);
  }
  STAMP(7)

  // epilogue: normalise, stage this wave's 32 x 128 outputs through its private LDS strip, store whole rows
  l += __shfl_xor(l, 32, 64);
  if (h == 0 && qrow < S) lse[(int64_t)bh * S + qrow] = (m2 + __log2f(l)) * 0.6931471805599453f;
#if F2_LROUNDED
  lr_ += __shfl_xor(lr_, 32, 64);
  l = lr_;
#endif
  store_rows_via_lds(sm + wid * (32 * ROWS_PITCH), oacc, 1.f / l, o + ((int64_t)b * S + q0 + wid * 32) * d + hh * HD, d, S - (q0 + wid * 32), lane);
  __syncthreads();   // the strips are read before the next item's first DMA
  STAMP(5)
  STAMP_COUNT(9)
  }   // items
#ifdef ATTN_STAMP
  if (threadIdx.x == 0 && stamp) {
#pragma unroll
    for (int i = 0; i < 10; ++i) stamp[blockIdx.x * 12 + i] = st_acc[i];
    stamp[blockIdx.x * 12 + 10] = __builtin_readcyclecounter() - st_t0;
  }
#endif
}

static int attn_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) n = p.multiProcessorCount;
    else n = 256;
  }
  const int m = (n - g_opt_reserve_cus) & ~7;     // persistent grids: leave the reserved CUs free, keep a multiple of 8
  return m < 8 ? 8 : m;
}

extern "C" int dmi_attention_fwd(const uint16_t* qkv, uint16_t* o, float* lse, int B, int H, int S, void* stream) {
  DMI_REQUIRE(qkv && o && lse, "attention_fwd: null pointer");
  DMI_REQUIRE(B > 0 && H > 0 && S > 0 && S % 8 == 0, "attention_fwd: S must be a multiple of 8 (S=%d)", S);
  DMI_REQUIRE((int64_t)S * 3 * H * HD * 2 < 0x7fffffff, "attention_fwd: sequence too long for 32-bit buffer offsets");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * QK_STAGE);
    (void)hipFuncSetAttribute((const void*)attn_fwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * QK_STAGE);
    attr_done = true;
  }
  {
    const int items = ((S + 127) / 128) * B * H;
    const int grid = items < 2 * attn_num_cus() ? items : 2 * attn_num_cus();   // two persistent blocks per CU
    const int perxcd = g_opt_attn_xcd && (B * H) % 8 == 0 && grid % 8 == 0;
    if (g_opt_attn_fwd == 0) attn_fwd_kernel<<<dim3(grid), dim3(256), 2 * QK_STAGE, (hipStream_t)stream>>>(qkv, o, lse, B, H, S, perxcd);
#ifdef ATTN_STAMP
    else attn_fwd2_kernel<<<dim3(grid), dim3(256), 2 * QK_STAGE, (hipStream_t)stream>>>(qkv, o, lse, B, H, S, perxcd, g_dbg_buf);
#else
    else attn_fwd2_kernel<<<dim3(grid), dim3(256), 2 * QK_STAGE, (hipStream_t)stream>>>(qkv, o, lse, B, H, S, perxcd);
#endif
  }
  DMI_CHECK_LAUNCH("attention_fwd");
  return DMI_OK;
}

// =====================================================================================
// backward
// =====================================================================================
// dQ kernel v2: same K / V natural-tile DMA ring as the forward.  S^T = K Q^T and dP^T = V dO^T read tile rows with
// ds_read_b128; dQ^T += K^T dS^T takes K^T fragments from the SAME K tile with hardware transpose reads.
// The kernel also produces delta[q] = sum_d dO[q,d] O[q,d] itself (a lane pair holds the whole dO row of its query as MFMA
// fragments; O is read in the same layout) and publishes the (lse, delta) pairs the dK/dV kernel streams in -- the
// separate delta pass (17 us per layer) is gone.
template <int rowstore>      // [r06] 1: whole-row epilogue stores through LDS strips; 0: the round-2 form (A/B)
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                             const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                             float* __restrict__ delta, float* __restrict__ stats,
                                                             bf16_t* __restrict__ dqkv, int B, int H, int S, int perxcd) {
  extern __shared__ __attribute__((aligned(16))) char sm[];  // 2 x QK_STAGE
  const int d = H * HD, ld3 = 3 * d;
  const int T = (S + 127) / 128;
  const AttnSched sched = attn_sched(T, B * H, perxcd);
  int tile_, bh;
  for (int round = 0; attn_item(sched, round, tile_, bh); ++round) {   // persistent: two blocks per CU walk their item lists
  const int qt = T - 1 - tile_;
  const int b = bh / H, hh = bh % H;
  const int q0 = qt * 128;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));   // opaque per item (see attn_bwd_dkv_kernel)
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, g4 = lane >> 4, l16 = lane & 15;
  const int qrow = q0 + wid * 32 + r;
  const int qrow_c = qrow < S ? qrow : S - 1;
  const bf16_t* qb = qkv + (int64_t)b * S * ld3 + hh * HD;
  const bf16_t* dob = d_o + (int64_t)b * S * d + hh * HD;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sm;
  const int nbytes = (int)(((int64_t)(S - 1) * ld3 + HD) * 2);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(qb + d), 0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(qb + 2 * d), 0, nbytes, 0x00020000);

  bf16x8 qf[8], dof[8], of_[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    qf[kk] = *(const bf16x8*)(qb + (int64_t)qrow_c * ld3 + 16 * kk + 8 * h);
    dof[kk] = *(const bf16x8*)(dob + (int64_t)qrow_c * d + 16 * kk + 8 * h);
    of_[kk] = *(const bf16x8*)(o + ((int64_t)b * S + qrow_c) * d + hh * HD + 16 * kk + 8 * h);
  }
  const float lse_q = lse[(int64_t)bh * S + qrow_c];
  const float lse2_q = lse_q * LOG2E_F;   // p = exp2(s * log2(e) - lse * log2(e)): one fma + one v_exp_f32 per element
  float delta_q = 0.f;
  {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float fo[8], fd[8];
      unpack8(__builtin_bit_cast(u32x4, of_[kk]), fo);
      unpack8(__builtin_bit_cast(u32x4, dof[kk]), fd);
#pragma unroll
      for (int j = 0; j < 8; ++j) delta_q += fo[j] * fd[j];
    }
    delta_q += __shfl_xor(delta_q, 32, 64);   // the other half of the row lives in lane ^ 32
    if (h == 0 && qrow < S) {
      const int64_t idx = (int64_t)bh * S + qrow;
      delta[idx] = delta_q;
      stats[2 * idx] = lse2_q;   // base-2 units: the dK/dV kernel computes exp2(s * log2(e) - this)
      stats[2 * idx + 1] = delta_q;
    }
  }

  int vo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 4, pc = c & 15;
    vo[i] = (row * ld3 + 8 * (pc ^ swz(row))) * 2;
  }
  auto stage = [&](int st, int key0) {
    char* base = sm + st * QK_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma16(rk, base + (wid * 64 + 256 * i) * 16, vo[i] + key0 * ld3 * 2);
      dma16(rv, base + 16384 + (wid * 64 + 256 * i) * 16, vo[i] + key0 * ld3 * 2);
    }
  };
  int ofa[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) ofa[kk] = r * 256 + (((2 * kk + h) ^ swz(r)) << 4);
  const int rr = l16 >> 2;
  // transposed-fragment offsets of d-tile 0; d-tile dt is the same offset with bits 6..7 XORed by dt (the chunk index dt * 4 + c,
  // c < 4, enters through an XOR swizzle).  Only these two stay in registers: the eight per-d-tile offsets did not fit next to
  // the accumulators and were spilled (19 VGPRs, reloaded inside the loop; a scratch reload waits with vmcnt(0), i.e. for every LDS-DMA in flight).
  // 256 VGPRs + 80 B of scratch -> 213 VGPRs, no scratch; the same time in isolation (profiles/r04ag_kbench_attn.log), 128 -> 113 us per layer inside the step (profiles/r04_step_breakdown.txt).
  unsigned oft0[2];
#pragma unroll
  for (int w2 = 0; w2 < 2; ++w2) {
    const int row = 4 * h + rr + 8 * w2;
    const int chunk = 2 * (g4 & 1) + ((l16 & 3) >> 1);
    oft0[w2] = row * 256 + ((chunk ^ swz(row)) << 4) + 8 * (l16 & 1);
  }

  f32x16 dq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[i][e] = 0.f;

  const int qlast = (q0 + 127 < S - 1) ? q0 + 127 : S - 1;
  const int nsteps = qlast / 64 + 1;
  const int wave_qmax = q0 + wid * 32 + 31;

  auto compute = [&](int st, int j) {
    if (64 * j > wave_qmax) return;
    const char* base = sm + st * QK_STAGE;
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
      // (K / V row fragments by inline asm one k-step ahead of their MFMAs -- what the round-6 forward kernel does -- measured SLOWER here
      // than hipcc's own read-wait-MFMA schedule of these C++ loads: backward 252.5 vs 249.5 us, profiles/r06_attn_ab.log.  Not kept.)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const bf16x8 ka = *(const bf16x8*)(base + kt2 * 8192 + ofa[kk]);
        const bf16x8 va = *(const bf16x8*)(base + 16384 + kt2 * 8192 + ofa[kk]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[kk], s, 0, 0, 0);     // S^T[key][q]
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[kk], dp, 0, 0, 0);  // dP^T[key][q]
      }
      // K^T fragments in pairs of d-tiles (8 VGPRs per set, two sets in flight): step q = (s2, d-tile pair)
      Tr2 tk[2];
      unsigned ob = st * QK_STAGE + kt2 * 8192;
      asm volatile("" : "+v"(ob));   // opaque: the address arithmetic below stays inside the loop (hoisted, it is 32 registers)
      const unsigned t0 = ob + oft0[0], t1 = ob + oft0[1];   // (tile base + offset) ^ (dt << 6): the base is a multiple of 4096
      tr2_issue(tk[0], lds0 + t0, lds0 + t1, lds0 + (t0 ^ 64u), lds0 + (t1 ^ 64u));
      float ds[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = 64 * j + kt2 * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const float pe = __builtin_amdgcn_exp2f((key > qrow) ? -INFINITY : __builtin_fmaf(s[e], LOG2E_F, -lse2_q));  // unconditional exp: no per-element branches
        ds[e] = pe * (dp[e] - delta_q);
      }
      bf16x8 dsb[2];
      dsb[0] = pack_bf8(ds);
      dsb[1] = pack_bf8(ds + 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int s2 = q >> 1, dp2 = (q & 1) * 2;  // d-tiles dp2, dp2+1
        Tr2& c = tk[q & 1];
        tr2_wait(c);
        if (q < 3) {
          const int s2n = (q + 1) >> 1, dn = ((q + 1) & 1) * 2;
          const unsigned u0 = t0 + s2n * 4096, u1 = t1 + s2n * 4096;
          tr2_issue(tk[(q + 1) & 1], lds0 + (u0 ^ (dn << 6)), lds0 + (u1 ^ (dn << 6)), lds0 + (u0 ^ ((dn + 1) << 6)), lds0 + (u1 ^ ((dn + 1) << 6)));
        }
        dq[dp2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(c.a0, c.a1), dsb[s2], dq[dp2], 0, 0, 0);          // dQ^T[d][q]
        dq[dp2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(c.b0, c.b1), dsb[s2], dq[dp2 + 1], 0, 0, 0);
      }
    }
  };

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int j = 0;
  for (; j + 2 <= nsteps; j += 2) {
    stage(1, 64 * (j + 1));
    compute(0, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 2 < nsteps) stage(0, 64 * (j + 2));
    compute(1, j + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (j < nsteps) compute(0, j);

  if constexpr (rowstore) {   // [r06] whole-row stores through a wave-private LDS strip (behind a barrier: the last tile's readers are done)
    __syncthreads();
    store_rows_via_lds(sm + wid * (32 * ROWS_PITCH), dq, 1.0f, dqkv + ((int64_t)b * S + q0 + wid * 32) * ld3 + hh * HD, ld3, S - (q0 + wid * 32), lane);
  } else if (qrow < S) {
    bf16_t* op = dqkv + ((int64_t)b * S + qrow) * ld3 + hh * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int dd = dt * 32 + 8 * q4 + 4 * h;
        *(u32x2*)(op + dd) = u32x2{pack2bf(dq[dt][4 * q4], dq[dt][4 * q4 + 1]), pack2bf(dq[dt][4 * q4 + 2], dq[dt][4 * q4 + 3])};
      }
  }
  __syncthreads();   // the last step's LDS reads / the strips are done before the next item's first DMA
  }   // items
}

// ---- dK/dV kernel ---------------------------------------------------------------------------------------------------
// Block = 128 keys (4 waves x 32), loop over 32-query tiles from the diagonal down.
//   S = Q K^T, dP = dO V^T           : A operands (rows = queries) read with ds_read_b128 from the NATURAL Q / dO tiles
//   dV^T += dO^T P, dK^T += Q^T dS   : A operands (rows = d) are the TRANSPOSE of the same tiles -> ds_read_b64_tr_b16
// so only two 8-KiB tiles (+ 256 B of (lse * log2 e, delta) pairs) stream per step, by LDS-DMA into a 4-slot ring.  LDS rows
// are 256 B linear (DMA), 16-B chunk index XOR ((row&3)<<2 | (row>>2)&3) on the source side: 16 consecutive rows hit 16
// distinct chunks (b128 reads conflict-free) and the 4 rows x 64 B of a transpose-read group cover all banks once.
// Every LDS read goes through inline asm with counted lgkmcnt waits (hipcc serialises the intrinsics behind in-flight
// LDS-DMA, see gemm.hip) -- which also makes a RUN-TIME ring slot index safe: a x4 unroll with literal slot offsets
// makes LICM hoist ~100 per-slot fragment addresses out of the loop.
// Software pipeline (one wave per SIMD: 128 accumulator registers for dV / dK leave no room for a second block): a step is
//   phase A:  S / dP MFMAs of tile t+1 (slot t+1)   interleaved with   softmax + dS of tile t (one element pair per two MFMAs)
//   phase B:  dV / dK MFMAs of tile t (slot t)       interleaved with   the DMA issue of tile t+3 and the stats fetch of tile t+1
// Measured (profiles/r02_attn_dkv_phases.log): the phases-one-after-the-other version took 211 us per call at (32, 4, 1280),
// this one 160 us, bit-identical results; per-step cycle counters: A 1240, B 527 (= its 16 MFMAs), DMA wait + barrier 200.
// A two-blocks-per-CU variant (<= 256 registers, 65 KiB LDS) spilled and was not faster.
#define DKV_STAGE 16640  // Q 8192 | dO 8192 | (lse * log2 e, delta) pairs 256
#define DKV_NSTAGE 4
struct St8 {
  f32x4 v[8];   // (lse, delta) pairs of this lane's 16 query rows: v[2g], v[2g+1] = rows 8g + 4h + {0,1}, {2,3}
};
__device__ __forceinline__ void st8_issue(St8& f, unsigned a) {
  asm volatile(
      "ds_read_b128 %0, %8\n\t"
      "ds_read_b128 %1, %8 offset:16\n\t"
      "ds_read_b128 %2, %8 offset:64\n\t"
      "ds_read_b128 %3, %8 offset:80\n\t"
      "ds_read_b128 %4, %8 offset:128\n\t"
      "ds_read_b128 %5, %8 offset:144\n\t"
      "ds_read_b128 %6, %8 offset:192\n\t"
      "ds_read_b128 %7, %8 offset:208"
      : "=&v"(f.v[0]), "=&v"(f.v[1]), "=&v"(f.v[2]), "=&v"(f.v[3]), "=&v"(f.v[4]), "=&v"(f.v[5]), "=&v"(f.v[6]), "=&v"(f.v[7])
      : "v"(a)
      : "memory");
}
template <int N>
__device__ __forceinline__ void st8_wait(St8& f) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(f.v[0]), "+v"(f.v[1]), "+v"(f.v[2]), "+v"(f.v[3]), "+v"(f.v[4]), "+v"(f.v[5]), "+v"(f.v[6]), "+v"(f.v[7])
               : "n"(N)
               : "memory");
}
template <int rowstore>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                               const float* __restrict__ stats /* [B,H,S,2] (lse * log2 e, delta) */,
                                                               bf16_t* __restrict__ dqkv, int B, int H, int S, int perxcd) {
  extern __shared__ __attribute__((aligned(16))) char sm[];  // V 32768 | 4 x DKV_STAGE
  const int d = H * HD, ld3 = 3 * d;
  const AttnSched sched = attn_sched((S + 127) / 128, B * H, perxcd);
  int ktile, bh;
  // persistent loop over this block's work items; every step of an item ends with a barrier, so the next item may overwrite
  // the LDS straight away
  for (int round = 0; attn_item(sched, round, ktile, bh); ++round) {
  const int b = bh / H, hh = bh % H;
  const int key0 = ktile * 128;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));   // opaque per item: otherwise every lane constant of the body (fragment offsets, DMA offsets, ...)
                                  // is hoisted out of the item loop and stays live across it -- 716 B/lane of spills measured
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, g4 = lane >> 4, l16 = lane & 15;
  const int krow = key0 + wid * 32 + r;
  const int krow_c = krow < S ? krow : S - 1;
  const bf16_t* qb = qkv + (int64_t)b * S * ld3 + hh * HD;
  const bf16_t* kb = qb + d;
  const bf16_t* vb = qb + 2 * d;
  const bf16_t* dob = d_o + (int64_t)b * S * d + hh * HD;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sm;

  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, (int)(((int64_t)(S - 1) * ld3 + HD) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc((void*)dob, 0, (int)(((int64_t)(S - 1) * d + HD) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, (int)(((int64_t)(S - 1) * ld3 + HD) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc((void*)(stats + (int64_t)bh * S * 2), 0, S * 8, 0x00020000);

  bf16x8 kf[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) kf[kk] = *(const bf16x8*)(kb + (int64_t)krow_c * ld3 + 16 * kk + 8 * h);

#pragma unroll
  for (int i = 0; i < 8; ++i) {   // resident V tile
    const int c = tid + 256 * i, row = c >> 4, pc = c & 15;
    int gr = key0 + row;
    gr = gr < S ? gr : S - 1;
    dma16(rv, sm + (wid * 64 + 256 * i) * 16, (gr * ld3 + 8 * (pc ^ swz(row))) * 2);
  }
  int voq[2], vod[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i, row = c >> 4, pc = c & 15;
    voq[i] = (row * ld3 + 8 * (pc ^ swz(row))) * 2;
    vod[i] = (row * d + 8 * (pc ^ swz(row))) * 2;
  }
  auto stage = [&](int st, int q0) {
    char* base = sm + 32768 + st * DKV_STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      dma16(rq, base + (wid * 64 + 256 * i) * 16, voq[i] + q0 * ld3 * 2);
      dma16(rdo, base + 8192 + (wid * 64 + 256 * i) * 16, vod[i] + q0 * d * 2);
    }
    dma4(rst, base + 16384, (q0 * 2 + lane) * 4);
  };
  auto stage_q = [&](int st, int q0, int i) {   // half of a stage's Q + dO chunks / its stats strip: issued between the dV / dK MFMAs
    char* base = sm + 32768 + st * DKV_STAGE;
    dma16(rq, base + (wid * 64 + 256 * i) * 16, voq[i] + q0 * ld3 * 2);
    dma16(rdo, base + 8192 + (wid * 64 + 256 * i) * 16, vod[i] + q0 * d * 2);
  };
  auto stage_st = [&](int st, int q0) {
    dma4(rst, sm + 32768 + st * DKV_STAGE + 16384, (q0 * 2 + lane) * 4);
  };
  int ofa[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) ofa[kk] = r * 256 + (((2 * kk + h) ^ swz(r)) << 4);
  const int rr = l16 >> 2;
  unsigned oft[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int w2 = 0; w2 < 2; ++w2) {
      const int row = 4 * h + rr + 8 * w2;
      const int chunk = dt * 4 + 2 * (g4 & 1) + ((l16 & 3) >> 1);
      oft[dt][w2] = row * 256 + ((chunk ^ swz(row)) << 4) + 8 * (l16 & 1);
    }

  f32x16 dv[4], dk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dv[i][e] = dk[i][e] = 0.f;
  f32x16 sA, dpA, sB, dpB;
#pragma unroll
  for (int e = 0; e < 16; ++e) sA[e] = dpA[e] = sB[e] = dpB[e] = 0.f;

  const int nqi = (S + 31) / 32;
  const int qi0 = key0 / 32;
  const int nsteps = nqi - qi0;

  unsigned aq[8];   // natural-fragment addresses in stage slot 0; the same offsets address this wave's rows of the resident V tile
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) aq[kk] = lds0 + 32768 + ofa[kk];
  const unsigned vrel = (unsigned)(wid * 8192 - 32768);
  // S / dP of the tile in ring slot st, nothing interleaved (first tile of a block)
  auto sdp_only = [&](int st, f32x16& s, f32x16& dp) {
    const unsigned so = st * DKV_STAGE;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
    Fr3 F[3];
    fr3_issue<0>(F[0], aq[0] + so, aq[0] + vrel);
    fr3_issue<0>(F[1], aq[1] + so, aq[1] + vrel);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      Fr3& c = F[kk % 3];
      if (kk == 0) fr3_wait<3>(c);
      else if (kk < 7) fr3_wait<3>(c, F[(kk + 2) % 3]);
      else fr3_wait<0>(c, F[(kk + 2) % 3]);
      if (kk + 2 < 8) fr3_issue<0>(F[(kk + 2) % 3], aq[kk + 2] + so, aq[kk + 2] + vrel);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c.q), kf[kk], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c.d), __builtin_bit_cast(bf16x8, c.v), dp, 0, 0, 0);
    }
  };
  // softmax + dS of the accumulator-element pair (2i, 2i+1) = query rows (e&3) + 8(e>>2) + 4h of the tile, this lane's key,
  // packed to bf16 (pw / dw = the B operands of the dV / dK MFMAs); the stats hold (lse * log2(e), delta) pairs.
  // MASK: only the four tiles that touch the block's diagonal need the causal predicate.  Rows past S need no mask: their
  // Q / dO rows and stats read as zeros (buffer bounds), so dS = 0 and P multiplies a zero dO row.
  auto softmax_pair = [&](auto mask_c, int i, const St8& stt, const f32x16& s, const f32x16& dp, int kd, unsigned* pw, unsigned* dw) {
    constexpr bool MASK = decltype(mask_c)::value;
#ifdef ATTN_DBG_DKV_NOSOFTMAX     // experiment: what the softmax / dS VALU chain of phase A costs (upper bound of any re-placement)
    pw[i] = pack2bf(s[2 * i], s[2 * i + 1]);
    dw[i] = pack2bf(dp[2 * i], dp[2 * i + 1]);
    asm volatile("" : "+v"(pw[i]), "+v"(dw[i]));
    return;
#endif
    const int g = i >> 1;                       // elements 4g + {0,1} (i even) or 4g + {2,3} (i odd)
    const f32x4 sv = stt.v[i];                  // (l2, delta) of rows 8g + 4h + 2(i&1) + {0, 1}
    float x0 = __builtin_fmaf(s[2 * i], LOG2E_F, -sv[0]);
    float x1 = __builtin_fmaf(s[2 * i + 1], LOG2E_F, -sv[2]);
    if constexpr (MASK) {
      x0 = (kd > 8 * g + 2 * (i & 1)) ? -INFINITY : x0;   // key > query
      x1 = (kd > 8 * g + 2 * (i & 1) + 1) ? -INFINITY : x1;
    }
    const float p0 = __builtin_amdgcn_exp2f(x0), p1 = __builtin_amdgcn_exp2f(x1);
    const float q0 = p0 * (dp[2 * i] - sv[1]), q1 = p1 * (dp[2 * i + 1] - sv[3]);
    pw[i] = pack2bf(p0, p1);
    dw[i] = pack2bf(q0, q1);
    asm volatile("" : "+v"(pw[i]), "+v"(dw[i]));   // pin here: otherwise the exponentials sink below the last MFMA
  };
  St8 stt;       // stats of the CURRENT tile: fetched during the previous step's dV / dK phase (the first one before the loop)
  Tr4 tdo, tq;   // transposed dO / Q fragments of the current tile's first 16 queries: fetched at the top of the step
  auto body = [&](int st, int qi, int t, f32x16& s, f32x16& dp, f32x16& sn, f32x16& dpn, auto has_next_c, auto mask_c) {
    constexpr bool has_next = decltype(has_next_c)::value;
    const unsigned lb = lds0 + 32768 + st * DKV_STAGE;
    const unsigned so = ((st + 1) & 3) * DKV_STAGE;
    // eight transposed-fragment base addresses of this slot; the four fragment sets of a step (dO^T / Q^T x first / second 16
    // queries) are immediate offsets from them
    const unsigned ta[8] = {lb + oft[0][0], lb + oft[0][1], lb + oft[1][0], lb + oft[1][1], lb + oft[2][0], lb + oft[2][1], lb + oft[3][0], lb + oft[3][1]};
    tr4_issue_off<8192>(tdo, ta);
    tr4_issue_off<0>(tq, ta);
    unsigned pw[8], dw[8];
    const int kd = krow - 32 * qi - 4 * h;   // key - (first query row of this lane in the tile)
    if constexpr (has_next) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sn[e] = dpn[e] = 0.f;
      Fr3 F[2];   // one operand triple ahead of the MFMAs, issued right after the wait for the current one: the two MFMAs + the
                  // softmax pair of a k-step cover the read; deeper rings (2, 3 triples ahead) measured the same or 1 % slower
      fr3_issue<0>(F[0], aq[0] + so, aq[0] + vrel);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        Fr3& c = F[kk & 1];
        if (kk == 0) fr3_wait<0>(c); else fr3_wait<0>(c, F[(kk + 1) & 1]);
        if (kk + 1 < 8) fr3_issue<0>(F[(kk + 1) & 1], aq[kk + 1] + so, aq[kk + 1] + vrel);
        sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c.q), kf[kk], sn, 0, 0, 0);
        dpn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c.d), __builtin_bit_cast(bf16x8, c.v), dpn, 0, 0, 0);
        softmax_pair(mask_c, kk, stt, s, dp, kd, pw, dw);
        // one MFMA, then half of the pair's VALU chain (two independent elements interleaved: a single resident wave has
        // nothing else to cover the VALU / transcendental latencies), twice
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) softmax_pair(mask_c, i, stt, s, dp, kd, pw, dw);
    }
    bf16x8 pb[2], dsb[2];
    pb[0] = __builtin_bit_cast(bf16x8, u32x4{pw[0], pw[1], pw[2], pw[3]});
    pb[1] = __builtin_bit_cast(bf16x8, u32x4{pw[4], pw[5], pw[6], pw[7]});
    dsb[0] = __builtin_bit_cast(bf16x8, u32x4{dw[0], dw[1], dw[2], dw[3]});
    dsb[1] = __builtin_bit_cast(bf16x8, u32x4{dw[4], dw[5], dw[6], dw[7]});
    // dV / dK phase: 16 MFMAs, no VALU work of its own -> the DMA issue of tile t+3 and the stats fetch of tile t+1 ride here.
    // Fragment sets: (tdo, tq) hold the first 16 queries (in since the top of the step); t2 takes dO^T of the second 16,
    // then tdo's registers are reused for Q^T of the second 16.
    const bool do_dma = t + 3 < nsteps;
    const int st3 = (st + 3) & 3, q3 = 32 * (qi + 3);
    Tr4 t2;
    {
      tr4_wait(tq);   // (tdo, tq) of the first 16 queries
      tr4_issue_off<8192 + 4096>(t2, ta);
      dv[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(tdo.a0, tdo.a1), pb[0], dv[0], 0, 0, 0);   // dV^T[d][key]
      dv[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(tdo.b0, tdo.b1), pb[0], dv[1], 0, 0, 0);
      if (do_dma) stage_q(st3, q3, 0);
      dv[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(tdo.c0, tdo.c1), pb[0], dv[2], 0, 0, 0);
      dv[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(tdo.d0, tdo.d1), pb[0], dv[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      tr4_issue_off<4096>(tdo, ta);   // Q^T, second 16
      dk[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(tq.a0, tq.a1), dsb[0], dk[0], 0, 0, 0);    // dK^T[d][key]
      dk[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(tq.b0, tq.b1), dsb[0], dk[1], 0, 0, 0);
      if (do_dma) stage_q(st3, q3, 1);
      dk[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(tq.c0, tq.c1), dsb[0], dk[2], 0, 0, 0);
      dk[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(tq.d0, tq.d1), dsb[0], dk[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      tr4_wait8(t2, tq);   // t2 (older) is in, the 8 reads of the new tdo may still be in flight
      if constexpr (has_next) st8_issue(stt, lds0 + 32768 + so + 16384 + 32 * h);   // stats of the next tile
      dv[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(t2.a0, t2.a1), pb[1], dv[0], 0, 0, 0);
      dv[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(t2.b0, t2.b1), pb[1], dv[1], 0, 0, 0);
      if (do_dma) stage_st(st3, q3);
      dv[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(t2.c0, t2.c1), pb[1], dv[2], 0, 0, 0);
      dv[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(t2.d0, t2.d1), pb[1], dv[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (has_next) tr4_wait8(tdo, t2); else tr4_wait(tdo);   // Q^T of the second 16 (the stats reads stay in flight)
      dk[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(tdo.a0, tdo.a1), dsb[1], dk[0], 0, 0, 0);
      dk[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(tdo.b0, tdo.b1), dsb[1], dk[1], 0, 0, 0);
      dk[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(tdo.c0, tdo.c1), dsb[1], dk[2], 0, 0, 0);
      dk[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat2(tdo.d0, tdo.d1), dsb[1], dk[3], 0, 0, 0);
      if constexpr (has_next) st8_wait<0>(stt);
    }
  };

  // Ring: at the barrier that opens step t, tiles t and t+1 have landed (every wave waited for its own DMAs), tile t+2 is
  // in flight; step t issues tile t+3 into the slot of tile t-1, whose last reader (dV / dK of step t-1) is behind the barrier.
#define DKV_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define DKV_BARRIER()                                  \
  do {                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                      \
  } while (0)
  if (nsteps > 0) stage(0, 32 * qi0);
  if (nsteps > 1) stage(1, 32 * (qi0 + 1));
  if (nsteps > 2) stage(2, 32 * (qi0 + 2));
  if (nsteps > 2) DKV_WAIT(5); else DKV_WAIT(0);
  DKV_BARRIER();
  if (nsteps > 0) {
    st8_issue(stt, lds0 + 32768 + 16384 + 32 * h);
    sdp_only(0, sA, dpA);   // (its final lgkmcnt(0) covers the stats reads)
  }
  int t = 0;
#define DKV_STEP(CS, CD, NS, ND, NEXT, MASK)                                \
  {                                                                         \
    body(t & 3, qi0 + t, t, CS, CD, NS, ND, std::integral_constant<bool, NEXT>{}, std::integral_constant<bool, MASK>{}); \
    if (nsteps - 1 - t >= 3) DKV_WAIT(5); else DKV_WAIT(0);                 \
    DKV_BARRIER();                                                          \
    ++t;                                                                    \
  }
  // tiles 0..3 of a block touch its diagonal (masked variant); pairs of steps keep the A / B accumulator roles fixed
  while (t < 4 && t + 3 <= nsteps) {
    DKV_STEP(sA, dpA, sB, dpB, true, true) DKV_STEP(sB, dpB, sA, dpA, true, true)
  }
  while (t + 3 <= nsteps) {   // two steps that both have a successor
    DKV_STEP(sA, dpA, sB, dpB, true, false) DKV_STEP(sB, dpB, sA, dpA, true, false)
  }
  if (nsteps - t == 2) {
    DKV_STEP(sA, dpA, sB, dpB, true, true) DKV_STEP(sB, dpB, sA, dpA, false, true)
  } else if (nsteps - t == 1) {
    DKV_STEP(sA, dpA, sB, dpB, false, true)
  }
#undef DKV_STEP
#undef DKV_WAIT
#undef DKV_BARRIER

  if constexpr (rowstore) {   // [r06] whole-row stores (every step ended with a barrier: the ring and the V tile have no readers left)
    bf16_t* gk = dqkv + ((int64_t)b * S + key0 + wid * 32) * ld3 + d + hh * HD;
    char* strip = sm + wid * (2 * 32 * ROWS_PITCH);
    store_rows_via_lds(strip, dk, 1.0f, gk, ld3, S - (key0 + wid * 32), lane);
    store_rows_via_lds(strip + 32 * ROWS_PITCH, dv, 1.0f, gk + d, ld3, S - (key0 + wid * 32), lane);
    __syncthreads();   // the strips are read before the next item's DMA overwrites them
  } else if (krow < S) {
    bf16_t* okp = dqkv + ((int64_t)b * S + krow) * ld3 + d + hh * HD;
    bf16_t* ovp = okp + d;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int dd = dt * 32 + 8 * q4 + 4 * h;
        *(u32x2*)(okp + dd) = u32x2{pack2bf(dk[dt][4 * q4], dk[dt][4 * q4 + 1]), pack2bf(dk[dt][4 * q4 + 2], dk[dt][4 * q4 + 3])};
        *(u32x2*)(ovp + dd) = u32x2{pack2bf(dv[dt][4 * q4], dv[dt][4 * q4 + 1]), pack2bf(dv[dt][4 * q4 + 2], dv[dt][4 * q4 + 3])};
      }
  }
  }   // items
}

// A two-pass form -- the dQ pass also writes P and dS (bf16, causal half of [2][B*H][S][S]) and the dK / dV pass streams them: two
// products per tile instead of four and a softmax -- was built, passed the same parity tests and measured SLOWER: 360 us against
// 268 us per layer at (32, 4, 1280), step 15.42 -> 15.78 ms (profiles/r04ah_kbench_attn.log, r04ah_ab_two_pass.log; with the P / dS
// pieces stored as the accumulators hold them, 8 bytes to 32 rows per instruction, 405 us, and 905 / 1820 us with write-through
// stores: r04ag_*).  The pair moves 0.84 GB per layer through HBM; recomputing S and dP on the matrix cores costs less than that
// traffic.  The code is on the git branch `r04-attn-two-pass`.
extern "C" int dmi_attention_bwd(const uint16_t* qkv, const uint16_t* o, const uint16_t* d_o, const float* lse, float* delta,
                                 uint16_t* dqkv, int B, int H, int S, void* stream) {
  DMI_REQUIRE(qkv && o && d_o && lse && delta && dqkv, "attention_bwd: null pointer");
  DMI_REQUIRE(B > 0 && H > 0 && S > 0 && S % 8 == 0, "attention_bwd: S must be a multiple of 8 (S=%d)", S);
  DMI_REQUIRE((int64_t)S * 3 * H * HD * 2 < 0x7fffffff, "attention_bwd: sequence too long for 32-bit buffer offsets");
  hipStream_t st = (hipStream_t)stream;
  float* stats = delta + (int64_t)B * H * S;  // delta scratch is [3][B,H,S]: delta | (lse, delta) pairs
  static bool attr_done = false;
  const int shm = 32768 + DKV_NSTAGE * DKV_STAGE;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * QK_STAGE);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * QK_STAGE);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    attr_done = true;
  }
  const int items = ((S + 127) / 128) * B * H;
  {
    const int grid = items < 2 * attn_num_cus() ? items : 2 * attn_num_cus();   // two persistent blocks per CU
    const int perxcd = g_opt_attn_xcd && (B * H) % 8 == 0 && grid % 8 == 0;
    if (g_opt_attn_bwd & 1) attn_bwd_dq_kernel<1><<<dim3(grid), dim3(256), 2 * QK_STAGE, st>>>(qkv, o, d_o, lse, delta, stats, dqkv, B, H, S, perxcd);
    else attn_bwd_dq_kernel<0><<<dim3(grid), dim3(256), 2 * QK_STAGE, st>>>(qkv, o, d_o, lse, delta, stats, dqkv, B, H, S, perxcd);
  }
  DMI_CHECK_LAUNCH("attention_bwd_dq");
  {
    const int grid = items < attn_num_cus() ? items : attn_num_cus();   // one persistent block per CU
    const int perxcd = g_opt_attn_xcd && (B * H) % 8 == 0 && grid % 8 == 0;
    if (g_opt_attn_bwd & 1) attn_bwd_dkv_kernel<1><<<dim3(grid), dim3(256), shm, st>>>(qkv, d_o, stats, dqkv, B, H, S, perxcd);
    else attn_bwd_dkv_kernel<0><<<dim3(grid), dim3(256), shm, st>>>(qkv, d_o, stats, dqkv, B, H, S, perxcd);
  }
  DMI_CHECK_LAUNCH("attention_bwd_dkv");
  return DMI_OK;
}

// =====================================================================================
// incremental decode: ONE query position against the key/value cache  
// =====================================================================================
// The reference scaffolds incremental inference (src/dalle_mtf/models.py:246-254: the new k, v replace row `position - 1`
// of the stored states; :281-285: the mask row of that position selects keys <= position) but never finishes the predict
// path.  Here the cache IS the forward pass's [B*S, 3d] projection buffer: the caller's QKV GEMM writes row `pos` of every
// batch element in place (row pitch S*3d), then this kernel computes, per (batch, head),
//     o = softmax(q_pos . K[0..pos]^T) V[0..pos]           (unscaled logits, as everywhere on this path)
// HBM-bound (2 * (pos+1) * 256 B per head): one block per (b, h), a wave per 64-key chunk, online softmax per wave, the sixteen
// waves merge through LDS.  (History at B*H = 128, S = 1280: four waves, key = lane, one value row in flight per wave: 46 us;
// sixteen waves, eight value rows in flight: 26 us; the row-coalesced form below: see profiles/r03_decode_kernel_stats.csv.)
// P is rounded to bf16 before P.V and the row sum is taken over the unrounded fp32 values -- the same places where the tiled
// forward kernel rounds -- so decode logits track full-forward logits to bf16 noise.
//
// Graph-replayable form (one captured HIP graph serves every position of the sampling loop): `pos_dev` -- the position read
// from device memory -- and `fresh` -- a [B, 3d] staging buffer at a FIXED address that the step's QKV GEMM wrote (a kernel
// argument baked into a graph cannot move along the cache).  With `fresh` the block takes q from it, stores the head's
// q | k | v into cache row `pos` for the steps to come, and reads key / value `pos` from the staging row itself (no reliance
// on the block seeing its own global stores).
#define DEC_WAVES 16   // 1024 threads: a (batch, head) pair streams <= S keys through ONE block, so the block must itself hold the loads in flight
// Lane (c = lane & 15, g = lane >> 4) owns head dims [8c, 8c + 8) and, per 64-key chunk, keys 4i + g (i = 0..15): every load
// instruction of the wave fetches four whole 256-B rows (16 B per lane) -- the first form read one row per LANE, 16 B at a time,
// and re-fetched each 128-B line from L2 up to eight times.  The 16 partial dot products per lane are combined across the 16
// lanes of a group by a reduce-scatter butterfly (15 exchanges): lane (c, g) ends up with the whole score of key 4c + g.
__global__ __launch_bounds__(64 * DEC_WAVES) void attn_decode_kernel(bf16_t* qkv, const bf16_t* __restrict__ fresh, bf16_t* __restrict__ o,
                                                                     int H, int S, int pos_arg, const int* __restrict__ pos_dev) {
  __shared__ float ps[DEC_WAVES][64];
  __shared__ float red_m[DEC_WAVES], red_l[DEC_WAVES];
  __shared__ float oacc[DEC_WAVES][HD];
  const int bh = blockIdx.x, b = bh / H, hh = bh % H;
  const int d = H * HD;
  const int64_t ld3 = 3 * (int64_t)d;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int pos = pos_dev ? *pos_dev : pos_arg;
  if (pos < 0 || pos >= S) return;                     // (block-uniform; the host checks the by-value form)
  bf16_t* base = qkv + (int64_t)b * S * ld3 + hh * HD;
  const bf16_t* fr = fresh ? fresh + (int64_t)b * ld3 + hh * HD : nullptr;
  float q[8];
  unpack8(*(const u32x4*)((fr ? fr : base + (int64_t)pos * ld3) + 8 * c), q);
  if (fr && threadIdx.x < 3 * HD / 8) {   // q | k | v of this head: 3 x 256 B -> cache row pos (read back by no one in this launch)
    const int part = threadIdx.x / (HD / 8), ch = threadIdx.x % (HD / 8);
    *(u32x4*)(base + (int64_t)pos * ld3 + part * d + ch * 8) = *(const u32x4*)(fr + part * d + ch * 8);
  }
  float m = -1e30f, l = 0.f;
  float oa[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int ch0 = wid; ch0 * 64 <= pos; ch0 += DEC_WAVES) {
    const int k0 = ch0 * 64;
    auto row_ptr = [&](int i, int which) -> const u32x4* {   // key k0 + 4i + g, clamped to pos; row pos comes from the staging buffer
      int key = k0 + 4 * i + g;
      key = key < pos ? key : pos;
      return (const u32x4*)((fr && key == pos) ? fr + which * d + 8 * c : base + which * d + (int64_t)key * ld3 + 8 * c);
    };
    u32x4 raw[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) raw[i] = *row_ptr(i, 1);
    float p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float f[8];
      unpack8(raw[i], f);
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) t = __builtin_fmaf(f[j], q[j], t);
      p[i] = t;
    }
    // reduce-scatter over the 16 lanes of a group: after the stage with mask w, a lane keeps the half of its values whose
    // index has bit w equal to its own bit w of c
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1) {
      const bool up = (c & w) != 0;
#pragma unroll
      for (int j = 0; j < w; ++j) {
        const float send = up ? p[j] : p[j + w];
        const float keep = up ? p[j + w] : p[j];
        p[j] = keep + __shfl_xor(send, w, 64);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) raw[i] = *row_ptr(i, 2);     // value rows requested before the softmax' wave reductions
    const int key = k0 + 4 * c + g;
    const bool valid = key <= pos;
    const float sc = valid ? p[0] : -1e30f;
    const float mn = fmaxf(m, wave_max(sc));
    const float alpha = __expf(m - mn);
    const float pe = valid ? __expf(sc - mn) : 0.f;
    l = l * alpha + wave_sum(pe);
    m = mn;
    ps[wid][4 * c + g] = bf2f(f2bf(pe));
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private strip: in-order LDS, no block barrier needed
#pragma unroll
    for (int j = 0; j < 8; ++j) oa[j] *= alpha;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float f[8];
      unpack8(raw[i], f);
      const float pj = ps[wid][4 * i + g];               // 0 for keys past pos (their rows were clamped to row pos)
#pragma unroll
      for (int j = 0; j < 8; ++j) oa[j] = __builtin_fmaf(pj, f[j], oa[j]);
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {                          // the four lane groups hold disjoint key subsets of the same dims
    oa[j] += __shfl_xor(oa[j], 16, 64);
    oa[j] += __shfl_xor(oa[j], 32, 64);
  }
  if (lane == 0) { red_m[wid] = m; red_l[wid] = l; }
  if (g == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) oacc[wid][8 * c + j] = oa[j];
  }
  __syncthreads();
  if (wid == 0) {
    float M = red_m[0];
#pragma unroll
    for (int w = 1; w < DEC_WAVES; ++w) M = fmaxf(M, red_m[w]);
    float L = 0.f, a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int w = 0; w < DEC_WAVES; ++w) {      // fixed order: deterministic
      const float sc = __expf(red_m[w] - M);
      L += red_l[w] * sc;
      a0 += oacc[w][2 * lane] * sc;
      a1 += oacc[w][2 * lane + 1] * sc;
    }
    const float inv = 1.f / L;
    *(unsigned*)(o + (int64_t)b * d + hh * HD + 2 * lane) = pack2bf(a0 * inv, a1 * inv);
  }
}

extern "C" int dmi_attention_decode(uint16_t* qkv, const uint16_t* fresh, uint16_t* o, int B, int H, int S, int pos, const int* pos_dev,
                                    void* stream) {
  DMI_REQUIRE(qkv && o, "attention_decode: null pointer");
  DMI_REQUIRE(B > 0 && H > 0 && S > 0, "attention_decode: bad shape");
  DMI_REQUIRE(pos_dev || (pos >= 0 && pos < S), "attention_decode: need 0 <= pos < S (pos=%d, S=%d)", pos, S);
  attn_decode_kernel<<<dim3((unsigned)(B * H)), dim3(64 * DEC_WAVES), 0, (hipStream_t)stream>>>(qkv, fresh, o, H, S, pos, pos_dev);
  DMI_CHECK_LAUNCH("attention_decode");
  return DMI_OK;
}
