// elementwise.hip -- HBM-bound kernels of the DALL-E train step (SURVEY.md §2.2 K1,K2,K7-K10).
// All loads/stores are 16 B per lane (8 bf16 / 4 fp32), one wave64 per row where rows are short,
// wave-shuffle reductions, deterministic two-stage reductions for gradients of shared parameters.
#include "common.h"
#include <stdarg.h>

// ------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "ok";
void dmi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* dmi_last_error_string(void) { return g_err; }
extern "C" int dmi_version(void) { return 100; }

// =====================================================================================
// K1 embedding   (src/dalle_mtf/models.py:186-219)
// =====================================================================================
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int* __restrict__ tokens,
                                                        const bf16_t* __restrict__ wte,
                                                        const bf16_t* __restrict__ wpe, bf16_t* __restrict__ x,
                                                        int64_t rows, int S, int d, int vocab, const int* __restrict__ pos_dev) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wid;
  if (row >= rows) return;
  int tok = tokens[row];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
  int s = (int)(row % S);
  if (pos_dev) {                                        // the decode step: every row sits at that one position (clamped to the table)
    s = *pos_dev;
    s = s < 0 ? 0 : (s >= S ? S - 1 : s);
  }
  const u32x4* a = (const u32x4*)(wte + (int64_t)tok * d);
  const u32x4* p = (const u32x4*)(wpe + (int64_t)s * d);
  u32x4* o = (u32x4*)(x + row * d);
  for (int c = lane; c < d / 8; c += 64) {
    float fa[8], fp[8];
    unpack8(a[c], fa);
    unpack8(p[c], fp);
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] += fp[j];
    o[c] = pack8(fa);
  }
}

extern "C" int dmi_embed_fwd(const int32_t* tokens, const uint16_t* wte, const uint16_t* wpe, uint16_t* x,
                             int64_t rows, int S, int d, int vocab, const int* pos_dev, void* stream) {
  DMI_REQUIRE(tokens && wte && wpe && x, "embed_fwd: null pointer");
  DMI_REQUIRE(d % 8 == 0 && rows > 0 && S > 0, "embed_fwd: d %% 8 != 0 or empty (d=%d rows=%lld)", d, (long long)rows);
  embed_fwd_kernel<<<dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream>>>(tokens, wte, wpe, x, rows, S, d, vocab, pos_dev);
  DMI_CHECK_LAUNCH("embed_fwd");
  return DMI_OK;
}

// dwpe[s, :] = sum_b dx[b, s, :]      (one wave per position; deterministic)
__global__ __launch_bounds__(256) void embed_bwd_wpe_kernel(const bf16_t* __restrict__ dx, float* __restrict__ dwpe,
                                                            int B, int S, int d) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int s = blockIdx.x * 4 + wid;
  if (s >= S) return;
  for (int c = lane; c < d / 8; c += 64) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < B; ++b) {
      float f[8];
      unpack8(*(const u32x4*)(dx + ((int64_t)b * S + s) * d + c * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
    float* o = dwpe + (int64_t)s * d + c * 8;
    *(f32x4*)o = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *(f32x4*)(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
  }
}
// ---- stable sort of the token ids (index plumbing for the scatter-add below) -----------------------------------------
// LSD radix sort with 4-bit digits over SORT_T = 2048 threads (8 blocks): thread g owns the contiguous chunk
// [g*cpt, (g+1)*cpt) of the current order.  Per pass three short launches: COUNT (every thread counts its digits into its own
// column of hist[16][SORT_T], no atomics), SCAN (one block: exclusive scan of the 16 * SORT_T counts in (digit, thread) order =
// the destinations), SCATTER (every thread writes its chunk in order) -> stable and deterministic.  ceil(log2(vocab) / 4)
// passes ping-pong between (sorted, perm) and the workspace; the last pass lands in the outputs.
// [r04] Rounds 2-3 ran this as ONE 1024-thread block for 0.62 ms on a side stream "hidden" under the forward.  It was not
// hidden: the forward's kernels are persistent (one or two blocks per CU, each with its own tile list), and the block whose CU
// the sort occupied started ~0.6 ms late -- a per-launch trace showed layer 0's forward taking 687 us instead of 410 us
// (attention 210 us instead of 70).  Moving it under the head's weight gradient moved the damage there (+340 us).  As 12
// launches of 3-5 us it disturbs nothing.
#define SORT_T 2048
#define SORT_BLK 256
__device__ __forceinline__ int sort_clamp(int k, int vocab) { return k < 0 ? 0 : (k >= vocab ? vocab - 1 : k); }
__global__ __launch_bounds__(SORT_BLK) void sort_count_kernel(const int* __restrict__ keys, int* __restrict__ hist, int n, int vocab,
                                                              int shift, int cpt) {
  __shared__ int col[16][SORT_BLK];
  const int t = threadIdx.x, g = blockIdx.x * SORT_BLK + t;
#pragma unroll
  for (int d = 0; d < 16; ++d) col[d][t] = 0;
  const int i0 = (int64_t)g * cpt < n ? g * cpt : n;
  const int i1 = i0 + cpt < n ? i0 + cpt : n;
  for (int i = i0; i < i1; ++i) col[(sort_clamp(keys[i], vocab) >> shift) & 15][t] += 1;
#pragma unroll
  for (int d = 0; d < 16; ++d) hist[d * SORT_T + g] = col[d][t];
}
// exclusive scan of hist[16 * SORT_T] in linear (digit-major) order, in place; thread t owns 16 * SORT_T / 1024 = 32 entries
__global__ __launch_bounds__(1024) void sort_scan_kernel(int* __restrict__ hist) {
  __shared__ int wtot[16];
  constexpr int PER = 16 * SORT_T / 1024;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  int* h = hist + t * PER;
  int tot = 0;
  for (int j = 0; j < PER; j += 4) {
    const int4 v = *(const int4*)(h + j);
    tot += (v.x + v.y) + (v.z + v.w);
  }
  int inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wtot[wid] = inc;
  __syncthreads();
  int base = inc - tot;
  for (int w = 0; w < wid; ++w) base += wtot[w];
  for (int j = 0; j < PER; j += 4) {
    int4 v = *(const int4*)(h + j);
    const int4 o = {base, base + v.x, base + v.x + v.y, base + v.x + v.y + v.z};
    base += (v.x + v.y) + (v.z + v.w);
    *(int4*)(h + j) = o;
  }
}
__global__ __launch_bounds__(SORT_BLK) void sort_scatter_kernel(const int* __restrict__ keys, const int* __restrict__ perm_in,
                                                                int* __restrict__ keys_out, int* __restrict__ perm_out,
                                                                const int* __restrict__ hist, int n, int vocab, int shift, int cpt) {
  __shared__ int col[16][SORT_BLK];
  const int t = threadIdx.x, g = blockIdx.x * SORT_BLK + t;
#pragma unroll
  for (int d = 0; d < 16; ++d) col[d][t] = hist[d * SORT_T + g];
  const int i0 = (int64_t)g * cpt < n ? g * cpt : n;
  const int i1 = i0 + cpt < n ? i0 + cpt : n;
  for (int i = i0; i < i1; ++i) {
    const int k = sort_clamp(keys[i], vocab);
    const int d = (k >> shift) & 15;
    const int pos = col[d][t];
    col[d][t] = pos + 1;
    keys_out[pos] = k;
    perm_out[pos] = perm_in ? perm_in[i] : i;
  }
}
extern "C" int64_t dmi_sort_tokens_workspace_bytes(int64_t n) { return 2 * n * 4 + (int64_t)16 * SORT_T * 4 + 256; }
extern "C" int dmi_sort_tokens(const int32_t* tokens, int32_t* sorted_tokens, int32_t* perm, int64_t n, int vocab,
                               void* workspace, void* stream) {
  DMI_REQUIRE(tokens && sorted_tokens && perm && workspace, "sort_tokens: null pointer");
  DMI_REQUIRE(n > 0 && n < (1ll << 30) && vocab > 0, "sort_tokens: bad sizes");
  int bits = 1;
  while ((1ll << bits) < vocab) ++bits;
  const int npass = (bits + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  int* tk = (int*)workspace;           // [n] keys | [n] perm | [16 * SORT_T] counts
  int* tp = tk + n;
  int* hist = (int*)(((uintptr_t)(tp + n) + 15) & ~(uintptr_t)15);      // 16-B aligned (vector loads in the scan)
  const int cpt = (int)((n + SORT_T - 1) / SORT_T);
  const dim3 grid(SORT_T / SORT_BLK), blk(SORT_BLK);
  for (int p = 0; p < npass; ++p) {
    const bool to_out = ((npass - 1 - p) & 1) == 0;          // the last pass lands in the outputs
    const int* sk = (p == 0) ? tokens : (to_out ? tk : sorted_tokens);
    const int* sp = (p == 0) ? nullptr : (to_out ? tp : perm);
    int* dk = to_out ? sorted_tokens : tk;
    int* dp = to_out ? perm : tp;
    sort_count_kernel<<<grid, blk, 0, st>>>(sk, hist, (int)n, vocab, 4 * p, cpt);
    sort_scan_kernel<<<dim3(1), dim3(1024), 0, st>>>(hist);
    sort_scatter_kernel<<<grid, blk, 0, st>>>(sk, sp, dk, dp, hist, (int)n, vocab, 4 * p, cpt);
  }
  DMI_CHECK_LAUNCH("sort_tokens");
  return DMI_OK;
}

// ---- scatter-add = gradient of mtf.gather (Appendix A.6), deterministic, no atomics -------------------------------
// Positions are visited in token-id order (sorted_tok ascending, perm = source row).  One wave per chunk of 32 sorted
// positions accumulates each run of equal ids in registers.  A run that lies entirely inside its chunk is stored
// directly (single writer).  The chunk's first / last run may continue in the neighbouring chunks (the padding id fills
// hundreds of chunks): those partial sums go to part[chunk][0 / 1][d], and a second kernel lets the wave of the chunk
// where such a run STARTS add the partials of the following chunks in chunk order.  dwte is zeroed here (ids absent
// from the batch have zero gradient).
#define EB_CH 32
__device__ __forceinline__ int eb_clamp(int t, int vocab) { return t < 0 ? 0 : (t >= vocab ? vocab - 1 : t); }
__global__ __launch_bounds__(256) void embed_bwd_wte_sorted_kernel(const int* __restrict__ sorted_tok,
                                                                   const int* __restrict__ perm,
                                                                   const bf16_t* __restrict__ dx,
                                                                   float* __restrict__ dwte, float* __restrict__ part,
                                                                   int64_t n, int d, int vocab) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + wid;
  const int64_t i0 = chunk * EB_CH;
  if (i0 >= n) return;
  const int64_t i1 = (i0 + EB_CH < n) ? i0 + EB_CH : n;
  const int first_tok = sorted_tok[i0];
  const bool head_open = (i0 > 0) && (sorted_tok[i0 - 1] == first_tok);
  for (int c = lane; c < d / 8; c += 64) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int cur = first_tok;
    bool open = head_open;   // the run being accumulated started before this chunk
    bool first = true;       // ... and is the chunk's first run
    for (int64_t i = i0; i < i1; ++i) {
      const int t = sorted_tok[i];
      if (t != cur) {   // run [.., i) ended inside the chunk
        float* dst = open ? part + ((int64_t)chunk * 2 + 0) * d + c * 8 : dwte + (int64_t)eb_clamp(cur, vocab) * d + c * 8;
        *(f32x4*)dst = f32x4{acc[0], acc[1], acc[2], acc[3]};
        *(f32x4*)(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        cur = t;
        open = false;
        first = false;
      }
      float f[8];
      unpack8(*(const u32x4*)(dx + (int64_t)perm[i] * d + c * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
    const bool tail_open = (i1 < n) && (sorted_tok[i1] == cur);
    float* dst;
    if (open || tail_open) dst = part + ((int64_t)chunk * 2 + (first ? 0 : 1)) * d + c * 8;
    else dst = dwte + (int64_t)eb_clamp(cur, vocab) * d + c * 8;
    *(f32x4*)dst = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *(f32x4*)(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
  }
}
// the chunk's LAST run, if it continues into the next chunk and did not start before this chunk, owns the row: add the
// slot-0 partials of the following chunks while they continue the same id (fixed order -> deterministic)
__global__ __launch_bounds__(256) void embed_bwd_wte_combine_kernel(const int* __restrict__ sorted_tok, float* __restrict__ dwte,
                                                                    const float* __restrict__ part, int64_t n, int d, int vocab) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + wid;
  const int64_t i0 = chunk * EB_CH;
  if (i0 >= n) return;
  const int64_t i1 = (i0 + EB_CH < n) ? i0 + EB_CH : n;
  if (i1 >= n) return;                                  // last chunk: nothing continues
  const int tok = sorted_tok[i1 - 1];
  if (sorted_tok[i1] != tok) return;                    // last run is closed at the tail
  const bool single = sorted_tok[i0] == tok;            // the whole chunk is one run
  if (single && i0 > 0 && sorted_tok[i0 - 1] == tok) return;   // a continuation, not a start
  const int slot = single ? 0 : 1;
  const int64_t nchunks = (n + EB_CH - 1) / EB_CH;
  // number of following chunks that continue the run: chunk k continues iff it starts with tok; it is the last one iff
  // the run ends inside it.  64 chunks are classified per ballot (the padding id spans hundreds of chunks; a serial walk
  // with two dependent loads per chunk cost 110 us on the dalle_example batch).
  int64_t cnt = 0;
  for (int64_t base = chunk + 1; base < nchunks; base += 64) {
    const int64_t k = base + lane;
    bool cont = false, last = true;
    if (k < nchunks) {
      const int64_t k0 = k * EB_CH, k1 = (k0 + EB_CH < n) ? k0 + EB_CH : n;
      cont = sorted_tok[k0] == tok;
      last = !(sorted_tok[k1 - 1] == tok && k1 < n && sorted_tok[k1] == tok);
    }
    const unsigned long long stop = __ballot(!cont || last);   // first chunk that is not a full continuation
    if (stop == 0ull) { cnt += 64; continue; }
    const int f = __ffsll((long long)stop) - 1;
    const unsigned long long contb = __ballot(cont);
    cnt += f + (((contb >> f) & 1ull) ? 1 : 0);                   // chunk f still contributes if it starts with tok
    break;
  }
  const float* p0 = part + ((int64_t)chunk * 2 + slot) * d;
  const float* pk = part + ((int64_t)(chunk + 1) * 2) * d;   // slot 0 of the following chunks, stride 2 d
  for (int c = lane; c < d / 4; c += 64) {
    f32x4 acc = *(const f32x4*)(p0 + c * 4);
    int64_t k = 0;
    for (; k + 4 <= cnt; k += 4) {   // four independent loads in flight, summed in chunk order
      const f32x4 v0 = *(const f32x4*)(pk + (k + 0) * 2 * d + c * 4);
      const f32x4 v1 = *(const f32x4*)(pk + (k + 1) * 2 * d + c * 4);
      const f32x4 v2 = *(const f32x4*)(pk + (k + 2) * 2 * d + c * 4);
      const f32x4 v3 = *(const f32x4*)(pk + (k + 3) * 2 * d + c * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = (((acc[j] + v0[j]) + v1[j]) + v2[j]) + v3[j];
    }
    for (; k < cnt; ++k) {
      const f32x4 v = *(const f32x4*)(pk + k * 2 * d + c * 4);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    *(f32x4*)(dwte + (int64_t)eb_clamp(tok, vocab) * d + c * 4) = acc;
  }
}

extern "C" int64_t dmi_embed_bwd_workspace_bytes(int B, int S, int d) {
  return (((int64_t)B * S + EB_CH - 1) / EB_CH) * 2 * d * 4 + 256;
}
extern "C" int dmi_embed_bwd(const int32_t* sorted_tokens, const int32_t* perm, const uint16_t* dx, float* dwte,
                             float* dwpe, int B, int S, int d, int vocab, void* workspace, void* stream) {
  DMI_REQUIRE(sorted_tokens && perm && dx && dwte && dwpe && workspace, "embed_bwd: null pointer");
  DMI_REQUIRE(d % 8 == 0 && B > 0 && S > 0 && vocab > 0, "embed_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  embed_bwd_wpe_kernel<<<dim3((S + 3) / 4), dim3(256), 0, st>>>(dx, dwpe, B, S, d);
  DMI_CHECK_LAUNCH("embed_bwd_wpe");
  DMI_REQUIRE(hipMemsetAsync(dwte, 0, (size_t)vocab * d * 4, st) == hipSuccess, "embed_bwd: memset failed");
  const int64_t n = (int64_t)B * S;
  const int64_t chunks = cdiv64(n, EB_CH);
  embed_bwd_wte_sorted_kernel<<<dim3((unsigned)cdiv64(chunks, 4)), dim3(256), 0, st>>>(sorted_tokens, perm, dx, dwte, (float*)workspace, n, d, vocab);
  DMI_CHECK_LAUNCH("embed_bwd_wte_sorted");
  embed_bwd_wte_combine_kernel<<<dim3((unsigned)cdiv64(chunks, 4)), dim3(256), 0, st>>>(sorted_tokens, dwte, (const float*)workspace, n, d, vocab);
  DMI_CHECK_LAUNCH("embed_bwd_wte_combine");
  return DMI_OK;
}

// =====================================================================================
// K2 LayerNorm (src/dalle_mtf/models.py:373-389, layers.py:30-33): biased variance of the
// centred row, eps inside rsqrt.  One wave per row, row cached in registers (d <= 512*NC).
// =====================================================================================
template <int NC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ g,
                                                     const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wid;
  if (row >= rows) return;
  const int nch = d / 8;
  float v[NC][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      unpack8(*(const u32x4*)(x + row * d + c * 8), v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    }
  }
  const float mu = wave_sum(sum) / (float)d;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[i][j] -= mu;
        sq += v[i][j] * v[i][j];
      }
    }
  }
  const float rs = rsqrtf(wave_sum(sq) / (float)d + eps);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      float fg[8], fb[8], o[8];
      unpack8(*(const u32x4*)(g + c * 8), fg);
      unpack8(*(const u32x4*)(b + c * 8), fb);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = v[i][j] * rs * fg[j] + fb[j];
      *(u32x4*)(y + row * d + c * 8) = pack8(o);
    }
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

extern "C" int dmi_layernorm_fwd(const uint16_t* x, const uint16_t* g, const uint16_t* b, uint16_t* y, float* mean,
                                 float* rstd, int64_t rows, int d, float eps, void* stream) {
  DMI_REQUIRE(x && g && b && y && mean && rstd, "layernorm_fwd: null pointer");
  DMI_REQUIRE(d % 8 == 0 && d <= 4096 && rows > 0, "layernorm_fwd: unsupported d=%d (need d%%8==0, d<=4096)", d);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv64(rows, 4)), blk(256);
  if (d <= 512) ln_fwd_kernel<1><<<grid, blk, 0, st>>>(x, g, b, y, mean, rstd, rows, d, eps);
  else if (d <= 1024) ln_fwd_kernel<2><<<grid, blk, 0, st>>>(x, g, b, y, mean, rstd, rows, d, eps);
  else if (d <= 2048) ln_fwd_kernel<4><<<grid, blk, 0, st>>>(x, g, b, y, mean, rstd, rows, d, eps);
  else ln_fwd_kernel<8><<<grid, blk, 0, st>>>(x, g, b, y, mean, rstd, rows, d, eps);
  DMI_CHECK_LAUNCH("layernorm_fwd");
  return DMI_OK;
}

// ---- generic deterministic partial reduce: out[c] = sum_{p<P} part[p*ncols + c]
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int P, int ncols) {
  __shared__ float sm[4][64];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float acc = 0.f;
  if (c < ncols) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // 4 independent chains keep 4+ loads in flight
    int p = grp;
    for (; p + 12 < P; p += 16) {
      a0 += part[(int64_t)p * ncols + c];
      a1 += part[(int64_t)(p + 4) * ncols + c];
      a2 += part[(int64_t)(p + 8) * ncols + c];
      a3 += part[(int64_t)(p + 12) * ncols + c];
    }
    for (; p < P; p += 4) a0 += part[(int64_t)p * ncols + c];
    acc = (a0 + a1) + (a2 + a3);
  }
  sm[grp][cl] = acc;
  __syncthreads();
  if (grp == 0 && c < ncols) out[c] = (sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl]);
}

#define LN_BWD_RPB 32  // rows per block (8 per wave): 1280 blocks at M = 40960 -> ~20 waves/CU hide the per-row latency chain
extern "C" int64_t dmi_layernorm_bwd_workspace_bytes(int64_t rows, int d) {
  return cdiv64(rows, LN_BWD_RPB) * 2 * (int64_t)d * 4;
}

// dx = rstd * (dy*g - mean_j(dy*g) - xhat * mean_j(dy*g*xhat)) (+ dres);  partial dg/db per block.
template <int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                     const bf16_t* __restrict__ g, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
                                                     bf16_t* __restrict__ dx, float* __restrict__ part,
                                                     int64_t rows, int d) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [4][2][d]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nch = d / 8;
  float ag[NC][8], ab[NC][8], fg[NC][8];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
#pragma unroll
    for (int j = 0; j < 8; ++j) ag[i][j] = ab[i][j] = 0.f;
    if (c < nch) unpack8(*(const u32x4*)(g + c * 8), fg[i]);
  }
  const int64_t r0 = (int64_t)blockIdx.x * LN_BWD_RPB;
  // software pipeline over this wave's rows: the raw loads of row it+1 are issued before the reductions of row it
  u32x4 rdy[NC], rx[NC], rr[NC];
  float mu_n = 0.f, rs_n = 0.f;
  auto fetch = [&](int64_t row) {
    if (row < rows) {
      mu_n = mean[row];
      rs_n = rstd[row];
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
          rdy[i] = *(const u32x4*)(dy + row * d + c * 8);
          rx[i] = *(const u32x4*)(x + row * d + c * 8);
          if (dres) rr[i] = *(const u32x4*)(dres + row * d + c * 8);
        }
      }
    }
  };
  fetch(r0 + wid);
  for (int it = 0; it < LN_BWD_RPB / 4; ++it) {
    const int64_t row = r0 + wid + 4 * it;
    if (row >= rows) break;
    const float mu = mu_n, rs = rs_n;
    float fy[NC][8], xh[NC][8], fr[NC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float fx[8];
        unpack8(rdy[i], fy[i]);
        unpack8(rx[i], fx);
        if (dres) unpack8(rr[i], fr[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[i][j] = (fx[j] - mu) * rs;
          ab[i][j] += fy[i][j];
          ag[i][j] += fy[i][j] * xh[i][j];
          fy[i][j] *= fg[i][j];  // dy*g
          s1 += fy[i][j];
          s2 += fy[i][j] * xh[i][j];
        }
      }
    }
    if (it + 1 < LN_BWD_RPB / 4) fetch(row + 4);
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs * (fy[i][j] - s1 - xh[i][j] * s2);
        if (dres) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += fr[i][j];
        }
        *(u32x4*)(dx + row * d + c * 8) = pack8(o);
      }
    }
  }
  // cross-wave combine, one partial row per block
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sm[(wid * 2 + 0) * d + c * 8 + j] = ag[i][j];
        sm[(wid * 2 + 1) * d + c * 8 + j] = ab[i][j];
      }
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * d; idx += 256) {
    const int which = idx / d, col = idx % d;
    float s = (sm[(0 * 2 + which) * d + col] + sm[(1 * 2 + which) * d + col]) +
              (sm[(2 * 2 + which) * d + col] + sm[(3 * 2 + which) * d + col]);
    part[(int64_t)blockIdx.x * 2 * d + idx] = s;
  }
}

// out layout of the reduce: [dg(d) | db(d)] -> two destination pointers.  16 columns x 16 row-groups per block
// (2d/16 blocks), 2 independent chains per thread.
__global__ __launch_bounds__(256) void ln_bwd_finish_kernel(const float* __restrict__ part, float* __restrict__ dg,
                                                            float* __restrict__ db, int P, int d) {
  __shared__ float sm[16][17];
  const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int64_t nc = 2 * (int64_t)d;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
  if (c < nc) {
    int p = grp;
    for (; p + 112 < P; p += 128) {  // eight independent loads in flight per thread
      const float* q = part + (int64_t)p * nc + c;
      a0 += q[0];
      a1 += q[16 * nc];
      a2 += q[32 * nc];
      a3 += q[48 * nc];
      a4 += q[64 * nc];
      a5 += q[80 * nc];
      a6 += q[96 * nc];
      a7 += q[112 * nc];
    }
    for (; p < P; p += 16) a0 += part[(int64_t)p * nc + c];
  }
  sm[grp][cl] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  __syncthreads();
  if (grp == 0 && c < nc) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += sm[q][cl];
    if (c < d) dg[c] = s;
    else db[c - d] = s;
  }
}

extern "C" int dmi_layernorm_bwd(const uint16_t* dy, const uint16_t* x, const uint16_t* g, const float* mean,
                                 const float* rstd, const uint16_t* dres, uint16_t* dx, float* dg, float* db,
                                 void* workspace, int64_t rows, int d, void* stream) {
  DMI_REQUIRE(dy && x && g && mean && rstd && dx && workspace && ((dg && db) || (!dg && !db)), "layernorm_bwd: null pointer");
  DMI_REQUIRE(d % 8 == 0 && d <= 2048 && rows > 0, "layernorm_bwd: unsupported d=%d (need d%%8==0, d<=2048)", d);
  hipStream_t st = (hipStream_t)stream;
  const int P = (int)cdiv64(rows, LN_BWD_RPB);
  const size_t shm = (size_t)4 * 2 * d * sizeof(float);
  float* part = (float*)workspace;
  dim3 grid(P), blk(256);
  if (d <= 512) ln_bwd_kernel<1><<<grid, blk, shm, st>>>(dy, x, g, mean, rstd, dres, dx, part, rows, d);
  else if (d <= 1024) ln_bwd_kernel<2><<<grid, blk, shm, st>>>(dy, x, g, mean, rstd, dres, dx, part, rows, d);
  else ln_bwd_kernel<4><<<grid, blk, shm, st>>>(dy, x, g, mean, rstd, dres, dx, part, rows, d);
  DMI_CHECK_LAUNCH("layernorm_bwd");
  if (dg == nullptr && db == nullptr) return DMI_OK;   // deferred: the caller reduces the partials with dmi_layernorm_bwd_finish_batch
  ln_bwd_finish_kernel<<<dim3((2 * d + 15) / 16), blk, 0, st>>>(part, dg, db, P, d);
  DMI_CHECK_LAUNCH("layernorm_bwd_finish");
  return DMI_OK;
}

// dg | db = column sums of P partial rows [2 d] (the reduce behind dmi_layernorm_bwd, for partials produced elsewhere: the fused
// LayerNorm backward of dmi_gemm_nt_lnbwd)
extern "C" int dmi_layernorm_bwd_finish_parts(const float* part, int P, float* dg, float* db, int d, void* stream) {
  DMI_REQUIRE(part && dg && db && P > 0 && d % 8 == 0 && d <= 2048, "layernorm_bwd_finish_parts: bad arguments");
  ln_bwd_finish_kernel<<<dim3((2 * d + 15) / 16), dim3(256), 0, (hipStream_t)stream>>>(part, dg, db, P, d);
  DMI_CHECK_LAUNCH("layernorm_bwd_finish_parts");
  return DMI_OK;
}

// The gain / bias gradients of several LayerNorms reduced in ONE launch: blockIdx.y = the LayerNorm.  Each of the 13 reduces of a
// dalle_example step is a 64-block kernel of ~7 us; the results are only needed by the gradient exchange / the optimizer.
#define LN_FINISH_MAX 16
struct LnFinishBatch {
  const float* part[LN_FINISH_MAX];
  float* dg[LN_FINISH_MAX];
  float* db[LN_FINISH_MAX];
  int P[LN_FINISH_MAX];
  int d;
};
__global__ __launch_bounds__(256) void ln_bwd_finish_batch_kernel(LnFinishBatch b) {
  __shared__ float sm[16][17];
  const int it = blockIdx.y;
  const float* __restrict__ part = b.part[it];
  const int P = b.P[it], d = b.d;
  const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int64_t nc = 2 * (int64_t)d;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
  if (c < nc) {   // the same fixed summation order as ln_bwd_finish_kernel: bit-identical results
    int p = grp;
    for (; p + 112 < P; p += 128) {
      const float* q = part + (int64_t)p * nc + c;
      a0 += q[0];
      a1 += q[16 * nc];
      a2 += q[32 * nc];
      a3 += q[48 * nc];
      a4 += q[64 * nc];
      a5 += q[80 * nc];
      a6 += q[96 * nc];
      a7 += q[112 * nc];
    }
    for (; p < P; p += 16) a0 += part[(int64_t)p * nc + c];
  }
  sm[grp][cl] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  __syncthreads();
  if (grp == 0 && c < nc) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += sm[q][cl];
    if (c < d) b.dg[it][c] = s;
    else b.db[it][c - d] = s;
  }
}
extern "C" int dmi_layernorm_bwd_finish_batch(const void* const* workspaces, float* const* dgs, float* const* dbs, const int64_t* rows,
                                              int n, int d, void* stream) {
  DMI_REQUIRE(workspaces && dgs && dbs && rows && n >= 1 && n <= LN_FINISH_MAX, "layernorm_bwd_finish_batch: 1..%d items", LN_FINISH_MAX);
  DMI_REQUIRE(d % 8 == 0 && d <= 2048, "layernorm_bwd_finish_batch: unsupported d=%d", d);
  LnFinishBatch b;
  b.d = d;
  for (int i = 0; i < n; ++i) {
    DMI_REQUIRE(workspaces[i] && dgs[i] && dbs[i] && rows[i] > 0, "layernorm_bwd_finish_batch: null pointer in item %d", i);
    b.part[i] = (const float*)workspaces[i]; b.dg[i] = dgs[i]; b.db[i] = dbs[i];
    b.P[i] = (int)cdiv64(rows[i], LN_BWD_RPB);
  }
  ln_bwd_finish_batch_kernel<<<dim3((2 * d + 15) / 16, n), dim3(256), 0, (hipStream_t)stream>>>(b);
  DMI_CHECK_LAUNCH("layernorm_bwd_finish_batch");
  return DMI_OK;
}

// =====================================================================================
// column sum (bias gradients)
// =====================================================================================
#define COLSUM_RPB 256
extern "C" int64_t dmi_colsum_workspace_bytes(int64_t M, int N) { return cdiv64(M, COLSUM_RPB) * (int64_t)N * 4; }

__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ Y, int ldy, float* __restrict__ part,
                                                     int64_t M, int N) {
  __shared__ float sm[4][64][8];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int cg = blockIdx.x * 64 + cl;  // group of 8 columns
  const int64_t r0 = (int64_t)blockIdx.y * COLSUM_RPB;
  const int64_t r1 = (r0 + COLSUM_RPB < M) ? r0 + COLSUM_RPB : M;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (cg * 8 < N) {
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      float f[8];
      unpack8(*(const u32x4*)(Y + r * ldy + cg * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sm[rl][cl][j] = acc[j];
  __syncthreads();
  if (rl == 0 && cg * 8 < N) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      part[(int64_t)blockIdx.y * N + cg * 8 + j] = (sm[0][cl][j] + sm[1][cl][j]) + (sm[2][cl][j] + sm[3][cl][j]);
  }
}

extern "C" int dmi_colsum(const uint16_t* Y, int ldy, float* out, int64_t M, int N, void* workspace, void* stream) {
  DMI_REQUIRE(Y && out && workspace, "colsum: null pointer");
  DMI_REQUIRE(N % 8 == 0 && ldy % 8 == 0 && M > 0, "colsum: N,ldy must be multiples of 8");
  hipStream_t st = (hipStream_t)stream;
  const int P = (int)cdiv64(M, COLSUM_RPB);
  colsum_kernel<<<dim3((N / 8 + 63) / 64, P), dim3(256), 0, st>>>(Y, ldy, (float*)workspace, M, N);
  DMI_CHECK_LAUNCH("colsum");
  reduce_partials_kernel<<<dim3((N + 63) / 64), dim3(256), 0, st>>>((const float*)workspace, out, P, N);
  DMI_CHECK_LAUNCH("colsum_reduce");
  return DMI_OK;
}

// =====================================================================================
// batched bf16 transpose: element (b,h,r,c) at in + b*sb + h*sh + r*sr + c  ->  out[((b*nh+h)*C + c)*R + r]
// =====================================================================================
// Rv = valid input rows, R = output row pitch (>= Rv; rows in [Rv, R) are written as zeros)
__device__ __forceinline__ void transpose_tile(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int Rv, int R,
                                               int C, int64_t sr, int r0, int c0, unsigned (*t)[33]) {
  const int tid = threadIdx.x;
  {
    const int chunk = tid & 7, rp = tid >> 3;
    const int ra = r0 + 2 * rp, c = c0 + 8 * chunk;
    u32x4 va = {0, 0, 0, 0}, vb = {0, 0, 0, 0};
    if (c < C) {
      if (ra < Rv) va = *(const u32x4*)(src + (int64_t)ra * sr + c);
      if (ra + 1 < Rv) vb = *(const u32x4*)(src + (int64_t)(ra + 1) * sr + c);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned lo = (va[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
      const unsigned hi = (vb[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
      t[8 * chunk + j][rp] = lo | (hi << 16);
    }
  }
  __syncthreads();
  {
    const int orow = tid >> 2, seg = tid & 3;
    const int c = c0 + orow, r = r0 + 16 * seg;
    if (c < C) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int rr = r + 8 * q;
        if (rr < R) {
          u32x4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = t[orow][8 * seg + 4 * q + i];
          *(u32x4*)(dst + (int64_t)c * R + rr) = v;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                        int nh, int Rv, int R, int C, int64_t sb, int64_t sh, int64_t sr) {
  __shared__ unsigned t[64][33];
  const int bh = blockIdx.z, b = bh / nh, h = bh % nh;
  transpose_tile(in + b * sb + h * sh, out + (int64_t)bh * C * R, Rv, R, C, sr, blockIdx.y * 64, blockIdx.x * 64, t);
}

// Many independent [R,C] -> [C,R] transposes inside one pair of buffers in ONE launch (the per-step refresh of the
// forward GEMMs' [out,in] weight copies was 25 launches of ~5 us each).  table[i] = {in_off, out_off, R, C,
// first_tile} in elements / 64x64 tiles, first_tile ascending.
__global__ __launch_bounds__(256) void transpose_batch_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                              const int64_t* __restrict__ table, int n) {
  __shared__ unsigned t[64][33];
  const int64_t tile = blockIdx.x;
  int i = 0;
  while (i + 1 < n && table[(i + 1) * 5 + 4] <= tile) ++i;
  const int64_t in_off = table[i * 5], out_off = table[i * 5 + 1];
  const int R = (int)table[i * 5 + 2], C = (int)table[i * 5 + 3];
  const int local = (int)(tile - table[i * 5 + 4]);
  const int tc = (C + 63) / 64;
  transpose_tile(in + in_off, out + out_off, R, R, C, C, (local / tc) * 64, (local % tc) * 64, t);
}
extern "C" int dmi_transpose_bf16_batch(const uint16_t* in_base, uint16_t* out_base, const int64_t* table, int n,
                                        int64_t total_tiles, void* stream) {
  DMI_REQUIRE(in_base && out_base && table && n > 0 && total_tiles > 0, "transpose_batch: bad args");
  transpose_batch_kernel<<<dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream>>>(in_base, out_base, table, n);
  DMI_CHECK_LAUNCH("transpose_batch");
  return DMI_OK;
}

static int dmi_transpose_padded(const uint16_t* in, uint16_t* out, int nb, int nh, int Rv, int Rp, int C,
                         int64_t in_b_stride, int64_t in_h_stride, int64_t in_r_stride, void* stream) {
  DMI_REQUIRE(in && out, "transpose: null pointer");
  DMI_REQUIRE(Rp % 8 == 0 && Rp >= Rv && C % 8 == 0 && in_r_stride % 8 == 0 && in_h_stride % 8 == 0 && in_b_stride % 8 == 0,
              "transpose: R, C and strides must be multiples of 8 (R=%d C=%d)", Rp, C);
  transpose_kernel<<<dim3((C + 63) / 64, (Rp + 63) / 64, nb * nh), dim3(256), 0, (hipStream_t)stream>>>(
      in, out, nh, Rv, Rp, C, in_b_stride, in_h_stride, in_r_stride);
  DMI_CHECK_LAUNCH("transpose");
  return DMI_OK;
}
extern "C" int dmi_transpose_bf16_padded(const uint16_t* in, uint16_t* out, int R_valid, int R_pitch, int C, void* stream) {
  return dmi_transpose_padded(in, out, 1, 1, R_valid, R_pitch, C, 0, 0, C, stream);
}
extern "C" int dmi_transpose_bf16(const uint16_t* in, uint16_t* out, int batch, int R, int C, void* stream) {
  return dmi_transpose_padded(in, out, batch, 1, R, R, C, (int64_t)R * C, 0, C, stream);
}

// =====================================================================================
// integer paths (bit-exact): label shift (models.py:407-410), token assembly (model_fns.py:76-77,118-119)
// =====================================================================================
__global__ void shift_labels_kernel(const int* __restrict__ tokens, int* __restrict__ labels, int B, int S, int eos) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * S) return;
  const int s = (int)(i % S);
  labels[i] = (s == S - 1) ? eos : tokens[i + 1];
}
extern "C" int dmi_shift_labels(const int32_t* tokens, int32_t* labels, int B, int S, int eos, void* stream) {
  DMI_REQUIRE(tokens && labels && B > 0 && S > 0, "shift_labels: bad args");
  const int64_t n = (int64_t)B * S;
  shift_labels_kernel<<<dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(tokens, labels, B, S, eos);
  DMI_CHECK_LAUNCH("shift_labels");
  return DMI_OK;
}

// one wave per (b, p): argmax over C fp32 logits, first max on ties (tf.math.argmax), + text_vocab
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const int* __restrict__ text,
                                                              const float* __restrict__ logits, int* __restrict__ out,
                                                              int B, int T, int P, int C, int text_vocab) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t idx = (int64_t)blockIdx.x * 4 + wid;
  const int S = T + P;
  if (idx >= (int64_t)B * S) return;
  const int b = (int)(idx / S), s = (int)(idx % S);
  if (s < T) {
    if (lane == 0) out[idx] = text[(int64_t)b * T + s];
    return;
  }
  const float* row = logits + ((int64_t)b * P + (s - T)) * C;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = row[c];
    if (v > best || (v == best && c < bi)) {
      best = v;
      bi = c;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if (lane == 0) out[idx] = (bi == 0x7fffffff ? 0 : bi) + text_vocab;
}
extern "C" int dmi_assemble_tokens(const int32_t* text, const float* vae_logits, int32_t* tokens_out, int B, int T,
                                   int P, int C, int text_vocab, void* stream) {
  DMI_REQUIRE(text && vae_logits && tokens_out && B > 0 && T >= 0 && P > 0 && C > 0, "assemble_tokens: bad args");
  const int64_t n = (int64_t)B * (T + P);
  assemble_tokens_kernel<<<dim3((unsigned)cdiv64(n, 4)), dim3(256), 0, (hipStream_t)stream>>>(text, vae_logits, tokens_out, B, T, P, C, text_vocab);
  DMI_CHECK_LAUNCH("assemble_tokens");
  return DMI_OK;
}

// =====================================================================================
// K7 cross entropy over bf16 logits (models.py:348-359; mtf softmax_cross_entropy_with_logits A.5)
// one 256-thread block per row; pass 1 online max/sum; pass 2 (L2-resident re-read) writes dz in place.
// =====================================================================================
// General path: 256 threads per row, two passes (second pass re-reads the row through the caches).
__global__ __launch_bounds__(256) void cross_entropy_kernel(bf16_t* __restrict__ z, int ldz,
                                                            const int* __restrict__ labels,
                                                            float* __restrict__ loss_rows, float* __restrict__ lse_out,
                                                            int V, float dz_scale) {
  __shared__ float sm_m[4], sm_s[4];
  const int64_t row = blockIdx.x;
  bf16_t* zr = z + row * ldz;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nch = (V + 7) / 8;
  float m = -INFINITY, s = 0.f;
  for (int c = tid; c < nch; c += 256) {
    float f[8];
    unpack8(*(const u32x4*)(zr + c * 8), f);
    float cm = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (c * 8 + j >= V) f[j] = -INFINITY;
      cm = fmaxf(cm, f[j]);
    }
    if (cm > m) {
      s *= __expf(m - cm);
      m = cm;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += __expf(f[j] - m);
  }
  float wm = wave_max(m);
  s *= (m == -INFINITY) ? 0.f : __expf(m - wm);
  s = wave_sum(s);
  if (lane == 0) {
    sm_m[wid] = wm;
    sm_s[wid] = s;
  }
  __syncthreads();
  const float bm = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
  float bs = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) bs += sm_s[w] * __expf(sm_m[w] - bm);
  const float lse = bm + __logf(bs);
  const int label = labels[row];
  if (tid == 0) {
    const float zl = (label >= 0 && label < V) ? bf2f(zr[label]) : 0.f;
    loss_rows[row] = lse - zl;
    if (lse_out) lse_out[row] = lse;
  }
  if (dz_scale == 0.f) return;
  __syncthreads();  // tid 0 read z[label] before anyone overwrites it
  for (int c = tid; c < ldz / 8; c += 256) {  // every column of the row incl. the pad: dz[pad] = 0
    float f[8];
    unpack8(*(const u32x4*)(zr + c * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = c * 8 + j;
      float p = (col < V) ? __expf(f[j] - lse) : 0.f;
      if (col == label) p -= 1.f;
      f[j] = p * dz_scale;
    }
    *(u32x4*)(zr + c * 8) = pack8(f);
  }
}

// Large-vocabulary path: 1024 threads per row, the row (<= NC*1024 chunks of 16 B) stays in registers between the
// statistics pass and the dlogits pass: one HBM read + one HBM write per element.
template <int NC>
__global__ __launch_bounds__(1024) void cross_entropy_reg_kernel(bf16_t* __restrict__ z, int ldz,
                                                                 const int* __restrict__ labels,
                                                                 float* __restrict__ loss_rows, float* __restrict__ lse_out,
                                                                 int V, float dz_scale) {
  __shared__ float sm_m[16], sm_s[16];
  const int64_t row = blockIdx.x;
  bf16_t* zr = z + row * ldz;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nch = ldz / 8;
  // the row lives in registers as fp32 (NC x 8 per thread); each element costs one unpack, one max, one fma + v_exp
  // (kept: the probabilities are exp(x - max) * (scale / sum), no second exp), one mul and half a pack.  Columns >= V
  // are set to -inf once at load time so no later pass carries a mask.
  float f[NC][8];
  float m = -INFINITY;
  {
    u32x4 raw[NC];  // all loads in flight before the first use
    const u32x4 ninf = {0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u};
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = tid + 1024 * i;
      raw[i] = (c < nch) ? *(const u32x4*)(zr + c * 8) : ninf;
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = tid + 1024 * i;
      unpack8(raw[i], f[i]);
      if (c * 8 + 8 > V) {  // only the chunk straddling V (and pad chunks); the empty asm keeps this a real branch
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) f[i][j] = (c * 8 + j < V) ? f[i][j] : -INFINITY;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, f[i][j]);
    }
  }
  m = wave_max(m);
  if (lane == 0) sm_m[wid] = m;
  __syncthreads();
  float bm = sm_m[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) bm = fmaxf(bm, sm_m[w]);
  constexpr float LOG2E = 1.4426950408889634f;
  const float bm2 = bm * LOG2E;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[i][j] = __builtin_amdgcn_exp2f(__builtin_fmaf(f[i][j], LOG2E, -bm2));
      s += f[i][j];
    }
  s = wave_sum(s);
  if (lane == 0) sm_s[wid] = s;
  __syncthreads();
  float bs = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) bs += sm_s[w];
  const float lse = bm + __logf(bs);
  const int label = labels[row];
  if (tid == 0) {
    const float zl = (label >= 0 && label < V) ? bf2f(zr[label]) : 0.f;  // row not overwritten yet (barrier below)
    loss_rows[row] = lse - zl;
    if (lse_out) lse_out[row] = lse;
  }
  if (dz_scale == 0.f) return;
  __syncthreads();
  const float k = dz_scale / bs;
  const int lch = (label >= 0 && label < V) ? (label >> 3) : -1, lj = label & 7;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = tid + 1024 * i;
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[i][j] *= k;
      if (c == lch) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) f[i][j] -= (j == lj) ? dz_scale : 0.f;
      }
      *(u32x4*)(zr + c * 8) = pack8(f[i]);
    }
  }
}

extern "C" int dmi_cross_entropy(uint16_t* z, int ldz, const int32_t* labels, float* loss_rows, float* lse,
                                 int64_t M, int V, float dz_scale, void* stream) {
  DMI_REQUIRE(z && labels && loss_rows, "cross_entropy: null pointer");
  DMI_REQUIRE(ldz % 8 == 0 && ldz >= ((V + 7) / 8) * 8 && M > 0 && V > 0, "cross_entropy: ldz must be a multiple of 8 and >= round_up(V,8)");
  hipStream_t st = (hipStream_t)stream;
  const int nch = ldz / 8;
  if (nch > 2048 && nch <= 8 * 1024) {
    if (nch <= 4 * 1024) cross_entropy_reg_kernel<4><<<dim3((unsigned)M), dim3(1024), 0, st>>>(z, ldz, labels, loss_rows, lse, V, dz_scale);
    else if (nch <= 7 * 1024) cross_entropy_reg_kernel<7><<<dim3((unsigned)M), dim3(1024), 0, st>>>(z, ldz, labels, loss_rows, lse, V, dz_scale);
    else cross_entropy_reg_kernel<8><<<dim3((unsigned)M), dim3(1024), 0, st>>>(z, ldz, labels, loss_rows, lse, V, dz_scale);
  } else {
    cross_entropy_kernel<<<dim3((unsigned)M), dim3(256), 0, st>>>(z, ldz, labels, loss_rows, lse, V, dz_scale);
  }
  DMI_CHECK_LAUNCH("cross_entropy");
  return DMI_OK;
}

// =====================================================================================
// Fused softmax head (training path): the cross entropy of models.py:348-359 without a logits round trip.
//   dmi_label_logit      zl[m] = x[m,:] . Wt[label[m],:] + bias[label[m]]        (fp32; also the per-row shift of the exponent)
//   dmi_gemm_nt_softmax  E[m,v] = bf16(exp(logit[m,v] - zl[m])) + partial row sums  (gemm.hip)
//   dmi_softmax_finish   S[m] = sum_v exp(.) ;  loss_row[m] = log S[m] + shift[m] - zl[m]  (= logsumexp - label logit, exactly the
//                        reference's loss_batch);  rowscale[m] = dz_scale / S[m];  E[m,label] -= S[m]  (so that dlogits = rowscale * E);
//                        Xs[m,:] = bf16(rowscale[m] * x[m,:])  (the weight-gradient GEMM's left operand: dW = Xs^T E)
// No shift (the engine's choice): exp(logit) is exact to rounding for logits within +-87 -- floating point keeps its relative
// precision, the row maximum is not needed.  A row whose sum leaves [1e-18, 1e18] (some logit > 41, or all < -41 - ln V) is flagged and
// redone exactly by softmax_fixup_kernel with the row maximum as the shift.  With the label logit as shift (rowshift = zl) the
// label entry is 1 and only a logit exceeding the label's by > 88 (a row loss > 88 nats) takes the fix-up path.
// =====================================================================================
__global__ __launch_bounds__(256) void label_logit_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ Wt,
                                                          int ldw, const bf16_t* __restrict__ bias, const int* __restrict__ labels,
                                                          float* __restrict__ zl, int* __restrict__ flag, int64_t M, int K, int V) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) flag[0] = 0;
  const int64_t row = (int64_t)blockIdx.x * 4 + wid;
  if (row >= M) return;
  int lab = labels[row];
  lab = lab < 0 ? 0 : (lab >= V ? V - 1 : lab);
  const bf16_t* xr = X + row * ldx;
  const bf16_t* wr = Wt + (int64_t)lab * ldw;
  float acc = 0.f;
  for (int c = lane; c < K / 8; c += 64) {
    float a[8], b[8];
    unpack8(*(const u32x4*)(xr + c * 8), a);
    unpack8(*(const u32x4*)(wr + c * 8), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_fmaf(a[j], b[j], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) zl[row] = acc + bf2f(bias[lab]);
}
extern "C" int dmi_label_logit(const uint16_t* X, int ldx, const uint16_t* Wt, int ldw, const uint16_t* bias,
                               const int32_t* labels, float* zl, int32_t* flag, int64_t M, int K, int V, void* stream) {
  DMI_REQUIRE(X && Wt && bias && labels && zl && flag, "label_logit: null pointer");
  DMI_REQUIRE(M > 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && V > 0, "label_logit: bad sizes");
  label_logit_kernel<<<dim3((unsigned)cdiv64(M, 4)), dim3(256), 0, (hipStream_t)stream>>>(X, ldx, Wt, ldw, bias, labels, zl, flag, M, K, V);
  DMI_CHECK_LAUNCH("label_logit");
  return DMI_OK;
}

struct SoftmaxFinishArgs {
  const float* part; int nparts;
  const float* zl;        // label logits (dmi_label_logit)
  const float* rowshift;  // nullable: the shift the GEMM epilogue subtracted in the exponent
  const int* labels;
  const bf16_t* X; int ldx;
  bf16_t* E; int lde;
  float* loss_rows; float* rowscale; bf16_t* rowscale_bf16; bf16_t* Xs;
  int* flag;
  int64_t M; int K, V;
  float dz_scale;
};
#define SF_ROWS 64
#define SF_PARTS 16
__global__ __launch_bounds__(256) void softmax_finish_kernel(SoftmaxFinishArgs a) {
  // 64 rows per block, the partials of a row summed by 16 threads (p = q mod 16) with 16-byte loads of 4 adjacent rows: 640
  // blocks x 4 waves for the dalle_example batch (the 160-block version left 2.5 waves per CU and ran latency-bound)
  __shared__ float sm[SF_PARTS][SF_ROWS];
  __shared__ float sc[SF_ROWS];
  const int tid = threadIdx.x, r4 = tid & 15, q = tid >> 4;   // thread: rows 4 r4 .. 4 r4 + 3, partials p = q (mod 16)
  const int64_t m0 = (int64_t)blockIdx.x * SF_ROWS, m = m0 + 4 * r4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (m + 3 < a.M && (a.M & 3) == 0) {
    for (int p = q; p < a.nparts; p += SF_PARTS) {   // fixed order: deterministic
      const f32x4 v = *(const f32x4*)(a.part + (int64_t)p * a.M + m);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
  } else {
    for (int j = 0; j < 4; ++j)
      if (m + j < a.M)
        for (int p = q; p < a.nparts; p += SF_PARTS) s[j] += a.part[(int64_t)p * a.M + m + j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) sm[q][4 * r4 + j] = s[j];
  __syncthreads();
  if (tid < SF_ROWS) {
    const int r = tid;
    const int64_t mr = m0 + r;
    float scale = 0.f;
    if (mr < a.M) {
      float S = 0.f;
#pragma unroll
      for (int k = 0; k < SF_PARTS; ++k) S += sm[k][r];
      // accept window [1e-18, 1e18] (|logsumexp| <= 41): inf / nan (an exponent overflowed) and vanished rows fall outside, and
      // so do rows that are merely LARGE -- for S ~ 1e33..3e38 rowscale = dz_scale / S would be an fp32/bf16 denormal (Xs and
      // rowscale_bf16 flush to 0: a silently dropped row) and the E.W^T accumulators of the input gradient could overflow.
      // Inside the window rowscale >= 1e-18 * dz_scale and E <= 1e18 keep > 60 binades of headroom; outside, the exact fix-up.
      const bool bad = !(S > 1.0e-18f && S < 1.0e18f);
      if (bad) a.flag[0] = 1;
      // loss = logsumexp - label logit = log S + shift - zl   (models.py:348-359: the reference's loss_batch)
      a.loss_rows[mr] = bad ? INFINITY : __logf(S) + (a.rowshift ? a.rowshift[mr] : 0.f) - a.zl[mr];
      scale = bad ? 0.f : a.dz_scale / S;
      if (a.dz_scale != 0.f) {
        a.rowscale[mr] = scale;
        a.rowscale_bf16[mr] = f2bf(scale);
        int lab = a.labels[mr];
        lab = lab < 0 ? 0 : (lab >= a.V ? a.V - 1 : lab);
        bf16_t* el = a.E + mr * a.lde + lab;
        *el = f2bf(bf2f(*el) - S);   // dlogits[m, label] = rowscale * (e_label - S) = dz_scale * (p_label - 1)
      }
    }
    sc[r] = scale;
  }
  __syncthreads();
  if (a.dz_scale == 0.f) return;
  const int cpr = a.K / 8;
  for (int idx = tid; idx < SF_ROWS * cpr; idx += 256) {
    const int rr = idx / cpr, ch = idx - rr * cpr;
    const int64_t mm = m0 + rr;
    if (mm >= a.M) break;
    float f[8];
    unpack8(*(const u32x4*)(a.X + mm * a.ldx + ch * 8), f);
    const float k = sc[rr];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] *= k;
    *(u32x4*)(a.Xs + mm * a.K + ch * 8) = pack8(f);
  }
}
// Exact redo of the flagged rows (loss_rows[m] == inf): logits recomputed from x and Wt with the row maximum as the shift.
#define FIX_MAXK 8192
__global__ __launch_bounds__(256) void softmax_fixup_kernel(SoftmaxFinishArgs a, const bf16_t* __restrict__ Wt, int ldw,
                                                            const bf16_t* __restrict__ bias, int N) {
  if (a.flag[0] == 0) return;
  __shared__ float xs[FIX_MAXK];
  __shared__ float red[4];
  __shared__ float zlab;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int64_t m = blockIdx.x; m < a.M; m += gridDim.x) {
    if (a.loss_rows[m] < 3.0e38f) continue;   // block-uniform
    __syncthreads();
    for (int k = tid; k < a.K; k += 256) xs[k] = bf2f(a.X[m * a.ldx + k]);
    int lab = a.labels[m];
    lab = lab < 0 ? 0 : (lab >= a.V ? a.V - 1 : lab);
    __syncthreads();
    auto logit = [&](int v) {
      const bf16_t* wr = Wt + (int64_t)v * ldw;
      float acc = 0.f;
      for (int c = lane; c < a.K / 8; c += 64) {
        float b[8];
        unpack8(*(const u32x4*)(wr + c * 8), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_fmaf(xs[c * 8 + j], b[j], acc);
      }
      return wave_sum(acc) + bf2f(bias[v]);
    };
    float mx = -INFINITY;
    for (int v = wid; v < a.V; v += 4) {
      const float z = logit(v);
      mx = fmaxf(mx, z);
      if (v == lab && lane == 0) zlab = z;
    }
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float zl = zlab;
    __syncthreads();
    float sum = 0.f;
    for (int v = wid; v < N; v += 4) {
      const float e = (v < a.V) ? __expf(logit(v) - mx) : 0.f;
      sum += e;
      if (lane == 0) a.E[m * a.lde + v] = f2bf(e);
    }
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    const float S = ((red[0] + red[1]) + red[2]) + red[3];
    const float scale = a.dz_scale / S;
    __syncthreads();   // E row complete (block scope) before the label entry is patched
    if (tid == 0) {
      a.loss_rows[m] = mx + __logf(S) - zl;
      if (a.dz_scale != 0.f) {
        a.rowscale[m] = scale;
        a.rowscale_bf16[m] = f2bf(scale);
        a.E[m * a.lde + lab] = f2bf(__expf(zl - mx) - S);
      }
    }
    if (a.dz_scale != 0.f)
      for (int k = tid * 8; k < a.K; k += 256 * 8) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = xs[k + j] * scale;
        *(u32x4*)(a.Xs + m * a.K + k) = pack8(f);
      }
  }
}
extern "C" int dmi_softmax_finish(const float* rowsum_part, int nparts, const float* label_logit, const float* rowshift,
                                  const int32_t* labels, const uint16_t* X, int ldx,
                                  const uint16_t* Wt, int ldw, const uint16_t* bias, uint16_t* E, int lde, int N,
                                  float* loss_rows, float* rowscale, uint16_t* rowscale_bf16, uint16_t* Xs, int32_t* flag,
                                  int64_t M, int K, int V, float dz_scale, void* stream) {
  DMI_REQUIRE(rowsum_part && label_logit && labels && X && Wt && bias && E && loss_rows && flag, "softmax_finish: null pointer");
  DMI_REQUIRE(dz_scale == 0.f || (rowscale && rowscale_bf16 && Xs), "softmax_finish: gradient outputs missing");
  DMI_REQUIRE(M > 0 && K % 8 == 0 && K <= FIX_MAXK && ldx % 8 == 0 && ldw % 8 == 0 && nparts > 0 && V > 0 && V <= N && N <= lde,
              "softmax_finish: bad sizes (K <= %d)", FIX_MAXK);
  SoftmaxFinishArgs a;
  a.part = rowsum_part; a.nparts = nparts; a.zl = label_logit; a.rowshift = rowshift; a.labels = labels; a.X = X; a.ldx = ldx; a.E = E; a.lde = lde;
  a.loss_rows = loss_rows; a.rowscale = rowscale; a.rowscale_bf16 = rowscale_bf16; a.Xs = Xs; a.flag = flag;
  a.M = M; a.K = K; a.V = V; a.dz_scale = dz_scale;
  hipStream_t st = (hipStream_t)stream;
  softmax_finish_kernel<<<dim3((unsigned)cdiv64(M, SF_ROWS)), dim3(256), 0, st>>>(a);
  DMI_CHECK_LAUNCH("softmax_finish");
  softmax_fixup_kernel<<<dim3(256), dim3(256), 0, st>>>(a, Wt, ldw, bias, N);   // returns at once unless a row was flagged
  DMI_CHECK_LAUNCH("softmax_fixup");
  return DMI_OK;
}

// =====================================================================================
// reductions + optimizer (src/optimizers.py:11-16, 82-89, 154-177)
// =====================================================================================
__device__ __forceinline__ float block_sum_256(float x, float* sm) {
  x = wave_sum(x);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = x;
  __syncthreads();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// one block, 1024 threads, four independent float4 chains per thread (fixed order -> deterministic)
__global__ __launch_bounds__(1024) void sum_f32_kernel(const float* __restrict__ x, int64_t n, float scale,
                                                       float* __restrict__ out) {
  __shared__ float sm[16];
  const int tid = threadIdx.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if ((((uintptr_t)x) & 15) == 0) {
    const int64_t n4 = n / 4;
    const f32x4* x4 = (const f32x4*)x;
    int64_t i = tid;
    for (; i + 3072 < n4; i += 4096) {
      const f32x4 v0 = x4[i], v1 = x4[i + 1024], v2 = x4[i + 2048], v3 = x4[i + 3072];
      a0 += (v0[0] + v0[1]) + (v0[2] + v0[3]);
      a1 += (v1[0] + v1[1]) + (v1[2] + v1[3]);
      a2 += (v2[0] + v2[1]) + (v2[2] + v2[3]);
      a3 += (v3[0] + v3[1]) + (v3[2] + v3[3]);
    }
    for (; i < n4; i += 1024) {
      const f32x4 v0 = x4[i];
      a0 += (v0[0] + v0[1]) + (v0[2] + v0[3]);
    }
    if (tid < (n & 3)) a1 += x[n4 * 4 + tid];
  } else {
    for (int64_t i = tid; i < n; i += 1024) a0 += x[i];
  }
  float t = wave_sum((a0 + a1) + (a2 + a3));
  if ((tid & 63) == 0) sm[tid >> 6] = t;
  __syncthreads();
  if (tid == 0) {
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) r += sm[w];
    out[0] = r * scale;
  }
}
extern "C" int dmi_sum_f32(const float* x, int64_t n, float scale, float* out, void* stream) {
  DMI_REQUIRE(x && out && n > 0, "sum_f32: bad args");
  sum_f32_kernel<<<dim3(1), dim3(1024), 0, (hipStream_t)stream>>>(x, n, scale, out);
  DMI_CHECK_LAUNCH("sum_f32");
  return DMI_OK;
}

#define SUMSQ_BLOCKS 2048
extern "C" int64_t dmi_sumsq_workspace_bytes(int64_t n) { (void)n; return SUMSQ_BLOCKS * 4; }
// (A sweep from the END of the buffer -- the flat gradient buffer is laid out in backward's completion order, so its tail is what the 256-MB
// Infinity Cache should still hold at clip time -- measured -0.03 ms per step on one box over three alternations, 0.00 on another over four,
// and slightly slower with Adam behind it in isolation: neutral, not kept.  profiles/r06_ab_sumsq_rev.log)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float sm[4];
  float acc = 0.f;
  const int64_t n4 = n / 4;
  const f32x4* g4 = (const f32x4*)g;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = g4[i];
    acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[n4 * 4 + threadIdx.x];
    acc += v * v;
  }
  const float t = block_sum_256(acc, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}
extern "C" int dmi_sumsq(const float* g, int64_t n, float* out, void* workspace, void* stream) {
  DMI_REQUIRE(g && out && workspace && n > 0, "sumsq: bad args");
  DMI_REQUIRE(((uintptr_t)g & 15) == 0, "sumsq: g must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  sumsq_kernel<<<dim3(SUMSQ_BLOCKS), dim3(256), 0, st>>>(g, n, (float*)workspace);
  DMI_CHECK_LAUNCH("sumsq");
  sum_f32_kernel<<<dim3(1), dim3(1024), 0, st>>>((const float*)workspace, SUMSQ_BLOCKS, 1.0f, out);
  DMI_CHECK_LAUNCH("sumsq_finish");
  return DMI_OK;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16_t* __restrict__ pb, int64_t n,
                                                   const float* __restrict__ gnorm_sq, float clip, float lr, float b1,
                                                   float b2, float eps, float wd, float gscale,
                                                   const float* __restrict__ lr_dev) {
  if (lr_dev) lr = lr_dev[0];   // graph replay: the scheduled learning rate lives in device memory
  float mult = gscale;
  if (gnorm_sq != nullptr && clip > 0.f) {
    const float gn = sqrtf(gnorm_sq[0]) * gscale;  // norm of the scaled gradient
    mult = gscale * (clip / fmaxf(gn, clip));
  }
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 pv = ((f32x4*)p)[i], gv = ((const f32x4*)g)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = gv[j] * mult;
      mv[j] = b1 * mv[j] + (1.f - b1) * gg;
      vv[j] = b2 * vv[j] + (1.f - b2) * gg * gg;
      float upd = mv[j] / (sqrtf(vv[j]) + eps);
      upd += wd * pv[j];
      pv[j] -= lr * upd;
    }
    ((f32x4*)p)[i] = pv;
    ((f32x4*)m)[i] = mv;
    ((f32x4*)v)[i] = vv;
    if (pb) *(u32x2*)(pb + i * 4) = u32x2{pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3])};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    const float gg = g[i] * mult;
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    float pp = p[i];
    pp -= lr * (mm / (sqrtf(vv) + eps) + wd * pp);
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
    if (pb) pb[i] = f2bf(pp);
  }
}
extern "C" int dmi_adam_step(float* p, const float* g, float* m, float* v, uint16_t* p_bf16, int64_t n,
                             const float* gnorm_sq, float clip, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float grad_scale, const float* lr_dev, void* stream) {
  DMI_REQUIRE(p && g && m && v && n > 0, "adam_step: bad args");
  DMI_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && (((uintptr_t)p_bf16) & 7) == 0,
              "adam_step: buffers must be 16-byte aligned");
  int64_t blocks = cdiv64(n / 4 + 1, 256);
  if (blocks > 4096) blocks = 4096;
  adam_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(p, g, m, v, p_bf16, n, gnorm_sq, clip, lr, beta1, beta2, eps, weight_decay, grad_scale, lr_dev);
  DMI_CHECK_LAUNCH("adam_step");
  return DMI_OK;
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = f2bf(in[i]);
}
extern "C" int dmi_cast_f32_bf16(const float* in, uint16_t* out, int64_t n, void* stream) {
  DMI_REQUIRE(in && out && n > 0, "cast: bad args");
  int64_t blocks = cdiv64(n, 256);
  if (blocks > 8192) blocks = 8192;
  cast_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(in, out, n);
  DMI_CHECK_LAUNCH("cast");
  return DMI_OK;
}

// =====================================================================================
// sampling of the next image token from one row of head logits
// =====================================================================================
// The reference stops at the logits (predict raises NotImplementedError, src/model_fns.py:135-136); the sampler around the
// incremental-inference hooks (src/dalle_mtf/models.py:246-254,281-285) is this repo's.  One block per sequence:
//   v[i] = (z[b, i] + bias[i]) / temperature;  top-k filter: keep v[i] >= (k-th largest v)  (ties kept, as a masked_fill
//   of v < kth would);  choice ~ softmax(v) over the kept entries, drawn as argmax_i (v[i] + Gumbel noise) -- the Gumbel-max
//   form of the same categorical draw -- with counter-based noise hash(seed, position, b, i): reproducible, no RNG state.
//   temperature <= 0: greedy (first maximum).
// The k-th largest value is found without a sort: 32 steps of a bitwise binary search on order-preserving integer keys
// (count(key >= candidate) >= k), a block-wide count per step.
// Everything that changes between calls may come from device memory (params_dev: {1/temperature or 0, top_k, seed lo, seed hi};
// pos_dev), and the choice is written where the next decode step reads its input token, so that decode + sampling replay as one
// HIP graph with no host round trip per position.
__device__ __forceinline__ unsigned order_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
#define SAMPLE_MAX_VOCAB 8192
__global__ __launch_bounds__(256) void sample_tokens_kernel(const bf16_t* __restrict__ z, int ldz, const bf16_t* __restrict__ bias, int nv,
                                                            float inv_temp, int top_k, uint64_t seed, const unsigned* __restrict__ params_dev,
                                                            int pos_arg, int* pos_dev, int advance, int token_offset,
                                                            int* __restrict__ next_tok, int* __restrict__ out, int out_ld, int out_col0) {
  __shared__ unsigned keys[SAMPLE_MAX_VOCAB];
  __shared__ int cnt[4];
  __shared__ float bval[4];
  __shared__ int bidx[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (params_dev) {
    inv_temp = __uint_as_float(params_dev[0]);
    top_k = (int)params_dev[1];
    seed = (uint64_t)params_dev[2] | ((uint64_t)params_dev[3] << 32);
  }
  const int counter = pos_dev ? *pos_dev : pos_arg;
  const bool greedy = !(inv_temp > 0.f);
  for (int i = tid; i < nv; i += 256) {
    float v = bf2f(z[(int64_t)b * ldz + i]);
    if (bias) v += bf2f(bias[i]);
    if (!greedy) v *= inv_temp;
    keys[i] = order_key(v);
  }
  __syncthreads();
  unsigned thr = 0u;
  if (!greedy && top_k > 0 && top_k < nv) {
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned cand = thr | (1u << bit);
      int c = 0;
      for (int i = tid; i < nv; i += 256) c += keys[i] >= cand;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
      if (lane == 0) cnt[wid] = c;
      __syncthreads();
      const int total = cnt[0] + cnt[1] + cnt[2] + cnt[3];
      __syncthreads();
      if (total >= top_k) thr = cand;   // block-uniform
    }
  }
  const uint64_t stream_key = splitmix64(seed ^ ((uint64_t)(unsigned)counter * 0xD2B74407B1CE6E93ull));
  float best = -INFINITY;
  int besti = nv;
  for (int i = tid; i < nv; i += 256) {   // ascending i per thread: strict > keeps the first maximum
    const unsigned k = keys[i];
    if (k < thr) continue;
    float s = key_value(k);
    if (!greedy) {
      const uint64_t h = splitmix64(stream_key + (((uint64_t)(unsigned)b << 32) | (unsigned)i));
      // 23 random bits so that the + 0.5 is exact in fp32: u in [2^-24, 1 - 2^-24], strictly inside (0, 1) -- with 24 bits
      // 16777215.5 rounds to 2^24, u == 1 and the Gumbel noise -log(-log u) is +inf (the entry then wins regardless of its logit)
      const float u = ((float)(unsigned)(h >> 41) + 0.5f) * (1.0f / 8388608.0f);
      s -= __logf(-__logf(u));
    }
    if (s > best) { best = s; besti = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (lane == 0) { bval[wid] = best; bidx[wid] = besti; }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (bval[w] > best || (bval[w] == best && bidx[w] < besti)) { best = bval[w]; besti = bidx[w]; }
    if (besti >= nv) besti = 0;          // all-NaN row: defined output
    if (next_tok) next_tok[b] = token_offset + besti;
    const int col = counter - out_col0;
    if (out && col >= 0 && col < out_ld) out[(int64_t)b * out_ld + col] = besti;
    if (advance && pos_dev) {            // the last block to finish moves the position on (every block has read it by then)
      __threadfence();
      if (atomicAdd(pos_dev + 1, 1) == (int)gridDim.x - 1) {
        pos_dev[1] = 0;
        pos_dev[0] = counter + 1;
      }
    }
  }
}
extern "C" int dmi_sample_tokens(const uint16_t* z, int ldz, const uint16_t* bias, int B, int nv, float temperature, int top_k,
                                 uint64_t seed, const uint32_t* params_dev, int pos, int32_t* pos_dev, int advance, int token_offset,
                                 int32_t* next_tok, int32_t* out, int out_ld, int out_col0, void* stream) {
  DMI_REQUIRE(z && (next_tok || out), "sample_tokens: null pointer");
  DMI_REQUIRE(B > 0 && nv > 0 && nv <= SAMPLE_MAX_VOCAB && ldz >= nv, "sample_tokens: need 0 < nv <= %d (nv=%d)", SAMPLE_MAX_VOCAB, nv);
  DMI_REQUIRE(!out || out_ld > 0, "sample_tokens: out_ld");
  DMI_REQUIRE(!advance || pos_dev, "sample_tokens: advance needs pos_dev");
  const float inv_temp = temperature > 0.f ? 1.f / temperature : 0.f;
  sample_tokens_kernel<<<dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream>>>(z, ldz, bias, nv, inv_temp, top_k, seed, params_dev, pos, pos_dev,
                                                                                advance, token_offset, next_tok, out, out_ld, out_col0);
  DMI_CHECK_LAUNCH("sample_tokens");
  return DMI_OK;
}

// fp32 logits of a head-output slice: out[b, i] = float(z[b, i]) + float(bias[i])  (src/dalle_mtf/models.py:394-395)
__global__ void logits_f32_kernel(const bf16_t* __restrict__ z, int ldz, const bf16_t* __restrict__ bias, float* __restrict__ out, int nv) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nv) return;
  float v = bf2f(z[(int64_t)b * ldz + i]);
  if (bias) v += bf2f(bias[i]);
  out[(int64_t)b * nv + i] = v;
}
extern "C" int dmi_logits_f32(const uint16_t* z, int ldz, const uint16_t* bias, float* out, int B, int nv, void* stream) {
  DMI_REQUIRE(z && out, "logits_f32: null pointer");
  DMI_REQUIRE(B > 0 && nv > 0 && ldz >= nv, "logits_f32: need B > 0, 0 < nv <= ldz (B=%d nv=%d ldz=%d)", B, nv, ldz);
  logits_f32_kernel<<<dim3((unsigned)((nv + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream>>>(z, ldz, bias, out, nv);
  DMI_CHECK_LAUNCH("logits_f32");
  return DMI_OK;
}
