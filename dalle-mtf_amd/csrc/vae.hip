// vae.hip -- data-movement and pointwise kernels of the discrete-VAE path (reference
// src/vae_tf/models.py:81-163, src/vae_tf/layers.py:4-25).
//
// Convolution lowering: every flavour the reference uses (tf.layers.conv2d 4x4 s2 / 3x3 s1 / 1x1 SAME,
// conv2d_transpose 4x4 s2 SAME; forward, input gradient, weight gradient) is a tap list feeding the MFMA GEMMs of
// gemm.hip (NT for fwd / dgrad, TN for wgrad).  Layers with C_in % 64 == 0 gather implicitly inside the GEMM
// (dmi_conv_gemm_nt / dmi_conv_wgrad_tn); the materialised im2col below serves the 3-channel input layer and is the
// bit-exact reference of the implicit kernels in the tests:
//   conv s1/s2 fwd      : taps (ky-pad, kx-pad), stride s           -> Y = col(X) . W^T
//   conv s1 dgrad       : taps (pad-ky, pad-kx) on dY               -> dX = col(dY) . Wd^T ,  Wd[ci][(k,co)]
//   conv s2 dgrad / conv-transpose fwd : 4 output-parity classes, 2x2 taps each, then a pixel interleave
//   wgrad               : dW[(k,ci)][co] = col(X)^T . dY   (TN GEMM; lands in the TF kernel layout)
// Activations are NHWC bf16 matrices [B*H*W, C] with C % 8 == 0 (the 3-channel image / reconstruction are padded
// to 8 channels with zeros), so every access is a 16-byte vector.
#include "common.h"

#define MAX_TAPS 16
struct TapList {
  int n;
  int dy[MAX_TAPS];
  int dx[MAX_TAPS];
};

// out[(b,oy,ox)][t*C + c] = x[b, oy*s + dy[t], ox*s + dx[t], c]   (0 outside the image); row pitch ldo >= n*C,
// columns [n*C, ldo) are zero-filled (K padding for the GEMM).
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, TapList taps,
                                                     int B, int H, int W, int C, int Ho, int Wo, int stride, int ldo) {
  const int cpr = ldo / 8;  // chunks per output row
  const int64_t total = (int64_t)B * Ho * Wo * cpr;
  const int cc = C / 8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int ch = (int)(idx % cpr);
    const int64_t m = idx / cpr;
    const int t = ch / cc, c8 = ch % cc;
    u32x4 v = {0, 0, 0, 0};
    if (t < taps.n) {
      const int ox = (int)(m % Wo);
      const int oy = (int)((m / Wo) % Ho);
      const int b = (int)(m / ((int64_t)Wo * Ho));
      const int iy = oy * stride + taps.dy[t], ix = ox * stride + taps.dx[t];
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const u32x4*)(x + (((int64_t)b * H + iy) * W + ix) * C + c8 * 8);
    }
    *(u32x4*)(out + m * ldo + ch * 8) = v;
  }
}

extern "C" int dmi_im2col(const uint16_t* x, uint16_t* out, int B, int H, int W, int C, int Ho, int Wo, int stride,
                          int ntaps, const int* dy, const int* dx, int ldo, void* stream) {
  DMI_REQUIRE(x && out && dy && dx, "im2col: null pointer");
  DMI_REQUIRE(C % 8 == 0 && ldo % 8 == 0 && ntaps >= 1 && ntaps <= MAX_TAPS && ldo >= ntaps * C, "im2col: need C%%8==0, ldo%%8==0, ldo>=ntaps*C (C=%d ntaps=%d ldo=%d)", C, ntaps, ldo);
  TapList t;
  t.n = ntaps;
  for (int i = 0; i < ntaps; ++i) { t.dy[i] = dy[i]; t.dx[i] = dx[i]; }
  const int64_t total = (int64_t)B * Ho * Wo * (ldo / 8);
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 65536) blocks = 65536;
  im2col_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(x, out, t, B, H, W, C, Ho, Wo, stride, ldo);
  DMI_CHECK_LAUNCH("im2col");
  return DMI_OK;
}

// out[a][t*Bn + b] = in[idx[t]][a][b], row pitch ldo >= nsel*Bn, tail zero-filled
// (weight re-layouts: [k][ci][co] -> [ci][(k',co)] for dgrad / output-parity GEMMs)
__global__ __launch_bounds__(256) void weight_gather_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, TapList sel,
                                                            int A, int Bn, int ldo) {
  const int64_t total = (int64_t)A * ldo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % ldo);
    const int a = (int)(i / ldo);
    const int t = j / Bn, b = j % Bn;
    out[i] = (t < sel.n) ? in[((int64_t)sel.dy[t] * A + a) * Bn + b] : (bf16_t)0;
  }
}
extern "C" int dmi_weight_gather(const uint16_t* in, uint16_t* out, int A, int Bn, int nsel, const int* idx, int ldo, void* stream) {
  DMI_REQUIRE(in && out && idx && nsel >= 1 && nsel <= MAX_TAPS && A > 0 && Bn > 0 && ldo >= nsel * Bn, "weight_gather: bad args");
  TapList t;
  t.n = nsel;
  for (int i = 0; i < nsel; ++i) { t.dy[i] = idx[i]; t.dx[i] = 0; }
  const int64_t total = (int64_t)A * ldo;
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 4096) blocks = 4096;
  weight_gather_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(in, out, t, A, Bn, ldo);
  DMI_CHECK_LAUNCH("weight_gather");
  return DMI_OK;
}

// Every per-step weight re-layout of a model in ONE launch (the small VAE configurations spent 50 launches of ~3 us on them):
// table[i] = {in_off, out_off, A, Bn, nsel, ldo, first_block, idx[16]} (int64; offsets in elements of in_base / out_base,
// first_block ascending); a block covers 2048 consecutive output elements of its item.
#define WG_ROW 23
__global__ __launch_bounds__(256) void weight_gather_batch_kernel(const bf16_t* __restrict__ in_base, bf16_t* __restrict__ out_base,
                                                                  const int64_t* __restrict__ table, int n) {
  const int64_t blk = blockIdx.x;
  int i = 0;
  while (i + 1 < n && table[(i + 1) * WG_ROW + 6] <= blk) ++i;
  const int64_t* row = table + (int64_t)i * WG_ROW;
  const bf16_t* in = in_base + row[0];
  bf16_t* out = out_base + row[1];
  const int A = (int)row[2], Bn = (int)row[3], nsel = (int)row[4], ldo = (int)row[5];
  const int64_t total = (int64_t)A * ldo;
  const int64_t base = (blk - row[6]) * 2048;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t e = base + k * 256 + threadIdx.x;
    if (e >= total) break;
    const int j = (int)(e % ldo);
    const int a = (int)(e / ldo);
    const int t = j / Bn, b = j % Bn;
    out[e] = (t < nsel) ? in[((int64_t)row[7 + t] * A + a) * Bn + b] : (bf16_t)0;
  }
}
extern "C" int dmi_weight_gather_batch(const uint16_t* in_base, uint16_t* out_base, const int64_t* table, int n, int64_t total_blocks,
                                       void* stream) {
  DMI_REQUIRE(in_base && out_base && table && n > 0 && total_blocks > 0, "weight_gather_batch: bad args");
  weight_gather_batch_kernel<<<dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream>>>(in_base, out_base, table, n);
  DMI_CHECK_LAUNCH("weight_gather_batch");
  return DMI_OK;
}

// out[b, 2t+py, 2u+px, :] = in[p = py*2+px][b, t, u, :]
__global__ __launch_bounds__(256) void pixel_interleave_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B,
                                                               int Ht, int Wt, int C) {
  const int cc = C / 8;
  const int64_t per = (int64_t)B * Ht * Wt * cc;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < 4 * per; idx += (int64_t)gridDim.x * 256) {
    const int p = (int)(idx / per);
    const int64_t r = idx % per;
    const int c8 = (int)(r % cc);
    const int64_t m = r / cc;
    const int u = (int)(m % Wt), t = (int)((m / Wt) % Ht), b = (int)(m / ((int64_t)Wt * Ht));
    const int py = p >> 1, px = p & 1;
    *(u32x4*)(out + ((((int64_t)b * 2 * Ht + 2 * t + py) * 2 * Wt) + 2 * u + px) * C + c8 * 8) = *(const u32x4*)(in + idx * 8);
  }
}
extern "C" int dmi_pixel_interleave(const uint16_t* in4, uint16_t* out, int B, int Ht, int Wt, int C, void* stream) {
  DMI_REQUIRE(in4 && out && C % 8 == 0, "pixel_interleave: bad args");
  const int64_t total = 4 * (int64_t)B * Ht * Wt * (C / 8);
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 65536) blocks = 65536;
  pixel_interleave_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(in4, out, B, Ht, Wt, C);
  DMI_CHECK_LAUNCH("pixel_interleave");
  return DMI_OK;
}

// image fp32 [N, Cin] -> bf16 [N, Cp] (zero padded channels)
__global__ __launch_bounds__(256) void pad_channels_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t N,
                                                           int Cin, int Cp) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N * Cp; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Cp);
    const int64_t n = i / Cp;
    out[i] = c < Cin ? f2bf(in[n * Cin + c]) : (bf16_t)0;
  }
}
extern "C" int dmi_pad_channels(const float* in, uint16_t* out, int64_t N, int Cin, int Cp, void* stream) {
  DMI_REQUIRE(in && out && N > 0 && Cin <= Cp, "pad_channels: bad args");
  int64_t blocks = cdiv64(N * Cp, 256);
  if (blocks > 16384) blocks = 16384;
  pad_channels_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(in, out, N, Cin, Cp);
  DMI_CHECK_LAUNCH("pad_channels");
  return DMI_OK;
}

// =====================================================================================
// Gumbel-softmax (src/vae_tf/layers.py:4-21): g = -log(-log u); y = softmax((logits + g)/T); hard: one-hot of the
// argmax (first max), straight-through gradient.  One wave per row; T columns (T % 8 == 0).
// Uniform noise is an INPUT (TF's Philox stream cannot be reproduced; parity is on identical noise).
// =====================================================================================
template <int NC>  // NC = ceil(T/8/64) chunks per lane
__global__ __launch_bounds__(256) void gumbel_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ u,
                                                         bf16_t* __restrict__ y, bf16_t* __restrict__ y_soft,
                                                         int* __restrict__ index, int64_t M, int T, float inv_temp, int hard,
                                                         const float* __restrict__ temp_dev) {
  if (temp_dev) inv_temp = 1.f / temp_dev[0];   // graph replay: the annealed temperature lives in device memory
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wid;
  if (row >= M) return;
  const int nch = T / 8;
  float v[NC][8];
  float mx = -INFINITY;
  int mi = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 l = *(const f32x4*)(logits + row * T + c * 8 + 4 * q);
        const f32x4 uu = *(const f32x4*)(u + row * T + c * 8 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = -logf(-logf(uu[e]));
          const float z = (l[e] + g) * inv_temp;
          v[i][4 * q + e] = z;
          const int col = c * 8 + 4 * q + e;
          if (z > mx || (z == mx && col < mi)) { mx = z; mi = col; }
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o, 64);
    const int oi = __shfl_xor(mi, o, 64);
    if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = __expf(v[i][e] - mx);
        s += v[i][e];
      }
    }
  }
  const float inv = 1.f / wave_sum(s);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      float p[8], o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        p[e] = v[i][e] * inv;
        o[e] = hard ? ((c * 8 + e == mi) ? 1.f : 0.f) : p[e];
      }
      *(u32x4*)(y_soft + row * T + c * 8) = pack8(p);
      *(u32x4*)(y + row * T + c * 8) = pack8(o);
    }
  }
  if (lane == 0 && index) index[row] = mi;
}
extern "C" int dmi_gumbel_softmax_fwd(const float* logits, const float* u, uint16_t* y, uint16_t* y_soft, int32_t* index,
                                      int64_t M, int T, float temperature, int hard, const float* temperature_dev, void* stream) {
  DMI_REQUIRE(logits && u && y && y_soft && M > 0 && T % 8 == 0 && T <= 4096 && (temperature > 0.f || temperature_dev),
              "gumbel_fwd: bad args (T=%d)", T);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv64(M, 4)), blk(256);
  const float it = temperature_dev ? 0.f : 1.f / temperature;
  if (T <= 512) gumbel_fwd_kernel<1><<<grid, blk, 0, st>>>(logits, u, y, y_soft, index, M, T, it, hard, temperature_dev);
  else if (T <= 1024) gumbel_fwd_kernel<2><<<grid, blk, 0, st>>>(logits, u, y, y_soft, index, M, T, it, hard, temperature_dev);
  else if (T <= 2048) gumbel_fwd_kernel<4><<<grid, blk, 0, st>>>(logits, u, y, y_soft, index, M, T, it, hard, temperature_dev);
  else gumbel_fwd_kernel<8><<<grid, blk, 0, st>>>(logits, u, y, y_soft, index, M, T, it, hard, temperature_dev);
  DMI_CHECK_LAUNCH("gumbel_fwd");
  return DMI_OK;
}

// dlogits = (1/T) * y_soft * (dy - sum_j dy_j y_soft_j)   (hard: straight-through => same formula on y_soft)
__global__ __launch_bounds__(256) void gumbel_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y_soft,
                                                         bf16_t* __restrict__ dlogits, int64_t M, int T, float inv_temp,
                                                         const float* __restrict__ temp_dev) {
  if (temp_dev) inv_temp = 1.f / temp_dev[0];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wid;
  if (row >= M) return;
  const int nch = T / 8;
  float dot = 0.f;
  for (int c = lane; c < nch; c += 64) {
    float a[8], p[8];
    unpack8(*(const u32x4*)(dy + row * T + c * 8), a);
    unpack8(*(const u32x4*)(y_soft + row * T + c * 8), p);
#pragma unroll
    for (int e = 0; e < 8; ++e) dot += a[e] * p[e];
  }
  dot = wave_sum(dot);
  for (int c = lane; c < nch; c += 64) {
    float a[8], p[8], o[8];
    unpack8(*(const u32x4*)(dy + row * T + c * 8), a);
    unpack8(*(const u32x4*)(y_soft + row * T + c * 8), p);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = inv_temp * p[e] * (a[e] - dot);
    *(u32x4*)(dlogits + row * T + c * 8) = pack8(o);
  }
}
extern "C" int dmi_gumbel_softmax_bwd(const uint16_t* dy, const uint16_t* y_soft, uint16_t* dlogits, int64_t M, int T,
                                      float temperature, const float* temperature_dev, void* stream) {
  DMI_REQUIRE(dy && y_soft && dlogits && M > 0 && T % 8 == 0 && (temperature > 0.f || temperature_dev), "gumbel_bwd: bad args");
  gumbel_bwd_kernel<<<dim3((unsigned)cdiv64(M, 4)), dim3(256), 0, (hipStream_t)stream>>>(
      dy, y_soft, dlogits, M, T, temperature_dev ? 0.f : 1.f / temperature, temperature_dev);
  DMI_CHECK_LAUNCH("gumbel_bwd");
  return DMI_OK;
}

// =====================================================================================
// MSE (src/vae_tf/layers.py:24-25): loss = mean((img - out)^2) over N*Cin values; out bf16 [N, Cp] (Cp >= Cin padded);
// d_out = 2 (out - img) * grad_scale / (N*Cin), pad channels 0.  partial sums -> dmi_sum_f32.
// =====================================================================================
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ img, const bf16_t* __restrict__ outp,
                                                  bf16_t* __restrict__ dout, float* __restrict__ part, int64_t N, int Cin, int Cp,
                                                  float gscale) {
  __shared__ float sm[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N * Cp; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Cp);
    const int64_t n = i / Cp;
    float d = 0.f;
    if (c < Cin) {
      d = bf2f(outp[i]) - img[n * Cin + c];
      acc += d * d;
    }
    if (dout) dout[i] = f2bf(d * gscale);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
#define MSE_BLOCKS 1024
extern "C" int64_t dmi_mse_workspace_bytes(void) { return MSE_BLOCKS * 4; }
extern "C" int dmi_sum_f32(const float* x, int64_t n, float scale, float* out, void* stream);
extern "C" int dmi_mse_loss(const float* img, const uint16_t* outp, uint16_t* dout, float* loss, int64_t N, int Cin, int Cp,
                            float grad_scale, void* workspace, void* stream) {
  DMI_REQUIRE(img && outp && loss && workspace && N > 0 && Cin <= Cp, "mse: bad args");
  const float gs = 2.f * grad_scale / ((float)N * (float)Cin);
  mse_kernel<<<dim3(MSE_BLOCKS), dim3(256), 0, (hipStream_t)stream>>>(img, outp, dout, (float*)workspace, N, Cin, Cp, gs);
  DMI_CHECK_LAUNCH("mse");
  return dmi_sum_f32((const float*)workspace, MSE_BLOCKS, 1.f / ((float)N * (float)Cin), loss, stream);
}

// dst += src  (fp32; the tied codebook receives gradients from the encoder and the decoder matmul)
__global__ __launch_bounds__(256) void add_f32_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] += src[i];
}
extern "C" int dmi_add_f32(float* dst, const float* src, int64_t n, void* stream) {
  DMI_REQUIRE(dst && src && n > 0, "add_f32: bad args");
  int64_t blocks = cdiv64(n, 256);
  if (blocks > 4096) blocks = 4096;
  add_f32_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(dst, src, n);
  DMI_CHECK_LAUNCH("add_f32");
  return DMI_OK;
}

// bf16 [N, Cp] -> fp32 [N, Cin] (reconstruction image for summaries)
__global__ __launch_bounds__(256) void unpad_channels_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int64_t N, int Cin, int Cp) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N * Cin; i += (int64_t)gridDim.x * 256)
    out[i] = bf2f(in[(i / Cin) * Cp + (i % Cin)]);
}
extern "C" int dmi_unpad_channels(const uint16_t* in, float* out, int64_t N, int Cin, int Cp, void* stream) {
  DMI_REQUIRE(in && out && N > 0 && Cin <= Cp, "unpad_channels: bad args");
  int64_t blocks = cdiv64(N * Cin, 256);
  if (blocks > 16384) blocks = 16384;
  unpad_channels_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(in, out, N, Cin, Cp);
  DMI_CHECK_LAUNCH("unpad_channels");
  return DMI_OK;
}

// =====================================================================================
// fp32 convolution for the TOKENISING encoder.  Inside dalle_model_fn the reference builds the VAE without use_bf16
// (src/model_fns.py:43-51), so the image tokens fed to DALL-E come from an fp32 encoder and an fp32 `x @ codebook`
// (src/vae_tf/models.py:81-120); argmax over bf16-computed logits flips tokens whose top-2 gap is below bf16 noise.
// This kernel is that path: implicit-im2col convolution on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32, 1/16 of
// the bf16 rate -- forward only, the image is tokenised once per step), weights in the TF layout [(tap, ci)][co] straight
// from the fp32 master buffer.
//   out[(b,oy,ox)][n] = sum_t sum_c x[b, oy*s + dy[t], ox*s + dx[t], c] * Wk[(t*C + c)][n]  + bias[n]  (ReLU) (+ residual)
// Block = 128 output pixels x 64 channels, 4 waves of 32 x 64; the K loop walks (tap, 8-channel chunk): a chunk lies in one
// tap (C % 8 == 0).  LDS: A chunk k-major [8][128] so both MFMA operands are read as 32 consecutive floats.
// The codebook product is the same kernel with one tap on a 1x1 "image".
// =====================================================================================
typedef __attribute__((ext_vector_type(4))) float v4f;
__global__ __launch_bounds__(256) void conv2d_f32_kernel(const float* __restrict__ x, const float* __restrict__ Wk,
                                                         const float* __restrict__ bias, const float* __restrict__ residual,
                                                         float* __restrict__ out, TapList taps, int B, int H, int W, int C,
                                                         int Ho, int Wo, int stride, int N, int relu) {
  __shared__ float As[8][128];
  __shared__ float Ws[8][64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t M = (int64_t)B * Ho * Wo;
  const int64_t m0 = (int64_t)blockIdx.x * 128;
  const int n0 = blockIdx.y * 64;
  // loader role: pixel p, channel half
  const int p = tid & 127, half = tid >> 7;
  const int64_t mp = m0 + p;
  const bool pok = mp < M;
  const int ox = (int)(mp % Wo), oy = (int)((mp / Wo) % Ho), b = (int)(mp / ((int64_t)Wo * Ho));
  const int wrow = tid >> 4, wc4 = (tid & 15) * 4;   // W loader (threads 0..127): row of the chunk, 4 columns
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const int cchunks = C / 8;
  for (int t = 0; t < taps.n; ++t) {
    const int iy = oy * stride + taps.dy[t], ix = ox * stride + taps.dx[t];
    const bool inside = pok && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const float* xp = x + (((int64_t)b * H + iy) * W + ix) * C + 4 * half;
    for (int cc = 0; cc < cchunks; ++cc) {
      v4f av = {0.f, 0.f, 0.f, 0.f};
      if (inside) av = *(const v4f*)(xp + cc * 8);
      v4f wv = {0.f, 0.f, 0.f, 0.f};
      if (tid < 128 && n0 + wc4 < N) wv = *(const v4f*)(Wk + ((int64_t)(t * C + cc * 8 + wrow)) * N + n0 + wc4);
      __syncthreads();   // previous chunk's fragments have been read
#pragma unroll
      for (int j = 0; j < 4; ++j) As[4 * half + j][p] = av[j];
      if (tid < 128) *(v4f*)&Ws[wrow][wc4] = wv;
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float a = As[2 * kk + (lane >> 5)][wid * 32 + (lane & 31)];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float w = Ws[2 * kk + (lane >> 5)][j * 32 + (lane & 31)];
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w, acc[j], 0, 0, 0);   // D[pixel][channel]
        }
      }
    }
  }
  // D layout: lane holds channel (lane & 31) for pixels (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + j * 32 + (lane & 31);
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t m = m0 + wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (m >= M) continue;
      float v = acc[j][e] + bv;
      if (relu) v = fmaxf(v, 0.f);
      if (residual) v += residual[m * N + n];
      out[m * N + n] = v;
    }
  }
}
extern "C" int dmi_conv2d_f32(const float* x, int B, int H, int W, int C, int Ho, int Wo, int stride, int ntaps, const int* dy,
                              const int* dx, const float* Wk, const float* bias, const float* residual, float* out, int N,
                              int relu, void* stream) {
  DMI_REQUIRE(x && Wk && out && dy && dx, "conv2d_f32: null pointer");
  DMI_REQUIRE(C % 8 == 0 && N % 4 == 0 && ntaps >= 1 && ntaps <= MAX_TAPS && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && stride >= 1,
              "conv2d_f32: need C%%8==0, N%%4==0, 1..16 taps (C=%d N=%d ntaps=%d)", C, N, ntaps);
  DMI_REQUIRE((((uintptr_t)x | (uintptr_t)Wk | (uintptr_t)out) & 15) == 0, "conv2d_f32: operands must be 16-byte aligned");
  TapList t;
  t.n = ntaps;
  for (int i = 0; i < ntaps; ++i) { t.dy[i] = dy[i]; t.dx[i] = dx[i]; }
  const int64_t M = (int64_t)B * Ho * Wo;
  conv2d_f32_kernel<<<dim3((unsigned)cdiv64(M, 128), (unsigned)((N + 63) / 64)), dim3(256), 0, (hipStream_t)stream>>>(
      x, Wk, bias, residual, out, t, B, H, W, C, Ho, Wo, stride, N, relu);
  DMI_CHECK_LAUNCH("conv2d_f32");
  return DMI_OK;
}

// tf.space_to_depth / tf.depth_to_space (NHWC, block size s; src/vae_tf/models.py:85-86,158-161), fp32, with channel
// padding: stacked[b, y, x, (dy*s + dx)*C + c] = img[b, y*s + dy, x*s + dx, c] for c < C; stacked channels [s*s*C, Cp) = 0.
__global__ __launch_bounds__(256) void space_to_depth_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int Hs,
                                                             int Ws_, int C, int s, int Cp) {
  const int64_t total = (int64_t)B * Hs * Ws_ * Cp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cp = (int)(i % Cp);
    const int64_t pix = i / Cp;
    float v = 0.f;
    if (cp < s * s * C) {
      const int c = cp % C, blk = cp / C, dy = blk / s, dx = blk % s;
      const int xx = (int)(pix % Ws_), yy = (int)((pix / Ws_) % Hs), b = (int)(pix / ((int64_t)Ws_ * Hs));
      v = img[(((int64_t)b * Hs * s + yy * s + dy) * (Ws_ * s) + xx * s + dx) * C + c];
    }
    out[i] = v;
  }
}
__global__ __launch_bounds__(256) void depth_to_space_kernel(const float* __restrict__ stacked, float* __restrict__ img, int B,
                                                             int Hs, int Ws_, int C, int s, int Cp) {
  const int64_t total = (int64_t)B * Hs * s * Ws_ * s * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    const int X = (int)(pix % (Ws_ * s)), Y = (int)((pix / (Ws_ * s)) % (Hs * s)), b = (int)(pix / ((int64_t)Ws_ * s * Hs * s));
    const int yy = Y / s, dy = Y % s, xx = X / s, dx = X % s;
    img[i] = stacked[(((int64_t)b * Hs + yy) * Ws_ + xx) * Cp + (dy * s + dx) * C + c];
  }
}
extern "C" int dmi_space_to_depth_f32(const float* img, float* stacked, int B, int Hs, int Ws_, int C, int s, int Cp, void* stream) {
  DMI_REQUIRE(img && stacked && B > 0 && Hs > 0 && Ws_ > 0 && C > 0 && s >= 1 && Cp >= s * s * C, "space_to_depth: bad args");
  int64_t blocks = cdiv64((int64_t)B * Hs * Ws_ * Cp, 256);
  if (blocks > 16384) blocks = 16384;
  space_to_depth_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(img, stacked, B, Hs, Ws_, C, s, Cp);
  DMI_CHECK_LAUNCH("space_to_depth");
  return DMI_OK;
}
extern "C" int dmi_depth_to_space_f32(const float* stacked, float* img, int B, int Hs, int Ws_, int C, int s, int Cp, void* stream) {
  DMI_REQUIRE(img && stacked && B > 0 && Hs > 0 && Ws_ > 0 && C > 0 && s >= 1 && Cp >= s * s * C, "depth_to_space: bad args");
  int64_t blocks = cdiv64((int64_t)B * Hs * s * Ws_ * s * C, 256);
  if (blocks > 16384) blocks = 16384;
  depth_to_space_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(stacked, img, B, Hs, Ws_, C, s, Cp);
  DMI_CHECK_LAUNCH("depth_to_space");
  return DMI_OK;
}
