"""why is the 128-channel 3x3 convolution of vae_coco (M = 262144, N = 128, K = 1152) at 300 TFLOP/s?  conv vs plain GEMM, taps, sizes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "dalle-mtf_amd"), os.path.join(ROOT, "tools")]
import torch
import dalle_hip as dh
from kbench import timeit, rb
DEV = "cuda"
taps9 = [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
def conv(B, H, C, N, taps, flags=0, tag=""):
    W = H
    x = rb(B * H * W, C); Wt = rb(N, len(taps) * C, scale=0.05); out = torch.empty(B * H * W, N, dtype=torch.bfloat16, device=DEV)
    bias = rb(N); res = rb(B * H * W, N)
    t = timeit(lambda: dh.conv_gemm_nt(x, B, H, W, C, H, W, 1, taps, Wt, len(taps) * C, out, N, N, flags, bias=bias if flags & 1 else None, residual=res if flags & 4 else None))
    fl = 2.0 * B * H * W * N * len(taps) * C
    print(f"conv {tag} B={B} H=W={H} C={C} N={N} taps={len(taps)} flags={flags}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)
def gemm(M, N, K, tag=""):
    A = rb(M, K); Bt = rb(N, K, scale=0.05); C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    t = timeit(lambda: dh.gemm_nt(A, K, Bt, K, C, N, M, N, K))
    print(f"gemm {tag} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF/s", flush=True)
conv(16, 128, 128, 128, taps9, 0, "3x3 128ch@128")
conv(16, 128, 128, 128, taps9, 3, "3x3 128ch@128")
conv(16, 128, 128, 128, [(0, 0)] * 9, 0, "9 x centre tap")
conv(16, 128, 128, 128, [(0, 0)], 0, "1x1")
conv(16, 64, 256, 256, taps9, 0, "3x3 256ch@64")
conv(16, 32, 512, 512, taps9, 0, "3x3 512ch@32")
conv(16, 128, 128, 256, taps9, 0, "3x3 128->256 @128")
gemm(262144, 128, 1152, "same shape, materialised A")
gemm(65536, 256, 2304)
gemm(16384, 512, 4608)
