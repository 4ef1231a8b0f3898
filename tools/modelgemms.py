"""Time every GEMM call shape of the dalle_example step with the library's own dispatch (A/B two builds via DALLE_HIP_LIB)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch, dalle_hip as dh
from kbench import timeit, rb
M, d, Vp = 40960, 512, 50816
tot = 0.0
def nt(name, N, K, flags, count):
    global tot
    A, Bt = rb(M, K), rb(N, K, scale=0.05)
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    bias, res = rb(N), rb(M, N)
    t = timeit(lambda: dh.gemm_nt(A, K, Bt, K, C, N, M, N, K, flags, bias=bias, residual=res, relu_src=res), iters=10)
    tot += t * count
    print(f"NT {name:14s} N={N:6d} K={K:6d} f={flags}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:6.0f} TF/s  x{count}", flush=True)
def tn(name, I, J, bias, count):
    global tot
    X, dY = rb(M, I), rb(M, J)
    dW = torch.empty(I, J, dtype=torch.float32, device="cuda")
    db = torch.empty(J, dtype=torch.float32, device="cuda") if bias else None
    w = torch.empty(dh.gemm_tn_workspace_bytes(M, I, J) + 1024, dtype=torch.uint8, device="cuda")
    t = timeit(lambda: dh.gemm_tn(X, I, dY, J, dW, M, I, J, w, dbias=db), iters=10)
    tot += t * count
    print(f"TN {name:14s} I={I:6d} J={J:6d} b={int(bias)}: {t*1e6:8.1f} us {2*M*I*J/t/1e12:6.0f} TF/s  x{count}", flush=True)
nt("qkv", 3 * d, d, 0, 6); nt("proj", d, d, 5, 6); nt("fc1", 4 * d, d, 3, 6); nt("fc2", d, 4 * d, 5, 6); nt("logits", Vp, d, 1, 1)
nt("d_fc2", 4 * d, d, 8, 6); nt("d_fc1", d, 4 * d, 0, 6); nt("d_proj", d, d, 0, 6); nt("d_qkv", d, 3 * d, 0, 6); nt("d_logits", d, Vp, 0, 1)
tn("w_fc2", 4 * d, d, True, 6); tn("w_fc1", d, 4 * d, True, 6); tn("w_proj", d, d, True, 6); tn("w_qkv", d, 3 * d, False, 6); tn("w_logits", d, Vp, True, 1)
print(f"sum over a step: {tot*1e3:.3f} ms")
