"""Vocabulary-head experiment: whole-batch vs row-chunked logits GEMM -> cross entropy -> dgrad GEMM, so each
chunk of logits/dlogits is consumed out of the 256 MiB Infinity Cache instead of HBM."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import dalle_hip as dh
from kbench import timeit, rb, ws

M, d, V, Vp = 40960, 512, 50771, 50816
x = rb(M, d)
Wt = rb(Vp, d, scale=0.02)        # [V, d]  (NT operand for logits)
W = Wt.t().contiguous()           # [d, V]  (NT operand for dgrad: dX = dZ . W^T -> B^T = W [d, Vp])
bias = rb(Vp)
z = torch.empty(M, Vp, dtype=torch.bfloat16, device="cuda")
dx = torch.empty(M, d, dtype=torch.bfloat16, device="cuda")
labels = torch.randint(0, V, (M,), dtype=torch.int32, device="cuda")
loss_rows = torch.empty(M, dtype=torch.float32, device="cuda")
dW = torch.empty(d, Vp, dtype=torch.float32, device="cuda")
db = torch.empty(Vp, dtype=torch.float32, device="cuda")
w = ws(dh.gemm_tn_workspace_bytes(M, d, Vp))
esz = 2


def off(t, rows, ld):
    return t.data_ptr() + rows * ld * t.element_size()


def run(chunk, tn_first=False):
    if tn_first and chunk >= M:
        pass
    for r0 in range(0, M, chunk):
        n = min(chunk, M - r0)
        dh.gemm_nt(off(x, r0, d), d, Wt, d, off(z, r0, Vp), Vp, n, Vp, d, dh.GEMM_BIAS, bias=bias)
        dh.cross_entropy(off(z, r0, Vp), Vp, off(labels, r0, 1), off(loss_rows, r0, 1), None, n, V, 1.0 / M)
        dh.gemm_nt(off(z, r0, Vp), Vp, W, Vp, off(dx, r0, d), d, n, d, Vp)
    dh.gemm_tn(x, d, z, Vp, dW, M, d, Vp, w, dbias=db)


def parts(chunk):
    ts = {}
    n = chunk
    ts["logits"] = timeit(lambda: dh.gemm_nt(x, d, Wt, d, z, Vp, n, Vp, d, dh.GEMM_BIAS, bias=bias))
    ts["ce"] = timeit(lambda: dh.cross_entropy(z, Vp, labels, loss_rows, None, n, V, 1.0 / M))
    ts["dgrad"] = timeit(lambda: dh.gemm_nt(z, Vp, W, Vp, dx, d, n, d, Vp))
    return ts


if __name__ == "__main__":
    for chunk in (M, 8192, 4096, 2048, 1024, M):
        t = timeit(lambda: run(chunk), iters=5, warm=2)
        print(f"chunk {chunk:6d}: head total {t*1e3:8.3f} ms", flush=True)
    for chunk in (M, 2048, 1024):
        ts = parts(chunk)
        print(f"isolated (back-to-back same buffer) rows={chunk}: " + "  ".join(f"{k} {v*1e6*M/chunk:9.1f} us/full-M" for k, v in ts.items()), flush=True)
    t = timeit(lambda: dh.gemm_tn(x, d, z, Vp, dW, M, d, Vp, w, dbias=db), iters=5)
    print(f"tn wgrad {t*1e6:9.1f} us")
