"""Sustained-load clock / power probe: runs one kernel in a loop for ~6 s while sampling rocm-smi."""
import sys, os, subprocess, threading, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch, dalle_hip as dh
from kbench import rb
M = 40960


from powerprobe_lib import run


def nt(N, K, **opts):
    A, Bt = rb(M, K), rb(N, K, scale=0.05)
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    def f():
        for k, v in opts.items(): dh.set_option(k, v)
        dh.gemm_nt(A, K, Bt, K, C, N, M, N, K, 0)
    return f, 2.0 * M * N * K


f, fl = nt(2048, 2048, nt4=0); run("nt2 128x128 N=2048 K=2048", f, fl)
f, fl = nt(2048, 2048, nt4=2); run("nt4 256x128 N=2048 K=2048", f, fl)
f, fl = nt(50816, 512, nt4=1); run("nt4 logits N=50816 K=512", f, fl)
dh.set_option("nt4", 1)
X, dY = rb(M, 2048), rb(M, 512)
dW = torch.empty(2048, 512, dtype=torch.float32, device="cuda")
w = torch.empty(dh.gemm_tn_workspace_bytes(M, 2048, 512) + 1024, dtype=torch.uint8, device="cuda")
run("tn 2048x512", lambda: dh.gemm_tn(X, 2048, dY, 512, dW, M, 2048, 512, w), 2.0 * M * 2048 * 512)
x = torch.randn(1 << 28, device="cuda")
run("torch copy 1 GiB (HBM)", lambda: x.clone(), 0.0, secs=3.0)
