"""DVFS evidence for DESIGN.md §4 "[r03] What bounds the step": the SAME long-K GEMM kernel on random and on zero-filled
operands, with rocm-smi power / sclk sampled during a ~5 s loop of each.  usage: python tools/dvfsprobe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch
import dalle_hip as dh
from powerprobe_lib import run

M = 40960


def gemm(N, K, zero, nt8):
    if zero:
        A = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
        Bt = torch.zeros(N, K, dtype=torch.bfloat16, device="cuda")
    else:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        Bt = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")

    def f():
        dh.set_option("nt8", 2 if nt8 else 0)
        dh.set_option("nt4", 0)
        dh.gemm_nt(A, K, Bt, K, C, N, M, N, K, 0)
    return f, 2.0 * M * N * K


for N, K in ((4096, 4096), (2048, 2048)):
    for nt8 in (1, 0):
        for zero in (0, 1, 0, 1):
            f, fl = gemm(N, K, zero, nt8)
            run(f"{'256x256' if nt8 else '128x128'} N={N} K={K} {'zeros ' if zero else 'random'}", f, fl, secs=4.0)
dh.set_option("nt8", 1)
dh.set_option("nt4", 1)
