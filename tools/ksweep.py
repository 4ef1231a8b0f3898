"""NT GEMM variant sweep on the model's shapes (+ a K sweep separating per-tile fixed cost from the main-loop rate)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch, dalle_hip as dh
from kbench import timeit, rb
M = 40960
VAR = (("nt2", dict(nt5=0, nt4=0)), ("nt4", dict(nt5=0, nt4=2)), ("nt5/256", dict(nt5=2)), ("nt5/128", dict(nt5=3)))
shapes = [(2048, 512), (2048, 1024), (2048, 2048), (2048, 4096), (1536, 512), (512, 512), (512, 1536), (512, 2048), (50816, 512), (512, 50816)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for N, K in shapes:
    A, Bt = rb(M, K), rb(N, K, scale=0.05)
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    line = f"N={N:6d} K={K:6d}: "
    for name, opts in VAR:
        for k, v in opts.items(): dh.set_option(k, v)
        t = timeit(lambda: dh.gemm_nt(A, K, Bt, K, C, N, M, N, K, 0), iters=10 if N * K > 2e7 else 20)
        line += f"{name} {t*1e6:8.1f} us {2*M*N*K/t/1e12:6.0f} TF | "
    print(line, flush=True)
    del A, Bt, C
