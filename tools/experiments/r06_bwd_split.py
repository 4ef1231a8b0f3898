"""[r06] the two backward kernels timed separately (events around each launch are not available through the ABI: the pair is timed, then the
dQ kernel alone by running the pair with S chosen so that ... no: simply rocprof-free -- time attention_bwd, and time it again with the dK/dV
launch being the only difference between two builds).  Here: pair time per build; DALLE_HIP_LIB selects the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch, dalle_hip as dh
from kbench import timeit, rb
print("lib:", os.environ.get("DALLE_HIP_LIB", "product"))
B, H, S = 32, 4, 1280
d = H * 128
qkv = rb(B * S, 3 * d, scale=0.3)
o = torch.empty(B * S, d, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
d_o = rb(B * S, d)
delta = torch.empty(3, B, H, S, dtype=torch.float32, device="cuda")
dqkv = torch.empty(B * S, 3 * d, dtype=torch.bfloat16, device="cuda")
dh.attention_fwd(qkv, o, lse, B, H, S)
for rep in range(3):
    tb = timeit(lambda: dh.attention_bwd(qkv, o, d_o, lse, delta, dqkv, B, H, S))
    print(f"bwd pair {tb*1e6:7.1f} us", flush=True)
