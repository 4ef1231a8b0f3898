"""attention fwd / bwd timing with an option toggled on and off (alternating, same buffers, identical results)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch, dalle_hip as dh
from kbench import timeit, rb
B, H, S = 32, 4, 1280
d = H * 128
qkv = rb(B * S, 3 * d, scale=0.3)
o = torch.empty(B * S, d, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
d_o = rb(B * S, d)
delta = torch.empty(3, B, H, S, dtype=torch.float32, device="cuda")
dqkv = torch.empty(B * S, 3 * d, dtype=torch.bfloat16, device="cuda")
ref = {}
for rep in range(3):
    for x in (1, 8, 16):
        dh.set_option("attn_xcd", x)
        tf = timeit(lambda: dh.attention_fwd(qkv, o, lse, B, H, S))
        tb = timeit(lambda: dh.attention_bwd(qkv, o, d_o, lse, delta, dqkv, B, H, S))
        print(f"attn_xcd={x}: fwd {tf*1e6:7.1f} us  bwd {tb*1e6:7.1f} us", flush=True)
        key = (o.float().sum().item(), dqkv.float().abs().sum().item())
        ref.setdefault("k", key)
        if key != ref["k"]:
            print("  (results differ from the first variant:", key, ref["k"], ")")
dh.set_option("attn_xcd", 8)
