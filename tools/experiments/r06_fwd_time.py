"""[r06] forward attention timing at three sequence lengths with the same flops (per-item vs per-step cost); DALLE_HIP_LIB selects the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch, dalle_hip as dh
from kbench import timeit, rb
vers = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
print("lib:", os.environ.get("DALLE_HIP_LIB", "product"))
for (B, H, S) in [(128, 4, 640), (32, 4, 1280), (8, 4, 2560)]:
    d = H * 128
    qkv = rb(B * S, 3 * d, scale=0.3)
    o = torch.empty(B * S, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
    out = []
    for rep in range(2):
        for ver in vers:
            dh.set_option("attn_fwd", ver)
            tf = timeit(lambda: dh.attention_fwd(qkv, o, lse, B, H, S))
            out.append(f"v{ver} {tf*1e6:7.1f} us")
    print(f"({B},{H},{S}): " + "  ".join(out), flush=True)
