"""[r06] accuracy of the two forward kernels against fp64 math: mean / rms / max error of O in units of the output's bf16 ulp, flat and peaked rows"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch, dalle_hip as dh
from r06_attn import run_fwd


def ref64(qkv, B, H, S):
    d = H * 128
    t = qkv.double().view(B, S, 3, H, 128)
    q, k, v = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    logits = q @ k.transpose(-1, -2)
    mask = torch.triu(torch.ones(S, S, dtype=torch.bool, device=qkv.device), 1)
    logits = logits.masked_fill(mask, float("-inf"))
    o = torch.softmax(logits, -1) @ v
    return o.permute(0, 2, 1, 3).reshape(B * S, d), torch.logsumexp(logits, -1)


for (B, H, S, scale) in [(2, 4, 1280, 0.3), (2, 4, 1280, 0.6), (2, 4, 1280, 1.0), (4, 2, 640, 0.15)]:
    torch.manual_seed(7)
    d = H * 128
    qkv = (torch.randn(B * S, 3 * d, device="cuda") * scale).to(torch.bfloat16)
    o_ref, lse_ref = ref64(qkv, B, H, S)
    best = o_ref.to(torch.bfloat16)                      # the correctly rounded result
    rnd = float((best.double() - o_ref).pow(2).mean().sqrt())
    msg = f"({B},{H},{S}) scale {scale}: rms(o) {float(o_ref.pow(2).mean().sqrt()):.3f}, rms of pure bf16 rounding {rnd:.3e} |"
    for ver in (0, 1):
        o, lse = run_fwd(ver, qkv, B, H, S)
        e = o.double() - o_ref
        msg += (f"  v{ver}: rms err {float(e.pow(2).mean().sqrt()):.3e} ({float(e.pow(2).mean().sqrt()) / rnd:.3f} x rounding), max {float(e.abs().max()):.4f}, "
                f"!= correctly rounded {float((o != best).float().mean()):.4f}, lse max {float((lse.double() - lse_ref).abs().max()):.2e}")
    print(msg, flush=True)
