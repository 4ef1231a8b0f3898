"""[r06 debug] where does the round-6 forward kernel go wrong on large-magnitude scores?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch, dalle_hip as dh
from r06_attn import ref_attn, run_fwd
from kbench import timeit, rb

for (B, H, S, scale, spike) in [(1, 1, 256, 1.0, False), (1, 1, 1280, 1.0, False), (1, 1, 1280, 0.6, False), (1, 2, 1280, 1.0, True), (1, 1, 1280, 0.3, True)]:
    torch.manual_seed(S + H)
    d = H * 128
    qkv = (torch.randn(B * S, 3 * d, device="cuda") * scale).to(torch.bfloat16)
    if spike:
        qkv[700, d:d + 128] = qkv[900, :128] * 3
    o_ref, lse_ref = ref_attn(qkv, B, H, S)
    o, lse = run_fwd(1, qkv, B, H, S)
    bad_rows = torch.nonzero(torch.isnan(o.float()).any(-1) | ((o.float() - o_ref).abs().max(-1).values > 0.05 * max(1.0, float(o_ref.abs().max())))).flatten()
    bad_lse = torch.nonzero(~torch.isfinite(lse.flatten()) | ((lse.flatten() - lse_ref.flatten()).abs() > 1e-2 * (1 + lse_ref.flatten().abs()))).flatten()
    print(f"({B},{H},{S}) scale {scale} spike {spike}: bad O rows {bad_rows.numel()} {bad_rows[:40].tolist()}  bad lse {bad_lse.numel()} {bad_lse[:40].tolist()}", flush=True)
    if bad_lse.numel():
        i = int(bad_lse[0])
        print("   first bad lse: got", float(lse.flatten()[i]), "ref", float(lse_ref.flatten()[i]), " row max logit", flush=True)
        t = qkv.float().view(B, S, 3, H, 128)
        hh, rr = divmod(i, S)
        s = t[0, rr, 0, hh] @ t[0, :rr + 1, 1, hh].t()
        for j0 in range(0, rr + 1, 64):
            print(f"      tile {j0//64}: max {float(s[j0:j0+64].max()):9.2f} (log2 units {float(s[j0:j0+64].max())*1.4427:9.2f})")

# per-item overhead vs per-step cost: same flops, different sequence lengths
for (B, H, S) in [(128, 4, 640), (32, 4, 1280), (8, 4, 2560), (2, 4, 5120)]:
    d = H * 128
    qkv = rb(B * S, 3 * d, scale=0.3)
    o = torch.empty(B * S, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
    fl = 4.0 * B * H * S * S * 128 / 2
    out = []
    for rep in range(2):
        for ver in (0, 1):
            dh.set_option("attn_fwd", ver)
            tf = timeit(lambda: dh.attention_fwd(qkv, o, lse, B, H, S))
            out.append(f"v{ver} {tf*1e6:7.1f} us")
    print(f"({B},{H},{S}): " + "  ".join(out), flush=True)
