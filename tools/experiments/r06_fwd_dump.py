"""[r06] dump / compare forward outputs of two builds on identical inputs: python r06_fwd_dump.py dump <path> | cmp <a> <b>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch
if sys.argv[1] == "dump":
    import dalle_hip as dh
    from r06_attn import run_fwd
    out = {}
    for (B, H, S, scale) in [(2, 4, 1280, 0.3), (1, 2, 1280, 1.0), (2, 2, 200, 0.6)]:
        torch.manual_seed(S)
        d = H * 128
        qkv = (torch.randn(B * S, 3 * d, device="cuda") * scale).to(torch.bfloat16)
        if scale == 1.0:
            qkv[700, d:d + 128] = qkv[900, :128] * 3
        o, lse = run_fwd(1, qkv, B, H, S)
        out[f"o_{S}"] = o.cpu(); out[f"lse_{S}"] = lse.cpu()
    torch.save(out, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        same = torch.equal(a[k], b[k])
        print(k, "bit-identical" if same else f"DIFFER: {int((a[k] != b[k]).sum())} of {a[k].numel()} elements, max abs {float((a[k].float() - b[k].float()).abs().max()):.3g}")
