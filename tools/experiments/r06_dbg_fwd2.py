"""[r06 debug] spike magnitude sweep for the round-6 forward kernel"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch, dalle_hip as dh
from r06_attn import ref_attn, run_fwd
B, H, S = 1, 1, 1280
d = 128
for keyrow, qrow in ((700, 900), (64 * 10 + 3, 900), (64 * 10 + 35, 901), (5, 70), (130, 140)):
  for mult in (0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 2.0, 3.0):
    torch.manual_seed(5)
    qkv = (torch.randn(B * S, 3 * d, device="cuda") * 1.0).to(torch.bfloat16)
    qkv[keyrow, d:2 * d] = (qkv[qrow, :128].float() * mult).to(torch.bfloat16)
    o_ref, lse_ref = ref_attn(qkv, B, H, S)
    o, lse = run_fwd(1, qkv, B, H, S)
    t = qkv.float().view(S, 3, 128)
    sp = float(t[qrow, 0] @ t[keyrow, 1]) * 1.4427
    prev = float((t[qrow, 0] @ t[:keyrow - keyrow % 64, 1].t()).max()) * 1.4427 if keyrow >= 64 else float("nan")
    bad = torch.nonzero(~torch.isfinite(lse.flatten()) | ((lse.flatten() - lse_ref.flatten()).abs() > 1e-2 * (1 + lse_ref.flatten().abs()))).flatten()
    eo = float((o.float() - o_ref).abs().nan_to_num(1e9).max())
    print(f"key {keyrow} q {qrow} mult {mult}: spike {sp:8.1f} log2 units (max of earlier tiles {prev:7.1f}) bad lse rows {bad[:8].tolist()} got {[float(lse.flatten()[i]) for i in bad[:3]]} ref {[float(lse_ref.flatten()[i]) for i in bad[:3]]} o err {eo:.3g}", flush=True)
