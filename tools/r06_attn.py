"""[r06] attention kernels: correctness of the round-6 forms against fp32 torch math and against the round-2 kernels on the same
inputs (several shapes incl. ragged S), then same-process alternating timings at the benchmark shape (32, 4, 1280).
usage: python tools/r06_attn.py [check] [time]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch
import dalle_hip as dh
from kbench import timeit, rb


def ref_attn(qkv, B, H, S):
    d = H * 128
    t = qkv.float().view(B, S, 3, H, 128)
    q, k, v = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    logits = q @ k.transpose(-1, -2)
    mask = torch.triu(torch.ones(S, S, dtype=torch.bool, device=qkv.device), 1)
    logits = logits.masked_fill(mask, float("-inf"))
    lse = torch.logsumexp(logits, -1)
    o = torch.softmax(logits, -1) @ v
    return o.permute(0, 2, 1, 3).reshape(B * S, d), lse


def run_fwd(ver, qkv, B, H, S):
    d = H * 128
    o = torch.full((B * S, d), float("nan"), dtype=torch.bfloat16, device="cuda")
    lse = torch.full((B, H, S), float("nan"), dtype=torch.float32, device="cuda")
    dh.set_option("attn_fwd", ver)
    dh.attention_fwd(qkv, o, lse, B, H, S)
    torch.cuda.synchronize()
    return o, lse


def run_bwd(ver, qkv, o, d_o, lse, B, H, S):
    d = H * 128
    delta = torch.zeros(3, B, H, S, dtype=torch.float32, device="cuda")
    dqkv = torch.full((B * S, 3 * d), float("nan"), dtype=torch.bfloat16, device="cuda")
    if "attn_bwd" in OPTS:
        dh.set_option("attn_bwd", ver)
    dh.attention_bwd(qkv, o, d_o, lse, delta, dqkv, B, H, S)
    torch.cuda.synchronize()
    return dqkv


OPTS = set()
if dh.get_option("attn_bwd") >= 0:
    OPTS.add("attn_bwd")


def check():
    bad = 0
    for (B, H, S, scale) in [(1, 1, 64, 0.3), (1, 2, 128, 0.3), (2, 4, 264, 0.3), (1, 4, 1280, 0.3), (2, 1, 72, 0.5), (1, 2, 200, 0.3),
                             (3, 2, 640, 0.2), (1, 8, 1000, 0.3), (1, 1, 8, 0.3), (1, 2, 1280, 1.0)]:
        torch.manual_seed(S + H)
        d = H * 128
        qkv = (torch.randn(B * S, 3 * d, device="cuda") * scale).to(torch.bfloat16)
        if scale == 1.0:   # a spike: one key far above the rest for a few queries (forces late rescales)
            qkv[700, d:d + 128] = qkv[900, :128] * 3
        o_ref, lse_ref = ref_attn(qkv, B, H, S)
        res = {}
        for ver in (0, 1):
            o, lse = run_fwd(ver, qkv, B, H, S)
            eo = float((o.float() - o_ref).abs().max())
            el = float((lse - lse_ref).abs().max())
            nan = bool(torch.isnan(o.float()).any() or torch.isnan(lse).any())
            res[ver] = (o, lse, eo, el, nan)
        dif = float((res[0][0].float() - res[1][0].float()).abs().max())
        frac = float((res[0][0] != res[1][0]).float().mean())
        tol = 2e-2 * max(1.0, float(o_ref.abs().max()))
        ok = (not res[1][4]) and res[1][2] <= tol and res[1][3] <= 2e-3 * max(1.0, float(lse_ref.abs().max()))
        bad += not ok
        print(f"fwd ({B},{H},{S}) scale {scale}: old err o {res[0][2]:.4g} lse {res[0][3]:.3g} | new err o {res[1][2]:.4g} lse {res[1][3]:.3g} "
              f"nan {res[1][4]} | new vs old max {dif:.4g}, differing elements {frac:.4f}  {'OK' if ok else 'FAIL'}", flush=True)
        if "attn_bwd" in OPTS:
            d_o = (torch.randn(B * S, d, device="cuda")).to(torch.bfloat16)
            o1, lse1 = res[1][0], res[1][1]
            g = {}
            for ver in (0, 1):
                g[ver] = run_bwd(ver, qkv, o1, d_o, lse1, B, H, S)
            qf = qkv.float().clone().requires_grad_(True)
            oo, _ = ref_attn(qf, B, H, S)
            (oo * d_o.float()).sum().backward()
            gr = qf.grad
            e0 = float((g[0].float() - gr).abs().max())
            e1 = float((g[1].float() - gr).abs().max())
            nan = bool(torch.isnan(g[1].float()).any())
            dif = float((g[0].float() - g[1].float()).abs().max())
            frac = float((g[0] != g[1]).float().mean())
            tol = 3e-2 * max(1.0, float(gr.abs().max()))
            ok = (not nan) and e1 <= max(tol, 1.5 * e0)
            bad += not ok
            print(f"bwd ({B},{H},{S}): old err {e0:.4g} | new err {e1:.4g} nan {nan} (grad absmax {float(gr.abs().max()):.3g}) | new vs old max {dif:.4g}, "
                  f"differing {frac:.4f}  {'OK' if ok else 'FAIL'}", flush=True)
    print("CHECK", "FAILED" if bad else "PASSED", flush=True)
    return bad


def time_all():
    B, H, S = 32, 4, 1280
    d = H * 128
    qkv = rb(B * S, 3 * d, scale=0.3)
    o = torch.empty(B * S, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
    d_o = rb(B * S, d)
    delta = torch.empty(3, B, H, S, dtype=torch.float32, device="cuda")
    dqkv = torch.empty(B * S, 3 * d, dtype=torch.bfloat16, device="cuda")
    fl = 4.0 * B * H * S * S * 128 / 2     # causal half of the dense-equivalent forward flops
    for rep in range(3):
        for ver in (0, 1):
            dh.set_option("attn_fwd", ver)
            tf = timeit(lambda: dh.attention_fwd(qkv, o, lse, B, H, S))
            msg = f"attn_fwd={ver}: fwd {tf*1e6:7.1f} us ({fl/tf*1e-12:6.1f} TF/s causal-algorithmic)"
            if "attn_bwd" in OPTS:
                dh.set_option("attn_bwd", ver)
            tb = timeit(lambda: dh.attention_bwd(qkv, o, d_o, lse, delta, dqkv, B, H, S))
            msg += f"  bwd[{ver if 'attn_bwd' in OPTS else 'r02'}] {tb*1e6:7.1f} us ({2.5*fl/tb*1e-12:6.1f} TF/s)"
            print(msg, flush=True)
    dh.set_option("attn_fwd", 1)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    rc = 0
    if "check" in what:
        rc = check()
    if "time" in what:
        time_all()
    sys.exit(1 if rc else 0)
