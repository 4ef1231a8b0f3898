"""[r06] randomised shapes for the round-6 changes against fp32 torch math and against the round-2 forms (one-off stress run, not part of the
suite): attention forward / backward at random (B, H, S) with random score scales and occasional late spikes; the fused LayerNorm backward
at random ragged M with sentinel guard rows behind every row-indexed operand (the round-5 advisor finding)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "dalle-mtf_amd"), os.path.join(ROOT, "tools")]
import torch
import dalle_hip as dh
from r06_attn import ref_attn
DEV = "cuda"
rng = random.Random(11)
bad = 0
for it in range(48):
    B, H = rng.randint(1, 3), rng.choice([1, 2, 3, 4, 8])
    S = 8 * rng.randint(1, 190)
    scale = rng.choice([0.1, 0.3, 0.3, 0.6, 1.0])
    d = H * 128
    torch.manual_seed(100 + it)
    qkv = (torch.randn(B * S, 3 * d, device=DEV) * scale).to(torch.bfloat16)
    spike = rng.random() < 0.4 and S >= 16
    if spike:
        for _ in range(rng.randint(1, 3)):
            b_, h_ = rng.randrange(B), rng.randrange(H)
            q_ = rng.randrange(1, S); k_ = rng.randrange(0, q_ + 1)
            qkv[b_ * S + k_, d + h_ * 128:d + (h_ + 1) * 128] = (qkv[b_ * S + q_, h_ * 128:(h_ + 1) * 128].float() * rng.choice([1.5, 3.0, 6.0]) / max(scale, 0.3) ** 2 * 0.3).to(torch.bfloat16)
    qf = qkv.float().clone().requires_grad_(True)
    o_ref, lse_ref = ref_attn(qf, B, H, S)
    d_o = torch.randn(B * S, d, device=DEV).to(torch.bfloat16)
    (o_ref * d_o.float()).sum().backward()
    g_ref = qf.grad
    res = {}
    for ver in (0, 1):
        dh.set_option("attn_fwd", ver); dh.set_option("attn_bwd", ver)
        o = torch.full((B * S, d), float("nan"), dtype=torch.bfloat16, device=DEV)
        lse = torch.full((B, H, S), float("nan"), dtype=torch.float32, device=DEV)
        dh.attention_fwd(qkv, o, lse, B, H, S)
        res[ver] = (o, lse)
    dh.set_option("attn_fwd", 1)
    o1, l1 = res[1]
    omax = max(1.0, float(o_ref.abs().max()))
    ok = bool(torch.isfinite(o1.float()).all() and torch.isfinite(l1).all())
    ok = ok and float((o1.float() - o_ref).abs().max()) <= 2.5e-2 * omax and float((l1 - lse_ref).abs().max()) <= 2e-3 * max(1.0, float(lse_ref.abs().max()))
    ok = ok and float((o1.float() - res[0][0].float()).abs().max()) <= 2.0 ** -6 * omax
    g = {}
    for ver in (0, 1):
        dh.set_option("attn_bwd", ver)
        dq = torch.full((B * S, 3 * d), float("nan"), dtype=torch.bfloat16, device=DEV)
        dh.attention_bwd(qkv, o1, d_o, l1, torch.zeros(3, B, H, S, dtype=torch.float32, device=DEV), dq, B, H, S)
        g[ver] = dq
    dh.set_option("attn_bwd", 1)
    gmax = max(1.0, float(g_ref.abs().max()))
    okb = torch.equal(g[0], g[1]) and bool(torch.isfinite(g[1].float()).all()) and float((g[1].float() - g_ref).abs().max()) <= 4e-2 * gmax
    bad += (not ok) + (not okb)
    print(f"attn ({B},{H},{S}) scale {scale} spike {spike}: fwd {'ok' if ok else 'FAIL'} (err {float((o1.float() - o_ref).abs().max()):.3g} of {omax:.3g}), "
          f"bwd {'ok' if okb else 'FAIL'} (err {float((g[1].float() - g_ref).abs().max()):.3g} of {gmax:.3g}, forms identical {torch.equal(g[0], g[1])})", flush=True)

def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)
for it in range(30):
    M = rng.randint(1, 3000); K = 64 * rng.randint(1, 33); N = 512; G = 176
    A, Bt = rnd(M, K, seed=it).to(DEV), rnd(N, K, scale=0.2, seed=it + 1).to(DEV)
    def guarded(t, fill):
        full = torch.full((M + G,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=DEV)
        full[:M] = t.to(DEV)
        return full, full[:M]
    xf, x = guarded(rnd(M, N, seed=it + 2), float("nan"))
    rf, dres = guarded(rnd(M, N, seed=it + 3), float("nan"))
    gam = (1 + 0.1 * torch.randn(N, generator=torch.Generator().manual_seed(it))).to(torch.bfloat16).to(DEV)
    y = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    mf, mean = guarded(torch.zeros(M), float("nan")); sf, rstd = guarded(torch.zeros(M), float("nan"))
    dh.layernorm_fwd(x, gam, torch.zeros(N, dtype=torch.bfloat16, device=DEV), y, mean, rstd, M, N)
    dy = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    dh.gemm_nt(A, K, Bt, K, dy, N, M, N, K)
    dx0 = torch.zeros_like(dy); dg0 = torch.zeros(N, device=DEV); db0 = torch.zeros(N, device=DEV)
    ws = torch.empty(int(dh.layernorm_bwd_workspace_bytes(M, N)) + 256, dtype=torch.uint8, device=DEV)
    dh.layernorm_bwd(dy, x, gam, mean, rstd, dres, dx0, dg0, db0, ws, M, N)
    SENT = 1.2345678e12
    d1f, dx1 = guarded(torch.full((M, N), float("nan"), dtype=torch.bfloat16), SENT)
    c2f, C2 = guarded(torch.full((M, N), float("nan"), dtype=torch.bfloat16), SENT)
    part = torch.empty(dh.gemm_nt_lnbwd_parts(M) * 2 * N, device=DEV)
    dg1 = torch.zeros(N, device=DEV); db1 = torch.zeros(N, device=DEV)
    B2 = rnd(N, N, scale=0.2, seed=it + 9).to(DEV)
    dh.gemm_nt_lnbwd(A, K, Bt, K, M, N, K, x, gam, mean, rstd, dres, dx1, part, dg=dg1, db=db1, B2=B2, ldb2=N, C2=C2)
    guard = torch.full((G, N), SENT, dtype=torch.bfloat16, device=DEV)
    dd = (dx1.float() - dx0.float()).abs()
    ok = (not torch.isnan(dx1.float()).any()) and (not torch.isnan(dg1).any()) and (not torch.isnan(db1).any())
    ok = ok and torch.equal(d1f[M:], guard) and torch.equal(c2f[M:], guard)
    ok = ok and float((dd > 2.0 ** -7 * dx0.float().abs() + 1e-3).float().mean()) == 0.0
    ok = ok and float((dg1 - dg0).abs().max()) <= 1e-3 * (1 + float(dg0.abs().max())) and float((db1 - db0).abs().max()) <= 1e-3 * (1 + float(db0.abs().max()))
    bad += not ok
    print(f"lnbwd M={M} (M % 160 = {M % 160}) K={K}: {'ok' if ok else 'FAIL'}", flush=True)
print("STRESS", "FAILED" if bad else "PASSED", bad)
