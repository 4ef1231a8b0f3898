"""randomised shapes for the round-5 kernels against their claimed equalities (one-off stress run, not part of the suite)"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "dalle-mtf_amd")]
import torch
import dalle_hip as dh
DEV = "cuda"
rng = random.Random(7)
def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)
bad = 0
for it in range(40):
    # ReLU bits
    M = rng.randint(1, 3000); N = 64 * rng.randint(1, 20); K = 128 * rng.randint(1, 5)
    A, Bt, bias = rnd(M, K, seed=it), rnd(N, K, scale=0.2, seed=it + 1), rnd(N, seed=it + 2)
    dY, W2 = rnd(M, K, seed=it + 3), rnd(N, K, scale=0.2, seed=it + 4)
    h = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV); h2 = torch.zeros_like(h)
    bits = torch.full((dh.relu_bits_bytes(M, N),), 0x55, dtype=torch.uint8, device=DEV)
    dh.gemm_nt_relu_bits(A, K, Bt, K, h, N, M, N, K, bias, bits)
    dh.gemm_nt(A, K, Bt, K, h2, N, M, N, K, dh.GEMM_BIAS | dh.GEMM_RELU, bias=bias)
    g1 = torch.zeros_like(h); g2 = torch.zeros_like(h)
    dh.gemm_nt_mask_bits(dY, K, W2, K, g1, N, M, N, K, bits)
    dh.gemm_nt(dY, K, W2, K, g2, N, M, N, K, dh.GEMM_RELU_MASK, relu_src=h)
    ok = torch.equal(h, h2) and torch.equal(g1, g2)
    bad += not ok
    if not ok: print("bits mismatch", M, N, K)
for it in range(25):
    # fused LayerNorm backward (+ chained product) vs the separate kernels
    M = rng.randint(1, 4000); K = 64 * rng.randint(1, 35); N = 512
    A, Bt, x, dres = rnd(M, K, seed=it), rnd(N, K, scale=0.2, seed=it + 1), rnd(M, N, seed=it + 2), rnd(M, N, seed=it + 3)
    gam = (1 + 0.1 * torch.randn(N, generator=torch.Generator().manual_seed(it))).to(torch.bfloat16).to(DEV)
    y = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV); mean = torch.zeros(M, device=DEV); rstd = torch.zeros(M, device=DEV)
    dh.layernorm_fwd(x, gam, torch.zeros(N, dtype=torch.bfloat16, device=DEV), y, mean, rstd, M, N)
    dy = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    dh.set_option("ntr", 0); dh.gemm_nt(A, K, Bt, K, dy, N, M, N, K); dh.set_option("ntr", 1)
    dx0 = torch.zeros_like(dy); dg0 = torch.zeros(N, device=DEV); db0 = torch.zeros(N, device=DEV)
    ws = torch.empty(int(dh.layernorm_bwd_workspace_bytes(M, N)) + 256, dtype=torch.uint8, device=DEV)
    dh.layernorm_bwd(dy, x, gam, mean, rstd, dres, dx0, dg0, db0, ws, M, N)
    part = torch.empty(dh.gemm_nt_lnbwd_parts(M) * 2 * N, device=DEV)
    dx1 = torch.full_like(dy, float("nan")); dg1 = torch.zeros(N, device=DEV); db1 = torch.zeros(N, device=DEV)
    B2 = rnd(N, N, scale=0.2, seed=it + 9); C2 = torch.full_like(dy, float("nan"))
    dh.gemm_nt_lnbwd(A, K, Bt, K, M, N, K, x, gam, mean, rstd, dres, dx1, part, dg=dg1, db=db1, B2=B2, ldb2=N, C2=C2)
    Cref = torch.zeros_like(dy); dh.gemm_nt(dx1, N, B2, N, Cref, N, M, N, N)
    d = (dx1.float() - dx0.float()).abs()
    ok = (not torch.isnan(dx1.float()).any()) and float((d > 2.0 ** -7 * dx0.float().abs() + 1e-3).float().mean()) == 0.0 and torch.equal(C2, Cref)
    ok = ok and float((dg1 - dg0).abs().max()) <= 1e-3 * (1 + float(dg0.abs().max())) and float((db1 - db0).abs().max()) <= 1e-3 * (1 + float(db0.abs().max()))
    bad += not ok
    if not ok: print("lnbwd mismatch", M, K, float(d.max()))
for it in range(20):
    # grouped weight gradients vs single launches
    M = rng.randint(64, 6000); n = rng.randint(1, 4)
    probs = []
    for k in range(n):
        I, J = 8 * rng.randint(1, 80), 8 * rng.randint(1, 80)
        q = dict(X=rnd(M, I, seed=it + k), ldx=I, dY=rnd(M, J, seed=it + k + 7), ldy=J, dW=torch.zeros(I, J, device=DEV), I=I, J=J,
                 ws=torch.empty(int(dh.gemm_tn_workspace_bytes(M, I, J)) + 256, dtype=torch.uint8, device=DEV))
        if rng.random() < 0.5: q["dbias"] = torch.zeros(J, device=DEV)
        probs.append(q)
    dh.gemm_tn_group(probs, M)
    for q in probs:
        ref = q["X"].float().t() @ q["dY"].float()
        e = float((q["dW"] - ref).abs().max()) / (1e-6 + float(ref.abs().max()))
        eb = 0.0
        if "dbias" in q:
            rb = q["dY"].float().sum(0); eb = float((q["dbias"] - rb).abs().max()) / (1e-6 + float(rb.abs().max()))
        if e > 2e-3 or eb > 2e-3:
            bad += 1; print("tn_group mismatch", M, q["I"], q["J"], e, eb)
torch.cuda.synchronize()
print("stress: failures", bad)
