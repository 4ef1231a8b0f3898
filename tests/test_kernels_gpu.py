"""Per-kernel parity on a real MI355X: every C-ABI entry point vs plain fp32 math on the SAME bf16-rounded
inputs (so the only differences are accumulation order and the final bf16 rounding).
Tolerances: bf16 outputs rtol 1.6e-2 (2 ulp of bf16) + small atol; fp32 outputs 2e-3 relative to the
contraction's magnitude; integer paths bit-exact."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import dalle_hip as dh  # noqa: E402  (path set up by conftest)

DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16)


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def close(got, ref, rtol, atol, what=""):
    got = got.float().cpu()
    ref = ref.float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = (err > tol)
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} out of tol; max err {float(err.max()):.4g} " \
                          f"at ref {float(ref.flatten()[err.argmax()]):.4g}; ref rms {float(ref.pow(2).mean().sqrt()):.4g}"


def ws(nbytes):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=DEV)


# ------------------------------------------------------------------ embedding

def test_embed_fwd_bwd():
    B, S, d, V = 3, 40, 512, 1000
    g = torch.Generator().manual_seed(1)
    tok = torch.randint(0, V, (B, S), generator=g, dtype=torch.int32)
    tok[:, -7:] = 5  # repeated ids -> atomic collisions
    wte, wpe = rnd(V, d, scale=0.02, seed=2), rnd(S, d, scale=0.01, seed=3)
    x = torch.empty(B * S, d, dtype=torch.bfloat16, device=DEV)
    dh.embed_fwd(tok.to(DEV), wte.to(DEV), wpe.to(DEV), x, S, d, V)
    ref = wte.float()[tok.long()] + wpe.float()[None, :S]
    close(x.view(B, S, d), ref, 8e-3, 1e-6, "embed_fwd")
    dx = rnd(B * S, d, seed=4)
    dwte = torch.full((V, d), 7.0, dtype=torch.float32, device=DEV)   # must be overwritten, zeros for absent ids
    dwpe = torch.empty(S, d, dtype=torch.float32, device=DEV)
    st, perm = _sorted(tok.view(-1).to(DEV), V)
    dh.embed_bwd(st, perm, dx.to(DEV), dwte, dwpe, B, S, d, V, ws(dh.embed_bwd_workspace_bytes(B, S, d)))
    ref_wte = torch.zeros(V, d).index_add_(0, tok.view(-1).long(), dx.float())
    close(dwte, ref_wte, 1e-5, 1e-5, "embed_bwd wte")
    close(dwpe, dx.float().view(B, S, d).sum(0), 1e-5, 1e-5, "embed_bwd wpe")


def _sorted(tok_flat, V):
    n = tok_flat.numel()
    st = torch.empty(n, dtype=torch.int32, device=DEV)
    perm = torch.empty(n, dtype=torch.int32, device=DEV)
    dh.sort_tokens(tok_flat, st, perm, n, V, ws(dh.sort_tokens_workspace_bytes(n)))
    return st, perm


@pytest.mark.parametrize("n,V", [(1, 5), (37, 16), (1024, 17), (5000, 50771), (40960, 50771), (2049, 70000), (4097, 2)])
def test_sort_tokens_is_stable(n, V):
    """multi-block radix sort (count / scan / scatter per 4-bit digit): bit-identical to a stable sort (ids and source
    positions), for 1..5 digit passes, ragged and empty per-thread chunks and heavy duplicates (the padding id)."""
    g = torch.Generator().manual_seed(n)
    tok = torch.randint(0, V, (n,), generator=g, dtype=torch.int32)
    tok[n // 3: 2 * n // 3] = V - 1
    st, perm = _sorted(tok.to(DEV), V)
    rs, rp = torch.sort(tok, stable=True)
    assert torch.equal(st.cpu(), rs) and torch.equal(perm.cpu().long(), rp)


def test_embed_bwd_long_runs_deterministic():
    """runs of one id spanning many 32-position chunks (padding) + runs ending exactly on chunk borders: equals the fp32
    index_add, and two launches agree bit for bit (no atomics)."""
    B, S, d, V = 4, 640, 256, 300
    g = torch.Generator().manual_seed(5)
    tok = torch.randint(0, V, (B, S), generator=g, dtype=torch.int32)
    tok[:, 100:500] = V - 1          # 1600 positions of one id
    tok[0, :64] = 7                  # exactly two chunks' worth of another id
    tok[1, :32] = 9
    dx = rnd(B * S, d, seed=6).to(DEV)
    st, perm = _sorted(tok.view(-1).to(DEV), V)
    outs = []
    for _ in range(2):
        dwte = torch.full((V, d), float("nan"), dtype=torch.float32, device=DEV)
        dwpe = torch.empty(S, d, dtype=torch.float32, device=DEV)
        dh.embed_bwd(st, perm, dx, dwte, dwpe, B, S, d, V, ws(dh.embed_bwd_workspace_bytes(B, S, d)))
        outs.append(dwte.cpu())
    assert torch.equal(outs[0], outs[1])
    ref = torch.zeros(V, d).index_add_(0, tok.view(-1).long(), dx.float().cpu())
    close(outs[0], ref, 1e-5, 2e-4, "embed_bwd long runs")


# ------------------------------------------------------------------ layernorm

@pytest.mark.parametrize("rows,d", [(7, 512), (130, 1024), (66, 2048), (5, 256)])
def test_layernorm_fwd_bwd(rows, d):
    x, g, b = rnd(rows, d, seed=1), bf(1 + 0.1 * torch.randn(d)), bf(0.1 * torch.randn(d))
    y = torch.empty(rows, d, dtype=torch.bfloat16, device=DEV)
    mean = torch.empty(rows, dtype=torch.float32, device=DEV)
    rstd = torch.empty(rows, dtype=torch.float32, device=DEV)
    dh.layernorm_fwd(x.to(DEV), g.to(DEV), b.to(DEV), y, mean, rstd, rows, d)
    xf = x.float().requires_grad_(True)
    gf, bff = g.float().requires_grad_(True), b.float().requires_grad_(True)
    ref = F.layer_norm(xf, (d,), gf, bff, 1e-5)
    close(y, ref.detach(), 1e-2, 1e-2, "ln_fwd")
    close(mean, x.float().mean(-1), 1e-5, 1e-5, "ln mean")
    dy = rnd(rows, d, seed=5)
    dres = rnd(rows, d, seed=6)
    ref.backward(dy.float())
    dx = torch.empty(rows, d, dtype=torch.bfloat16, device=DEV)
    dg = torch.empty(d, dtype=torch.float32, device=DEV)
    db = torch.empty(d, dtype=torch.float32, device=DEV)
    w = ws(dh.layernorm_bwd_workspace_bytes(rows, d))
    dh.layernorm_bwd(dy.to(DEV), x.to(DEV), g.to(DEV), mean, rstd, dres.to(DEV), dx, dg, db, w, rows, d)
    close(dx, xf.grad + dres.float(), 1.6e-2, 2e-2, "ln_bwd dx")
    close(dg, gf.grad, 1e-3, 1e-3 * math.sqrt(rows), "ln_bwd dg")
    close(db, bff.grad, 1e-3, 1e-3 * math.sqrt(rows), "ln_bwd db")
    dx2 = torch.empty_like(dx)
    dh.layernorm_bwd(dy.to(DEV), x.to(DEV), g.to(DEV), mean, rstd, None, dx2, dg, db, w, rows, d)
    close(dx2, xf.grad, 1.6e-2, 2e-2, "ln_bwd dx (no residual)")


def test_layernorm_bwd_deferred_finish_batch():
    """[r05-prep] dmi_layernorm_bwd with dg = db = NULL leaves the per-block partials in the workspace; one
    dmi_layernorm_bwd_finish_batch launch reduces several LayerNorms -- bit-identical to the immediate form (same summation order),
    different row counts per item, 1 and 15 items."""
    d = 512
    cases = [(4097, 1), (64, 2), (40960, 3)] + [(256 + 32 * i, 10 + i) for i in range(13)]
    ref, items, keep = [], [], []
    for rows, seed in cases:
        x, dy, g = rnd(rows, d, seed=seed), rnd(rows, d, seed=seed + 100), bf(1 + 0.1 * torch.randn(d))
        xd, dyd, gd = x.to(DEV), dy.to(DEV), g.to(DEV)
        y = torch.zeros(rows, d, dtype=torch.bfloat16, device=DEV)
        mean = torch.zeros(rows, dtype=torch.float32, device=DEV)
        rstd = torch.zeros(rows, dtype=torch.float32, device=DEV)
        dh.layernorm_fwd(xd, gd, torch.zeros(d, dtype=torch.bfloat16, device=DEV), y, mean, rstd, rows, d)
        dx0, dx1 = (torch.zeros(rows, d, dtype=torch.bfloat16, device=DEV) for _ in range(2))
        dg0, db0, dg1, db1 = (torch.full((d,), float("nan"), dtype=torch.float32, device=DEV) for _ in range(4))
        ws0, ws1 = (ws(dh.layernorm_bwd_workspace_bytes(rows, d)) for _ in range(2))
        dh.layernorm_bwd(dyd, xd, gd, mean, rstd, None, dx0, dg0, db0, ws0, rows, d)
        dh.layernorm_bwd(dyd, xd, gd, mean, rstd, None, dx1, None, None, ws1, rows, d)
        assert torch.equal(dx0, dx1)
        ref.append((dg0, db0))
        items.append((ws1, dg1, db1, rows))
        keep.append((xd, dyd, gd, mean, rstd))
    dh.layernorm_bwd_finish_batch(items[:1], d)
    dh.layernorm_bwd_finish_batch(items[1:], d)
    for (dg0, db0), (_, dg1, db1, _) in zip(ref, items):
        assert torch.equal(dg0, dg1) and torch.equal(db0, db1)


@pytest.mark.parametrize("M,K,with_res", [(160, 512, True), (1000, 1536, True), (2100, 2048, False), (40960, 1536, True), (333, 256, True),
                                          (512, 1536, True), (161, 512, False), (2559, 512, True)])
def test_gemm_nt_lnbwd_fused_layernorm_backward(M, K, with_res):
    """[r05] dmi_gemm_nt_lnbwd (full-row tiles, N = 512): dx, dgamma, dbeta equal dmi_gemm_nt followed by dmi_layernorm_bwd up to the
    summation order of the reductions (dx within one bf16 ulp almost everywhere), and fp32 autograd of LayerNorm applied to the
    bf16-rounded product; ragged last tile, K with (K / 32) % 3 == 2."""
    N = 512
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.2, seed=2)
    x, dres = rnd(M, N, seed=3), rnd(M, N, seed=4)
    gam = bf(1 + 0.1 * torch.randn(N, generator=torch.Generator().manual_seed(5)))
    Ad, Bd, gd = A.to(DEV), Bt.to(DEV), gam.to(DEV)
    # [r06] every row-indexed operand of the fused call is the first M rows of an allocation with G sentinel rows behind it: NaN behind the
    # inputs (a read past M would poison dgamma / dbeta through the column partials), a bit pattern behind dx that must survive the call
    # (advisor finding, round 5: on a ragged last tile the row-tile term sat in the scalar offset, outside the descriptor's range check)
    G = 192
    def guarded(t, fill):
        full = torch.full((M + G,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=DEV)
        full[:M] = t.to(DEV)
        return full, full[:M]
    x_full, xd = guarded(x, float("nan"))
    r_full, rd = guarded(dres, float("nan"))
    if not with_res:
        rd = None
    y = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    mean_full, mean = guarded(torch.zeros(M, dtype=torch.float32), float("nan"))
    rstd_full, rstd = guarded(torch.zeros(M, dtype=torch.float32), float("nan"))
    dh.layernorm_fwd(xd, gd, torch.zeros(N, dtype=torch.bfloat16, device=DEV), y, mean, rstd, M, N)
    # two-kernel form
    dy = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    dh.gemm_nt(Ad, K, Bd, K, dy, N, M, N, K)
    dx0 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    dg0, db0 = (torch.zeros(N, dtype=torch.float32, device=DEV) for _ in range(2))
    dh.layernorm_bwd(dy, xd, gd, mean, rstd, rd, dx0, dg0, db0, ws(dh.layernorm_bwd_workspace_bytes(M, N)), M, N)
    # fused
    SENT = 1.2345678e12    # (bf16(SENT) is a fixed bit pattern no product here produces)
    dx1_full, dx1 = guarded(torch.full((M, N), float("nan"), dtype=torch.bfloat16), SENT)
    dg1, db1 = (torch.full((N,), float("nan"), dtype=torch.float32, device=DEV) for _ in range(2))
    part = torch.full((dh.gemm_nt_lnbwd_parts(M) * 2 * N,), float("nan"), dtype=torch.float32, device=DEV)
    dh.gemm_nt_lnbwd(Ad, K, Bd, K, M, N, K, xd, gd, mean, rstd, rd, dx1, part, dg=dg1, db=db1)
    assert not torch.isnan(dx1.float()).any() and not torch.isnan(dg1).any() and not torch.isnan(db1).any()
    guard = torch.full((G, N), SENT, dtype=torch.bfloat16, device=DEV)
    assert torch.equal(dx1_full[M:], guard), "dmi_gemm_nt_lnbwd stored dx rows past M"
    assert torch.isnan(x_full[M:].float()).all() and torch.isnan(mean_full[M:]).all() and torch.isnan(rstd_full[M:]).all()
    d = (dx1.float() - dx0.float()).abs()
    assert float((d > 2.0 ** -7 * dx0.float().abs() + 1e-3).float().mean()) == 0.0, "dx differs from the two-kernel form by more than one bf16 ulp"
    assert float((dx1 != dx0).float().mean()) < 2e-2, float((dx1 != dx0).float().mean())
    close(dg1, dg0.cpu(), 1e-4, 1e-4 * math.sqrt(M), "dgamma vs two-kernel form")
    close(db1, db0.cpu(), 1e-4, 1e-4 * math.sqrt(M), "dbeta vs two-kernel form")
    # fp32 autograd on the rounded product
    xf = x.float().requires_grad_(True)
    gf, bff = gam.float().requires_grad_(True), torch.zeros(N, requires_grad=True)
    F.layer_norm(xf, (N,), gf, bff, 1e-5).backward(dy.float().cpu())
    close(dx1, xf.grad + (dres.float() if with_res else 0), 1.6e-2, 2e-2, "dx vs autograd")
    close(dg1, gf.grad, 2e-3, 2e-3 * math.sqrt(M), "dgamma vs autograd")
    close(db1, bff.grad, 2e-3, 2e-3 * math.sqrt(M), "dbeta vs autograd")
    with pytest.raises(dh.DalleHipError):
        dh.gemm_nt_lnbwd(Ad, K, Bd, K, M, 256, K, xd, gd, mean, rstd, rd, dx1, part)
    # chained form: the product that consumes dx in the same launch -- dx, the partials and C2 = dx . B2^T bit-identical to the
    # unchained call followed by dmi_gemm_nt on the stored dx
    B2 = rnd(N, N, scale=0.2, seed=7).to(DEV)
    dx2_full, dx2 = guarded(torch.full((M, N), float("nan"), dtype=torch.bfloat16), SENT)
    C2_full, C2 = guarded(torch.full((M, N), float("nan"), dtype=torch.bfloat16), SENT)
    dg2, db2 = (torch.full((N,), float("nan"), dtype=torch.float32, device=DEV) for _ in range(2))
    dh.gemm_nt_lnbwd(Ad, K, Bd, K, M, N, K, xd, gd, mean, rstd, rd, dx2, part, dg=dg2, db=db2, B2=B2, ldb2=N, C2=C2)
    assert torch.equal(dx2, dx1) and torch.equal(dg2, dg1) and torch.equal(db2, db1)
    Cref = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    dh.gemm_nt(dx1, N, B2, N, Cref, N, M, N, N)
    assert torch.equal(C2, Cref), "the chained product must equal dmi_gemm_nt on the stored dx"
    assert torch.equal(dx2_full[M:], guard) and torch.equal(C2_full[M:], guard), "the chained form stored rows past M"


def _gemm_ref(A, Bt, bias=None, relu=False, residual=None, relu_src=None):
    C = A.float() @ Bt.float().t()
    if bias is not None:
        C = C + bias.float()
    if relu:
        C = torch.relu(C)
    if residual is not None:
        C = C + residual.float()
    if relu_src is not None:
        C = C * (relu_src.float() > 0)
    return C


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 128), (1024, 512, 512), (257, 1160, 192), (130, 128, 448)])
def test_gemm_nt_plain(M, N, K):
    A, Bt = rnd(M, K, seed=1), rnd(N, K, seed=2)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    dh.gemm_nt(A.to(DEV), K, Bt.to(DEV), K, C, N, M, N, K)
    close(C, _gemm_ref(A, Bt), 1.6e-2, 2e-2 * math.sqrt(K / 64), "gemm_nt")


def test_gemm_nt_epilogues():
    M, N, K = 384, 640, 256
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.1, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    hsrc = torch.relu(rnd(M, N, seed=5))
    rs = torch.rand(M, generator=torch.Generator().manual_seed(6)) * 3 - 1
    Ad, Bd = A.to(DEV), Bt.to(DEV)
    tol = dict(rtol=1.6e-2, atol=2e-2)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    dh.gemm_nt(Ad, K, Bd, K, C, N, M, N, K, dh.GEMM_BIAS, bias=bias.to(DEV))
    close(C, _gemm_ref(A, Bt, bias), what="bias", **tol)
    dh.gemm_nt(Ad, K, Bd, K, C, N, M, N, K, dh.GEMM_BIAS | dh.GEMM_RELU, bias=bias.to(DEV))
    close(C, _gemm_ref(A, Bt, bias, relu=True), what="bias+relu", **tol)
    dh.gemm_nt(Ad, K, Bd, K, C, N, M, N, K, dh.GEMM_BIAS | dh.GEMM_RESIDUAL, bias=bias.to(DEV), residual=res.to(DEV))
    close(C, _gemm_ref(A, Bt, bias, residual=res), what="bias+residual", **tol)
    dh.gemm_nt(Ad, K, Bd, K, C, N, M, N, K, dh.GEMM_RELU_MASK, relu_src=hsrc.to(DEV))
    close(C, _gemm_ref(A, Bt, relu_src=hsrc), what="relu mask", **tol)
    dh.gemm_nt(Ad, K, Bd, K, C, N, M, N, K, dh.GEMM_ROWSCALE, rowscale=rs.to(DEV))
    close(C, _gemm_ref(A, Bt) * rs[:, None], what="row scale", **tol)
    Cf = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    dh.gemm_nt(Ad, K, Bd, K, Cf, N, M, N, K, dh.GEMM_OUT_F32)
    close(Cf, _gemm_ref(A, Bt), 1e-3, 1e-3, "f32 out")
    # strided operands (lda > K): the QKV-style slices
    A2 = rnd(M, 3 * K, seed=7)
    dh.gemm_nt(A2.to(DEV)[:, K:], 3 * K, Bd, K, C, N, M, N, K)
    close(C, _gemm_ref(A2[:, K:2 * K], Bt), what="strided A", **tol)


@pytest.mark.parametrize("M,N,K,flags", [(32, 512, 512, 0), (32, 1536, 512, 0), (3, 2048, 512, 3), (32, 512, 2048, 5), (17, 48, 64, 1),
                                         (1, 16, 192, 5), (32, 512, 576, 3)])
def test_gemm_nt_skinny_rows(M, N, K, flags):
    """M <= 32 products (the decode step) run on the weight-streaming kernel (8 waves split K, 16 columns per block): against
    the fp32 product and against the tiled kernel (same inputs, different summation order -> bf16-rounding differences only);
    lda / ldc larger than the logical widths as in the decode step."""
    A, Bt = rnd(M, K + 64, seed=1), rnd(N, K, scale=0.2, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N + 16, seed=4)
    kw = dict(bias=bias.to(DEV) if flags & 1 else None, residual=res.to(DEV) if flags & 4 else None)
    ref = _gemm_ref(A[:, :K], Bt, bias if flags & 1 else None, relu=bool(flags & 2), residual=res[:, :N] if flags & 4 else None)
    outs = []
    for sk in (1, 0):
        dh.set_option("skinny", sk)
        C = torch.full((M + 1, N + 16), 7.0, dtype=torch.bfloat16, device=DEV)
        dh.gemm_nt(A.to(DEV), K + 64, Bt.to(DEV), K, C, N + 16, M, N, K, flags, **kw)
        close(C[:M, :N], ref, 1.6e-2, 2e-2 * math.sqrt(K / 64), f"gemm_nt skinny={sk}")
        assert bool((C[M:] == 7.0).all()) and bool((C[:, N:] == 7.0).all()), "wrote outside the [M, N] block"
        outs.append(C[:M, :N].float().cpu())
    dh.set_option("skinny", 1)
    scale = float(ref.abs().max())
    assert float((outs[0] - outs[1]).abs().max()) <= 1.6e-2 * scale


@pytest.mark.parametrize("M,N,K,flags", [(32, 1536, 512, 0), (5, 2048, 512, 3), (32, 512, 128, 0), (17, 64, 2048, 1), (1, 16, 64, 3), (32, 48, 1024, 0)])
def test_ln_gemm_nt_decode_shapes(M, N, K, flags):
    """dmi_ln_gemm_nt (LayerNorm in the prologue of the M <= 32 product) against fp32 math with the bf16 rounding of LN's output,
    and against the two separate kernels (dmi_layernorm_fwd + dmi_gemm_nt): same rounding points, so the results differ by
    summation order only."""
    X = rnd(M, K + 8, scale=2.0, seed=1) + 0.7
    gam, bet = rnd(K, seed=2) + 1.0, rnd(K, scale=0.3, seed=3)
    Bt, bias = rnd(N, K, scale=0.2, seed=4), rnd(N, seed=5)
    xf = X[:, :K].float()
    mu = xf.mean(-1, keepdim=True)
    xn = ((xf - mu) * torch.rsqrt(((xf - mu) ** 2).mean(-1, keepdim=True) + 1e-5) * gam.float() + bet.float()).to(torch.bfloat16)
    ref = _gemm_ref(xn, Bt, bias if flags & 1 else None, relu=bool(flags & 2))
    Xd, Bd = X.to(DEV), Bt.to(DEV)
    C = torch.full((M + 1, N + 16), 7.0, dtype=torch.bfloat16, device=DEV)
    dh.ln_gemm_nt(Xd, K + 8, gam.to(DEV), bet.to(DEV), Bd, K, C, N + 16, M, N, K, flags, bias=bias.to(DEV) if flags & 1 else None)
    close(C[:M, :N], ref, 1.6e-2, 3e-2 * math.sqrt(K / 64), "ln_gemm_nt")
    assert bool((C[M:] == 7.0).all()) and bool((C[:, N:] == 7.0).all()), "wrote outside the [M, N] block"
    y = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    st = [torch.empty(M, dtype=torch.float32, device=DEV) for _ in range(2)]
    dh.layernorm_fwd(Xd[:, :K].contiguous(), gam.to(DEV), bet.to(DEV), y, st[0], st[1], M, K)
    assert float((y.float().cpu() - xn.float()).abs().max()) <= 2e-2 * float(xn.float().abs().max())
    C2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    dh.gemm_nt(y, K, Bd, K, C2, N, M, N, K, flags, bias=bias.to(DEV) if flags & 1 else None)
    scale = float(ref.abs().max())
    assert float((C[:M, :N].float() - C2.float()).abs().max()) <= 1.6e-2 * scale
    with pytest.raises(dh.DalleHipError):      # not a decode-step shape: refused, not silently slow
        dh.ln_gemm_nt(Xd, K + 8, gam.to(DEV), bet.to(DEV), Bd, K, C, N + 16, 33, N, K, 0)


def test_embed_fwd_position_from_device_memory():
    """pos_dev of dmi_embed_fwd: every row takes wpe[*pos_dev] (the decode step) -- equals the by-value form on a one-row table"""
    B, d, V, S = 6, 128, 50, 40
    toks = torch.randint(0, V, (B,), dtype=torch.int32)
    wte, wpe = rnd(V, d, seed=1), rnd(S, d, seed=2)
    a = torch.empty(B, d, dtype=torch.bfloat16, device=DEV)
    b = torch.empty(B, d, dtype=torch.bfloat16, device=DEV)
    for pos in (0, 17, S - 1):
        dh.embed_fwd(toks.to(DEV), wte.to(DEV), wpe.to(DEV)[pos:pos + 1], a, 1, d, V)
        dh.embed_fwd(toks.to(DEV), wte.to(DEV), wpe.to(DEV), b, S, d, V, pos_dev=torch.tensor([pos], dtype=torch.int32, device=DEV))
        assert torch.equal(a, b)
        close(a, wte[toks.long()].float() + wpe[pos].float(), 1e-2, 1e-2, "embed at pos")
    dh.embed_fwd(toks.to(DEV), wte.to(DEV), wpe.to(DEV), b, S, d, V, pos_dev=torch.tensor([S + 5], dtype=torch.int32, device=DEV))
    assert torch.equal(a, b)            # a position past the table is clamped to its last row, never read out of bounds


@pytest.mark.parametrize("M,N,K,flags", [(300, 256, 128, 0), (1000, 1160, 192, 1), (4000, 2568, 128, 5), (2048, 1024, 512, 3),
                                         (515, 136, 64, 8), (700, 264, 128, 32)])
def test_gemm_nt4_tile_256(M, N, K, flags):
    """256x128x32-tile kernel (forced), incl. M/N tails: same results as the 128x128x64 kernel (identical k order)."""
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.2, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    hsrc = torch.relu(rnd(M, N, seed=5))
    rs = torch.rand(M, generator=torch.Generator().manual_seed(6)) + 0.5
    kw = dict(bias=bias.to(DEV) if flags & 1 else None, residual=res.to(DEV) if flags & 4 else None,
              relu_src=hsrc.to(DEV) if flags & 8 else None, rowscale=rs.to(DEV) if flags & 32 else None)
    ref = _gemm_ref(A, Bt, bias if flags & 1 else None, relu=bool(flags & 2), residual=res if flags & 4 else None,
                    relu_src=hsrc if flags & 8 else None)
    if flags & 32:
        ref = ref * rs[:, None]
    outs = []
    for nt4 in (2, 0):
        dh.set_option("nt4", nt4)
        C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        dh.gemm_nt(A.to(DEV), K, Bt.to(DEV), K, C, N, M, N, K, flags, **kw)
        close(C, ref, 1.6e-2, 2e-2 * math.sqrt(K / 64), f"gemm_nt nt4={nt4}")
        outs.append(C.cpu())
    dh.set_option("nt4", 1)
    assert torch.equal(outs[0], outs[1]), "256-row-tile and 128-row-tile kernels must be bit-identical"


@pytest.mark.parametrize("M,N,K,flags", [(300, 256, 128, 0), (1000, 1160, 192, 1), (2100, 520, 320, 5), (512, 512, 1024, 3),
                                         (515, 136, 64, 8), (700, 264, 128, 32), (256, 256, 64, 16)])
def test_gemm_nt8_tile_256x256(M, N, K, flags):
    """256x256x64-tile 8-wave kernel (forced), incl. M/N tails, odd K-step counts and every epilogue: bit-identical to the
    128x128x64 kernel (identical k order)."""
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.2, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    hsrc = torch.relu(rnd(M, N, seed=5))
    rs = torch.rand(M, generator=torch.Generator().manual_seed(6)) + 0.5
    kw = dict(bias=bias.to(DEV) if flags & 1 else None, residual=res.to(DEV) if flags & 4 else None,
              relu_src=hsrc.to(DEV) if flags & 8 else None, rowscale=rs.to(DEV) if flags & 32 else None)
    ref = _gemm_ref(A, Bt, bias if flags & 1 else None, relu=bool(flags & 2), residual=res if flags & 4 else None,
                    relu_src=hsrc if flags & 8 else None)
    if flags & 32:
        ref = ref * rs[:, None]
    outs = []
    dh.set_option("nt4", 0)
    try:
        for nt8 in (2, 0):
            dh.set_option("nt8", nt8)
            C = torch.zeros(M, N, dtype=torch.float32 if flags & 16 else torch.bfloat16, device=DEV)
            dh.gemm_nt(A.to(DEV), K, Bt.to(DEV), K, C, N, M, N, K, flags, **kw)
            close(C, ref, 1.6e-2, 2e-2 * math.sqrt(K / 64), f"gemm_nt nt8={nt8}")
            outs.append(C.cpu())
    finally:
        dh.set_option("nt8", 1)
        dh.set_option("nt4", 1)
    assert torch.equal(outs[0], outs[1]), "256x256 and 128x128 tile kernels must be bit-identical"


@pytest.mark.parametrize("M,N,K,flags", [(300, 256, 128, 0), (1000, 1160, 256, 1), (2100, 520, 384, 3), (515, 136, 128, 8),
                                         (700, 264, 128, 32), (6000, 5000, 512, 0), (40960, 1536, 512, 0)])
def test_gemm_nt8p_persistent_tile_256x256(M, N, K, flags):
    """[r04] persistent 256x256 kernel with the register epilogue (forced): several tiles per block incl. M / N tails, one
    tile per block, fewer tiles than blocks, every register-form epilogue -- bit-identical to the 128x128x64 kernel (same k
    order; the register epilogue rounds the same fp32 values)."""
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.2, seed=2)
    bias = rnd(N, seed=3)
    hsrc = torch.relu(rnd(M, N, seed=5)) if flags & 8 else None
    rs = torch.rand(M, generator=torch.Generator().manual_seed(6)) + 0.5
    kw = dict(bias=bias.to(DEV) if flags & 1 else None, relu_src=hsrc.to(DEV) if flags & 8 else None,
              rowscale=rs.to(DEV) if flags & 32 else None)
    Ad, Bd = A.to(DEV), Bt.to(DEV)
    outs = []
    dh.set_option("nt4", 0)
    dh.set_option("nt8", 0)
    try:
        for p8 in (2, 0):
            dh.set_option("nt8p", p8)
            C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            dh.gemm_nt(Ad, K, Bd, K, C, N, M, N, K, flags, **kw)
            outs.append(C)
    finally:
        dh.set_option("nt8p", 1)
        dh.set_option("nt8", 1)
        dh.set_option("nt4", 1)
    if M * N <= 4_000_000:
        ref = _gemm_ref(A, Bt, bias if flags & 1 else None, relu=bool(flags & 2), relu_src=hsrc)
        if flags & 32:
            ref = ref * rs[:, None]
        close(outs[0], ref, 1.6e-2, 2e-2 * math.sqrt(K / 64), "gemm_nt nt8p")
    assert torch.equal(outs[0], outs[1]), "persistent 256x256 and 128x128 tile kernels must be bit-identical"


@pytest.mark.parametrize("M,shapes", [(40960, [(512, 512, True), (512, 1536, False)]), (2560, [(512, 512, True), (512, 1536, False)]),
                                      (1000, [(136, 200, True), (264, 72, True), (128, 128, False)]), (300, [(2048, 512, True)]),
                                      (5000, [(512, 2048, True), (2048, 512, True), (512, 512, True), (512, 1536, False)])])
@pytest.mark.parametrize("wide", [1, 0])
def test_gemm_tn_group_matches_fp32(M, shapes, wide):
    """[r06] wide = 1: the library may run the group on 128 x 256 tiles (dmi_gemm_tn_group_plan > 0: the four-problem case here and
    the FFN pair), wide = 0: always the 128 x 128 tiles.
    [r05] dmi_gemm_tn_group: several weight gradients over the same M rows in one launch (the out-projection + QKV pair of a block
    at the benchmark shape, ragged tiles, one problem, four problems, split and unsplit plans), each vs the fp32 product of the same
    bf16 operands, bias gradients as column sums; and vs dmi_gemm_tn to fp32 summation order; deferred and immediate reduces agree
    bit for bit."""
    dh.set_option("tn_wide", wide)
    try:
        _group_matches_fp32(M, shapes, wide)
    finally:
        dh.set_option("tn_wide", 1)


def _group_matches_fp32(M, shapes, wide):
    plan = dh.gemm_tn_group_plan([(I, J) for I, J, _ in shapes], M)
    assert (plan > 0) == (wide == 1 and len(shapes) == 4 and M == 5000), plan
    probs, refs = [], []
    for k, (I, J, with_bias) in enumerate(shapes):
        X, dY = rnd(M, I, seed=10 + k), rnd(M, J, seed=20 + k)
        q = dict(X=X.to(DEV), ldx=I, dY=dY.to(DEV), ldy=J, dW=torch.full((I, J), 7.0, dtype=torch.float32, device=DEV), I=I, J=J,
                 ws=torch.empty(int(dh.gemm_tn_workspace_bytes(M, I, J)) + 256, dtype=torch.uint8, device=DEV))
        if with_bias:
            q["dbias"] = torch.full((J,), 7.0, dtype=torch.float32, device=DEV)
        probs.append(q)
        refs.append((X.float().t() @ dY.float(), dY.float().sum(0)))
    dh.gemm_tn_group(probs, M)
    first = [(q["dW"].clone(), q["dbias"].clone() if "dbias" in q else None) for q in probs]
    for q, (rw, rb) in zip(probs, refs):
        close(q["dW"], rw, 2e-3, 2e-3 * math.sqrt(M), "group dW")
        if "dbias" in q:
            close(q["dbias"], rb, 2e-3, 2e-3 * math.sqrt(M), "group dbias")
        single = torch.zeros_like(q["dW"])
        sb = torch.zeros(q["J"], dtype=torch.float32, device=DEV) if "dbias" in q else None
        dh.gemm_tn(q["X"], q["ldx"], q["dY"], q["ldy"], single, M, q["I"], q["J"], q["ws"], dbias=sb)
        close(q["dW"], single.cpu(), 1e-4, 1e-4 * math.sqrt(M), "group vs single")
    for q in probs:
        q["dW"].fill_(3.0)
        if "dbias" in q:
            q["dbias"].fill_(3.0)
    dfr = dh.DeferredReduces()
    dh.gemm_tn_group(probs, M, deferred=dfr)
    dfr.run()
    for q, (w0, b0) in zip(probs, first):
        assert torch.equal(q["dW"], w0) and (b0 is None or torch.equal(q["dbias"], b0))


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (2100, 576, 384), (515, 1088, 128), (40960, 2048, 512)])
def test_relu_mask_as_bits_is_bit_identical(M, N, K):
    """[r05] dmi_gemm_nt_relu_bits / dmi_gemm_nt_mask_bits (FFN-1 forward emits one bit per output, the FFN-2 input gradient applies
    them): h equals dmi_gemm_nt(BIAS | RELU), the masked product equals dmi_gemm_nt(RELU_MASK, relu_src = h) -- bit for bit, incl.
    M / N tails, several tiles per block and fewer tiles than blocks; outputs that are exactly zero are masked."""
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.2, seed=2)
    bias = rnd(N, seed=3)
    dY, W2 = rnd(M, K, seed=4), rnd(N, K, scale=0.2, seed=5)      # the gradient product has its own operands, the same [M, N] output
    Ad, Bd, bd, dYd, W2d = A.to(DEV), Bt.to(DEV), bias.to(DEV), dY.to(DEV), W2.to(DEV)
    h = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    bits = torch.full((dh.relu_bits_bytes(M, N),), 0xAA, dtype=torch.uint8, device=DEV)
    dh.gemm_nt_relu_bits(Ad, K, Bd, K, h, N, M, N, K, bd, bits)
    h2 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    dh.gemm_nt(Ad, K, Bd, K, h2, N, M, N, K, dh.GEMM_BIAS | dh.GEMM_RELU, bias=bd)
    assert torch.equal(h, h2)
    frac = float((h > 0).float().mean())
    assert 0.3 < frac < 0.7, frac
    g1 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    dh.gemm_nt_mask_bits(dYd, K, W2d, K, g1, N, M, N, K, bits)
    g2 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    dh.gemm_nt(dYd, K, W2d, K, g2, N, M, N, K, dh.GEMM_RELU_MASK, relu_src=h)
    assert torch.equal(g1, g2)
    assert float((g1[h <= 0] != 0).float().sum()) == 0
    with pytest.raises(dh.DalleHipError):      # widths that are no multiple of 64: refused, the caller keeps the relu_src form
        dh.gemm_nt_relu_bits(Ad, K, Bd, K, h, N, M, N - 8, K, bd, bits)


@pytest.mark.parametrize("M,K,flags", [(160, 64, 0), (1000, 192, 5), (2100, 512, 5), (40960, 1536, 0), (5000, 2048, 1), (333, 256, 4)])
def test_gemm_ntr_full_row_tiles(M, K, flags):
    """[r04] full-row 160x512 tiles (forced; N = 512): k-step counts of every residue mod 3 (three-buffer ring), an M tail,
    bias / residual in the register epilogue -- bit-identical to the 128x128x64 kernel and equal to fp32 math on the same
    bf16 operands."""
    N = 512
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.2, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    kw = dict(bias=bias.to(DEV) if flags & 1 else None, residual=res.to(DEV) if flags & 4 else None)
    Ad, Bd = A.to(DEV), Bt.to(DEV)
    outs = []
    try:
        for ntr in (2, 0):
            dh.set_option("ntr", ntr)
            C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            dh.gemm_nt(Ad, K, Bd, K, C, N, M, N, K, flags, **kw)
            outs.append(C)
    finally:
        dh.set_option("ntr", 1)
    if M * N <= 4_000_000:
        ref = _gemm_ref(A, Bt, bias if flags & 1 else None, residual=res if flags & 4 else None)
        close(outs[0], ref, 1.6e-2, 2e-2 * math.sqrt(max(K, 64) / 64), "gemm_nt ntr")
    assert torch.equal(outs[0], outs[1]), "full-row and 128x128 tile kernels must be bit-identical"


@pytest.mark.parametrize("M,K,with_res", [(160, 512, True), (1000, 2048, True), (2100, 512, False), (40960, 512, True),
                                          (2100, 256, True), (1000, 1024, True)])   # (K / 32) % 3 == 2: the last k-step reloads into buffer 0
def test_gemm_nt_ln_fused_layernorm(M, K, with_res):
    """[r04] dmi_gemm_nt_ln (full-row tiles, N = 512): C is bit-identical to dmi_gemm_nt with the same bias / residual; Y and the
    row statistics equal dmi_layernorm_fwd applied to that C up to the summation order of the statistics (<= 1 bf16 ulp on Y),
    and fp32 LayerNorm math on the same rounded C."""
    N = 512
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.2, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    gam = (torch.rand(N, generator=torch.Generator().manual_seed(5)) + 0.5).to(torch.bfloat16)
    bet = rnd(N, seed=6)
    Ad, Bd, bd, rd = A.to(DEV), Bt.to(DEV), bias.to(DEV), (res.to(DEV) if with_res else None)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    mean = torch.zeros(M, dtype=torch.float32, device=DEV)
    rstd = torch.zeros(M, dtype=torch.float32, device=DEV)
    dh.gemm_nt_ln(Ad, K, Bd, K, C, N, M, N, K, gam.to(DEV), bet.to(DEV), Y, N, mean, rstd, bias=bd, residual=rd)
    C2 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    dh.set_option("ntr", 0)
    try:
        dh.gemm_nt(Ad, K, Bd, K, C2, N, M, N, K, dh.GEMM_BIAS | (dh.GEMM_RESIDUAL if with_res else 0), bias=bd, residual=rd)
    finally:
        dh.set_option("ntr", 1)
    assert torch.equal(C, C2), "the fused kernel's C must be bit-identical to the plain product"
    Y2 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    m2 = torch.zeros(M, dtype=torch.float32, device=DEV)
    r2 = torch.zeros(M, dtype=torch.float32, device=DEV)
    dh.layernorm_fwd(C2, gam.to(DEV), bet.to(DEV), Y2, m2, r2, M, N)
    close(mean, m2.cpu(), 1e-5, 1e-6, "mean vs layernorm_fwd")
    close(rstd, r2.cpu(), 1e-5, 1e-7, "rstd vs layernorm_fwd")
    dy = (Y.float() - Y2.float()).abs()
    assert float((dy > 2.0 ** -7 * Y2.float().abs() + 1e-6).float().mean()) == 0.0, "Y differs from layernorm_fwd by more than one bf16 ulp"
    assert float((Y != Y2).float().mean()) < 5e-3, float((Y != Y2).float().mean())   # only where the statistics' last bits tip a rounding
    cf = C.float().cpu()
    ref = torch.nn.functional.layer_norm(cf, (N,), gam.float(), bet.float(), 1e-5)
    close(Y, ref, 1.6e-2, 2e-2, "Y vs fp32 layer_norm")
    with pytest.raises(dh.DalleHipError):      # other widths: refused, the caller keeps the two-kernel form
        dh.gemm_nt_ln(Ad, K, Bd, K, C, N, M, 256, K, gam.to(DEV), bet.to(DEV), Y, N, mean, rstd)


def test_gemm_nt8_splitk_rowscale():
    """the head input-gradient form: long K split in 4, 256x256 tiles (auto: 4 x 64 tiles = 256 blocks), row scale in the reduce"""
    M, N, K = 2048, 2048, 4096 * 4
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.05, seed=2)
    rs = torch.rand(M, generator=torch.Generator().manual_seed(3)) + 0.5
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    w = ws(dh.gemm_nt_splitk_workspace_bytes(M, N, 4))
    dh.gemm_nt_splitk(A.to(DEV), K, Bt.to(DEV), K, C, M, N, K, 4, w, rowscale=rs.to(DEV))
    close(C, _gemm_ref(A, Bt) * rs[:, None], 1.6e-2, 2e-2 * math.sqrt(K / 64) * 0.05, "nt8 split-K")


@pytest.mark.parametrize("M,N,K,ns", [(300, 256, 1024, 2), (1024, 512, 4096, 2), (130, 128, 448, 3)])
def test_gemm_nt_splitk(M, N, K, ns):
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.1, seed=2)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    w = ws(dh.gemm_nt_splitk_workspace_bytes(M, N, ns))
    dh.gemm_nt_splitk(A.to(DEV), K, Bt.to(DEV), K, C, M, N, K, ns, w)
    close(C, _gemm_ref(A, Bt), 1.6e-2, 2e-2 * math.sqrt(K / 64) * 0.1, "gemm_nt_splitk")
    rs = torch.rand(M, generator=torch.Generator().manual_seed(3)) * 2 - 0.5
    dh.gemm_nt_splitk(A.to(DEV), K, Bt.to(DEV), K, C, M, N, K, ns, w, rowscale=rs.to(DEV))
    close(C, _gemm_ref(A, Bt) * rs[:, None], 1.6e-2, 2e-2 * math.sqrt(K / 64) * 0.1, "gemm_nt_splitk row scale")


@pytest.mark.parametrize("tail,tn8", [(1, 0), (0, 0), (1, 2), (1, 1)])
@pytest.mark.parametrize("M,I,J", [(256, 128, 128), (544, 256, 384), (4096, 512, 256), (1000, 128, 1160), (72, 136, 200),
                                   (48, 128, 128), (100, 64, 72), (10000, 512, 512), (5000, 520, 776),
                                   (200, 1024, 8320), (330, 640, 13320)])   # 520 / 525 tiles: the row-split tail launch
def test_gemm_tn(tail, tn8, M, I, J):
    """tn8 = 0: 128x128-tile kernel (with / without the row-split tail launch); 2: 256x256-tile 8-wave kernel forced
    (incl. ragged tiles and tiny M); 1: automatic choice."""
    dh.set_option("tn_tail", tail)
    dh.set_option("tn8", tn8)
    try:
        X, dY = rnd(M, I, seed=1), rnd(M, J, seed=2)
        dW = torch.full((I, J), 7.0, dtype=torch.float32, device=DEV)
        w = ws(dh.gemm_tn_workspace_bytes(M, I, J))
        db = torch.full((J,), 3.0, dtype=torch.float32, device=DEV)
        dh.gemm_tn(X.to(DEV), I, dY.to(DEV), J, dW, M, I, J, w, dbias=db)
        ref = X.float().t() @ dY.float()
        close(dW, ref, 2e-3, 2e-3 * math.sqrt(M), f"gemm_tn tail={tail}")
        close(db, dY.float().sum(0), 1e-4, 1e-3 * math.sqrt(M), f"gemm_tn fused bias grad tail={tail}")
        dW2 = torch.zeros_like(dW)
        dh.gemm_tn(X.to(DEV), I, dY.to(DEV), J, dW2, M, I, J, w)          # without the bias output
        assert torch.equal(dW2, dW), "gemm_tn must be deterministic and independent of the bias option"
        # weighted column sums (the fused-softmax head's bias gradient): dbias = w^T dY, w bf16 [M]
        wv = rnd(M, seed=3)
        db2 = torch.full((J,), 3.0, dtype=torch.float32, device=DEV)
        dW3 = torch.zeros_like(dW)
        dh.gemm_tn(X.to(DEV), I, dY.to(DEV), J, dW3, M, I, J, w, dbias=db2, bias_weights=wv.to(DEV))
        assert torch.equal(dW3, dW)
        close(db2, wv.float() @ dY.float(), 1e-4, 1e-3 * math.sqrt(M), f"gemm_tn weighted bias grad tail={tail}")
    finally:
        dh.set_option("tn_tail", 1)
        dh.set_option("tn8", 0)


@pytest.mark.parametrize("path", ["gang stream-K", "128x128 + tail", "128x128 whole", "256x256"])
def test_gemm_tn_rows_beyond_2GiB(path):
    """[r06] The weight gradient of the vocabulary projection at the BENCHMARK batch: dY = 40 960 rows x 50 816 columns of bf16 = 4.16 GB, i.e.
    every row from 21 130 on lies more than 2 GiB into the operand.  Rounds 1-5 accumulated the row step in 32-bit per-lane buffer offsets
    under a descriptor clipped to 2^31 - 1 bytes, so those rows were range-checked to ZERO and the head's weight / bias gradient came from
    52 % of the tokens -- unseen, because every oracle comparison runs at <= 2 sequences.  Here only rows past that point (and three before
    it) are non-zero in X, and the bias weights are non-zero only there: every path must reproduce their contribution exactly."""
    M, I, J = 40960, 256, 50816     # (I >= 256: the workspace then also covers the forced 256x256 plan)
    g = torch.Generator().manual_seed(5)
    hot = torch.tensor([5, 21000, 21129, 21130, 21131, 30000, 30063, 40959])
    X = torch.zeros(M, I, dtype=torch.bfloat16, device=DEV)
    X[hot.to(DEV)] = torch.randn(len(hot), I, generator=g).to(torch.bfloat16).to(DEV)
    Y = torch.empty(M, J, dtype=torch.bfloat16, device=DEV)
    for r0 in range(0, M, 4096):     # (row-dependent values, generated in pieces: 4 GB of bf16)
        Y[r0:r0 + 4096] = (torch.randn(4096, 1, generator=g) * 0.5 + torch.randn(1, J, generator=g) * 0.25).to(torch.bfloat16).to(DEV)
    wv = torch.zeros(M, dtype=torch.bfloat16, device=DEV)
    wv[hot.to(DEV)] = (torch.rand(len(hot), generator=g) + 0.5).to(torch.bfloat16).to(DEV)
    ref = X[hot.to(DEV)].float().t() @ Y[hot.to(DEV)].float()
    rb = wv[hot.to(DEV)].float() @ Y[hot.to(DEV)].float()
    w = ws(dh.gemm_tn_workspace_bytes(M, I, J))
    opts = {"gang stream-K": dict(tn_wide=1, tn_tail=1, tn8=0), "128x128 + tail": dict(tn_wide=0, tn_tail=1, tn8=0),
            "128x128 whole": dict(tn_wide=0, tn_tail=0, tn8=0), "256x256": dict(tn_wide=0, tn_tail=1, tn8=2)}[path]
    for k, v in opts.items():
        dh.set_option(k, v)
    try:
        dW = torch.full((I, J), float("nan"), device=DEV)
        db = torch.full((J,), float("nan"), device=DEV)
        dh.gemm_tn(X, I, Y, J, dW, M, I, J, w, dbias=db, bias_weights=wv)
        torch.cuda.synchronize()
    finally:
        dh.set_option("tn_wide", 1); dh.set_option("tn_tail", 1); dh.set_option("tn8", 0)
    assert float(ref.abs().max()) > 1.0
    close(dW, ref, 1e-5, 1e-4, f"gemm_tn {path}: rows beyond 2 GiB")
    close(db, rb, 1e-5, 1e-4, f"gemm_tn {path}: weighted bias sums beyond 2 GiB")


def test_head_forward_and_input_gradient_rows_beyond_2GiB():
    """[r06] The same question for the other two kernels that touch the 4.16-GB E = exp(logits) matrix at the benchmark batch -- the
    vocabulary projection that writes it and the input gradient that reads it (both NT kernels with per-tile descriptor bases, which is why
    they were right all along): rows on both sides of the 2-GiB line against fp32."""
    M, K, Vp = 40960, 512, 50816
    g = torch.Generator().manual_seed(9)
    X = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    Wt = (torch.randn(Vp, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    bias = (torch.randn(Vp, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    E = torch.empty(M, Vp, dtype=torch.bfloat16, device=DEV)
    part = torch.empty(dh.gemm_nt_softmax_partials(Vp), M, dtype=torch.float32, device=DEV)
    dh.gemm_nt_softmax(X, K, Wt, K, bias, None, E, Vp, part, M, Vp, K)
    rows = torch.tensor([0, 21129, 21130, 21131, 30000, 40959], device=DEV)
    ref = torch.exp(X[rows].float() @ Wt.float().t() + bias.float())
    close(E[rows], ref, 1.0e-2, 1e-6, "vocabulary projection rows beyond 2 GiB")
    close(part[:, rows].sum(0), ref.sum(1), 2e-3, 1e-3, "row sums beyond 2 GiB")
    # input gradient dX = rowscale * (E . W): W = [K, Vp]
    W = Wt.t().contiguous()
    rs = (torch.rand(M, generator=g) + 0.5).to(DEV)
    dX = torch.full((M, K), float("nan"), dtype=torch.bfloat16, device=DEV)
    dh.gemm_nt(E, Vp, W, Vp, dX, K, M, K, Vp, dh.GEMM_ROWSCALE, rowscale=rs)
    refd = (E[rows].float() @ W.float().t()) * rs[rows, None]
    close(dX[rows], refd, 1.6e-2, 2e-2 * float(refd.abs().max()), "head input gradient rows beyond 2 GiB")
    assert not torch.isnan(dX.float()).any()


def test_gemm_tn_group_wide_ffn_pair_is_bit_identical_to_two_launches():
    """[r06] the two FFN gradients of a block as one grouped launch on 128 x 256 tiles: 32 + 32 tiles x 8 row splits -- the split count
    of each single launch, and the wide tile keeps the k order (32-row chunks in row order), so the grouped launch has the single
    launches' BITS (slab by slab); with the 128 x 128 group (64 + 64 tiles x 4 splits) only to summation order."""
    M, d = 8192, 512
    X2, dY2, X1, dY1 = rnd(M, 4 * d, seed=1).to(DEV), rnd(M, d, seed=2).to(DEV), rnd(M, d, seed=3).to(DEV), rnd(M, 4 * d, seed=4).to(DEV)
    w2, w1 = ws(dh.gemm_tn_workspace_bytes(M, 4 * d, d)), ws(dh.gemm_tn_workspace_bytes(M, d, 4 * d))
    single = []
    for X, dY, I, J, w in ((X2, dY2, 4 * d, d, w2), (X1, dY1, d, 4 * d, w1)):
        dW = torch.zeros(I, J, dtype=torch.float32, device=DEV); db = torch.zeros(J, dtype=torch.float32, device=DEV)
        dh.gemm_tn(X, I, dY, J, dW, M, I, J, w, dbias=db)
        single.append((dW, db))
    assert dh.gemm_tn_group_plan([(4 * d, d), (d, 4 * d)], M) == 8
    for wide in (1, 0):
        dh.set_option("tn_wide", wide)
        try:
            probs = [dict(X=X2, ldx=4 * d, dY=dY2, ldy=d, dW=torch.full((4 * d, d), float("nan"), device=DEV), I=4 * d, J=d, ws=w2,
                          dbias=torch.full((d,), float("nan"), device=DEV)),
                     dict(X=X1, ldx=d, dY=dY1, ldy=4 * d, dW=torch.full((d, 4 * d), float("nan"), device=DEV), I=d, J=4 * d, ws=w1,
                          dbias=torch.full((4 * d,), float("nan"), device=DEV))]
            dh.gemm_tn_group(probs, M)
        finally:
            dh.set_option("tn_wide", 1)
        for q, (dW, db) in zip(probs, single):
            if wide:
                assert torch.equal(q["dW"], dW) and torch.equal(q["dbias"], db)
            else:
                close(q["dW"], dW, 1e-4, 1e-4 * math.sqrt(M), "128x128 group vs single")


@pytest.mark.parametrize("reserve", [0, 16])
@pytest.mark.parametrize("M,I,J,weighted", [(200, 512, 33288, True), (33, 512, 40000, False), (1000, 256, 66000, True), (75, 1024, 17000, False),
                                            (500, 520, 28000, True), (4096, 512, 50816, True)])
def test_gemm_tn_gang_stream_k(reserve, M, I, J, weighted):
    """[r06] unsplit gradients with at least as many 256-column stripes as gangs (the head's shape class) run as a gang stream-K on
    128 x 256 tiles (gemm_tn_wide_sk_kernel): against fp32, against the 128x128 kernel -- stripes no gang boundary cuts are BIT-IDENTICAL
    (same k order), a cut stripe is the fp32 sum of its two pieces (<= 2 ulp of the accumulated magnitude from the one-chain sum) --,
    run to run bit-identical (a two-addend atomic sum has no order), over NaN-filled outputs (the zeroing covers exactly the cut stripes),
    with ragged I / J / M, the weighted bias sums, and with reserved CUs (another gang count)."""
    g = torch.Generator().manual_seed(M + I + J)
    X = torch.randn(M, I, generator=g).to(torch.bfloat16).to(DEV)
    dY = torch.randn(M, J, generator=g).to(torch.bfloat16).to(DEV)
    wv = (torch.rand(M, generator=g) + 0.5).to(torch.bfloat16).to(DEV) if weighted else None
    w = ws(dh.gemm_tn_workspace_bytes(M, I, J))

    def run(wide):
        dh.set_option("tn_wide", wide)
        dW = torch.full((I, J), float("nan"), dtype=torch.float32, device=DEV)
        db = torch.full((J,), float("nan"), dtype=torch.float32, device=DEV)
        dh.gemm_tn(X, I, dY, J, dW, M, I, J, w, dbias=db, bias_weights=wv)
        torch.cuda.synchronize()
        return dW, db
    dh.set_option("reserve_cus", reserve)
    dh.set_option("tn_tail", 0)    # the reference arm: whole 128x128 tiles, no row-split tail stripe (which sums in another order)
    try:
        d0, b0 = run(0)
        d1, b1 = run(1)
        d2, b2 = run(1)
    finally:
        dh.set_option("tn_wide", 1)
        dh.set_option("tn_tail", 1)
        dh.set_option("reserve_cus", 0)
    assert not torch.isnan(d1).any() and not torch.isnan(b1).any()
    assert torch.equal(d1, d2) and torch.equal(b1, b2), "the gang stream-K must be deterministic"
    ref = X.float().t() @ dY.float()
    close(d1, ref, 2e-3, 2e-3 * math.sqrt(M), "gang stream-K vs fp32")
    close(b1, (wv.float() @ dY.float()) if weighted else dY.float().sum(0), 1e-4, 1e-3 * math.sqrt(M), "gang stream-K bias sums")
    # which stripes does a gang boundary cut?  (the library's plan: 2 blocks per CU less the reserved CUs, whole gangs per XCD)
    wti, wtj, nsteps = (I + 127) // 128, (J + 255) // 256, (M + 31) // 32
    gangs = (2 * ((256 - reserve) & ~7) // wti) & ~7
    assert wtj >= gangs, "shape does not take the gang stream-K path"
    U = wtj * nsteps
    cut = torch.zeros(wtj, dtype=torch.bool)
    for k in range(1, gangs):
        b = U * k // gangs
        if b % nsteps:
            cut[b // nsteps] = True
    assert 0 < int(cut.sum()) < wtj or nsteps == 1
    colcut = cut.repeat_interleave(256)[:J].to(DEV)
    assert torch.equal(d1[:, ~colcut], d0[:, ~colcut]), "uncut stripes must have the 128x128 kernel's bits"
    assert torch.equal(b1[~colcut], b0[~colcut])
    mag = (X.float().abs().t() @ dY.float().abs())
    assert float(((d1 - d0).abs() / (mag + 1e-6)).max()) <= 2.5e-7, float(((d1 - d0).abs() / (mag + 1e-6)).max())
    assert int((d1[:, colcut] != d0[:, colcut]).sum()) > 0 or nsteps == 1   # (the cut stripes really are two-piece sums)


def test_gemm_tn_deferred_batch_reduce_is_bit_identical():
    """four split weight gradients (with / without bias) whose slab reduces are deferred and run as ONE launch == the same
    calls with their own reduce launches, bit for bit; an unsplit problem defers nothing."""
    M = 6000
    shapes = [(256, 768, True), (256, 256, True), (1024, 256, False), (256, 1024, True)]
    deferred = dh.DeferredReduces()
    outs, refs = [], []
    for k, (I, J, wb) in enumerate(shapes):
        X, dY = rnd(M, I, seed=10 + k).to(DEV), rnd(M, J, seed=20 + k).to(DEV)
        rW = torch.zeros(I, J, dtype=torch.float32, device=DEV)
        rb = torch.zeros(J, dtype=torch.float32, device=DEV) if wb else None
        dh.gemm_tn(X, I, dY, J, rW, M, I, J, ws(dh.gemm_tn_workspace_bytes(M, I, J)), dbias=rb)
        refs.append((rW, rb))
        dW = torch.full((I, J), 5.0, dtype=torch.float32, device=DEV)
        db = torch.full((J,), 2.0, dtype=torch.float32, device=DEV) if wb else None
        own_ws = ws(dh.gemm_tn_workspace_bytes(M, I, J))
        dh.gemm_tn(X, I, dY, J, dW, M, I, J, own_ws, dbias=db, deferred=deferred)
        outs.append((dW, db, own_ws))
    assert deferred.n == 7          # 3 x (bias + weights) + 1 x weights
    deferred.run()
    assert deferred.n == 0
    for (dW, db, _), (rW, rb) in zip(outs, refs):
        assert torch.equal(dW, rW)
        if db is not None:
            assert torch.equal(db, rb)
    X, dY = rnd(500, 1024, seed=1).to(DEV), rnd(500, 8320, seed=2).to(DEV)     # 520 tiles: unsplit (tail launch) -> nothing deferred
    dW = torch.zeros(1024, 8320, dtype=torch.float32, device=DEV)
    dh.gemm_tn(X, 1024, dY, 8320, dW, 500, 1024, 8320, ws(dh.gemm_tn_workspace_bytes(500, 1024, 8320)), deferred=deferred)
    assert deferred.n == 0
    close(dW, X.float().cpu().t() @ dY.float().cpu(), 2e-3, 2e-3 * math.sqrt(500), "unsplit with deferred arg")


# ------------------------------------------------------------------ fused softmax head

def _head_case(M, K, V, seed, big=None):
    g = torch.Generator().manual_seed(seed)
    Vp = (V + 127) // 128 * 128
    X = (torch.randn(M, K, generator=g)).to(torch.bfloat16)
    Wt = torch.zeros(Vp, K)
    Wt[:V] = torch.randn(V, K, generator=g) * (2.0 / math.sqrt(K))
    Wt = Wt.to(torch.bfloat16)
    bias = torch.full((Vp,), -30000.0)
    bias[:V] = torch.randn(V, generator=g) * 0.5
    bias = bias.to(torch.bfloat16)
    labels = torch.randint(0, V, (M,), generator=g, dtype=torch.int32)
    if big is not None:   # rows whose label logit is > 88 below the row maximum: the exponent overflows -> exact fix-up path
        for r in big:
            v_hi = (int(labels[r]) + 1) % V
            Wt[v_hi] = (X[r].float() * (200.0 / float(X[r].float().pow(2).sum()))).to(torch.bfloat16)
            Wt[int(labels[r])] = (-X[r].float() * (100.0 / float(X[r].float().pow(2).sum()))).to(torch.bfloat16)
    return X, Wt, bias, labels, Vp


def _run_head(X, Wt, bias, labels, V, Vp, dz_scale, shift=True):
    M, K = X.shape
    Xd, Wd, bd, ld = X.to(DEV), Wt.to(DEV), bias.to(DEV), labels.to(DEV)
    zl = torch.empty(M, dtype=torch.float32, device=DEV)
    flag = torch.full((1,), 5, dtype=torch.int32, device=DEV)
    dh.label_logit(Xd, K, Wd, K, bd, ld, zl, flag, M, K, V)
    nparts = dh.gemm_nt_softmax_partials(Vp)
    part = torch.full((nparts, M), float("nan"), dtype=torch.float32, device=DEV)
    E = torch.zeros(M, Vp, dtype=torch.bfloat16, device=DEV)
    dh.gemm_nt_softmax(Xd, K, Wd, K, bd, zl if shift else None, E, Vp, part, M, Vp, K)
    loss = torch.empty(M, dtype=torch.float32, device=DEV)
    rsc = torch.empty(M, dtype=torch.float32, device=DEV)
    rsb = torch.empty(M, dtype=torch.bfloat16, device=DEV)
    Xs = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    dh.softmax_finish(part, nparts, zl, zl if shift else None, ld, Xd, K, Wd, K, bd, E, Vp, Vp, loss, rsc, rsb, Xs, flag, M, K, V, dz_scale)
    return zl, E, loss, rsc, rsb, Xs, int(flag.item())


@pytest.mark.parametrize("shift", [False, True])
@pytest.mark.parametrize("nt4", [0, 2, "nt8p"])
@pytest.mark.parametrize("M,K,V", [(300, 128, 1000), (1024, 512, 5000), (77, 256, 777), (257, 64, 200)])
def test_fused_softmax_head(M, K, V, nt4, shift):
    """label logit -> exp-epilogue GEMM -> finish: loss_rows = logsumexp - label logit, rowscale * E = dz_scale * (softmax -
    onehot), Xs = rowscale * X; vs fp32 math on the same bf16 inputs.  Both NT tilings; without an exponent shift (the
    engine's mode) and with the label logit as shift."""
    X, Wt, bias, labels, Vp = _head_case(M, K, V, seed=M)
    dz_scale = 1.0 / M
    if nt4 == "nt8p":
        if K % 128:
            pytest.skip("the persistent 256x256 kernel needs K % 128 == 0")
        dh.set_option("nt8p", 2)
    else:
        dh.set_option("nt4", nt4)
    try:
        zl, E, loss, rsc, rsb, Xs, flag = _run_head(X, Wt, bias, labels, V, Vp, dz_scale, shift)
    finally:
        dh.set_option("nt4", 1)
        dh.set_option("nt8p", 1)
    assert flag == 0
    z = X.float() @ Wt.float()[:V].t() + bias.float()[:V]
    lab = labels.long()
    close(zl, z[torch.arange(M), lab], 1e-4, 1e-4, "label logit")
    ref_loss = torch.logsumexp(z, -1) - z[torch.arange(M), lab]
    close(loss, ref_loss, 1e-4, 2e-4, "loss rows")
    p = torch.softmax(z, -1)
    dz_ref = (p - F.one_hot(lab, V).float()) * dz_scale
    dz = E.float().cpu()[:, :V] * rsc.cpu()[:, None]
    assert float(E.float()[:, V:].abs().max()) == 0.0, "pad columns must be exactly zero"
    # bf16 rounding of E: 2^-9 relative on each entry; entries below 1e-3 of the row's largest are noise-level
    close(dz, dz_ref, 1.6e-2, 1e-3 * dz_scale, "dlogits = rowscale * E")
    close(Xs, X.float() * rsc.cpu()[:, None], 1.6e-2, 1e-12, "Xs")
    close(rsb, rsc.cpu(), 8e-3, 0.0, "rowscale bf16")
    assert abs(float((dz.sum(-1)).abs().max())) <= 2e-2 * dz_scale   # rows of dlogits sum to ~0


@pytest.mark.parametrize("shift", [False, True])
def test_fused_softmax_head_overflow_rows_are_redone_exactly(shift):
    """rows with a logit of +200 (label logit -100) overflow exp(logit [- label logit]); they are flagged from their sum and
    recomputed with the row maximum as the shift: same loss / dlogits as fp32 math, other rows untouched."""
    M, K, V = 130, 128, 500
    X, Wt, bias, labels, Vp = _head_case(M, K, V, seed=11, big=[3, 64, 129])
    dz_scale = 0.25
    zl, E, loss, rsc, rsb, Xs, flag = _run_head(X, Wt, bias, labels, V, Vp, dz_scale, shift)
    assert flag == 1
    z = X.float() @ Wt.float()[:V].t() + bias.float()[:V]
    lab = labels.long()
    ref_loss = torch.logsumexp(z, -1) - z[torch.arange(M), lab]
    assert float(ref_loss[3]) > 88
    close(loss, ref_loss, 1e-4, 2e-3, "loss rows")
    dz_ref = (torch.softmax(z, -1) - F.one_hot(lab, V).float()) * dz_scale
    dz = E.float().cpu()[:, :V] * rsc.cpu()[:, None]
    close(dz, dz_ref, 1.6e-2, 1e-3 * dz_scale, "dlogits incl. redone rows")
    close(Xs, X.float() * rsc.cpu()[:, None], 1.6e-2, 1e-12, "Xs")


def test_fused_softmax_head_large_but_finite_rows_take_the_exact_path():
    """a row whose largest logit is ~80 has a FINITE sum (e^80 = 5.5e34) but rowscale = dz_scale / S would be a
    denormal: Xs and the bf16 rowscale flush to zero and the row silently drops out of dW / db.  Such rows must be flagged and
    redone with the row maximum as the shift, like overflowing ones."""
    M, K, V = 130, 128, 500
    X, Wt, bias, labels, Vp = _head_case(M, K, V, seed=12)
    lab = labels.long()
    for r in (5, 77):
        v_hi = (int(lab[r]) + 1) % V
        Wt[v_hi] = (X[r].float() * (80.0 / float(X[r].float().pow(2).sum()))).to(torch.bfloat16)
    dz_scale = 1.0 / (32 * 1280)
    zl, E, loss, rsc, rsb, Xs, flag = _run_head(X, Wt, bias, labels, V, Vp, dz_scale, shift=False)
    assert flag == 1
    z = X.float() @ Wt.float()[:V].t() + bias.float()[:V]
    assert 70 < float(z[5].max()) < 88
    ref_loss = torch.logsumexp(z, -1) - z[torch.arange(M), lab]
    close(loss, ref_loss, 1e-4, 2e-3, "loss rows")
    dz_ref = (torch.softmax(z, -1) - F.one_hot(lab, V).float()) * dz_scale
    dz = E.float().cpu()[:, :V] * rsc.cpu()[:, None]
    close(dz, dz_ref, 1.6e-2, 1e-3 * dz_scale, "dlogits incl. redone rows")
    assert float(rsb.float().cpu()[5]) > 0 and float(Xs.float().cpu()[5].abs().max()) > 0, "row 5 must not flush to zero"
    close(Xs, X.float() * rsc.cpu()[:, None], 1.6e-2, 1e-12, "Xs")


def test_colsum_and_transpose():
    M, N = 1000, 520
    Y = rnd(M, N, seed=1)
    out = torch.empty(N, dtype=torch.float32, device=DEV)
    dh.colsum(Y.to(DEV), N, out, M, N, ws(dh.colsum_workspace_bytes(M, N)))
    close(out, Y.float().sum(0), 1e-4, 1e-3, "colsum")
    X = rnd(3, 72, 200, seed=2)
    T = torch.zeros(3, 200, 72, dtype=torch.bfloat16, device=DEV)
    dh.transpose(X.to(DEV), T, 3, 72, 200)
    assert torch.equal(T.cpu(), X.transpose(1, 2).contiguous()), "transpose"


# ------------------------------------------------------------------ attention

def _attn_ref(qkv, B, H, S):
    d = H * 128
    t = qkv.float().view(B, S, 3, H, 128)
    q, k, v = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))  # [B,H,S,128]
    logits = q @ k.transpose(-1, -2)
    mask = torch.triu(torch.ones(S, S, dtype=torch.bool), 1)
    logits = logits.masked_fill(mask, float("-inf"))
    lse = torch.logsumexp(logits, -1)
    o = torch.softmax(logits, -1) @ v
    return o.permute(0, 2, 1, 3).reshape(B * S, d), lse


def test_transpose_batch_and_fast_sums():
    """one-launch transposes of several ragged matrices living in one flat buffer (bit-exact), and the single-block
    vectorised sum (fixed order) for aligned / unaligned / tail lengths."""
    shapes = [(512, 1536), (72, 200), (8, 8), (2048, 512), (200, 72)]
    src = rnd(sum(r * c for r, c in shapes), seed=21).to(DEV)
    dst = torch.zeros_like(src)
    rows, off, tile = [], 0, 0
    for r, c in shapes:
        rows.append([off, off, r, c, tile])
        off += r * c
        tile += ((r + 63) // 64) * ((c + 63) // 64)
    table = torch.tensor(rows, dtype=torch.int64, device=DEV)
    dh.transpose_batch(src, dst, table, len(shapes), tile)
    off = 0
    for r, c in shapes:
        want = src[off:off + r * c].view(r, c).t().contiguous().view(-1)
        assert torch.equal(dst[off:off + r * c], want), (r, c)
        off += r * c
    for n in (1, 3, 4, 1000, 40960, 40963, 2048):
        x = torch.randn(n + 1, generator=torch.Generator().manual_seed(n)).to(DEV)
        out = torch.zeros(1, device=DEV)
        for view in (x[:n], x[1:n + 1]):       # second view is only 4-byte aligned
            dh.sum_f32(view, n, 0.5, out)
            ref = float(view.double().sum()) * 0.5
            assert abs(float(out) - ref) <= 1e-5 * max(1.0, abs(ref)) + 2e-4 * (n ** 0.5) * 1e-2, (n, float(out), ref)


@pytest.mark.parametrize("xcd", [8, 1, 0, 3])
@pytest.mark.parametrize("B,H,S", [(1, 1, 128), (2, 2, 272), (1, 2, 384), (1, 1, 72), (3, 1, 384), (5, 2, 200)])
def test_attention_fwd_bwd(B, H, S, xcd):
    """persistent attention blocks; attn_xcd != 0: per-XCD item lists when the (batch, head) count divides by 8 (default),
    0 (or any other count): one serpentine over all blocks.  These grids are smaller than the chip: one item per block."""
    dh.set_option("attn_xcd", xcd)
    try:
        _attention_fwd_bwd(B, H, S)
    finally:
        dh.set_option("attn_xcd", 8)


@pytest.mark.parametrize("xcd", [8, 0])
@pytest.mark.parametrize("B,H,S", [(4, 2, 272), (8, 8, 1152)])
def test_attention_persistent_schedules(B, H, S, xcd):
    """(batch, head) counts that divide by 8 (per-XCD item lists) and, at (8, 8, 1152), 576 items: more than the 256 dK/dV
    / 512 forward and dQ blocks of the chip, so every block walks several rounds of its serpentine list."""
    dh.set_option("attn_xcd", xcd)
    try:
        _attention_fwd_bwd(B, H, S)
    finally:
        dh.set_option("attn_xcd", 8)


@pytest.mark.parametrize("B,H,S", [(1, 2, 520), (1, 1, 1280), (2, 1, 40), (1, 1, 8)])
def test_attention_bwd_ring_lengths(B, H, S):
    """dK/dV kernel: the software pipeline's head / steady-state / tail paths (1, 2, 3 and many 32-query steps per
    block, sequence ends inside a tile, masked diagonal tiles followed by mask-free ones)."""
    _attention_fwd_bwd(B, H, S)


def _attention_fwd_bwd(B, H, S):
    d = H * 128
    # q small (the reference folds 1/sqrt(k) into Wq's init), k/v O(1): logits O(1)
    g = torch.Generator().manual_seed(S)
    qkv = torch.randn(B * S, 3, H, 128, generator=g)
    qkv[:, 0] *= 0.12
    qkv = qkv.view(B * S, 3 * d).to(torch.bfloat16)
    qkv_d = qkv.to(DEV)
    o = torch.zeros(B * S, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(B, H, S, dtype=torch.float32, device=DEV)
    dh.attention_fwd(qkv_d, o, lse, B, H, S)
    qr = qkv.float().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, B, H, S)
    close(lse, lse_ref.detach(), 2e-3, 2e-3, "attn lse")
    close(o, o_ref.detach(), 1.6e-2, 1.5e-2, "attn fwd o")
    d_o = rnd(B * S, d, seed=9)
    o_ref.backward(d_o.float())
    d_o_d = d_o.to(DEV)
    delta = torch.zeros(3, B, H, S, dtype=torch.float32, device=DEV)
    dqkv = torch.zeros(B * S, 3 * d, dtype=torch.bfloat16, device=DEV)
    dh.attention_bwd(qkv_d, o, d_o_d, lse, delta, dqkv, B, H, S)
    gref = qr.grad.view(B * S, 3, d)
    got = dqkv.float().cpu().view(B * S, 3, d)
    for i, nm in enumerate("qkv"):
        scale = float(gref[:, i].abs().max())
        close(got[:, i], gref[:, i], 3e-2, 2e-2 * scale, f"attn bwd d{nm}")
    return dqkv


def test_persistent_grids_with_reserved_cus():
    """[r04] option reserve_cus (set by the engine when world_size > 1: the RCCL channels of the gradient exchange need CUs, and a
    persistent block whose CU is taken starts late): the persistent kernels run on 240 instead of 256 blocks -- same results,
    bit for bit (attention forward / backward vs fp32 autograd at a shape with many items per block; persistent 256x256 NT
    kernel; the full-row kernel steps aside)."""
    base = _attention_fwd_bwd(8, 4, 640)
    A, Bt = rnd(6000, 512, seed=1), rnd(5000, 512, scale=0.2, seed=2)
    C0 = torch.zeros(6000, 5000, dtype=torch.bfloat16, device=DEV)
    C1 = torch.zeros(6000, 5000, dtype=torch.bfloat16, device=DEV)
    dh.set_option("nt8p", 2)
    dh.gemm_nt(A.to(DEV), 512, Bt.to(DEV), 512, C0, 5000, 6000, 5000, 512, 0)
    dh.set_option("reserve_cus", 16)
    try:
        assert dh.get_option("reserve_cus") == 16
        again = _attention_fwd_bwd(8, 4, 640)
        dh.gemm_nt(A.to(DEV), 512, Bt.to(DEV), 512, C1, 5000, 6000, 5000, 512, 0)
    finally:
        dh.set_option("reserve_cus", 0)
        dh.set_option("nt8p", 1)
    assert torch.equal(base, again) and torch.equal(C0, C1)


@pytest.mark.parametrize("B,H,S", [(2, 4, 264), (1, 4, 1280), (2, 1, 72)])
def test_attention_round6_forms_vs_round2_kernels(B, H, S):
    """[r06] options attn_fwd / attn_bwd select the round-2 kernels (0) or the round-6 forms (1, default).  Backward: coalesced prologue
    fetches and whole-row epilogue stores through LDS strips move the same numbers -- bit-identical.  Forward: the software-pipelined kernel
    keeps its running maximum as an integer power of two and defers rescales, i.e. P is rounded to bf16 at another scale: outputs agree to
    one bf16 ulp of the output range, lse to fp32 rounding; both forms are checked against fp32 autograd by _attention_fwd_bwd."""
    out = {}
    for ver in (0, 1):
        dh.set_option("attn_fwd", ver)
        dh.set_option("attn_bwd", ver)
        try:
            dqkv = _attention_fwd_bwd(B, H, S)
            d = H * 128
            g = torch.Generator().manual_seed(S + 1)
            qkv = (torch.randn(B * S, 3 * d, generator=g) * 0.3).to(torch.bfloat16).to(DEV)
            o = torch.zeros(B * S, d, dtype=torch.bfloat16, device=DEV)
            lse = torch.zeros(B, H, S, dtype=torch.float32, device=DEV)
            dh.attention_fwd(qkv, o, lse, B, H, S)
            dq2 = torch.zeros(B * S, 3 * d, dtype=torch.bfloat16, device=DEV)
            if ver == 1:
                o_use, lse_use = out[0][1], out[0][2]       # the same saved forward for both backward forms
            else:
                o_use, lse_use = o, lse
            dh.attention_bwd(qkv, o_use, rnd(B * S, d, seed=3).to(DEV), lse_use, torch.zeros(3, B, H, S, dtype=torch.float32, device=DEV), dq2, B, H, S)
            out[ver] = (dqkv, o.clone(), lse.clone(), dq2)
        finally:
            dh.set_option("attn_fwd", 1)
            dh.set_option("attn_bwd", 1)
    assert torch.equal(out[0][3], out[1][3]), "backward forms must be bit-identical on the same inputs"
    do_ = (out[0][1].float() - out[1][1].float()).abs()
    assert float(do_.max()) <= 2.0 ** -7 * max(1.0, float(out[0][1].float().abs().max())), float(do_.max())
    close(out[1][2], out[0][2], 1e-5, 1e-5, "lse, round-6 vs round-2 forward")


@pytest.mark.parametrize("keyrow,qrow,mult", [(700, 900, 3.0), (5, 70, 2.0), (643, 900, 3.0), (675, 901, 1.5), (130, 140, 3.0), (1279, 1279, 3.0)])
def test_attention_fwd_late_spike(keyrow, qrow, mult):
    """[r06] one key far above everything a query has seen before (score ~ mult * |q|^2, hundreds of base-2 exponent units above the
    running maximum), in the lower and in the upper half-row lanes, in the first tile, in a steady-state tile and on the diagonal: the
    deferred rescale must fire with an exact (possibly flushed-to-zero) factor and the row maximum must cover BOTH half rows -- the first
    build took it from the lower half only (a compiler fold of the permlane32 swap builtin), invisible on ordinary scores, inf here."""
    B, H, S = 1, 1, 1280
    torch.manual_seed(5)
    qkv = torch.randn(S, 3 * 128).to(torch.bfloat16)
    qkv[keyrow, 128:256] = (qkv[qrow, :128].float() * mult).to(torch.bfloat16)
    o = torch.zeros(S, 128, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(1, 1, S, dtype=torch.float32, device=DEV)
    dh.attention_fwd(qkv.to(DEV), o, lse, B, H, S)
    o_ref, lse_ref = _attn_ref(qkv, B, H, S)
    assert torch.isfinite(lse).all() and torch.isfinite(o.float()).all()
    close(lse, lse_ref, 1e-5, 2e-3, "lse with a late spike")
    close(o, o_ref, 1.6e-2, 3e-2, "o with a late spike")


def test_attention_row0_kat():
    """causal mask: query 0 attends only to key 0 -> o[0] == v[0] exactly (bf16 round trip)."""
    B, H, S = 1, 1, 128
    qkv = rnd(S, 3 * 128, seed=11).to(DEV)
    o = torch.zeros(S, 128, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(1, 1, S, dtype=torch.float32, device=DEV)
    dh.attention_fwd(qkv, o, lse, B, H, S)
    assert torch.equal(o[0].cpu(), qkv[0, 256:].cpu())


# ------------------------------------------------------------------ cross entropy, labels, tokens

def test_shift_labels_and_assemble_tokens_bit_exact():
    from oracle import dalle_oracle as do
    B, T, P, C = 3, 16, 9, 37
    g = torch.Generator().manual_seed(3)
    text = torch.randint(0, 100, (B, T), generator=g, dtype=torch.int32)
    logits = torch.randn(B, P, C, generator=g)
    logits[0, 0, 5] = logits[0, 0, 9] = 50.0     # tie -> first index
    logits[1, 2, :] = 0.25                        # all equal -> 0
    out = torch.zeros(B, T + P, dtype=torch.int32, device=DEV)
    dh.assemble_tokens(text.to(DEV), logits.to(DEV), out, B, T, P, C, 100)
    ref = do.assemble_tokens(text.numpy(), do.image_tokens_from_logits(logits.numpy().reshape(B, 3, 3, C)), 100)
    assert np.array_equal(out.cpu().numpy(), ref)
    lab = torch.zeros_like(out)
    dh.shift_labels(out, lab, B, T + P, 999)
    assert np.array_equal(lab.cpu().numpy(), do.shift_labels(ref, 999))


@pytest.mark.parametrize("M,V,ld", [(5, 1000, 1024), (64, 777, 784), (3, 50771, 50816)])
def test_cross_entropy(M, V, ld):
    g = torch.Generator().manual_seed(V)
    z = torch.full((M, ld), -30000.0)
    z[:, :V] = torch.randn(M, V, generator=g) * 2
    z = z.to(torch.bfloat16)
    labels = torch.randint(0, V, (M,), generator=g, dtype=torch.int32)
    zd = z.to(DEV).clone()
    loss_rows = torch.zeros(M, dtype=torch.float32, device=DEV)
    lse = torch.zeros(M, dtype=torch.float32, device=DEV)
    scale = 1.0 / 1234.0
    dh.cross_entropy(zd, ld, labels.to(DEV), loss_rows, lse, M, V, scale)
    zf = z.float()[:, :V].requires_grad_(True)
    ref = F.cross_entropy(zf, labels.long(), reduction="none")
    close(loss_rows, ref.detach(), 1e-4, 1e-4, "ce loss")
    (ref.sum() * scale).backward()
    close(zd[:, :V], zf.grad, 1e-2, 1e-7, "ce dz")
    assert float(zd[:, V:].float().abs().max()) == 0.0 if ld > V else True
    tot = torch.zeros(1, dtype=torch.float32, device=DEV)
    dh.sum_f32(loss_rows, M, 1.0 / M, tot)
    assert abs(float(tot) - float(ref.mean().detach())) < 1e-4


def test_uniform_logits_loss_is_log_v():
    M, V, ld = 4, 512, 512
    z = torch.zeros(M, ld, dtype=torch.bfloat16, device=DEV)
    labels = torch.arange(M, dtype=torch.int32, device=DEV)
    loss_rows = torch.zeros(M, dtype=torch.float32, device=DEV)
    dh.cross_entropy(z, ld, labels, loss_rows, None, M, V, 0.0)
    assert torch.allclose(loss_rows.cpu(), torch.full((M,), math.log(V)), atol=1e-5)


# ------------------------------------------------------------------ optimizer

def test_sumsq_adam_cast():
    from oracle import dalle_oracle as do
    from collections import OrderedDict
    n = 100003
    g = torch.Generator().manual_seed(0)
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.01
    m, v = torch.randn(n, generator=g) * 0.01, torch.rand(n, generator=g) * 1e-4
    pad = (-n) % 4
    pd, gd, md, vd = (torch.cat([t, torch.zeros(pad)]).to(DEV) for t in (p, gr, m, v))
    out = torch.zeros(1, dtype=torch.float32, device=DEV)
    dh.sumsq(gd, n, out, ws(dh.sumsq_workspace_bytes(n)))
    assert abs(float(out) - float((gr.double() ** 2).sum())) < 1e-4 * float((gr.double() ** 2).sum())
    pb = torch.zeros(n + pad, dtype=torch.bfloat16, device=DEV)
    dh.adam_step(pd, gd, md, vd, pb, n, out, 1.0, 1e-3, 0.9, 0.999, 1e-6, 0.01, 1.0)
    P = OrderedDict(w=p.numpy().copy()); G = OrderedDict(w=gr.numpy().copy())
    Mm = OrderedDict(w=m.numpy().copy()); Vv = OrderedDict(w=v.numpy().copy())
    Gc, gn = do.clip_by_global_norm(G, 1.0)
    do.adam_step(P, Gc, Mm, Vv, 1e-3, 0.9, 0.999, 1e-6, 0.01)
    assert np.allclose(pd.cpu().numpy()[:n], P["w"], rtol=1e-5, atol=1e-6)
    assert np.allclose(md.cpu().numpy()[:n], Mm["w"], rtol=1e-5, atol=1e-7)
    assert np.allclose(vd.cpu().numpy()[:n], Vv["w"], rtol=1e-5, atol=1e-9)
    assert torch.equal(pb[:n].cpu(), pd[:n].cpu().to(torch.bfloat16))
    c = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    dh.cast_f32_bf16(pd, c, n)
    assert torch.equal(c.cpu(), pd[:n].cpu().to(torch.bfloat16))


@pytest.mark.parametrize("B,H,W,C,N,stride,kind,flags", [
    (2, 12, 12, 64, 128, 1, "3x3", 0), (1, 16, 20, 128, 72, 1, "3x3rev", 4), (3, 8, 8, 64, 64, 1, "3x3", 8),
    (2, 16, 16, 64, 136, 2, "4x4", 1), (2, 9, 7, 64, 64, 1, "par0", 1), (1, 10, 6, 128, 64, 1, "par3", 0), (2, 6, 6, 192, 64, 1, "3x3", 3)])
def test_conv_gemm_nt_equals_im2col_gemm(B, H, W, C, N, stride, kind, flags):
    """implicit-im2col convolution == dmi_im2col + dmi_gemm_nt, bit for bit (same k order and kernel arithmetic): 3x3 SAME,
    reversed taps (input gradient), 4x4 stride-2 SAME, 2x2 output-parity taps of the transposed conv; ragged M / N tiles;
    every epilogue used by the VAE."""
    taps = {"3x3": [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)],
            "3x3rev": [(1 - ky, 1 - kx) for ky in range(3) for kx in range(3)],
            "4x4": [(ky - 1, kx - 1) for ky in range(4) for kx in range(4)],
            "par0": [(0, 0), (0, -1), (-1, 0), (-1, -1)], "par3": [(1, 1), (1, 0), (0, 1), (0, 0)]}[kind]
    Ho, Wo = (H // stride, W // stride)
    K = len(taps) * C
    x = rnd(B * H * W, C, seed=1).to(DEV)
    Wt = rnd(N, K, scale=0.1, seed=2).to(DEV)
    bias, res = rnd(N, seed=3).to(DEV), rnd(B * Ho * Wo, N, seed=4).to(DEV)
    src = torch.relu(rnd(B * Ho * Wo, N, seed=5)).to(DEV)
    kw = dict(bias=bias if flags & 1 else None, residual=res if flags & 4 else None, relu_src=src if flags & 8 else None)
    col = torch.zeros(B * Ho * Wo, K, dtype=torch.bfloat16, device=DEV)
    dh.im2col(x, col, B, H, W, C, Ho, Wo, stride, taps, K)
    ref = torch.zeros(B * Ho * Wo, N, dtype=torch.bfloat16, device=DEV)
    dh.set_option("nt4", 0)
    try:
        dh.gemm_nt(col, K, Wt, K, ref, N, B * Ho * Wo, N, K, flags, **kw)
    finally:
        dh.set_option("nt4", 1)
    out = torch.zeros_like(ref)
    dh.conv_gemm_nt(x, B, H, W, C, Ho, Wo, stride, taps, Wt, K, out, N, N, flags, **kw)
    assert torch.equal(out, ref)
    assert float(out.float().abs().max()) > 0


@pytest.mark.parametrize("B,H,W,C,N,stride,kind,bias", [(2, 16, 16, 64, 128, 1, "3x3", True), (3, 8, 32, 128, 72, 1, "3x3", False),
                                                        (2, 32, 32, 64, 64, 2, "4x4", True), (1, 64, 64, 192, 136, 1, "3x3", True)])
def test_conv_wgrad_tn_equals_im2col_gemm_tn(B, H, W, C, N, stride, kind, bias):
    """implicit-im2col weight gradient == dmi_im2col + dmi_gemm_tn, bit for bit (same row split, same k order), incl. the
    fused bias gradient, 64-channel layers (a 128-wide tile spans two taps) and partial last k-tiles."""
    taps = {"3x3": [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)],
            "4x4": [(ky - 1, kx - 1) for ky in range(4) for kx in range(4)]}[kind]
    Ho, Wo = H // stride, W // stride
    K, M = len(taps) * C, B * Ho * Wo
    x = rnd(B * H * W, C, seed=1).to(DEV)
    dY = rnd(M, N, seed=2).to(DEV)
    col = torch.zeros(M, K, dtype=torch.bfloat16, device=DEV)
    dh.im2col(x, col, B, H, W, C, Ho, Wo, stride, taps, K)
    ref = torch.zeros(K, N, dtype=torch.float32, device=DEV)
    rb = torch.zeros(N, dtype=torch.float32, device=DEV) if bias else None
    w = ws(dh.gemm_tn_workspace_bytes(M, K, N))
    dh.gemm_tn(col, K, dY, N, ref, M, K, N, w, dbias=rb)
    out = torch.full((K, N), 3.0, dtype=torch.float32, device=DEV)
    ob = torch.full((N,), 3.0, dtype=torch.float32, device=DEV) if bias else None
    dh.conv_wgrad_tn(x, B, H, W, C, Ho, Wo, stride, taps, dY, N, N, out, ws(dh.conv_wgrad_tn_workspace_bytes(M, K, N)), dbias=ob)
    assert torch.equal(out, ref)
    if bias:
        assert torch.equal(ob, rb)
    assert float(out.abs().max()) > 0
