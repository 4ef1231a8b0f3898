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
    dwte = torch.zeros(V, d, dtype=torch.float32, device=DEV)
    dwpe = torch.empty(S, d, dtype=torch.float32, device=DEV)
    dh.embed_bwd(tok.to(DEV), dx.to(DEV), dwte, dwpe, B, S, d, V)
    ref_wte = torch.zeros(V, d).index_add_(0, tok.view(-1).long(), dx.float())
    close(dwte, ref_wte, 1e-5, 1e-5, "embed_bwd wte")
    close(dwpe, dx.float().view(B, S, d).sum(0), 1e-5, 1e-5, "embed_bwd wpe")


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


# ------------------------------------------------------------------ GEMMs

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


VARIANTS = {"nt2": dict(nt2=1, glds=1), "glds": dict(nt2=0, glds=1), "regstage": dict(nt2=0, glds=0)}


def _set_variant(v):
    for k, val in VARIANTS[v].items():
        dh.set_option(k, val)


@pytest.mark.parametrize("glds", list(VARIANTS))
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 128), (1024, 512, 512), (257, 1160, 192), (130, 128, 448)])
def test_gemm_nt_plain(glds, M, N, K):
    _set_variant(glds)
    try:
        A, Bt = rnd(M, K, seed=1), rnd(N, K, seed=2)
        C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        dh.gemm_nt(A.to(DEV), K, Bt.to(DEV), K, C, N, M, N, K)
        close(C, _gemm_ref(A, Bt), 1.6e-2, 2e-2 * math.sqrt(K / 64), f"gemm_nt glds={glds}")
    finally:
        _set_variant("nt2")


@pytest.mark.parametrize("glds", list(VARIANTS))
def test_gemm_nt_epilogues(glds):
    _set_variant(glds)
    try:
        M, N, K = 384, 640, 256
        A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.1, seed=2)
        bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
        hsrc = torch.relu(rnd(M, N, seed=5))
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
        Cf = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        dh.gemm_nt(Ad, K, Bd, K, Cf, N, M, N, K, dh.GEMM_OUT_F32)
        close(Cf, _gemm_ref(A, Bt), 1e-3, 1e-3, "f32 out")
        # strided operands (lda > K): the QKV-style slices
        A2 = rnd(M, 3 * K, seed=7)
        dh.gemm_nt(A2.to(DEV)[:, K:], 3 * K, Bd, K, C, N, M, N, K)
        close(C, _gemm_ref(A2[:, K:2 * K], Bt), what="strided A", **tol)
    finally:
        _set_variant("nt2")


@pytest.mark.parametrize("M,N,K,flags", [(300, 256, 128, 0), (1000, 1160, 192, 1), (4000, 2568, 128, 5), (2048, 1024, 512, 3), (515, 136, 64, 8)])
def test_gemm_nt4_tile_256(M, N, K, flags):
    """256x128x32-tile kernel (forced), incl. M/N tails: same results as the 128x128x64 kernel (identical k order)."""
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.2, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    hsrc = torch.relu(rnd(M, N, seed=5))
    kw = dict(bias=bias.to(DEV) if flags & 1 else None, residual=res.to(DEV) if flags & 4 else None,
              relu_src=hsrc.to(DEV) if flags & 8 else None)
    ref = _gemm_ref(A, Bt, bias if flags & 1 else None, relu=bool(flags & 2), residual=res if flags & 4 else None,
                    relu_src=hsrc if flags & 8 else None)
    outs = []
    for nt4 in (2, 0):
        dh.set_option("nt4", nt4)
        C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        dh.gemm_nt(A.to(DEV), K, Bt.to(DEV), K, C, N, M, N, K, flags, **kw)
        close(C, ref, 1.6e-2, 2e-2 * math.sqrt(K / 64), f"gemm_nt nt4={nt4}")
        outs.append(C.cpu())
    dh.set_option("nt4", 1)
    assert torch.equal(outs[0], outs[1]), "256-row-tile and 128-row-tile kernels must be bit-identical"


@pytest.mark.parametrize("M,N,K,flags", [(300, 256, 64, 0), (1000, 1160, 192, 1), (4000, 2568, 128, 5), (2048, 1024, 512, 3),
                                         (515, 136, 64, 8), (700, 384, 320, 0), (260, 128, 1024, 4), (1024, 512, 2112, 0)])
@pytest.mark.parametrize("mode", [2, 3])
def test_gemm_nt5_hand_scheduled(M, N, K, flags, mode):
    """3-stage-ring kernel with asm-scheduled fragment reads (forced; mode 2 = 256x128 tiles, 3 = 128x128), K-step counts
    2..66 (all residues mod 3), M/N tails and every epilogue: bit-identical to the 128x128x64 kernel (same k order)."""
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.2, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    hsrc = torch.relu(rnd(M, N, seed=5))
    kw = dict(bias=bias.to(DEV) if flags & 1 else None, residual=res.to(DEV) if flags & 4 else None,
              relu_src=hsrc.to(DEV) if flags & 8 else None)
    ref = _gemm_ref(A, Bt, bias if flags & 1 else None, relu=bool(flags & 2), residual=res if flags & 4 else None,
                    relu_src=hsrc if flags & 8 else None)
    outs = []
    saved = dh.get_option("nt5")
    dh.set_option("nt4", 0)
    try:
        for nt5 in (mode, 0):
            dh.set_option("nt5", nt5)
            C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            dh.gemm_nt(A.to(DEV), K, Bt.to(DEV), K, C, N, M, N, K, flags, **kw)
            close(C, ref, 1.6e-2, 2e-2 * math.sqrt(K / 64), f"gemm_nt nt5={nt5}")
            outs.append(C.cpu())
    finally:
        dh.set_option("nt5", saved)
        dh.set_option("nt4", 1)
    assert torch.equal(outs[0], outs[1]), "hand-scheduled and compiler-scheduled kernels must be bit-identical"


@pytest.mark.parametrize("M,N,K,flags", [(4000, 2568, 128, 0), (8192, 1280, 256, 5), (3000, 3000 // 8 * 8, 384, 3)])
def test_gemm_nt_persistent(M, N, K, flags):
    """> 512 tiles and an even number of K-steps: the persistent kernel (nt3) path, incl. M/N tails; must agree with nt2."""
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.2, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    kw = dict(bias=bias.to(DEV) if flags & 1 else None, residual=res.to(DEV) if flags & 4 else None)
    ref = _gemm_ref(A, Bt, bias if flags & 1 else None, relu=bool(flags & 2), residual=res if flags & 4 else None)
    outs = []
    dh.set_option("nt4", 0)
    for nt3 in (1, 0):
        dh.set_option("nt3", nt3)
        C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        dh.gemm_nt(A.to(DEV), K, Bt.to(DEV), K, C, N, M, N, K, flags, **kw)
        close(C, ref, 1.6e-2, 2e-2 * math.sqrt(K / 64), f"gemm_nt nt3={nt3}")
        outs.append(C.cpu())
    dh.set_option("nt3", 0)
    dh.set_option("nt4", 1)
    assert torch.equal(outs[0], outs[1]), "persistent and per-tile kernels must be bit-identical"


@pytest.mark.parametrize("M,N,K,ns", [(300, 256, 1024, 2), (1024, 512, 4096, 2), (130, 128, 448, 3)])
def test_gemm_nt_splitk(M, N, K, ns):
    A, Bt = rnd(M, K, seed=1), rnd(N, K, scale=0.1, seed=2)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    w = ws(dh.gemm_nt_splitk_workspace_bytes(M, N, ns))
    dh.gemm_nt_splitk(A.to(DEV), K, Bt.to(DEV), K, C, M, N, K, ns, w)
    close(C, _gemm_ref(A, Bt), 1.6e-2, 2e-2 * math.sqrt(K / 64) * 0.1, "gemm_nt_splitk")


@pytest.mark.parametrize("trread", [1, 0])
@pytest.mark.parametrize("M,I,J", [(256, 128, 128), (544, 256, 384), (4096, 512, 256), (1000, 128, 1160), (72, 136, 200),
                                   (48, 128, 128), (100, 64, 72), (10000, 512, 512),
                                   (200, 1024, 8320), (330, 640, 13320)])   # 520 / 525 tiles: the stream-K launch
def test_gemm_tn(trread, M, I, J):
    """trread 1: hardware-transpose-read kernel; 0: explicit transposes + NT fallback."""
    dh.set_option("tn_trread", trread)
    try:
        X, dY = rnd(M, I, seed=1), rnd(M, J, seed=2)
        dW = torch.full((I, J), 7.0, dtype=torch.float32, device=DEV)
        w = ws(dh.gemm_tn_workspace_bytes(M, I, J))
        db = torch.full((J,), 3.0, dtype=torch.float32, device=DEV)
        dh.gemm_tn(X.to(DEV), I, dY.to(DEV), J, dW, M, I, J, w, dbias=db)
        ref = X.float().t() @ dY.float()
        close(dW, ref, 2e-3, 2e-3 * math.sqrt(M), f"gemm_tn trread={trread}")
        close(db, dY.float().sum(0), 1e-4, 1e-3 * math.sqrt(M), f"gemm_tn fused bias grad trread={trread}")
        dW2 = torch.zeros_like(dW)
        dh.gemm_tn(X.to(DEV), I, dY.to(DEV), J, dW2, M, I, J, w)          # without the bias output
        assert torch.equal(dW2, dW), "gemm_tn must be deterministic and independent of the bias option"
    finally:
        dh.set_option("tn_trread", 1)


def test_gemm_tn_grouped_is_bit_identical():
    """four weight gradients of different shapes (split / unsplit, with and without bias) in one grouped launch ==
    four dmi_gemm_tn calls, bit for bit; repeated launches reuse the same host problem array."""
    M = 2000
    shapes = [(256, 768, True), (256, 256, True), (1024, 256, False), (256, 1024, True), (64, 72, False)]
    items, refs = [], []
    for k, (I, J, wb) in enumerate(shapes):
        X, dY = rnd(M, I, seed=10 + k).to(DEV), rnd(M, J, seed=20 + k).to(DEV)
        dW = torch.full((I, J), 5.0, dtype=torch.float32, device=DEV)
        db = torch.full((J,), 2.0, dtype=torch.float32, device=DEV) if wb else None
        items.append((X, I, dY, J, dW, M, I, J, db))
        rW, rb = torch.zeros_like(dW), (torch.zeros_like(db) if wb else None)
        dh.gemm_tn(X, I, dY, J, rW, M, I, J, ws(dh.gemm_tn_workspace_bytes(M, I, J)), dbias=rb)
        refs.append((rW, rb))
    probs = dh.tn_problems(items)
    w = ws(dh.gemm_tn_grouped_workspace_bytes(probs))
    for _ in range(2):
        dh.gemm_tn_grouped(probs, w)
        for (X, I, dY, J, dW, M_, I_, J_, db), (rW, rb) in zip(items, refs):
            assert torch.equal(dW, rW), (I, J)
            if db is not None:
                assert torch.equal(db, rb), (I, J)
            close(dW, X.float().t() @ dY.float(), 2e-3, 2e-3 * math.sqrt(M), "grouped tn")


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
    # strided per-head transpose: qkv [B*S, 3d] -> vt [B,H,128,S]
    B, S, H = 2, 48, 3
    d = H * 128
    qkv = rnd(B * S, 3 * d, seed=3).to(DEV)
    vt = torch.zeros(B, H, 128, S, dtype=torch.bfloat16, device=DEV)
    dh.transpose_strided(qkv.data_ptr() + 2 * d * 2, vt, B, H, S, 128, S * 3 * d, 128, 3 * d)
    ref = qkv.cpu().view(B, S, 3, H, 128)[:, :, 2].permute(0, 2, 3, 1).contiguous()
    assert torch.equal(vt.cpu(), ref), "strided transpose"


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


def _transposes(qkv, B, H, S):
    d = H * 128
    outs = []
    for i in range(3):
        t = torch.zeros(B, H, 128, S, dtype=torch.bfloat16, device=DEV)
        dh.transpose_strided(qkv.data_ptr() + i * d * 2, t, B, H, S, 128, S * 3 * d, 128, 3 * d)
        outs.append(t)
    return outs


@pytest.mark.parametrize("xcd", [8, 1, 0, 3])
@pytest.mark.parametrize("B,H,S", [(1, 1, 128), (2, 2, 272), (1, 2, 384), (1, 1, 72), (3, 1, 384), (5, 2, 200)])
def test_attention_fwd_bwd(B, H, S, xcd):
    """xcd = G >= 1: per-XCD block ranges, groups of G (batch, head) pairs visited heaviest-tile-first (default 8);
    0: plain grid order.  Grids of 9 and 20 blocks exercise uneven per-XCD ranges and partial groups."""
    dh.set_option("attn_xcd", xcd)
    try:
        _attention_fwd_bwd(B, H, S)
    finally:
        dh.set_option("attn_xcd", 8)


def _attention_fwd_bwd(B, H, S):
    d = H * 128
    # q small (the reference folds 1/sqrt(k) into Wq's init), k/v O(1): logits O(1)
    g = torch.Generator().manual_seed(S)
    qkv = torch.randn(B * S, 3, H, 128, generator=g)
    qkv[:, 0] *= 0.12
    qkv = qkv.view(B * S, 3 * d).to(torch.bfloat16)
    qkv_d = qkv.to(DEV)
    qt, kt, vt = _transposes(qkv_d, B, H, S)
    o = torch.zeros(B * S, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(B, H, S, dtype=torch.float32, device=DEV)
    dh.attention_fwd(qkv_d, vt, o, lse, B, H, S)
    qr = qkv.float().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, B, H, S)
    close(lse, lse_ref.detach(), 2e-3, 2e-3, "attn lse")
    close(o, o_ref.detach(), 1.6e-2, 1.5e-2, "attn fwd o")
    d_o = rnd(B * S, d, seed=9)
    o_ref.backward(d_o.float())
    d_o_d = d_o.to(DEV)
    dot = torch.zeros(B, H, 128, S, dtype=torch.bfloat16, device=DEV)
    dh.transpose_strided(d_o_d.data_ptr(), dot, B, H, S, 128, S * d, 128, d)
    delta = torch.zeros(3, B, H, S, dtype=torch.float32, device=DEV)
    dqkv = torch.zeros(B * S, 3 * d, dtype=torch.bfloat16, device=DEV)
    dh.attention_bwd(qkv_d, qt, kt, o, d_o_d, dot, lse, delta, dqkv, B, H, S)
    gref = qr.grad.view(B * S, 3, d)
    got = dqkv.float().cpu().view(B * S, 3, d)
    for i, nm in enumerate("qkv"):
        scale = float(gref[:, i].abs().max())
        close(got[:, i], gref[:, i], 3e-2, 2e-2 * scale, f"attn bwd d{nm}")


def test_attention_row0_kat():
    """causal mask: query 0 attends only to key 0 -> o[0] == v[0] exactly (bf16 round trip)."""
    B, H, S = 1, 1, 128
    qkv = rnd(S, 3 * 128, seed=11).to(DEV)
    _, _, vt = _transposes(qkv, B, H, S)
    o = torch.zeros(S, 128, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(1, 1, S, dtype=torch.float32, device=DEV)
    dh.attention_fwd(qkv, vt, o, lse, B, H, S)
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
    assert abs(float(tot) - float(ref.mean())) < 1e-4


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
