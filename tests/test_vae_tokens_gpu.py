"""The VAE as DALL-E's tokenizer: image tokens = argmax of the encoder logits (bit-exact integer
path downstream).  The reference runs that encoder in fp32 (src/model_fns.py:43-51 does not forward use_bf16), so the
product's tokenising path is the exact-fp32 convolution kernel; the bf16 training encoder's disagreement with the fp32 oracle
is COUNTED and reported ("argmax of bf16 VAE logits may differ from fp32 only where top-2 gap < bf16 ulp --
count and report").  Also: stack_factor > 1 (space_to_depth / depth_to_space), recompute_grad, and the stride-2 implicit
convolution directly against the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import dalle_hip as dh  # noqa: E402
from oracle import vae_oracle as vo  # noqa: E402

DEV = "cuda"


def _tf_same_conv(x_nhwc, k_hwio, bias, stride):
    return vo.conv2d_same(torch.as_tensor(x_nhwc), torch.as_tensor(k_hwio), None if bias is None else torch.as_tensor(bias), stride).numpy()


@pytest.mark.parametrize("B,H,C,N,k,s", [(2, 16, 8, 64, 4, 2), (3, 8, 64, 128, 3, 1), (1, 32, 16, 36, 4, 2), (2, 5, 24, 72, 3, 1)])
def test_conv2d_f32_matches_torch_fp32(B, H, C, N, k, s):
    """exact-fp32 MFMA convolution vs F.conv2d with TF SAME padding: fp32 round-off only (1e-5 relative)."""
    g = torch.Generator().manual_seed(H * C)
    x = torch.randn(B, H, H, C, generator=g)
    w = torch.randn(k, k, C, N, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    res = torch.randn(B * (-(-H // s)) ** 2, N, generator=g)
    Ho = -(-H // s)
    taps = [(ky - 1, kx - 1) for ky in range(k) for kx in range(k)]
    out = torch.empty(B * Ho * Ho, N, device=DEV)
    for relu, use_res in ((False, False), (True, False), (False, True)):
        dh.conv2d_f32(x.to(DEV).contiguous(), B, H, H, C, Ho, Ho, s, taps, w.reshape(k * k * C, N).to(DEV).contiguous(), b.to(DEV),
                      res.to(DEV) if use_res else None, out, N, relu=relu)
        ref = torch.from_numpy(_tf_same_conv(x.numpy(), w.numpy(), b.numpy(), s)).reshape(-1, N)
        if relu:
            ref = torch.relu(ref)
        if use_res:
            ref = ref + res
        err = float((out.cpu() - ref).abs().max())
        assert err <= 2e-5 * max(1.0, float(ref.abs().max())), (relu, use_res, err)


@pytest.mark.parametrize("s,C,Cp", [(2, 3, 16), (4, 3, 48), (2, 5, 24), (1, 3, 8)])
def test_space_depth_kernels_match_tf_semantics(s, C, Cp):
    B, Hs = 2, 6
    g = torch.Generator().manual_seed(s)
    img = torch.randn(B, Hs * s, Hs * s, C, generator=g)
    st = torch.full((B * Hs * Hs, Cp), 9.0, device=DEV)
    dh.space_to_depth_f32(img.to(DEV).contiguous(), st, B, Hs, Hs, C, s, Cp)
    ref = vo.space_to_depth(img, s).reshape(B * Hs * Hs, s * s * C)
    assert torch.equal(st.cpu()[:, :s * s * C], ref) and float(st[:, s * s * C:].abs().max() if Cp > s * s * C else 0) == 0
    back = torch.empty(B, Hs * s, Hs * s, C, device=DEV)
    dh.depth_to_space_f32(st, back, B, Hs, Hs, C, s, Cp)
    assert torch.equal(back.cpu(), img)


def test_vae_example_token_mismatch_count_bf16_vs_fp32_encoder():
    """configs/vae_example.json's stack ([[3,64],[3,128],[3,256]], 32x32, 512 tokens), B = 32 synthetic images:
    tokens from (a) the fp32 oracle, (b) the fp32 HIP tokenising encoder, (c) the bf16 HIP training encoder.
    (b) must equal (a) except where the oracle's top-2 logit gap is at fp32 round-off level; (c)'s mismatches are
    counted and printed (they sit where the top-2 gap is below the bf16 error of the logits)."""
    from parity import save_report
    from src.vae_tf import DiscreteVAE
    c = dict(num_tokens=512, dimensions=32, convblocks=[[3, 64], [3, 128], [3, 256]])
    cfg = vo.VaeConfig(**c)
    P = vo.init_params(cfg, seed=3, bias_perturb=0.02)
    B = 32
    img = vo.synthetic_images(B, 32, seed=0)
    ref = vo.encoder({k: torch.tensor(v) for k, v in P.items()}, torch.tensor(img), cfg).numpy()        # [B,4,4,512]
    tok_ref = np.argmax(ref, -1).reshape(B, -1)
    srt = np.sort(ref, -1)
    gap = (srt[..., -1] - srt[..., -2]).reshape(B, -1)
    vae = DiscreteVAE(batch_size=B, mode="eval", **c)
    vae.load_reference_params(P)
    imgd = torch.from_numpy(img).to(DEV)
    lb = vae.forward(imgd, return_logits=True).cpu().numpy().copy()
    vae.fp32_tokens = True
    lf = vae.forward(imgd, return_logits=True).cpu().numpy().copy()
    tok_b, tok_f = np.argmax(lb, -1).reshape(B, -1), np.argmax(lf, -1).reshape(B, -1)
    err_b, err_f = float(np.abs(lb - ref).max()), float(np.abs(lf - ref).max())
    mis_b, mis_f = int((tok_b != tok_ref).sum()), int((tok_f != tok_ref).sum())
    rep = dict(positions=int(tok_ref.size), logits_absmax=float(np.abs(ref).max()), bf16_logit_max_err=err_b, fp32_logit_max_err=err_f,
               bf16_token_mismatches=mis_b, fp32_token_mismatches=mis_f,
               bf16_mismatch_max_top2_gap=float(gap[tok_b != tok_ref].max()) if mis_b else 0.0,
               oracle_min_top2_gap=float(gap.min()))
    print("[vae tokens]", rep)
    save_report("parity_vae_tokens.json", rep)
    assert err_f <= 1e-4 * max(1.0, float(np.abs(ref).max())), err_f
    assert np.array_equal(tok_f[gap > 4 * err_f], tok_ref[gap > 4 * err_f])
    assert mis_f <= int((gap <= 4 * err_f).sum())
    # bf16 encoder: every mismatch must sit below the logits' bf16 error bound
    assert np.array_equal(tok_b[gap > 2 * err_b], tok_ref[gap > 2 * err_b])
    # and the downstream integer path is bit-exact on the fp32 tokens
    T, Ptok = 16, 16
    text = torch.randint(0, 50257, (B, T), dtype=torch.int32, device=DEV)
    out = torch.empty(B, T + Ptok, dtype=torch.int32, device=DEV)
    dh.assemble_tokens(text, torch.from_numpy(lf).to(DEV).contiguous(), out, B, T, Ptok, 512, 50258)
    from oracle import dalle_oracle as do
    want = do.assemble_tokens(text.cpu().numpy(), do.image_tokens_from_logits(lf), 50258)
    assert np.array_equal(out.cpu().numpy(), want)


def test_stride2_implicit_conv_vs_oracle():
    """4x4 stride-2 SAME convolution with C_in = 64 (the implicit-im2col NT kernel, no column matrix) directly against the
    oracle's F.conv2d restatement, plus its weight gradient (implicit TN kernel) against autograd."""
    import math
    B, H, C, N = 4, 16, 64, 128
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, H, C, generator=g).to(torch.bfloat16)
    w = (torch.randn(4, 4, C, N, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    taps = [(ky - 1, kx - 1) for ky in range(4) for kx in range(4)]
    Ho = H // 2
    wf = w.reshape(16 * C, N).t().contiguous()           # [co][(k,ci)]
    out = torch.empty(B * Ho * Ho, N, dtype=torch.bfloat16, device=DEV)
    dh.conv_gemm_nt(x.to(DEV).contiguous().view(-1, C), B, H, H, C, Ho, Ho, 2, taps, wf.to(DEV), 16 * C, out, N, N, dh.GEMM_BIAS,
                    bias=b.to(DEV))
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    ref = vo.conv2d_same(xr, wr, b.float(), 2)
    err = float((out.float().cpu().view(B, Ho, Ho, N) - ref.detach()).abs().max())
    assert err <= 1.6e-2 * float(ref.abs().max()) + 1e-2, err
    dy = torch.randn(B * Ho * Ho, N, generator=g).to(torch.bfloat16)
    ref.backward(dy.float().view(B, Ho, Ho, N))
    dW = torch.empty(16 * C, N, dtype=torch.float32, device=DEV)
    db = torch.empty(N, dtype=torch.float32, device=DEV)
    ws = torch.empty(dh.conv_wgrad_tn_workspace_bytes(B * Ho * Ho, 16 * C, N) + 256, dtype=torch.uint8, device=DEV)
    dh.conv_wgrad_tn(x.to(DEV).contiguous().view(-1, C), B, H, H, C, Ho, Ho, 2, taps, dy.to(DEV), N, N, dW, ws, dbias=db)
    gw = wr.grad.reshape(16 * C, N)
    assert float((dW.cpu() - gw).abs().max()) <= 2e-3 * math.sqrt(B * Ho * Ho)
    assert float((db.cpu() - dy.float().sum(0)).abs().max()) <= 1e-3 * math.sqrt(B * Ho * Ho)


@pytest.mark.parametrize("stack,recompute", [(2, False), (1, True), (2, True)])
def test_vae_stack_factor_and_recompute_vs_oracle(stack, recompute):
    """stack_factor = 2: space_to_depth in front of the encoder, depth_to_space behind the decoder (reference
    vae_tf/models.py:85-86,158-161); recompute_grad: residual branches re-run in backward -- gradients bit-identical to the
    stored-activation run, loss / reconstruction / gradients vs the fp32 oracle as in the golden test."""
    from src.vae_tf import DiscreteVAE
    c = dict(num_tokens=64, dimensions=32, convblocks=[[2, 64], [2, 64]], stack_factor=stack)
    cfg = vo.VaeConfig(**c)
    P = vo.init_params(cfg, seed=5, bias_perturb=0.02)
    B = 2
    img = vo.synthetic_images(B, 32, seed=1)
    u = vo.synthetic_uniforms((B, cfg.grid, cfg.grid, 64), seed=2)
    loss_o, g_o, out_o = vo.loss_and_grads(P, img, u, cfg, hard=False, temp=0.8)
    vae = DiscreteVAE(batch_size=B, recompute_grad=recompute, **c)
    vae.load_reference_params(P)
    back = vae.export_reference()
    for k in P:
        assert back[k].shape == P[k].shape and np.allclose(back[k], P[k]), k
    imgd = torch.from_numpy(img).to(DEV)
    loss, recon = vae.forward(imgd, return_recon_loss=True, hard_gumbel=False, temperature=0.8, noise=torch.from_numpy(u), need_grad=True)
    assert abs(float(loss) - loss_o) <= 3e-2 * loss_o, (float(loss), loss_o)
    assert recon.shape == (B, 32, 32, 3)
    assert float(np.abs(recon.cpu().numpy() - out_o).max()) <= 5e-2 * max(1.0, float(np.abs(out_o).max()))
    vae.backward()
    gh = vae.export_reference(vae.g)
    for k in g_o:
        rel = np.linalg.norm(gh[k].astype(np.float64) - g_o[k]) / max(np.linalg.norm(g_o[k]), 1e-30)
        assert rel <= 0.1, (k, rel)
    if recompute:
        ref = DiscreteVAE(batch_size=B, recompute_grad=False, **c)
        ref.load_reference_params(P)
        ref.forward(imgd, return_recon_loss=True, hard_gumbel=False, temperature=0.8, noise=torch.from_numpy(u), need_grad=True)
        ref.backward()
        assert torch.equal(ref.g, vae.g), "recompute_grad must not change a single bit of the gradients"
