"""End-to-end parity of the DALL-E train step (fwd + bwd + clip + Adam) on the MI355X vs the CPU oracle on
identical weights/tokens.  bf16 compute vs the fp32 oracle: loss within 1e-2 relative, every gradient tensor
within 6e-2 relative L2 (SURVEY.md §8(c) 'Tolerances to state')."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_small_step_matches_oracle():
    from parity import check_report, compare_step
    check_report(compare_step(verbose=True))


def test_ref_faithful_seq_272_step():
    """S = 256 + 16 (the reference-faithful CIFAR grid, src/model_fns.py:68), ragged tiles (272 = 2*128 + 16)."""
    from parity import check_report, compare_step
    check_report(compare_step(n_embd=256, n_heads=2, n_layers=1, text_vocab=500, image_vocab=32, T=256, P=16, B=2,
                              seed=3, steps=1))


def test_export_roundtrip_and_label_kat():
    from oracle import dalle_oracle as do
    from src.dalle_mtf.engine import DalleEngine
    cfg = do.DalleConfig(128, 40, 8, 8, 8, 1, 1)
    P0 = do.init_params(cfg, seed=5, perturb=0.1)
    eng = DalleEngine(128, 1, 1, 40, 8, 8, 8, batch_size=1, hparams=dict(lr=1e-3, train_steps=10))
    eng.load_reference_params(P0)
    back = eng.export_reference()
    for k in P0:
        assert np.array_equal(back[k], P0[k]), k
    tok = torch.randint(0, 48, (1, 16), dtype=torch.int32, device="cuda")
    eng.forward(tok, need_grad=False)
    lab = eng.labels.cpu().numpy()
    assert np.array_equal(lab, do.shift_labels(tok.cpu().numpy(), cfg.eos_token_id))
    assert lab[0, -1] == cfg.total_tokens - 1


def test_eval_logits_match_oracle():
    from collections import OrderedDict
    from oracle import dalle_oracle as do
    from src.dalle_mtf.engine import DalleEngine
    cfg = do.DalleConfig(128, 100, 20, 24, 40, 2, 1)
    P0 = do.init_params(cfg, seed=9, perturb=0.05)
    eng = DalleEngine(128, 2, 1, 100, 20, 24, 40, batch_size=2, hparams=dict(lr=1e-3, train_steps=10))
    eng.load_reference_params(P0)
    tokens = do.assemble_tokens(do.synthetic_captions(2, 24, 100), do.synthetic_image_tokens(2, 40, 20), 100)
    eng.forward(torch.from_numpy(tokens).cuda(), need_grad=False)
    got = eng.logits().cpu().numpy()
    Pt = OrderedDict((k, torch.tensor(v)) for k, v in P0.items())
    _, _, ref = do.forward(Pt, tokens, cfg, bf16=False, return_logits=True)
    ref = ref.numpy()
    assert np.abs(got - ref).max() <= 3e-2 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()


def test_microbatched_step_equals_full_batch_step():
    """Serialized training step (src/model_fns.py:141-166, src/dalle_mtf/models.py:356): 4 micro-batches of 1 row
    accumulated locally == one step on the 4-row batch -- same loss, same gradients (fp32 accumulation-order
    tolerance) and therefore the same Adam update; also equals the oracle's full-batch loss."""
    from collections import OrderedDict
    from oracle import dalle_oracle as do
    from src.dalle_mtf.engine import DalleEngine
    cfg = do.DalleConfig(128, 100, 20, 24, 40, 2, 1)
    P0 = do.init_params(cfg, seed=11, perturb=0.05)
    tokens = do.assemble_tokens(do.synthetic_captions(4, 24, 100, seed=1), do.synthetic_image_tokens(4, 40, 20, seed=2), 100)
    tok = torch.from_numpy(tokens).cuda()
    hp = dict(lr=1e-3, train_steps=10, warmup_steps=0)
    full = DalleEngine(128, 2, 1, 100, 20, 24, 40, batch_size=4, hparams=dict(hp))
    full.load_reference_params(P0)
    loss_full = float(full.train_step(tok))
    g_full = full.g.clone()
    mb = DalleEngine(128, 2, 1, 100, 20, 24, 40, batch_size=1, global_batch_size=1, hparams=dict(hp, num_microbatches=4))
    mb.load_reference_params(P0)
    loss_mb = float(mb.train_step(tok))
    assert mb.global_step == 1
    assert abs(loss_mb - loss_full) <= 2e-3 * abs(loss_full), (loss_mb, loss_full)
    num = float((mb.g - g_full).norm())
    den = float(g_full.norm())
    assert num <= 2e-2 * den, (num, den)          # bf16 activations: per-row kernels see different tilings
    assert float((mb.p - full.p).abs().max()) <= 2.5e-3  # |Adam step| <= ~3.2 lr without bias correction
    Pt = OrderedDict((k, torch.tensor(v)) for k, v in P0.items())
    ref_loss, _ = do.forward(Pt, tokens, cfg, bf16=False)[:2]
    assert abs(loss_mb - float(ref_loss)) <= 1e-2 * abs(float(ref_loss))


def test_head_dgrad_tail_split_matches_single_launch(monkeypatch):
    """M = 320 row tiles of 256 of the [M, d] head input gradient = 1.25 residencies of the 256 CUs: the last 64 tiles run as a
    second launch with K (the vocabulary) split.  Same gradients as the single launch up to the fp32 summation order of the
    K parts."""
    from oracle import dalle_oracle as do
    from src.dalle_mtf.engine import DalleEngine
    B, T, P = 80, 24, 1000          # S = 1024, M = 81920 = 320 x 256; V = 8000 + 191 + 1 = 8192
    tokens = torch.from_numpy(do.assemble_tokens(do.synthetic_captions(B, T, 8000, seed=1),
                                                 do.synthetic_image_tokens(B, P, 191, seed=2), 8000)).cuda()
    grads = []
    for flag in ("0", "1"):
        monkeypatch.setenv("DALLE_DGRAD_TAIL", flag)
        eng = DalleEngine(256, 1, 2, 8000, 191, T, P, batch_size=B, hparams=dict(lr=1e-3, train_steps=10))
        eng.init_params(seed=7)
        loss = float(eng.forward(tokens, need_grad=True))
        eng.backward()
        torch.cuda.synchronize()
        grads.append((loss, eng.g.clone()))
        del eng
        torch.cuda.empty_cache()
    assert grads[0][0] == grads[1][0]
    num, den = float((grads[0][1] - grads[1][1]).norm()), float(grads[0][1].norm())
    assert 0 < den and num <= 5e-3 * den, (num, den)
    assert num > 0, "the tail-split launch was not taken (identical bits)"


def test_recompute_grad_is_bit_identical():
    """hparams['recompute_grad'] (mtf.recompute_grad around every block, src/dalle_mtf/models.py:342-343): block activations live
    in one shared set of buffers and backward() re-runs each block's forward -- same loss, same gradients, bit for bit,
    and the same parameters after one optimizer step."""
    from oracle import dalle_oracle as do
    from src.dalle_mtf.engine import DalleEngine
    tokens = torch.from_numpy(do.assemble_tokens(do.synthetic_captions(2, 16, 300, seed=1),
                                                 do.synthetic_image_tokens(2, 112, 64, seed=2), 300)).cuda()
    out = []
    for rc in (False, True):
        eng = DalleEngine(256, 3, 2, 300, 64, 16, 112, batch_size=2, hparams=dict(lr=1e-3, train_steps=10, warmup_steps=0,
                                                                                recompute_grad=rc))
        eng.init_params(seed=3)
        loss = float(eng.train_step(tokens))
        out.append((loss, eng.g.clone(), eng.p.clone()))
        if rc:
            assert eng.h[0].data_ptr() == eng.h[2].data_ptr() and eng.qkv[0].data_ptr() == eng.qkv[1].data_ptr()
        del eng
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])


def test_fused_layernorm_forms_equal_the_separate_kernels():
    """[r05] n_embd = 512: LayerNorm forward fused into the products that end in the residual stream (dmi_gemm_nt_ln) and LayerNorm
    backward fused into the input-gradient products that feed a LayerNorm (dmi_gemm_nt_lnbwd), each against the separate-kernel
    step on the same weights and tokens: same loss to fp32 rounding, every gradient tensor within 0.06 % (backward form) / 1.1 %
    (forward form: the fused kernels sum their row reductions in another order, Y may differ by one bf16 ulp, which flips a few ReLU
    bits) relative L2, the same parameters after clip + Adam up to one sign flip of a near-zero gradient; batched and immediate gain / bias reduces of the fused
    backward are bit-identical."""
    from oracle import dalle_oracle as do
    from src.dalle_mtf.engine import DalleEngine
    tokens = torch.from_numpy(do.assemble_tokens(do.synthetic_captions(2, 64, 500, seed=1),
                                                 do.synthetic_image_tokens(2, 192, 120, seed=2), 500)).cuda()
    out = {}
    for tag, hp in (("sep", dict(fuse_ln=False, fuse_lnbwd=False)), ("fwd", dict(fuse_ln=True, fuse_lnbwd=False)),
                    ("bwd", dict(fuse_ln=False, fuse_lnbwd=True)), ("bwd_now", dict(fuse_ln=False, fuse_lnbwd=True, lnbwd_batch_finish=False)),
                    ("both", dict(fuse_ln=True, fuse_lnbwd=True))):
        eng = DalleEngine(512, 3, 4, 500, 120, 64, 192, batch_size=2, hparams=dict(lr=1e-3, train_steps=10, warmup_steps=0, **hp))
        assert eng.fuse_ln == hp["fuse_ln"] and eng.fuse_lnbwd == hp["fuse_lnbwd"]
        eng.init_params(seed=3)
        loss = float(eng.train_step(tokens))
        out[tag] = (loss, eng.export_reference(eng.g), eng.p.clone())
        del eng
    assert torch.equal(out["bwd"][2], out["bwd_now"][2]) and out["bwd"][0] == out["bwd_now"][0]
    ref = out["sep"]
    # Error model instead of a pin:
    #   backward form: same formula, other summation order of the row / column reductions -> fp32 rounding of dgamma / dbeta and a dx that
    #   differs by one bf16 ulp on < 2 % of its elements: every tensor within 1e-3.  (Round 5 measured 5.9e-4 HERE -- at M = 512 rows,
    #   a ragged last tile, the epilogue then read rows past M through the scalar offset; with that fixed the two forms agree to 1.3e-7.)
    #   forward form: Y differs from dmi_layernorm_fwd's by one bf16 ulp (2^-8 relative) on a fraction p ~ 4e-3 of its elements.  Through
    #   FFN-1 (K = 512, weights ~ 0.02) that moves a pre-activation by ~ sqrt(p K) * 2^-8 * 0.02 ~ 1e-4 against a spread of
    #   sqrt(K) * 0.02 ~ 0.45, so a fraction f ~ 2 * 1e-4 * 0.4 / 0.45 / 2 ~ 1e-4 of the ReLU mask bits flips, and a flipped fraction f costs
    #   sqrt(f) ~ 1e-2 in relative L2 on everything upstream of that ReLU.  WHICH bits flip is chaotic (any change of rounding anywhere in
    #   the forward moves it: measured 0.0108 in round 5, 0.0186 with the round-6 attention forward), so the bound is 3 x the model's 1e-2.
    BOUND = dict(fwd=0.03, bwd=1e-3, both=0.03)
    res = {}
    for tag in ("fwd", "bwd", "both"):
        loss, g, p = out[tag]
        worst = max((float(np.linalg.norm(g[k] - ref[1][k]) / (np.linalg.norm(ref[1][k]) + 1e-30)), k) for k in g)
        res[tag] = (abs(loss - ref[0]) / abs(ref[0]), worst, float((p - ref[2]).abs().max()))
        print(tag, "loss rel", res[tag][0], "worst gradient tensor vs the separate kernels:", worst, "max parameter difference", res[tag][2], flush=True)
    for tag, (dl, worst, dp) in res.items():
        assert dl <= 2e-5, (tag, dl)
        assert worst[0] <= BOUND[tag], (tag, worst)
        assert dp <= 6.5e-3, (tag, dp)      # Adam without bias correction: |step| = 3.16 lr at step 0, twice that where a near-zero gradient's sign flips
