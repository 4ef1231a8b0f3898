"""Parity at the shapes BASELINE.json names: the HIP engine vs the fp32 CPU oracle at the exact
`dalle_example` configuration (n_embd 512, 4 heads, 6 layers, 256 + 1024 positions, V = 50 771) and at one layer of the
1.3B shape (n_embd 2048, 16 heads), plus the attention kernels alone at (B, H, S) = (1, 4, 1280) and (1, 16, 1280).
Each test prints the per-tensor relative-L2 table and stores it under gpurun_out/ (committed copies: profiles/r02_parity_*).
Tolerances = measured on MI355X + 25 %."""
import math
from collections import OrderedDict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16_ORACLE_GRAD_TOL = 0.034     # measured 0.0272 (profiles/r03_parity_dalle_example.json) + 25 %
# [r04] against the TEACHER-FORCED bf16 oracle (forward = the engine's own stored activations, backward = the oracle's
# arithmetic, oracle/dalle_oracle.py _force): provisional bound, replaced by measured + 25 % once
# profiles/r04_parity_dalle_example.json exists
FORCED_ORACLE_GRAD_TOL = 0.022   # measured 0.0176 (layer_5/attn/k; every other tensor <= 0.0067) + 25 %
FORCED_FA_ORACLE_GRAD_TOL = 0.0085   # measured 0.0066 with the flash-style delta in the oracle (every tensor) + 25 %
DALLE_EXAMPLE = dict(n_embd=512, n_heads=4, n_layers=6, text_vocab=50258, image_vocab=512, T=256, P=1024)


def test_dalle_example_shape_step_vs_fp32_oracle():
    """loss, every gradient tensor and one clip + Adam update at the headline shape (B = 2: 2560 rows -> the 256x128 NT
    tiling of the vocabulary projection, the row-split weight-gradient tail and the fused softmax head all engage)."""
    from parity import check_report, compare_step, save_report
    rep = compare_step(B=2, seed=21, steps=1, perturb=0.02, bf16_oracle=True, per_tensor=True, bf16_grad_oracle=True, **DALLE_EXAMPLE)
    save_report("parity_dalle_example.json", rep)
    # measured on MI355X (profiles/r02_parity_dalle_example.json): loss 9e-6 relative, grad norm 1.4e-4, worst tensor 0.033
    # (layer_5/mlp/mlp_linear_1/kernel).  The 3 % floor is the ReLU: a pre-activation within bf16 noise of 0 flips its mask
    # bit against the fp32 oracle, and a flipped fraction f of the mask costs sqrt(f) in relative L2 upstream of it -- the
    # tensors downstream of the last ReLU (mlp_linear_2, to_logits/*) sit at 0.3-0.6 %.
    check_report(rep, loss_rtol=2e-4, grad_tol=0.042, gn_rtol=2e-3)
    assert rep["steps"][0]["head_fixup_flag"] == 0
    # [r03] the same gradients against the bf16-EMULATING oracle (dalle_oracle `bf16=True`: activations and weight copies
    # rounded to bf16 wherever the reference's bf_16 policy rounds them, src/dalle_mtf/ops.py:76-82; fp32 math in between).
    # Its forward agrees with the kernels' to fp32 summation order (loss 1.7e-6 relative vs 1.1e-5 against the fp32 oracle), so
    # what remains is the BACKWARD: the engine rounds every gradient activation (dh, dqkv, d_o, dx) to bf16, the oracle's
    # autograd keeps them in fp32.  Measured: worst tensor 0.0272 vs 0.0337 against the fp32 oracle (same tensor,
    # layer_5/mlp/mlp_linear_1/kernel) -- i.e. ~80 % of the headline gradient error is bf16 rounding of backward activations
    # (inherent to the bf_16 policy, not to a kernel) and ~20 % is forward dtype error incl. ReLU mask flips.
    s0 = rep["steps"][0]
    assert abs(s0["loss_hip"] - s0["loss_oracle_bf16"]) <= 2e-4 * abs(s0["loss_oracle_bf16"]), s0
    assert s0["worst_grad_rel_l2_vs_bf16_oracle"][0] <= BF16_ORACLE_GRAD_TOL, s0["worst_grad_rel_l2_vs_bf16_oracle"]
    # [r04] The bf16 oracle has ALWAYS rounded its backward tensors to bf16 (autograd's backward of the up-cast casts the
    # gradient to bf16, tests/test_oracle.py::test_bf16_oracle_rounds_gradients_and_teacher_forcing_is_consistent) -- round 3's
    # attribution of the 2.7 % to "fp32 backward activations in the oracle" was wrong.  What separates two faithful bf16
    # implementations is the FORWARD: different fp32 summation orders flip bf16 roundings and ReLU mask bits, and each flipped
    # mask fraction f costs sqrt(f) upstream.  Teacher forcing removes it: the oracle's forward takes the engine's own stored
    # activations (every LayerNorm output, q | k | v, attention output, both residual sums, the FFN hidden layer), its backward
    # is its own -- what remains is the backward arithmetic of the whole 6-layer chain.
    # Measured on MI355X (profiles/r04_parity_dalle_example.json): worst tensor 0.0176 (layer_5/attn/k, then layer_5/attn/q
    # 0.0169), EVERY other tensor <= 0.0067.  Those two are attributed: rounding dS (mode "+ds") or keeping dP in fp32 ("+dp32")
    # changes nothing (0.0176), but with the softmax backward written the flash-attention way -- delta = rowsum(dO * O) taken
    # from the bf16-ROUNDED output instead of sum_j P_j dP_j (mode "+fa", oracle/dalle_oracle.py _FlashCore) -- the oracle
    # agrees with the kernels to 0.0066 on every tensor.  The kernels' one deviation from the reference's arithmetic order is
    # therefore the flash formulation of delta (its rounding error enters every dS of a row with one sign); it is inherent to
    # not materialising P, and bounded here.  Bounds = measured + 25 %.
    print("free-running bf16 oracle, fp32 weight gradients:", s0["worst_grad_rel_l2_vs_bf16_fp32w_oracle"],
          "\nteacher-forced bf16 oracle:", s0["worst_grad_rel_l2_vs_forced_bf16_oracle"],
          "\nteacher-forced, fp32 weight gradients:", s0["worst_grad_rel_l2_vs_forced_fp32w_oracle"],
          "\nteacher-forced, flash-style delta:", s0["worst_grad_rel_l2_vs_forced_fp32w_fa_oracle"], flush=True)
    assert s0["worst_grad_rel_l2_vs_forced_fp32w_oracle"][0] <= FORCED_ORACLE_GRAD_TOL, s0["worst_grad_rel_l2_vs_forced_fp32w_oracle"]
    assert s0["worst_grad_rel_l2_vs_forced_fp32w_fa_oracle"][0] <= FORCED_FA_ORACLE_GRAD_TOL, s0["worst_grad_rel_l2_vs_forced_fp32w_fa_oracle"]
    tab = s0["grad_rel_l2_vs_forced_fp32w_oracle"]
    # [r05] the q / k gradients of EVERY layer carry the delta formulation's error (layer_5 0.0115 / 0.0110, layer_4 0.0086 / 0.0085 with
    # this round's fused LayerNorm forms; layer_5 0.0176 before) and are held to the flash-style bound above; every other tensor
    # agrees to 0.5 % with the plain teacher-forced oracle
    others = max(v for k, v in tab.items() if not (k.endswith("attn/q") or k.endswith("attn/k")))
    assert others <= FORCED_FA_ORACLE_GRAD_TOL, others
    # [r06] ... and the q / k gradients keep EXPLICIT per-layer bounds against the plain forced oracle as well (advisor, round 5: the flash-style
    # bound alone would not catch drift of the delta formulation's error).  Mechanism: delta_kernel - delta_reference = dO . e_O, with e_O the
    # error of the STORED attention output -- so these numbers measure how far O is from correctly rounded, amplified by the depth of the
    # gradient path.  History of layer_5 (q, k): 0.0176 / 0.0169 (round 4), 0.0115 / 0.0110 (round 5), 0.0157 / 0.0159 with the round-6 forward
    # kernel while it normalised O by the sum of the UNROUNDED probabilities (its integer running maximum no longer makes the dominant
    # probability exactly 1, so that term's bf16 rounding stopped cancelling: O at 1.47 x pure rounding, tools/experiments/r06_fwd_accuracy.py),
    # 0.0096 / 0.0096 since O is normalised by the sum of the ROUNDED probabilities (1.05 x).  Bounds = measured + 25 %.
    QK_FORCED_BOUND = [0.0080, 0.0047, 0.0063, 0.0089, 0.0108, 0.0120]       # measured 0.0063, 0.0037, 0.0050, 0.0071, 0.0086, 0.0096
    for l, bound in enumerate(QK_FORCED_BOUND):
        worst = max(tab[f"layer_{l}/attn/q"], tab[f"layer_{l}/attn/k"])
        assert worst <= bound, (l, worst, bound)


def test_dalle_example_shape_eval_logits_vs_fp32_oracle():
    from oracle import dalle_oracle as do
    from parity import save_report
    from src.dalle_mtf.engine import DalleEngine
    c = DALLE_EXAMPLE
    cfg = do.DalleConfig(c["n_embd"], c["text_vocab"], c["image_vocab"], c["T"], c["P"], c["n_layers"], c["n_heads"])
    P0 = do.init_params(cfg, seed=77, perturb=0.02)
    eng = DalleEngine(c["n_embd"], c["n_layers"], c["n_heads"], c["text_vocab"], c["image_vocab"], c["T"], c["P"], batch_size=1,
                      hparams=dict(lr=1e-3, train_steps=10, num_microbatches=4))   # eval ignores the micro-batch count
    eng.load_reference_params(P0)
    tokens = do.assemble_tokens(do.synthetic_captions(1, c["T"], c["text_vocab"], seed=5),
                                do.synthetic_image_tokens(1, c["P"], c["image_vocab"], seed=6), c["text_vocab"])
    loss_h = float(eng.forward(torch.from_numpy(tokens).cuda(), need_grad=False))
    got = eng.logits().cpu().numpy()
    Pt = OrderedDict((k, torch.tensor(v)) for k, v in P0.items())
    loss_o, _, ref = do.forward(Pt, tokens, cfg, bf16=False, return_logits=True)
    ref = ref.numpy()
    err = float(np.abs(got - ref).max())
    rel = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    agree = float((got.argmax(-1) == ref.argmax(-1)).mean())
    print(dict(max_abs_err=err, rel_l2=rel, ref_absmax=float(np.abs(ref).max()), loss_hip=loss_h, loss_oracle=float(loss_o),
               argmax_agreement=agree))
    save_report("parity_dalle_example_logits.json", dict(max_abs_err=err, rel_l2=rel, loss_hip=loss_h, loss_oracle=float(loss_o),
                                                         argmax_agreement=agree))
    # measured: max |err| 0.022 (logits up to ~1.9), rel-L2 0.0082, argmax agreement 99.5 %, loss 6e-5 relative
    assert err <= 2.8e-2 * max(1.0, float(np.abs(ref).max())), err
    assert rel <= 1.1e-2, rel
    assert abs(loss_h - float(loss_o)) <= 2e-4 * abs(float(loss_o))


def test_1p3b_layer_shape_step_vs_fp32_oracle():
    """one transformer block + the vocabulary head at the 1.3B width (n_embd 2048, 16 heads, K = 2048 GEMMs: the 128x128x64
    NT tiling, head dim 128 x 16 heads in attention)."""
    from parity import check_report, compare_step, save_report
    rep = compare_step(n_embd=2048, n_heads=16, n_layers=1, text_vocab=50258, image_vocab=512, T=256, P=1024, B=1, seed=31,
                       steps=1, perturb=0.02, bf16_oracle=False, per_tensor=True)
    save_report("parity_1p3b_layer.json", rep)
    # measured (profiles/r02_parity_1p3b_layer.json): loss 1.5e-6 relative, grad norm 5e-5, worst tensor 0.060 (wpe; every
    # tensor upstream of the block's ReLU is at 0.042-0.060, mlp_linear_2 / to_logits at 0.1-0.6 %: the same mask-flip floor as
    # in the dalle_example test, somewhat higher at this width)
    check_report(rep, loss_rtol=2e-4, grad_tol=0.075, gn_rtol=2e-3)


def test_1p3b_two_layer_step_vs_oracles():
    """[r04] BASELINE config 5's width with MORE than one block: n_embd 2048, 16 heads, L = 2, V = 50 771, B = 1 -- the
    256x256 NT tile takes every K >= 2048 product of both blocks (forward, input gradients) and the residual gradient stream
    crosses a block boundary.  Against the fp32 oracle and the teacher-forced bf16 oracle."""
    from parity import check_report, compare_step, save_report
    rep = compare_step(n_embd=2048, n_heads=16, n_layers=2, text_vocab=50258, image_vocab=512, T=256, P=1024, B=1, seed=41,
                       steps=1, perturb=0.02, bf16_oracle=False, bf16_grad_oracle=True, per_tensor=True)
    save_report("parity_1p3b_two_layers.json", rep)
    # measured (profiles/r04_parity_1p3b_two_layers.json): vs the fp32 oracle worst tensor 0.0292 (wpe), vs the free-running bf16
    # oracle 0.0213, TEACHER-FORCED 0.0043 (wpe; attention q / k 0.002); bounds = measured + 25 %
    check_report(rep, loss_rtol=2e-4, grad_tol=0.037, gn_rtol=2e-3)
    s0 = rep["steps"][0]
    print("teacher-forced bf16 oracle:", s0["worst_grad_rel_l2_vs_forced_fp32w_oracle"], flush=True)
    assert s0["worst_grad_rel_l2_vs_forced_fp32w_oracle"][0] <= 0.0055, s0["worst_grad_rel_l2_vs_forced_fp32w_oracle"]


def test_dalle_coco_block_step_vs_oracles():
    """[r04] one block at the `dalle_coco` shape that profiles/r0x_bench_dalle_coco.json times: n_embd 1024, 8 heads,
    V = 50258 + 2048 + 1 = 52 307 (the vae_coco codebook), 256 + 1024 positions."""
    from parity import check_report, compare_step, save_report
    rep = compare_step(n_embd=1024, n_heads=8, n_layers=1, text_vocab=50258, image_vocab=2048, T=256, P=1024, B=1, seed=51,
                       steps=1, perturb=0.02, bf16_oracle=False, bf16_grad_oracle=True, per_tensor=True)
    save_report("parity_dalle_coco_block.json", rep)
    # measured (profiles/r04_parity_dalle_coco_block.json): vs the fp32 oracle 0.0682 (layer_0/attn/k), free-running bf16 oracle
    # 0.0501, TEACHER-FORCED 0.0065 (wpe); bounds = measured + 25 %
    check_report(rep, loss_rtol=2e-4, grad_tol=0.085, gn_rtol=2e-3)
    s0 = rep["steps"][0]
    print("teacher-forced bf16 oracle:", s0["worst_grad_rel_l2_vs_forced_fp32w_oracle"], flush=True)
    assert s0["worst_grad_rel_l2_vs_forced_fp32w_oracle"][0] <= 0.0082, s0["worst_grad_rel_l2_vs_forced_fp32w_oracle"]


@pytest.mark.parametrize("B,H,S", [(1, 4, 1280), (1, 16, 1280)])
def test_attention_kernels_at_headline_shape(B, H, S):
    """attention forward + backward alone vs fp32 autograd at the benchmark's (H, S): 10 query tiles x 20 key tiles per
    head, every causal tile class, all XCD-grouped block orders in one grid."""
    from test_kernels_gpu import _attention_fwd_bwd
    _attention_fwd_bwd(B, H, S)


def test_engine_vs_the_references_own_headline_digest():
    """[r05: run, bounds = measured + 25 %] the engine against tests/golden/ref_callsite_dalle_headline.npz directly: what the reference's
    own files (src/dalle_mtf/*.py over the shims of oracle/refshim) compute at the exact dalle_example architecture, B = 1 -- loss,
    the norm of every gradient tensor and the small gradients, with the bounds of the oracle comparison above (the oracle equals
    that digest to 1e-6, tests/test_reference_callsite.py)."""
    import importlib.util
    import os
    from src.dalle_mtf.engine import DalleEngine
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_ref_callsite_golden", os.path.join(here, "golden", "make_ref_callsite_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    z = np.load(os.path.join(here, "golden", "ref_callsite_dalle_headline.npz"))
    cfg, weights, tokens = gen.case_inputs(gen.HEADLINE)
    hp = gen.HEADLINE["hp"]
    eng = DalleEngine(hp["n_embd"], hp["n_layers"], hp["n_heads"], hp["text_vocab_size"], hp["image_vocab_size"], hp["text_seq_len"],
                      hp["image_seq_len"], batch_size=1, hparams=dict(lr=hp["lr"], train_steps=hp["train_steps"], warmup_steps=hp["warmup_steps"],
                                                                      gradient_clipping=hp["gradient_clipping"]))
    eng.load_reference_params(weights)
    loss = float(eng.forward(torch.from_numpy(tokens).cuda(), need_grad=True).item())
    eng.backward()
    eng.wait_grads()
    gh = eng.export_reference(eng.g)
    assert abs(loss - float(z["loss"])) <= 2e-4 * float(z["loss"]), (loss, float(z["loss"]))
    norms = np.array([np.linalg.norm(gh[k].astype(np.float64)) for k in weights])
    rel = np.abs(norms - z["grad_norms"]) / (z["grad_norms"] + 1e-30)
    print("worst gradient-norm deviation vs the reference digest:", float(rel.max()), list(weights)[int(rel.argmax())])
    assert rel.max() < 0.0067            # measured 0.0053 (layer_3/attn/q)
    worst = max((float(np.linalg.norm(gh[k[5:]] - z[k]) / (np.linalg.norm(z[k]) + 1e-30)), k) for k in z.files if k.startswith("grad:"))
    print("worst small gradient vs the reference digest:", worst)
    assert worst[0] < 0.032             # measured 0.0255 (layer_5/norm_2/g)
    # per-position loss (reference: loss_batch of src/dalle_mtf/models.py:348-352, before the mean)
    lr_ = eng.loss_rows.float().cpu().numpy().reshape(1, -1)
    dl = np.abs(lr_ - z["loss_batch"])
    print("per-position loss vs the reference digest: max abs", float(dl.max()), "mean abs", float(dl.mean()))
    assert dl.max() <= PERPOS_LOSS_ABS and dl.mean() <= PERPOS_LOSS_MEAN_ABS, (float(dl.max()), float(dl.mean()))
    # logits (evaluation forward): the three stored rows, every position's maximum and arg-max
    eng.forward(torch.from_numpy(tokens).cuda(), need_grad=False)
    lg = eng.logits().cpu().numpy()
    pos = list(gen.HEADLINE_POSITIONS)
    rows_rel = float(np.linalg.norm(lg[:, pos, :] - z["logits_rows"]) / np.linalg.norm(z["logits_rows"]))
    dmax = float(np.abs(lg.max(-1) - z["logits_max"]).max())
    agree = float((lg.argmax(-1) == z["logits_argmax"]).mean())
    # where the arg-max differs the reference's own top value must be within the bf16 logit error of ours
    ref_top_here = np.take_along_axis(lg, z["logits_argmax"][..., None].astype(np.int64), -1)[..., 0]
    gap = float((lg.max(-1) - ref_top_here).max())
    print("logit rows rel-L2", rows_rel, "max |max-logit diff|", dmax, "arg-max agreement", agree, "largest top-gap at a disagreement", gap)
    assert rows_rel <= LOGIT_ROWS_REL and dmax <= LOGIT_MAX_ABS and agree >= ARGMAX_AGREE and gap <= LOGIT_MAX_ABS
    del eng
    torch.cuda.empty_cache()


# bounds of test_engine_vs_the_references_own_headline_digest: measured on an MI355X + 25 % (profiles/r05_parity_ref_digest.log)
# measured: per-position loss max 0.0132 / mean 0.0033 abs; logit rows rel-L2 0.0091; max-logit 0.0185 abs; arg-max agreement 0.9969
# (4 of 1280 positions, all exact bf16 ties in the engine's logits: top-gap 0.0); gradient norms 0.0053; small gradients 0.0255
PERPOS_LOSS_ABS, PERPOS_LOSS_MEAN_ABS = 0.0165, 0.0042
LOGIT_ROWS_REL, LOGIT_MAX_ABS, ARGMAX_AGREE = 0.0114, 0.0232, 0.99


def _headline_engine(B, seed=1234, hp=None):
    from src.dalle_mtf.engine import DalleEngine
    eng = DalleEngine(512, 6, 4, 50258, 512, 256, 1024, batch_size=B, global_batch_size=32,
                      hparams=dict(lr=1e-3, train_steps=100000, warmup_steps=3000, gradient_clipping=1.0, **(hp or {})))
    eng.init_params(seed=seed)
    eng.global_step = 1500
    return eng


def test_production_dispatch_at_the_benchmark_batch_is_bit_identical_to_the_128x128_kernels():
    """The step bench.py times (dalle_example, B = 32, S = 1280: M = 40 960), three arms on the same weights and tokens:
      prod   the production dispatch: layer products on gemm_ntr / gemm_nt8p / gemm_nt8 (launch_nt selects them from M >= 20 480 or
             >= 512 tiles on; the oracle-checked B <= 2 steps run gemm_nt2), LayerNorm forward / backward fused into the N = 512 products;
      plain  ntr = nt8 = nt4 = 0 and nt8p for the softmax head only (which the B <= 2 steps also run on nt8p).  This arm covers the
             PLAIN products (QKV, FFN-1, FFN-2 input gradient, the head's three products): every NT kernel claims the 128x128 kernel's
             bits (same k order) and the composed step is held to it bit for bit -- loss, per-position losses, every gradient, the
             parameters after clip + Adam.  It does NOT cover the five N = 512 products: dmi_gemm_nt_ln / dmi_gemm_nt_lnbwd launch
             gemm_ntr_kernel<0, 5, 1|2|3> whatever `ntr` says, so both arms run the same fused kernels there; nor the head's weight
             gradient, which both arms run as the gang stream-K (deterministic; held to the 128x128 kernel by
             test_kernels_gpu.py::test_gemm_tn_gang_stream_k: uncut stripes bit-identical, cut stripes the fp32 sum of two pieces);
      sep    [r06] fuse_ln = fuse_lnbwd = False: plain dmi_gemm_nt + dmi_layernorm_fwd / _bwd for those five products -- the composed
             B = 32 backward through gemm_ntr<0,5,2|3> at 256 blocks against something other than itself.  The fused forms sum their
             row reductions in another order (Y / dx within one bf16 ulp, which flips a few ReLU bits), so this arm is held to the bounds
             test_fused_layernorm_forms_equal_the_separate_kernels derives at the small shape: loss 2e-5 relative, every gradient tensor
             0.03 relative L2 (a flipped ReLU-mask fraction f ~ 1e-4 costs sqrt(f) ~ 1e-2; x 3), parameters within 6.5 lr.
    And the first two sequences' per-position losses equal those of a B = 2 engine on the same rows (whose gradients
    test_dalle_example_shape_step_vs_fp32_oracle compares with the oracle): the link from the benchmarked dispatch to the oracle."""
    import dalle_hip as dh
    from oracle import dalle_oracle as do
    B = 32
    tokens = torch.from_numpy(do.assemble_tokens(do.synthetic_captions(B, 256, 50258, seed=1),
                                                 do.synthetic_image_tokens(B, 1024, 512, seed=2), 50258)).cuda()
    names = ("ntr", "nt8p", "nt8", "nt4")
    saved = {n: dh.get_option(n) for n in names}
    out = {}
    try:
        for arm in ("prod", "plain", "sep"):
            for n in names:   # (nt8p = 3: the softmax head alone stays on the persistent kernel -- its register epilogue adds the row-sum
                dh.set_option(n, saved[n] if arm != "plain" else (3 if n == "nt8p" else 0))   # partials in its own fixed order, by design)
            eng = _headline_engine(B, hp=dict(fuse_ln=False, fuse_lnbwd=False) if arm == "sep" else None)
            assert eng.fuse_ln == (arm != "sep") and eng.fuse_lnbwd == (arm != "sep")
            loss = float(eng.train_step(tokens).item())
            torch.cuda.synchronize()
            out[arm] = (loss, eng.g.clone() if arm != "sep" else None, eng.p.clone(), eng.loss_rows.clone(), eng.export_reference(eng.g), eng.learning_rate(1500))
            del eng
            torch.cuda.empty_cache()
    finally:
        for n in names:
            dh.set_option(n, saved[n])
    prod, plain, sep = out["prod"], out["plain"], out["sep"]
    assert prod[0] == plain[0], (prod[0], plain[0])
    assert torch.equal(prod[3], plain[3]), "per-position losses differ"
    assert torch.equal(prod[1], plain[1]), float((prod[1] - plain[1]).abs().max())
    assert torch.equal(prod[2], plain[2])
    # [r06] fused vs separate LayerNorm forms at the benchmark batch
    dl = abs(prod[0] - sep[0]) / abs(sep[0])
    worst = max((float(np.linalg.norm(prod[4][k] - sep[4][k]) / (np.linalg.norm(sep[4][k]) + 1e-30)), k) for k in sep[4])
    dp = float((prod[2] - sep[2]).abs().max())
    print("B = 32, fused vs separate LayerNorm forms: loss rel", dl, "worst gradient tensor", worst, "max parameter difference", dp,
          "lr", prod[5], flush=True)
    assert dl <= 2e-5, dl
    assert worst[0] <= 0.03, worst
    assert dp <= 6.5 * prod[5] + 1e-7, (dp, prod[5])
    small = _headline_engine(2)
    small.forward(tokens[:2].contiguous(), need_grad=True)
    torch.cuda.synchronize()
    a, b = prod[3][:2 * 1280], small.loss_rows
    nd = int((a != b).sum())
    print("per-position losses, B = 32 dispatch vs B = 2 engine: differing positions", nd, "max abs", float((a - b).abs().max()))
    assert torch.equal(a, b)
    del small
    torch.cuda.empty_cache()


def test_benchmark_batch_gradients_equal_the_sum_of_the_two_sequence_gradients():
    """[r06] The link that was missing between the benchmarked batch and the oracle on the BACKWARD side: the loss is a mean over tokens and
    every sequence's activations are batch-independent (per-position losses are bit-equal across batch sizes, test above), so the gradient of
    the B = 32 step must be the SUM of the sixteen B = 2 gradients on the same sequences (both engines divide by the tokens of the GLOBAL batch
    of 32: _headline_engine passes global_batch_size = 32) -- and the B = 2 step is what
    test_dalle_example_shape_step_vs_fp32_oracle compares with the oracle tensor by tensor.  This is the test that would have caught the
    2-GiB defect of rounds 1-5 (the head's weight / bias gradient built from the first 21 130 of 40 960 rows: relative error 0.7 on those two
    tensors); with it fixed every gradient tensor agrees to bf16 / summation-order level."""
    from oracle import dalle_oracle as do
    B = 32
    tokens = torch.from_numpy(do.assemble_tokens(do.synthetic_captions(B, 256, 50258, seed=1),
                                                 do.synthetic_image_tokens(B, 1024, 512, seed=2), 50258)).cuda()
    big = _headline_engine(B)
    big.forward(tokens, need_grad=True)
    big.backward(allreduce=False)
    torch.cuda.synchronize()
    gb = big.export_reference(big.g)
    del big
    torch.cuda.empty_cache()
    small = _headline_engine(2)
    acc = None
    for i in range(0, B, 2):
        small.forward(tokens[i:i + 2].contiguous(), need_grad=True)
        small.backward(allreduce=False)
        torch.cuda.synchronize()
        gs = small.export_reference(small.g)
        acc = {k: v.astype(np.float64) for k, v in gs.items()} if acc is None else {k: acc[k] + gs[k] for k in acc}
    del small
    torch.cuda.empty_cache()
    worst = max((float(np.linalg.norm(gb[k] - acc[k]) / (np.linalg.norm(acc[k]) + 1e-30)), k) for k in gb)
    head = {k: float(np.linalg.norm(gb[k] - acc[k]) / (np.linalg.norm(acc[k]) + 1e-30)) for k in gb if "to_logits" in k}
    print("B = 32 gradient vs the sum of sixteen B = 2 gradients: worst tensor", worst, "head tensors", head, flush=True)
    # bf16 products summed in another order (row splits, tile plans differ with M): a few 1e-3; the defect was 0.7 on the head's tensors
    assert worst[0] <= 5e-3, worst                       # measured 9.5e-4 (layer_0/attn/k)
    assert all(v <= 1e-3 for v in head.values()), head   # measured 2.9e-5 (kernel), 5.8e-5 (bias)


@pytest.mark.parametrize("n_embd,n_heads,n_layers,image_vocab,B", [(2048, 16, 2, 512, 32), (1024, 8, 2, 2048, 16)])
def test_secondary_widths_batch_gradients_equal_the_sum_of_the_two_sequence_gradients(n_embd, n_heads, n_layers, image_vocab, B):
    """[r06] The same size-independent property at the widths of the other two transformer configurations and THEIR per-GPU batches
    (1.3B dimensions at B = 32: the 4.16-GB dY again, the FFN gradients on the gang stream-K with 8 / 32 gangs; dalle_coco's width at
    B = 16), two blocks deep: gradient of the batch == sum of the gradients of its pairs of sequences.  (The pairs are what the
    width-specific oracle tests cover: test_1p3b_two_layer_step_vs_oracles, tests/test_dalle_step_gpu.py.)"""
    from oracle import dalle_oracle as do
    from src.dalle_mtf.engine import DalleEngine
    tokens = torch.from_numpy(do.assemble_tokens(do.synthetic_captions(B, 256, 50258, seed=3),
                                                 do.synthetic_image_tokens(B, 1024, image_vocab, seed=4), 50258)).cuda()

    def make(b):
        eng = DalleEngine(n_embd, n_layers, n_heads, 50258, image_vocab, 256, 1024, batch_size=b, global_batch_size=B,
                          hparams=dict(lr=1e-3, train_steps=100000, warmup_steps=3000, gradient_clipping=1.0))
        eng.init_params(seed=77)
        return eng
    big = make(B)
    big.forward(tokens, need_grad=True)
    big.backward(allreduce=False)
    torch.cuda.synchronize()
    gb = big.export_reference(big.g)
    del big
    torch.cuda.empty_cache()
    small = make(2)
    acc = None
    for i in range(0, B, 2):
        small.forward(tokens[i:i + 2].contiguous(), need_grad=True)
        small.backward(allreduce=False)
        torch.cuda.synchronize()
        gs = small.export_reference(small.g)
        acc = {k: v.astype(np.float64) for k, v in gs.items()} if acc is None else {k: acc[k] + gs[k] for k in acc}
    del small
    torch.cuda.empty_cache()
    worst = max((float(np.linalg.norm(gb[k] - acc[k]) / (np.linalg.norm(acc[k]) + 1e-30)), k) for k in gb)
    print(f"n_embd {n_embd} B = {B}: gradient vs the sum of the B = 2 gradients, worst tensor", worst, flush=True)
    assert worst[0] <= 5e-3, worst


def _trajectory(start, steps=10, seed=4321):
    """`steps` free-running optimizer steps at the exact dalle_example architecture on one B = 1 batch, engine and fp32 CPU oracle each
    carrying their OWN parameters and Adam slots from identical initial weights; schedule position `start` of configs/dalle_example.json
    (lr 1e-3, 3000 warm-up steps, cosine to 100 000, clip 1.0)."""
    from oracle import dalle_oracle as do
    from src.dalle_mtf.engine import DalleEngine
    c = DALLE_EXAMPLE
    cfg = do.DalleConfig(c["n_embd"], c["text_vocab"], c["image_vocab"], c["T"], c["P"], c["n_layers"], c["n_heads"])
    hp = dict(lr=1e-3, train_steps=100000, warmup_steps=3000, gradient_clipping=1.0)
    P0 = do.init_params(cfg, seed=seed, perturb=0.02)
    tokens = do.assemble_tokens(do.synthetic_captions(1, c["T"], c["text_vocab"], seed=11),
                                do.synthetic_image_tokens(1, c["P"], c["image_vocab"], seed=12), c["text_vocab"])
    eng = DalleEngine(c["n_embd"], c["n_layers"], c["n_heads"], c["text_vocab"], c["image_vocab"], c["T"], c["P"], batch_size=1, hparams=hp)
    assert eng.fuse_ln and eng.fuse_lnbwd          # the default (fused) dispatch
    eng.load_reference_params(P0)
    eng.global_step = start
    tok_d = torch.from_numpy(tokens).cuda()
    Po = {k: v.copy() for k, v in P0.items()}
    m = {k: np.zeros_like(v) for k, v in P0.items()}
    v = {k: np.zeros_like(v) for k, v in P0.items()}
    rows = []
    for step in range(steps):
        loss_o, g = do.loss_and_grads(Po, tokens, cfg, bf16=False)
        gn_o = math.sqrt(sum(float((g[k].astype(np.float64) ** 2).sum()) for k in g))
        gc, _ = do.clip_by_global_norm(g, hp["gradient_clipping"])
        lr_o = do.learning_rate(start + step, hp["lr"], hp["train_steps"], hp["warmup_steps"])
        do.adam_step(Po, gc, m, v, lr_o)
        assert abs(lr_o - eng.learning_rate(start + step)) <= 1e-9
        loss_h = float(eng.train_step(tok_d).item())
        rows.append(dict(start=start, step=step, loss_hip=loss_h, loss_oracle=float(loss_o), rel=abs(loss_h - float(loss_o)) / abs(float(loss_o)),
                         grad_norm_hip=eng.grad_norm(), grad_norm_oracle=gn_o, lr=lr_o))
        print(rows[-1], flush=True)
    ph = eng.export_reference(eng.p)
    drift = max(float(np.abs(ph[k] - Po[k]).max()) for k in Po)
    del eng
    torch.cuda.empty_cache()
    return rows, drift


def test_ten_step_trajectory_at_the_dalle_example_shape_vs_fp32_oracle():
    """[r06] SURVEY.md §8(c): "bf16 compute vs fp32 oracle: |dloss| <= 1e-2 relative over first 10 steps" (reference:
    src/dalle_mtf/models.py:397-416 loss, src/optimizers.py:34-104 clip + schedule + Adam without bias correction), on the default --
    fused -- dispatch at the exact dalle_example architecture, B = 1, S = 1280.
    (a) THE CONTRACT: the first ten steps of training as the shipped config runs them -- global steps 0..9 of a 3000-step linear warm-up
        (lr = 0, 3.3e-7, ..., 3e-6): |dloss| <= 1e-2 relative at every step (measured <= 1e-4) and the parameters of the two runs stay
        within 2 x 6.6 x sum(lr) of each other (Adam without bias correction moves a weight by lr * m / sqrt(v) with
        |m| / sqrt(v) <= (1 - 0.9^t) / sqrt(1 - 0.999^t): 3.16 at t = 1, 6.5 at t = 10; a noise-level gradient whose sign differs
        moves it the other way).
    (b) A STRESS the contract does not ask for, kept because it is the only trajectory in the suite that actually descends: ten steps from
        schedule position 1500 (lr 5e-4) on that single sequence -- the loss falls 11.0 -> 5.0, i.e. the model memorises the batch with
        sign-like 1.6e-3 updates per weight and step.  Steps 0..4 agree to 2e-4 (asserted 1e-3).  From step 5 the fit OVERSHOOTS: the
        oracle's own gradient norm jumps 1.08 -> 2.80 -> 3.18 -> 1.01 and the engine's 2.01 -> 1.04 -> 0.77 -> 1.64 -- the same oscillation
        one step apart, which is what a bf16 perturbation of a marginally stable trajectory does -- and the losses differ by up to 1.7 %
        for three steps before they meet again (6e-4 at step 8, 2.4e-3 at step 9).  That phase is chaotic (it also depends on the host's
        fp32 summation order inside the CPU oracle), so only a sanity bound is asserted there: 1e-1 on steps 5..9, reported in full."""
    from parity import save_report
    rows_a, drift_a = _trajectory(0)
    lr_sum = sum(r["lr"] for r in rows_a)
    for r in rows_a:
        assert r["rel"] <= 1e-2, r
    assert drift_a <= 2 * 6.6 * lr_sum + 1e-7, (drift_a, lr_sum)
    rows_b, _ = _trajectory(1500)
    save_report("parity_dalle_example_trajectory.json", dict(contract=rows_a, contract_param_drift=drift_a, stress=rows_b))
    assert rows_b[-1]["loss_oracle"] < rows_b[0]["loss_oracle"] - 0.5, "the stress trajectory must actually move"
    for r in rows_b:
        assert r["rel"] <= (1e-3 if r["step"] < 5 else 1e-1), r
