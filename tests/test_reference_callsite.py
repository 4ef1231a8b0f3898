"""The oracle against the REFERENCE'S OWN CODE: tests/golden/ref_callsite_dalle.npz holds what the
reference's src/dalle_mtf/models.py + layers.py + ops.py + src/optimizers.py compute when executed, unmodified, over the
tensorflow / mesh-tensorflow shims of oracle/refshim (tests/golden/make_ref_callsite_golden.py).  The oracle must reproduce every
number of it from the same weights and tokens: logits, per-position loss, loss, the gradient of every variable, the learning
rate, the clip, the Adam update -- i.e. its restatement of the reference's call graph is checked against the call graph itself.
(The shims restate the third-party primitives, SURVEY.md Appendix A: those stay unpinned, and DESIGN.md §2 says so.)

CPU only; where the reference checkout exists (the authoring container) the fixture is also regenerated and compared."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

from oracle import dalle_oracle as do
from oracle import refshim
from oracle import vae_oracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "ref_callsite_dalle.npz")
FIXTURE_VAE = os.path.join(HERE, "golden", "ref_callsite_vae.npz")

_spec = importlib.util.spec_from_file_location("make_ref_callsite_golden", os.path.join(HERE, "golden", "make_ref_callsite_golden.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


@pytest.fixture(scope="module")
def blob():
    z = np.load(FIXTURE)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def vblob():
    z = np.load(FIXTURE_VAE)
    return {k: z[k] for k in z.files}


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def test_fixture_cases_are_the_generator_cases(blob):
    assert json.loads(str(blob["cases"])) == json.loads(json.dumps(gen.CASES))


@pytest.mark.parametrize("name", ["a", "b"])
def test_variable_table_is_the_references(blob, name):
    """names, creation order, shapes and initialiser constants of the variables the reference's code creates (mtf.get_variable /
    mtf.layers.dense / attention_params_simple call sites) == oracle.param_specs"""
    case = gen.CASES[name]
    cfg, _, _ = gen.case_inputs(case)
    ref = json.loads(str(blob[name + "/variables"]))
    spec = do.param_specs(cfg)
    assert list(ref.keys()) == list(spec.keys())
    for n, (shape, kind, std) in spec.items():
        rshape, rkind, rconst = ref[n]
        assert tuple(rshape) == tuple(shape), n
        if kind == "normal":
            assert rkind == "normal" and rconst == pytest.approx(std, rel=1e-12), (n, rconst, std)
        else:
            assert rkind == "constant" and rconst == (1.0 if kind == "ones" else 0.0), n


@pytest.mark.parametrize("name", ["a", "b"])
def test_oracle_reproduces_the_reference_step(blob, name):
    """fp32: forward, gradients, schedule, clip and Adam of the oracle == the reference's, to fp32 summation-order noise"""
    case = gen.CASES[name]
    hp = case["hp"]
    cfg, weights, tokens = gen.case_inputs(case)
    g = lambda k: blob[name + "/" + k]   # noqa: E731
    P = {n: torch.tensor(a) for n, a in weights.items()}
    loss, loss_batch, logits = do.forward(P, tokens, cfg, return_logits=True)
    assert _rel(logits.numpy(), g("logits")) < 5e-6
    assert np.abs(loss_batch.numpy() - g("loss_batch")).max() < 2e-5
    assert abs(float(loss) - float(g("loss"))) < 5e-6 * abs(float(g("loss")))
    loss2, grads = do.loss_and_grads(weights, tokens, cfg)
    for n, gr in grads.items():
        assert _rel(gr, g("grad:" + n)) < 2e-5, (n, _rel(gr, g("grad:" + n)))
    # one optimizer step at the case's global step
    params = {n: a.copy() for n, a in weights.items()}
    m = {n: np.zeros_like(a) for n, a in weights.items()}
    v = {n: np.zeros_like(a) for n, a in weights.items()}
    _, gnorm, lr = do.train_step(params, m, v, tokens, cfg, case["step"], hp)
    assert lr == pytest.approx(float(g("lr")), rel=2e-6)
    clipped, _ = do.clip_by_global_norm(grads, hp["gradient_clipping"])
    np.testing.assert_allclose([np.linalg.norm(c.astype(np.float64)) for c in clipped.values()], g("clipped_norms"), rtol=2e-5, atol=1e-12)
    np.testing.assert_allclose([np.linalg.norm(params[n].astype(np.float64)) for n in grads], g("after_norms"), rtol=1e-6)
    for k in gen.FULL:
        k = k.format(last=hp["n_layers"] - 1)
        assert _rel(clipped[k], g("clipped:" + k)) < 2e-5, k
        # the update is lr * m' / (sqrt(v') + eps) [+ decay]: compare the STEP, not the parameter it is a small part of
        step_o, step_r = params[k] - weights[k], g("after:" + k) - weights[k]
        assert _rel(step_o, step_r) < 2e-4, (k, _rel(step_o, step_r))
        assert _rel(m[k], g("after:" + k + "/adam_m")) < 2e-5 and _rel(v[k], g("after:" + k + "/adam_v")) < 4e-5, k
    if hp.get("weight_decay"):    # the decayed / not decayed split of exclude_from_weight_decay = ["norm", "bias"] was exercised
        assert do.use_weight_decay("layer_0/attn/q", hp["weight_decay"]) and not do.use_weight_decay("layer_0/norm_1/g", hp["weight_decay"])


def test_bf16_reference_sits_within_the_bf16_band(blob):
    """case c: the reference with "bf_16": true over shims that round EVERY mtf op's output to bfloat16, against the oracle's bf16
    mode (which rounds at tensor boundaries and keeps e.g. the LayerNorm arithmetic in fp32).  Two different bf16 evaluations of
    one graph: the test bounds their distance (loss 2e-3, logits 3e-2 relative) -- informational, it pins no arithmetic."""
    case = gen.CASES["c"]
    cfg, weights, tokens = gen.case_inputs(case)
    P = {n: torch.tensor(a) for n, a in weights.items()}
    loss, _, logits = do.forward(P, tokens, cfg, bf16=True, return_logits=True)
    assert abs(float(loss) - float(blob["c/loss"])) < 2e-3 * float(blob["c/loss"])
    assert _rel(logits.numpy(), blob["c/logits"]) < 3e-2


@pytest.mark.parametrize("name", ["v1", "v2", "v3"])
def test_vae_oracle_reproduces_the_reference(vblob, name):
    """src/vae_tf/models.py + layers.py executed over the TF shim (tf.layers.conv2d / conv2d_transpose SAME, tf.get_variable scopes,
    the tied codebook, Gumbel noise from injected uniforms) vs oracle/vae_oracle.py: variable names / order / shapes, encoder
    logits, reconstruction, loss and the gradient of every variable; hard (straight-through) and soft Gumbel, stack_factor 1 / 2."""
    case = gen.VAE_CASES[name]
    cfg, weights, img, u = gen.vae_case_inputs(case)
    g = lambda k: vblob[name + "/" + k]   # noqa: E731
    ref_vars = json.loads(str(g("variables")))
    assert list(ref_vars.keys()) == list(vo.param_specs(cfg).keys())
    for n, shape in vo.param_specs(cfg).items():
        assert tuple(ref_vars[n]) == tuple(shape), n
    P = {n: torch.tensor(a) for n, a in weights.items()}
    logits = vo.forward(P, torch.tensor(img), cfg, return_logits=True)
    assert _rel(logits.numpy(), g("logits")) < 5e-6
    loss, out = vo.forward(P, torch.tensor(img), cfg, torch.tensor(u), return_recon_loss=True, hard_gumbel=case["hard"],
                           temperature=case["temperature"])
    assert _rel(out.numpy(), g("reconstruction")) < 5e-6 and abs(float(loss) - float(g("loss"))) < 5e-6 * float(g("loss"))
    _, grads = vo.loss_and_grads(weights, img, u, cfg, hard=case["hard"], temp=case["temperature"])[:2]
    for n, gr in grads.items():
        assert _rel(gr, g("grad:" + n)) < 2e-5, (n, _rel(gr, g("grad:" + n)))


def test_vae_oracle_reproduces_the_reference_at_the_vae_example_architecture(vblob):
    """configs/vae_example.json's network (three stride-2 stages of 64 / 128 / 256 channels with two residual layers each, 512 tokens,
    32x32 images), B = 2: digest of the reference's run vs the same digest of the oracle's"""
    case = gen.VAE_HEADLINE
    cfg, weights, img, u = gen.vae_case_inputs(case)
    P = {n: torch.tensor(a, requires_grad=True) for n, a in weights.items()}
    logits = vo.forward(P, torch.tensor(img), cfg, return_logits=True)
    loss, out = vo.forward(P, torch.tensor(img), cfg, torch.tensor(u), return_recon_loss=True, hard_gumbel=True, temperature=1.0)
    loss.backward()
    d = gen.vae_digest(loss.detach().numpy(), out.detach().numpy(), logits.detach().numpy(), {n: p.grad.numpy() for n, p in P.items()})
    g = lambda k: vblob["vh/" + k]   # noqa: E731
    assert _rel(d["logits"], g("logits")) < 5e-6 and _rel(d["reconstruction"], g("reconstruction")) < 5e-6
    assert abs(float(d["loss"]) - float(g("loss"))) < 5e-6 * float(g("loss"))
    np.testing.assert_allclose(d["grad_norms"], g("grad_norms"), rtol=2e-5)
    for k in d:
        if k.startswith("grad:"):
            assert _rel(d[k], g(k)) < 5e-5, k


@pytest.mark.skipif(not refshim.available(), reason="the reference checkout is not on this machine")
def test_vae_headline_fixture_is_what_the_reference_computes_here(vblob):
    for k, a in gen.run_vae_headline().items():
        np.testing.assert_allclose(a, vblob["vh/" + k], rtol=1e-5, atol=1e-7, err_msg=k)


@pytest.mark.skipif(not refshim.available(), reason="the reference checkout is not on this machine")
def test_the_references_recompute_grad_changes_no_number(vblob):
    """v3 = v1 with recompute_grad: the reference's custom-gradient recompute (row v5) must reproduce v1 bit for bit"""
    for k in vblob:
        if k.startswith("v1/") and not k.endswith("variables"):
            np.testing.assert_array_equal(vblob[k], vblob["v3/" + k[3:]], err_msg=k)


@pytest.mark.parametrize("name", ["v1", "v2", "v3"])
def test_vae_fixture_is_what_the_reference_computes_here(vblob, name):
    out = gen.run_vae_case(gen.VAE_CASES[name])
    for k, a in out.items():
        ref = vblob[name + "/" + k]
        if a.dtype.kind in "US":
            assert str(a) == str(ref), k
        else:
            np.testing.assert_allclose(a, ref, rtol=1e-5, atol=1e-7, err_msg=k)


@pytest.mark.skipif(not refshim.available(), reason="the reference checkout is not on this machine (it never is on the GPU box)")
@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_fixture_is_what_the_reference_computes_here(blob, name):
    """re-executes the reference's files over the shims and compares with the committed fixture"""
    out = gen.run_case(gen.CASES[name])
    for k, a in out.items():
        ref = blob[name + "/" + k]
        if a.dtype.kind in "US":
            assert str(a) == str(ref), k
        else:
            np.testing.assert_allclose(a, ref, rtol=1e-5, atol=1e-7, err_msg=k)


@pytest.mark.skipif(not refshim.available(), reason="the reference checkout is not on this machine")
def test_shims_leave_sys_modules_clean():
    import sys
    before = {k for k in sys.modules if k.split(".")[0] in ("tensorflow", "mesh_tensorflow", "_dalle_mtf_reference")}
    with refshim.installed():
        m = refshim.reference_module("dalle_mtf.models")
        assert m.__file__.startswith(refshim.DEFAULT_ROOT)
    after = {k for k in sys.modules if k.split(".")[0] in ("tensorflow", "mesh_tensorflow", "_dalle_mtf_reference")}
    assert before == after
    # the reference tree is read-only for this build: importing it must not drop byte-code caches into it
    for sub in ("", "dalle_mtf", "vae_tf", "utils"):
        assert not os.path.exists(os.path.join(refshim.DEFAULT_ROOT, "src", sub, "__pycache__")), sub


@pytest.mark.skipif(not refshim.available(), reason="the reference checkout is not on this machine")
def test_input_helpers_are_the_references():
    """src/input_fns.py:4-38 executed over the TF shim against the product's numpy restatement (dalle-mtf_amd/src/input_fns.py):
    crop_center_and_resize -- including the reference's swapped w / h names and its box [(1-wn)/2, (1-hn)/2, wn, hn] --,
    decode_img's normalisation, truncate_or_pad_label (pad by text_seq_len, keep the first text_seq_len ids)."""
    import io
    import sys
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dalle-mtf_amd"))
    from src import input_fns as prod
    rng = np.random.default_rng(0)
    with refshim.installed():
        ref = refshim.reference_module("input_fns")
        for (H, W) in ((40, 40), (48, 30), (21, 64), (16, 16)):
            img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
            np.testing.assert_array_equal(ref.crop_center_and_resize(torch.as_tensor(img), 16).numpy(), prod.crop_center_and_resize(img, 16))
        for n in (0, 3, 8, 20):
            lab = rng.integers(0, 100, size=n)
            params = {"text_seq_len": 8, "padding_id": 99}
            np.testing.assert_array_equal(ref.truncate_or_pad_label(torch.as_tensor(lab), params).numpy(), prod.truncate_or_pad_label(lab, params))
        buf = io.BytesIO()
        Image.fromarray(rng.integers(0, 256, size=(24, 36, 3), dtype=np.uint8)).save(buf, format="JPEG", quality=90)
        np.testing.assert_allclose(ref.decode_img(buf.getvalue(), 16, channels=3).numpy(), prod.decode_img(buf.getvalue(), 16, channels=3), atol=1e-5)


@pytest.mark.skipif(not refshim.available(), reason="the reference checkout is not on this machine")
def test_product_api_extends_the_references_signatures():
    """the drop-in boundary, mechanically: every parameter of the reference's DALLE / DiscreteVAE constructors and forward methods,
    of its input functions and config loader exists in the product's counterpart at the same position with the same default
    (the product appends device / process-group arguments)"""
    import inspect
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dalle-mtf_amd"))
    import src.dalle_mtf.models as pm
    import src.input_fns as pi
    import src.utils.utils as pu
    import src.vae_tf.models as pv
    with refshim.installed():
        rm, rv = refshim.reference_module("dalle_mtf.models"), refshim.reference_module("vae_tf.models")
        ri, ru = refshim.reference_module("input_fns"), refshim.reference_module("utils.utils")
        pairs = [(rm.DALLE.__init__, pm.DALLE.__init__), (rm.DALLE.forward, pm.DALLE.forward),
                 (rv.DiscreteVAE.__init__, pv.DiscreteVAE.__init__), (rv.DiscreteVAE.forward, pv.DiscreteVAE.forward),
                 (ri.dalle_input_fn, pi.dalle_input_fn), (ri.vae_input_fn, pi.vae_input_fn),
                 (ri.truncate_or_pad_label, pi.truncate_or_pad_label), (ri.read_labeled_tfrecord, pi.read_labeled_tfrecord),
                 (ri.read_tfrecord, pi.read_tfrecord), (ru.fetch_model_params, pu.fetch_model_params)]
        for r, p in pairs:
            rp, pp = list(inspect.signature(r).parameters.values()), list(inspect.signature(p).parameters.values())
            assert len(pp) >= len(rp), (r.__qualname__, rp, pp)
            for a, b in zip(rp, pp):
                assert (a.name, a.default, a.kind) == (b.name, b.default, b.kind), (r.__qualname__, a, b)


def test_oracle_reproduces_the_reference_at_the_headline_shape():
    """the exact `dalle_example` architecture of BASELINE.json (d = 512, 6 layers, 4 heads, S = 256 + 1024, V = 50 771), B = 1, fp32:
    tests/golden/ref_callsite_dalle_headline.npz is a digest of what the reference's files compute over the shims (generated once:
    70 s, 14 GB); the oracle's forward / backward reduced to the same digest must agree -- the shape the GPU parity tests
    (tests/test_headline_parity_gpu.py) compare the HIP engine with this oracle at."""
    z = np.load(os.path.join(HERE, "golden", "ref_callsite_dalle_headline.npz"))
    assert json.loads(str(z["case"])) == json.loads(json.dumps(gen.HEADLINE))
    cfg, weights, tokens = gen.case_inputs(gen.HEADLINE)
    P = {n: torch.tensor(a, requires_grad=True) for n, a in weights.items()}
    loss, loss_batch, logits = do.forward(P, tokens, cfg, return_logits=True)
    loss.backward()
    grads = {n: p.grad.numpy() for n, p in P.items()}
    d = gen.headline_digest(loss.detach().numpy(), loss_batch.detach().numpy(), logits.detach().numpy(), grads)
    assert abs(float(d["loss"]) - float(z["loss"])) < 5e-6 * float(z["loss"])
    assert np.abs(d["loss_batch"] - z["loss_batch"]).max() < 5e-5
    assert _rel(d["logits_rows"], z["logits_rows"]) < 5e-6 and _rel(d["logits_max"], z["logits_max"]) < 5e-6
    assert (d["logits_argmax"] != z["logits_argmax"]).mean() < 0.002      # ties within fp32 noise only
    np.testing.assert_allclose(d["grad_norms"], z["grad_norms"], rtol=2e-5)
    for k in z.files:
        if k.startswith("grad:"):
            assert _rel(d[k], z[k]) < 5e-5, (k, _rel(d[k], z[k]))
    lr = do.learning_rate(gen.HEADLINE["step"], 1e-3, 100000, 3000)
    assert lr == pytest.approx(float(z["lr"]), rel=2e-6)


@pytest.mark.skipif(not (refshim.available() and os.environ.get("DALLE_REFSHIM_HEADLINE", "0") != "0"),
                    reason="re-executing the reference at the headline shape takes 70 s and 14 GB: DALLE_REFSHIM_HEADLINE=1")
def test_headline_fixture_is_what_the_reference_computes_here():
    z = np.load(os.path.join(HERE, "golden", "ref_callsite_dalle_headline.npz"))
    out = gen.run_headline()
    for k, a in out.items():
        np.testing.assert_allclose(a, z[k], rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.fixture(scope="module")
def fblob():
    z = np.load(os.path.join(HERE, "golden", "ref_callsite_model_fns.npz"))
    return {k: z[k] for k in z.files}


def test_oracle_reproduces_the_references_vae_model_fn(fblob):
    """the reference's `vae_model_fn` (src/model_fns_tf.py:9-114) called itself: temperature anneal from the global step (:40-45),
    train_gumbel_hard, scope "vae", tf.train.AdamOptimizer(lr) -- vs oracle.temperature / loss_and_grads / tf_adam_step (row v6)"""
    assert json.loads(str(fblob["cases"])) == json.loads(json.dumps(gen.FN_CASES))
    c = gen.FN_CASES["fv"]
    p = c["params"]
    cfg, w, img, u = gen.fn_vae_inputs(c)
    temp = vo.temperature(c["step"], p)
    assert temp == pytest.approx(1.0 - 0.3 * 0.5, rel=1e-6)
    loss, grads = vo.loss_and_grads(w, img, u, cfg, hard=p["train_gumbel_hard"], temp=temp)[:2]
    assert abs(float(loss) - float(fblob["fv/loss"])) < 5e-6 * float(fblob["fv/loss"])
    m, v = {n: np.zeros_like(a) for n, a in w.items()}, {n: np.zeros_like(a) for n, a in w.items()}
    after = {n: a.copy() for n, a in w.items()}
    vo.tf_adam_step(after, grads, m, v, int(fblob["fv/t"]), p["lr"])
    for n in w:
        assert _rel(grads[n], fblob["fv/grad:" + n]) < 2e-5, n
        assert _rel(after[n] - w[n], fblob["fv/after:" + n] - w[n]) < 2e-4, n


def test_oracle_reproduces_the_references_dalle_model_fn(fblob):
    """the reference's `dalle_model_fn` (src/model_fns.py:55-236) called itself with images and caption ids: the tokens it assembles
    (VAE encoder arg-max, reshape, + text_vocab_size, concat with the text: rows a1 / a2), image_seq_len, the loss, the variables
    after its update ops and the global-step increment -- vs the oracle's pipeline of the same steps.  Also records what the
    reference restores: nothing -- initialize_vae_weights (:66) collects the variables under "vae" BEFORE vae.forward (:73)
    creates them (the product restores the VAE checkpoint for real)."""
    c = gen.FN_CASES["fd"]
    p = c["params"]
    vcfg, vw, cfg, dw, img, text = gen.fn_dalle_inputs(c)
    logits = vo.forward({n: torch.tensor(a) for n, a in vw.items()}, torch.tensor(img), vcfg, return_logits=True)
    tokens = do.assemble_tokens(text, do.image_tokens_from_logits(logits.numpy()), p["text_vocab_size"])
    np.testing.assert_array_equal(tokens, fblob["fd/tokens"])
    assert fblob["fd/tokens"].dtype == np.int32 and cfg.image_seq_len == vcfg.grid ** 2
    P2 = {n: a.copy() for n, a in dw.items()}
    m, v = {n: np.zeros_like(a) for n, a in dw.items()}, {n: np.zeros_like(a) for n, a in dw.items()}
    loss, _, _ = do.train_step(P2, m, v, tokens, cfg, c["step"], p)
    assert abs(loss - float(fblob["fd/loss"])) < 5e-6 * float(fblob["fd/loss"])
    for n in dw:
        assert _rel(P2[n] - dw[n], fblob["fd/after:" + n] - dw[n]) < 2e-4, n
    assert int(fblob["fd/next_global_step"]) == c["step"] + 1
    assert json.loads(str(fblob["fd/restore_requests"])) == [[p["vae_checkpoint_path"], []]]


@pytest.mark.skipif(not refshim.available(), reason="the reference checkout is not on this machine")
def test_model_fn_fixture_is_what_the_reference_computes_here(fblob, capsys):
    out = gen.run_fn_cases()
    capsys.readouterr()      # the reference prints its parameter count and dimension names
    for k, a in out.items():
        if a.dtype.kind in "US":
            assert str(a) == str(fblob[k]), k
        else:
            np.testing.assert_allclose(a, fblob[k], rtol=1e-5, atol=1e-7, err_msg=k)


@pytest.mark.skipif(not refshim.available(), reason="the reference checkout is not on this machine")
def test_the_gpu_goldens_are_what_the_reference_computes():
    """tests/golden/dalle_small.npz and vae_small.npz were generated from the ORACLE (tests/golden/make_golden.py) and are what the
    GPU tests compare the HIP path with (tests/test_golden.py::test_hip_matches_*_golden).  Here the reference's own files are run
    on the same configurations, weights and inputs: the committed goldens must be what the reference computes -- which makes
    those GPU tests comparisons against the reference's call graph."""
    from oracle.refshim import harness
    _s = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(_s)
    _s.loader.exec_module(mg)
    # DALL-E: n_embd 128, one head (head dim 128: the engine's), 2 layers, 24 + 40 positions
    z = np.load(os.path.join(HERE, "golden", "dalle_small.npz"))
    cfg = do.DalleConfig(**mg.DALLE_SMALL)
    P = do.init_params(cfg, seed=77, perturb=0.05)
    tokens = do.assemble_tokens(do.synthetic_captions(2, cfg.text_seq_len, cfg.text_vocab_size, seed=5),
                                do.synthetic_image_tokens(2, cfg.image_seq_len, cfg.image_vocab_size, seed=6), cfg.text_vocab_size)
    np.testing.assert_array_equal(tokens, z["tokens"])
    hp = dict(mg.DALLE_SMALL, bf_16=False, **mg.HP)
    r = harness.run_dalle_step(hp, P, tokens, global_step=1)
    assert _rel(r["logits"], z["logits"]) < 5e-6 and np.abs(r["loss_batch"] - z["loss_batch"]).max() < 2e-5
    assert abs(float(r["loss"]) - float(z["loss"])) < 5e-6 * float(z["loss"]) and float(r["lr"]) == pytest.approx(float(z["lr"]), rel=2e-6)
    for k in z.files:
        if k.startswith("grad:"):
            assert _rel(r["grads"][k[5:]], z[k]) < 2e-5, k
        if k.startswith("after:"):
            assert _rel(r["updated"][k[6:]] - P[k[6:]], z[k] - P[k[6:]]) < 2e-4, k
    # VAE: 16x16 images, two stages, 64 tokens; hard and soft Gumbel from the same injected uniforms
    z = np.load(os.path.join(HERE, "golden", "vae_small.npz"))
    vcfg = vo.VaeConfig(**mg.VAE_SMALL)
    VP = vo.init_params(vcfg, seed=11, bias_perturb=0.02)
    hpv = dict(num_tokens=mg.VAE_SMALL["num_tokens"], n_embd=512, hidden_dim=64, convblocks=mg.VAE_SMALL["convblocks"], stack_factor=1)
    for tag, hard, temp in (("hard", True, 1.0), ("soft", False, 0.7)):
        rv = harness.run_vae_step(hpv, VP, z["img"], z["u"], hard_gumbel=hard, temperature=temp)
        assert _rel(rv["logits"], z["logits"]) < 5e-6 and _rel(rv["reconstruction"], z["recon_" + tag]) < 5e-6
        assert abs(float(rv["loss"]) - float(z["loss_" + tag])) < 5e-6 * float(z["loss_" + tag])
        for n, g in rv["grads"].items():
            assert _rel(g, z["grad_%s:%s" % (tag, n)]) < 2e-5, (tag, n)
    np.testing.assert_array_equal(np.argmax(rv["logits"], -1).reshape(2, -1).astype(np.int32), z["tokens"])


@pytest.mark.skipif(not refshim.available(), reason="the reference checkout is not on this machine")
def test_oracle_equals_the_reference_on_random_small_configurations():
    """twelve random small configurations (depth, heads, widths, vocabularies, sequence split, batch; decay kind, warm-up incl. 0,
    decay end, clip norm, Adam constants, weight decay, global step): the reference's files over the shims vs the oracle, live"""
    from oracle.refshim import harness
    rng = np.random.default_rng(2024)
    for trial in range(12):
        heads = int(rng.integers(1, 5))
        kv = int(rng.choice([8, 16, 24]))
        hp = dict(n_embd=heads * kv, text_vocab_size=int(rng.integers(8, 40)), image_vocab_size=int(rng.integers(4, 20)),
                  text_seq_len=int(rng.integers(1, 7)), image_seq_len=int(rng.integers(1, 10)), n_layers=int(rng.integers(1, 4)),
                  n_heads=heads, bf_16=False, lr=float(rng.choice([1e-4, 3e-4, 1e-3])), train_steps=int(rng.integers(50, 400)),
                  warmup_steps=int(rng.choice([0, 10, 60])), lr_decay=str(rng.choice(["cosine", "linear", "none"])),
                  gradient_clipping=float(rng.choice([0.25, 1.0, 100.0])), weight_decay=float(rng.choice([0.0, 0.01])),
                  beta_1=float(rng.choice([0.9, 0.8])), beta_2=float(rng.choice([0.999, 0.95])), epsilon=float(rng.choice([1e-6, 1e-8])),
                  recompute_grad=bool(rng.integers(0, 2)))
        if rng.integers(0, 2):
            hp["lr_decay_end"] = int(hp["train_steps"] // 2)
        step = int(rng.integers(0, hp["train_steps"] + 20))
        batch = int(rng.integers(1, 4))
        cfg = do.DalleConfig(hp["n_embd"], hp["text_vocab_size"], hp["image_vocab_size"], hp["text_seq_len"], hp["image_seq_len"],
                             hp["n_layers"], hp["n_heads"])
        w = do.init_params(cfg, seed=trial, perturb=0.05)
        tokens = do.assemble_tokens(do.synthetic_captions(batch, cfg.text_seq_len, cfg.text_vocab_size, seed=trial + 1),
                                    do.synthetic_image_tokens(batch, cfg.image_seq_len, cfg.image_vocab_size, seed=trial + 2), cfg.text_vocab_size)
        r = harness.run_dalle_step(hp, w, tokens, global_step=step)
        assert list(r["variables"]) == list(do.param_specs(cfg)), (trial, hp)
        P2 = {n: a.copy() for n, a in w.items()}
        m, v = {n: np.zeros_like(a) for n, a in w.items()}, {n: np.zeros_like(a) for n, a in w.items()}
        loss, gnorm, lr = do.train_step(P2, m, v, tokens, cfg, step, hp)
        assert abs(loss - float(r["loss"])) < 5e-6 * abs(float(r["loss"])), (trial, hp)
        assert lr == pytest.approx(float(r["lr"]), rel=3e-6, abs=1e-12), (trial, hp, step)
        for n in w:
            a, b = P2[n] - w[n], r["updated"][n] - w[n]
            # the step m' / (sqrt(v') + eps) of an entry whose gradient is at the fp32 noise of the backward is ill-conditioned
            # (eps 1e-8), and the step is recovered from the rounded parameter: compare the tensors in L2
            assert _rel(a, b) < 1e-3 or np.abs(a - b).max() <= 4 * np.spacing(np.abs(w[n]).max()), (trial, n, _rel(a, b), hp, step)


@pytest.mark.skipif(not refshim.available(), reason="the reference checkout is not on this machine")
def test_vae_oracle_equals_the_reference_on_random_small_configurations():
    """eight random VAE configurations (1-3 stages, 1-3 layers per stage incl. stages without residual layers, channel widths,
    codebook size, stack_factor 1 / 2 / 4, hard / soft Gumbel, temperature, recompute_grad), live"""
    from oracle.refshim import harness
    rng = np.random.default_rng(77)
    for trial in range(8):
        nst = int(rng.integers(1, 4))
        blocks = [[int(rng.integers(1, 4)), int(rng.choice([8, 16, 24]))] for _ in range(nst)]
        sf = int(rng.choice([1, 2, 4]))
        size = (2 ** nst) * sf * int(rng.integers(1, 3))
        hp = dict(num_tokens=int(rng.choice([16, 40])), n_embd=64, hidden_dim=16, convblocks=blocks, stack_factor=sf,
                  recompute_grad=bool(rng.integers(0, 2)))
        hard, temp, batch = bool(rng.integers(0, 2)), float(rng.choice([0.5, 1.0, 2.0])), int(rng.integers(1, 3))
        cfg = vo.VaeConfig(hp["num_tokens"], size, blocks, stack_factor=sf)
        w = vo.init_params(cfg, seed=trial, bias_perturb=0.05)
        img = vo.synthetic_images(batch, size, seed=trial + 1)
        u = vo.synthetic_uniforms((batch, cfg.grid, cfg.grid, cfg.num_tokens), seed=trial + 2)
        r = harness.run_vae_step(hp, w, img, u, hard_gumbel=hard, temperature=temp)
        assert list(r["variables"]) == list(vo.param_specs(cfg)), (trial, hp)
        loss, grads = vo.loss_and_grads(w, img, u, cfg, hard=hard, temp=temp)[:2]
        assert abs(float(loss) - float(r["loss"])) < 5e-6 * float(r["loss"]), (trial, hp)
        for n in w:
            assert _rel(grads[n], r["grads"][n]) < 2e-5, (trial, n, hp)


@pytest.mark.skipif(not refshim.available(), reason="the reference checkout is not on this machine")
def test_fetch_model_params_is_the_references():
    """src/utils/utils.py:13-17 executed on every shipped config: same keys and values, missing key -> None, as the product's"""
    import glob
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dalle-mtf_amd"))
    import src.utils.utils as pu
    with refshim.installed():
        ru = refshim.reference_module("utils.utils")
        files = sorted(glob.glob(os.path.join(os.path.dirname(HERE), "configs", "*.json"))) + \
            sorted(glob.glob(os.path.join(refshim.DEFAULT_ROOT, "configs", "*.json")))
        assert len(files) >= 8
        for f in files:
            a, b = ru.fetch_model_params(f), pu.fetch_model_params(f)
            assert dict(a) == dict(b) and a["no_such_key"] is None and b["no_such_key"] is None, f
