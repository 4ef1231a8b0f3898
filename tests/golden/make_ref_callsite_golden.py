"""Generates tests/golden/ref_callsite_dalle.npz and ref_callsite_vae.npz by EXECUTING THE REFERENCE'S OWN FILES
(src/dalle_mtf/models.py, layers.py, ops.py, src/optimizers.py; src/vae_tf/models.py, layers.py -- imported from /root/reference,
nothing is copied) over the tensorflow / mesh-tensorflow shims of
oracle/refshim (run from the repo root: `python tests/golden/make_ref_callsite_golden.py`; needs the reference checkout, i.e.
the authoring container -- the committed .npz is what travels).

Every array in the file is an output of the reference's code: loss, per-position loss, logits, the gradient of every trainable
variable (mtf.gradients), the learning rate of its schedule, the clipped gradients and the variables / Adam slots after
get_optimizer's update ops.  Inputs are reproducible from the stored hyper-parameters and seeds through the oracle's
init_params / synthetic_* helpers (tests/test_reference_callsite.py does that), so the file holds outputs only.

Cases:
  a   2 layers, 2 heads (kv 16), fp32, step 40 of a 100-step warm-up, cosine decay, clip 1.0
  b   3 layers, 4 heads (kv 12), fp32, recompute_grad (mtf.recompute_grad path, models.py:342-343), linear decay past the warm-up,
      weight decay 0.01 (exclude_from_weight_decay ["norm", "bias"], optimizers.py:82-89), clip 0.5
  c   case a with "bf_16": true -- every mtf op's output rounded to bfloat16 (master weights bf16): informational, compared with
      the oracle's coarser bf16 emulation under a loose bound.
  h   the exact `dalle_example` architecture of BASELINE.json (d 512, 6 layers, 4 heads, S = 256 + 1024, V = 50 771), B = 1, fp32,
      step 1500 of the 3000-step warm-up: a digest (loss, per-position loss, the logits of three positions and every position's
      arg-max / max, the norm of every gradient, the small gradients in full) in ref_callsite_dalle_headline.npz; 70 s and 14 GB
      to generate, so only regenerated with --headline
model_fn cases (ref_callsite_model_fns.npz): the reference's `vae_model_fn` and `dalle_model_fn` THEMSELVES are called
(src/model_fns_tf.py:9-114, src/model_fns.py:55-236; TPUEstimator / saver / hook objects are inert records):
  fv  vae_model_fn, TRAIN, global step 30 of a 100-step temperature anneal 1.0 -> 0.5, train_gumbel_hard false: loss, every
      gradient, the variables after tf.train.AdamOptimizer's first step
  fd  dalle_model_fn, TRAIN, global step 40: the VAE built from params["vae_params"], images tokenised (arg-max, + text_vocab_size,
      concat: :72-77,118-119), image_seq_len (:68), loss, the variables after the update ops, the next global step -- and the
      (empty) variable list the reference hands to tf.train.init_from_checkpoint, because it restores BEFORE it builds the VAE
VAE cases (16x16 images, two stride-2 stages with residual stacks, 32 codebook tokens; Gumbel uniforms injected):
  v1  hard Gumbel (straight-through), temperature 0.7
  v2  stack_factor 2 (space_to_depth / depth_to_space), soft Gumbel, temperature 1.0
  v3  v1 with "recompute_grad": true -- the reference's tf.custom_gradient / GradientTape recompute hack (src/vae_tf/models.py:8-43,
      100,148) executed; same numbers as v1 by construction, which the test asserts
  vh  the exact `vae_example` architecture (32x32 images, convblocks [[3,64],[3,128],[3,256]], 512 tokens), B = 2, hard Gumbel:
      encoder logits, reconstruction and loss in full, the norm of every gradient, the small gradients in full"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dalle_oracle as do  # noqa: E402
from oracle import vae_oracle as vo  # noqa: E402
from oracle.refshim import available, harness  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ref_callsite_dalle.npz")
OUT_VAE = os.path.join(HERE, "ref_callsite_vae.npz")
OUT_HEADLINE = os.path.join(HERE, "ref_callsite_dalle_headline.npz")
OUT_FNS = os.path.join(HERE, "ref_callsite_model_fns.npz")
HEADLINE = dict(hp=dict(n_embd=512, text_vocab_size=50258, image_vocab_size=512, text_seq_len=256, image_seq_len=1024, n_layers=6, n_heads=4,
                        bf_16=False, lr=1e-3, train_steps=100000, warmup_steps=3000, gradient_clipping=1.0),
                batch=1, step=1500, seeds=(1234, 1, 2))
HEADLINE_POSITIONS = (0, 255, 1279)


def headline_digest(loss, loss_batch, logits, grads, lr=None):
    """the compact form both sides are reduced to (the test applies it to the oracle's outputs)"""
    out = {"loss": np.float32(loss), "loss_batch": np.asarray(loss_batch, np.float32)}
    logits = np.asarray(logits)
    out["logits_rows"] = logits[:, list(HEADLINE_POSITIONS), :].astype(np.float32)
    out["logits_argmax"] = logits.argmax(-1).astype(np.int32)
    out["logits_max"] = logits.max(-1).astype(np.float32)
    out["grad_norms"] = np.array([np.linalg.norm(np.asarray(g, np.float64)) for g in grads.values()], np.float64)
    for k, g in grads.items():
        if g.size <= 2048:
            out["grad:" + k] = np.asarray(g, np.float32)
    if lr is not None:
        out["lr"] = np.float32(lr)
    return out


def run_headline():
    cfg, weights, tokens = case_inputs(HEADLINE)
    r = harness.run_dalle_step(HEADLINE["hp"], weights, tokens, global_step=HEADLINE["step"])
    return headline_digest(r["loss"], r["loss_batch"], r["logits"], r["grads"], r["lr"])

CASES = {
    "a": dict(hp=dict(n_embd=32, text_vocab_size=40, image_vocab_size=16, text_seq_len=6, image_seq_len=10, n_layers=2, n_heads=2,
                      bf_16=False, lr=3e-4, train_steps=1000, warmup_steps=100, gradient_clipping=1.0),
              batch=3, step=40, seeds=(7, 1, 2)),
    "b": dict(hp=dict(n_embd=48, text_vocab_size=50, image_vocab_size=12, text_seq_len=5, image_seq_len=9, n_layers=3, n_heads=4,
                      bf_16=False, lr=1e-3, train_steps=2000, warmup_steps=100, lr_decay="linear", gradient_clipping=0.5,
                      weight_decay=0.01, recompute_grad=True),
              batch=2, step=700, seeds=(11, 3, 4)),
    "c": dict(hp=dict(n_embd=32, text_vocab_size=40, image_vocab_size=16, text_seq_len=6, image_seq_len=10, n_layers=2, n_heads=2,
                      bf_16=True, lr=3e-4, train_steps=1000, warmup_steps=100, gradient_clipping=1.0),
              batch=3, step=40, seeds=(7, 1, 2)),
}


FULL = ("embedding/wte", "positional_embedding/wpe", "layer_0/norm_1/g", "layer_0/attn/q", "layer_{last}/attn/o",
        "layer_{last}/attn/compute_output_bias/o_b", "layer_0/mlp/mlp_linear_1/kernel", "layer_{last}/mlp/mlp_linear_2/bias",
        "to_logits/layer_norm/b", "to_logits/linear_out/kernel")


def case_inputs(case):
    """(cfg, weights, tokens) of a case -- from the oracle's own helpers, so that the test can rebuild them"""
    hp = case["hp"]
    cfg = do.DalleConfig(hp["n_embd"], hp["text_vocab_size"], hp["image_vocab_size"], hp["text_seq_len"], hp["image_seq_len"],
                         hp["n_layers"], hp["n_heads"])
    ws, ts, is_ = case["seeds"]
    weights = do.init_params(cfg, seed=ws, perturb=0.05)
    text = do.synthetic_captions(case["batch"], cfg.text_seq_len, cfg.text_vocab_size, seed=ts)
    img = do.synthetic_image_tokens(case["batch"], cfg.image_seq_len, cfg.image_vocab_size, seed=is_)
    return cfg, weights, do.assemble_tokens(text, img, cfg.text_vocab_size)


def run_case(case):
    """name -> array: what the reference computes for the case"""
    cfg, weights, tokens = case_inputs(case)
    r = harness.run_dalle_step(case["hp"], weights, tokens, global_step=case["step"])
    out = {"loss": r["loss"], "loss_batch": r["loss_batch"], "logits": r["logits"], "lr": r["lr"]}
    out["variables"] = np.array(json.dumps({k: list(v) for k, v in r["variables"].items()}))
    for k, g in r["grads"].items():
        out["grad:" + k] = g
    # the clip multiplier is one number: store what it did to every tensor as norms, and a few tensors in full; the updated
    # variables in full for the kinds that differ (decayed / not decayed, every scope), their Adam slots for the same few
    out["clipped_norms"] = np.array([np.linalg.norm(g.astype(np.float64)) for g in r["clipped_grads"].values()], np.float64)
    for k in FULL:
        k = k.format(last=case["hp"]["n_layers"] - 1)
        out["clipped:" + k] = r["clipped_grads"][k]
        for suffix in ("", "/adam_m", "/adam_v"):
            out["after:" + k + suffix] = r["updated"][k + suffix]
    out["after_norms"] = np.array([np.linalg.norm(r["updated"][k].astype(np.float64)) for k in r["grads"]], np.float64)
    return out


VAE_CASES = {
    "v1": dict(hp=dict(num_tokens=32, n_embd=64, hidden_dim=16, convblocks=[[2, 16], [3, 24]], stack_factor=1), size=16, batch=2,
               hard=True, temperature=0.7, seeds=(5, 3, 9)),
    "v2": dict(hp=dict(num_tokens=32, n_embd=64, hidden_dim=16, convblocks=[[2, 16], [3, 24]], stack_factor=2), size=16, batch=2,
               hard=False, temperature=1.0, seeds=(6, 4, 10)),
    "v3": dict(hp=dict(num_tokens=32, n_embd=64, hidden_dim=16, convblocks=[[2, 16], [3, 24]], stack_factor=1, recompute_grad=True),
               size=16, batch=2, hard=True, temperature=0.7, seeds=(5, 3, 9)),
}


VAE_HEADLINE = dict(hp=dict(num_tokens=512, n_embd=512, hidden_dim=64, convblocks=[[3, 64], [3, 128], [3, 256]], stack_factor=1), size=32,
                    batch=2, hard=True, temperature=1.0, seeds=(4321, 0, 7))


def vae_case_inputs(case):
    hp = case["hp"]
    cfg = vo.VaeConfig(hp["num_tokens"], case["size"], hp["convblocks"], stack_factor=hp["stack_factor"])
    ws, is_, us = case["seeds"]
    weights = vo.init_params(cfg, seed=ws, bias_perturb=0.05)
    img = vo.synthetic_images(case["batch"], case["size"], seed=is_)
    u = vo.synthetic_uniforms((case["batch"], cfg.grid, cfg.grid, cfg.num_tokens), seed=us)
    return cfg, weights, img, u


def run_vae_case(case):
    cfg, weights, img, u = vae_case_inputs(case)
    r = harness.run_vae_step(case["hp"], weights, img, u, hard_gumbel=case["hard"], temperature=case["temperature"])
    out = {"loss": r["loss"], "reconstruction": r["reconstruction"], "logits": r["logits"]}
    out["variables"] = np.array(json.dumps({k: list(v) for k, v in r["variables"].items()}))
    for k, g in r["grads"].items():
        out["grad:" + k] = g
    return out


def vae_digest(loss, reconstruction, logits, grads):
    out = {"loss": np.float32(loss), "reconstruction": np.asarray(reconstruction, np.float32), "logits": np.asarray(logits, np.float32)}
    out["grad_norms"] = np.array([np.linalg.norm(np.asarray(g, np.float64)) for g in grads.values()], np.float64)
    for k, g in grads.items():
        if g.size <= 4096:
            out["grad:" + k] = np.asarray(g, np.float32)
    return out


def run_vae_headline():
    case = VAE_HEADLINE
    cfg, weights, img, u = vae_case_inputs(case)
    r = harness.run_vae_step(case["hp"], weights, img, u, hard_gumbel=case["hard"], temperature=case["temperature"])
    return vae_digest(r["loss"], r["reconstruction"], r["logits"], r["grads"])


_FN_VAE = dict(num_tokens=32, dim=64, hidden_dim=16, convblocks=[[2, 16], [3, 24]], stack_factor=1, model_path="runs/vae")
FN_CASES = {
    "fv": dict(params=dict(num_tokens=32, n_embd=64, hidden_dim=16, convblocks=[[2, 16], [3, 24]], stack_factor=1, lr=1e-3,
                           dataset={"image_size": 16}, train_batch_size=2, eval_batch_size=2, temp_start=1.0, temp=0.5,
                           temp_anneal_steps=100, train_gumbel_hard=False, eval_gumbel_hard=True, model_path="runs/vae"),
               step=30, batch=2, seeds=(5, 3, 9)),
    "fd": dict(params=dict(n_embd=32, text_vocab_size=40, image_vocab_size=32, text_seq_len=6, n_layers=2, n_heads=2, bf_16=False, lr=3e-4,
                           train_steps=1000, warmup_steps=100, gradient_clipping=1.0, dataset={"image_size": 16}, train_batch_size=3,
                           eval_batch_size=3, vae_params=_FN_VAE, vae_checkpoint_path="runs/vae/model.ckpt-10", mesh_shape="data:1",
                           layout="batch_dim:data", use_tpu=False, gpu_ids=["device:CPU:0"], model_path="runs/dalle",
                           steps_per_checkpoint=100),
               step=40, batch=3, seeds=(5, 3, 7, 1)),
}


def fn_vae_inputs(case):
    p = case["params"]
    cfg = vo.VaeConfig(p["num_tokens"], p["dataset"]["image_size"], p["convblocks"], stack_factor=p["stack_factor"])
    ws, is_, us = case["seeds"]
    return (cfg, vo.init_params(cfg, seed=ws, bias_perturb=0.05), vo.synthetic_images(case["batch"], p["dataset"]["image_size"], seed=is_),
            vo.synthetic_uniforms((case["batch"], cfg.grid, cfg.grid, cfg.num_tokens), seed=us))


def fn_dalle_inputs(case):
    p = case["params"]
    vp = p["vae_params"]
    vcfg = vo.VaeConfig(vp["num_tokens"], p["dataset"]["image_size"], vp["convblocks"], stack_factor=vp["stack_factor"])
    vs, is_, ds, ts = case["seeds"]
    cfg = do.DalleConfig(p["n_embd"], p["text_vocab_size"], p["image_vocab_size"], p["text_seq_len"], vcfg.grid ** 2, p["n_layers"], p["n_heads"])
    return (vcfg, vo.init_params(vcfg, seed=vs, bias_perturb=0.05), cfg, do.init_params(cfg, seed=ds, perturb=0.05),
            vo.synthetic_images(case["batch"], p["dataset"]["image_size"], seed=is_),
            do.synthetic_captions(case["batch"], p["text_seq_len"], p["text_vocab_size"], seed=ts))


def run_fn_cases():
    out = {}
    c = FN_CASES["fv"]
    cfg, w, img, u = fn_vae_inputs(c)
    r = harness.run_vae_model_fn(c["params"], w, img, u, global_step=c["step"])
    out["fv/loss"], out["fv/reconstruction"], out["fv/t"] = r["loss"], r["reconstruction"], np.int32(r["t"])
    for k in r["grads"]:
        out["fv/grad:" + k], out["fv/after:" + k] = r["grads"][k], r["updated"][k]
    c = FN_CASES["fd"]
    vcfg, vw, cfg, dw, img, text = fn_dalle_inputs(c)
    r = harness.run_dalle_model_fn(c["params"], vw, dw, img, text, global_step=c["step"])
    out["fd/loss"], out["fd/tokens"], out["fd/next_global_step"] = r["loss"], r["tokens"], np.int32(r["next_global_step"])
    out["fd/restore_requests"] = np.array(json.dumps(r["restore_requests"]))
    for k, a in r["updated"].items():
        if not k.endswith(("/adam_m", "/adam_v")):
            out["fd/after:" + k] = a
    return out


def main():
    if not available():
        raise SystemExit("the reference checkout is not here (DALLE_REFERENCE_ROOT / /root/reference): nothing to execute")
    blob = {"cases": np.array(json.dumps(CASES))}
    for name, case in CASES.items():
        out = run_case(case)
        for k, a in out.items():
            blob[name + "/" + k] = a
        print("case %s: loss %.6f  lr %.6e  %d arrays" % (name, float(out["loss"]), float(out["lr"]), len(out)))
    np.savez_compressed(OUT, **blob)
    print(OUT, os.path.getsize(OUT), "bytes")
    blob = {"cases": np.array(json.dumps(VAE_CASES))}
    for name, case in VAE_CASES.items():
        out = run_vae_case(case)
        for k, a in out.items():
            blob[name + "/" + k] = a
        print("case %s: loss %.6f  %d arrays" % (name, float(out["loss"]), len(out)))
    out = run_vae_headline()
    for k, a in out.items():
        blob["vh/" + k] = a
    print("case vh: loss %.6f  %d arrays" % (float(out["loss"]), len(out)))
    np.savez_compressed(OUT_VAE, **blob)
    print(OUT_VAE, os.path.getsize(OUT_VAE), "bytes")
    out = run_fn_cases()
    np.savez_compressed(OUT_FNS, cases=np.array(json.dumps(FN_CASES)), **out)
    print("model_fn cases: vae loss %.6f, dalle loss %.6f" % (float(out["fv/loss"]), float(out["fd/loss"])), OUT_FNS, os.path.getsize(OUT_FNS), "bytes")
    if "--headline" in sys.argv:
        out = run_headline()
        np.savez_compressed(OUT_HEADLINE, case=np.array(json.dumps(HEADLINE)), **out)
        print("headline: loss %.6f  lr %.6e" % (float(out["loss"]), float(out["lr"])), OUT_HEADLINE, os.path.getsize(OUT_HEADLINE), "bytes")


if __name__ == "__main__":
    main()
