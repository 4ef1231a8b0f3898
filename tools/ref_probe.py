#!/usr/bin/env python
"""ref_probe.py -- pin the oracle against the REAL reference wherever it can run (SURVEY.md §8(c) pin (4)).

The reference (EleutherAI/DALLE-mtf) is Python on tensorflow==2.4.0 + mesh_tensorflow==0.1.18 (requirements.txt:1-2).  Neither
is installable in the authoring container or on the GPU box (no network, no Python-3.10 wheel for TF 2.4), so the oracle under
oracle/ is unpinned at the level of the third-party primitives (round 4 pins its CALL GRAPH by executing the reference's own files
over shims, oracle/refshim + tests/test_reference_callsite.py).  This script is the other half of that statement: on any machine where `import tensorflow, mesh_tensorflow` works AND a checkout of the reference is available
(DALLE_REFERENCE_ROOT, default /root/reference) it

  dump   builds the UNMODIFIED reference model class (src/dalle_mtf/models.py:141-416, `DALLE`) on `mesh_shape data:1`,
         `layout batch_dim:data`, device CPU:0 (PlacementMeshImpl, as src/model_fns.py:88-91 does off-TPU), assigns the
         oracle's initial weights (oracle.dalle_oracle.init_params, same seed as the golden generator) to the reference's
         variables by their TF names, and writes loss, loss_batch, logits and every variable's gradient
         (mtf.gradients, as src/optimizers.py:34) to tests/golden/ref_dalle_small.npz.  tests/test_golden.py then compares
         the ORACLE against that file (test_oracle_vs_reference_dump, skipped while the file does not exist) -- the moment the
         file exists the oracle is pinned to the reference, not to itself;
  time   times the same reference train graph (forward + gradients + the reference optimizer's update ops,
         src/optimizers.py:11-104) on the host cores for bench.py's cpu_baseline with kind = "reference";
  check  (default) only reports whether the above is possible here, as one JSON line, exit status 0 either way.

bench.py calls `probe()` (never reads the reference tree unless TF imports) and falls back to timing the oracle
(kind = "port").  Status in this repository: UNEXERCISED beyond `check` -- the TF branch has never run because no machine
available to the build has TensorFlow; it is written against the reference's call signatures and must be treated as a
best-effort recipe by whoever first runs it."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_ROOT = os.environ.get("DALLE_REFERENCE_ROOT", "/root/reference")
SMALL = dict(n_embd=256, text_vocab_size=300, image_vocab_size=64, text_seq_len=16, image_seq_len=112, n_layers=2, n_heads=2)
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_dalle_small.npz")


def probe():
    """{"available": bool, "reason": str}: can the reference itself be executed in this process environment?"""
    try:
        import tensorflow  # noqa: F401
    except Exception as e:  # ImportError, or a broken wheel
        return {"available": False, "reason": f"import tensorflow failed: {type(e).__name__}: {e}"}
    try:
        import mesh_tensorflow  # noqa: F401
    except Exception as e:
        return {"available": False, "reason": f"import mesh_tensorflow failed: {type(e).__name__}: {e}"}
    if not os.path.isfile(os.path.join(REF_ROOT, "src", "dalle_mtf", "models.py")):
        return {"available": False, "reason": f"no reference checkout at {REF_ROOT} (set DALLE_REFERENCE_ROOT)"}
    return {"available": True, "reason": "tensorflow + mesh_tensorflow import and the reference checkout is present"}


def _tf_name_to_oracle(name):
    """TF variable name ([MTF-RECALL] for the attention scope) -> oracle parameter name."""
    n = name.split(":")[0]
    for suffix in ("/adam_m", "/adam_v"):
        if n.endswith(suffix):
            return None
    return n


def _build(cfg, batch, hp, train):
    """the reference graph on one CPU device; returns (tf graph handles).  Follows src/model_fns.py:77-189."""
    import mesh_tensorflow as mtf
    import tensorflow.compat.v1 as tf
    sys.path.insert(0, REF_ROOT)
    from src.dalle_mtf.models import DALLE          # the reference's own class, unmodified
    from src.optimizers import get_optimizer
    tf.disable_v2_behavior()
    graph = mtf.Graph()
    mesh = mtf.Mesh(graph, "my_mesh", None)
    mesh_shape = mtf.convert_to_shape("data:1")
    layout_rules = mtf.convert_to_layout_rules("batch_dim:data")
    mesh_impl = mtf.placement_mesh_impl.PlacementMeshImpl(mesh_shape, layout_rules, ["device:CPU:0"])
    params = dict(hp, num_microbatches=1, bf_16=False)
    model = DALLE(n_embd=cfg["n_embd"], text_vocab_size=cfg["text_vocab_size"], image_vocab_size=cfg["image_vocab_size"],
                  text_seq_len=cfg["text_seq_len"], image_seq_len=cfg["image_seq_len"], n_layers=cfg["n_layers"],
                  n_heads=cfg["n_heads"], batch_size=batch, bf_16=False, mode="train", params=params)
    S = cfg["text_seq_len"] + cfg["image_seq_len"]
    tok_ph = tf.placeholder(tf.int32, [batch, S], name="tokens")
    shape = mtf.Shape([model.dimensions["batch_dim"], model.dimensions["total_seq_dim"]])
    feats = {"tokens": mtf.import_fully_replicated(mesh, tok_ph, shape, name="text_inputs")}
    loss, loss_batch, logits = model.forward(feats, return_loss=True, return_logits=True)
    tvars = graph.trainable_variables
    grads = mtf.gradients([loss], [v.outputs[0] for v in tvars])
    update_ops = None
    if train:
        tf.train.get_or_create_global_step()
        _, update_ops, _ = get_optimizer(mesh, loss, params, variable_dtype=model.variable_dtype)
    lowering = mtf.Lowering(graph, {mesh: mesh_impl}, autostack=False)
    out = dict(tok_ph=tok_ph, loss=lowering.export_to_tf_tensor(loss), loss_batch=lowering.export_to_tf_tensor(loss_batch),
               logits=lowering.export_to_tf_tensor(logits), tvars=tvars,
               grads=[None if g is None else lowering.export_to_tf_tensor(g) for g in grads],
               restore=lowering.copy_masters_to_slices())
    if train:
        out["train_op"] = tf.group([lowering.lowered_operation(op) for op in update_ops])
    return tf, out


def _inputs(cfg, batch, seed=0):
    sys.path.insert(0, ROOT)
    from oracle import dalle_oracle as do        # checker-side data generator only (tools/ is not the product path)
    ocfg = do.DalleConfig(cfg["n_embd"], cfg["text_vocab_size"], cfg["image_vocab_size"], cfg["text_seq_len"],
                          cfg["image_seq_len"], cfg["n_layers"], cfg["n_heads"])
    P = do.init_params(ocfg, seed=1234 + seed, perturb=0.05)
    text = do.synthetic_captions(batch, cfg["text_seq_len"], cfg["text_vocab_size"], seed=seed + 1)
    img = do.synthetic_image_tokens(batch, cfg["image_seq_len"], cfg["image_vocab_size"], seed=seed + 2)
    return P, do.assemble_tokens(text, img, cfg["text_vocab_size"])


def _assign(tf, sess, P):
    """oracle weights -> reference master variables, by name; every reference variable must be covered."""
    missing = []
    for v in tf.global_variables():
        key = _tf_name_to_oracle(v.name)
        if key is None or v.name.startswith("global_step"):
            continue
        if key not in P:
            missing.append(v.name)
            continue
        v.load(P[key].reshape(v.shape.as_list()).astype(v.dtype.as_numpy_dtype), sess)
    if missing:
        raise RuntimeError(f"reference variables without an oracle counterpart (fix the Appendix-B name map): {missing}")


def dump(path=GOLDEN, batch=2):
    import numpy as np
    cfg = SMALL
    hp = dict(lr=1e-3, train_steps=1000, warmup_steps=2, gradient_clipping=1.0)
    tf, g = _build(cfg, batch, hp, train=False)
    P, tokens = _inputs(cfg, batch)
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        _assign(tf, sess, P)
        sess.run(g["restore"])
        fetch = [g["loss"], g["loss_batch"], g["logits"]] + [x for x in g["grads"] if x is not None]
        vals = sess.run(fetch, {g["tok_ph"]: tokens})
    out = {"tokens": tokens, "loss": vals[0], "loss_batch": vals[1], "logits": vals[2]}
    names = [v.name for v, x in zip(g["tvars"], g["grads"]) if x is not None]
    for n, a in zip(names, vals[3:]):
        out["grad/" + _tf_name_to_oracle(n)] = a
    np.savez_compressed(path, **out)
    return {"written": path, "loss": float(vals[0]), "tensors": len(names)}


def time_reference(cfg=None, batch=1, budget_s=20.0):
    """train steps/s of the reference graph on this host (all cores TF gives it)."""
    cfg = cfg or dict(n_embd=512, text_vocab_size=50258, image_vocab_size=512, text_seq_len=256, image_seq_len=1024,
                      n_layers=6, n_heads=4)
    hp = dict(lr=1e-3, train_steps=100000, warmup_steps=3000, gradient_clipping=1.0)
    tf, g = _build(cfg, batch, hp, train=True)
    _, tokens = _inputs(cfg, batch)
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        sess.run(g["restore"])
        sess.run([g["loss"], g["train_op"]], {g["tok_ph"]: tokens})     # warm-up (graph optimisation, allocator)
        t0, n = time.time(), 0
        while time.time() - t0 < budget_s and n < 8:
            sess.run([g["loss"], g["train_op"]], {g["tok_ph"]: tokens})
            n += 1
        dt = (time.time() - t0) / max(n, 1)
    S = cfg["text_seq_len"] + cfg["image_seq_len"]
    return {"value": batch * S / dt, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "reference",
            "sample": f"{n} train steps of B={batch} x S={S} of the unmodified mesh-tensorflow reference on device:CPU:0, {dt:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("cmd", nargs="?", default="check", choices=["check", "dump", "time"])
    a = ap.parse_args()
    st = probe()
    if a.cmd == "check" or not st["available"]:
        print(json.dumps(st))
        return 0
    print(json.dumps(dump() if a.cmd == "dump" else time_reference()))
    return 0


if __name__ == "__main__":
    sys.exit(main())
