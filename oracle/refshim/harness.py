"""harness -- drives the reference's DALL-E train step over the shims the way src/model_fns.py drives it.

TEST INFRASTRUCTURE ONLY.  `run_dalle_step` follows the reference's own call sequence:
  src/model_fns.py:96-108    DALLE(n_embd=..., text_vocab_size=..., ..., batch_size=..., bf_16=..., mode=..., params=params)
  src/model_fns.py:80-94     graph = mtf.Graph(); mesh = mtf.Mesh(graph, "my_mesh")
  src/model_fns.py:118-123   tokens -> mtf.import_fully_replicated(mesh, x, Shape([batch_dim, total_seq_dim]), name=key)
  src/model_fns.py:166       loss, loss_batch = model.forward(mtf_features, return_loss=True)
  src/model_fns.py:181       _, update_ops, var_grads = get_optimizer(mesh, loss, params, variable_dtype=model.variable_dtype)
and returns numpy arrays keyed by the reference's variable names."""
from collections import OrderedDict, defaultdict

import numpy as np
import torch

from . import installed, mtfshim, reference_module, tfshim


def run_dalle_step(hparams, weights, tokens, global_step=0, root=None, with_optimizer=True, return_logits=True):
    """hparams: the config keys the reference reads (n_embd, text_vocab_size, image_vocab_size, text_seq_len, n_layers, n_heads,
    bf_16, lr, train_steps, warmup_steps, gradient_clipping, ...) + image_seq_len; weights: name -> array for every
    non-constant-initialised variable; tokens: int [B, text_seq_len + image_seq_len] (text ids | image ids + text_vocab_size,
    as src/model_fns.py:119 assembles them)."""
    kw = {} if root is None else {"root": root}
    with installed(**kw):
        models = reference_module("dalle_mtf.models")
        params = defaultdict(lambda: None, dict(hparams))          # src/utils/utils.py:13-17
        params.setdefault("num_microbatches", 1)                     # src/model_fns.py:153
        tokens = np.asarray(tokens)
        batch_size = tokens.shape[0]
        mtfshim.inject_variables({k: np.asarray(v) for k, v in weights.items()})
        tfshim.set_global_step(global_step)
        model = models.DALLE(
            n_embd=params["n_embd"],
            text_vocab_size=params["text_vocab_size"],
            image_vocab_size=params["image_vocab_size"],
            text_seq_len=params["text_seq_len"],
            image_seq_len=params["image_seq_len"],
            n_layers=params["n_layers"],
            n_heads=params["n_heads"],
            batch_size=batch_size,
            bf_16=params["bf_16"],
            mode="train",
            params=params,
        )
        graph = mtfshim.Graph()
        mesh = mtfshim.Mesh(graph, "my_mesh")
        mtf_shape = mtfshim.Shape([model.dimensions["batch_dim"], model.dimensions["total_seq_dim"]])
        features = {"tokens": mtfshim.import_fully_replicated(mesh, torch.as_tensor(tokens.astype(np.int32)), mtf_shape, name="text_inputs")}
        if return_logits:
            loss, loss_batch, logits = model.forward(features, return_loss=True, return_logits=True)
        else:
            (loss, loss_batch), logits = model.forward(features, return_loss=True), None
        out = OrderedDict()
        out["loss"] = loss.value.detach().numpy().copy()
        lb_dims = loss_batch.shape.dimension_names
        lb = loss_batch.value.detach()
        if lb_dims != ["batch_dim", "total_seq_dim"]:            # the label gather leaves [total_seq_dim, batch_dim]
            lb = lb.permute([lb_dims.index("batch_dim"), lb_dims.index("total_seq_dim")])
        out["loss_batch"] = lb.numpy().copy()
        if logits is not None:
            assert logits.shape.dimension_names == ["batch_dim", "total_seq_dim", "vocab_dim"], logits.shape
            out["logits"] = logits.value.detach().numpy().copy()
        variables = graph.trainable_variables
        out["variables"] = OrderedDict((v.name, (tuple(v.shape.to_integer_list), v.initializer.kind,
                                                 float(getattr(v.initializer, "stddev", getattr(v.initializer, "value", 0.0)))))
                                       for v in variables)
        if with_optimizer:
            optimizers = reference_module("optimizers")
            lr, update_ops, var_grads = optimizers.get_optimizer(mesh, loss, params, variable_dtype=model.variable_dtype)
            out["lr"] = lr.value.detach().numpy().copy()
            # var_grads: the CLIPPED fp32 gradients in trainable-variable order (src/optimizers.py:44,100-103)
            out["clipped_grads"] = OrderedDict((v.name, g.value.detach().numpy().copy()) for v, g in zip(variables, var_grads))
            new = OrderedDict()
            for op in update_ops:
                new[op.variable.name] = op.new_value.numpy().copy()
            out["updated"] = new
        raw = mtfshim.gradients([loss], [v.outputs[0] for v in variables])
        out["grads"] = OrderedDict((v.name, g.value.detach().numpy().copy()) for v, g in zip(variables, raw))
        return out


def run_vae_step(hparams, weights, images, uniforms, hard_gumbel=True, temperature=1.0, root=None):
    """The reference's discrete VAE (src/vae_tf/models.py:46-184, layers.py) as src/model_fns_tf.py:16-56 builds and calls it:
    DiscreteVAE(num_tokens=..., dim=..., hidden_dim=..., input_channels=..., convblocks=..., recompute_grad=..., use_bf16=...,
    stack_factor=..., dimensions=H) and, under tf.variable_scope("vae"), model.forward(features, return_recon_loss=True,
    temperature=temp, hard_gumbel=gumbel).  weights: oracle names (without the "vae/" scope); uniforms: the Gumbel noise source
    tf.random_uniform would draw.  Returns loss, reconstruction, encoder logits, the gradient of every variable."""
    kw = {} if root is None else {"root": root}
    with installed(**kw):
        vae_models = reference_module("vae_tf.models")
        params = defaultdict(lambda: None, dict(hparams))
        images = torch.as_tensor(np.asarray(images, dtype=np.float32))
        tfshim.inject_variables({"vae/" + k: np.asarray(v) for k, v in weights.items()}, uniforms=[np.asarray(uniforms)])
        model = vae_models.DiscreteVAE(
            num_tokens=params["num_tokens"],
            dim=params["n_embd"],
            hidden_dim=params["hidden_dim"],
            input_channels=params.get("input_channels", 3),
            convblocks=params.get("convblocks", [(3, 64), (3, 128), (3, 256)]),
            recompute_grad=params.get("recompute_grad", False),
            use_bf16=params.get("use_bf16", False),
            stack_factor=params.get("stack_factor", 1),
            dimensions=images.shape[1],
        )
        with tfshim.variable_scope("vae"):
            loss, reconstruction = model.forward(images, return_recon_loss=True, temperature=temperature, hard_gumbel=hard_gumbel)
        variables = tfshim.created_variables()
        with tfshim.variable_scope("vae"):          # the reuse path of src/model_fns.py:72-77 (tokenising the images)
            logits = model.forward(images, return_logits=True)
        assert list(tfshim.created_variables()) == list(variables), "the second forward created variables"
        gs = torch.autograd.grad(loss, list(variables.values()), allow_unused=True)
        out = OrderedDict(loss=loss.detach().numpy().copy(), reconstruction=reconstruction.detach().numpy().copy(),
                          logits=logits.detach().numpy().copy())
        out["variables"] = OrderedDict((k[len("vae/"):], tuple(v.shape)) for k, v in variables.items())
        out["grads"] = OrderedDict((k[len("vae/"):], (g if g is not None else torch.zeros_like(v)).numpy().copy())
                                   for (k, v), g in zip(variables.items(), gs))
        return out


def run_vae_model_fn(params, weights, images, uniforms, global_step=0, adam_state=None, mode="train", root=None):
    """Calls the reference's `vae_model_fn(features, labels, mode, params)` (src/model_fns_tf.py:9-114) itself: it builds the VAE
    from the config keys, picks the Gumbel flavour for the mode, anneals the temperature from the global step, runs the forward
    under scope "vae" and hands the loss to tf.train.AdamOptimizer / CrossShardOptimizer.  Returns the loss, the temperature-
    dependent reconstruction digest and what the optimizer's minimize() produced (gradients, Adam slots, updated variables)."""
    kw = {} if root is None else {"root": root}
    with installed(**kw):
        fns = reference_module("model_fns_tf")
        p = defaultdict(lambda: None, dict(params))
        tfshim.inject_variables({"vae/" + k: np.asarray(v) for k, v in weights.items()}, uniforms=[np.asarray(uniforms)])
        tfshim.set_global_step(global_step)
        if adam_state:
            tfshim.adam_state.update(adam_state)
        feats = torch.as_tensor(np.asarray(images, dtype=np.float32))
        spec = fns.vae_model_fn(feats, feats, {"train": tfshim.estimator.ModeKeys.TRAIN, "eval": tfshim.estimator.ModeKeys.EVAL}[mode], p)
        out = OrderedDict(loss=spec.loss.detach().numpy().copy())
        reconstruction = spec.eval_metrics[1][3]
        out["reconstruction"] = reconstruction.detach().numpy().copy()
        strip = lambda d: OrderedDict((k[len("vae/"):], v.detach().numpy().copy()) for k, v in d.items())   # noqa: E731
        out["grads"], out["updated"] = strip(spec.train_op["grads"]), strip(spec.train_op["updated"])
        out["m"], out["v"], out["t"] = strip(spec.train_op["m"]), strip(spec.train_op["v"]), spec.train_op["t"]
        return out


def run_dalle_model_fn(params, vae_weights, dalle_weights, images, text, global_step=0, mode="train", root=None):
    """Calls the reference's `dalle_model_fn(features, labels, mode, params)` (src/model_fns.py:55-236) itself -- features = images,
    labels = caption ids: it builds the VAE from params["vae_params"], tokenises the images (arg-max of the encoder logits, reshape,
    + text_vocab_size, concat with the text: :72-77,118-119), derives image_seq_len (:68), builds the mtf graph / mesh, the DALLE
    model, the loss, the optimizer and the update ops.  Returns the loss, the tokens it assembled, the updated variables and what
    it asked tf.train.init_from_checkpoint to restore."""
    kw = {} if root is None else {"root": root}
    with installed(**kw):
        fns = reference_module("model_fns")
        p = defaultdict(lambda: None, dict(params))
        tfshim.inject_variables({"vae/" + k: np.asarray(v) for k, v in vae_weights.items()})
        mtfshim.inject_variables({k: np.asarray(v) for k, v in dalle_weights.items()})
        tfshim.set_global_step(global_step)
        captured = {}
        real_import = mtfshim.import_fully_replicated

        def spy_import(mesh, tf_tensor, shape, name=None):      # the tokens the model_fn assembled, as it hands them to mtf
            captured[name] = torch.as_tensor(tf_tensor).clone()
            return real_import(mesh, tf_tensor, shape, name=name)

        import sys
        sys.modules["mesh_tensorflow"].import_fully_replicated = spy_import
        spec = fns.dalle_model_fn(torch.as_tensor(np.asarray(images, dtype=np.float32)), torch.as_tensor(np.asarray(text, dtype=np.int32)),
                                  {"train": tfshim.estimator.ModeKeys.TRAIN, "eval": tfshim.estimator.ModeKeys.EVAL}[mode], p)
        out = OrderedDict(loss=torch.as_tensor(spec.loss).detach().numpy().copy())
        out["tokens"] = captured["text_inputs"].numpy().copy()
        out["restore_requests"] = list(tfshim.restore_requests)
        if spec.train_op is not None:
            upd = OrderedDict()
            for op in spec.train_op:
                if hasattr(op, "variable"):
                    upd[op.variable.name] = op.new_value.numpy().copy()
                else:
                    out["next_global_step"] = int(op.new_value)
            out["updated"] = upd
        return out
