"""get_optimizer / clip_by_global_norm surface of src/optimizers.py re-hosted on the engine's fused
kernels: dmi_sumsq (global norm), dmi_adam_step (clip multiplier + Adam without bias correction,
AdamWeightDecayOptimizer semantics, src/optimizers.py:82-89,154-177).  Adafactor (optimizers.py:91-97)
is not selected by any shipped config and is out of scope."""


def clip_by_global_norm(engine, clip_norm):
    """Returns the device scalar ||g||^2; the multiplier clip/max(||g||, clip) (optimizers.py:11-16) is applied
    inside the Adam kernel so gradients are read once."""
    import dalle_hip as dh
    dh.sumsq(engine.g, engine.lay.total, engine.gnorm_sq, engine.ws)
    return engine.gnorm_sq


def get_optimizer(engine, params):
    """Returns (learning_rate_fn, update_op): update_op() = backward-complete -> all-reduce wait -> clip -> Adam."""
    name = (params.get("optimizer") or "adam").lower()
    if name != "adam":
        raise ValueError(f"{name} not recognized (only adam is built; adafactor is out of scope)")
    for k in ("lr", "train_steps", "lr_decay_end", "lr_decay", "warmup_steps", "gradient_clipping", "weight_decay",
              "beta_1", "beta_2", "epsilon"):
        if k in params and params[k] is not None:
            engine.hp[k] = params[k]
    if "gradient_clipping" not in engine.hp:
        engine.hp["gradient_clipping"] = 1.0

    def update_op():
        return engine.optimizer_step()
    return engine.learning_rate, update_op
