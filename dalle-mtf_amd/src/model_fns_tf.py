"""vae_model_fn -- same (features, labels, mode, params) contract as the reference (src/model_fns_tf.py:9-114):
builds the DiscreteVAE, anneals the Gumbel temperature (:40-45), forward with reconstruction loss (:48-56),
tf.train.AdamOptimizer(lr) semantics + mean over replicas (CrossShardOptimizer, :58-66), loss/image summaries
(:68-107).  predict raises NotImplementedError as upstream (:30-31)."""
import torch

from .estimator import CheckpointSaverHook, EstimatorSpec, latest_checkpoint
from .utils import ModeKeys, SummaryWriter, mode_to_str
from .vae_tf import DiscreteVAE


def _dist_info():
    from .dp import dist_setup
    return dist_setup()    # (world, rank, process group, RCCL communicator behind the C ABI or None)


def temperature_schedule(step, params):
    """reference model_fns_tf.py:40-45."""
    if params.get("temp_anneal_steps", None):
        warmup_frac = min(float(step) / params["temp_anneal_steps"], 1.0)
        return params["temp_start"] - warmup_frac * (params["temp_start"] - params["temp"])
    t = params.get("temp")
    return 1.0 if t is None else t


def _build(params, mode_str):
    world, rank, pg, comm = _dist_info()
    H = params["dataset"]["image_size"]
    gbs = params[f"{mode_str}_batch_size"]
    assert gbs % world == 0
    n_channels = params.get("input_channels") or 3
    model = DiscreteVAE(
        num_tokens=params["num_tokens"],
        dim=params["n_embd"],              # None in the VAE configs; unused by the tf VAE
        hidden_dim=params["hidden_dim"],
        input_channels=n_channels,
        convblocks=params.get("convblocks") or [(3, 64), (3, 128), (3, 256)],
        recompute_grad=params.get("recompute_grad") or False,
        use_bf16=params.get("use_bf16") or False,
        stack_factor=params.get("stack_factor") or 1,
        dimensions=H, batch_size=gbs // world, mode=mode_str, process_group=pg, world_size=world, comm=comm)
    ck = latest_checkpoint(params["model_path"]) if params.get("model_path") else None
    if ck is not None:
        model.load_state_dict(torch.load(ck, map_location="cpu"))
    else:
        model.init_params(seed=params.get("seed") or 4321)
    if world > 1:
        import torch.distributed as dist
        for buf in (model.p, model.m, model.v):
            model.reducer.broadcast(buf, root=0)
        box = [model.global_step]
        dist.broadcast_object_list(box, src=0, group=pg)
        model.global_step = int(box[0])
        model.refresh_compute_copies(cast=True)
    saver = CheckpointSaverHook(params.get("model_path"), params.get("steps_per_checkpoint"), model.state_dict,
                                max_to_keep=params.get("max_checkpoints") or 5, is_chief=(rank == 0))
    writer = SummaryWriter(params["model_path"]) if (params.get("model_path") and rank == 0) else None
    return dict(model=model, saver=saver, writer=writer, rank=rank)


def vae_model_fn(features, labels, mode, params):
    mode_str = mode_to_str(mode)
    if mode == ModeKeys.PREDICT:
        raise NotImplementedError
    assert mode in (ModeKeys.TRAIN, ModeKeys.EVAL)
    key = f"_vae_state_{mode_str}"
    if params.get(key) is None:
        if mode == ModeKeys.EVAL and params.get("_vae_state_train") is not None and \
                params["eval_batch_size"] == params["train_batch_size"]:
            params[key] = params["_vae_state_train"]
        else:
            params[key] = _build(params, mode_str)
    st = params[key]
    model = st["model"]
    model.mode = mode_str
    if mode == ModeKeys.EVAL and st is not params.get("_vae_state_train"):
        tr = params.get("_vae_state_train")       # a separately sized eval model scores the CURRENT weights
        if tr is not None:
            if st.get("_synced_step") != tr["model"].global_step:
                model.p.copy_(tr["model"].p)
                model.global_step = tr["model"].global_step
                model.refresh_compute_copies(cast=True)
                st["_synced_step"] = tr["model"].global_step
        else:
            ck = latest_checkpoint(params["model_path"]) if params.get("model_path") else None
            if ck is not None and st.get("_synced_ckpt") != ck:
                model.load_state_dict(torch.load(ck, map_location="cpu"))
                st["_synced_ckpt"] = ck
    train_gumbel = params.get("train_gumbel_hard")
    eval_gumbel = params.get("eval_gumbel_hard")
    train_gumbel = True if train_gumbel is None else train_gumbel
    eval_gumbel = True if eval_gumbel is None else eval_gumbel
    gumbel = train_gumbel if mode == ModeKeys.TRAIN else eval_gumbel
    temp = temperature_schedule(model.global_step, params)
    denormalize = lambda x: (x + 1) / 2          # reference model_fns_tf.py:71,83
    imgs = features["inputs"] if isinstance(features, dict) else features
    if mode == ModeKeys.EVAL:
        loss, reconstruction = model.forward(features, return_recon_loss=True, temperature=temp, hard_gumbel=gumbel, need_grad=False)
        if st["writer"] is not None and float(loss) < 1e-9:      # the reference's record_if(loss < 1e-9) gate (:88)
            st["writer"].images(model.global_step, "eval/input_image", denormalize(imgs))
            st["writer"].images(model.global_step, "eval/reconstruction_image", denormalize(reconstruction))
        return EstimatorSpec(mode=mode, loss=loss, eval_metrics={"_loss": loss})

    # TRAIN: forward + backward + optimizer run inside train_op as ONE step (DiscreteVAE.train_step: a replayed HIP graph on a
    # single GPU); spec.loss is the device scalar that step fills in -- the loss of this batch before the update, as upstream.
    def train_op():
        model.train_step(features, params["lr"], hard_gumbel=gumbel, temperature=temp)
        return model.global_step

    host_call = None
    if st["writer"] is not None:      # rank 0 only: loss (+ temperature), input and reconstruction images (reference :68-78)
        def host_call_fn(step, loss, temperature, input):
            st["writer"].scalars(step, loss=loss, temperature=temperature)
            st["writer"].images(step, "input_image", denormalize(input))
            st["writer"].images(step, "reconstruction_image", denormalize(model.reconstruction()))   # of the step just run
        host_call = (host_call_fn, {"loss": model.loss[0], "temperature": temp, "input": imgs})
    return EstimatorSpec(mode=mode, loss=model.loss[0], train_op=train_op, host_call=host_call, training_hooks=[st["saver"]])
