"""dalle_model_fn -- same (features, labels, mode, params) contract as the reference (src/model_fns.py:55-236):
features = images [B,H,W,C] fp32 in [-1,1], labels = caption ids [B,text_seq_len] int32.
VAE-encode -> argmax tokens -> concat with text (+text_vocab_size) -> DALLE forward -> (TRAIN) backward,
gradient all-reduce over the data axis, clip + Adam -> EstimatorSpec(loss, train_op, host_call, hooks).

The TF graph/session split becomes: the model is built once and cached on `params`; each call runs the
forward eagerly and returns a spec whose train_op() runs backward + optimizer and returns the new
global step.  predict raises NotImplementedError exactly like the reference (model_fns.py:135-136)."""
import os

import torch

import dalle_hip as dh
from .dalle_mtf import DALLE
from .estimator import CheckpointSaverHook, EstimatorSpec, latest_checkpoint
from .optimizers import get_optimizer
from .utils import ModeKeys, create_host_call, get_graph_info, mode_to_str, parse_mesh_shape, scalar_summary


def _dist_info():
    from .dp import dist_setup
    return dist_setup()    # (world, rank, process group, RCCL communicator behind the C ABI or None)


def load_vae_model(params, mode_str):
    """reference model_fns.py:35-52: builds the DiscreteVAE from params['vae_params'] and resolves the checkpoint
    (explicit vae_checkpoint_path or the latest under the VAE's model_path)."""
    from .vae_tf import DiscreteVAE
    vae_checkpoint_path = params.get("vae_checkpoint_path")
    vae_params = params.get("vae_params")
    assert vae_params is not None, "vae model config must be supplied"
    if vae_checkpoint_path is None:
        vae_checkpoint_path = latest_checkpoint(vae_params["model_path"])
    if vae_checkpoint_path is None and vae_params.get("model_path"):      # a run directory written by the reference's tf.train.Saver
        from .data.tf_checkpoint import latest_tf_checkpoint
        vae_checkpoint_path = latest_tf_checkpoint(vae_params["model_path"])
    if vae_checkpoint_path is None and not params.get("allow_random_vae"):
        raise AssertionError("pretrained vae needed for training")
    D = params["dataset"]["image_size"]
    vae_model = DiscreteVAE(
        num_tokens=vae_params["num_tokens"],
        dim=vae_params.get("dim"),
        hidden_dim=vae_params.get("hidden_dim"),
        input_channels=vae_params.get("input_channels") or 3,
        convblocks=vae_params.get("convblocks") or [(3, 64), (3, 128), (3, 256)],
        stack_factor=vae_params.get("stack_factor") or 1,
        dimensions=D,
        batch_size=params.get("_tokenizer_batch") or params[f"{mode_str}_batch_size"] // max(_dist_info()[0], 1),
        mode="eval")
    # the reference does not forward use_bf16 here, so its tokenising encoder is fp32: same here unless switched off
    vae_model.fp32_tokens = bool(params.get("vae_tokens_fp32", True))
    return vae_model, vae_checkpoint_path


def initialize_vae_weights(vae, checkpoint_path):
    """reference model_fns.py:11-32: restore every variable under scope 'vae' by name."""
    if checkpoint_path is None:
        vae.init_params()
        return
    if os.path.exists(str(checkpoint_path) + ".index"):
        # a TensorFlow checkpoint written by the reference itself (tf.train.Saver, variables under scope "vae/"): restore by
        # name exactly as reference model_fns.py:11-32 does -- read without TensorFlow by src/data/tf_checkpoint.py
        from .data.tf_checkpoint import load_model_variables
        P = load_model_variables(str(checkpoint_path), scope="vae/") or load_model_variables(str(checkpoint_path))
        vae.load_reference_params(P)
        return
    sd = torch.load(checkpoint_path, map_location="cpu")
    vae.load_reference_params({k: (v.numpy() if torch.is_tensor(v) else v) for k, v in sd["vae_variables"].items()})


def serialize_num_microbatches(batch_per_replica, sequence_length, tokens_per_microbatch_per_replica=None):
    """mtf.transformer.utils.serialize_num_microbatches (mesh_tensorflow 0.1.18) for the batch_dim:data layout:
    sequences per micro-batch = max(1, tokens_per_mb // seq_len); count = batch_per_replica // that."""
    if not tokens_per_microbatch_per_replica:
        return 1
    microbatch_size = max(1, int(tokens_per_microbatch_per_replica) // int(sequence_length))
    return max(1, batch_per_replica // microbatch_size)


def _build(params, mode_str):
    world, rank, pg, comm = _dist_info()
    mesh = parse_mesh_shape(params.get("mesh_shape"))
    gbs = params[f"{mode_str}_batch_size"]
    assert gbs % world == 0, f"{mode_str}_batch_size {gbs} must divide over {world} data-parallel ranks"
    local_bs = gbs // world
    state = {"world": world, "rank": rank}
    if params.get("synthetic_image_tokens"):
        image_seq_len = int(params["synthetic_image_tokens"])
    else:
        vp = params["vae_params"]
        cb = vp.get("convblocks") or [(3, 64), (3, 128), (3, 256)]
        # reference model_fns.py:68 (from the VAE's configuration: the micro-batch count below sizes the tokenizer)
        image_seq_len = (params["dataset"]["image_size"] // (2 ** len(cb))) ** 2 // ((vp.get("stack_factor") or 1) ** 2)
    nmb = 1
    if mode_str == "train":
        nmb = serialize_num_microbatches(local_bs, params["text_seq_len"] + image_seq_len,
                                         params.get("tokens_per_mb_per_replica"))
        assert local_bs % nmb == 0, f"per-replica batch {local_bs} does not split into {nmb} micro-batches"
    if params.get("synthetic_image_tokens"):
        state["vae"] = None
    else:
        # the tokenizer is sized for ONE ENGINE batch (a micro-batch in training): dalle_model_fn feeds it engine-sized
        # chunks, so any row count the engine accepts (train: nmb x engine batch; eval: any multiple) also tokenizes
        params["_tokenizer_batch"] = local_bs // nmb
        vae, ckpt = load_vae_model(params, mode_str)
        initialize_vae_weights(vae, ckpt)
        assert image_seq_len == (vae.H // (2 ** len(vae.convblocks))) ** 2 // (vae.stack_factor ** 2)
        state["vae"] = vae
    model = DALLE(n_embd=params["n_embd"], text_vocab_size=params["text_vocab_size"],
                  image_vocab_size=params["image_vocab_size"], text_seq_len=params["text_seq_len"],
                  image_seq_len=image_seq_len, n_layers=params["n_layers"], n_heads=params["n_heads"],
                  batch_size=local_bs // nmb, bf_16=params["bf_16"], mode=mode_str, params=params,
                  process_group=pg, world_size=world, global_batch_size=gbs // nmb, comm=comm)
    eng = model.engine
    eng.hp["num_microbatches"] = nmb   # reference model_fns.py:141-154 (1 when tokens_per_mb_per_replica is unset)
    params["num_microbatches"] = nmb
    state["local_bs"] = local_bs
    ck = latest_checkpoint(params["model_path"]) if params.get("model_path") else None
    if ck is not None:
        eng.load_state_dict(torch.load(ck, map_location="cpu")["dalle"])
    elif params.get("tf_checkpoint"):
        # warm start from a checkpoint of the reference (`<prefix>.index` + `.data-*`, variable names of SURVEY Appendix B)
        from .data.tf_checkpoint import global_step_of, load_model_variables
        eng.load_reference_params(load_model_variables(params["tf_checkpoint"]))
        eng.global_step = global_step_of(params["tf_checkpoint"])
    else:
        eng.init_params(seed=params.get("seed") or 1234)
    if world > 1:
        # every rank starts from rank 0's weights, Adam state AND step (the LR schedule and the loop length depend on it)
        import torch.distributed as dist
        for buf in (eng.p, eng.m, eng.v):
            eng.reducer.broadcast(buf, root=0)
        box = [eng.global_step]
        dist.broadcast_object_list(box, src=0, group=pg)
        eng.global_step = int(box[0])
        eng.refresh_compute_copies(cast=True)
    lr_fn, update_op = get_optimizer(eng, params)
    if rank == 0:
        get_graph_info(model.variables())
    saver = CheckpointSaverHook(params.get("model_path"), params.get("steps_per_checkpoint"),
                                lambda: {"dalle": eng.state_dict()}, max_to_keep=params.get("max_checkpoints") or 5,
                                is_chief=(rank == 0))
    state.update(model=model, lr_fn=lr_fn, update_op=update_op, saver=saver, image_seq_len=image_seq_len)
    return state


def dalle_model_fn(features, labels, mode, params):
    mode_str = mode_to_str(mode)
    if mode == ModeKeys.PREDICT:
        raise NotImplementedError
    assert mode in (ModeKeys.TRAIN, ModeKeys.EVAL)
    key = f"_dalle_state_{mode_str}"
    if params.get(key) is None:
        tr = params.get("_dalle_state_train")
        if mode == ModeKeys.EVAL and tr is not None and \
                (params["eval_batch_size"] // max(tr["world"], 1)) % tr["model"].engine.B == 0:
            params[key] = tr       # evaluated in engine-sized chunks below: no second engine, never stale
        else:
            params[key] = _build(params, mode_str)
    st = params[key]
    model, eng = st["model"], st["model"].engine
    if mode == ModeKeys.EVAL and st is not params.get("_dalle_state_train"):
        # a separately built eval engine scores the CURRENT weights: copied from the train engine when there is one, reloaded
        # from the newest checkpoint otherwise (CheckpointSaverHook.save ends with a barrier, so the file is complete)
        tr = params.get("_dalle_state_train")
        if tr is not None:
            if st.get("_synced_step") != tr["model"].engine.global_step:
                eng.p.copy_(tr["model"].engine.p)
                eng.refresh_compute_copies(cast=True)
                st["_synced_step"] = tr["model"].engine.global_step
        else:
            ck = latest_checkpoint(params["model_path"]) if params.get("model_path") else None
            if ck is not None and st.get("_synced_ckpt") != ck:
                eng.load_state_dict(torch.load(ck, map_location="cpu")["dalle"])
                st["_synced_ckpt"] = ck
    model.mode = mode_str
    dev = eng.dev
    nmb = eng.hp.get("num_microbatches", 1) if mode == ModeKeys.TRAIN else 1
    T, P = eng.T, st["image_seq_len"]
    B = labels.numel() // T   # rows handed in: n_microbatches x engine batch in TRAIN; any multiple of it in EVAL
    assert B % eng.B == 0 and (mode != ModeKeys.TRAIN or B == eng.B * nmb), \
        f"got {B} rows for an engine batch of {eng.B} (x {nmb} micro-batches)"
    text = labels.to(device=dev, dtype=torch.int32).reshape(B, T)
    tokens = torch.empty(B, T + P, dtype=torch.int32, device=dev)
    if st["vae"] is not None:
        # tokens = argmax(vae_logits, -1) (model_fns.py:72-77); + text_vocab_size; concat (model_fns.py:118-119)
        vae = st["vae"]
        feats = (features["inputs"] if isinstance(features, dict) else features).to(dev)
        assert B % vae.B == 0, f"{B} images for a tokenizer batch of {vae.B}"
        for c0 in range(0, B, vae.B):   # the tokenizer is sized for one engine batch
            vae_logits = vae.forward(feats[c0:c0 + vae.B], return_logits=True)        # [b,g,g,T] fp32
            dh.assemble_tokens(text[c0:c0 + vae.B], vae_logits.contiguous(), tokens[c0:c0 + vae.B], vae.B, T, P,
                               vae_logits.shape[-1], eng.text_vocab_size)
    else:
        img = features["image_tokens"] if isinstance(features, dict) else features
        tokens[:, :T] = text
        tokens[:, T:] = img.to(device=dev, dtype=torch.int32).reshape(B, P) + eng.text_vocab_size
    if nmb > 1:
        # serialized training step (model_fns.py:156-166): forward/backward per micro-batch inside train_op, gradients
        # accumulated locally and reduced once; spec.loss is filled in by train_op
        if getattr(eng, "loss_acc", None) is None:
            eng.loss_acc = torch.zeros_like(eng.loss)

        def train_op_mb():
            eng.train_step(tokens)
            return eng.global_step
        scalar_summary("loss", eng.loss_acc[0])
        scalar_summary("lr", st["lr_fn"]())
        return EstimatorSpec(mode=mode, loss=eng.loss_acc[0], train_op=train_op_mb,
                             host_call=create_host_call(params["model_path"]) if (params.get("model_path") and st["rank"] == 0) else None,
                             training_hooks=[st["saver"]])
    if mode == ModeKeys.EVAL:
        # the engine may have been sized for one training micro-batch: evaluate the batch in engine-sized chunks (equal
        # sizes, so the mean of the chunk means is the batch mean, src/dalle_mtf/models.py:354)
        total = None
        for c in range(B // eng.B):
            l, _ = model.forward({"tokens": tokens[c * eng.B:(c + 1) * eng.B]}, return_loss=True)
            total = l.clone() if total is None else total + l
        return EstimatorSpec(mode=mode, loss=total / (B // eng.B))
    loss, _loss_batch = model.forward({"tokens": tokens}, return_loss=True)
    scalar_summary("loss", loss)
    scalar_summary("lr", st["lr_fn"]())

    def train_op():
        eng.backward()
        st["update_op"]()
        return eng.global_step
    return EstimatorSpec(mode=mode, loss=loss, train_op=train_op, host_call=create_host_call(params["model_path"])
                         if (params.get("model_path") and st["rank"] == 0) else None, training_hooks=[st["saver"]])
