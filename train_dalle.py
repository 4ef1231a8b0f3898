"""train_dalle.py -- same CLI and config contract as the reference entry point (train_dalle.py:12-103):
    python train_dalle.py --model dalle_example [--new] [--gpu_ids ...]
Multi-GPU data parallel = one process per GPU:
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_dalle.py --model dalle_example
(`mesh_shape data:N` / `layout batch_dim:data` describe exactly this; the unused `model` axis is ignored.)"""
import argparse
import os
import sys
from functools import partial

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))

import torch  # noqa: E402
from src.utils import *  # noqa: E402,F401,F403
from src.model_fns import dalle_model_fn  # noqa: E402
from src.input_fns import dalle_input_fn  # noqa: E402
from src.data import get_tokenizer  # noqa: E402
from src.estimator import Estimator, load_global_step_from_checkpoint_dir  # noqa: E402


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument("--tpu", type=str, help="Accepted for CLI compatibility; TPUs are not a target of this build.")
    parser.add_argument("--gpu_ids", nargs="+", type=str, default=["device:GPU:0"],
                        help="Kept for CLI compatibility: with one process per GPU the device is LOCAL_RANK.")
    parser.add_argument("--model", type=str, default=None, help="JSON file that contains model parameters.")
    parser.add_argument("--new", action="store_true", help="If set, deletes previous checkpoint, if it exists, and "
                                                           "starts a new training run")
    args = parser.parse_args()
    assert args.model is not None, "Model must be set"
    return args


def main():
    args = parse_args()
    logging = setup_logging(args, rank=int(os.environ.get("RANK", "0")))
    params = fetch_model_params(args.model)
    params["vae_params"] = fetch_model_params(params["vae_model"])
    assert params["model_type"].lower() == "dalle", f'model_type {params["model_type"]} not recognized'
    assert args.tpu is None, "TPUs are not supported by the MI355X build"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        from src.dp import barrier, init_process_group
        init_process_group(local_rank)

    if args.new and int(os.environ.get("RANK", "0")) == 0:
        maybe_remove_gs_or_filepath(params["model_path"])
    if world > 1:
        barrier()      # nobody looks for a checkpoint before rank 0 has cleared the directory

    current_step = int(load_global_step_from_checkpoint_dir(params["model_path"]))
    logging.info(f"Current step: {current_step}")

    mesh_shape = parse_mesh_shape(params["mesh_shape"])
    params["num_cores"] = world
    params["use_tpu"] = False
    params["gpu_ids"] = args.gpu_ids
    if "model" in mesh_shape:
        logging.info("mesh axis 'model' is not mapped by any layout rule in the reference configs; ignored")
    tokenizer = get_tokenizer(params["tokenizer"], vocab_size=params["text_vocab_size"])
    assert len(tokenizer) == params["text_vocab_size"], \
        f"tokenizer vocab size {len(tokenizer)} must equal model vocab size {params['text_vocab_size']}"
    params["padding_id"] = tokenizer.encode(tokenizer.pad_token)[0]
    params["batch_size"] = params["train_batch_size"] // world
    params["dp_rank"], params["dp_world"] = int(os.environ.get("RANK", "0")), world

    estimator = Estimator(model_fn=dalle_model_fn, model_dir=params["model_path"], params=params,
                          log_every=min(params["iterations"] or 100, 100), logger=logging)
    try:
        has_predict_or_eval_steps = params["predict_steps"] > 0 or params["eval_steps"] > 0
        while current_step < params["train_steps"]:
            nxt = params["train_steps"]
            if has_predict_or_eval_steps:
                nxt = min(current_step + params["steps_per_checkpoint"], params["train_steps"])
            current_step = estimator.train(input_fn=partial(dalle_input_fn, eval=False), max_steps=nxt)
            if params["predict_steps"] > 0:
                raise NotImplementedError
            if params["eval_steps"] > 0:
                estimator.evaluate(input_fn=partial(dalle_input_fn, eval=True), steps=params["eval_steps"])
    finally:
        estimator.close()      # stops and joins the input producer: no thread may be alive at interpreter exit


if __name__ == "__main__":
    main()
