"""Host utilities mirroring src/utils/utils.py of the reference (config loading, mode strings, logging,
scalar summaries) re-hosted on plain Python/PyTorch.  TPU-only pieces (simd_mesh_setup, utils.py:163-182)
are out of scope."""
import json
import logging
import os
import sys
from collections import defaultdict
from shutil import rmtree
from urllib.parse import urlparse

__all__ = ["fetch_model_params", "yes_or_no", "mode_to_str", "remove_gs_or_filepath", "maybe_remove_gs_or_filepath",
           "get_n_trainable_vars", "get_graph_info", "setup_logging", "scalar_summary", "ModeKeys", "SummaryWriter",
           "parse_mesh_shape", "create_host_call"]

_REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


class ModeKeys:
    """tf.estimator.ModeKeys stand-in (same string values)."""
    TRAIN = "train"
    EVAL = "eval"
    PREDICT = "infer"


def fetch_model_params(model):
    """reference utils.py:13-17: a name under ./configs/ or a path ending in .json; missing keys read as None."""
    model_path = model if model.endswith(".json") else f"./configs/{model}.json"
    if not os.path.exists(model_path) and not model.endswith(".json"):
        model_path = os.path.join(_REPO_ROOT, "configs", f"{model}.json")
    with open(model_path) as f:
        params = json.load(f)
    return defaultdict(lambda: None, params)


def yes_or_no(question):
    while True:
        reply = str(input(question + ' (y/n): ')).lower().strip()
        if reply[:1] == 'y':
            return True
        if reply[:1] == 'n':
            return False


def mode_to_str(mode):
    """reference utils.py:29-37."""
    if mode == ModeKeys.PREDICT:
        return "predict"
    elif mode == ModeKeys.EVAL:
        return "eval"
    elif mode == ModeKeys.TRAIN:
        return "train"
    raise ValueError(f"Invalid mode {mode}")


def remove_gs_or_filepath(path):
    if urlparse(path).scheme == "gs":
        raise ValueError("gs:// paths are not supported by the MI355X build; use a local model_path")
    if os.path.exists(path):
        rmtree(path)


def maybe_remove_gs_or_filepath(path):
    """reference utils.py:48-52 (interactive confirmation before deleting model_path)."""
    if yes_or_no(f"Are you sure you want to remove '{path}' to start afresh?"):
        remove_gs_or_filepath(path)
    else:
        sys.exit()


def get_n_trainable_vars(variables):
    """reference utils.py:55-69; `variables` maps name -> shape."""
    total = 0
    for shape in variables.values():
        n = 1
        for s in shape:
            n *= int(s)
        total += n
    print(f"\n\nN PARAMS:\n{total:,}\n\n")
    return total


def get_graph_info(variables):
    return get_n_trainable_vars(variables)


def setup_logging(args, logdir="logs", rank=0):
    """reference utils.py:184-195: file + stdout logger named after the config.  One process per GPU here: only rank 0
    writes the log file and the console (the other ranks log warnings and above to stderr)."""
    name = os.path.splitext(os.path.basename(args.model))[0]
    logger = logging.getLogger("dalle_mtf_amd")
    logger.propagate = False
    if rank == 0:
        os.makedirs(logdir, exist_ok=True)
        logger.setLevel(logging.INFO)
        logger.handlers = [logging.FileHandler(f"{logdir}/{name}.log"), logging.StreamHandler(sys.stdout)]
    else:
        logger.setLevel(logging.WARNING)
        logger.handlers = [logging.StreamHandler(sys.stderr)]
    return logger


def parse_mesh_shape(mesh_shape: str):
    """'data:16,model:2' -> {'data': 16, 'model': 2} (mtf.convert_to_shape, src/model_fns.py:81)."""
    out = {}
    for part in (mesh_shape or "").split(","):
        if ":" in part:
            k, v = part.split(":")
            out[k.strip()] = int(v)
    return out


_SUMMARIES = {}


def scalar_summary(name, x):
    """reference utils.py:216-227: remember a named scalar (device tensor or float) for the host call."""
    _SUMMARIES[name] = x
    return x


class SummaryWriter:
    """tf2.summary.create_file_writer stand-in (reference src/utils/utils.py:103-161, src/model_fns_tf.py:68-96): scalars and
    images go to a TensorBoard event file `events.out.tfevents.<time>.<host>` under model_dir -- TFRecord framing of
    tensorflow.Event protos, written without TensorFlow (src/data/tfrecord.py supplies the framing, CRCs and the proto wire
    helpers) -- and, for people without TensorBoard, to `summaries.jsonl` + `images/*.png`.
    Event {1: wall_time (double), 2: step (int64), 3: file_version (string) | 5: Summary};
    Summary {1: repeated Value {1: tag, 2: simple_value (float) | 4: Image {1: height, 2: width, 3: colorspace, 4: PNG bytes}}}."""

    def __init__(self, model_dir):
        import socket
        import time
        os.makedirs(model_dir, exist_ok=True)
        self.path = os.path.join(model_dir, "summaries.jsonl")
        self.events_path = os.path.join(model_dir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}")
        self._write_event(0, file_version=b"brain.Event:2")

    def _write_event(self, step, file_version=None, summary=None):
        import struct
        import time
        from ..data.tfrecord import _ld, _varint, masked_crc32c
        ev = bytes([(1 << 3) | 1]) + struct.pack("<d", time.time()) + bytes([(2 << 3) | 0]) + _varint(int(step))
        if file_version is not None:
            ev += _ld(3, file_version)
        if summary is not None:
            ev += _ld(5, summary)
        hdr = struct.pack("<Q", len(ev))
        with open(self.events_path, "ab") as f:
            f.write(hdr + struct.pack("<I", masked_crc32c(hdr)) + ev + struct.pack("<I", masked_crc32c(ev)))

    def scalars(self, step, **kv):
        import struct
        from ..data.tfrecord import _ld
        rec = {"step": int(step)}
        summary = b""
        for k, v in kv.items():
            rec[k] = float(v.item() if hasattr(v, "item") else v)
            summary += _ld(1, _ld(1, k.encode()) + bytes([(2 << 3) | 5]) + struct.pack("<f", rec[k]))
        with open(self.path, "a") as f:
            f.write(json.dumps(rec) + "\n")
        self._write_event(step, summary=summary)

    def images(self, step, name, x, max_images=3):
        """tf2.summary.image (reference src/model_fns_tf.py:74-75,89-90; TF writes max_outputs = 3 images):
        x NHWC in [0, 1] -> event-file Image values tagged <name>/image/<i> and model_dir/images/<name>_<step>_<i>.png"""
        import io
        import numpy as np
        from PIL import Image
        from ..data.tfrecord import _ld, _varint
        d = os.path.join(os.path.dirname(self.path), "images")
        os.makedirs(d, exist_ok=True)
        a = x[:max_images].detach().float().cpu().numpy() if hasattr(x, "detach") else np.asarray(x[:max_images], np.float32)
        a = (np.clip(a, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
        summary = b""
        for i, im in enumerate(a):
            pil = Image.fromarray(im[..., 0] if im.shape[-1] == 1 else im[..., :3])
            pil.save(os.path.join(d, f"{name.replace('/', '_')}_{int(step)}_{i}.png"))
            buf = io.BytesIO()
            pil.save(buf, format="PNG")
            img = (bytes([(1 << 3) | 0]) + _varint(im.shape[0]) + bytes([(2 << 3) | 0]) + _varint(im.shape[1]) +
                   bytes([(3 << 3) | 0]) + _varint(1 if im.shape[-1] == 1 else 3) + _ld(4, buf.getvalue()))
            summary += _ld(1, _ld(1, f"{name}/image/{i}".encode()) + _ld(4, img))
        self._write_event(step, summary=summary)


def create_host_call(model_dir):
    """reference utils.py:103-161: returns (fn, args) that writes the collected scalar summaries."""
    if not _SUMMARIES:
        return None
    writer = SummaryWriter(model_dir)

    def host_call_fn(global_step, **tensors):
        writer.scalars(global_step, **{k: v for k, v in tensors.items() if "image" not in k})
    return host_call_fn, dict(_SUMMARIES)
