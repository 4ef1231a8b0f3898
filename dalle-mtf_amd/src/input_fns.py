"""Input pipelines.  The reference's tf.data TFRecord/JPEG readers (src/input_fns.py:15-120) are the
"next" row (f)1 of SURVEY.md §8; the hot-path metric uses synthetic batches:
CIFAR-shaped uint8 images normalised (x-127.5)/127.5 (input_fns.py:20) and random captions right-padded
with padding_id to text_seq_len (input_fns.py:32-38)."""
import numpy as np
import torch


def truncate_or_pad_label(label, params):
    """reference input_fns.py:32-38."""
    T, pad = params["text_seq_len"], params["padding_id"]
    label = np.asarray(label, dtype=np.int32).reshape(-1)
    out = np.full((T,), pad, dtype=np.int32)
    n = min(label.shape[0], T)
    out[:n] = label[:n]
    return out


def _synthetic_captions(rng, B, T, pad):
    out = np.full((B, T), pad, dtype=np.int32)
    for b in range(B):
        n = int(rng.integers(1, T + 1))
        out[b, :n] = rng.integers(0, pad, size=n, dtype=np.int32)
    return out


def _is_synthetic(path):
    return (not path) or str(path).startswith("synthetic") or str(path).startswith("gs://")


def dalle_input_fn(params, eval=False):
    """yields (image [B,H,W,C] fp32 in [-1,1], caption ids [B,text_seq_len] int32) forever."""
    ds = params["dataset"]
    path = ds["train_path"] if not eval else ds["eval_path"]
    if not _is_synthetic(path):
        raise NotImplementedError("TFRecord input is SURVEY.md §8(f) rank 1 (next row); use a 'synthetic' dataset path")
    B = params["batch_size"] if params.get("batch_size") else params["eval_batch_size" if eval else "train_batch_size"]
    size, ch = ds["image_size"], params.get("n_channels") or 3
    pad = params["padding_id"] if params.get("padding_id") is not None else params["text_vocab_size"] - 1
    rng = np.random.default_rng(1 if not eval else 101)

    def gen():
        while True:
            img = rng.integers(0, 256, size=(B, size, size, ch), dtype=np.uint8)
            img = (img.astype(np.float32) - 127.5) / 127.5
            yield torch.from_numpy(img), torch.from_numpy(_synthetic_captions(rng, B, params["text_seq_len"], pad))
    return gen()


def vae_input_fn(params, eval=False):
    """yields (image, image) forever ("returns image twice", input_fns.py:64,100)."""
    ds = params["dataset"]
    path = ds["train_path"] if not eval else ds["eval_path"]
    if not _is_synthetic(path):
        raise NotImplementedError("JPEG/TFRecord input is SURVEY.md §8(f) rank 1 (next row); use a 'synthetic' dataset path")
    B = params["batch_size"] if params.get("batch_size") else params["eval_batch_size" if eval else "train_batch_size"]
    size, ch = ds["image_size"], params.get("n_channels") or 3
    rng = np.random.default_rng(0 if not eval else 100)

    def gen():
        while True:
            img = rng.integers(0, 256, size=(B, size, size, ch), dtype=np.uint8)
            img = torch.from_numpy((img.astype(np.float32) - 127.5) / 127.5)
            yield img, img
    return gen()
