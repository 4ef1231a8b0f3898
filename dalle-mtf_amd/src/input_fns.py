"""Input pipelines (reference src/input_fns.py:1-120).

Two sources behind the reference's two entry points `dalle_input_fn` / `vae_input_fn`:
  * real data: TFRecord shards of tf.train.Example{"image": JPEG bytes, "caption": int64[]} (written by
    src/data/create_tfrecords.py:47-56) or, for the VAE, a glob of JPEG files.  Framing/proto parsing is in
    data/tfrecord.py; JPEG decode is PIL's libjpeg (pixel-exactness against tf.image.decode_jpeg's IDCT / upsampling settings is
    UNVERIFIED: TensorFlow cannot run here); the reference's
    tf.data graph (file shuffle -> 4-way interleave -> map -> shuffle(5*batch) -> batch(drop_remainder) -> prefetch
    -> repeat, input_fns.py:23-29,104-120) is restated as a Python generator with a decode thread pool and a
    bounded prefetch queue feeding the GPU step;
  * `synthetic*` / `gs://` paths: seeded synthetic batches of the same shapes (the bench metric's data, §8(d)).
"""
import atexit
import glob as _glob
import queue
import threading
import weakref
from concurrent.futures import ThreadPoolExecutor
import io

import numpy as np
import torch

from .data.tfrecord import read_records, decode_example


def crop_center_and_resize(img, size):
    """reference input_fns.py:4-12 over tf.image.crop_and_resize (bilinear, extrapolation 0).

    Kept bug-for-bug: the reference names s[0] (rows) `w` and s[1] (cols) `h`, and passes the box as
    [(1-wn)/2, (1-hn)/2, wn, hn] where TF expects [y1, x1, y2, x2]; for square inputs that is the identity box
    [0,0,1,1] (a plain bilinear resize), for non-square inputs it is whatever that expression selects.
    img: [H,W,C] uint8/float; returns [size,size,C] float32."""
    img = np.asarray(img)
    H, W = img.shape[0], img.shape[1]
    w, h = H, W
    c = max(w, h)
    wn, hn = h / c, w / c
    y1, x1, y2, x2 = np.float32((1 - wn) / 2), np.float32((1 - hn) / 2), np.float32(wn), np.float32(hn)
    f32 = np.float32
    src = img.astype(np.float32)
    idx = np.arange(size, dtype=np.float32)
    if size > 1:
        in_y = y1 * f32(H - 1) + idx * ((y2 - y1) * f32(H - 1) / f32(size - 1))
        in_x = x1 * f32(W - 1) + idx * ((x2 - x1) * f32(W - 1) / f32(size - 1))
    else:
        in_y = np.full((1,), f32(0.5) * (y1 + y2) * f32(H - 1), dtype=np.float32)
        in_x = np.full((1,), f32(0.5) * (x1 + x2) * f32(W - 1), dtype=np.float32)
    oky = (in_y >= 0) & (in_y <= H - 1)
    okx = (in_x >= 0) & (in_x <= W - 1)
    cy, cx = np.clip(in_y, 0, H - 1), np.clip(in_x, 0, W - 1)
    top, bot = np.floor(cy).astype(np.int64), np.ceil(cy).astype(np.int64)
    lef, rig = np.floor(cx).astype(np.int64), np.ceil(cx).astype(np.int64)
    ly = (cy - top.astype(np.float32))[:, None, None]
    lx = (cx - lef.astype(np.float32))[None, :, None]
    tl, tr = src[top][:, lef], src[top][:, rig]
    bl, br = src[bot][:, lef], src[bot][:, rig]
    t = tl + (tr - tl) * lx
    b = bl + (br - bl) * lx
    out = t + (b - t) * ly
    out = out * (oky[:, None, None] & okx[None, :, None])
    return out.astype(np.float32)


def decode_img(img_bytes, size, channels=3):
    """reference input_fns.py:15-21: decode_jpeg(channels) -> crop_center_and_resize -> (x - 127.5) / 127.5."""
    from PIL import Image
    im = Image.open(io.BytesIO(img_bytes))
    im = im.convert("L" if channels == 1 else "RGB")
    arr = np.asarray(im)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    out = crop_center_and_resize(arr, size)
    return (out - np.float32(127.5)) / np.float32(127.5)


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
    """only an explicit 'synthetic...' path selects the seeded synthetic batches; an empty path or a gs:// URL carried over
    from a reference config is an error, not silent noise training (there is no GCS access here)"""
    p = str(path or "")
    if p.startswith("synthetic"):
        return True
    if not p or p.startswith("gs://"):
        raise ValueError(f"dataset path {path!r}: use a local glob of *.tfrecords / image files, or 'synthetic' for seeded "
                         "synthetic batches (gs:// buckets are not reachable from this build)")
    return False


def _stream_seed(params, base, eval):
    """seed of a training stream: salted with the data-parallel rank (ranks must not draw identical synthetic batches) and
    with the step the run (re)started from (a resumed run must not replay the stream's head)"""
    if eval:
        return base
    return base + 7919 * int(params.get("dp_rank", 0)) + 104729 * int(params.get("_input_start_step", 0))


def _batch_size(params, eval):
    return params["batch_size"] if params.get("batch_size") else params["eval_batch_size" if eval else "train_batch_size"]


def _dp_shard(params):
    """(rank, world): data-parallel ranks read disjoint element streams (element i goes to rank i % world)."""
    return int(params.get("dp_rank", 0)), int(params.get("dp_world", 1))


def read_labeled_tfrecord(params):
    """reference input_fns.py:41-55: Example -> (image fp32 [size,size,C], caption int32 [text_seq_len])."""
    size, ch = params["dataset"]["image_size"], params.get("n_channels") or 3

    def read_fn(example):
        feats = decode_example(example)
        image = decode_img(feats["image"][0], size, ch)
        label = truncate_or_pad_label(feats.get("caption", []), params)
        return image, label
    return read_fn


def read_tfrecord(params):
    """reference input_fns.py:58-68: Example -> (image, image)."""
    size, ch = params["dataset"]["image_size"], params.get("n_channels") or 3

    def read_fn(example):
        image = decode_img(decode_example(example)["image"][0], size, ch)
        return image, image
    return read_fn


def _interleave_records(files, cycle_length=4, rank=0, world=1):
    """tf.data parallel_interleave(TFRecordDataset, cycle_length=4, sloppy=False, block_length=1): round-robin one
    record at a time over up to cycle_length open files; an exhausted file's slot is refilled with the next file.
    Data-parallel sharding happens HERE: element n of the interleaved stream belongs to rank n % world, and the records
    of other ranks are seeked past without being read or CRC-checked."""
    pending = list(files)
    n = [0]

    def mine():
        return n[0] % world == rank

    slots = []
    while pending and len(slots) < cycle_length:
        slots.append(read_records(pending.pop(0), want=mine))
    i = 0
    while slots:
        i %= len(slots)
        try:
            rec = next(slots[i])
            n[0] += 1
            if rec is not None:
                yield rec
            i += 1
        except StopIteration:
            if pending:
                slots[i] = read_records(pending.pop(0), want=mine)
            else:
                slots.pop(i)


def _shuffle(it, buffer_size, rng):
    """tf.data shuffle(buffer_size): fill a buffer, emit a uniformly random slot, refill it."""
    buf = []
    for x in it:
        if len(buf) < buffer_size:
            buf.append(x)
            continue
        j = int(rng.integers(0, buffer_size))
        yield buf[j]
        buf[j] = x
    while buf:
        j = int(rng.integers(0, len(buf)))
        yield buf.pop(j)


_LIVE_PREFETCH = weakref.WeakSet()


def _close_live_prefetchers():
    """atexit: a producer still running when the interpreter finalises is unwound by force (pthread_exit through C++
    frames of torch/numpy -> std::terminate -> SIGABRT after an otherwise clean run).  atexit callbacks run before
    finalisation freezes daemon threads, so stopping and JOINING every live producer here makes the exit clean even
    when an entry point forgot close()."""
    for p in list(_LIVE_PREFETCH):
        p.close()


atexit.register(_close_live_prefetchers)


class _Prefetch:
    """Bounded background producer (tf.data prefetch): decodes on a thread pool so JPEG work overlaps the GPU step.
    close() stops the producer, drains the queue and joins the thread (idempotent; also a context manager)."""

    def __init__(self, make_epoch, map_fn, batch, shuffle_buf, seed, workers=8, depth=4, shard=(0, 1)):
        self._q = queue.Queue(maxsize=depth)
        self._stop = threading.Event()
        self._args = (make_epoch, map_fn, batch, shuffle_buf, seed, workers, shard)
        self._t = threading.Thread(target=self._run, daemon=True, name="dalle-prefetch")
        _LIVE_PREFETCH.add(self)
        self._t.start()

    def _put(self, item):
        """bounded put that gives up when the consumer has closed the stream"""
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def _run(self):
        make_epoch, map_fn, batch, shuffle_buf, seed, workers, (rank, world) = self._args
        rng = np.random.default_rng(seed)
        pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="dalle-decode")
        try:
            def mapped():             # ordered map with a bounded number of decodes in flight
                inflight, ahead = [], workers * 4
                for raw in make_epoch(rank, world):          # make_epoch yields this rank's elements only
                    inflight.append(pool.submit(map_fn, raw))
                    if len(inflight) >= ahead:
                        yield inflight.pop(0).result()
                for f in inflight:
                    yield f.result()

            while not self._stop.is_set():     # .repeat() sits after batch(): every epoch drops its remainder
                stream = mapped()
                if shuffle_buf:
                    stream = _shuffle(stream, shuffle_buf, rng)
                window, produced = [], 0
                for el in stream:
                    if self._stop.is_set():
                        return
                    window.append(el)
                    if len(window) < batch:
                        continue
                    a = torch.from_numpy(np.stack([w[0] for w in window]))
                    b = torch.from_numpy(np.stack([w[1] for w in window]))
                    window, produced = [], produced + 1
                    if not self._put((a, b)):
                        return
                if produced == 0:
                    raise ValueError(f"input pipeline: an epoch holds fewer than batch_size={batch} elements")
        except BaseException as e:  # surfaced to the consumer
            if not self._stop.is_set():
                self._put(e)
        finally:
            pool.shutdown(wait=True, cancel_futures=True)

    def __iter__(self):
        return self

    def __next__(self):
        if self._stop.is_set():
            raise StopIteration
        item = self._q.get()
        if isinstance(item, BaseException):
            raise item
        return item

    def close(self):
        self._stop.set()
        t = self._t
        while t.is_alive():            # a producer blocked in put() wakes within its 0.1 s timeout; drain so it can leave
            try:
                while True:
                    self._q.get_nowait()
            except queue.Empty:
                pass
            t.join(timeout=0.05)
        _LIVE_PREFETCH.discard(self)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _file_order(files, eval, seed):
    """Dataset.shuffle(file_count, reshuffle_each_iteration=False): one fixed permutation (train only)."""
    files = sorted(files)
    if not eval and len(files) > 1:
        files = [files[i] for i in np.random.default_rng(seed).permutation(len(files))]
    return files


def _synthetic_dalle(params, eval):
    ds = params["dataset"]
    B = _batch_size(params, eval)
    size, ch = ds["image_size"], params.get("n_channels") or 3
    pad = params["padding_id"] if params.get("padding_id") is not None else params["text_vocab_size"] - 1
    rng = np.random.default_rng(_stream_seed(params, 1, eval) if not eval else 101)
    while True:
        img = rng.integers(0, 256, size=(B, size, size, ch), dtype=np.uint8)
        img = (img.astype(np.float32) - 127.5) / 127.5
        yield torch.from_numpy(img), torch.from_numpy(_synthetic_captions(rng, B, params["text_seq_len"], pad))


def _synthetic_vae(params, eval):
    ds = params["dataset"]
    B = _batch_size(params, eval)
    size, ch = ds["image_size"], params.get("n_channels") or 3
    rng = np.random.default_rng(_stream_seed(params, 0, eval) if not eval else 100)
    while True:
        img = rng.integers(0, 256, size=(B, size, size, ch), dtype=np.uint8)
        img = torch.from_numpy((img.astype(np.float32) - 127.5) / 127.5)
        yield img, img


def dalle_input_fn(params, eval=False):
    """reference input_fns.py:104-120.  Yields (image [B,H,W,C] fp32 in [-1,1], caption ids [B,text_seq_len] int32)
    forever."""
    ds = params["dataset"]
    path = ds["train_path"] if not eval else ds["eval_path"]
    if _is_synthetic(path):
        return _synthetic_dalle(params, eval)
    files = _glob.glob(path)
    if not files:
        raise FileNotFoundError(f"dalle_input_fn: no TFRecord files match {path!r}")
    seed = int(params.get("input_seed", 0))
    files = _file_order(files, eval, seed)
    B = _batch_size(params, eval)
    sseed = seed + 1 + 104729 * int(params.get("_input_start_step", 0) if not eval else 0)
    return _Prefetch(lambda r, w: _interleave_records(files, 4, r, w), read_labeled_tfrecord(params), B,
                     0 if eval else B * 5, sseed, shard=_dp_shard(params))


def vae_input_fn(params, eval=False):
    """reference input_fns.py:68-101.  Yields (image, image) forever ("returns image twice")."""
    ds = params["dataset"]
    path = ds["train_path"] if not eval else ds["eval_path"]
    if _is_synthetic(path):
        return _synthetic_vae(params, eval)
    files = _glob.glob(path)
    if not files:
        raise FileNotFoundError(f"vae_input_fn: nothing matches {path!r}")
    seed = int(params.get("input_seed", 0))
    files = _file_order(files, eval, seed)
    B = _batch_size(params, eval)
    size = ds["image_size"]
    sseed = seed + 1 + 104729 * int(params.get("_input_start_step", 0) if not eval else 0)
    if ds.get("tfrecords"):
        return _Prefetch(lambda r, w: _interleave_records(files, 4, r, w), read_tfrecord(params), B,
                         0 if eval else B * 5, sseed, shard=_dp_shard(params))

    def _process_path(file_path):          # input_fns.py:92-96 (decode_img with its default 3 channels)
        with open(file_path, "rb") as f:
            img = decode_img(f.read(), size)
        return img, img
    return _Prefetch(lambda r, w: iter(files[r::w]), _process_path, B, 0 if eval else B * 5, sseed, shard=_dp_shard(params))
