"""Input pipeline: TFRecord framing, tf.train.Example wire format, JPEG decode +
the reference's crop/resize/normalise/pad semantics (reference src/input_fns.py:4-66, src/data/create_tfrecords.py)."""
import io
import json
import os
import struct

import numpy as np
import pytest

from src.data import tfrecord as tfr
from src.data.create_tfrecords import create_paired_dataset, create_random_dataset, serialize_example
from src import input_fns


# ---------------------------------------------------------------- CRC-32C / framing known answers

def test_crc32c_known_answers():
    # RFC 3720 appendix B.4 test vectors for CRC-32C (Castagnoli)
    assert tfr.crc32c(b"123456789") == 0xE3069283
    assert tfr.crc32c(bytes(32)) == 0x8A9136AA
    assert tfr.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tfr.crc32c(bytes(range(32))) == 0x46DD794E
    assert tfr.crc32c(b"") == 0


def test_record_framing_roundtrip_and_corruption(tmp_path):
    recs = [b"", b"a", os.urandom(1000), b"x" * 70000]
    p = str(tmp_path / "t.tfrecords")
    tfr.write_records(p, recs)
    assert list(tfr.read_records(p, verify_crc=True)) == recs
    raw = open(p, "rb").read()
    # layout of the first (empty) record: len=0, crc(len), crc(data)
    assert raw[:8] == struct.pack("<Q", 0) and len(raw) == sum(16 + len(r) for r in recs)
    bad = bytearray(raw)
    bad[16 + 12 + 1] ^= 1  # flip a payload bit of record 2
    open(p, "wb").write(bytes(bad))
    with pytest.raises(IOError):
        list(tfr.read_records(p, verify_crc=True))
    open(p, "wb").write(raw[:-3])
    with pytest.raises(IOError):
        list(tfr.read_records(p))


# ---------------------------------------------------------------- Example proto vs the real protobuf runtime

def _example_class():
    """Build tensorflow.Example's message classes from its published schema with google.protobuf, as an
    independent implementation of the wire format."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fdp = descriptor_pb2.FileDescriptorProto()
    fdp.name, fdp.package, fdp.syntax = "example_test.proto", "tensorflow", "proto3"
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = fdp.message_type.add()
        m.name = name
        return m

    def field(m, name, num, typ, label=T.LABEL_OPTIONAL, type_name=None, oneof=None):
        f = m.field.add()
        f.name, f.number, f.type, f.label = name, num, typ, label
        if type_name:
            f.type_name = type_name
        if oneof is not None:
            f.oneof_index = oneof
        return f

    field(msg("BytesList"), "value", 1, T.TYPE_BYTES, T.LABEL_REPEATED)
    field(msg("FloatList"), "value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED)
    field(msg("Int64List"), "value", 1, T.TYPE_INT64, T.LABEL_REPEATED)
    feat = msg("Feature")
    feat.oneof_decl.add().name = "kind"
    field(feat, "bytes_list", 1, T.TYPE_MESSAGE, type_name=".tensorflow.BytesList", oneof=0)
    field(feat, "float_list", 2, T.TYPE_MESSAGE, type_name=".tensorflow.FloatList", oneof=0)
    field(feat, "int64_list", 3, T.TYPE_MESSAGE, type_name=".tensorflow.Int64List", oneof=0)
    feats = msg("Features")
    entry = feats.nested_type.add()
    entry.name = "FeatureEntry"
    entry.options.map_entry = True
    field(entry, "key", 1, T.TYPE_STRING)
    field(entry, "value", 2, T.TYPE_MESSAGE, type_name=".tensorflow.Feature")
    field(feats, "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".tensorflow.Features.FeatureEntry")
    field(msg("Example"), "features", 1, T.TYPE_MESSAGE, type_name=".tensorflow.Features")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("tensorflow.Example"))


def test_example_wire_format_against_protobuf():
    Example = _example_class()
    jpeg = os.urandom(333)
    caption = [0, 1, 127, 128, 50256, 2 ** 40, -1]
    # ours -> protobuf
    ex = Example()
    ex.ParseFromString(serialize_example(jpeg, caption))
    assert ex.features.feature["image"].bytes_list.value[0] == jpeg
    assert list(ex.features.feature["caption"].int64_list.value) == caption
    # protobuf -> ours
    ex2 = Example()
    ex2.features.feature["image"].bytes_list.value.append(jpeg)
    ex2.features.feature["caption"].int64_list.value.extend(caption)
    ex2.features.feature["score"].float_list.value.extend([0.5, -2.0])
    got = tfr.decode_example(ex2.SerializeToString())
    assert got["image"] == [jpeg] and got["caption"] == caption and got["score"] == [0.5, -2.0]
    # empty caption (VarLenFeature with no values) parses to []
    ex3 = Example()
    ex3.features.feature["image"].bytes_list.value.append(b"z")
    ex3.features.feature["caption"].int64_list.SetInParent()
    assert tfr.decode_example(ex3.SerializeToString())["caption"] == []


# ---------------------------------------------------------------- crop_and_resize / decode semantics

def _crop_and_resize_loops(img, box, size):
    """tf.image.crop_and_resize (bilinear, extrapolation_value 0) written as the kernel's scalar loops
    (tensorflow/core/kernels/image/crop_and_resize_op.cc, CropAndResize functor)."""
    H, W, C = img.shape
    y1, x1, y2, x2 = [np.float32(v) for v in box]
    out = np.zeros((size, size, C), np.float32)
    hs = (y2 - y1) * np.float32(H - 1) / np.float32(size - 1) if size > 1 else np.float32(0)
    ws = (x2 - x1) * np.float32(W - 1) / np.float32(size - 1) if size > 1 else np.float32(0)
    for y in range(size):
        in_y = y1 * np.float32(H - 1) + np.float32(y) * hs if size > 1 else np.float32(0.5) * (y1 + y2) * np.float32(H - 1)
        if in_y < 0 or in_y > H - 1:
            continue
        top, bot = int(np.floor(in_y)), int(np.ceil(in_y))
        ly = in_y - np.float32(top)
        for x in range(size):
            in_x = x1 * np.float32(W - 1) + np.float32(x) * ws if size > 1 else np.float32(0.5) * (x1 + x2) * np.float32(W - 1)
            if in_x < 0 or in_x > W - 1:
                continue
            lef, rig = int(np.floor(in_x)), int(np.ceil(in_x))
            lx = in_x - np.float32(lef)
            for c in range(C):
                tl, tr = np.float32(img[top, lef, c]), np.float32(img[top, rig, c])
                bl, br = np.float32(img[bot, lef, c]), np.float32(img[bot, rig, c])
                t = tl + (tr - tl) * lx
                b = bl + (br - bl) * lx
                out[y, x, c] = t + (b - t) * ly
    return out


@pytest.mark.parametrize("shape,size", [((32, 32, 3), 32), ((40, 40, 3), 16), ((17, 17, 1), 32),
                                        ((24, 48, 3), 16), ((48, 24, 3), 16), ((9, 9, 3), 1)])
def test_crop_center_and_resize_matches_scalar_kernel(shape, size):
    rng = np.random.default_rng(sum(shape) + size)
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    H, W = shape[:2]
    c = max(H, W)
    wn, hn = W / c, H / c                      # reference input_fns.py:5-8 with its (w,h) = (rows, cols) naming
    box = [(1 - wn) / 2, (1 - hn) / 2, wn, hn]
    want = _crop_and_resize_loops(img, box, size)
    got = input_fns.crop_center_and_resize(img, size)
    assert got.shape == (size, size, shape[2]) and got.dtype == np.float32
    np.testing.assert_array_equal(got, want)


def test_square_same_size_is_identity():
    img = np.random.default_rng(3).integers(0, 256, size=(32, 32, 3), dtype=np.uint8)
    np.testing.assert_array_equal(input_fns.crop_center_and_resize(img, 32), img.astype(np.float32))


def _jpeg(arr, quality=95):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr.squeeze() if arr.shape[-1] == 1 else arr).save(buf, format="JPEG", quality=quality)
    return buf.getvalue()


def test_decode_img_range_channels():
    rng = np.random.default_rng(5)
    smooth = np.clip(np.add.outer(np.arange(32) * 6, np.arange(32) * 2)[:, :, None] + np.array([0, 20, 40]), 0, 255)
    smooth = smooth.astype(np.uint8)
    out = input_fns.decode_img(_jpeg(smooth), 32, 3)
    assert out.shape == (32, 32, 3) and out.dtype == np.float32
    assert out.min() >= -1.0 and out.max() <= 1.0
    # lossy codec, smooth image: within a few grey levels of the source after (x-127.5)/127.5
    assert np.abs(out - (smooth.astype(np.float32) - 127.5) / 127.5).max() < 8 / 127.5
    g = input_fns.decode_img(_jpeg(smooth[:, :, :1]), 16, 3)      # grayscale file decoded with channels=3
    assert g.shape == (16, 16, 3) and np.array_equal(g[..., 0], g[..., 1])
    assert input_fns.decode_img(_jpeg(smooth), 8, 1).shape == (8, 8, 1)
    del rng


def test_truncate_or_pad_label():
    p = {"text_seq_len": 6, "padding_id": 99}
    assert input_fns.truncate_or_pad_label([1, 2, 3], p).tolist() == [1, 2, 3, 99, 99, 99]
    assert input_fns.truncate_or_pad_label(list(range(10)), p).tolist() == [0, 1, 2, 3, 4, 5]
    assert input_fns.truncate_or_pad_label([], p).tolist() == [99] * 6
    assert input_fns.truncate_or_pad_label([7], p).dtype == np.int32


# ---------------------------------------------------------------- end-to-end datasets

class _Tok:
    def encode(self, text):
        return [ord(ch) % 50 for ch in text]


def _make_dataset(tmp_path, n=23, per_file=5):
    from PIL import Image
    raw = tmp_path / "raw"
    os.makedirs(raw)
    rng = np.random.default_rng(0)
    for i in range(n):
        arr = np.full((20 + i % 3, 20 + i % 3, 3), i * 10 % 256, np.uint8)   # constant colour encodes the index
        arr[..., 1] = (i * 7) % 256
        Image.fromarray(arr).save(raw / f"img_{i:03d}.jpg", quality=100)
    jsonl = create_random_dataset(str(raw / "*.jpg"), tmp_path / "paired", max_images_per_folder=10, words_per_caption=4)
    paths = create_paired_dataset(jsonl, "T", tmp_path / "rec", examples_per_tfrecord=per_file, tokenizer=_Tok())
    del rng
    return paths


def _params(tmp_path, **kw):
    p = {"dataset": {"train_path": str(tmp_path / "rec" / "T_*.tfrecords"), "eval_path": str(tmp_path / "rec" / "T_*.tfrecords"),
                     "image_size": 16, "tfrecords": True},
         "batch_size": 4, "n_channels": 3, "text_seq_len": 8, "padding_id": 50, "text_vocab_size": 51}
    p.update(kw)
    return p


def test_create_paired_dataset_and_interleave(tmp_path):
    paths = _make_dataset(tmp_path)
    assert [os.path.basename(p) for p in paths] == [f"T_{k}.tfrecords" for k in range(5)]
    counts = [len(list(tfr.read_records(p, verify_crc=True))) for p in paths]
    assert counts == [5, 5, 5, 5, 3]
    items = [json.loads(l) for l in open(tmp_path / "paired" / "captions_data.jsonl")]
    ex0 = tfr.decode_example(next(tfr.read_records(paths[0])))
    assert ex0["caption"] == _Tok().encode(items[0]["caption"][0])
    assert ex0["image"][0] == open(tmp_path / "paired" / items[0]["image_path"], "rb").read()
    # 4-way round-robin interleave: a,b,c,d,a,b,c,d,... then the 5th file joins when a slot frees up
    order = [tfr.decode_example(r)["caption"] for r in input_fns._interleave_records(paths, 4)]
    per_file = [[tfr.decode_example(r)["caption"] for r in tfr.read_records(p)] for p in paths]
    assert order[:8] == [per_file[0][0], per_file[1][0], per_file[2][0], per_file[3][0],
                         per_file[0][1], per_file[1][1], per_file[2][1], per_file[3][1]]
    assert len(order) == 23 and sorted(map(tuple, order)) == sorted(tuple(c) for f in per_file for c in f)


def test_dalle_input_fn_eval_is_deterministic_and_drops_remainder(tmp_path):
    _make_dataset(tmp_path)
    it = input_fns.dalle_input_fn(_params(tmp_path), eval=True)
    batches = [next(it) for _ in range(7)]
    it.close()
    img, cap = batches[0]
    assert tuple(img.shape) == (4, 16, 16, 3) and str(img.dtype) == "torch.float32"
    assert tuple(cap.shape) == (4, 8) and str(cap.dtype) == "torch.int32"
    assert float(img.min()) >= -1 and float(img.max()) <= 1
    # 23 elements, batch 4, drop_remainder -> 5 batches per epoch, then the epoch repeats exactly (no shuffle in eval)
    assert all(bool((batches[0][k] == batches[5][k]).all()) for k in (0, 1))
    assert all(bool((batches[1][k] == batches[6][k]).all()) for k in (0, 1))
    assert not bool((batches[0][0] == batches[1][0]).all())


def test_dalle_input_fn_train_shuffles_and_dp_shards_are_disjoint(tmp_path):
    _make_dataset(tmp_path)
    seen = []
    for rank in range(2):
        it = input_fns.dalle_input_fn(_params(tmp_path, dp_rank=rank, dp_world=2, batch_size=2), eval=False)
        imgs = [next(it)[0] for _ in range(5)]     # 5 batches of 2 = one epoch of this rank's 11/12 elements
        it.close()
        keys = set()
        for b in imgs:
            for im in b:
                keys.add((round(float(im[8, 8, 0]) * 127.5 + 127.5), round(float(im[8, 8, 1]) * 127.5 + 127.5)))
        seen.append(keys)
    assert len(seen[0]) == 10 and len(seen[1]) == 10
    assert not (seen[0] & seen[1])


def test_vae_input_fn_jpeg_glob_and_tfrecords(tmp_path):
    _make_dataset(tmp_path)
    p = _params(tmp_path)
    a, b = next(input_fns.vae_input_fn(p, eval=True))
    assert a is b or bool((a == b).all())
    assert tuple(a.shape) == (4, 16, 16, 3)
    p2 = _params(tmp_path)
    p2["dataset"] = {"train_path": str(tmp_path / "raw" / "*.jpg"), "eval_path": str(tmp_path / "raw" / "*.jpg"), "image_size": 16}
    a2, _ = next(input_fns.vae_input_fn(p2, eval=True))
    assert tuple(a2.shape) == (4, 16, 16, 3)
    # sorted glob, no shuffle in eval: first image is img_000 (constant colour 0 in R)
    assert abs(float(a2[0, 8, 8, 0]) * 127.5 + 127.5) < 3


def test_missing_and_too_small_datasets_fail_loudly(tmp_path):
    p = _params(tmp_path)
    with pytest.raises(FileNotFoundError):
        input_fns.dalle_input_fn(p)
    _make_dataset(tmp_path, n=3, per_file=5)
    with pytest.raises(ValueError):
        next(input_fns.dalle_input_fn(_params(tmp_path, batch_size=4), eval=True))


# ---------------------------------------------------------------- property tests (hypothesis)

from hypothesis import given, settings, strategies as st  # noqa: E402

_i64 = st.integers(min_value=-(2 ** 63), max_value=2 ** 63 - 1)


@settings(max_examples=60, deadline=None)
@given(st.dictionaries(st.text(st.characters(min_codepoint=33, max_codepoint=126), min_size=1, max_size=12),
                       st.one_of(st.lists(st.binary(max_size=40), min_size=1, max_size=4),
                                 st.lists(_i64, min_size=1, max_size=20)),
                       max_size=5))
def test_example_codec_roundtrip_property(features):
    """any mix of bytes / int64 features survives encode -> decode, and the real protobuf runtime reads the same values."""
    buf = tfr.encode_example(features)
    assert tfr.decode_example(buf) == {k: list(v) for k, v in features.items()}
    ex = _example_class()()
    ex.ParseFromString(buf)
    for k, v in features.items():
        f = ex.features.feature[k]
        got = list(f.bytes_list.value) if isinstance(v[0], bytes) else list(f.int64_list.value)
        assert got == list(v)


@settings(max_examples=25, deadline=None)
@given(st.lists(st.binary(max_size=300), max_size=12))
def test_record_container_roundtrip_property(tmp_path_factory, records):
    p = str(tmp_path_factory.mktemp("rec") / "x.tfrecords")
    tfr.write_records(p, records)
    assert list(tfr.read_records(p, verify_crc=True)) == records


@settings(max_examples=40, deadline=None)
@given(st.lists(st.integers(min_value=0, max_value=50256), max_size=600), st.integers(min_value=1, max_value=300))
def test_truncate_or_pad_label_property(ids, T):
    out = input_fns.truncate_or_pad_label(ids, {"text_seq_len": T, "padding_id": 50257})
    assert out.shape == (T,) and out.dtype == np.int32
    n = min(len(ids), T)
    assert out[:n].tolist() == ids[:n] and (out[n:] == 50257).all()


# ---------------------------------------------------------------- producer lifetime (round-2 regression: SIGABRT at interpreter exit)

def test_prefetch_close_joins_the_producer_and_is_idempotent(tmp_path):
    import threading
    input_fns._close_live_prefetchers()     # the atexit hook itself: joins whatever earlier tests left running
    assert not [t for t in threading.enumerate() if t.name.startswith(("dalle-prefetch", "dalle-decode"))]
    _make_dataset(tmp_path)
    it = input_fns.dalle_input_fn(_params(tmp_path), eval=False)
    next(it)
    assert it._t.is_alive() and it in input_fns._LIVE_PREFETCH
    it.close()
    assert not it._t.is_alive() and it not in input_fns._LIVE_PREFETCH
    assert not [t for t in threading.enumerate() if t.name.startswith(("dalle-prefetch", "dalle-decode"))]
    it.close()                       # idempotent
    with pytest.raises(StopIteration):
        next(it)
    with input_fns.vae_input_fn(_params(tmp_path), eval=True) as it2:    # context-manager form
        next(it2)
    assert not it2._t.is_alive()


def test_estimator_close_and_context_manager_join_the_train_stream(tmp_path):
    from src.estimator import Estimator
    _make_dataset(tmp_path)

    class Spec:
        loss, training_hooks, host_call = 0.0, [], None

        def __init__(self):
            self.n = 0

        def train_op(self):
            self.n += 1
            return self.n
    spec = Spec()
    with Estimator(lambda f, l, mode, params: spec, None, _params(tmp_path), log_every=10 ** 9) as est:
        est.train(lambda params: input_fns.dalle_input_fn(params, eval=False), max_steps=3)
        t = est._train_it._t
        assert t.is_alive()
        est.evaluate(lambda params: input_fns.dalle_input_fn(params, eval=True), steps=2)
    assert not t.is_alive() and est._train_it is None


def test_interpreter_exit_is_clean_with_a_live_producer(tmp_path):
    """A program that never calls close() must still exit with rc 0: the atexit hook stops and joins every live producer
    before interpreter finalisation can unwind it by force (GPUTEST_r02: 'terminate called without an active exception')."""
    import subprocess
    import sys
    _make_dataset(tmp_path)
    prog = (
        "import sys, json; sys.path[:0] = %r\n"
        "from src import input_fns\n"
        "p = json.loads(%r)\n"
        "it = input_fns.dalle_input_fn(p, eval=False)\n"
        "next(it); next(it)\n"
        "print('done', flush=True)\n"       # exits with the producer mid-epoch
    ) % ([os.path.dirname(os.path.dirname(os.path.abspath(input_fns.__file__)))], json.dumps(_params(tmp_path)))
    for _ in range(6):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "done" in r.stdout, (r.returncode, r.stderr[-2000:])
        assert "terminate called" not in r.stderr


def test_dp_shard_skips_foreign_records_without_reading_them(tmp_path, monkeypatch):
    paths = _make_dataset(tmp_path)
    full = list(input_fns._interleave_records(paths, 4))
    crcs = []
    real = tfr.masked_crc32c
    monkeypatch.setattr(tfr, "masked_crc32c", lambda d: (crcs.append(len(d)), real(d))[1])
    mine = list(input_fns._interleave_records(paths, 4, rank=1, world=3))
    assert mine == full[1::3]
    payload_crcs = [n for n in crcs if n != 8]
    assert len(payload_crcs) == len(mine)            # one payload CRC per OWNED record; the others were seeked past
    assert len([n for n in crcs if n == 8]) == len(full)   # every length header is still verified
