"""Paired image/caption TFRecord writer (reference src/data/create_tfrecords.py:16-175), TensorFlow-free: records
are framed and serialised by data/tfrecord.py, byte-compatible with tf.io.TFRecordWriter + tf.train.Example.

Differences from the reference, on purpose:
  * shards are named {name}_0, {name}_1, ... (the reference re-opens {name}_{tfrecord_count} *before* bumping the
    counter, create_tfrecords.py:152-155, so its second shard overwrites its first);
  * `create_random_dataset` draws captions from a built-in word list (the reference downloads one; no network here);
  * `reencode` uses PIL at quality 94 instead of cv2.
`caption = tokenizer.encode(item["caption"][0])` (create_tfrecords.py:163) is kept: captions are stored as a list
of alternatives in the jsonl and the first one is used.
"""
import glob
import io
import json
import os
import random
import shutil
from pathlib import Path, PurePath

from .tfrecord import encode_example, masked_crc32c
from .tokenizer_utils import get_tokenizer
import struct


def dump_jsonl(data, output_path, append=False):
    mode = 'a+' if append else 'w'
    with open(output_path, mode, encoding='utf-8') as f:
        for line in data:
            f.write(json.dumps(line, ensure_ascii=False) + '\n')


def load_jsonl(input_path):
    data = []
    with open(input_path, 'r', encoding='utf-8') as f:
        for line in f:
            data.append(json.loads(line.rstrip('\n|\r')))
    return data


def serialize_example(image, caption):
    """{'image': BytesList[jpeg], 'caption': Int64List[ids]} -> serialized tf.train.Example."""
    return encode_example({"image": [bytes(image)], "caption": [int(c) for c in caption]})


class TFRecordWriter:
    def __init__(self, path):
        self._f = open(path, "wb")

    def write(self, data: bytes):
        hdr = struct.pack("<Q", len(data))
        self._f.write(hdr + struct.pack("<I", masked_crc32c(hdr)) + data + struct.pack("<I", masked_crc32c(data)))

    def close(self):
        self._f.close()


_WORDS = ("a an the red green blue small large old young cat dog bird horse boat plane tree house street "
          "river mountain sitting standing running flying on under near behind with and of in photo painting").split()


def create_random_dataset(path_to_images, out_dir, max_images_per_folder=1000, words_per_caption=50, seed=0):
    """Paired image/text folder with random captions in the layout create_paired_dataset reads (testing aid)."""
    rnd = random.Random(seed)
    out_dir = Path(out_dir)
    jsonl_path = out_dir / "captions_data.jsonl"
    os.makedirs(out_dir, exist_ok=True)
    images = sorted(glob.glob(path_to_images))
    folder_count = 0
    sub_folder = None
    for i, image in enumerate(images):
        if i % max_images_per_folder == 0:
            sub_folder = out_dir / str(folder_count)
            os.makedirs(sub_folder, exist_ok=True)
            folder_count += 1
        image = Path(image)
        caption = " ".join(rnd.choice(_WORDS) for _ in range(words_per_caption))
        data = {"caption": [caption], "image_path": str(sub_folder.relative_to(out_dir) / image.name)}
        shutil.copy(image, sub_folder)
        dump_jsonl([data], jsonl_path, append=True)
    return jsonl_path


def create_paired_dataset(path_to_jsonl, name, out_dir, examples_per_tfrecord=1000, tokenizer=None, reencode=False):
    """jsonl of {"image_path": relative path, "caption": [text, ...]} -> {out_dir}/{name}_{k}.tfrecords."""
    if tokenizer is None:
        tokenizer = get_tokenizer()
    out_dir = Path(out_dir)
    os.makedirs(out_dir, exist_ok=True)
    if isinstance(path_to_jsonl, (PurePath, str)):
        path_to_jsonl = [path_to_jsonl]
    if not isinstance(path_to_jsonl, list):
        raise TypeError("path_to_jsonl type not recognized, should be str, path, or list")
    tfrecord_count, example_count = 0, 0
    paths = [str(out_dir / f"{name}_{tfrecord_count}.tfrecords")]
    writer = TFRecordWriter(paths[-1])
    for path in path_to_jsonl:
        path = Path(path)
        for item in load_jsonl(path):
            if example_count % examples_per_tfrecord == 0 and example_count != 0:
                writer.close()
                tfrecord_count += 1
                paths.append(str(out_dir / f"{name}_{tfrecord_count}.tfrecords"))
                writer = TFRecordWriter(paths[-1])
            image_path = path.parent / item["image_path"]
            if reencode:
                from PIL import Image
                buf = io.BytesIO()
                Image.open(image_path).convert("RGB").save(buf, format="JPEG", quality=94)
                img = buf.getvalue()
            else:
                with open(image_path, "rb") as f:
                    img = f.read()
            caption = tokenizer.encode(item["caption"][0])
            writer.write(serialize_example(img, caption))
            example_count += 1
    writer.close()
    return paths
