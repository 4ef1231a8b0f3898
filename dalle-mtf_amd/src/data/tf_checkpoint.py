"""TensorFlow V2 checkpoint ("tensor bundle") reader / writer without TensorFlow ("optional importer for
TF-checkpoint variable names for cross-framework parity").

The reference saves and restores its variables with `tf.train.Saver` (src/model_fns.py:11-32 restores the VAE's variables
by name under scope `vae/`; :204-229 saves the DALL-E model's): files `<prefix>.index` + `<prefix>.data-0000N-of-0000M`.

Format (tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc}, tensorflow/core/lib/io/{table_builder,format,block}.cc,
tensorflow/core/protobuf/tensor_bundle.proto):
  * `.index` is an immutable LevelDB-style table: data blocks of prefix-compressed (key, value) entries
    [varint32 shared | varint32 non_shared | varint32 value_len | key suffix | value] followed by the block's restart array
    (uint32 offsets, uint32 count) and a 5-byte trailer (1 byte compression type, 4 bytes masked CRC-32C of block + type);
    an index block maps a separator key >= the last key of each data block to its BlockHandle (varint64 offset, size); the
    48-byte footer holds the metaindex and index handles and the magic 0xdb4775248b80fb57.
  * key "" -> BundleHeaderProto {1: num_shards, 2: endianness, 3: VersionDef}; every other key is a tensor name ->
    BundleEntryProto {1: dtype, 2: TensorShapeProto{2: dim{1: size}}, 3: shard_id, 4: offset, 5: size, 6: fixed32 crc32c}.
  * `.data-*` shards hold the raw little-endian tensor bytes at [offset, offset + size).
Snappy-compressed blocks (type 1) are not supported -- BundleWriter writes uncompressed tables.

STATUS: written from the format's published definition and verified only by round trip (tests/test_host_logic.py) plus the
format's known constants; it has NOT been checked against a file written by TensorFlow, because TensorFlow cannot be installed
here.  The reader verifies every block and tensor CRC, so a layout misunderstanding fails loudly rather than loading noise."""
from __future__ import annotations

import os
import warnings
import struct
from collections import OrderedDict
from typing import Dict, Iterator, List, Tuple

import numpy as np

from .tfrecord import _fields, _ld, _read_varint, _varint, masked_crc32c

TABLE_MAGIC = 0xDB4775248B80FB57
# tensorflow/core/framework/types.proto
DT = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8"), 14: "bfloat16", 19: np.dtype("<f2")}
DT_CODE = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9, np.dtype("float16"): 19}


# ------------------------------------------------------------------ table (.index)

def _read_block(buf: bytes, offset: int, size: int, verify=True) -> bytes:
    block = buf[offset:offset + size]
    ctype = buf[offset + size]
    (crc,) = struct.unpack("<I", buf[offset + size + 1:offset + size + 5])
    if verify and crc != masked_crc32c(block + bytes([ctype])):
        raise IOError("tensor bundle index: block checksum mismatch")
    if ctype != 0:
        raise NotImplementedError("tensor bundle index: compressed table blocks are not supported")
    return block


def _block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    (nrestart,) = struct.unpack("<I", block[-4:])
    end = len(block) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _handle(buf: bytes, pos: int):
    off, pos = _read_varint(buf, pos)
    size, pos = _read_varint(buf, pos)
    return off, size, pos


def read_table(path: str) -> "OrderedDict[bytes, bytes]":
    buf = open(path, "rb").read()
    if len(buf) < 48:
        raise IOError(f"{path}: too short for a table footer")
    footer = buf[-48:]
    (magic,) = struct.unpack("<Q", footer[40:])
    if magic != TABLE_MAGIC:
        raise IOError(f"{path}: bad table magic {magic:#x}")
    _, _, p = _handle(footer, 0)              # metaindex (unused)
    ioff, isize, _ = _handle(footer, p)
    out: "OrderedDict[bytes, bytes]" = OrderedDict()
    for _, hv in _block_entries(_read_block(buf, ioff, isize)):
        boff, bsize, _ = _handle(hv, 0)
        for k, v in _block_entries(_read_block(buf, boff, bsize)):
            out[k] = v
    return out


def _build_block(entries: List[Tuple[bytes, bytes]], restart_interval=16) -> bytes:
    out, restarts, last = bytearray(), [], b""
    for n, (k, v) in enumerate(entries):
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    out += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path: str, items: "OrderedDict[bytes, bytes]", block_size=4096):
    keys = sorted(items)
    f = bytearray()
    index: List[Tuple[bytes, bytes]] = []

    def emit(block: bytes) -> bytes:
        off = len(f)
        f.extend(block + b"\x00" + struct.pack("<I", masked_crc32c(block + b"\x00")))
        return _varint(off) + _varint(len(block))

    cur: List[Tuple[bytes, bytes]] = []
    size = 0
    for k in keys:
        cur.append((k, items[k]))
        size += len(k) + len(items[k]) + 6
        if size >= block_size:
            index.append((cur[-1][0], emit(_build_block(cur))))
            cur, size = [], 0
    if cur:
        index.append((cur[-1][0], emit(_build_block(cur))))
    meta = emit(_build_block([]))
    idx = emit(_build_block(index, restart_interval=1))
    footer = meta + idx
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    f.extend(footer)
    open(path, "wb").write(bytes(f))


# ------------------------------------------------------------------ bundle

def _shape_proto(shape) -> bytes:
    return b"".join(_ld(2, _varint((1 << 3) | 0) + _varint(int(d))) for d in shape)


def _parse_entry(v: bytes):
    e = dict(dtype=0, shape=[], shard=0, offset=0, size=0, crc=None)
    for f, wt, val in _fields(v):
        if f == 1:
            e["dtype"] = val
        elif f == 2:
            for f2, _, dim in _fields(val):
                if f2 == 2:
                    e["shape"].append(next((x for ff, _, x in _fields(dim) if ff == 1), 0))
        elif f == 3:
            e["shard"] = val
        elif f == 4:
            e["offset"] = val
        elif f == 5:
            e["size"] = val
        elif f == 6:
            (e["crc"],) = struct.unpack("<I", val)
        elif f == 7:
            raise NotImplementedError("tensor bundle: sliced (partitioned) variables are not supported")
    return e


def list_variables(prefix: str) -> "OrderedDict[str, Tuple[str, tuple]]":
    tab = read_table(prefix + ".index")
    out = OrderedDict()
    for k, v in tab.items():
        if k == b"":
            continue
        e = _parse_entry(v)
        out[k.decode()] = (str(DT.get(e["dtype"], e["dtype"])), tuple(e["shape"]))
    return out


def load_checkpoint(prefix: str, verify_crc=True, select=None) -> "OrderedDict[str, np.ndarray]":
    """{variable name: array} of a TF V2 checkpoint `<prefix>.index` / `<prefix>.data-*`.  bfloat16 tensors come back as
    float32 (exactly representable).  `select` (callable name -> bool) restricts the load BEFORE any byte of the data shards is
    read or checksummed (a 1.3B reference checkpoint holds ~15 GB incl. its Adam slots).  Entries of a dtype this reader does
    not decode (DT_STRING: `_CHECKPOINTABLE_OBJECT_GRAPH`, save counters) are skipped with a warning, not fatal."""
    tab = read_table(prefix + ".index")
    nshards = 1
    for f, _, val in _fields(tab.get(b"", b"")):
        if f == 1:
            nshards = val
        if f == 2 and val != 0:
            raise NotImplementedError("tensor bundle: big-endian bundles are not supported")
    shards = {}
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, v in tab.items():
        if k == b"":
            continue
        if select is not None and not select(k.decode()):
            continue
        e = _parse_entry(v)
        dt = DT.get(e["dtype"])
        if dt is None:
            warnings.warn(f"tensor bundle {prefix}: skipping {k.decode()!r} (dtype enum {e['dtype']} is not decoded by this reader)")
            continue
        if e["shard"] not in shards:
            shards[e["shard"]] = open(f"{prefix}.data-{e['shard']:05d}-of-{nshards:05d}", "rb")
        f = shards[e["shard"]]
        f.seek(e["offset"])
        raw = f.read(e["size"])                       # only the selected tensors' bytes are read and checksummed
        if len(raw) != e["size"]:
            raise IOError(f"{prefix}: tensor {k.decode()!r} runs past the end of its data shard")
        if verify_crc and e["crc"] is not None and e["crc"] != masked_crc32c(raw):
            raise IOError(f"{prefix}: tensor {k.decode()!r} fails its checksum")
        if isinstance(dt, str):      # bfloat16: the high half of a float32
            a = (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)
        else:
            a = np.frombuffer(raw, dtype=dt)
        out[k.decode()] = a.reshape(e["shape"]).copy()
    for f in shards.values():
        f.close()
    return out


def save_checkpoint(prefix: str, variables: Dict[str, np.ndarray]):
    """Write {name: array} as a single-shard TF V2 checkpoint (float32 / float64 / int32 / int64 / float16)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items: "OrderedDict[bytes, bytes]" = OrderedDict()
    items[b""] = _varint((1 << 3) | 0) + _varint(1) + _varint((2 << 3) | 0) + _varint(0) + _ld(3, _varint((1 << 3) | 0) + _varint(1))
    data = bytearray()
    for name in sorted(variables):
        a = np.asarray(variables[name], order="C")      # (ascontiguousarray would promote a scalar to shape (1,))
        code = DT_CODE.get(a.dtype)
        if code is None:
            raise NotImplementedError(f"save_checkpoint: dtype {a.dtype} of {name!r}")
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
        entry = _varint((1 << 3) | 0) + _varint(code) + _ld(2, _shape_proto(a.shape))
        if len(data):
            entry += _varint((4 << 3) | 0) + _varint(len(data))
        entry += _varint((5 << 3) | 0) + _varint(len(raw)) + _varint((6 << 3) | 5) + struct.pack("<I", masked_crc32c(raw))
        items[name.encode()] = entry
        data += raw
    open(f"{prefix}.data-00000-of-00001", "wb").write(bytes(data))
    write_table(prefix + ".index", items)


def latest_tf_checkpoint(model_dir: str):
    """tf.train.latest_checkpoint: the `checkpoint` state file's model_checkpoint_path, else the highest-numbered .index."""
    state = os.path.join(model_dir, "checkpoint")
    if os.path.exists(state):
        for line in open(state):
            if line.startswith("model_checkpoint_path:"):
                p = line.split(":", 1)[1].strip().strip('"')
                return p if os.path.isabs(p) else os.path.join(model_dir, p)
    best, step = None, -1
    for f in os.listdir(model_dir) if os.path.isdir(model_dir) else []:
        if f.endswith(".index") and "-" in f:
            try:
                s = int(f[:-6].rsplit("-", 1)[1])
            except ValueError:
                continue
            if s > step:
                best, step = os.path.join(model_dir, f[:-6]), s
    return best


def load_model_variables(prefix: str, scope: str = "") -> "OrderedDict[str, np.ndarray]":
    """The trainable variables of a reference checkpoint under `scope` (e.g. "vae/" -- reference src/model_fns.py:11-32
    restores exactly those), scope stripped, optimizer slots (`<var>/adam_m`, `<var>/adam_v`, Adam's beta powers) and
    `global_step` dropped: what DalleEngine.load_reference_params / DiscreteVAE.load_reference_params take."""
    def wanted(name):      # decided on the index alone: slots and other scopes are never read from the data shards
        if not name.startswith(scope):
            return False
        n = name[len(scope):]
        return not (n == "global_step" or n.endswith(("/adam_m", "/adam_v", "/Adam", "/Adam_1")) or n in ("beta1_power", "beta2_power"))

    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, a in load_checkpoint(prefix, select=wanted).items():
        out[name[len(scope):]] = a.astype(np.float32) if a.dtype.kind == "f" else a
    return out


def global_step_of(prefix: str) -> int:
    v = load_checkpoint(prefix, select=lambda n: n == "global_step").get("global_step")     # one 8-byte entry, not the bundle
    return int(np.asarray(v).reshape(-1)[0]) if v is not None else 0
