"""TFRecord container + tf.train.Example wire format, dependency-free (the reference relies on TensorFlow for both:
writer src/data/create_tfrecords.py:34-56, reader src/input_fns.py:41-66).

Record framing (tensorflow/core/lib/io/record_writer.cc): u64 length | u32 masked_crc32c(length) | data |
u32 masked_crc32c(data), little endian; mask(c) = ((c >> 15 | c << 17) + 0xa282ead8) mod 2^32, CRC-32C (Castagnoli).
Example schema (tensorflow/core/example/{example,feature}.proto):
  Example{1: Features}  Features{1: map<string, Feature>}  Feature{1: BytesList | 2: FloatList | 3: Int64List}
  BytesList{1: repeated bytes}  FloatList{1: repeated float [packed]}  Int64List{1: repeated int64 [packed]}
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Union

import numpy as np

_MASK_DELTA = 0xA282EAD8


def _make_table():
    poly = 0x82F63B78
    tab = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if (c & 1) else (c >> 1)
        tab[i] = c
    return tab


_TABLE = _make_table()


def _crc32c_py(data: bytes) -> int:
    """bytewise table CRC (4 MB/s): only for tiny inputs (the 8-byte length header) or when the native library is absent
    (dataset-writing tools on a machine without the build)"""
    c = 0xFFFFFFFF
    tab = _TABLE
    for b in data:
        c = int(tab[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


_native = None


def _native_crc():
    """dmi_crc32c from libdalle_hip.so (slicing-by-8 on the host, ~2 GB/s): a per-byte Python loop caps a decode thread
    at ~100 images/s and holds the GIL the kernel-launching thread needs."""
    global _native
    if _native is None:
        try:
            from dalle_hip import lib
            _native = lib().dmi_crc32c
        except Exception:
            _native = False
    return _native


def crc32c(data: bytes) -> int:
    fn = _native_crc() if len(data) > 64 else None
    if fn:
        return int(fn(bytes(data), len(data)))
    return _crc32c_py(data)


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


# ------------------------------------------------------------------ container

def write_records(path: str, records: List[bytes]):
    with open(path, "wb") as f:
        for data in records:
            hdr = struct.pack("<Q", len(data))
            f.write(hdr)
            f.write(struct.pack("<I", masked_crc32c(hdr)))
            f.write(data)
            f.write(struct.pack("<I", masked_crc32c(data)))


def read_records(path: str, verify_crc: bool = True, want=None) -> Iterator[bytes]:
    """Yields each record's payload.  `want()` (optional) is asked once per record, just before its payload would be
    read: when it returns False the payload is seeked past (no read, no CRC) and None is yielded in its place, so a
    data-parallel rank pays only for the records it owns."""
    with open(path, "rb") as f:
        while True:
            hdr = f.read(12)
            if len(hdr) == 0:
                return
            if len(hdr) < 12:
                raise IOError(f"{path}: truncated record header")
            n, hcrc = struct.unpack("<QI", hdr)
            if verify_crc and hcrc != masked_crc32c(hdr[:8]):
                raise IOError(f"{path}: corrupt record length")
            if want is not None and not want():
                f.seek(n + 4, 1)
                yield None
                continue
            data = f.read(n)
            if len(data) < n:
                raise IOError(f"{path}: truncated record")
            raw = f.read(4)
            if len(raw) < 4:
                raise IOError(f"{path}: truncated record")
            (dcrc,) = struct.unpack("<I", raw)
            if verify_crc and dcrc != masked_crc32c(data):
                raise IOError(f"{path}: corrupt record data")
            yield data


# ------------------------------------------------------------------ protobuf wire helpers

def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1  # int64 two's complement
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf: bytes, pos: int):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not (b & 0x80):
            return val, pos
        shift += 7


def _ld(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _fields(buf: bytes):
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            val = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield field, wt, val


FeatureValue = Union[List[bytes], List[int], List[float]]


def encode_example(features: Dict[str, FeatureValue]) -> bytes:
    """bytes values -> BytesList, ints -> Int64List (packed), floats -> FloatList (packed)."""
    entries = b""
    for name, vals in features.items():
        vals = list(vals)
        if vals and isinstance(vals[0], (bytes, bytearray)):
            feat = _ld(1, b"".join(_ld(1, bytes(v)) for v in vals))
        elif vals and isinstance(vals[0], float):
            feat = _ld(2, _ld(1, struct.pack(f"<{len(vals)}f", *vals)))
        else:
            feat = _ld(3, _ld(1, b"".join(_varint(int(v)) for v in vals)))
        entry = _ld(1, name.encode()) + _ld(2, feat)
        entries += _ld(1, entry)
    return _ld(1, entries)


def decode_example(buf: bytes) -> Dict[str, FeatureValue]:
    out: Dict[str, FeatureValue] = {}
    for f, _, features in _fields(buf):
        if f != 1:
            continue
        for f2, _, entry in _fields(features):
            if f2 != 1:
                continue
            name, feat = None, b""
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    name = v.decode()
                elif f3 == 2:
                    feat = v
            vals: FeatureValue = []
            for kind, _, lst in _fields(feat):
                if kind == 1:      # BytesList
                    vals = [bytes(v) for ff, _, v in _fields(lst) if ff == 1]
                elif kind == 2:    # FloatList
                    fl: List[float] = []
                    for ff, wt, v in _fields(lst):
                        if ff == 1 and wt == 2:
                            fl += list(struct.unpack(f"<{len(v) // 4}f", v))
                        elif ff == 1 and wt == 5:
                            fl.append(struct.unpack("<f", v)[0])
                    vals = fl
                elif kind == 3:    # Int64List
                    il: List[int] = []
                    for ff, wt, v in _fields(lst):
                        if ff != 1:
                            continue
                        if wt == 2:
                            p = 0
                            while p < len(v):
                                x, p = _read_varint(v, p)
                                il.append(x - (1 << 64) if x >= (1 << 63) else x)
                        else:
                            il.append(v - (1 << 64) if v >= (1 << 63) else v)
                    vals = il
            if name is not None:
                out[name] = vals
    return out
