"""Reader / writer for the TensorFlow "tensor bundle" (V2 checkpoint) format.

The reference restores its weights with ``tf.train.Saver.restore`` (reference
``network.py:47`` builds the Saver, ``network.py:122`` restores) from
``final_model/roomnet.index`` + ``roomnet.data-00000-of-00001``.  TensorFlow is
not available on the MI355X hosts, so this module re-implements the on-disk
format from its published layout:

* ``<prefix>.index`` is a LevelDB-style SSTable: data blocks of prefix-compressed
  key/value entries, one index block, a 48-byte footer ending in the magic
  ``0xdb4775248b80fb57``.  Every block is followed by a 1-byte compression type
  (0 = none) and a masked CRC32C of block+type.
* key ``""`` maps to a ``BundleHeaderProto``; every other key is a variable name
  mapping to a ``BundleEntryProto`` {dtype, shape, shard_id, offset, size, crc32c}.
* ``<prefix>.data-XXXXX-of-YYYYY`` holds the raw little-endian tensor bytes.

Only what the RoomNet inference path needs is supported (float32/int32/int64
tensors, uncompressed blocks, no slices); anything else raises ``BundleError``.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from typing import Dict, Iterator, List, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_FOOTER_LEN = 48
_BLOCK_TRAILER = 5
_MASK_DELTA = 0xA282EAD8

# DataType enum values of tensorflow/core/framework/types.proto that we accept
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8")}
_DTYPE_IDS = {v: k for k, v in _DTYPES.items()}


class BundleError(IOError):
    """Raised for a missing, corrupt or unsupported checkpoint bundle."""


# --------------------------------------------------------------------- crc32c
def _make_crc_table() -> np.ndarray:
    poly = 0x82F63B78
    tbl = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (poly if c & 1 else 0)
        tbl[i] = c
    return tbl


_CRC_TABLE = _make_crc_table()
_CRC_TABLE_LIST = [int(x) for x in _CRC_TABLE]


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C (Castagnoli), bit-reflected, as used by LevelDB / TensorFlow."""
    c = crc ^ 0xFFFFFFFF
    tbl = _CRC_TABLE_LIST
    for b in data:
        c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked: int) -> int:
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------ varint / proto
def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        if pos >= len(buf):
            raise BundleError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise BundleError("varint too long")


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _proto_fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """Yield (field_number, wire_type, value) for a protobuf message."""
    pos = 0
    n = len(buf)
    while pos < n:
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _get_varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise BundleError("unsupported protobuf wire type %d" % wt)
        yield field, wt, val


@dataclass
class BundleEntry:
    """One ``BundleEntryProto``: where a tensor lives in the data shard."""
    name: str
    dtype: np.dtype
    shape: Tuple[int, ...]
    shard_id: int
    offset: int
    size: int
    crc32c: int  # masked, as stored

    @property
    def num_elements(self) -> int:
        n = 1
        for d in self.shape:
            n *= d
        return n


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims: List[int] = []
    for field, wt, val in _proto_fields(buf):
        if field == 2 and wt == 2:  # TensorShapeProto.Dim
            size = 0
            for f2, _w2, v2 in _proto_fields(val):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(int(size))
        elif field == 3 and val:
            raise BundleError("unknown-rank tensor in bundle")
    return tuple(dims)


def _parse_entry(name: str, buf: bytes) -> BundleEntry:
    dtype_id, shape, shard, offset, size, crc = 0, (), 0, 0, 0, 0
    for field, wt, val in _proto_fields(buf):
        if field == 1:
            dtype_id = val
        elif field == 2:
            shape = _parse_shape(val)
        elif field == 3:
            shard = val
        elif field == 4:
            offset = val
        elif field == 5:
            size = val
        elif field == 6:
            crc = val
        elif field == 7:
            raise BundleError("sliced tensor %r is not supported" % name)
    if dtype_id not in _DTYPES:
        raise BundleError("tensor %r has unsupported dtype id %d" % (name, dtype_id))
    return BundleEntry(name, _DTYPES[dtype_id], shape, shard, offset, size, crc)


# ------------------------------------------------------------------- sstable
def _read_block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
    end = offset + size
    if end + _BLOCK_TRAILER > len(buf):
        raise BundleError("block handle out of range")
    block = buf[offset:end]
    ctype = buf[end]
    if verify:
        stored = struct.unpack_from("<I", buf, end + 1)[0]
        if unmask_crc(stored) != crc32c(buf[offset:end + 1]):
            raise BundleError("index block checksum mismatch at offset %d" % offset)
    if ctype != 0:
        raise BundleError("compressed index blocks are not supported (type %d)" % ctype)
    return block


def _block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise BundleError("block too small")
    num_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * num_restarts
    if limit < 0:
        raise BundleError("bad restart array")
    pos = 0
    key = b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key):
            raise BundleError("bad key prefix length")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        value = block[pos:pos + vlen]
        pos += vlen
        yield key, value


def _read_index_file(path: str, verify: bool = True) -> Tuple[Dict[str, int], Dict[str, BundleEntry]]:
    try:
        with open(path, "rb") as f:
            buf = f.read()
    except OSError as e:
        raise BundleError("cannot open checkpoint index %r: %s" % (path, e)) from e
    if len(buf) < _FOOTER_LEN:
        raise BundleError("%r is too short to be a checkpoint index" % path)
    footer = buf[-_FOOTER_LEN:]
    if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
        raise BundleError("%r: bad table magic (not a TF checkpoint index)" % path)
    pos = 0
    _mi_off, pos = _get_varint(footer, pos)
    _mi_size, pos = _get_varint(footer, pos)
    idx_off, pos = _get_varint(footer, pos)
    idx_size, pos = _get_varint(footer, pos)
    index_block = _read_block(buf, idx_off, idx_size, verify)
    header: Dict[str, int] = {}
    entries: Dict[str, BundleEntry] = {}
    for _k, handle in _block_entries(index_block):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        for key, value in _block_entries(_read_block(buf, boff, bsize, verify)):
            if key == b"":
                for field, _wt, val in _proto_fields(value):
                    if field == 1:
                        header["num_shards"] = val
                    elif field == 2:
                        header["endianness"] = val
                    elif field == 3:
                        for f2, _w2, v2 in _proto_fields(val):
                            if f2 == 1:
                                header["producer"] = v2
                header.setdefault("num_shards", 0)
                header.setdefault("endianness", 0)
            else:
                name = key.decode("utf-8")
                entries[name] = _parse_entry(name, value)
    if not header:
        raise BundleError("%r has no bundle header" % path)
    if header["endianness"] != 0:
        raise BundleError("big-endian bundles are not supported")
    return header, entries


class BundleReader:
    """Random access to the tensors of ``<prefix>.index`` / ``<prefix>.data-*``."""

    def __init__(self, prefix: str, verify_index: bool = True):
        self.prefix = prefix
        self.header, self.entries = _read_index_file(prefix + ".index", verify_index)
        self._shards: Dict[int, bytes] = {}

    def keys(self) -> List[str]:
        return sorted(self.entries)

    def __contains__(self, name: str) -> bool:
        return name in self.entries

    def _shard(self, shard_id: int) -> bytes:
        if shard_id not in self._shards:
            path = "%s.data-%05d-of-%05d" % (self.prefix, shard_id, self.header["num_shards"])
            try:
                with open(path, "rb") as f:
                    self._shards[shard_id] = f.read()
            except OSError as e:
                raise BundleError("cannot open checkpoint data %r: %s" % (path, e)) from e
        return self._shards[shard_id]

    def raw(self, name: str) -> bytes:
        if name not in self.entries:
            raise KeyError("tensor %r not found in checkpoint %r" % (name, self.prefix))
        e = self.entries[name]
        data = self._shard(e.shard_id)
        if e.offset + e.size > len(data):
            raise BundleError("tensor %r extends past the end of its data shard" % name)
        return data[e.offset:e.offset + e.size]

    def verify(self, name: str) -> bool:
        return mask_crc(crc32c(self.raw(name))) == self.entries[name].crc32c

    def get(self, name: str, verify: bool = True) -> np.ndarray:
        e = self.entries[name]
        raw = self.raw(name)
        if len(raw) != e.num_elements * e.dtype.itemsize:
            raise BundleError("tensor %r: size %d does not match shape %s" % (name, len(raw), e.shape))
        if verify and mask_crc(crc32c(raw)) != e.crc32c:
            raise BundleError("tensor %r: CRC32C mismatch (corrupt checkpoint)" % name)
        return np.frombuffer(raw, dtype=e.dtype).reshape(e.shape).copy()

    def load_all(self, verify: bool = True) -> Dict[str, np.ndarray]:
        return {k: self.get(k, verify) for k in self.keys()}


def checkpoint_exists(prefix: str) -> bool:
    return os.path.isfile(prefix + ".index")


# -------------------------------------------------------------------- writer
def _put_field(field: int, wt: int, payload: bytes) -> bytes:
    return _put_varint((field << 3) | wt) + payload


def _encode_entry(e: BundleEntry) -> bytes:
    shape = b""
    for d in e.shape:
        dim = _put_field(1, 0, _put_varint(d)) if d else b""
        shape += _put_field(2, 2, _put_varint(len(dim)) + dim)
    out = _put_field(1, 0, _put_varint(_DTYPE_IDS[np.dtype(e.dtype)]))
    out += _put_field(2, 2, _put_varint(len(shape)) + shape)
    if e.shard_id:
        out += _put_field(3, 0, _put_varint(e.shard_id))
    if e.offset:
        out += _put_field(4, 0, _put_varint(e.offset))
    out += _put_field(5, 0, _put_varint(e.size))
    out += _put_field(6, 5, struct.pack("<I", e.crc32c))
    return out


def _build_block(items: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out = bytearray()
    restarts: List[int] = []
    last = b""
    for i, (k, v) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            for a, b in zip(last, k):
                if a != b:
                    break
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v))
        out += k[shared:] + v
        last = k
    if not restarts:
        restarts.append(0)
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _with_trailer(block: bytes) -> bytes:
    return block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00")))


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """Write ``tensors`` as a single-shard bundle readable by ``BundleReader``
    (and laid out like ``tf.train.Saver.save`` output: tensors back-to-back in
    lexicographic key order; reference ``network.py:94-97``)."""
    data = bytearray()
    items: List[Tuple[bytes, bytes]] = []
    header = _put_field(1, 0, _put_varint(1)) + _put_field(3, 2, b"\x02" + _put_field(1, 0, _put_varint(1)))
    items.append((b"", header))
    for name in sorted(tensors):
        arr = np.ascontiguousarray(tensors[name])
        dt = arr.dtype.newbyteorder("<")
        if dt not in _DTYPE_IDS:
            raise BundleError("unsupported dtype %s for %r" % (arr.dtype, name))
        raw = arr.astype(dt, copy=False).tobytes()
        e = BundleEntry(name, dt, tuple(arr.shape), 0, len(data), len(raw), mask_crc(crc32c(raw)))
        data += raw
        items.append((name.encode("utf-8"), _encode_entry(e)))
    data_block = _build_block(items)
    out = bytearray(_with_trailer(data_block))
    metaindex_off = len(out)
    metaindex = _build_block([])
    out += _with_trailer(metaindex)
    index_off = len(out)
    sep = items[-1][0] + b"\x00" if len(items) > 1 else b"\x00"
    index_block = _build_block([(sep, _put_varint(0) + _put_varint(len(data_block)))])
    out += _with_trailer(index_block)
    footer = _put_varint(metaindex_off) + _put_varint(len(metaindex))
    footer += _put_varint(index_off) + _put_varint(len(index_block))
    footer = footer.ljust(40, b"\x00") + struct.pack("<Q", TABLE_MAGIC)
    out += footer
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
