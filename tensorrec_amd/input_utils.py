"""
The "standard TensorRec data format" and its TFRecord files, without TensorFlow (tensorrec/input_utils.py:8-142).

The reference moves every matrix around as the 5-tuple ``(row_index int64[nnz], col_index int64[nnz], values
float32[nnz], d0 int64, d1 int64)`` (input_utils.py:30-38) -- as a ``tf.data.Dataset`` in memory, or as ONE
``tf.train.Example`` with the features ``row_index``, ``col_index``, ``values``, ``d0``, ``d1`` inside a TFRecord file
(:86-105, :115-121).  ``TensorRecDataset`` is that 5-tuple as a plain object; the functions below keep the reference's
names and read / write the same files.

File format (restated from the published TFRecord / protobuf specifications, TensorFlow itself is absent here, so files
written by TensorFlow are NOT part of the test fixtures -- parity with TF-written files is pinned only by the format's
known answers: CRC-32C test vectors, the mask formula and hand-assembled records in tests/test_input_utils.py):

    record  := uint64le length | uint32le masked_crc32c(length bytes) | payload | uint32le masked_crc32c(payload)
    masked  := ((crc >> 15 | crc << 17) + 0xa282ead8) mod 2^32
    payload := Example{1: Features{1: map<string, Feature>}},  Feature{2: FloatList{1: packed float32},
               3: Int64List{1: packed varint}}   (unpacked repeated fields are accepted when reading)

Varint coding of the index arrays is vectorised in NumPy; the checksum is the native ``trec_crc32c``.
"""
from __future__ import annotations

import struct

import numpy as np
from scipy import sparse as sp

from . import _native as N

_MASK_DELTA = 0xa282ead8


class TensorRecDataset(object):
    """One matrix in the standard TensorRec format (what a ``tf.data.Dataset`` of input_utils.py:22-40 yields)."""

    def __init__(self, row_index, col_index, values, d0, d1):
        self.row_index = np.ascontiguousarray(row_index, dtype=np.int64)
        self.col_index = np.ascontiguousarray(col_index, dtype=np.int64)
        self.values = np.ascontiguousarray(values, dtype=np.float32)
        self.d0, self.d1 = int(d0), int(d1)
        if not (len(self.row_index) == len(self.col_index) == len(self.values)):
            raise ValueError("row_index, col_index and values must have the same length")

    def as_tuple(self):
        return self.row_index, self.col_index, self.values, self.d0, self.d1

    def to_sparse_matrix(self):
        return sp.coo_matrix((self.values, (self.row_index, self.col_index)), shape=(self.d0, self.d1)).tocsr()


def create_tensorrec_dataset_from_sparse_matrix(sparse_matrix):
    """(input_utils.py:22-40)"""
    if not isinstance(sparse_matrix, sp.coo_matrix):
        sparse_matrix = sp.coo_matrix(sparse_matrix)
    return TensorRecDataset(sparse_matrix.row, sparse_matrix.col, sparse_matrix.data, sparse_matrix.shape[0],
                            sparse_matrix.shape[1])


def get_dimensions_from_tensorrec_dataset(dataset):
    """(input_utils.py:57-71)"""
    return dataset.d0, dataset.d1


# ------------------------------------------------------------------------------------------------ checksums
def crc32c(data, crc=0):
    buf = data if isinstance(data, bytes) else bytes(data)
    return N.load().trec_crc32c(buf, len(buf), crc & 0xffffffff) & 0xffffffff


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xffffffff) + _MASK_DELTA) & 0xffffffff


# ------------------------------------------------------------------------------------------------ protobuf wire format
def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7f
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _encode_varints(values):
    """int64 array -> concatenated base-128 varints (two's complement for negatives, as protobuf int64)."""
    v = np.ascontiguousarray(values, dtype=np.int64).view(np.uint64)
    if v.size == 0:
        return b""
    nbytes = np.ones(v.shape, np.int64)
    for j in range(1, 10):
        nbytes += (v >= np.uint64(1 << (7 * j))).astype(np.int64)
    ends = np.cumsum(nbytes)
    starts = ends - nbytes
    out = np.zeros(int(ends[-1]), np.uint8)
    for j in range(10):
        sel = nbytes > j
        if not sel.any():
            break
        chunk = ((v[sel] >> np.uint64(7 * j)) & np.uint64(0x7f)).astype(np.uint8)
        chunk |= ((nbytes[sel] > j + 1).astype(np.uint8) << 7)
        out[starts[sel] + j] = chunk
    return out.tobytes()


def _decode_varints(buf):
    """Concatenated varints -> int64 array."""
    b = np.frombuffer(buf, dtype=np.uint8)
    if b.size == 0:
        return np.zeros(0, np.int64)
    last = (b & 0x80) == 0
    if not last[-1]:
        raise ValueError("truncated varint")
    group = np.concatenate([[0], np.cumsum(last)[:-1]])              # value index of every byte
    starts = np.concatenate([[0], np.nonzero(last)[0][:-1] + 1])
    pos = np.arange(b.size) - starts[group]
    if pos.max() > 9:
        raise ValueError("varint longer than 10 bytes")
    parts = (b & 0x7f).astype(np.uint64) << (np.uint64(7) * pos.astype(np.uint64))
    return np.add.reduceat(parts, starts).view(np.int64)            # disjoint bit ranges: sum == or


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _feature_int64(values):
    return _ld(3, _ld(1, _encode_varints(values)))                   # Feature.int64_list{value packed}


def _feature_float(values):
    return _ld(2, _ld(1, np.ascontiguousarray(values, dtype='<f4').tobytes()))     # Feature.float_list{value packed}


def _example(features):
    entries = b"".join(_ld(1, _ld(1, name.encode()) + _ld(2, feat)) for name, feat in features)   # map entries
    return _ld(1, entries)                                           # Example.features


def _fields(buf):
    """Iterate (field number, wire type, value) over one message; length-delimited values are memoryviews."""
    mv = memoryview(buf)
    i, n = 0, len(mv)
    while i < n:
        key = 0
        shift = 0
        while True:
            byte = mv[i]
            i += 1
            key |= (byte & 0x7f) << shift
            shift += 7
            if not byte & 0x80:
                break
        field, wt = key >> 3, key & 7
        if wt == 0:
            j = i
            while mv[j] & 0x80:
                j += 1
            yield field, wt, mv[i:j + 1]
            i = j + 1
        elif wt == 2:
            ln = 0
            shift = 0
            while True:
                byte = mv[i]
                i += 1
                ln |= (byte & 0x7f) << shift
                shift += 7
                if not byte & 0x80:
                    break
            yield field, wt, mv[i:i + ln]
            i += ln
        elif wt == 5:
            yield field, wt, mv[i:i + 4]
            i += 4
        elif wt == 1:
            yield field, wt, mv[i:i + 8]
            i += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)


def _parse_feature(buf):
    for field, wt, val in _fields(buf):
        if field == 3:                                               # Int64List
            chunks = [bytes(v) for f, w, v in _fields(val) if f == 1]
            return _decode_varints(b"".join(chunks))
        if field == 2:                                               # FloatList: packed (wt 2) or one float per tag (wt 5)
            chunks = [bytes(v) for f, w, v in _fields(val) if f == 1]
            return np.frombuffer(b"".join(chunks), dtype='<f4').astype(np.float32)
        if field == 1:
            raise ValueError("bytes_list features are not part of the TensorRec record schema")
    return np.zeros(0, np.int64)


def _parse_example(payload):
    out = {}
    for field, _, features in _fields(payload):
        if field != 1:
            continue
        for f2, _, entry in _fields(features):
            if f2 != 1:
                continue
            name, feat = None, None
            for f3, _, val in _fields(entry):
                if f3 == 1:
                    name = bytes(val).decode()
                elif f3 == 2:
                    feat = val
            if name is not None and feat is not None:
                out[name] = _parse_feature(feat)
    return out


# ------------------------------------------------------------------------------------------------ TFRecord files
def write_tfrecord_from_tensorrec_dataset(tfrecord_path, dataset):
    """(input_utils.py:74-106): one record holding one Example with the five features."""
    payload = _example([
        ('row_index', _feature_int64(dataset.row_index)),
        ('col_index', _feature_int64(dataset.col_index)),
        ('values', _feature_float(dataset.values)),
        ('d0', _feature_int64([dataset.d0])),
        ('d1', _feature_int64([dataset.d1])),
    ])
    header = struct.pack('<Q', len(payload))
    with open(tfrecord_path, 'wb') as file:
        file.write(header)
        file.write(struct.pack('<I', masked_crc32c(header)))
        file.write(payload)
        file.write(struct.pack('<I', masked_crc32c(payload)))
    return tfrecord_path


def write_tfrecord_from_sparse_matrix(tfrecord_path, sparse_matrix):
    """(input_utils.py:43-54)"""
    dataset = create_tensorrec_dataset_from_sparse_matrix(sparse_matrix=sparse_matrix)
    return write_tfrecord_from_tensorrec_dataset(tfrecord_path=tfrecord_path, dataset=dataset)


def _read_records(tfrecord_path, verify=True):
    with open(tfrecord_path, 'rb') as file:
        while True:
            header = file.read(8)
            if not header:
                return
            if len(header) != 8:
                raise ValueError("%s: truncated record header" % tfrecord_path)
            (length,) = struct.unpack('<Q', header)
            (hcrc,) = struct.unpack('<I', file.read(4))
            if verify and hcrc != masked_crc32c(header):
                raise ValueError("%s: corrupted record length" % tfrecord_path)
            payload = file.read(length)
            tail = file.read(4)
            if len(payload) != length or len(tail) != 4:
                raise ValueError("%s: truncated record" % tfrecord_path)
            if verify and struct.unpack('<I', tail)[0] != masked_crc32c(payload):
                raise ValueError("%s: corrupted record data" % tfrecord_path)
            yield payload


def create_tensorrec_dataset_from_tfrecord(tfrecord_path, verify=True):
    """(input_utils.py:109-142).  The reference's files hold one record; if a file holds several (one matrix each, as
    TFRecordDataset would yield them one by one) the first is returned -- the graph consumes one element per run."""
    for payload in _read_records(tfrecord_path, verify):
        feats = _parse_example(payload)
        missing = [k for k in ('row_index', 'col_index', 'values', 'd0', 'd1') if k not in feats]
        if missing:
            raise ValueError("%s: record lacks the features %s" % (tfrecord_path, missing))
        if len(feats['d0']) != 1 or len(feats['d1']) != 1:
            raise ValueError("%s: d0 / d1 must be scalars" % tfrecord_path)
        return TensorRecDataset(feats['row_index'], feats['col_index'], feats['values'], feats['d0'][0], feats['d1'][0])
    raise ValueError("%s holds no record" % tfrecord_path)
