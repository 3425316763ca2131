"""The on-disk input format of the path, byte for byte, without TensorFlow.

* TFRecord framing (what tf.python_io.TFRecordWriter / tf_record_iterator produce, dataset_.py:764,
  serialize.py:246-256):  u64le length | u32le masked_crc32c(length) | payload | u32le masked_crc32c(payload)
  with masked(c) = ((c >> 15 | c << 17) + 0xa282ead8) mod 2^32 over CRC-32C (Castagnoli).
* payload = serialized tf.train.Example { features { feature: map<string, Feature> } } where
  Feature = oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3 }.
  The reference writes 'height', 'width', 'depth' (int64), 'label' (int64 list) and 'image_raw'
  (H*W*C uint8, HWC, BGR) per frame (serialize.py:246-256) and reads them back at dataset_.py:100-133.
* '<paths file>.tfrecord.size' sidecar: tab separated items / type / cpi / fpc / labelcount
  (serialize.py:138-151, dataset_.py:701-756).
"""
import itertools
import os
import struct
from ast import literal_eval

import numpy as np

# ---- CRC-32C ----------------------------------------------------------------------------------------
_POLY = 0x82F63B78


def _make_tables():
    t0 = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (_POLY if c & 1 else 0)
        t0[i] = c
    tabs = [t0]
    for _ in range(7):
        prev = tabs[-1]
        tabs.append((prev >> 8) ^ t0[prev & 0xFF])
    return [[int(v) for v in t] for t in tabs]


_T = _make_tables()


def crc32c(data, crc=0):
    """Slicing-by-8 CRC-32C; crc32c(b'123456789') == 0xE3069283."""
    crc ^= 0xFFFFFFFF
    mv = memoryview(data)
    n = len(mv)
    i = 0
    t0, t1, t2, t3, t4, t5, t6, t7 = _T
    n8 = n - (n % 8)
    if n8:
        for lo, hi in struct.iter_unpack("<II", mv[:n8]):
            lo ^= crc
            crc = (t7[lo & 0xFF] ^ t6[(lo >> 8) & 0xFF] ^ t5[(lo >> 16) & 0xFF] ^ t4[lo >> 24] ^
                   t3[hi & 0xFF] ^ t2[(hi >> 8) & 0xFF] ^ t1[(hi >> 16) & 0xFF] ^ t0[hi >> 24])
        i = n8
    for b in mv[i:]:
        crc = t0[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- record framing ----------------------------------------------------------------------------------
class TFRecordWriter:
    def __init__(self, path):
        self.f = open(path, "wb")

    def write(self, payload):
        hdr = struct.pack("<Q", len(payload))
        self.f.write(hdr)
        self.f.write(struct.pack("<I", masked_crc32c(hdr)))
        self.f.write(payload)
        self.f.write(struct.pack("<I", masked_crc32c(payload)))

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class tf_record_iterator:
    """Iterates payloads like tf.python_io.tf_record_iterator (dataset_.py:764).  verify_crc checks both CRCs
    (TF checks them too); skip(n) fast-forwards by header only (dataset_.py:772-811 reads and discards)."""

    def __init__(self, path, verify_crc=True):
        self.f = open(path, "rb")
        self.verify = verify_crc
        self.size = os.fstat(self.f.fileno()).st_size          # record lengths come from the file: none may reach past its end

    def __iter__(self):
        return self

    def _header(self):
        hdr = self.f.read(12)
        if len(hdr) == 0:
            raise StopIteration
        if len(hdr) < 12:
            raise IOError("truncated TFRecord header")
        length, = struct.unpack("<Q", hdr[:8])
        if self.verify and struct.unpack("<I", hdr[8:])[0] != masked_crc32c(hdr[:8]):
            raise IOError("corrupted TFRecord length CRC")
        if length > self.size - self.f.tell() - 4:             # payload + its CRC must fit in what is left (as vlh's record_fits)
            raise IOError("truncated TFRecord: a record of %d bytes at offset %d reaches past the end of the file (%d bytes)" %
                          (length, self.f.tell() - 12, self.size))
        return length

    def __next__(self):
        length = self._header()
        payload = self.f.read(length)
        tail = self.f.read(4)
        if len(payload) < length or len(tail) < 4:
            raise IOError("truncated TFRecord payload")
        if self.verify and struct.unpack("<I", tail)[0] != masked_crc32c(payload):
            raise IOError("corrupted TFRecord payload CRC")
        return payload

    def skip(self, count):
        for i in range(count):
            try:
                length = self._header()
            except StopIteration:                              # a bare StopIteration would end somebody's for-loop silently
                raise IOError("truncated TFRecord: the file ends after %d of the %d records to skip" % (i, count))
            self.f.seek(length + 4, 1)

    def close(self):
        self.f.close()


# ---- minimal protobuf codec for tf.train.Example ---------------------------------------------------------
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _ld(field, payload):   # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_example(features):
    """features: {name: bytes | [bytes] | int | [int] | float | [float] | np.float32 array}.  Keys are emitted in
    sorted order (protobuf map order is unspecified; TF's C++ serializer also sorts deterministically)."""
    body = b""
    for key in sorted(features):
        v = features[key]
        if isinstance(v, (bytes, bytearray)):
            v = [bytes(v)]
        elif isinstance(v, (int, np.integer, float, np.floating)):
            v = [v]
        v = list(v)
        if v and isinstance(v[0], (bytes, bytearray)):
            feat = _ld(1, b"".join(_ld(1, bytes(x)) for x in v))                           # BytesList
        elif v and isinstance(v[0], (float, np.floating)):
            feat = _ld(2, _ld(1, struct.pack("<%df" % len(v), *v)))                        # FloatList (packed)
        else:
            feat = _ld(3, _ld(1, b"".join(_varint(int(x)) for x in v)))                    # Int64List (packed)
        entry = _ld(1, key.encode()) + _ld(2, feat)
        body += _ld(1, entry)                                                              # Features.feature map entry
    return _ld(1, body)                                                                    # Example.features


def _fields(buf):
    pos, end = 0, len(buf)
    while pos < end:
        tag, pos = _read_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 2:
            ln, pos = _read_varint(buf, pos)
            yield field, wt, buf[pos:pos + ln]
            pos += ln
        elif wt == 0:
            v, pos = _read_varint(buf, pos)
            yield field, wt, v
        elif wt == 5:
            yield field, wt, buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            yield field, wt, buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported wire type %d" % wt)


def decode_example(payload):
    """-> {name: list of bytes | ints | floats} (accepts packed and unpacked encodings)."""
    out = {}
    mv = memoryview(payload)
    for f1, _, features in _fields(mv):
        if f1 != 1:
            continue
        for f2, _, entry in _fields(features):
            if f2 != 1:
                continue
            key, feat = None, None
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    key = bytes(v).decode()
                elif f3 == 2:
                    feat = v
            vals = []
            for kind, _, lst in _fields(feat):
                for f5, wt, v in _fields(lst):
                    if kind == 1:
                        vals.append(bytes(v))
                    elif kind == 2:
                        vals.extend(struct.unpack("<%df" % (len(v) // 4), v) if wt == 2 else struct.unpack("<f", v))
                    elif kind == 3:
                        if wt == 2:
                            p = 0
                            while p < len(v):
                                x, p = _read_varint(v, p)
                                vals.append(x - (1 << 64) if x >> 63 else x)
                        else:
                            vals.append(v - (1 << 64) if v >> 63 else v)
            out[key] = vals
    return out


def frame_example(frame_u8, label):
    """serialize.py:246-256: one frame -> Example payload."""
    h, w, c = frame_u8.shape
    label = list(label) if isinstance(label, (list, tuple, np.ndarray)) else [label]
    return encode_example({"height": h, "width": w, "depth": c, "label": [int(l) for l in label],
                           "image_raw": np.ascontiguousarray(frame_u8, np.uint8).tobytes()})


def parse_frame_example(payload):
    """dataset_.py:100-133: -> (uint8 [H,W,C] view, label list)."""
    ex = decode_example(payload)
    h, w, c = ex["height"][0], ex["width"][0], ex["depth"][0]
    img = np.frombuffer(ex["image_raw"][0], np.uint8).reshape(h, w, c)
    return img, list(ex["label"])


def vector_example(vector, label):
    """serialize_vectors_to_tfrecord (serialize.py:258-266): one float32 vector -> Example payload
    {dimension, label, vector_raw}."""
    v = np.ascontiguousarray(vector, np.float32).ravel()
    label = list(label) if isinstance(label, (list, tuple, np.ndarray)) else [label]
    return encode_example({"dimension": int(v.size), "label": [int(l) for l in label], "vector_raw": v.tobytes()})


def parse_vector_example(payload):
    """deserialize_vector (dataset_.py:137-168): -> (float32 [dim] view, label list); the stored dimension must match."""
    ex = decode_example(payload)
    vec = np.frombuffer(ex["vector_raw"][0], np.float32)
    dim = int(ex["dimension"][0])
    if dim != len(vec):
        raise ValueError("Deserialized vector length %d but dimension stored is %d." % (len(vec), dim))
    return vec, list(ex["label"])


# ---- .size sidecar -----------------------------------------------------------------------------------------
def write_size_file(path, num_items, mode, clips_per_item, fpc, max_num_labels):
    """serialize.py:138-151; cpi is run-length encoded [(count, cpv), ...]."""
    with open(path, "w") as f:
        f.write("items\t%d\n" % num_items)
        f.write("type\t%s\n" % mode)
        cpi = [(len(list(g)), k) for k, g in itertools.groupby(clips_per_item)] if clips_per_item is not None else None
        f.write("cpi\t%s\n" % str(cpi))
        f.write("fpc\t%s\n" % str(fpc))
        f.write("labelcount\t%s\n" % str(max_num_labels))


def read_size_file(path):
    """dataset_.py:701-756 -> dict(items, type, cpv list | None, fpc | None, labelcount).  The reference eval()s the
    fields; literal_eval accepts the same literals without executing code."""
    d = {}
    with open(path, "r") as f:
        for line in f:
            if line.strip():
                k, v = line.strip().split("\t")
                d[k.strip()] = v.strip()
    items = int(literal_eval(d["items"]))
    cpv = literal_eval(d["cpi"])
    fpc = literal_eval(d["fpc"])
    if isinstance(cpv, list):
        if cpv and isinstance(cpv[0], tuple):
            cpv = [item for num, item in cpv for _ in range(num)]
        if len(cpv) != items:
            raise ValueError("Read %d items but got cpv list of size %d" % (items, len(cpv)))
    return {"items": items, "type": d["type"], "cpv": cpv, "fpc": fpc, "labelcount": int(d["labelcount"])}
