"""The input format, byte for byte: CRC-32C known answers, TFRecord framing, the tf.train.Example codec
(cross-checked against google.protobuf with the same schema built at run time), the .size sidecar and the
native batch reader (libvltf_host.so) against the pure-Python restatement."""
import os
import struct

import numpy as np
import pytest

from vltf_amd import _hostio, tfrecord as T


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(_hostio.LIB_PATH):
        import __graft_entry__ as g
        g.build()


def test_crc32c_known_answers():
    # RFC 3720 B.4 vectors
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    for data in (b"", b"a", b"123456789", bytes(range(256)) * 37 + b"xyz"):
        assert _hostio.lib().vlh_crc32c(data, len(data)) == T.crc32c(data)
        assert _hostio.masked_crc32c(data) == T.masked_crc32c(data)
    c = T.crc32c(b"123456789")
    assert T.masked_crc32c(b"123456789") == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def example_classes():
    """tf.train.Example / Features / Feature / *List rebuilt with descriptor_pb2 (tensorflow/core/example/*.proto)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="vltf_test_example.proto", package="vltf_test", syntax="proto3")
    F = descriptor_pb2.FieldDescriptorProto

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = tname
        return m
    msg("BytesList", [("value", 1, F.TYPE_BYTES, F.LABEL_REPEATED, None)])
    msg("FloatList", [("value", 1, F.TYPE_FLOAT, F.LABEL_REPEATED, None)])
    msg("Int64List", [("value", 1, F.TYPE_INT64, F.LABEL_REPEATED, None)])
    feat = msg("Feature", [("bytes_list", 1, F.TYPE_MESSAGE, F.LABEL_OPTIONAL, ".vltf_test.BytesList"),
                           ("float_list", 2, F.TYPE_MESSAGE, F.LABEL_OPTIONAL, ".vltf_test.FloatList"),
                           ("int64_list", 3, F.TYPE_MESSAGE, F.LABEL_OPTIONAL, ".vltf_test.Int64List")])
    feats = msg("Features", [("feature", 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".vltf_test.Features.FeatureEntry")])
    entry = feats.nested_type.add(name="FeatureEntry")
    entry.options.map_entry = True
    entry.field.add(name="key", number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    entry.field.add(name="value", number=2, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name=".vltf_test.Feature")
    msg("Example", [("features", 1, F.TYPE_MESSAGE, F.LABEL_OPTIONAL, ".vltf_test.Features")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("vltf_test.Example"))


def test_example_codec_against_protobuf():
    Example = example_classes()
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 256, (6, 7, 3), dtype=np.uint8)
    ours = T.frame_example(frame, [5, 2])
    ex = Example()
    ex.ParseFromString(ours)                                    # protobuf accepts our bytes
    f = ex.features.feature
    assert f["height"].int64_list.value[0] == 6 and f["width"].int64_list.value[0] == 7 and f["depth"].int64_list.value[0] == 3
    assert list(f["label"].int64_list.value) == [5, 2] and f["image_raw"].bytes_list.value[0] == frame.tobytes()
    ref = Example()                                            # and we accept protobuf's bytes
    ref.features.feature["height"].int64_list.value.append(6)
    ref.features.feature["width"].int64_list.value.append(7)
    ref.features.feature["depth"].int64_list.value.append(3)
    ref.features.feature["label"].int64_list.value.extend([5, 2])
    ref.features.feature["image_raw"].bytes_list.value.append(frame.tobytes())
    ref.features.feature["score"].float_list.value.extend([0.5, -2.0])
    img, lab = T.parse_frame_example(ref.SerializeToString())
    assert np.array_equal(img, frame) and lab == [5, 2]
    assert T.decode_example(ref.SerializeToString())["score"] == [0.5, -2.0]
    assert T.decode_example(T.encode_example({"neg": [-3, 1 << 40]}))["neg"] == [-3, 1 << 40]


def write_frames(path, frames, labels):
    with T.TFRecordWriter(path) as w:
        for fr, lab in zip(frames, labels):
            w.write(T.frame_example(fr, lab))


def test_framing_roundtrip_and_native_reader(tmp_path):
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (7, 12, 10, 3), dtype=np.uint8)
    labels = [[i % 4] for i in range(7)]
    labels[3] = [3, 9, 1]
    path = str(tmp_path / "a.tfrecord")
    write_frames(path, frames, labels)
    raw = open(path, "rb").read()
    n, = struct.unpack("<Q", raw[:8])
    assert struct.unpack("<I", raw[8:12])[0] == T.masked_crc32c(raw[:8])
    assert struct.unpack("<I", raw[12 + n:16 + n])[0] == T.masked_crc32c(raw[12:12 + n])
    got = [T.parse_frame_example(p) for p in T.tf_record_iterator(path)]
    assert len(got) == 7 and all(np.array_equal(g[0], f) for g, f in zip(got, frames)) and [g[1] for g in got] == labels
    imgs, labs, off = _hostio.read_frames(path, 0, 4, (12, 10, 3))
    assert np.array_equal(imgs, frames[:4]) and labs == labels[:4]
    imgs2, labs2, off2 = _hostio.read_frames(path, off, 3, (12, 10, 3))
    assert np.array_equal(imgs2, frames[4:]) and labs2 == labels[4:] and off2 == len(raw)
    assert _hostio.skip_records(path, 0, 4) == off
    it = T.tf_record_iterator(path)
    it.skip(4)
    assert np.array_equal(T.parse_frame_example(next(it))[0], frames[4])
    with pytest.raises(EOFError) as e:                          # ragged: fewer records than asked
        _hostio.read_frames(path, off, 5, (12, 10, 3))
    assert e.value.records_read == 3
    with pytest.raises(_hostio.HostIOError):                     # wrong geometry
        _hostio.read_frames(path, 0, 1, (12, 11, 3))
    empty = str(tmp_path / "empty.tfrecord")
    open(empty, "wb").close()
    assert list(T.tf_record_iterator(empty)) == []
    with pytest.raises(EOFError):
        _hostio.read_frames(empty, 0, 1, (12, 10, 3))


def test_threaded_native_reader_matches_serial(tmp_path):
    """vlh_read_frames_mt: same images / labels / offsets / EOF accounting / first-error semantics as one thread."""
    rng = np.random.default_rng(4)
    n = 53
    frames = rng.integers(0, 256, (n, 9, 8, 3), dtype=np.uint8)
    labels = [[int(rng.integers(0, 300))] * (1 + i % 3) for i in range(n)]          # label >= 128: records differ in length
    path = str(tmp_path / "m.tfrecord")
    write_frames(path, frames, labels)
    size = os.path.getsize(path)
    for threads in (2, 3, 8, 64):
        a = _hostio.read_frames(path, 0, 40, (9, 8, 3), threads=1)
        b = _hostio.read_frames(path, 0, 40, (9, 8, 3), threads=threads)
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] == labels[:40] and a[2] == b[2]
        c = _hostio.read_frames(path, b[2], 13, (9, 8, 3), threads=threads)
        assert np.array_equal(c[0], frames[40:]) and c[2] == size
        with pytest.raises(EOFError) as e:                          # ragged tail: 13 records left, 30 asked
            _hostio.read_frames(path, b[2], 30, (9, 8, 3), threads=threads)
        assert e.value.records_read == 13
    raw = bytearray(open(path, "rb").read())
    raw[a[2] + 12 + 40] ^= 0x01                                     # corrupt the payload of record 40
    bad = str(tmp_path / "mbad.tfrecord")
    open(bad, "wb").write(raw)
    with pytest.raises(_hostio.HostIOError, match="record 40"):
        _hostio.read_frames(bad, 0, n, (9, 8, 3), threads=8)
    ok, _, _ = _hostio.read_frames(bad, 0, 40, (9, 8, 3), threads=8)   # the records before it are fine
    assert np.array_equal(ok, frames[:40])
    truncated = str(tmp_path / "mtrunc.tfrecord")
    open(truncated, "wb").write(bytes(raw[:a[2] - 7]))               # cut inside record 39
    with pytest.raises(EOFError) as e:
        _hostio.read_frames(truncated, 0, 40, (9, 8, 3), threads=4)
    assert e.value.records_read == 39


def test_corruption_is_detected(tmp_path):
    frames = np.zeros((2, 4, 4, 3), np.uint8)
    path = str(tmp_path / "c.tfrecord")
    write_frames(path, frames, [[0], [1]])
    raw = bytearray(open(path, "rb").read())
    n, = struct.unpack("<Q", raw[:8])
    raw[12 + n // 2] ^= 0x10                                    # flip a bit inside record 0's image bytes
    bad = str(tmp_path / "bad.tfrecord")
    open(bad, "wb").write(raw)
    with pytest.raises(IOError):
        list(T.tf_record_iterator(bad))
    with pytest.raises(_hostio.HostIOError):
        _hostio.read_frames(bad, 0, 2, (4, 4, 3))
    imgs, _, _ = _hostio.read_frames(bad, 0, 2, (4, 4, 3), verify_crc=False)      # like TF with CRC checks off
    assert imgs.shape == (2, 4, 4, 3)


def test_untrusted_record_lengths(tmp_path):
    """Lengths come from the file: a length near 2^64 (wraps `len + 4`), one that merely runs past the end of the file, and a
    malformed varint length inside the Example must all fail cleanly in the single-threaded, multi-threaded and skip paths."""
    frames = np.arange(3 * 4 * 4 * 3, dtype=np.uint8).reshape(3, 4, 4, 3)
    path = str(tmp_path / "l.tfrecord")
    write_frames(path, frames, [[0], [1], [2]])
    good = bytearray(open(path, "rb").read())
    n, = struct.unpack("<Q", good[:8])
    rec = 12 + n + 4
    for bogus in (2 ** 64 - 3, 2 ** 63 + 5, len(good) * 2, rec * 3 - 16 - rec + 1):     # the last: one byte too long for record 1
        raw = bytearray(good)
        raw[rec:rec + 8] = struct.pack("<Q", bogus)                 # record 1's length (its CRC now mismatches too)
        bad = str(tmp_path / "lbad.tfrecord")
        open(bad, "wb").write(raw)
        for threads in (1, 2):
            with pytest.raises((EOFError, _hostio.HostIOError)):
                _hostio.read_frames(bad, 0, 3, (4, 4, 3), verify_crc=False, threads=threads)
        with pytest.raises(_hostio.HostIOError):
            _hostio.skip_records(bad, 0, 3)
        assert _hostio.skip_records(bad, 0, 1) == rec              # the record before it is still reachable
    # a huge varint length inside the payload (field length > remaining bytes)
    raw = bytearray(good)
    raw[12 + 1:12 + 3] = b"\xff\xff"                                # the Features length varint of record 0 -> far past the record
    bad = str(tmp_path / "lbad2.tfrecord")
    open(bad, "wb").write(raw)
    with pytest.raises(_hostio.HostIOError):
        _hostio.read_frames(bad, 0, 1, (4, 4, 3), verify_crc=False, threads=1)


def test_python_reader_rejects_untrusted_lengths(tmp_path):
    """The pure-Python tf_record_iterator (the reader of vectors datasets: dataset_._read_vectors, skip, vector_dim) trusts no
    length either: an oversized length fails as a truncated file instead of a MemoryError / OverflowError from f.read, skip()
    notices the end of the file, and running out of records inside skip() is an IOError, not a bare StopIteration."""
    from vltf_amd import tfrecord as T
    path = str(tmp_path / "v.tfrecord")
    with T.TFRecordWriter(path) as w:
        for i in range(3):
            w.write(T.encode_example({"dimension": 4, "label": [i], "vector_raw": np.arange(4, dtype=np.float32).tobytes()}))
    good = bytearray(open(path, "rb").read())
    n, = struct.unpack("<Q", good[:8])
    rec = 12 + n + 4
    for bogus in (2 ** 64 - 3, 2 ** 63 + 5, len(good) * 2, len(good) - rec - 16 + 1):
        raw = bytearray(good)
        raw[rec:rec + 8] = struct.pack("<Q", bogus)
        bad = str(tmp_path / "vbad.tfrecord")
        open(bad, "wb").write(raw)
        it = T.tf_record_iterator(bad, verify_crc=False)
        next(it)                                                     # record 0 is intact
        with pytest.raises(IOError, match="truncated"):
            next(it)
        it.close()
        it = T.tf_record_iterator(bad, verify_crc=False)
        with pytest.raises(IOError, match="truncated"):
            it.skip(2)
        it.close()
    it = T.tf_record_iterator(path)
    with pytest.raises(IOError, match="ends after 3 of the 5"):
        it.skip(5)
    it.close()
    open(str(tmp_path / "cut.tfrecord"), "wb").write(good[:rec + 20])    # file cut inside record 1's payload
    it = T.tf_record_iterator(str(tmp_path / "cut.tfrecord"))
    next(it)
    with pytest.raises(IOError, match="truncated"):
        next(it)
    it.close()


def test_size_file(tmp_path):
    p = str(tmp_path / "x.size")
    T.write_size_file(p, 5, "video", [2, 2, 3, 3, 3], 16, 1)
    assert open(p).read() == "items\t5\ntype\tvideo\ncpi\t[(2, 2), (3, 3)]\nfpc\t16\nlabelcount\t1\n"     # serialize.py:138-151
    d = T.read_size_file(p)
    assert d == {"items": 5, "type": "video", "cpv": [2, 2, 3, 3, 3], "fpc": 16, "labelcount": 1}
    T.write_size_file(p, 3, "image", None, None, 2)
    d = T.read_size_file(p)
    assert d["cpv"] is None and d["fpc"] is None
    open(p, "w").write("items\t2\ntype\tvideo\ncpi\t[(3, 1)]\nfpc\t4\nlabelcount\t1\n")
    with pytest.raises(ValueError):
        T.read_size_file(p)
