"""ctypes binding of libvltf_host.so (include/vltf_host.h): native TFRecord batch reading."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvltf_host.so")
p, i32, i64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t

SIGNATURES = {
    "vlh_last_error": (C.c_char_p, []),
    "vlh_crc32c": (C.c_uint32, [p, sz]),
    "vlh_masked_crc32c": (C.c_uint32, [p, sz]),
    "vlh_read_frames": (i64, [C.c_char_p, i64, i32, i32, p, i64, p, p, i32, p, p]),
    "vlh_read_frames_mt": (i64, [C.c_char_p, i64, i32, i32, p, i64, p, p, i32, p, p, i32]),
    "vlh_skip_records": (i64, [C.c_char_p, i64, i64, i32]),
}
_lib = None


class HostIOError(IOError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HostIOError("libvltf_host.so not found at %s -- run __graft_entry__.build()" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def masked_crc32c(data):
    b = bytes(data)
    return int(lib().vlh_masked_crc32c(b, len(b)))


def default_threads():
    """Reader threads: VLTF_READ_THREADS, else min(8, cores the cgroup grants this process)."""
    env = os.environ.get("VLTF_READ_THREADS")
    if env:
        return max(1, int(env))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(8, n))


def read_frames(path, offset, count, shape, max_labels=8, verify_crc=True, out=None, threads=None):
    """Reads `count` frame records -> (images uint8 [count, H, W, C], labels list[list[int]], new_offset).
    Raises EOFError with .records_read when the file ends early (dataset_.py:179-181 rewinds on that).
    threads: payload work is split over that many native threads (None = default_threads())."""
    h, w, c = shape
    nbytes = h * w * c
    images = out if out is not None else np.empty((count, h, w, c), np.uint8)
    dims = np.zeros((count, 3), np.int32)
    labels = np.zeros((count, max_labels), np.int64)
    lcount = np.zeros(count, np.int32)
    nread = C.c_int32(0)
    rc = lib().vlh_read_frames_mt(path.encode(), offset, count, int(verify_crc), images.ctypes.data, nbytes, dims.ctypes.data,
                                  labels.ctypes.data, max_labels, lcount.ctypes.data, C.byref(nread),
                                  default_threads() if threads is None else int(threads))
    if rc == -1:
        e = EOFError("end of %s after %d of %d records" % (path, nread.value, count))
        e.records_read = nread.value
        raise e
    if rc < 0:
        raise HostIOError("vlh_read_frames(%s): %s" % (path, lib().vlh_last_error().decode()))
    if not (dims == np.array([h, w, c])).all():
        raise HostIOError("record shape %s differs from the expected %s" % (dims[0].tolist(), [h, w, c]))
    return images, [labels[i, :min(lcount[i], max_labels)].tolist() for i in range(count)], int(rc)


def skip_records(path, offset, count, verify_crc=False):
    rc = lib().vlh_skip_records(path.encode(), offset, count, int(verify_crc))
    if rc < 0:
        raise HostIOError("vlh_skip_records(%s): %s" % (path, lib().vlh_last_error().decode()))
    return int(rc)
