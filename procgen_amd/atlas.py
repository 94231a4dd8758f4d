"""
Reader for the ".atlas" pack written by procgen_amd/csrc (image_io.h): a zlib-compressed cache of decoded
images keyed by their path relative to resource_root.  Data-format helper only (used by tooling/tests).
"""
import struct
import zlib

import numpy as np


def read_atlas(path):
    """Returns {relpath: (format, ndarray[h, w] uint32 0xAARRGGBB)}."""
    out = {}
    with open(path, "rb") as f:
        d = f.read()
    assert d[:8] == b"PGATLAS1", "bad atlas magic"
    (cnt,) = struct.unpack_from("<I", d, 8)
    off = 12
    for _ in range(cnt):
        (nl,) = struct.unpack_from("<I", d, off)
        off += 4
        name = d[off:off + nl].decode()
        off += nl
        w, h, fmt, zl = struct.unpack_from("<IIII", d, off)
        off += 16
        px = np.frombuffer(zlib.decompress(d[off:off + zl]), dtype=np.uint32).reshape(h, w)
        off += zl
        out[name] = (fmt, px)
    return out
