"""CPU: the product's PNG decoder / atlas pack against Qt 5.9.7's QImage(path).convertToFormat(...) CRCs
(tests/golden/asset_crc32_qt597.txt, produced by tests/tools/qt_image_hashes.py under PyQt5 5.9.7)."""
import os
import zlib

import numpy as np
import pytest

from helpers import REPO
from procgen_amd.atlas import read_atlas


def _golden(golden_dir):
    out = {}
    for line in open(os.path.join(golden_dir, "asset_crc32_qt597.txt")):
        parts = line.rstrip("\n").rsplit(" ", 4)
        out[(parts[0], int(parts[1]))] = (int(parts[2]), int(parts[3]), int(parts[4], 16))
    return out


def test_atlas_pack_matches_qt(golden_dir):
    path = os.path.join(REPO, "procgen_amd", "data", "coinrun.atlas")
    if not os.path.exists(path):
        pytest.skip("atlas not baked (run __graft_entry__.build())")
    gold = _golden(golden_dir)
    pack = read_atlas(path)
    assert len(pack) >= 120
    for key, (fmt, px) in pack.items():
        name = key[:-3] if key.endswith("|bg") else key
        w, h, crc = gold[(name, fmt)]
        assert px.shape == (h, w), key
        assert zlib.crc32(np.ascontiguousarray(px).tobytes()) & 0xFFFFFFFF == crc, f"pixels of {key} differ from Qt"


def test_oracle_side_decoder_matches_qt(golden_dir):
    """The test-side Pillow + qPremultiply decoder used to feed the oracle, on a sample of files."""
    root = "/root/reference/procgen/data/assets"
    if not os.path.isdir(root):
        pytest.skip("PNG tree not present")
    import oracle_env

    gold = _golden(golden_dir)
    keys = sorted(gold)[::23]
    for name, fmt in keys:
        px = oracle_env.decode_png_qt(os.path.join(root, name), fmt == 1)
        assert zlib.crc32(np.ascontiguousarray(px, dtype=np.uint32).tobytes()) & 0xFFFFFFFF == gold[(name, fmt)][2], name
