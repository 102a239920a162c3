"""The twelve bundled image/annotation pairs at their own resolution (tests/golden/dataset/, written by
tests/golden/make_golden.py --dataset): decoded, lossless PNG files + a manifest of oracle-output hashes."""
import json
import os

import numpy as np

from golden_util import GOLDEN_DIR, sha

DATASET_DIR = os.path.join(GOLDEN_DIR, "dataset")
with open(os.path.join(DATASET_DIR, "manifest.json")) as _f:
    MANIFEST = json.load(_f)
PAIRS = sorted(MANIFEST)


def image_path(name):
    return os.path.join(DATASET_DIR, f"{name}.png")


def annotation_path(name):
    return os.path.join(DATASET_DIR, f"{name}_ann.png")


def load_pair(name):
    """(bgr u8 [rows, cols, 3] as cv::imread would hand it over, annotation u8 [rows, cols], manifest entry)."""
    from PIL import Image
    rgb = np.array(Image.open(image_path(name)).convert("RGB"))
    ann = np.array(Image.open(annotation_path(name)).convert("L"))
    e = MANIFEST[name]
    assert sha(rgb) == e["rgb_sha"] and sha(ann) == e["annotation_sha"], f"{name}: fixture pixels changed"
    return np.ascontiguousarray(rgb[..., ::-1]), ann, e
