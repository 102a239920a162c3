import glob
import hashlib
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(f)[:-8] for f in glob.glob(os.path.join(GOLDEN_DIR, "*_256.npz")))
LEVELS = 3


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, f"{name}_256.npz"), allow_pickle=False)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def edited_from_ch0(ch0):
    e = np.zeros(ch0.shape + (3,), np.uint8)
    e[..., 0] = ch0
    return e
