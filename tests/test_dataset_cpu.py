"""The bundled dataset fixtures (tests/golden/dataset/) and the oracle, without a GPU: the committed pixels hash to the
manifest, the pyramid rule of src/main.cpp:95,103 gives the recorded level sizes, and the oracle cascade on this machine
reproduces the hashes recorded when the fixtures were made (both FP-contraction variants of the finest level for two pairs)."""
import numpy as np
import pytest

from cascade_ref import Cascade, pyramid_levels
from dataset_util import MANIFEST, PAIRS, load_pair
from golden_util import sha


def test_manifest_is_complete():
    assert len(PAIRS) == 12
    for name in PAIRS:
        e = MANIFEST[name]
        assert pyramid_levels(e["rows"], e["cols"]) == e["levels"] == len(e["sizes"]) == len(e["depth_sha_c1"]) == len(e["depth_sha_c0"])
        assert set(e["labels"]) <= {0, 64, 128, 192, 254}                      # src/main.cpp:41-42 (key 4 paints 254, not 255)
        assert 0.03 < e["coverage"] < 0.3


@pytest.mark.parametrize("name", PAIRS)
def test_oracle_cascade_reproduces_the_recorded_hashes(oracle, lut, name):
    bgr, ann, e = load_pair(name)
    c = Cascade(oracle, bgr, ann, lut, 1, threads=min(8, oracle.max_threads()))
    c.estimate(1000)
    assert [list(s) for s in c.size] == e["sizes"] and [list(g.shape) for g in c.gray] == e["gray_sizes"]
    assert [sha(g) for g in c.gray] == e["gray_sha"]
    assert [sha(c.depth[l]) for l in range(c.P)] == e["depth_sha_c1"]
    assert sha(c.depth_u8) == e["depth_u8_sha_c1"]
    assert sha(oracle.desaturate(bgr, c.gray[0], c.depth[0], 1)) == e["desaturate_sha"]
    assert sha(oracle.haze(bgr, c.depth[0], 1)) == e["haze_sha"]


@pytest.mark.parametrize("name", ["Flower", "WomanParasol"])
def test_uncontracted_variant_and_recorded_spread(oracle, lut, name):
    """What nvcc's -fmad would change: recorded per pair as the level-0 max-abs spread between the two variants."""
    bgr, ann, e = load_pair(name)
    c0 = Cascade(oracle, bgr, ann, lut, 0, threads=min(8, oracle.max_threads())); c0.estimate(1000)
    assert [sha(c0.depth[l]) for l in range(c0.P)] == e["depth_sha_c0"] and sha(c0.depth_u8) == e["depth_u8_sha_c0"]
    c1 = Cascade(oracle, bgr, ann, lut, 1, threads=min(8, oracle.max_threads())); c1.estimate(1000)
    assert float(np.abs(c1.depth[0] - c0.depth[0]).max()) == e["spread_c0_c1_level0"]
