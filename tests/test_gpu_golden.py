"""-m gpu: the HIP path against the committed golden vectors (no oracle code involved at run time)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_app

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_cornell_golden_images_and_first_hits(twk):
    gold = np.load(os.path.join(GOLDEN, "oracle_cornell.npz"))
    app = load_app(twk, "system_rtigo3_cornell_box_c1.txt", "scene_rtigo3_cornell_box_c1.txt", (64, 64))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.debugCapture(True)
    dev.render(0)
    assert np.array_equal(_bits(dev.getOutputBufferHost()), _bits(gold["c1_64_spp1"]))
    tbg, ids = dev.debugReadFirstHits()
    assert np.array_equal(ids, gold["c1_64_firsthit_ids"])
    hit = ids[:, 0] >= 0
    assert np.array_equal(_bits(tbg[hit]), _bits(gold["c1_64_firsthit_tbg"][hit]))
    dev.render(1)
    assert np.array_equal(_bits(dev.getOutputBufferHost()), _bits(gold["c1_64_spp2"]))
    dev.close()

    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (64, 36))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    for it in range(2):
        dev.render(it)
    assert np.array_equal(_bits(dev.getOutputBufferHost()), _bits(gold["c2_64x36_spp2"]))
    dev.close()
