"""The whole neighbourhood of the hot path in one run (examples/cached_inference_and_eval.py): producer -> cache on disk
-> batched inference -> HOI mAP -> exporters.  The synthetic ground truth makes the top cell of every image a true
positive that outranks everything else of its class in that image, so every class with ground truth has AP > 0."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cached_inference_and_eval(tmp_path):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import cached_inference_and_eval as ex
    outputs, summ, cells = ex.main(n_images=6, batch=3, out_dir=str(tmp_path))
    assert len(outputs) == 6 and all(o["boxes_h"].shape == (4 * 9, 4) for o in outputs)
    ap = summ["ap"]
    assert float(ap.max()) <= 1.0 + 1e-9 and float(ap.sum()) > 0
    assert cells.shape == (600, 6)
    files = sorted(os.listdir(tmp_path))
    assert "vcoco_results.pkl" in files and sum(f.endswith(".skgfc") for f in files) == 2 \
        and sum(f.endswith(".json") for f in files) == 6
