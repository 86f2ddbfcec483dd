"""detection JSON / feature-shard formats (CPU) and MultiScaleRoIAlign parity (gpu)."""
import numpy as np
import pytest
import torch

from skghoi_amd import cache


def test_detection_json_roundtrip(tmp_path):
    p = tmp_path / "HICO_test2015_00000001.json"
    boxes = np.array([[1.5, 2.0, 30.25, 40.0], [0, 0, 10, 10]]); scores = [0.9, 0.21]; labels = [49, 3]
    cache.write_detections_json(p, boxes, scores, labels)
    d = cache.read_detections_json(p)
    assert d["boxes"].dtype == torch.float32 and d["labels"].dtype == torch.int64
    assert torch.equal(d["boxes"], torch.tensor(boxes, dtype=torch.float32)) and d["labels"].tolist() == labels
    import json
    raw = json.load(open(p))
    assert set(raw) == {"boxes", "scores", "labels"}           # the reference's keys (adamixer_preprocessing.py:99-135)


@pytest.mark.parametrize("dtype,tol", [("fp32", 0.0), ("fp16", 2e-3), ("bf16", 1.6e-2)])
def test_feature_shard_roundtrip(tmp_path, dtype, tol):
    rs = np.random.RandomState(0)
    pooled = [rs.standard_normal((n, 8, 3, 3)).astype(np.float32) for n in (4, 0, 7)]
    glob = rs.standard_normal((3, 16)).astype(np.float32)
    hw = [(800, 1200), (600, 800), (640, 480)]
    path = tmp_path / "shard.skgfc"
    cache.write_feature_shard(path, pooled, glob, hw, dtype=dtype)
    sh = cache.FeatureShard(str(path))
    assert (sh.n_images, sh.n_boxes, sh.C, sh.pool, sh.gdim) == (3, 11, 8, 3, 16) and sh.payload_off % 4096 == 0
    x, g, shapes, counts = sh.batch(0, 3, "cpu")
    want = np.concatenate(pooled)
    assert x.shape == (11, 8, 3, 3) and counts == [4, 0, 7] and shapes == hw
    err = np.abs(x.numpy() - want).max() / np.abs(want).max()
    assert err <= tol if tol else np.array_equal(x.numpy(), want)
    assert np.array_equal(g.numpy().reshape(3, 16), glob)
    x2, _, _, c2 = sh.batch(2, 3, "cpu")
    assert c2 == [7] and x2.shape[0] == 7


@pytest.mark.gpu
def test_multiscale_roi_align_matches_oracle():
    from oracle import roi_align_oracle as RO
    from skghoi_amd.roi_pool import MultiScaleRoIAlign
    g = torch.Generator().manual_seed(0)
    shapes = [(200, 320), (192, 256)]
    feats = [torch.randn(2, 6, 200 // s, 320 // s, generator=g) for s in (4, 8, 16, 32)]
    boxes = [torch.tensor([[10.3, 20.1, 150.7, 180.2], [0., 0., 319., 199.], [100., 50., 104., 53.],
                           [250., 10., 318., 60.], [5., 5., 5.5, 5.2]]),
             torch.tensor([[30., 40., 90., 160.], [-5., -3., 40., 30.], [200., 150., 330., 210.]])]
    want = RO.multiscale_roi_align(feats, boxes, shapes, 7, 2)
    pool = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    got = pool({str(i): f.cuda() for i, f in enumerate(feats)}, [b.cuda() for b in boxes], shapes).cpu()
    assert got.shape == want.shape == (8, 6, 7, 7)
    assert (got - want).abs().max().item() <= 2e-5
    assert pool.scales == [0.25, 0.125, 0.0625, 0.03125] and (pool.k_min, pool.k_max) == (2, 5)
    # single level
    p1 = MultiScaleRoIAlign(["3"], 7, 2)
    got1 = p1({"3": feats[3].cuda()}, [b.cuda() for b in boxes], shapes).cpu()
    want1 = RO.multiscale_roi_align(feats[3:], boxes, shapes, 7, 2)
    assert (got1 - want1).abs().max().item() <= 2e-5


@pytest.mark.gpu
def test_multiscale_roi_align_backward_matches_oracle_autograd():
    """Gradient with respect to the four feature maps (the reference trains the detector through this pooling) against
    CPU autograd of the oracle's loop restatement, boxes on all four levels, partly outside the image, one degenerate."""
    from oracle import roi_align_oracle as RO
    from skghoi_amd.roi_pool import MultiScaleRoIAlign
    g = torch.Generator().manual_seed(1)
    shapes = [(200, 320), (192, 256)]
    feats = [torch.randn(2, 3, 200 // s, 320 // s, generator=g) for s in (4, 8, 16, 32)]
    boxes = [torch.tensor([[10.3, 20.1, 150.7, 180.2], [0., 0., 319., 199.], [100., 50., 104., 53.], [5., 5., 5.5, 5.2]]),
             torch.tensor([[30., 40., 90., 160.], [-5., -3., 40., 30.], [200., 150., 330., 210.]])]
    dout = torch.randn(7, 3, 7, 7, generator=g)
    fr = [f.clone().requires_grad_(True) for f in feats]
    RO.multiscale_roi_align(fr, boxes, shapes, 7, 2).backward(dout)
    fd = [f.cuda().requires_grad_(True) for f in feats]
    pool = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    out = pool({str(i): f for i, f in enumerate(fd)}, [b.cuda() for b in boxes], shapes)
    assert out.requires_grad
    out.backward(dout.cuda())
    for l, (a, b) in enumerate(zip(fd, fr)):
        want = b.grad if b.grad is not None else torch.zeros_like(b)
        assert a.grad is not None and (a.grad.cpu() - want).abs().max().item() <= 2e-5, l
    assert sum(float(b.grad.abs().sum()) > 0 for b in fr if b.grad is not None) >= 3      # several levels really used
    # without gradients requested nothing is recorded
    with torch.no_grad():
        assert not pool({str(i): f for i, f in enumerate(fd)}, [b.cuda() for b in boxes], shapes).requires_grad


@pytest.mark.gpu
def test_cached_pipeline_equals_direct(tmp_path):
    """feature maps -> MultiScaleRoIAlign -> head   ==   producer -> shard on disk -> reader -> head (fp32, bit-equal)."""
    from collections import OrderedDict
    import cases, gpu_run
    from skghoi_amd.roi_pool import MultiScaleRoIAlign
    case = cases.build_case("ragged3")
    case["C"], case["p"] = 256, 7
    head = gpu_run.build_head(case)
    head.box_roi_pool = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    g = torch.Generator().manual_seed(1)
    B = len(case["detections"])
    feats = OrderedDict((str(i), torch.randn(B, 256, 800 // s, 1200 // s, generator=g).cuda())
                        for i, s in enumerate((4, 8, 16, 32)))
    det = gpu_run.to_cuda(case["detections"])
    torch.manual_seed(3)
    with torch.no_grad():
        direct = head(feats, det, case["shapes"])
    path = str(tmp_path / "s.skgfc")
    kept = cache.produce_shard(head, feats, det, case["shapes"], path)
    sh = cache.FeatureShard(path)
    pooled, gl, hw, counts = sh.batch(0, B, "cuda")
    assert counts == [len(k["boxes"]) for k in kept] and hw == case["shapes"]
    cp = cache.CachedPool(); cp.pooled = pooled
    head.box_roi_pool = cp
    torch.manual_seed(3)
    with torch.no_grad():
        cached = head({"3": gl}, det, hw)
    assert len(direct) == len(cached)
    for a, b in zip(direct, cached):
        for k in a:
            assert torch.equal(a[k], b[k]), k
