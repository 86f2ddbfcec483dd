"""The captured-graph small-batch path (skghoi_amd/small.py) against the eager path of the same head: the graph holds the
same kernel sequence, so every result tensor must be BIT-identical -- on the capturing call, on replays, and on replays
with other contents of the same shape (other boxes / scores / classes / image sizes / TransH tables).  The eager path
itself is pinned to the reference's goldens in test_parity_gpu.py."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import cases
import gpu_run
import helpers
from skghoi_amd import synth

pytestmark = pytest.mark.gpu

SMALL_CASES = [c for c in cases.EVAL_CASES if c not in ("eval_targets", "many12")]


def _run(head, case, seed, small, det=None, shapes=None, feat3=None, buckets=False, capture_after=1):
    eng = head.engine()
    eng.debug = False
    eng.small_batch_max = 8 if small else 0
    eng.small_batch_buckets = buckets            # exact-shape plans are bit-identical to eager; bucket plans: see below
    eng.small_capture_after = capture_after      # 1: capture at the first sighting (these tests count captures and replays)
    det = gpu_run.to_cuda(case["detections"]) if det is None else det
    feat3 = case["feat3"].cuda() if feat3 is None else feat3
    feats = OrderedDict((k, feat3) for k in "0123")
    with torch.no_grad():
        torch.manual_seed(seed)
        res = head(feats, det, case["shapes"] if shapes is None else shapes)
        after = torch.empty(3).uniform_()
    K = case["cfg"]["K"]
    last = eng.last
    logits = last["logits"][:, :K + 1].clone() if "logits" in last else None
    return res, logits, after


def _same(a, b):
    assert len(a) == len(b)
    for ra, rb in zip(a, b):
        assert set(ra) == set(rb)
        for k in ra:
            assert ra[k].shape == rb[k].shape and ra[k].dtype == rb[k].dtype, k
            assert torch.equal(ra[k], rb[k]), k


@pytest.mark.parametrize("name", SMALL_CASES)
def test_graph_path_is_bit_identical_to_eager(name, precision):
    case = cases.build_case(name)
    head = gpu_run.build_head(case).eval()
    want, wl, wa = _run(head, case, 11, small=False)
    got, gl, ga = _run(head, case, 11, small=True)          # capturing call
    _same(got, want)
    assert torch.equal(ga, wa)                               # the host RNG was consumed identically
    if wl is not None:
        assert torch.equal(gl, wl)
    runner = head.engine()._small
    for seed in (12, 13):                                    # replays, other TransH tables
        want, wl, _ = _run(head, case, seed, small=False)
        got, gl, _ = _run(head, case, seed, small=True)
        _same(got, want)
        if wl is not None:
            assert torch.equal(gl, wl)
    lay_active = head.engine().last["layout"].n_active
    if lay_active:
        assert runner.misses == 1 and runner.hits == 2, (runner.misses, runner.hits)
    # and the result still matches the reference's golden (same comparison as the eager parity test, result level)
    g = helpers.load_golden(name)
    res, _, _ = _run(head, case, case["rng_seed"], small=True)
    assert len(res) == int(g["n_results"])
    for b, r in enumerate(res):
        for k in ("index", "prediction", "object"):
            if "res%d.%s" % (b, k) in g:
                assert np.array_equal(r[k].cpu().numpy(), g["res%d.%s" % (b, k)]), (b, k)
        if name != "nanbox" and "res%d.scores" % b in g and g["res%d.scores" % b].size:
            assert np.abs(r["scores"].cpu().numpy() - g["res%d.scores" % b]).max() <= 1e-5


def test_replay_with_other_contents_of_the_same_shape(precision):
    """One plan, many images: detections with other boxes, scores and object classes (another number of scored cells),
    another image size and other feature maps replay the plan captured for the first image."""
    case = cases.build_case("full20")
    head = gpu_run.build_head(case).eval()

    class Pool(torch.nn.Module):
        pooled = None

        def forward(self, features, boxes, image_shapes):
            return self.pooled

    head.box_roi_pool = Pool()
    runner = None
    cells = set()
    for i, hw in enumerate([(800, 1200), (640, 960), (800, 1200), (480, 640)]):
        im = synth.make_image(3000 + i, n_h=20, n_o=20)
        scale = torch.tensor([hw[1] / 1200.0, hw[0] / 800.0] * 2)
        det = [dict(boxes=(im["boxes"] * scale).cuda(), labels=im["labels"].cuda(), scores=im["scores"].cuda())]
        Pool.pooled = im["pooled"].cuda()
        feat3 = im["feat3"].cuda()
        want, wl, _ = _run(head, case, 50 + i, small=False, det=det, shapes=[hw], feat3=feat3)
        got, gl, _ = _run(head, case, 50 + i, small=True, det=det, shapes=[hw], feat3=feat3)
        _same(got, want)
        assert torch.equal(gl, wl)
        assert got[0]["prior"].is_contiguous() and got[0]["prior"].shape[0] == 2
        cells.add(int(got[0]["index"].numel()))
        runner = head.engine()._small
    assert runner.misses == 1 and runner.hits == 3
    assert len(cells) > 1                                   # the number of scored cells really changed between replays


def test_small_path_reference_quirks_and_errors():
    case = cases.build_case("skips_raise")                  # eval, batch 2, skipped image after one with pairs: HEAD:327
    head = gpu_run.build_head(case).eval()
    with pytest.raises(IndexError):
        _run(head, case, 3, small=True)
    head2 = gpu_run.build_head(case, reference_quirks=False).eval()
    want, _, _ = _run(head2, case, 3, small=False)
    got, _, _ = _run(head2, case, 3, small=True)
    _same(got, want)
    assert len(got) == 2 and got[1]["index"].numel() == 0
    # a batch without any pair takes the eager route from inside the small path
    only = dict(cases.build_case("skips_eval"))
    only["detections"] = only["detections"][:1]; only["feat3"] = only["feat3"][:1]; only["shapes"] = only["shapes"][:1]
    head3 = gpu_run.build_head(only, reference_quirks=False).eval()
    got, _, _ = _run(head3, only, 3, small=True)
    assert len(got) == 1 and got[0]["boxes_h"].shape == (0, 4)


def test_an_exact_shape_is_captured_at_its_second_sighting():
    """The default for exact-shape plans (batches of 2..8 images): the first call with a shape tuple takes the eager path and
    only notes the shape, the second captures, the third replays -- a stream whose shapes never repeat never pays a capture.
    All three results are bit-identical to eager."""
    case = cases.build_case("ragged3")
    head = gpu_run.build_head(case).eval()
    runner = None
    for i, want_stats in enumerate([(1, 0, 0), (1, 1, 0), (1, 1, 1)]):
        want, _, _ = _run(head, case, 20 + i, small=False)
        got, _, _ = _run(head, case, 20 + i, small=True, capture_after=2)
        _same(got, want)
        runner = head.engine()._small
        st = runner.stats()
        assert (st["deferred"], st["misses"], st["hits"]) == want_stats, (i, st)
    assert len(case["detections"]) >= 2 and not runner.sightings          # (a multi-image batch; the noted shape was consumed)


def test_plans_are_dropped_when_weights_change():
    case = cases.build_case("tiny")
    head = gpu_run.build_head(case).eval()
    a, la, _ = _run(head, case, 5, small=True)
    runner = head.engine()._small
    assert len(runner.plans) == 1
    head.box_pair_predictor.weight.data.mul_(2.0)
    b, lb, _ = _run(head, case, 5, small=True)
    assert runner.misses == 2 and len(runner.plans) == 1     # re-captured against the re-packed weights
    K = case["cfg"]["K"]
    bias = head.box_pair_predictor.bias.detach()
    assert torch.allclose(lb[:, :K] - bias, 2.0 * (la[:, :K] - bias), rtol=1e-4, atol=1e-5)
    want, lw, _ = _run(head, case, 5, small=False)
    _same(b, want)


def test_graph_path_with_exact_row_exponent_passes(monkeypatch):
    """fp16x2 with engine.EXACT_ROW_SCALE: every split-operand GEMM of the captured plan has a row-exponent pass in front
    of it, writing into a buffer the plan keeps (the capture's two branches share one pool).  Capturing call and replays
    must equal the eager path bit for bit."""
    from skghoi_amd import engine as _engine
    monkeypatch.setattr(_engine, "EXACT_ROW_SCALE", True)
    monkeypatch.setattr(gpu_run, "PRECISION", "fp16x2")
    case = cases.build_case("tiny")
    head = gpu_run.build_head(case).eval()
    assert head.precision == "fp16x2"
    for seed in (21, 22, 23):
        want, wl, _ = _run(head, case, seed, small=False)
        got, gl, _ = _run(head, case, seed, small=True)
        _same(got, want)
        if wl is not None:
            assert torch.equal(gl, wl)
    plans = list(head.engine()._small.plans.values())
    assert plans and all(len(p.amax_keep) > 0 for p in plans)


def _close(a, b, atol=2e-6, rtol=2e-5):
    """Bucket plans: integer outputs identical, floats up to the summation order of split-K reductions."""
    assert len(a) == len(b)
    for ra, rb in zip(a, b):
        assert set(ra) == set(rb)
        for k in ra:
            assert ra[k].shape == rb[k].shape and ra[k].dtype == rb[k].dtype, k
            if ra[k].dtype.is_floating_point:
                assert torch.allclose(ra[k], rb[k], atol=atol, rtol=rtol), (k, (ra[k] - rb[k]).abs().max().item())
            else:
                assert torch.equal(ra[k], rb[k]), k


def _single_image(i, n_h, n_o, C, p, hw=(800, 1200)):
    im = synth.make_image(7000 + i, n_h=n_h, n_o=n_o, out_channels=C, pool=p)
    det = [dict(boxes=im["boxes"].cuda(), labels=im["labels"].cuda(), scores=im["scores"].cuda())]
    return det, im["pooled"].cuda(), im["feat3"].cuda()


def test_bucket_plans_serve_every_shape_of_the_bucket(precision):
    """One captured plan per BUCKET of (humans, nodes): images of different shapes inside a bucket replay the same plan --
    larger after smaller, smaller after larger (stale rows of the earlier image stay in the unused tail) -- and every
    result equals the eager path's: indices / predictions / objects identical, scores and boxes to rounding."""
    case = cases.build_case("tiny")
    head = gpu_run.build_head(case).eval()

    class Pool(torch.nn.Module):
        pooled = None

        def forward(self, features, boxes, image_shapes):
            assert sum(len(b) for b in boxes) == self.pooled.shape[0]
            return self.pooled

    head.box_roi_pool = Pool()
    shapes = [(5, 8), (6, 9), (5, 7), (6, 6), (5, 9), (6, 9)]            # (humans, objects): all in the bucket (6, 15)
    from skghoi_amd.small import capacity
    assert len({(capacity(h, 15), capacity(h + o, 30)) for h, o in shapes}) == 1
    for i, (nh, no) in enumerate(shapes):
        det, pooled, feat3 = _single_image(i, nh, no, case["C"], case["p"])
        Pool.pooled = pooled
        want, wl, wa = _run(head, case, 90 + i, small=False, det=det, shapes=[(800, 1200)], feat3=feat3)
        got, gl, ga = _run(head, case, 90 + i, small=True, det=det, shapes=[(800, 1200)], feat3=feat3, buckets=True)
        _close(got, want)
        assert torch.equal(ga, wa)
        assert gl.shape == wl.shape and torch.allclose(gl, wl, atol=2e-6, rtol=2e-5)
        assert got[0]["boxes_h"].shape == (nh * (nh + no - 1), 4)
    st = head.engine()._small.stats()
    assert st["captures"] == 1 and st["misses"] == 1 and st["hits"] == len(shapes) - 1, st
    # and the golden of the reference still holds through a bucket plan
    g = helpers.load_golden("tiny")
    head2 = gpu_run.build_head(case).eval()
    res, _, _ = _run(head2, case, case["rng_seed"], small=True, buckets=True)
    for k in ("index", "prediction", "object"):
        assert np.array_equal(res[0][k].cpu().numpy(), g["res0.%s" % k]), k
    assert np.abs(res[0]["scores"].cpu().numpy() - g["res0.scores"]).max() <= 1e-5


def test_plan_eviction_retire_and_recapture_give_eager_results():
    """max_plans = 2 with three shapes in turn: every call beyond the second evicts the least recently used plan, the
    evicted plan is destroyed on the next capture's idle-device teardown, and a shape that comes back is captured again.
    Every result -- first capture, replay of a survivor, re-capture after eviction -- is bit-identical to the eager path
    (exact-shape plans)."""
    case = cases.build_case("tiny")
    head = gpu_run.build_head(case).eval()

    class Pool(torch.nn.Module):
        pooled = None

        def forward(self, features, boxes, image_shapes):
            return self.pooled

    head.box_roi_pool = Pool()
    eng = head.engine()
    from skghoi_amd.small import SmallBatchRunner
    eng._small = SmallBatchRunner(eng, max_plans=2)
    order = [(2, 3), (3, 2), (2, 3), (4, 4), (3, 2), (2, 3), (4, 4)]
    imgs = {}
    for i, shp in enumerate(sorted(set(order))):
        imgs[shp] = _single_image(40 + i, shp[0], shp[1], case["C"], case["p"])
    for i, shp in enumerate(order):
        det, pooled, feat3 = imgs[shp]
        Pool.pooled = pooled
        want, wl, _ = _run(head, case, 300 + i, small=False, det=det, shapes=[(800, 1200)], feat3=feat3)
        got, gl, _ = _run(head, case, 300 + i, small=True, det=det, shapes=[(800, 1200)], feat3=feat3)
        _same(got, want)
        assert torch.equal(gl, wl)
        assert head.engine()._small is eng._small and len(eng._small.plans) <= 2
    st = eng._small.stats()
    # (2,3) (3,2) miss; (2,3) hit; (4,4) miss, evicts (3,2); (3,2) miss, evicts (2,3); (2,3) miss, evicts (4,4); (4,4) miss
    assert st["misses"] == 6 and st["hits"] == 1 and st["evictions"] == 4 and st["captures"] == 6, st
    assert len(eng._small.retired) <= 1                     # everything but the last eviction has been buried


def test_engine_rebuild_tears_plans_down_through_the_idle_path():
    """capture -> a threshold change rebuilds the HeadEngine -> capture again: the old runner's plans are closed by the
    head (idle-device teardown), not dropped by refcount next to the new capture; results stay equal to eager."""
    case = cases.build_case("tiny")
    head = gpu_run.build_head(case).eval()
    a, _, _ = _run(head, case, 5, small=True)
    old = head.engine()._small
    assert old is not None and len(old.plans) == 1
    head.box_score_thresh = 0.21                            # engine() builds a new HeadEngine on the next call
    b, lb, _ = _run(head, case, 5, small=True)
    new = head.engine()._small
    assert new is not old and len(old.plans) == 0 and not old.retired and len(new.plans) == 1
    want, lw, _ = _run(head, case, 5, small=False)
    _same(b, want)


@pytest.mark.parametrize("name", ["tiny", "nanbox", "nms", "iter1", "iter0", "full20"])
def test_bucket_plans_reproduce_the_reference_goldens(name, precision):
    """Every single-image eval case through a BUCKET plan (capacity >= the image's graph; padded tails in every row space):
    integer outputs equal the eager path's and the reference's golden, floats to rounding -- including the zero-area boxes of
    `nanbox` (NaN scrub next to the zero-filled padding), NMS + top-k truncation, num_iter 0 / 1 and the full 20 x 20 graph."""
    case = cases.build_case(name)
    assert len(case["detections"]) == 1
    head = gpu_run.build_head(case).eval()
    want, wl, wa = _run(head, case, case["rng_seed"], small=False)
    got, gl, ga = _run(head, case, case["rng_seed"], small=True, buckets=True)
    got2, gl2, _ = _run(head, case, case["rng_seed"], small=True, buckets=True)              # replay
    plan = next(iter(head.engine()._small.plans.values()))
    assert plan.caps is not None and head.engine()._small.stats()["captures"] == 1
    _close(got, want); _close(got2, want)
    assert torch.equal(ga, wa)
    for a, b in zip(got, got2):
        for k in a:
            assert torch.equal(a[k], b[k]), k                                              # capture call == replay, bit for bit
    g = helpers.load_golden(name)
    for k in ("index", "prediction", "object"):
        if "res0.%s" % k in g:                                                             # (output-only fixtures hold fewer keys)
            assert np.array_equal(got[0][k].cpu().numpy(), g["res0.%s" % k]), k
    if name != "nanbox":
        assert np.abs(got[0]["scores"].cpu().numpy() - g["res0.scores"]).max() <= 1e-5
    K = case["cfg"]["K"]
    # (nanbox: the scrubbed +-FLT_MAX features drive the logits to ~1e9; the bar is relative there)
    assert np.abs(gl[:, :K].cpu().numpy() - g["logits_p"]).max() <= 1e-4 * max(1.0, np.abs(g["logits_p"]).max())


class _StreamPool(torch.nn.Module):
    pooled = None

    def forward(self, features, boxes, image_shapes):
        assert sum(len(b) for b in boxes) == self.pooled.shape[0]
        return self.pooled


def _image_stream(case, shapes, base):
    out = []
    for i, (nh, no) in enumerate(shapes):
        det, pooled, feat3 = _single_image(base + i, nh, no, case["C"], case["p"])
        out.append((det, pooled, OrderedDict((k, feat3) for k in "0123")))
    return out


def _loop(head, imgs, seed, ahead, order=None, decoys=None):
    """The test loop of utils.py:157-167 over `imgs`; ahead: hand image i + 1 to the head once forward i is enqueued."""
    eng = head.engine()
    eng.debug = False
    eng.small_batch_max, eng.small_batch_buckets, eng.small_capture_after = 8, True, 1
    uploaded = torch.cuda.Event(); uploaded.record()
    res = []
    with torch.no_grad():
        torch.manual_seed(seed)
        for k, (det, pooled, feats) in enumerate(imgs):
            head.box_roi_pool.pooled = pooled
            res.append(head(feats, det, [(800, 1200)]))
            if ahead and k + 1 < len(imgs):
                nxt = imgs[k + 1][0] if decoys is None else decoys[k]
                assert head.prefetch_eval(nxt, after=uploaded) is True
        tail = torch.empty(3).uniform_()
    return res, tail


def test_look_ahead_selection_gives_the_plain_loops_results():
    """InteractionHead.prefetch_eval: the next image's selection / count read-back / TransH table draw beside the forward in
    flight.  Results are bit-identical to the loop without look-ahead (same plans, same kernels, same tables), the global
    CPU generator ends at the same position, and every forward but the first claimed its look-ahead."""
    case = cases.build_case("tiny")
    head = gpu_run.build_head(case).eval()
    head.box_roi_pool = _StreamPool()
    shapes = [(5, 8), (2, 3), (6, 9), (1, 4), (5, 7), (3, 3), (6, 6), (1, 1), (4, 9)]
    imgs = _image_stream(case, shapes, 400)
    want, wt = _loop(head, imgs, 31, ahead=False)
    st0 = dict(head.engine()._small.stats())
    got, gt = _loop(head, imgs, 31, ahead=True)
    for a, b in zip(got, want):
        _same(a, b)
    assert torch.equal(gt, wt)
    st = head.engine()._small.stats()
    assert st["look_ahead_hits"] - st0["look_ahead_hits"] == len(imgs) - 1, st
    assert st["captures"] == st0["captures"]                 # the second pass replayed the first pass's plans


def test_an_unclaimed_look_ahead_leaves_no_trace():
    """A look-ahead for detections the next forward is NOT called with (a loop that skips an image, an interleaved training
    step, an eager-size batch) is dropped: the generator goes back to where it stood, the forward prepares for itself."""
    case = cases.build_case("tiny")
    head = gpu_run.build_head(case).eval()
    head.box_roi_pool = _StreamPool()
    imgs = _image_stream(case, [(5, 8), (2, 3), (6, 9), (3, 4)], 500)
    decoys = [d for d, _, _ in _image_stream(case, [(4, 4), (1, 2), (2, 2)], 600)]
    want, wt = _loop(head, imgs, 32, ahead=False)
    got, gt = _loop(head, imgs, 32, ahead=True, decoys=decoys)
    for a, b in zip(got, want):
        _same(a, b)
    assert torch.equal(gt, wt)
    assert head.engine()._small.stats()["look_ahead_hits"] == 0
    # training mode / batches of several images: nothing to look ahead for
    assert head.prefetch_eval(imgs[0][0] + imgs[1][0]) is False
    # detections the forward will refuse: the look-ahead declines quietly, the forward raises from its usual place
    big = [dict(boxes=torch.rand(1100, 4).cuda() * 100, labels=torch.full((1100,), 49).cuda(), scores=torch.rand(1100).cuda())]
    assert head.prefetch_eval(big) is False
    assert head.prefetch_eval([{k: v.cpu() for k, v in imgs[0][0][0].items()}]) is False
    from skghoi_amd import _capi
    with pytest.raises(_capi.SkgError):
        with torch.no_grad():
            head(imgs[0][2], big, [(800, 1200)])
    head.train()
    assert head.prefetch_eval(imgs[0][0]) is False
    head.eval()
    # ... and a pending look-ahead is dropped by a forward of another kind (here: the eager path of a larger batch size cap)
    torch.manual_seed(5)
    assert head.prefetch_eval(imgs[0][0]) is True
    head.engine().small_batch_max = 0
    head.box_roi_pool.pooled = imgs[1][1]
    with torch.no_grad():
        r1 = head(imgs[1][2], imgs[1][0], [(800, 1200)])
    t1 = torch.empty(2).uniform_()
    torch.manual_seed(5)
    with torch.no_grad():
        r2 = head(imgs[1][2], imgs[1][0], [(800, 1200)])
    t2 = torch.empty(2).uniform_()
    _same(r1, r2)
    assert torch.equal(t1, t2)


def test_trainer_test_loop_with_and_without_look_ahead():
    """trainer.test (utils.py:148-198) over a loader of single cached images: the device evaluator and the host restatement
    agree, with the one-image look-ahead and without."""
    from skghoi_amd import evaluate, trainer
    case = dict(cases.build_case("tiny"))
    case["o2v"] = synth.hico_object_to_verb()               # HICO-DET's valid (object, verb) pairs: every scored cell is an HOI
    head = gpu_run.build_head(case).eval()
    shapes = [(5, 8), (2, 3), (6, 9), (1, 4), (3, 3), (4, 6)]
    lut = evaluate.hico_object_n_verb_to_interaction()
    raw = []
    for i, (nh, no) in enumerate(shapes):
        im = synth.make_image(7400 + i, n_h=nh, n_o=no, out_channels=case["C"], pool=case["p"])
        det = dict(boxes=im["boxes"], labels=im["labels"], scores=im["scores"])
        tg = synth.make_targets(det, 49, synth.hico_object_to_verb(), 900 + i, n_gt=3)
        hoi = lut[tg["object"], tg["labels"]]
        keep = hoi >= 0
        target = dict(boxes_h=tg["boxes_h"][keep], boxes_o=tg["boxes_o"][keep], hoi=hoi[keep].long())
        raw.append((im, det, target))
    num_gt = [0] * 600
    for _, _, t in raw:
        for h in t["hoi"].tolist():
            num_gt[h] += 1

    class Loader:
        def __iter__(self):
            for im, det, target in raw:
                # (the pooled rows ride with the batch; a pool module that looks them up by identity of the feature map)
                yield (OrderedDict((k, im["feat3"]) for k in "0123"), [det], [im["hw"]], [target])

    class Pool(torch.nn.Module):
        def forward(self, features, boxes, image_shapes):
            f = features["3"]
            for im, _, _ in raw:
                if f.shape == im["feat3"].shape and torch.equal(f.cpu(), im["feat3"]):
                    return im["pooled"].cuda()
            raise AssertionError("unknown image")

    head.box_roi_pool = Pool()
    summaries = []
    for look, dev_eval in ((False, False), (True, True), (True, False), (False, True)):
        ev = evaluate.DeviceHOIEvaluator(num_gt, lut) if dev_eval else evaluate.HOIEvaluator(num_gt, lut)
        torch.manual_seed(77)
        summaries.append(trainer.test(head, Loader(), ev, device="cuda", lookahead=look))
    for s in summaries[1:]:
        assert torch.allclose(s["ap"].double(), summaries[0]["ap"].double(), atol=1e-9)
    assert head.engine()._small.stats()["look_ahead_hits"] == 2 * (len(shapes) - 1)
