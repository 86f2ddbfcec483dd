"""Randomised GPU parity of the integer / compare-heavy kernels against the oracle (bit-exact bar):
class-wise NMS + top-k selection on cluttered detections, pair enumeration, 46-d spatial encodings on arbitrary
(also degenerate) boxes, and the scoring/compaction kernel via the full head with random object->verb tables."""
import numpy as np
import pytest
import torch

import cases
import gpu_run
import helpers
from oracle import skg_oracle as O
from skghoi_amd import synth

pytestmark = pytest.mark.gpu


def _cluttered(rs, n, human_idx, n_cls):
    """Boxes in clusters (heavy overlap inside a cluster), few classes, quantised scores (many ties)."""
    k = max(1, n // 4)
    centres = rs.uniform(50, 700, (k, 2)); sizes = rs.uniform(30, 300, (k, 2))
    c = rs.randint(0, k, n)
    xy = centres[c] + rs.normal(0, 12, (n, 2)); wh = np.abs(sizes[c] + rs.normal(0, 15, (n, 2))) + 1
    boxes = np.concatenate([xy - wh / 2, xy + wh / 2], 1).astype(np.float32)
    labels = rs.choice([human_idx, human_idx, 3, 7, 11][:n_cls], n).astype(np.int64)
    scores = (rs.randint(1, 20, n) / 20.0).astype(np.float32)
    return dict(boxes=torch.from_numpy(boxes), labels=torch.from_numpy(labels), scores=torch.from_numpy(scores))


@pytest.mark.parametrize("seed", range(4))
def test_preprocess_nms_topk_bit_exact(seed):
    rs = np.random.RandomState(100 + seed)
    dets = [_cluttered(rs, int(n), 49, 5) for n in rs.randint(0, 120, 24)]
    dets.append(_cluttered(rs, 700, 49, 3))                      # large candidate set (bitonic sort of 1024)
    case = cases.build_case("tiny")
    case.update(max_human=int(rs.randint(1, 8)), max_object=int(rs.randint(1, 8)),
                box_score_thresh=float(rs.choice([0.2, 0.35, 0.5])), box_nms_thresh=float(rs.choice([0.3, 0.5, 0.7])))
    head = gpu_run.build_head(case)
    got = head.preprocess(gpu_run.to_cuda(dets), None)
    want = O.preprocess(dets, None, 49, case["box_score_thresh"], case["box_nms_thresh"], case["max_human"],
                        case["max_object"])
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert torch.equal(g["labels"].cpu(), w["labels"])
        assert torch.equal(g["boxes"].cpu(), w["boxes"]) and torch.equal(g["scores"].cpu(), w["scores"])


@pytest.mark.parametrize("seed", range(6))
def test_random_graphs_full_head_vs_oracle(seed, precision):
    """Random (n_h, n_o), random verb tables, overlapping boxes incl. zero-size ones: indices bit-exact, logits 1e-4."""
    rs = np.random.RandomState(200 + seed)
    case = cases.build_case("tiny")
    K = 24 if seed == 1 else 117
    if K == 24:
        case["cfg"] = cases.VCOCO
    case["o2v"] = synth.random_object_to_verb(81, K, per_class=int(rs.randint(1, 6)), seed=seed)
    hidx = case["cfg"]["human_idx"]
    dets = []
    for _ in range(4):
        d = _cluttered(rs, int(rs.randint(2, 30)), hidx, 5)
        d["labels"][d["labels"] == 49] = hidx
        if rs.rand() < 0.5 and len(d["boxes"]):
            d["boxes"][0, 2:] = d["boxes"][0, :2]                 # zero-area box
        dets.append(d)
    case["detections"] = dets
    case["feat3"] = torch.from_numpy(rs.standard_normal((4, 256, 3, 5)).astype(np.float32))
    case["shapes"] = [(int(rs.randint(400, 900)), int(rs.randint(400, 1300))) for _ in range(4)]
    case["box_nms_thresh"] = 0.6
    case["max_human"], case["max_object"] = 6, 9
    try:
        want = helpers.flatten_oracle(case, *helpers.run_oracle(case))
    except IndexError:
        with pytest.raises(IndexError):
            gpu_run.run_head(case)
        return
    got = gpu_run.run_head(case)
    for k, w in want.items():
        if k.startswith("res") and k.split(".")[1] in ("index", "prediction", "object"):
            assert np.array_equal(got[k], w), k
    scale = max(1.0, float(np.abs(want["logits_p"]).max())) if want["logits_p"].size else 1.0
    if want["logits_p"].size:
        assert np.abs(got["logits_p"] - want["logits_p"]).max() <= 1e-4 * scale
    for b in range(int(want["n_results"])):
        if want["res%d.scores" % b].size:
            assert np.abs(got["res%d.scores" % b] - want["res%d.scores" % b]).max() <= 1e-5


# seeds of cases.build_case("train_random@<seed>") whose gradients are well conditioned as a cross-device comparison (no
# ReLU input within rounding noise of zero: tools/case_conditioning.py reports <= 3.3e-5 for these, 5e-3 for seed 2)
TRAIN_SEEDS = [4, 5, 6, 8, 9, 10]


@pytest.mark.parametrize("seed", TRAIN_SEEDS)
def test_random_training_batches_match_oracle_autograd(seed):
    """Seeded random training batches (2-5 images of 0..5 humans / 0..6 objects, images without a human or without ground
    truth, HICO-DET's and V-COCO's head): association labels and result indices bit-exact, the three losses to 1e-5, the
    gradients of all 408 parameters against CPU autograd of the oracle -- the trainer's route (gradients written straight
    into the arena, no autograd engine) on even seeds, the autograd-backed fused step on odd ones."""
    case = cases.build_case("train_random@%d" % seed)
    mode = "direct" if seed % 2 == 0 else True
    flat, grads = gpu_run.run_train_with_grads(case, fused=mode)
    want, losses, oflat = helpers.oracle_train_grads(case, with_flat=True)
    for k in ("hoi_loss", "interactiveness_loss", "transH_loss"):
        assert abs(float(flat[k]) - losses[k]) <= 1e-5 * max(1.0, abs(losses[k])), (k, float(flat[k]), losses[k])
    assert int(flat["n_results"]) == int(oflat["n_results"])
    for b in range(int(oflat["n_results"])):
        for k in ("index", "prediction", "object", "labels", "unary_labels"):
            assert np.array_equal(flat["res%d.%s" % (b, k)], oflat["res%d.%s" % (b, k)]), (b, k)
        assert np.abs(flat["res%d.scores" % b] - oflat["res%d.scores" % b]).max(initial=0.0) <= 1e-5
    assert set(want) == set(grads), set(want) ^ set(grads)
    worst = (0.0, "")
    for k, w in want.items():
        scale = max(np.abs(w).max(), 1e-6)
        err = max(np.abs(grads[k] - w).max() - 1e-9, 0.0) / scale
        worst = max(worst, (err, k))
        # the adjacency weight's gradient is a sum of large cancelling terms at this width: a few ulp on the pooled features
        # move it by up to 3e-5 of its scale under the oracle itself (same tool), hence the wider bar for that one tensor
        bar = 3e-4 if k == "box_pair_head.adjacency.weight" else 1e-4
        assert err <= bar, "%s: rel err %.3e (|grad| max %.3e)" % (k, err, scale)
    print("seed %d: max relative gradient error %.3e (%s)" % (seed, worst[0], worst[1]))


def test_training_batch_without_a_positive_pair_ends_like_the_reference():
    """No positive pair in the whole batch (here: humans only, so no ground truth): the reference divides both focal terms by
    n_p = 0 (HEAD:165, 192-200) -- its training loop then stops at the NaN guard (utils.py:219) -- and the TransH term, in
    the semantics the oracle restates, views an empty score vector as [-1, 0] and raises RuntimeError.  The
    head must end the same way on every training route: RuntimeError, or losses that are not finite -- never a fault, a
    hang or a finite loss."""
    case = cases.build_case("train_no_positive")
    with pytest.raises(RuntimeError):
        helpers.oracle_train_grads(case)
    for mode in ("direct", True, False):
        try:
            flat, _ = gpu_run.run_train_with_grads(cases.build_case("train_no_positive"), fused=mode)
        except RuntimeError:
            continue
        assert not np.isfinite(float(flat["hoi_loss"])), (mode, float(flat["hoi_loss"]))
    torch.cuda.synchronize()
