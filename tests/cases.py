"""Golden / parity cases for the interaction-head hot path (shared by the golden generator, the oracle tests and the
GPU parity tests).  All inputs come from numpy RandomState seeds (see skghoi_amd/synth.py), so only reference
OUTPUTS are committed under tests/golden/.

Edge cases follow SURVEY.md section 8(c): n_h == 0, n == 1, only humans, zero-area box (NaN scrub), score ties,
NMS-active input, V-COCO configuration (K = 24, human_idx = 1), skipped images inside a batch (Q9 offset bug).
"""
import numpy as np
import torch

from skghoi_amd import synth

HICO = dict(K=117, human_idx=49, num_obj=80)
VCOCO = dict(K=24, human_idx=1, num_obj=81)


def _o2v(cfg):
    if cfg["K"] == 117:
        return synth.hico_object_to_verb()
    return synth.random_object_to_verb(cfg["num_obj"], cfg["K"], per_class=5, seed=11)


def _det(img):
    return dict(boxes=img["boxes"], labels=img["labels"], scores=img["scores"])


def _grid_images(specs, C, p, human_idx, num_obj, seed0):
    return [synth.make_image(seed0 + i, n_h=nh, n_o=no, out_channels=C, pool=p, human_idx=human_idx,
                             num_obj_classes=num_obj) for i, (nh, no) in enumerate(specs)]


def _nms_image(C, p, human_idx):
    """Hand-built detections: duplicates (IoU > 0.5 same class), a low score, same box different class,
    score ties, more humans than max_human."""
    rs = np.random.RandomState(77)
    base = np.array([[50, 60, 250, 400], [60, 70, 255, 410],      # human dup pair (IoU ~0.87)
                     [300, 50, 500, 420], [700, 100, 900, 500], [920, 120, 1100, 480], [30, 450, 200, 780],
                     [400, 450, 600, 700], [405, 455, 600, 700],  # object dup pair, same class
                     [400, 450, 600, 700],                        # same box, different class -> kept
                     [650, 520, 800, 760], [820, 540, 1000, 770], [1020, 560, 1180, 790],
                     [210, 470, 380, 760], [300, 430, 395, 445]], np.float32)
    labels = np.array([human_idx, human_idx, human_idx, human_idx, human_idx, human_idx,
                       3, 3, 7, 12, 12, 30, 60, 61], np.int64)
    scores = np.array([0.9, 0.8, 0.7, 0.7, 0.6, 0.5,  0.85, 0.95, 0.4, 0.3, 0.15, 0.7, 0.7, 0.21], np.float32)
    n = len(base)
    return dict(boxes=torch.from_numpy(base), labels=torch.from_numpy(labels), scores=torch.from_numpy(scores),
                feat3=torch.from_numpy(rs.standard_normal((1, 256, 6, 9)).astype(np.float32)), hw=(800, 1200),
                pooled_seed=771)


def pooled_for(case, n_rows):
    """Pooled box features [n_rows, C, p, p] in the head's post-preprocess row order (the roi-pool stand-in)."""
    rs = np.random.RandomState(case["pooled_seed"])
    return torch.from_numpy(rs.standard_normal((n_rows, case["C"], case["p"], case["p"])).astype(np.float32))


def build_case(name):
    """Returns a dict: cfg (K, human_idx), o2v, C, p, max_human, max_object, num_iter, training,
    detections (list), feat3 [B,256,h,w], shapes, targets|None, weight_seed, rng_seed, pooled_seed."""
    c = dict(name=name, C=8, p=2, max_human=15, max_object=15, num_iter=2, training=False, targets=None,
             weight_seed=3, rng_seed=1234, pooled_seed=99, box_score_thresh=0.2, box_nms_thresh=0.5)
    # "<case>@<seed>": the same case with another image seed (developer aid: tools/debug_grad_case.py looks for seeds whose
    # gradients are reproducible across hosts -- no ReLU input within rounding noise of zero)
    name, _, alt_seed = name.partition("@")
    cfg = HICO
    small_feat = dict(feat_hw=(6, 9))
    if name == "tiny":
        imgs = _grid_images([(3, 4)], 8, 2, 49, 80, 100)
    elif name == "ragged3":
        imgs = _grid_images([(2, 3), (4, 0), (1, 2)], 8, 2, 49, 80, 200)
    elif name == "skips_eval":
        # eval + batch > 1 + a skipped image: the reference's postprocess zips over the (only-skipped) label list
        # (HEAD:298-310) and returns ONE empty result ("Q14")
        imgs = _grid_images([(0, 3), (2, 2)], 8, 2, 49, 80, 300)
    elif name == "skips_raise":
        # same quirk, other order: the first image has pairs -> IndexError at HEAD:327
        imgs = _grid_images([(2, 2), (0, 3)], 8, 2, 49, 80, 310)
    elif name == "train_skips":
        # training batch with skipped images: Q9 (no `counter += n` on skip, HEAD:829-839)
        imgs = _grid_images([(0, 3), (2, 2), (1, 0), (3, 1)], 8, 2, 49, 80, 320)
        c.update(training=True)
    elif name == "nanbox":
        imgs = _grid_images([(2, 3)], 8, 2, 49, 80, 400)
        b = imgs[0]["boxes"].clone()
        lab = imgs[0]["labels"]
        hi = int(torch.nonzero(lab == 49)[0]); oi = int(torch.nonzero(lab != 49)[0])
        b[hi, 2] = b[hi, 0]; b[hi, 3] = b[hi, 1]            # zero-area human (0/0 IoU on its self pair)
        b[oi, 2] = b[oi, 0]                                  # zero-width object
        imgs[0]["boxes"] = b
    elif name == "vcoco":
        cfg = VCOCO
        imgs = _grid_images([(2, 4), (3, 3)], 8, 2, 1, 81, 500)
    elif name == "nms":
        imgs = [_nms_image(8, 2, 49)]
        c.update(max_human=3, max_object=4)
    elif name == "iter1":
        imgs = _grid_images([(2, 2)], 8, 2, 49, 80, 600)
        c.update(num_iter=1)
    elif name == "iter0":
        imgs = _grid_images([(2, 2)], 8, 2, 49, 80, 600)
        c.update(num_iter=0)
    elif name == "full20":
        imgs = _grid_images([(20, 20)], 256, 7, 49, 80, 1000)
        c.update(C=256, p=7, max_human=20, max_object=20, weight_seed=0)
    elif name == "full15x2":
        imgs = _grid_images([(17, 18), (15, 12)], 256, 7, 49, 80, 1100)      # default top-15 truncation active
        c.update(C=256, p=7, weight_seed=0)
    elif name == "full20x3":
        # three full-width images walked in two chunks (chunk_images = 2): the BASELINE graph shape across a chunk
        # boundary, output-only fixture
        imgs = _grid_images([(20, 20), (20, 20), (20, 20)], 256, 7, 49, 80, 1200)
        c.update(C=256, p=7, max_human=20, max_object=20, weight_seed=0, chunk_images=2)
    elif name == "many8":
        # eight one-image chunks: more chunks than the ring of TransH staging buffers (4), drawn inline
        imgs = _grid_images([(2, 3), (1, 2), (3, 1), (2, 2), (1, 4), (4, 2), (2, 1), (3, 3)], 8, 2, 49, 80, 1300)
        c.update(chunk_images=1)
    elif name == "many12":
        # six two-image chunks drawn by the helper thread (more than INLINE_IMAGES images), ring wrap-around
        imgs = _grid_images([(2, 3), (1, 2), (3, 1), (2, 2), (1, 4), (4, 2), (2, 1), (3, 3), (1, 1), (2, 4), (3, 2),
                             (1, 3)], 8, 2, 49, 80, 1400)
        c.update(chunk_images=2)
    elif name == "eval_targets":
        # eval mode WITH targets (validation): labels are associated and the sampling RNG is consumed (HEAD:933-963)
        imgs = _grid_images([(3, 4), (2, 3)], 8, 2, 49, 80, 700)
        c.update(eval_targets=True)
    elif name == "train_tiny":
        imgs = _grid_images([(3, 4), (2, 3)], 8, 2, 49, 80, 700)
        c.update(training=True)
    elif name == "train_full20x4":
        # BASELINE config 3 at full size: the reference's per-GPU training batch (main:158) of four full-width 20 x 20
        # images with ground truth appended (M ~ 3200 grid rows: the 128 x 128 tiles, split-K targets, grouped dX|dW
        # launches and the N = 4096 fc_2 product of the fused step).  Output-only fixture + gradient samples taken from
        # the live reference's autograd.
        imgs = _grid_images([(20, 20)] * 4, 256, 7, 49, 80, 1500)
        c.update(C=256, p=7, max_human=20, max_object=20, weight_seed=0, training=True, n_gt=4)
    elif name == "train_vcoco":
        # BASELINE config 5's TRAINING half: the V-COCO head (K = 24, human_idx = 1: main:73-76, cache.py:165-168) at full
        # width -- the classifier block is 25 columns, every K-dependent shape of the step changes (logit leading dimension,
        # dW of the classifier, the TransH relation tables, the split targets).  Gradient samples from the live reference.
        # (image seed 1603: of 1600-1604 the best conditioned -- with 1600 and 1601 one fc_2 unit of attention_head sits within
        #  rounding noise of its ReLU's zero, and whichever side a host's matrix products land on decides 4e-3 of that
        #  branch's bias gradient: oracle on the GPU box's host vs the reference here, bit-identical losses;
        #  tools/debug_grad_case.py, tools/case_conditioning.py)
        cfg = VCOCO
        imgs = _grid_images([(12, 10), (7, 13)], 256, 7, 1, 81, int(alt_seed or 1603))
        c.update(C=256, p=7, weight_seed=0, training=True, n_gt=4)
    elif name == "train_ragged_full":
        # a ragged full-width training batch at the reference's DEFAULT caps (15 / 15): an image over both caps (17 humans
        # + GT, 18 objects: truncation), a SKIPPED image in the middle (no human: the Q9 offset bug at C = 256, p = 7 --
        # HEAD:829-839 -- shifts the pooled rows of everything behind it), a small one and a one-pair image.
        imgs = _grid_images([(17, 18), (0, 5), (3, 9), (1, 1)], 256, 7, 49, 80, int(alt_seed or 1700))
        c.update(C=256, p=7, weight_seed=0, training=True, n_gt=4)
    elif name == "train_no_positive":
        # humans only: pairs exist (human-human) but there is no ground truth, hence no positive pair in the whole batch --
        # the reference's TransH term raises (HEAD:207-235: an empty score vector viewed as [-1, 0])
        imgs = _grid_images([(3, 0), (2, 0)], 8, 2, 49, 80, 330)
        c.update(training=True)
    elif name == "train_random":
        # "train_random@<seed>": a seeded random training batch at test width -- 2 to 5 images with 0..5 humans and 0..6
        # objects each (images without a human are SKIPPED: Q9; an image of one human and nothing else has no pair either),
        # 1..5 ground-truth pairs per image, V-COCO's head every third seed.  No fixture: the GPU step is compared with the
        # oracle's autograd on the same host (tests/test_random_parity_gpu.py).
        seed = int(alt_seed or 0)
        rs = np.random.RandomState(77000 + seed)
        vcoco = seed % 3 == 2
        if vcoco:
            cfg = VCOCO
        while True:
            shapes = [(int(rs.randint(0, 6)), int(rs.randint(0, 7))) for _ in range(int(rs.randint(2, 6)))]
            if any(h >= 1 and o >= 1 for h, o in shapes):          # (a batch without a positive pair raises, HEAD:207-235)
                break
        imgs = _grid_images(shapes, 8, 2, 1 if vcoco else 49, 81 if vcoco else 80, 78000 + 10 * seed)
        c.update(training=True, n_gt=int(rs.randint(1, 6)), rng_seed=4000 + seed, weight_seed=3 + seed % 2)
    else:
        raise KeyError(name)
    c["cfg"] = cfg
    c["o2v"] = _o2v(cfg)
    c["detections"] = [_det(i) for i in imgs]
    c["feat3"] = torch.cat([i["feat3"] for i in imgs])
    c["shapes"] = [i["hw"] for i in imgs]
    if c["training"] or c.get("eval_targets"):
        c["targets"] = [synth.make_targets(d, cfg["human_idx"], c["o2v"], 900 + k, n_gt=c.get("n_gt", 3))
                        for k, d in enumerate(c["detections"])]
    return c


EVAL_CASES = ["tiny", "ragged3", "skips_eval", "nanbox", "vcoco", "nms", "iter1", "iter0", "full20", "full15x2",
              "eval_targets", "full20x3", "many8", "many12"]
TRAIN_CASES = ["train_tiny", "train_skips"]
FULL_TRAIN_CASE = "train_full20x4"        # full-size training step: losses / labels / samples + gradient samples
# every full-width (C = 256, p = 7) training fixture: the uniform BASELINE shape, the V-COCO head, a ragged batch with a skip
FULL_TRAIN_CASES = [FULL_TRAIN_CASE, "train_vcoco", "train_ragged_full"]
OUTPUT_ONLY = ["full20", "full15x2", "full20x3", "many8", "many12"] + FULL_TRAIN_CASES   # no bulky intermediates
GRAD_SAMPLES = 512                        # entries kept per parameter gradient in those fixtures (evenly strided)
RAISING_CASES = ["skips_raise"]
ALL_CASES = EVAL_CASES + TRAIN_CASES + FULL_TRAIN_CASES


def grad_sample(flat_grad):
    """The fixed sample of a parameter gradient the full-size fixture stores: GRAD_SAMPLES evenly strided entries of the
    flattened tensor (all of it when smaller)."""
    n = flat_grad.shape[0]
    stride = max(1, n // GRAD_SAMPLES)
    return flat_grad[::stride][:GRAD_SAMPLES]
