#!/usr/bin/env python3
"""End-to-end walk through the hot path and its neighbours on synthetic data (needs an MI355X):

  FPN-like feature maps + detections -> producer (device NMS/top-k, MultiScaleRoIAlign, global pool) -> feature shard
  + per-image detection JSON on disk -> batched inference from the cache -> 600-way HOI scores -> 11-point mAP
  against synthetic ground truth -> HICO-DET .mat cells / V-COCO-style pickle.

Mirrors what a user of the reference does with hicodet/detections/adamixer_preprocessing.py, utils.test and cache.py.
"""
import argparse
import os
import sys
import tempfile
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from skghoi_amd import GraphHead, InteractionHead, cache, evaluate, runtime, synth
from skghoi_amd.roi_pool import MultiScaleRoIAlign

runtime.configure()           # process-level HIP runtime settings, before the first GPU use


def main(n_images=16, batch=8, out_dir=None, seed=0):
    dev = torch.device("cuda", 0)
    out_dir = out_dir or tempfile.mkdtemp(prefix="skg_cache_")
    o2v = synth.hico_object_to_verb()
    lut = evaluate.hico_object_n_verb_to_interaction()
    gh = GraphHead(256, 7, 1024, 1024, 117, 49, o2v)
    head = InteractionHead(MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2), gh, torch.nn.Linear(2048, 1),
                           torch.nn.Linear(2048, 117), human_idx=49, num_classes=117).to(dev).eval()
    head.load_state_dict(synth.make_state_dict(117, 256, 7, seed=seed))
    H, W = 800, 1216
    g = torch.Generator().manual_seed(seed)
    imgs = [synth.make_image(5000 + i, n_h=4, n_o=6, hw=(H, W)) for i in range(n_images)]
    dets = [dict(boxes=i["boxes"].to(dev), labels=i["labels"].to(dev), scores=i["scores"].to(dev)) for i in imgs]
    shapes = [(H, W)] * n_images
    # ---- producer: one shard per batch + the reference's per-image detection JSON
    shards = []
    for lo in range(0, n_images, batch):
        hi = min(n_images, lo + batch)
        feats = OrderedDict((str(l), torch.randn(hi - lo, 256, H // s, W // s, generator=g).to(dev))
                            for l, s in enumerate((4, 8, 16, 32)))
        path = os.path.join(out_dir, "shard_%04d.skgfc" % lo)
        kept = cache.produce_shard(head, feats, dets[lo:hi], shapes[lo:hi], path)
        for k, d in enumerate(kept):
            cache.write_detections_json(os.path.join(out_dir, "img_%06d.json" % (lo + k)), d["boxes"].cpu(),
                                        d["scores"].cpu(), d["labels"].cpu())
        shards.append((lo, hi, path))
    # ---- inference from the cache
    pool = cache.CachedPool()
    head.box_roi_pool = pool
    outputs = []
    torch.manual_seed(seed)
    with torch.no_grad():
        for lo, hi, path in shards:
            sh = cache.FeatureShard(path)
            pooled, gl, hw, counts = sh.batch(0, hi - lo, dev)
            det = [cache.read_detections_json(os.path.join(out_dir, "img_%06d.json" % i), dev) for i in range(lo, hi)]
            pool.pooled = pooled
            outputs += head({"3": gl}, det, hw)
    # ---- synthetic ground truth: the top-scoring cell of every image is "correct"
    num_gt = [0] * 600
    targets = []
    for out in outputs:
        j = int(out["scores"].argmax())
        p = int(out["index"][j])
        hoi = int(lut[int(out["object"][p]), int(out["prediction"][j])])
        num_gt[hoi] += 1
        targets.append(dict(boxes_h=out["boxes_h"][p:p + 1].cpu(), boxes_o=out["boxes_o"][p:p + 1].cpu(),
                            hoi=torch.tensor([hoi])))
    ev = evaluate.HOIEvaluator(num_gt, lut)
    for out, t in zip(outputs, targets):
        ev.add(out, t)
    summ = ev.summary()
    cells = evaluate.hicodet_mat_cells(outputs, list(range(n_images)), n_images, lut)
    actions = ["verb%d obj" % v for v in range(117)]
    evaluate.save_vcoco_pickle(evaluate.vcoco_results(outputs[:2], [1, 2], actions), out_dir)
    n_cls = int((summ["ap"] > 0).sum())
    print("images %d, scored cells %d, classes with AP>0: %d, mAP over classes with GT: %.4f, cache dir %s" % (
        n_images, sum(len(o["scores"]) for o in outputs), n_cls,
        float(summ["ap"][torch.tensor(num_gt) > 0].mean()), out_dir))
    return outputs, summ, cells


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    main(a.images, a.batch, a.out)
