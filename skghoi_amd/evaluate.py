"""600-way HOI mapping, box-pair association, 11-point mAP, and the HICO-DET / V-COCO result exporters
 -- the consumers of the interaction head's result dicts.

Reference: utils.py:148-198 (`test`: per-image association + pocket.utils.DetectionAPMeter(600, algorithm='11P')),
test/adamixer_transH_spatital_r50_test.py:30-33, 66-70 (rare / non-rare split at 10 training annotations),
cache.py:28-95 (per-object .mat files of [boxes_h | boxes_o | score] in pixel-index convention) and cache.py:97-143 +
cache_template.py:2-15 (V-COCO pickle of CacheTemplate dicts, protocol 2).  `pocket` (BoxPairAssociation,
DetectionAPMeter) is a third-party package absent from the image and unpinned by the reference; its published
semantics are restated here -- parity is unpinned at that boundary (oracle: oracle/eval_oracle.py, plain loops).

Everything works on whatever device the result tensors live on (the association is a handful of small IoU matrices);
nothing here is on the timed hot path.
"""
import json
import os
import pickle

import numpy as np
import torch

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def hico_object_n_verb_to_interaction():
    """[80][117] LUT -> HOI index or -1 (HICODet.object_n_verb_to_interaction, hicodet/hicodet.py:139-153), built from
    the 600 (verb, object) pairs in HOI order (skghoi_amd/data/hico_object_to_verb.json)."""
    with open(os.path.join(_DATA, "hico_object_to_verb.json")) as f:
        pairs = json.load(f)["hoi_verb_object"]
    lut = torch.full((80, 117), -1, dtype=torch.int64)
    for i, (v, o) in enumerate(pairs):
        lut[o, v] = i
    return lut


def box_iou(b1, b2):
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1]); a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2]); rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


def associate_pairs(gt_h, gt_o, det_h, det_o, scores, min_iou=0.5):
    """pocket.utils.BoxPairAssociation(min_iou)((gt_h, gt_o), (det_h, det_o), scores) -> binary labels [n_det].

    A detected pair matches the ground-truth pair with which min(IoU_h, IoU_o) is largest, if that is >= min_iou; among
    the detections matched to one ground-truth pair the highest-scoring one is the true positive."""
    n_det = det_h.shape[0]
    labels = torch.zeros(n_det, device=scores.device, dtype=scores.dtype)
    if n_det == 0 or gt_h.shape[0] == 0:
        return labels
    iou = torch.min(box_iou(gt_h, det_h), box_iou(gt_o, det_o))            # [n_gt, n_det]
    max_iou, max_idx = iou.max(0)
    match = torch.where(max_iou >= min_iou, max_idx, torch.full_like(max_idx, -1))
    for g in match.unique().tolist():
        if g < 0:
            continue
        det = torch.nonzero(match == g).squeeze(1)
        labels[det[scores[det].argmax()]] = 1
    return labels


class DetectionAPMeter:
    """pocket.utils.DetectionAPMeter(num_cls, num_gt=..., algorithm='11P'): per-class lists of (score, label),
    AP = mean over t in {0, 0.1, ..., 1} of max precision at recall >= t (float64)."""

    def __init__(self, num_cls, num_gt=None, algorithm="11P"):
        if algorithm not in ("11P", "AUC"):
            raise ValueError("unknown algorithm %s" % algorithm)
        self.num_cls = num_cls
        # num_gt=None (the training / validation meter of utils.py:208, 285): recall is taken over the positives among the
        # logged detections of the class
        self.num_gt = None if num_gt is None else [int(v) for v in num_gt]
        self.algorithm = algorithm
        self.reset()

    def reset(self):
        self._scores = [[] for _ in range(self.num_cls)]
        self._labels = [[] for _ in range(self.num_cls)]

    def append(self, scores, classes, labels):
        scores = scores.detach().cpu().double(); classes = classes.detach().cpu().long(); labels = labels.detach().cpu().double()
        for c in classes.unique().tolist():
            m = classes == c
            self._scores[c].append(scores[m]); self._labels[c].append(labels[m])

    @staticmethod
    def _ap(scores, labels, num_gt, algorithm):
        if num_gt == 0 or scores.numel() == 0:
            return 0.0
        order = torch.argsort(scores, descending=True, stable=True)
        lab = labels[order]
        tp = torch.cumsum(lab, 0); fp = torch.cumsum(1 - lab, 0)
        prec = tp / (tp + fp); rec = tp / num_gt
        if algorithm == "11P":
            ap = 0.0
            for t in torch.linspace(0, 1, 11, dtype=torch.float64):
                m = rec >= t
                if m.any():
                    ap += float(prec[m].max()) / 11
            return ap
        # area under the interpolated precision-recall curve
        ap, prev_r, mx = 0.0, 0.0, 0.0
        for i in range(len(prec) - 1, -1, -1):
            mx = max(mx, float(prec[i])); prec[i] = mx
        for p, r in zip(prec.tolist(), rec.tolist()):
            ap += p * (r - prev_r); prev_r = r
        return ap

    def eval(self):
        out = torch.zeros(self.num_cls, dtype=torch.float64)
        for c in range(self.num_cls):
            if self._scores[c]:
                lab = torch.cat(self._labels[c])
                n_gt = int(lab.sum()) if self.num_gt is None else self.num_gt[c]
                out[c] = self._ap(torch.cat(self._scores[c]), lab, n_gt, self.algorithm)
        return out


class DeviceAPMeter:
    """The training-mAP / validation meter of the reference's engine (utils.py:208, 229, 263-282: every iteration the
    batch's (scores, prediction, labels) are moved to the host, all-gathered over the ranks and appended to a
    DetectionAPMeter(num_classes, algorithm='11P') on rank 0) without the per-iteration host synchronisation and collective:
    `append` keeps the batch's three DEVICE tensors (no copy, no sync: the step loop stays asynchronous); `eval()` -- called
    once, at the end of an epoch -- gathers the ranks' logs in one padded all_gather, sorts on the device and computes the
    per-class 11-point APs with skg_eval_ap11_f64 (float64), on every rank.  Same numbers as DetectionAPMeter over the same
    detections; ties between equal scores of one class resolve by rank, then arrival order (the reference: iteration, then
    rank).  num_gt=None like the reference's meters: recall over the positives among the logged detections."""

    def __init__(self, num_cls, num_gt=None, device="cuda", group=None):
        self.num_cls = int(num_cls)
        self.num_gt = None if num_gt is None else [int(v) for v in num_gt]
        self.device = torch.device(device)
        self.group = group
        self.reset()

    def reset(self):
        self._log = []

    def append(self, scores, classes, labels):
        if scores.numel():
            self._log.append((scores.detach(), classes.detach(), labels.detach()))

    def append_results(self, results):
        """results: the head's per-image dicts of one batch with `labels` (training mode, or eval mode with targets)."""
        res = [r for r in results if "labels" in r and r["scores"].numel()]
        if res:
            self.append(torch.cat([r["scores"] for r in res]), torch.cat([r["prediction"] for r in res]),
                        torch.cat([r["labels"] for r in res]))

    def __len__(self):
        return sum(int(s.shape[0]) for s, _, _ in self._log)

    def eval(self):
        import torch.distributed as dist
        from . import _capi
        from .engine import _stream
        dev = self.device
        if self._log:
            mine = torch.stack([torch.cat([s.to(dev).float() for s, _, _ in self._log]),
                                torch.cat([c.to(dev).float() for _, c, _ in self._log]),
                                torch.cat([l.to(dev).float() for _, _, l in self._log])], dim=1)      # [L, 3]
        else:
            mine = torch.zeros(0, 3, device=dev)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            from .dist import _all_gather_ragged
            mine = torch.cat(_all_gather_ragged(mine.contiguous(), self.group))
        ap = torch.zeros(self.num_cls, dtype=torch.float64, device=dev)
        if mine.shape[0] == 0:
            return ap.cpu()
        if dev.type != "cuda":                                  # (CPU tensors: rehearsals over gloo) the host meter
            m = DetectionAPMeter(self.num_cls, self.num_gt)
            m.append(mine[:, 0], mine[:, 1].long(), mine[:, 2])
            return m.eval()
        scores, cls, labels = mine[:, 0].contiguous(), mine[:, 1].long(), mine[:, 2].contiguous()
        if ((cls < 0) | (cls >= self.num_cls)).any():
            raise IndexError("a logged prediction is outside [0, %d)" % self.num_cls)
        o1 = torch.sort(scores, descending=True, stable=True).indices
        o2 = torch.sort(cls[o1], stable=True).indices
        order = o1[o2]
        lab_sorted = labels[order].contiguous()
        class_off = torch.zeros(self.num_cls + 1, dtype=torch.int64, device=dev)
        class_off[1:] = torch.cumsum(torch.bincount(cls, minlength=self.num_cls), 0)
        if self.num_gt is None:
            num_gt = torch.zeros(self.num_cls, device=dev, dtype=torch.float64).index_add_(0, cls, labels.double()).long()
        else:
            num_gt = torch.as_tensor(self.num_gt, dtype=torch.int64, device=dev)
        thr = torch.linspace(0, 1, 11, dtype=torch.float64).to(dev)
        _capi.check(_capi.lib().skg_eval_ap11_f64(lab_sorted.data_ptr(), class_off.data_ptr(), num_gt.data_ptr(),
                                                  self.num_cls, thr.data_ptr(), ap.data_ptr(), _stream()),
                    "skg_eval_ap11_f64")
        return ap.cpu()


class HOIEvaluator:
    """The loop body of utils.test (utils.py:170-196) for any batch size: feed the head's result dict and the image's
    target (`boxes_h`, `boxes_o`, `hoi`), read full / rare / non-rare mAP at the end."""

    def __init__(self, num_gt_test, object_n_verb_to_interaction=None, min_iou=0.5, num_anno_train=None):
        self.lut = object_n_verb_to_interaction if object_n_verb_to_interaction is not None \
            else hico_object_n_verb_to_interaction()
        self.meter = DetectionAPMeter(int(self.lut.max()) + 1, num_gt=num_gt_test, algorithm="11P")
        self.min_iou = min_iou
        self.num_anno_train = None if num_anno_train is None else torch.as_tensor(num_anno_train)

    def interactions_of(self, output):
        idx = output["index"]
        return self.lut.to(idx.device)[output["object"][idx], output["prediction"]]

    def add(self, output, target):
        idx = output["index"]
        boxes_h = output["boxes_h"][idx]; boxes_o = output["boxes_o"][idx]
        scores = output["scores"]
        inter = self.interactions_of(output)
        if (inter < 0).any():
            raise IndexError("a predicted (object, verb) pair is not a valid interaction")
        labels = torch.zeros_like(scores)
        t_hoi = target["hoi"].to(scores.device)
        for h in inter.unique().tolist():
            gt = torch.nonzero(t_hoi == h).squeeze(1)
            det = torch.nonzero(inter == h).squeeze(1)
            if len(gt):
                labels[det] = associate_pairs(target["boxes_h"].to(scores.device)[gt].view(-1, 4),
                                              target["boxes_o"].to(scores.device)[gt].view(-1, 4),
                                              boxes_h[det].view(-1, 4), boxes_o[det].view(-1, 4),
                                              scores[det].view(-1), self.min_iou)
        self.meter.append(scores, inter, labels)
        return labels

    def summary(self):
        ap = self.meter.eval()
        out = dict(ap=ap, full=float(ap.mean()))
        if self.num_anno_train is not None:                      # test/..._test.py:30-33
            rare = torch.nonzero(self.num_anno_train < 10).squeeze(1)
            non_rare = torch.nonzero(self.num_anno_train >= 10).squeeze(1)
            out["rare"] = float(ap[rare].mean()); out["non_rare"] = float(ap[non_rare].mean())
        return out


class DeviceHOIEvaluator:
    """utils.test (utils.py:148-198) on the device, for batches of any size: interaction lookup + box-pair association in
    one launch per batch (skg_eval_associate_f32), detections kept on the device, and at the end two stable device sorts
    plus one launch for the 600 eleven-point APs (skg_eval_ap11_f64, float64).  Same numbers as HOIEvaluator (the host
    restatement the oracle pins); use it when the head runs batched and the results should not travel to the host."""

    def __init__(self, num_gt_test, object_n_verb_to_interaction=None, min_iou=0.5, num_anno_train=None, device="cuda"):
        self.device = torch.device(device)
        lut = object_n_verb_to_interaction if object_n_verb_to_interaction is not None \
            else hico_object_n_verb_to_interaction()
        self.n_obj, self.n_verb = lut.shape
        self.num_cls = int(lut.max()) + 1
        self.lut = lut.to(self.device, torch.int32).contiguous()
        self.num_gt = torch.as_tensor([int(v) for v in num_gt_test], dtype=torch.int64, device=self.device)
        self.min_iou = float(min_iou)
        self.num_anno_train = None if num_anno_train is None else torch.as_tensor(num_anno_train)
        self.thr = torch.linspace(0, 1, 11, dtype=torch.float64).to(self.device)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._scores, self._hoi, self._labels = [], [], []

    def add(self, outputs, targets):
        """outputs: the head's result dicts of a batch (HEAD:317-322); targets: per image {boxes_h, boxes_o, hoi}.
        Returns the per-image label tensors (views of one device tensor)."""
        from . import _capi
        from .engine import _stream
        dev = self.device
        n = len(outputs)
        if n == 0:
            return []
        ppi = [int(o["boxes_h"].shape[0]) for o in outputs]; cpi = [int(o["scores"].shape[0]) for o in outputs]
        cat = lambda k, dt=None: torch.cat([o[k].reshape(-1, *o[k].shape[1:]) for o in outputs]).to(dev).contiguous()
        boxes_h = cat("boxes_h").float(); boxes_o = cat("boxes_o").float(); obj = cat("object").long()
        index = cat("index").long(); pred = cat("prediction").long(); scores = cat("scores").float().contiguous()
        gts = [int(t["boxes_h"].shape[0]) for t in targets]
        gt_h = torch.cat([t["boxes_h"].reshape(-1, 4) for t in targets]).to(dev).float().contiguous()
        gt_o = torch.cat([t["boxes_o"].reshape(-1, 4) for t in targets]).to(dev).float().contiguous()
        gt_hoi = torch.cat([t["hoi"].reshape(-1) for t in targets]).to(dev).long().contiguous()
        if gt_h.shape[0] == 0:
            gt_h = torch.zeros(1, 4, device=dev); gt_o = torch.zeros(1, 4, device=dev)
            gt_hoi = torch.full((1,), -1, dtype=torch.int64, device=dev)
        offs = np.zeros(3 * (n + 1), np.int32)
        offs[1:n + 1] = np.cumsum(ppi)                               # pair_off (n used) | cell_off | gt_off
        offs[n + 2:2 * n + 2] = np.cumsum(cpi); offs[2 * n + 3:] = np.cumsum(gts)
        offs_d = torch.from_numpy(offs).to(dev)
        L = int(sum(cpi))
        hoi = torch.empty(max(L, 1), dtype=torch.int32, device=dev); labels = torch.zeros(max(L, 1), device=dev)
        pad4 = lambda t: t if t.shape[0] else torch.zeros((1,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        boxes_h, boxes_o, obj, index, pred, scores = map(pad4, (boxes_h, boxes_o, obj, index, pred, scores))
        _capi.check(_capi.lib().skg_eval_associate_f32(
            boxes_h.data_ptr(), boxes_o.data_ptr(), obj.data_ptr(), offs_d.data_ptr(), index.data_ptr(), pred.data_ptr(),
            scores.data_ptr(), offs_d.data_ptr() + 4 * (n + 1), n, self.lut.data_ptr(), self.n_obj, self.n_verb,
            gt_h.data_ptr(), gt_o.data_ptr(), gt_hoi.data_ptr(), offs_d.data_ptr() + 8 * (n + 1), self.min_iou,
            hoi.data_ptr(), labels.data_ptr(), self.status.data_ptr(), _stream()), "skg_eval_associate_f32")
        self._scores.append(scores[:L]); self._hoi.append(hoi[:L]); self._labels.append(labels[:L])
        return list(labels[:L].split(cpi))

    def summary(self):
        from . import _capi
        from .engine import _stream
        dev = self.device
        if int(self.status.item()):
            raise _capi.SkgError("an image has %d ground-truth pairs; skg_eval_associate_f32 keeps at most 2048 per image"
                                 % int(self.status.item()))
        ap = torch.zeros(self.num_cls, dtype=torch.float64, device=dev)
        if self._scores:
            scores = torch.cat(self._scores); hoi = torch.cat(self._hoi).long(); labels = torch.cat(self._labels)
            if (hoi < 0).any():
                raise IndexError("a predicted (object, verb) pair is not a valid interaction")
            o1 = torch.sort(scores, descending=True, stable=True).indices      # score descending, arrival order on ties
            o2 = torch.sort(hoi[o1], stable=True).indices                      # ... then class ascending, order kept
            order = o1[o2]
            lab_sorted = labels[order].contiguous()
            class_off = torch.zeros(self.num_cls + 1, dtype=torch.int64, device=dev)
            class_off[1:] = torch.cumsum(torch.bincount(hoi, minlength=self.num_cls), 0)
            _capi.check(_capi.lib().skg_eval_ap11_f64(lab_sorted.data_ptr(), class_off.data_ptr(), self.num_gt.data_ptr(),
                                                      self.num_cls, self.thr.data_ptr(), ap.data_ptr(), _stream()),
                        "skg_eval_ap11_f64")
        ap = ap.cpu()
        out = dict(ap=ap, full=float(ap.mean()))
        if self.num_anno_train is not None:                      # test/..._test.py:30-33
            rare = torch.nonzero(self.num_anno_train < 10).squeeze(1)
            non_rare = torch.nonzero(self.num_anno_train >= 10).squeeze(1)
            out["rare"] = float(ap[rare].mean()); out["non_rare"] = float(ap[non_rare].mean())
        return out


# ----------------------------------------------------------------------------------------------- exporters
def hicodet_mat_cells(outputs, image_indices, n_images, lut=None, num_hoi=600):
    """cache.py:28-83: object array [num_hoi, n_images] whose cell (hoi, image) is [n, 9] = boxes_h | boxes_o | score
    in pixel-index convention (x2, y2 minus 1), (0, 0) arrays where empty.  `outputs[k]` belongs to image
    `image_indices[k]`."""
    lut = hico_object_n_verb_to_interaction() if lut is None else lut
    cells = np.empty((num_hoi, n_images), dtype=object)
    for out, j in zip(outputs, image_indices):
        idx = out["index"].cpu()
        bh = out["boxes_h"].cpu()[idx].clone(); bo = out["boxes_o"].cpu()[idx].clone()
        bh[:, 2:] -= 1; bo[:, 2:] -= 1
        inter = lut[out["object"].cpu()[idx], out["prediction"].cpu()]
        sc = out["scores"].cpu()
        perm = inter.argsort()
        bh, bo, inter, sc = bh[perm], bo[perm], inter[perm], sc[perm]
        cls, cnt = inter.unique(return_counts=True)
        n = 0
        for c, k in zip(cls.tolist(), cnt.tolist()):
            cells[c, j] = torch.cat([bh[n:n + k], bo[n:n + k], sc[n:n + k, None]], dim=1).numpy()
            n += k
    for i in range(num_hoi):
        for j in range(n_images):
            if cells[i, j] is None:
                cells[i, j] = np.zeros((0, 0))
    return cells


def save_hicodet_mats(cells, cache_dir, object_to_interaction, coco2hico):
    """cache.py:84-95: one detections_XX.mat per COCO object id with the rows of that object's interactions."""
    import scipy.io as sio
    for object_idx in coco2hico:
        sio.savemat(os.path.join(cache_dir, "detections_{}.mat".format(str(object_idx).zfill(2))),
                    dict(all_boxes=cells[object_to_interaction[coco2hico[object_idx]]]))


def _cache_template_cls():
    """The record class lives in the top-level module `cache_template` (same import path as the reference's), so the
    pickles are interchangeable with the reference's writer / vcoco_evaluation.py."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.append(root)
    return importlib.import_module("cache_template").CacheTemplate


def vcoco_results(outputs, image_ids, actions):
    """cache.py:117-137: one CacheTemplate per scored (pair, action); `actions[a]` is e.g. 'hold obj'."""
    allr = []
    CT = _cache_template_cls()
    for out, image_id in zip(outputs, image_ids):
        idx = out["index"].cpu()
        bh = out["boxes_h"].cpu()[idx]; bo = out["boxes_o"].cpu()[idx]
        for h, o, s, a in zip(bh, bo, out["scores"].cpu(), out["prediction"].cpu()):
            a_name = actions[int(a)].split()
            r = CT(image_id=image_id, person_box=h.tolist())
            r[a_name[0] + "_agent"] = s.item()
            r["_".join(a_name)] = o.tolist() + [s.item()]
            allr.append(r)
    return allr


def save_vcoco_pickle(results, cache_dir):
    """cache.py:141-143: protocol 2 for the Python-2 vsrl_eval."""
    with open(os.path.join(cache_dir, "vcoco_results.pkl"), "wb") as f:
        pickle.dump(results, f, 2)
