"""Drop-in replacement for the reference module `heads/adamixer_transH_spatial_r50_head.py`.

Exports the same names with the same constructor keywords, forward signatures, return structures and state_dict keys
(SURVEY.md section 8b / Appendix A), so `models/adamixer_transH_spatial_r50_models.py:25`

    from adamixer_transH_spatial_r50_head import InteractionHead, GraphHead

keeps working when this directory (or <repo>/heads, which re-exports it) is put on sys.path instead of the
reference's.  The arithmetic runs in hand-written gfx950 kernels (libskghoi_hip.so, include/skghoi.h) driven by
skghoi_amd/engine.py; there is no eager/CPU fallback -- CPU tensors or a missing library raise.

Reference file:line for every class is given in its docstring (HEAD = the reference head file).
"""
from collections import OrderedDict
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist
from torch import nn, Tensor
from torch.nn import Module

from skghoi_amd import _capi
from skghoi_amd.engine import HeadEngine, current_stream_of, on_device, on_stream

__all__ = ["InteractionHead", "GraphHead", "MultiBranchFusion", "MessageMBF", "transH_head"]


class MultiBranchFusion(Module):
    """Parameter container of the reference's MultiBranchFusion (HEAD:431-474): `cardinality` branches of
    fc_1 (appearance -> sub), fc_2 (spatial -> sub), fc_3 (sub -> representation).  The engine runs the 16 branches
    as three stacked GEMMs; calling the module directly does the same for a single (appearance, spatial) pair."""

    def __init__(self, appearance_size: int, spatial_size: int, representation_size: int, cardinality: int) -> None:
        super().__init__()
        self.cardinality = cardinality
        sub_repr_size = int(representation_size / cardinality)
        assert sub_repr_size * cardinality == representation_size, \
            "The given representation size should be divisible by cardinality"
        self.fc_1 = nn.ModuleList([nn.Linear(appearance_size, sub_repr_size) for _ in range(cardinality)])
        self.fc_2 = nn.ModuleList([nn.Linear(spatial_size, sub_repr_size) for _ in range(cardinality)])
        self.fc_3 = nn.ModuleList([nn.Linear(sub_repr_size, representation_size) for _ in range(cardinality)])

    def _stacked(self, dev):
        w1 = torch.cat([l.weight for l in self.fc_1]).detach().to(dev).float().contiguous()
        b1 = torch.cat([l.bias for l in self.fc_1]).detach().to(dev).float()
        w2 = torch.cat([l.weight for l in self.fc_2]).detach().to(dev).float().contiguous()
        b2 = torch.cat([l.bias for l in self.fc_2]).detach().to(dev).float()
        w3 = torch.cat([l.weight for l in self.fc_3], dim=1).detach().to(dev).float().contiguous()
        b3 = torch.stack([l.bias for l in self.fc_3]).sum(0).detach().to(dev).float()
        return w1, b1, w2, b2, w3, b3

    def forward(self, appearance: Tensor, spatial: Tensor) -> Tensor:
        """HEAD:469-474 for 2-D inputs; appearance may have one row (broadcast) or as many rows as spatial."""
        from skghoi_amd.engine import gemm
        if appearance.device.type != "cuda":
            raise _capi.SkgError("MultiBranchFusion runs on a HIP device only")
        dev = spatial.device
        w1, b1, w2, b2, w3, b3 = self._stacked(dev)
        M = spatial.shape[0]
        R = w3.shape[0]
        ka = (appearance.shape[1] + 3) // 4 * 4
        a = torch.zeros(appearance.shape[0], ka, device=dev); a[:, :appearance.shape[1]] = appearance
        w1p = torch.zeros(w1.shape[0], ka, device=dev); w1p[:, :w1.shape[1]] = w1
        f1 = torch.empty(appearance.shape[0], R, device=dev)
        gemm(a, w1p, b1, f1, appearance.shape[0], R, ka, _capi.EPI_BIAS)
        idx = (torch.zeros(M, dtype=torch.int32, device=dev) if appearance.shape[0] == 1
               else torch.arange(M, dtype=torch.int32, device=dev))
        s = spatial.float().contiguous()
        t = torch.empty(M, R, device=dev)
        gemm(s, w2, b2, t, M, R, s.shape[1], _capi.EPI_MUL_RELU, P=f1, p_idx=idx, ldp=R)
        out = torch.empty(M, R, device=dev)
        gemm(t, w3, b3, out, M, R, R, _capi.EPI_BIAS_RELU)
        return out


class MessageMBF(MultiBranchFusion):
    """Parameter container of the reference's MessageMBF (HEAD:476-530).  The message computation itself is fused
    into the engine's graph pass (fc_2 GEMM with fused product, aggregation before fc_3)."""

    def __init__(self, appearance_size: int, spatial_size: int, representation_size: int, node_type: str,
                 cardinality: int) -> None:
        super().__init__(appearance_size, spatial_size, representation_size, cardinality)
        if node_type not in ("human", "object"):
            raise ValueError("Unknown node type \"{}\"".format(node_type))
        self.node_type = node_type

    def forward(self, *args) -> Tensor:
        raise _capi.SkgError("MessageMBF is evaluated inside GraphHead's fused graph pass, not standalone")


class transH_head(Module):
    """Holds the TransH hyper-parameters (HEAD:533-556).  The reference builds a fresh random TransH per image in
    its forward (HEAD:574-580); here the tables are drawn on the host with the identical RNG sequence
    (skghoi_amd/transh.py) and the scores are computed by skg_transh_scores_f32."""

    def __init__(self, transh_dim: int = 200, transh_p_norm: int = 2, transh_norm_flag: bool = True,
                 human_idx: int = 49, num_object: int = 80, num_cls: int = 117) -> None:
        super().__init__()
        self.transh_dim = transh_dim
        self.transh_p_norm = transh_p_norm
        self.transh_norm_flag = transh_norm_flag
        self.device = "cuda"
        self.human_idx = human_idx
        self.num_object = num_object
        self.num_cls = num_cls


class GraphHead(Module):
    """Graphical model head (HEAD:586-993): same constructor, sub-module names and state_dict keys."""

    def __init__(self, out_channels: int, roi_pool_size: int, node_encoding_size: int, representation_size: int,
                 num_cls: int, human_idx: int, object_class_to_target_class: List[list], fg_iou_thresh: float = 0.5,
                 num_iter: int = 2) -> None:
        super().__init__()
        self.out_channels = out_channels
        self.roi_pool_size = roi_pool_size
        self.node_encoding_size = node_encoding_size
        self.representation_size = representation_size
        self.num_cls = num_cls
        self.human_idx = human_idx
        self.object_class_to_target_class = object_class_to_target_class
        self.fg_iou_thresh = fg_iou_thresh
        self.num_iter = num_iter
        if node_encoding_size != 1024 or representation_size != 1024:
            # the reference hard-wires 1024 + 50 into fc_head/fc_tail and adds [.,representation] messages to
            # [.,node_encoding] nodes (HEAD:694-701, 912-914): no other width can run there either
            raise ValueError("node_encoding_size and representation_size must be 1024 (HEAD:694-701)")
        self.box_head = nn.Sequential(
            nn.Flatten(start_dim=1),
            nn.Linear(out_channels * roi_pool_size ** 2, node_encoding_size), nn.ReLU(),
            nn.Linear(node_encoding_size, node_encoding_size), nn.ReLU())
        self.adjacency = nn.Linear(representation_size, 1)
        self.sub_to_obj = MessageMBF(node_encoding_size, 1024, representation_size, node_type="human", cardinality=16)
        self.obj_to_sub = MessageMBF(node_encoding_size, 1024, representation_size, node_type="object", cardinality=16)
        self.norm_h = nn.LayerNorm(node_encoding_size)
        self.norm_o = nn.LayerNorm(node_encoding_size)
        self.spatial_head = nn.Sequential(nn.Linear(46, 128), nn.ReLU(), nn.Linear(128, 256), nn.ReLU(),
                                          nn.Linear(256, 1024), nn.ReLU())
        self.attention_head = MultiBranchFusion(node_encoding_size * 2, 1024, representation_size, cardinality=16)
        self.avg_pool = nn.AdaptiveAvgPool2d(output_size=1)
        self.attention_head_g = MultiBranchFusion(256, 1024, representation_size, cardinality=16)
        self.transh_head = transH_head(transh_dim=50, transh_p_norm=2, transh_norm_flag=True,
                                       human_idx=self.human_idx, num_object=80, num_cls=self.num_cls)
        self.fc_head = nn.Sequential(nn.Linear(1074, 1024), nn.ReLU())
        self.fc_tail = nn.Sequential(nn.Linear(1074, 1024), nn.ReLU())
        self._engine = None

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_engine"] = None                 # run-time cache (streams, packed weights): never copied or pickled
        return state

    # The engine of a stand-alone GraphHead (InteractionHead installs its own, with its predictor/suppressor).
    def _own_engine(self):
        if self._engine is None:
            self._engine = HeadEngine(self, nn.Identity(), nn.Identity(), self.human_idx, self.num_cls, 0.5, 0.2,
                                      _capi.MAX_NODES // 2, _capi.MAX_NODES // 2)
        return self._engine

    def compute_prior_scores(self, x: Tensor, y: Tensor, scores: Tensor, object_class: Tensor) -> Tensor:
        """HEAD:721-767, dense [2, M, K] form (the fused path emits the non-zero cells directly)."""
        dev = scores.device
        vt = self._own_engine().verbs(dev)
        p = 1.0 if self.training else 2.8
        s_h = scores[x].pow(p); s_o = scores[y].pow(p)
        cls = object_class[y].long()
        off = vt.off.long(); flat = vt.flat.long()
        cnt = off[cls + 1] - off[cls]
        pair = torch.repeat_interleave(torch.arange(len(x), device=dev), cnt)
        start = torch.repeat_interleave(off[cls], cnt)
        within = torch.arange(len(pair), device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
        verb = flat[start + within]
        prior = torch.zeros(2, len(x), self.num_cls, device=dev)
        prior[0, pair, verb] = s_h[pair]
        prior[1, pair, verb] = s_o[pair]
        return prior

    def forward(self, features: OrderedDict, image_shapes: List[Tuple[int, int]], box_features: Tensor,
                box_coords: List[Tensor], box_labels: List[Tensor], box_scores: List[Tensor],
                targets: Optional[List[dict]] = None):
        """HEAD:769-993 with the reference's list-of-tensors interface (8 lists in eval, 12 in training)."""
        eng = self._own_engine()
        pre = _pre_from_lists(box_coords, box_labels, box_scores, self.human_idx)
        if self.training:
            assert targets is not None, "Targets should be passed during training"
            from skghoi_amd.train_graph import graph_train
            return graph_train(eng, self, features["3"], image_shapes, box_features, pre, targets)[0]
        g = eng.graph(features["3"], image_shapes, box_features, pre, training=False)
        return _graph_lists(self, eng, pre, g)


def _pre_from_lists(box_coords, box_labels, box_scores, human_idx):
    """Builds the engine's packed detection record from already-preprocessed per-image lists (HEAD:822-841)."""
    from skghoi_amd.engine import Preprocessed
    dev = box_coords[0].device
    if dev.type != "cuda":
        raise _capi.SkgError("the interaction head runs on a HIP device only (got %s)" % dev)
    pre = Preprocessed()
    pre.device = dev
    pre.B = len(box_coords)
    pre.sizes = [int(len(b)) for b in box_coords]
    pre.boxes = torch.cat([b.reshape(-1, 4) for b in box_coords]).float().contiguous()
    pre.labels = torch.cat(box_labels).long().contiguous()
    pre.scores = torch.cat(box_scores).float().contiguous()
    is_h = (pre.labels == human_idx)
    img = torch.repeat_interleave(torch.arange(pre.B, device=dev), torch.tensor(pre.sizes, device=dev))
    n_h = torch.zeros(pre.B, dtype=torch.int64, device=dev).index_add_(0, img, is_h.long()).cpu().numpy()
    pre.n = np.asarray(pre.sizes, dtype=np.int64)
    pre.n_h = n_h.astype(np.int64)
    pre.L = None
    off = 0
    lab_h = pre.labels.cpu()
    for b in range(pre.B):
        if pre.n_h[b] > 0 and pre.n[b] > 1 and not bool(torch.all(lab_h[off:off + int(pre.n_h[b])] == human_idx)):
            raise ValueError("Human detections are not permuted to the top")
        off += int(pre.n[b])
    return pre


def _graph_lists(gh, eng, pre, g):
    lay = g["layout"]
    dev = pre.device
    K = gh.num_cls
    feats, bh, bo, oc, labels, prior = [], [], [], [], [], []
    a = 0
    pf = g.get("pair_features")
    for b in range(lay.n_visit):
        if lay.skipped[b]:
            feats.append(torch.zeros(0, 2 * gh.representation_size, device=dev))
            bh.append(torch.zeros(0, 4, device=dev)); bo.append(torch.zeros(0, 4, device=dev))
            oc.append(torch.zeros(0, device=dev, dtype=torch.int64))
            prior.append(torch.zeros(2, 0, K, device=dev)); labels.append(torch.zeros(0, K, device=dev))
            continue
        m = lay.meta[a]
        p0, P = int(m["pair_off"]), int(m["n_h"]) * (int(m["n"]) - 1)
        xk = g["x_keep"][p0:p0 + P]; yk = g["y_keep"][p0:p0 + P]
        b0 = int(m["box_off"]); n = int(m["n"])
        coords = pre.boxes[b0:b0 + n]; lab = pre.labels[b0:b0 + n]; sc = pre.scores[b0:b0 + n]
        feats.append(pf[p0:p0 + P])
        bh.append(coords[xk]); bo.append(coords[yk]); oc.append(lab[yk])
        prior.append(gh.compute_prior_scores(xk, yk, sc, lab))
        a += 1
    return feats, bh, bo, oc, labels, prior, [], []


_GRAPHS_OK = []


def _graphs_allowed():
    """skghoi_amd.runtime.graphs_allowed(), asked once per process (with a warning when the answer is no)."""
    if not _GRAPHS_OK:
        from skghoi_amd import runtime
        ok = runtime.graphs_allowed()
        if not ok:
            import warnings
            warnings.warn("GPU_MAX_HW_QUEUES < 3: the HIP runtime's graph path has crashed in this configuration; small eval "
                          "batches enqueue their kernels one by one instead of replaying captured graphs")
        _GRAPHS_OK.append(ok)
    return _GRAPHS_OK[0]


class _Prefetch:
    """A training batch being prepared ahead of its forward (InteractionHead.prefetch_train): the steps of
    train_fused.prepare_steps, each resumed on the head's side stream.  advance() runs up to the next host
    synchronisation point (the caller does other host work meanwhile: the kernel whose result that point reads is
    running); finish() runs the rest and returns the Prepared batch."""

    def __init__(self, head, gen, dev, stream, inputs):
        self.head, self.gen, self.dev, self.stream, self.inputs = head, gen, dev, stream, inputs
        self.prep = None
        self.done = False

    def advance(self):
        if self.done:
            return False
        with on_device(self.dev), on_stream(self.stream):
            try:
                next(self.gen)
                return True
            except StopIteration as end:
                self.prep = end.value
                if self.prep.norm is not None:
                    self.prep.norm.get()                 # data parallel: the side stream, not the step's, waits for the n_p all-reduce
                self.prep.ready = torch.cuda.Event()
                self.prep.ready.record(self.stream)
                self.done = True
                return False

    def finish(self):
        while self.advance():
            pass
        return self.prep

    def abandon(self):
        self.gen.close()
        self.done = True


class _Ready:
    """A finished preparation handed back for the next `forward` (same interface as _Prefetch)."""

    def __init__(self, prep):
        self.prep, self.inputs = prep, prep.inputs

    def finish(self):
        return self.prep

    def advance(self):
        return False

    def abandon(self):
        pass


class InteractionHead(Module):
    """Interaction head that constructs and classifies box pairs (HEAD:29-429): same constructor keywords, forward
    signature and result dictionaries.

    Extra keyword (not in the reference, defaults reproduce it):
      precision: str = "fp32" -- dense layers of the inference forward:
        "fp32" (default): the exact fp32 MFMA (v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain) everywhere -- the
          reference's own arithmetic.
        "fp16x2" (opt-in): operands split into two fp16 numbers each (22 significant bits, narrower than fp32's 24),
          three fp16-MFMA passes with fp32 accumulation -- ~1e-6 relative to the exact path on this head, far inside
          the 1e-4 logit bar, at about twice the speed; tiles that leave the fp16 range or contain inf / nan are
          recomputed exactly (skghoi_amd/csrc/skg_gemm.hip).
        "bf16": TRAINING-mode dense layers with bf16 operands (fp32 accumulation, fp32 master weights / gradients /
          activations); inference as "fp16x2".  Training otherwise always uses the exact fp32 path.
      reference_quirks: bool = True -- reproduce (a) the node-offset bug on skipped images and (b) the
        eval-mode label-list zip with skipped images in a batch > 1 (HEAD:298-310: truncated results / IndexError).
        With False every image gets its own (possibly empty) result.
    """

    def __init__(self, box_roi_pool: Module, box_pair_head: Module, box_pair_suppressor: Module,
                 box_pair_predictor: Module, human_idx: int, num_classes: int, box_nms_thresh: float = 0.5,
                 box_score_thresh: float = 0.2, max_human: int = 15, max_object: int = 15,
                 distributed: bool = False, reference_quirks: bool = True, precision: str = "fp32") -> None:
        super().__init__()
        self.box_roi_pool = box_roi_pool
        self.box_pair_head = box_pair_head
        self.box_pair_suppressor = box_pair_suppressor
        self.box_pair_predictor = box_pair_predictor
        self.num_classes = num_classes
        self.human_idx = human_idx
        self.box_nms_thresh = box_nms_thresh
        self.box_score_thresh = box_score_thresh
        self.max_human = max_human
        self.max_object = max_object
        self.distributed = distributed
        self.reference_quirks = reference_quirks
        if precision not in ("fp32", "bf16", "fp16x2"):
            raise ValueError("precision must be 'fp32', 'fp16x2' or 'bf16'")
        self.precision = precision
        self.fused_training = True      # False: training through autograd over per-layer Functions (skghoi_amd/train_graph.py)
        self.grad_mode = "autograd"     # "direct": the fused step writes p.grad itself (skghoi_amd/train_fused.py, StepFn)
        self._engine = None

    # run-time caches (engine with its streams / captured plans, parameter arena bookkeeping, prefetch state): never part of
    # a copy or a pickle of the module -- they are rebuilt on first use
    _RUNTIME = ("_engine", "_stacked", "_pf_stream", "_prefetched", "_prefetched2", "_last_train", "_last_train_plan",
                "_train_params")

    def __getstate__(self):
        state = dict(self.__dict__)
        for k in self._RUNTIME:
            if k in state:
                state[k] = None
        return state

    def engine(self) -> HeadEngine:
        e = self._engine
        if e is None or e.gh is not self.box_pair_head or e.predictor is not self.box_pair_predictor \
                or e.suppressor is not self.box_pair_suppressor \
                or e.max_human != self.max_human or e.max_object != self.max_object \
                or e.box_nms_thresh != float(self.box_nms_thresh) or e.box_score_thresh != float(self.box_score_thresh):
            old = e
            e = HeadEngine(self.box_pair_head, self.box_pair_predictor, self.box_pair_suppressor, self.human_idx,
                           self.num_classes, self.box_nms_thresh, self.box_score_thresh, self.max_human,
                           self.max_object, faithful_skip_offset=self.reference_quirks)
            if old is not None:
                # the engine being replaced owns captured launch plans (hipGraphs): they go through the one teardown path
                # -- idle device, before anything of a new capture exists (skghoi_amd/small.py) -- not by refcount
                e.small_batch_max, e.small_batch_buckets = old.small_batch_max, old.small_batch_buckets
                if old._small is not None:
                    old._small.close()
                    old._small = None
            self._engine = e
        e.faithful_skip_offset = self.reference_quirks
        e.precision = self.precision
        return e

    # ------------------------------------------------------------------------------------------ HEAD:92-151
    def preprocess(self, detections: List[dict], targets: List[dict], append_gt: Optional[bool] = None) -> List[dict]:
        if append_gt is None:
            append_gt = self.training
        pre = self.engine().preprocess(detections, targets, append_gt, self.training)
        return [dict(boxes=b.view(-1, 4), labels=l.view(-1), scores=s.view(-1)) for b, l, s in
                zip(pre.boxes.split(pre.sizes), pre.labels.split(pre.sizes), pre.scores.split(pre.sizes))]

    # ------------------------------------------------------------------------------------------ HEAD:237-337
    def postprocess(self, logits_p: Tensor, logits_s: Tensor, prior: List[Tensor], boxes_h: List[Tensor],
                    boxes_o: List[Tensor], object_class: List[Tensor], labels: List[Tensor]) -> List[dict]:
        """List-based form kept for API compatibility (the fused forward scores on the device in one kernel)."""
        num_boxes = [len(b) for b in boxes_h]
        weights = torch.sigmoid(logits_s).squeeze(1).split(num_boxes)
        scores = torch.sigmoid(logits_p).split(num_boxes)
        if len(labels) == 0:
            labels = [None for _ in range(len(num_boxes))]
        results = []
        for w, s, p, b_h, b_o, o, l in zip(weights, scores, prior, boxes_h, boxes_o, object_class, labels):
            x, y = torch.nonzero(p[0]).unbind(1)
            r = dict(boxes_h=b_h, boxes_o=b_o, index=x, prediction=y,
                     scores=s[x, y] * p[:, x, y].prod(dim=0) * w[x].detach(), object=o, prior=p[:, x, y], weights=w)
            if l is not None:
                r["labels"] = l[x, y]
                r["unary_labels"] = l.sum(dim=1).clamp(max=1)
            results.append(r)
        return results

    # ------------------------------------------------------------------------------------------ HEAD:341-429
    def forward(self, features: OrderedDict, detections: List[dict], image_shapes: List[Tuple[int, int]],
                targets: Optional[List[dict]] = None) -> List[dict]:
        if self.training:
            assert targets is not None, "Targets should be passed during training"
        dev = features["3"].device
        if dev.type != "cuda":
            raise _capi.SkgError("the interaction head runs on a HIP device only (features on %s)" % dev)
        for det in detections:
            if det["boxes"].device != dev:
                raise _capi.SkgError("detections on %s but features on %s" % (det["boxes"].device, dev))
        with on_device(dev):                   # kernels are enqueued on the current stream OF THE INPUTS' device
            return self._forward(features, detections, image_shapes, targets)

    def _forward(self, features, detections, image_shapes, targets):
        if self.training or targets is not None:
            self._drop_eval_look_ahead()
        if self.training:
            assert targets is not None, "Targets should be passed during training"
            return self._forward_train(features, detections, image_shapes, targets)
        if targets is not None:
            # eval mode with targets (validation): the reference associates labels and consumes the sampling RNG
            # whenever targets are given (HEAD:933-963); served by the training-mode graph pass without autograd,
            # without GT boxes appended (HEAD:105-106) and with the eval score power (HEAD:742)
            with torch.no_grad():
                if self.fused_training:
                    # the native preparation + the native launch plan's forward, in exact fp32 (round 5: validation batches no
                    # longer walk the round-1 autograd graph)
                    from skghoi_amd import train_fused
                    if train_fused.supported(self):
                        out = train_fused.validate_forward(self, self.engine(), features, detections, image_shapes, targets,
                                                           prep=self._take_prefetched(detections, image_shapes, targets))
                        if out is not None:
                            return out
                return self._forward_train(features, detections, image_shapes, targets, with_losses=False)
        eng = self.engine()
        if eng.small_batch_max and len(detections) <= eng.small_batch_max and _graphs_allowed():
            # a few images (the reference evaluates ONE per forward, utils.py:166-167): replay the captured launch plan
            # of this batch shape instead of ~40 individually enqueued kernels (skghoi_amd/small.py)
            from skghoi_amd.small import SmallBatchRunner
            if eng._small is None:
                eng._small = SmallBatchRunner(eng)
            if eng._small.eligible(self, detections, targets):
                return eng._small.forward(self, features, detections, image_shapes)
        self._drop_eval_look_ahead()
        pre = eng.preprocess(detections, targets, False, False, check_weights=True)
        return self._forward_eager(pre, features, image_shapes)

    def _forward_eager(self, pre, features, image_shapes):
        """Eval forward from packed detections on: every kernel enqueued individually, chunked over the batch."""
        eng = self.engine()
        box_coords = list(pre.boxes.split(pre.sizes))
        box_features = self.box_roi_pool(features, box_coords, image_shapes)
        g = eng.graph(features["3"], image_shapes, box_features, pre, training=False)
        lay = g["layout"]
        if lay.n_visit == 0:
            raise RuntimeError("torch.cat(): expected a non-empty list of Tensors")     # HEAD:408 on an empty batch
        dev = pre.device
        if lay.n_active:
            logits = eng.classify(g["pair_features"], checked=True)
            r = eng.score(logits, pre, g, False)
            eng.last = dict(g, logits=logits)
        else:
            r = None
            eng.last = dict(g)
        return self._results(lay, r, dev)

    # ------------------------------------------------------------------------------------------ prefetch
    def prefetch_eval(self, detections: List[dict], after=None) -> bool:
        """The reference's test loop (utils.py:157-167) calls the network once per image; a loop that knows its NEXT image
        can hand that image's detections over as soon as the current forward is enqueued: detection selection (HEAD:92-151),
        the read-back of its counts and the image's TransH table draw (HEAD:574-580) then run on the side stream beside
        the forward in flight, and the next `forward(features, detections, image_shapes)` called with this same list starts
        at its launch plan.  Single images in eval mode only (anything else returns False and does nothing); results and
        the position of the global CPU generator are those of the loop without look-ahead, also when the next call turns out
        to be for other detections (skghoi_amd/small.py, SmallBatchRunner.look_ahead; `after`: see there).
        `trainer.test` drives it."""
        if self.training or not detections or len(detections) != 1:
            return False
        dev = detections[0]["boxes"].device
        if dev.type != "cuda":
            return False                       # (the forward reports it, from where a loop without look-ahead sees it)
        eng = self.engine()
        if not (eng.small_batch_max and _graphs_allowed()):
            return False
        from skghoi_amd.small import SmallBatchRunner
        if eng._small is None:
            eng._small = SmallBatchRunner(eng)
        with on_device(dev):
            return eng._small.look_ahead(self, detections, after=after)

    def _drop_eval_look_ahead(self):
        small = getattr(self.__dict__.get("_engine"), "_small", None)
        if small is not None:
            small.drop_look_ahead()

    def prefetch_train(self, detections: List[dict], image_shapes: List[Tuple[int, int]], targets: List[dict],
                       after=None, arena=False, deep=False) -> bool:
        """Prepares the NEXT training batch while the GPU is busy with the current step: detection selection (HEAD:92-151),
        pairs + spatial encoding, label association and the host RNG draws of the forward (TransH tables, negative
        permutations) run now, on a high-priority side stream; the next `forward(features, detections, image_shapes,
        targets)` called with these same objects picks the result up instead of paying the forward's two host
        synchronisations with an idle GPU.

        Ordering against whoever produced the inputs: the side stream first waits for `after` -- an event recorded behind
        the inputs' producer (a loader's non-blocking uploads); `trainer.train_step` passes one recorded at its entry, i.e.
        behind everything enqueued before the step and in front of the step's own kernels.  Without it the side stream waits
        for everything the caller's current stream holds right now (always safe; the preparation then starts behind the step
        in flight).  The inputs are marked as in use by the side stream (record_stream), so their memory is not recycled
        under the preparation.  The host RNG is consumed here, in the order the forward would: call it once per batch, in
        batch order, with no other consumer of the global generator in between.  Returns False (nothing done) when the head
        is not in the fused training configuration.
        arena=True (trainer.train_step only): the batch's device tensors are carved out of a reusable block
        (train_fused.PrepArena, three slots in rotation) instead of one allocation each; the caller MUST call
        train_fused.release_prepared(head) on the step's stream once the step that consumed the batch has enqueued its last
        kernel -- the slot is reused behind that point.
        deep=True (trainer.train_step's two-batch look-ahead): the batch AFTER the next one.  Its handle waits in a second
        place (`promote_prefetched` moves it up one step later) and nothing of it touches the host RNG until it has been
        promoted and finished -- the draws stay in batch order."""
        from skghoi_amd import train_fused
        if not (self.fused_training and train_fused.supported(self)) or not detections or targets is None:
            return False
        dev = detections[0]["boxes"].device
        if dev.type != "cuda":
            raise _capi.SkgError("the interaction head runs on a HIP device only (detections on %s)" % dev)
        eng = self.engine()
        side = self._prefetch_stream(dev)
        if after is None:
            after = torch.cuda.Event()
            after.record(current_stream_of(dev))
        side.wait_event(after)
        for group in (detections, targets):
            for d in group:
                for t in d.values():
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(side)
        slot = None
        if arena:
            ring = self.__dict__.get("_prep_slots")
            if ring is None or ring[0].dev != dev:
                ring = self.__dict__["_prep_slots"] = [train_fused.PrepArena(dev) for _ in range(3)]
                self.__dict__["_prep_slot_next"] = 0
            k = self.__dict__["_prep_slot_next"]
            self.__dict__["_prep_slot_next"] = (k + 1) % len(ring)
            slot = ring[k]
            with on_device(dev), on_stream(side):
                slot.begin(side, current_stream_of(dev))
        # (eval mode: the preparation of a VALIDATION batch -- no GT boxes appended, eval score power)
        h = _Prefetch(self, train_fused.prepare_steps(self, eng, detections, image_shapes, targets, arena=slot,
                                                      training=self.training), dev,
                      side, (detections, image_shapes, targets))
        h.advance()                            # launches the detection-selection kernel; its counts are read later
        if deep:
            old = self.__dict__.get("_prefetched2")
            if old is not None:
                old.abandon()
            self._prefetched2 = h
        else:
            self._prefetched = h
        return h

    def promote_prefetched(self, detections, targets):
        """The preparation started one step ago with prefetch_train(deep=True) becomes the NEXT batch's, if it was made from
        exactly these objects; returns its handle (advance() / finish()), else None (a stale one is abandoned: it has not
        drawn from the host RNG yet)."""
        h = self.__dict__.get("_prefetched2")
        self._prefetched2 = None
        if h is None:
            return None
        d, _s, t = h.inputs
        if d is detections and t is targets and getattr(self, "_prefetched", None) is None:
            self._prefetched = h
            return h
        h.abandon()
        return None

    def fused_step(self, features, detections, image_shapes, targets, after_forward=None, defer_backward=False,
                   adamw=None):
        """Forward + backward of one training step without the autograd engine (skghoi_amd.train_fused.fused_step) for a
        trainer that owns the loop: gradients of the summed losses land in `p.grad` (overwritten, not accumulated).
        Returns the result list with the (detached) loss dict appended, or None when this call has to go through
        `forward` + `backward()` instead.
        defer_backward=True: the backward's launches may still be on their way to the stream when this returns (the
        library's worker thread issues them); the caller runs `skghoi_amd.train_fused.join_backward()` before it
        enqueues anything behind the gradients (trainer.train_step does, in front of the optimizer).
        adamw (with defer_backward): trainer.SkgAdamW.backward_slices -- when it yields slices for this step, the worker
        thread also issues the optimizer's update chunk by chunk inside the backward; the slices then sit in
        self._adamw_in_backward for the caller to acknowledge (SkgAdamW.backward_done) instead of calling step()."""
        from skghoi_amd import train_fused
        self.__dict__.pop("_adamw_in_backward", None)
        if not (self.training and self.fused_training and train_fused.supported(self)) or targets is None:
            return None
        dev = features["3"].device
        if dev.type != "cuda":
            return None
        with on_device(dev):
            out, prep = train_fused.fused_step(self, self.engine(), features, detections, image_shapes, targets,
                                               prep=self._take_prefetched(detections, image_shapes, targets),
                                               after_forward=after_forward, defer_backward=defer_backward, adamw=adamw)
            if out is None and prep is not None:
                self._prefetched = _Ready(prep)       # prepared but not consumed here: the `forward` that follows takes it
        return out

    def _prefetch_stream(self, dev):
        st = getattr(self, "_pf_stream", None)
        if st is None or st.device != dev:
            from skghoi_amd import engine as _engine
            st = self._pf_stream = _engine.shared_side_stream(dev, priority=-1)     # one per process (see there)
        return st

    def _take_prefetched(self, detections, image_shapes, targets):
        """The Prepared batch of prefetch_train if it was made from exactly these objects (identity), else None."""
        prep = getattr(self, "_prefetched", None)
        self._prefetched = None
        if prep is None:
            return None
        d, s, t = prep.inputs
        if not (d is detections and t is targets):
            prep.abandon()
            return None
        prep = prep.finish()
        d, s, t = prep.inputs
        if d is detections and t is targets and (s is image_shapes or list(s) == list(image_shapes)):
            if prep.slot is not None:
                self.__dict__["_prep_slot_in_use"] = prep.slot      # (released by the trainer: train_fused.release_prepared)
            return prep
        return None

    # ------------------------------------------------------------------------------------------ training (HEAD:380-429)
    def _forward_train(self, features, detections, image_shapes, targets, with_losses=True):
        from skghoi_amd import autograd as _ag
        from skghoi_amd.train_graph import graph_train
        linear = _ag.linear_bf16 if self.precision == "bf16" else _ag.linear
        eng = self.engine()
        pre = None
        if with_losses and self.training and self.fused_training:
            # the training step proper: one autograd.Function around the hand-written forward / backward kernels
            from skghoi_amd import train_fused
            if train_fused.supported(self):
                out, prep = train_fused.train_forward(self, eng, features, detections, image_shapes, targets,
                                                      prep=self._take_prefetched(detections, image_shapes, targets))
                if out is not None:
                    return out
                pre = prep.pre                 # no image with pairs: the generic path continues from the selection
        if pre is None:
            pre = eng.preprocess(detections, targets, self.training, self.training)
        box_coords = list(pre.boxes.split(pre.sizes))
        box_features = self.box_roi_pool(features, box_coords, image_shapes)
        from skghoi_amd import dist as _skd
        norm = []
        # the three n_p normalisers (HEAD:162-172, 190-199, 219-228) leave as ONE 3-element all-reduce as soon as the
        # labels exist, and come back as a device tensor when the loss scalars are formed: no barrier, no .item()
        ex = getattr(self, "grad_exchange", None)       # (its communicator and group: the SAME collective as peers on the fused route)
        on_counts = (lambda c: norm.append(_skd.start_normalisers(
            c, self.distributed, group=getattr(ex, "group", None), force=getattr(self, "force_collectives", False),
            native=getattr(ex, "native", None)))) if with_losses else None
        (feats, bh, bo, oc, labels, prior, pos, neg, he, te, re, rn), lay, P = graph_train(
            eng, self.box_pair_head, features["3"], image_shapes, box_features, pre, targets, on_counts=on_counts)
        if len(feats) == 0:
            raise RuntimeError("torch.cat(): expected a non-empty list of Tensors")
        pf = P["PF"] if P is not None else torch.cat(feats)       # skipped images contribute zero rows
        if isinstance(self.box_pair_predictor, nn.Linear) and isinstance(self.box_pair_suppressor, nn.Linear):
            logits_p = linear(pf, self.box_pair_predictor.weight, self.box_pair_predictor.bias)
            logits_s = linear(pf, self.box_pair_suppressor.weight, self.box_pair_suppressor.bias)
        else:
            logits_p = self.box_pair_predictor(pf); logits_s = self.box_pair_suppressor(pf)
        results = self._postprocess_packed(logits_p, logits_s, P, lay, pf.device)
        if not with_losses:
            return results
        n_p = norm[0].get()
        results.append(dict(
            hoi_loss=self.compute_interaction_classification_loss(results, n_p=n_p[0]),
            interactiveness_loss=self.compute_interactiveness_loss(results, n_p=n_p[1]),
            transH_loss=self.compute_transH_loss(pos, neg, he, re, rn, te, results, n_p=n_p[2])))
        return results

    def _postprocess_packed(self, logits_p, logits_s, P, lay, dev):
        """postprocess (HEAD:237-337) for the whole training batch at once; per-image dicts are views."""
        K = self.num_classes
        results = []
        if P is not None:
            A = len(P["ppi"])
            weights = torch.sigmoid(logits_s).squeeze(1)
            scores = torch.sigmoid(logits_p)
            prior, lab = P["prior"], P["labels"]
            x, y = torch.nonzero(prior[0]).unbind(1)
            action = scores[x, y] * prior[:, x, y].prod(dim=0) * weights[x].detach()
            pxy = prior[:, x, y]
            lxy = lab[x, y]
            unary = lab.sum(dim=1).clamp(max=1)
            cpi = torch.bincount(P["pair_img"][x], minlength=A).tolist()
            pair_off = torch.from_numpy(lay.meta["pair_off"].astype("int64")).to(dev)
            xl = x - pair_off[P["pair_img"][x]]
            ppi = P["ppi"]
            w_s, u_s = weights.split(ppi), unary.split(ppi)
            bh_s, bo_s, ob_s = P["boxes_h"].split(ppi), P["boxes_o"].split(ppi), P["object"].split(ppi)
            x_s, y_s, a_s, l_s = xl.split(cpi), y.split(cpi), action.split(cpi), lxy.split(cpi)
            p_s = pxy.split(cpi, dim=1)
        a = 0
        for b in range(lay.n_visit):
            if lay.skipped[b]:
                results.append(self._empty_result(dev, with_labels=True))
                continue
            results.append(dict(boxes_h=bh_s[a], boxes_o=bo_s[a], index=x_s[a], prediction=y_s[a], scores=a_s[a],
                                object=ob_s[a], prior=p_s[a], weights=w_s[a], labels=l_s[a], unary_labels=u_s[a]))
            a += 1
        return results

    def _n_p(self, labels):
        """HEAD:162-172 for a caller that invokes one loss method on its own: number of non-zero labels, averaged over
        the ranks (the barrier is dropped: all_reduce is already a synchronising collective).  The forward pass does
        not come through here -- it fuses the three normalisers into one collective (skghoi_amd/dist.py)."""
        n_p = len(torch.nonzero(labels))
        if self.distributed:
            world_size = dist.get_world_size()
            t = torch.as_tensor([n_p], device="cuda", dtype=torch.float64)
            dist.all_reduce(t)
            n_p = (t / world_size).item()
        return n_p

    def compute_interaction_classification_loss(self, results: List[dict], n_p=None) -> Tensor:
        """HEAD:153-177.  n_p: pre-reduced normaliser (0-d device tensor) from the forward's fused all-reduce."""
        from skghoi_amd.ops import binary_focal_loss
        labels = torch.cat([r["labels"] for r in results]); scores = torch.cat([r["scores"] for r in results])
        if n_p is None:
            n_p = self._n_p(labels)
        return binary_focal_loss(scores, labels, reduction="sum", gamma=0.2) / n_p

    def compute_interactiveness_loss(self, results: List[dict], n_p=None) -> Tensor:
        """HEAD:180-205."""
        from skghoi_amd.ops import binary_focal_loss
        weights = torch.cat([r["weights"] for r in results]); labels = torch.cat([r["unary_labels"] for r in results])
        if n_p is None:
            n_p = self._n_p(labels)
        return binary_focal_loss(weights, labels, reduction="sum", gamma=2.0) / n_p

    def compute_transH_loss(self, positive_scores, negative_scores, head, relation, relation_norm, tail,
                            results: List[dict], n_p=None) -> Tensor:
        """HEAD:207-235 with the semantics the code intends (the committed call passes six arguments to a
        one-argument NegativeSampling.forward and raises TypeError): NegativeSampling splits
        score = cat[pos, neg] in halves shaped [M, 1] (heads/NegativeSampling.py:30-40, 52-56) and applies
        MarginLoss(margin=1): mean(max(p - n, -margin)) + margin (heads/MarginLoss.py:28-36); regul_rate = 0."""
        if n_p is None:
            n_p = self._n_p(torch.cat([r["unary_labels"] for r in results]))
        score = torch.cat([torch.cat(positive_scores), torch.cat(negative_scores)])
        half = len(score) // 2
        p = score[:half].view(-1, half).permute(1, 0); n = score[half:].view(-1, half).permute(1, 0)
        margin = 1.0
        loss = torch.max(p - n, torch.tensor([-margin], device=score.device)).mean() + margin
        return loss / n_p

    def _empty_result(self, dev, with_labels):
        d = dict(boxes_h=torch.zeros(0, 4, device=dev), boxes_o=torch.zeros(0, 4, device=dev),
                 index=torch.zeros(0, dtype=torch.int64, device=dev),
                 prediction=torch.zeros(0, dtype=torch.int64, device=dev), scores=torch.zeros(0, device=dev),
                 object=torch.zeros(0, dtype=torch.int64, device=dev), prior=torch.zeros(2, 0, device=dev),
                 weights=torch.zeros(0, device=dev))
        if with_labels:
            d["labels"] = torch.zeros(0, device=dev)
            d["unary_labels"] = torch.zeros(0, device=dev)
        return d

    def _results(self, lay, r, dev, train_extras=None):
        """Per-image result dicts (HEAD:317-322) as views of the packed arrays.  train_extras = (labels at the scored
        cells [L], unary labels [sumP]): training-mode results carry `labels` / `unary_labels` (HEAD:323-327)."""
        n_skipped = int(lay.skipped[:lay.n_visit].sum())
        # (training appends a label entry for every image, HEAD:838 / 934: nothing is truncated there)
        quirk = self.reference_quirks and n_skipped > 0 and train_extras is None
        # HEAD:298-310: in eval only skipped images append to the label list; postprocess then zips over it
        n_out = min(lay.n_visit, n_skipped) if quirk else lay.n_visit
        if quirk:
            for b in range(n_out):
                if not lay.skipped[b]:
                    raise IndexError("index is out of bounds for dimension with size 0")      # HEAD:327
        results = []
        if r is not None:
            ppi = [int(v) for v in lay.pairs_per_image]; cpi = [int(v) for v in lay.cells_per_image]
            Mp, Lt = lay.sum_p, lay.sum_l
            bh = r["boxes_h"][:Mp].split(ppi); bo = r["boxes_o"][:Mp].split(ppi)
            ob = r["object"][:Mp].split(ppi); wt = r["weights"][:Mp].split(ppi)
            ix = r["index"][:Lt].split(cpi); pr = r["prediction"][:Lt].split(cpi); sc = r["scores"][:Lt].split(cpi)
            pri = r["prior"][:, :Lt].split(cpi, dim=1)
            if train_extras is not None:
                lb = train_extras[0][:Lt].split(cpi); un = train_extras[1][:Mp].split(ppi)
        a = 0
        for b in range(n_out):
            if lay.skipped[b]:
                results.append(self._empty_result(dev, with_labels=quirk or train_extras is not None))
                continue
            d = dict(boxes_h=bh[a], boxes_o=bo[a], index=ix[a], prediction=pr[a], scores=sc[a],
                     object=ob[a], prior=pri[a], weights=wt[a])
            if train_extras is not None:
                d["labels"] = lb[a]; d["unary_labels"] = un[a]
            results.append(d)
            a += 1
        return results
