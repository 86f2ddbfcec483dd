"""The drop-in proof through the reference's own door.

models/adamixer_transH_spatial_r50_models.py does
    sys.path.append('<...>/heads')                                                    (models:24)
    from adamixer_transH_spatial_r50_head import InteractionHead, GraphHead           (models:25)
builds both with the keyword sets of models:164-191 and calls
    results = self.interaction_head(box_feature, detections, images.image_sizes, targets)   (models:103-104)
from GenericHOINetwork.forward.  These tests do exactly that against <repo>/heads:
  * CPU (fresh interpreter, so that neither the package nor the reference's module of the same name is loaded yet):
    the import line, both constructors with the reference's keywords, the 408-key state_dict a reference checkpoint
    carries, the optimizer grouping by parameter-name prefix (main:112-120)
  * gpu: a GenericHOINetwork-shaped wrapper (features from a stand-in backbone/neck, MultiScaleRoIAlign as
    models:158-162 builds it) in eval and in training, checked against the CPU oracle fed with the oracle's RoIAlign.
"""
import json
import os
import subprocess
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import json, sys
sys.dont_write_bytecode = True
sys.path.append(%(heads)r)                                                   # models:24
from adamixer_transH_spatial_r50_head import InteractionHead, GraphHead      # models:25
import adamixer_transH_spatial_r50_head as mod
import torch
from torch import nn
sys.path.insert(0, %(root)r)
from skghoi_amd import synth

# models:137-148 defaults of SpatiallyConditionedGraph / the HICO-DET main (main:73-76)
object_to_action = synth.hico_object_to_verb()
human_idx, num_classes = 49, 117
output_size, sampling_ratio = 7, 2
node_encoding_size = representation_size = 1024
fg_iou_thresh, num_iterations = 0.5, 2
box_nms_thresh, box_score_thresh, max_human, max_object, distributed = 0.5, 0.2, 15, 15, False
out_channels = 256

box_roi_pool = nn.Identity()                    # MultiScaleRoIAlign is injected by the caller (models:158-162)
box_pair_head = GraphHead(                      # models:164-174, keyword for keyword
    out_channels=out_channels,
    roi_pool_size=output_size,
    node_encoding_size=node_encoding_size,
    representation_size=representation_size,
    num_cls=num_classes,
    human_idx=human_idx,
    object_class_to_target_class=object_to_action,
    fg_iou_thresh=fg_iou_thresh,
    num_iter=num_iterations
)
box_pair_predictor = nn.Linear(representation_size * 2, num_classes)     # models:176
box_pair_suppressor = nn.Linear(representation_size * 2, 1)              # models:177
interaction_head = InteractionHead(             # models:179-191, keyword for keyword
    box_roi_pool=box_roi_pool,
    box_pair_head=box_pair_head,
    box_pair_suppressor=box_pair_suppressor,
    box_pair_predictor=box_pair_predictor,
    num_classes=num_classes,
    human_idx=human_idx,
    box_nms_thresh=box_nms_thresh,
    box_score_thresh=box_score_thresh,
    max_human=max_human,
    max_object=max_object,
    distributed=distributed
)


class Net(nn.Module):                            # the attribute name the optimizer groups key on (main:112-120)
    def __init__(self):
        super().__init__()
        self.detector_backbone = nn.Linear(2, 2)
        self.interaction_head = interaction_head


net = Net()
sd = synth.make_state_dict(num_classes, out_channels, output_size, seed=0)   # the reference's 408 keys and shapes
missing, unexpected = interaction_head.load_state_dict(sd, strict=True)
ckpt = {"model_state_dict": {"interaction_head." + k: v for k, v in sd.items()}}
ckpt["model_state_dict"].update({"detector_backbone." + k: v for k, v in net.detector_backbone.state_dict().items()})
net.load_state_dict(ckpt["model_state_dict"])                                # main:89 / test:55
head_params = [n for n, p in net.named_parameters() if "interaction_head" in n and p.requires_grad]
print(json.dumps(dict(file=mod.__file__, n_keys=len(interaction_head.state_dict()), n_head_params=len(head_params),
                      missing=list(missing), unexpected=list(unexpected),
                      attrs=[n for n, _ in interaction_head.named_children()],
                      gh_attrs=[n for n, _ in box_pair_head.named_children()],
                      defaults=[interaction_head.max_human, interaction_head.max_object, box_pair_head.num_iter])))
'''


def test_reference_import_line_and_constructors():
    code = _CHILD % dict(heads=os.path.join(ROOT, "heads"), root=ROOT)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)                                   # only the sys.path.append of models:24 finds the module
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp", timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert os.path.samefile(r["file"], os.path.join(ROOT, "heads", "adamixer_transH_spatial_r50_head.py"))
    assert r["n_keys"] == 408 == r["n_head_params"] and not r["missing"] and not r["unexpected"]
    assert r["attrs"] == ["box_roi_pool", "box_pair_head", "box_pair_suppressor", "box_pair_predictor"]
    assert r["gh_attrs"] == ["box_head", "adjacency", "sub_to_obj", "obj_to_sub", "norm_h", "norm_o", "spatial_head",
                             "attention_head", "avg_pool", "attention_head_g", "transh_head", "fc_head", "fc_tail"]
    assert r["defaults"] == [15, 15, 2]


@pytest.mark.reference
def test_child_attribute_order_matches_reference_module():
    """The attribute (hence state_dict / checkpoint) order asserted above is the reference's own."""
    from oracle import ref_import
    from skghoi_amd import synth
    ref = ref_import.build_reference_head(117, 49, synth.hico_object_to_verb(), 256, 7, 15, 15)
    assert [n for n, _ in ref.named_children()] == ["box_roi_pool", "box_pair_head", "box_pair_suppressor",
                                                    "box_pair_predictor"]
    assert [n for n, _ in ref.box_pair_head.named_children()] == [
        "box_head", "adjacency", "sub_to_obj", "obj_to_sub", "norm_h", "norm_o", "spatial_head", "attention_head",
        "avg_pool", "attention_head_g", "transh_head", "fc_head", "fc_tail"]


# ------------------------------------------------------------------------------------------------------ GPU: the caller
class _ImageList:
    def __init__(self, tensors, image_sizes):
        self.tensors, self.image_sizes = tensors, image_sizes


class _Backbone(nn.Module):
    """Stand-in for detector_backbone + detector_neck (models:94-95): four 256-channel maps at strides 4..32."""

    def __init__(self, C=256):
        super().__init__()
        self.proj = nn.ModuleList([nn.Conv2d(3, C, 1) for _ in range(4)])

    def forward(self, x):
        return [p(nn.functional.avg_pool2d(x, s)) for p, s in zip(self.proj, (4, 8, 16, 32))]


class GenericHOINetworkLike(nn.Module):
    """models/adamixer_transH_spatial_r50_models.py:27-110 without mmdet and the image transform: same attribute names,
    same call into the head."""

    def __init__(self, detector_backbone, detector_neck, interaction_head):
        super().__init__()
        self.detector_backbone = detector_backbone
        self.detector_neck = detector_neck
        self.interaction_head = interaction_head

    def forward(self, images, detections, targets=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        features = self.detector_backbone(images.tensors)
        features = self.detector_neck(features)
        box_feature = OrderedDict()
        box_feature['0'] = features[0]
        box_feature['1'] = features[1]
        box_feature['2'] = features[2]
        box_feature['3'] = features[3]
        return self.interaction_head(box_feature, detections, images.image_sizes, targets)


def _build(training):
    sys.path.append(os.path.join(ROOT, "heads"))
    import importlib
    mod = importlib.import_module("adamixer_transH_spatial_r50_head")
    if not os.path.samefile(mod.__file__, os.path.join(ROOT, "heads", "adamixer_transH_spatial_r50_head.py")):
        pytest.skip("the reference's module of the same name is loaded in this process (build container)")
    from skghoi_amd import synth
    from skghoi_amd.roi_pool import MultiScaleRoIAlign
    o2v = synth.hico_object_to_verb()
    pool = MultiScaleRoIAlign(featmap_names=['0', '1', '2', '3'], output_size=7, sampling_ratio=2)   # models:158-162
    gh = mod.GraphHead(out_channels=256, roi_pool_size=7, node_encoding_size=1024, representation_size=1024,
                       num_cls=117, human_idx=49, object_class_to_target_class=o2v, fg_iou_thresh=0.5, num_iter=2)
    head = mod.InteractionHead(box_roi_pool=pool, box_pair_head=gh, box_pair_suppressor=nn.Linear(2048, 1),
                               box_pair_predictor=nn.Linear(2048, 117), num_classes=117, human_idx=49,
                               box_nms_thresh=0.5, box_score_thresh=0.2, max_human=15, max_object=15,
                               distributed=False)
    sd = synth.make_state_dict(117, 256, 7, seed=0)
    head.load_state_dict(sd)
    torch.manual_seed(11)
    net = GenericHOINetworkLike(_Backbone(), nn.Identity(), head).cuda().train(training)
    imgs = [synth.make_image(2000 + i, n_h=nh, n_o=no) for i, (nh, no) in enumerate([(3, 4), (2, 5)])]
    det = [dict(boxes=i["boxes"], labels=i["labels"], scores=i["scores"]) for i in imgs]
    shapes = [(800, 1200), (800, 1200)]
    g = torch.Generator().manual_seed(5)
    images = _ImageList(torch.randn(2, 3, 800, 1216, generator=g).cuda(), shapes)
    targets = [synth.make_targets(d, 49, o2v, 950 + k, n_gt=3) for k, d in enumerate(det)]
    return net, sd, o2v, images, det, targets, shapes


def _cuda(x):
    if torch.is_tensor(x):
        return x.cuda()
    if isinstance(x, dict):
        return {k: _cuda(v) for k, v in x.items()}
    return [_cuda(v) for v in x]


@pytest.mark.gpu
def test_generic_hoi_network_shaped_caller_eval():
    from oracle import roi_align_oracle as RO
    from oracle import skg_oracle as O
    net, sd, o2v, images, det, targets, shapes = _build(training=False)
    with torch.no_grad():
        torch.manual_seed(77)
        results = net(images, _cuda(det))                                            # models:103-104, targets=None
        feats = [f.cpu() for f in net.detector_neck(net.detector_backbone(images.tensors))]
    assert len(results) == 2
    for r in results:
        assert set(r) == {"boxes_h", "boxes_o", "index", "prediction", "scores", "object", "prior", "weights"}   # HEAD:317-322
    torch.manual_seed(77)
    with torch.no_grad():
        want, extras = O.interaction_head_forward(
            sd, feats[3], det, shapes, lambda coords: RO.multiscale_roi_align(feats, coords, shapes, 7, 2),
            117, 49, o2v, max_human=15, max_object=15)
    last = net.interaction_head.engine().last
    assert (last["logits"][:, :117].cpu() - extras["logits_p"]).abs().max().item() <= 1e-4
    for r, w in zip(results, want):
        for k in ("index", "prediction", "object"):
            assert torch.equal(r[k].cpu(), w[k]), k
        assert (r["scores"].cpu() - w["scores"]).abs().max().item() <= 1e-5
        assert torch.equal(r["boxes_h"].cpu(), w["boxes_h"])


@pytest.mark.gpu
def test_generic_hoi_network_shaped_caller_train():
    from skghoi_amd import trainer
    net, sd, o2v, images, det, targets, shapes = _build(training=True)
    with pytest.raises(ValueError):
        net(images, _cuda(det))                                                      # models:86-87
    opt = trainer.build_optimizer(net, lr=1e-4)                                      # main:109-127 grouping by name
    assert len(opt.param_groups) == 2 and len(opt.param_groups[0]["params"]) == 408
    torch.manual_seed(78)
    out = net(images, _cuda(det), _cuda(targets))
    loss_dict = out.pop()                                                            # utils.py:216
    assert set(loss_dict) == {"hoi_loss", "interactiveness_loss", "transH_loss"} and len(out) == 2
    for r in out:
        assert {"labels", "unary_labels"} <= set(r)
    total = sum(loss for loss in loss_dict.values())                                 # utils.py:221
    assert torch.isfinite(total)
    total.backward()
    assert net.detector_backbone.proj[3].weight.grad is not None                     # gradients reach the feature maps
    assert all(p.grad is not None for n, p in net.interaction_head.named_parameters())
    opt.step()
