"""GPU tests at the BASELINE workload shape and of the engine's robustness:
  * B = 256 images of 20 x 20 (two 128-image chunks, the exact bench workload): images 0, 127, 128 and 255 of the batch
    against single-image forwards of the same head under the same TransH RNG position (chunk-boundary neighbours)
  * packed-weight invalidation on ANY change of the live parameters (also writes that bypass the version counters)
  * the preprocess kernel's candidate limit through the C ABI."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import cases
import gpu_run
from skghoi_amd import _capi, synth, transh

pytestmark = pytest.mark.gpu


class _Pool(torch.nn.Module):
    def __init__(self, pooled):
        super().__init__()
        self.pooled = pooled

    def forward(self, features, boxes, image_shapes):
        return self.pooled


def _batch(B, n_h=20, n_o=20):
    imgs = [synth.make_image(1000 + i, n_h=n_h, n_o=n_o) for i in range(B)]
    det = [dict(boxes=i["boxes"].cuda(), labels=i["labels"].cuda(), scores=i["scores"].cuda()) for i in imgs]
    pooled = torch.cat([i["pooled"] for i in imgs]).cuda()
    feat3 = torch.cat([i["feat3"] for i in imgs]).cuda()
    return det, pooled, feat3, [i["hw"] for i in imgs]


def test_b256_images_match_single_image_runs(precision):
    B, K, seed = 256, 117, 4242
    case = cases.build_case("full20")
    head = gpu_run.build_head(case).eval()
    det, pooled, feat3, shapes = _batch(B)
    head.box_roi_pool = _Pool(pooled)
    assert head.engine().chunk_images == 128
    feats = OrderedDict((k, feat3) for k in "0123")
    with torch.no_grad():
        torch.manual_seed(seed)
        res = head(feats, det, shapes)
        logits = head.engine().last["logits"][:, :K + 1].clone()     # (columns K+1.. pad the row to 16 bytes)
    assert len(res) == B
    for b in (0, 127, 128, 255):
        head.box_roi_pool = _Pool(pooled[40 * b:40 * (b + 1)])
        f1 = OrderedDict((k, feat3[b:b + 1]) for k in "0123")
        with torch.no_grad():
            torch.manual_seed(seed)
            transh.draw_batch(K, b)                      # the b images ahead of it consumed this much of the RNG stream
            r1 = head(f1, det[b:b + 1], shapes[b:b + 1])[0]
            l1 = head.engine().last["logits"][:, :K + 1]
        for k in ("index", "prediction", "object"):
            assert torch.equal(res[b][k], r1[k]), (b, k)
        assert torch.equal(res[b]["boxes_h"], r1["boxes_h"]) and torch.equal(res[b]["boxes_o"], r1["boxes_o"])
        lb = logits[780 * b:780 * (b + 1)]
        # same kernels, same per-row arithmetic; only the tile a row lands in differs (which does not enter a row's sum)
        assert (lb - l1).abs().max().item() <= 1e-5, (b, (lb - l1).abs().max().item())
        assert (res[b]["scores"] - r1["scores"]).abs().max().item() <= 1e-6


def _tiny_forward(head, case):
    det = gpu_run.to_cuda(case["detections"])
    feats = OrderedDict((k, case["feat3"].cuda()) for k in "0123")
    with torch.no_grad():
        torch.manual_seed(5)
        r = head(feats, det, case["shapes"])
    return head.engine().last["logits"][:, :case["cfg"]["K"] + 1].clone(), r


def test_packed_weights_follow_any_parameter_change(precision):
    """The engine keeps re-laid copies of the parameters: every route of changing a live parameter must reach the next
    forward -- in-place under no_grad, through `.data` (bumps no version counter), load_state_dict, a replaced
    Parameter object, a replaced sub-module."""
    case = cases.build_case("tiny")
    head = gpu_run.build_head(case).eval()
    base, _ = _tiny_forward(head, case)
    again, _ = _tiny_forward(head, case)
    assert torch.equal(base, again)
    pw0 = head.engine()._pw
    assert head.engine()._pw is pw0                        # nothing changed: no re-pack
    K = case["cfg"]["K"]

    p = head.box_pair_predictor.weight
    p.data.mul_(2.0)                                       # invisible to p._version
    out, _ = _tiny_forward(head, case)
    assert head.engine()._pw is not pw0
    bias = head.box_pair_predictor.bias.detach()
    assert torch.allclose(out[:, :K] - bias, 2.0 * (base[:, :K] - bias), rtol=1e-4, atol=1e-5)
    p.data.mul_(0.5)
    back, _ = _tiny_forward(head, case)
    assert torch.allclose(back, base, rtol=0, atol=1e-6)

    mid = head.box_pair_head.attention_head.fc_2[7].weight            # a middle parameter of a stacked MBF
    mid.data.add_(0.01)
    out2, _ = _tiny_forward(head, case)
    assert (out2 - base).abs().max().item() > 1e-6
    mid.data.sub_(0.01)

    sd = {k: v.clone() for k, v in head.state_dict().items()}
    sd["box_pair_suppressor.bias"] = sd["box_pair_suppressor.bias"] + 1.0
    head.load_state_dict(sd)
    out3, _ = _tiny_forward(head, case)
    assert torch.allclose(out3[:, K], base[:, K] + 1.0, atol=1e-5)

    head.box_pair_suppressor.bias = torch.nn.Parameter(head.box_pair_suppressor.bias.detach() + 1.0)   # new object
    out4, _ = _tiny_forward(head, case)
    assert torch.allclose(out4[:, K], base[:, K] + 2.0, atol=1e-5)

    new_pred = torch.nn.Linear(2048, K).cuda()
    head.box_pair_predictor = new_pred                      # replaced module
    out5, _ = _tiny_forward(head, case)
    assert (out5[:, :K] - out4[:, :K]).abs().max().item() > 1e-4


def test_preprocess_candidate_limit_is_reported_not_overrun():
    """more than SKG_MAX_DET_PER_IMAGE candidates must not run past the kernel's LDS arrays.  Straight
    through the C ABI (the Python wrapper would refuse earlier)."""
    lib = _capi.lib()
    n_bad, n_ok = _capi.MAX_DET_PER_IMAGE + 1, 7
    n = n_bad + n_ok
    g = torch.Generator().manual_seed(3)
    xy = torch.rand(n, 2, generator=g) * 500
    boxes = torch.cat([xy, xy + 20 + torch.rand(n, 2, generator=g) * 50], 1).cuda()
    scores = (torch.rand(n, generator=g) * 0.7 + 0.3).cuda()
    labels = torch.randint(0, 80, (n,), generator=g).cuda()
    labels[n_bad] = 49
    det_off = torch.tensor([0, n_bad, n], dtype=torch.int32).cuda()
    nverbs = torch.ones(80, dtype=torch.int32).cuda()
    index = torch.full((2, 30), 7, dtype=torch.int32).cuda()
    count = torch.zeros(2, 4, dtype=torch.int32).cuda()
    rc = lib.skg_preprocess_f32(boxes.data_ptr(), scores.data_ptr(), labels.data_ptr(), det_off.data_ptr(), 2, 49,
                                0.2, 0.5, 15, 15, nverbs.data_ptr(), 80, 2.8, index.data_ptr(), count.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    c = count.cpu().numpy()
    assert list(c[0]) == [-1, -1, -1, n_bad] and np.all(index[0].cpu().numpy() == -1)
    assert c[1][0] >= 1 and 1 <= c[1][1] <= n_ok            # the next image is processed normally
    # and the Python entry point raises instead of returning garbage
    head = gpu_run.build_head(cases.build_case("tiny")).eval()
    det = [dict(boxes=boxes[:n_bad], labels=labels[:n_bad], scores=scores[:n_bad])]
    with pytest.raises(_capi.SkgError):
        head.preprocess(det, None)
