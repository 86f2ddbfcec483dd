"""HOI mapping, association, 11-point AP and the exporters against the loop-level oracle (CPU)."""
import pickle

import numpy as np
import pytest
import torch

from oracle import eval_oracle as EO
from skghoi_amd import evaluate as ev, synth


def _rand_boxes(rs, n):
    xy = rs.uniform(0, 300, (n, 2)); wh = rs.uniform(20, 200, (n, 2))
    return torch.tensor(np.concatenate([xy, xy + wh], 1), dtype=torch.float32)


def test_hoi_lut_matches_object_to_verb():
    lut = ev.hico_object_n_verb_to_interaction()
    o2v = synth.hico_object_to_verb()
    assert lut.shape == (80, 117) and int((lut >= 0).sum()) == 600 and sorted(lut[lut >= 0].tolist()) == list(range(600))
    for o in range(80):
        assert sorted(torch.nonzero(lut[o] >= 0).squeeze(1).tolist()) == sorted(o2v[o])


@pytest.mark.parametrize("seed", range(6))
def test_association_matches_oracle(seed):
    rs = np.random.RandomState(seed)
    n_gt, n_det = rs.randint(0, 5), rs.randint(0, 12)
    gh, go = _rand_boxes(rs, n_gt), _rand_boxes(rs, n_gt)
    # detections = jittered copies of gt pairs + random ones
    dh = torch.cat([gh[rs.randint(0, max(n_gt, 1), n_det // 2)] + 3 if n_gt else gh[:0], _rand_boxes(rs, n_det - (n_det // 2 if n_gt else 0))])
    do = torch.cat([go[rs.randint(0, max(n_gt, 1), n_det // 2)] + 2 if n_gt else go[:0], _rand_boxes(rs, n_det - (n_det // 2 if n_gt else 0))])
    sc = torch.tensor(rs.uniform(0, 1, len(dh)), dtype=torch.float32)
    got = ev.associate_pairs(gh, go, dh, do, sc)
    want = EO.associate(gh.numpy(), go.numpy(), dh.numpy(), do.numpy(), sc.numpy())
    assert np.array_equal(got.numpy(), want)


def test_ap_meter_matches_oracle_and_known_answers():
    rs = np.random.RandomState(1)
    m = ev.DetectionAPMeter(3, num_gt=[5, 2, 0])
    all_s, all_c, all_l = [], [], []
    for _ in range(4):
        s = torch.tensor(rs.uniform(0, 1, 9)); c = torch.tensor(rs.randint(0, 3, 9)); l = torch.tensor((rs.uniform(0, 1, 9) > 0.6).astype(np.float64))
        m.append(s, c, l); all_s.append(s); all_c.append(c); all_l.append(l)
    ap = m.eval()
    S, Cc, L = torch.cat(all_s).numpy(), torch.cat(all_c).numpy(), torch.cat(all_l).numpy()
    for c, ng in enumerate([5, 2, 0]):
        want = EO.ap_11p(S[Cc == c].tolist(), L[Cc == c].tolist(), ng)
        assert abs(float(ap[c]) - want) < 1e-12
    # known answers: perfect ranking with all gt found -> 1; one TP out of two gt at rank 1 -> 6/11
    assert ev.DetectionAPMeter._ap(torch.tensor([0.9, 0.8]), torch.tensor([1., 1.]), 2, "11P") == pytest.approx(1.0)
    assert ev.DetectionAPMeter._ap(torch.tensor([0.9, 0.8]), torch.tensor([1., 0.]), 2, "11P") == pytest.approx(6 / 11)


def test_evaluator_and_exporters(tmp_path):
    lut = ev.hico_object_n_verb_to_interaction()
    o2v = synth.hico_object_to_verb()
    bh = torch.tensor([[10., 10., 100., 200.], [300., 20., 380., 220.]]); bo = torch.tensor([[50., 60., 150., 160.], [200., 100., 320., 180.]])
    obj = torch.tensor([3, 17])
    idx = torch.tensor([0, 0, 1]); pred = torch.tensor([o2v[3][0], o2v[3][1], o2v[17][0]])
    out = dict(boxes_h=bh, boxes_o=bo, index=idx, prediction=pred, scores=torch.tensor([0.9, 0.4, 0.7]), object=obj)
    hoi0 = int(lut[3, o2v[3][0]]); hoi2 = int(lut[17, o2v[17][0]])
    target = dict(boxes_h=torch.tensor([[12., 11., 98., 205.]]), boxes_o=torch.tensor([[52., 58., 149., 161.]]),
                  hoi=torch.tensor([hoi0]))
    num_gt = [0] * 600; num_gt[hoi0] = 1; num_gt[hoi2] = 1
    e = ev.HOIEvaluator(num_gt, num_anno_train=[5 if i % 2 else 50 for i in range(600)])
    labels = e.add(out, target)
    assert labels.tolist() == [1.0, 0.0, 0.0]
    s = e.summary()
    assert s["ap"][hoi0] == pytest.approx(1.0) and s["ap"][hoi2] == 0.0 and s["full"] == pytest.approx(1 / 600)
    assert "rare" in s and "non_rare" in s
    cells = ev.hicodet_mat_cells([out], [2], n_images=4)
    assert cells.shape == (600, 4) and cells[hoi0, 2].shape == (1, 9) and cells[hoi0, 0].shape == (0, 0)
    assert cells[hoi0, 2][0, 2] == 99.0 and cells[hoi0, 2][0, 8] == pytest.approx(0.9)      # x2 - 1, score
    res = ev.vcoco_results([out], [4711], ["hold obj"] * 117)
    assert len(res) == 3 and res[0]["image_id"] == 4711 and res[0]["hold_agent"] == pytest.approx(0.9)
    assert res[0]["hold_obj"][:4] == bo[0].tolist() and res[0]["sit_agent"] == 0. and res[0]["sit_instr"] == [0., 0., .1, .1, 0.]
    ev.save_vcoco_pickle(res, str(tmp_path))
    back = pickle.load(open(tmp_path / "vcoco_results.pkl", "rb"))
    assert len(back) == 3 and back[1]["person_box"] == bh[0].tolist()
    assert type(back[0]).__module__ == "cache_template" and back[0]["carry_obj"] == [0., 0., .1, .1, 0.]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_device_evaluator_matches_host_and_oracle(seed):
    """skg_eval_associate_f32 + skg_eval_ap11_f64 (DeviceHOIEvaluator) against the host evaluator (torch restatement) and,
    per image and class, the loop-level oracle: labels bit-equal, the 600 APs equal in float64."""
    rs = np.random.RandomState(seed)
    lut = ev.hico_object_n_verb_to_interaction()
    o2v = synth.hico_object_to_verb()
    n_img = 5
    num_gt = [0] * 600
    outs, tgts = [], []
    for b in range(n_img):
        n_pairs = rs.randint(0, 14) if b != 3 else 0                 # one image without detections
        bh = _rand_boxes(rs, n_pairs); bo = _rand_boxes(rs, n_pairs)
        obj = torch.tensor(rs.randint(0, 80, n_pairs), dtype=torch.int64)
        index, pred = [], []
        for p in range(n_pairs):
            vs = o2v[int(obj[p])]
            for v in rs.choice(vs, size=min(len(vs), rs.randint(1, 4)), replace=False):
                index.append(p); pred.append(int(v))
        index = torch.tensor(index, dtype=torch.int64); pred = torch.tensor(pred, dtype=torch.int64)
        scores = torch.tensor(np.round(rs.uniform(0, 1, len(index)), 1), dtype=torch.float32)      # many ties
        # ground truth: jittered copies of some detected pairs (same class), some duplicated, plus unrelated ones
        gh, go, ghoi = [], [], []
        for c in rs.permutation(len(index))[:rs.randint(0, 6)]:
            p = int(index[c])
            for _ in range(rs.randint(1, 3)):
                gh.append(bh[p] + float(rs.uniform(-4, 4))); go.append(bo[p] + float(rs.uniform(-4, 4)))
                ghoi.append(int(lut[int(obj[p]), int(pred[c])]))
        for _ in range(rs.randint(0, 3)):
            gh.append(_rand_boxes(rs, 1)[0]); go.append(_rand_boxes(rs, 1)[0]); ghoi.append(int(rs.randint(0, 600)))
        for h in ghoi:
            num_gt[h] += 1
        tgts.append(dict(boxes_h=torch.stack(gh) if gh else torch.zeros(0, 4), boxes_o=torch.stack(go) if go else torch.zeros(0, 4),
                         hoi=torch.tensor(ghoi, dtype=torch.int64)))
        outs.append(dict(boxes_h=bh, boxes_o=bo, index=index, prediction=pred, scores=scores, object=obj))
    host = ev.HOIEvaluator(num_gt, num_anno_train=[5 if i % 3 else 50 for i in range(600)])
    want_labels = [host.add(o, t) for o, t in zip(outs, tgts)]
    dev = ev.DeviceHOIEvaluator(num_gt, num_anno_train=[5 if i % 3 else 50 for i in range(600)])
    cu = lambda d: {k: v.cuda() for k, v in d.items()}
    got = dev.add([cu(o) for o in outs[:2]], [cu(t) for t in tgts[:2]]) + dev.add([cu(o) for o in outs[2:]], tgts[2:])
    for b in range(n_img):
        assert torch.equal(got[b].cpu(), want_labels[b]), b
        # and the oracle, class by class
        o, t = outs[b], tgts[b]
        inter = lut[o["object"][o["index"]], o["prediction"]]
        for h in inter.unique().tolist():
            det = torch.nonzero(inter == h).squeeze(1); g = torch.nonzero(t["hoi"] == h).squeeze(1)
            if len(g):
                w = EO.associate(t["boxes_h"][g].numpy(), t["boxes_o"][g].numpy(), o["boxes_h"][o["index"]][det].numpy(),
                                 o["boxes_o"][o["index"]][det].numpy(), o["scores"][det].numpy())
                assert np.array_equal(got[b].cpu().numpy()[det.numpy()], w), (b, h)
    sh, sd = host.summary(), dev.summary()
    assert torch.equal(sh["ap"], sd["ap"]) and sh["full"] == sd["full"] and sh["rare"] == sd["rare"]
    assert float(sd["ap"].sum()) > 0


VCOCO_ACTIONS = ["hold obj", "sit instr", "ride instr", "look obj", "hit instr", "hit obj", "eat obj", "eat instr",
                 "jump instr", "lay instr", "talk_on_phone instr", "carry obj", "throw obj", "catch obj", "cut instr",
                 "cut obj", "work_on_computer instr", "ski instr", "surf instr", "skateboard instr", "drink instr",
                 "kick obj", "read obj", "snowboard instr"]           # VCOCO.actions (K = 24, cache.py:165-168)


def test_vcoco_pickle_is_consumable_the_way_vsrl_eval_reads_it(tmp_path):
    """the pickle goes through the reader of the external V-COCO toolkit (restated in
    oracle/vsrl_consumer.py): 26 actions x roles per record, defaults for everything the record does not carry, and a
    known-answer role AP computed from what was read."""
    from oracle import vsrl_consumer as VC
    bh = torch.tensor([[10., 10., 110., 210.], [300., 20., 380., 220.]]); bo = torch.tensor([[50., 60., 150., 160.], [200., 100., 320., 180.]])
    a_hold, a_cut_i, a_cut_o = VCOCO_ACTIONS.index("hold obj"), VCOCO_ACTIONS.index("cut instr"), VCOCO_ACTIONS.index("cut obj")
    out = dict(boxes_h=bh, boxes_o=bo, index=torch.tensor([0, 1, 1]), prediction=torch.tensor([a_hold, a_cut_i, a_cut_o]),
               scores=torch.tensor([0.9, 0.6, 0.3]), object=torch.tensor([40, 44]))
    res = ev.vcoco_results([out], [4711], VCOCO_ACTIONS)
    ev.save_vcoco_pickle(res, str(tmp_path))
    dets = pickle.load(open(tmp_path / "vcoco_results.pkl", "rb"))
    for d in dets:                                                   # every key the reader asks for resolves
        for action, roles in VC.ACTION_ROLES:
            for r in roles:
                v = d[action + "_" + r]
                assert np.isscalar(v) if r == "agent" else len(v) == 5
    assert {a.split()[0] for a in VCOCO_ACTIONS} <= {a for a, _ in VC.ACTION_ROLES}
    agents, roles = VC.collect_detections_for_image(dets, 4711)
    assert agents.shape == (3, 4 + 26) and roles.shape == (3, 130, 2)
    hold, cut = 0, 14
    assert agents[0, :4].tolist() == bh[0].tolist() and agents[0, 4 + hold] == pytest.approx(0.9)
    assert roles[0, 5 * hold:5 * hold + 5, 0].tolist() == pytest.approx(bo[0].tolist() + [0.9])
    assert agents[1, 4 + cut] == pytest.approx(0.6) and agents[2, 4 + cut] == pytest.approx(0.3)
    assert roles[1, 5 * cut:5 * cut + 5, 0].tolist() == pytest.approx(bo[1].tolist() + [0.6])      # cut instr = role 1
    assert roles[2, 5 * cut:5 * cut + 5, 1].tolist() == pytest.approx(bo[1].tolist() + [0.3])      # cut obj   = role 2
    assert roles[0, 5 * cut:5 * cut + 5, 0].tolist() == pytest.approx([0., 0., .1, .1, 0.])        # template default
    assert VC.collect_detections_for_image(dets, 1)[0].shape == (0, 30)
    gt = [dict(image_id=4711, person_box=[12., 11., 108., 205.], role_box=[52., 58., 149., 161.])]
    assert VC.role_ap(dets, gt, "hold", 1) == pytest.approx(1.0)                # found at rank 1
    assert VC.role_ap(dets, gt, "cut", 2) == pytest.approx(0.0)                 # the cut detection is another pair


def _meter_case(seed, n=400, ncls=7):
    rs = np.random.RandomState(seed)
    sc = torch.tensor(np.round(rs.uniform(0, 1, n), 2), dtype=torch.float32)          # two decimals: plenty of score ties
    cl = torch.tensor(rs.randint(0, ncls, n)); cl[cl == 3] = 2                         # class 3 never predicted
    lb = torch.tensor((rs.uniform(0, 1, n) < 0.3).astype(np.float32)); lb[cl == 5] = 0  # class 5 without a positive
    return sc, cl, lb


def test_training_meter_without_num_gt_matches_the_loop_oracle():
    """The reference's training / validation meters carry no ground-truth counts (utils.py:208, 285): recall over the
    positives among the logged detections.  Host meter and the asynchronous meter (CPU tensors here) against the oracle."""
    sc, cl, lb = _meter_case(0)
    host = ev.DetectionAPMeter(7)
    dev = ev.DeviceAPMeter(7, device="cpu")
    for lo in range(0, 400, 64):                                    # appended batch by batch, like the training loop
        host.append(sc[lo:lo + 64], cl[lo:lo + 64], lb[lo:lo + 64]); dev.append(sc[lo:lo + 64], cl[lo:lo + 64], lb[lo:lo + 64])
    a, b = host.eval(), dev.eval()
    assert torch.equal(a, b) and len(dev) == 400
    for c in range(7):
        m = cl == c
        want = EO.ap_11p(sc[m].double().numpy(), lb[m].double().numpy(), int(lb[m].sum())) if m.any() else 0.0
        assert abs(float(a[c]) - want) < 1e-12, c
    assert float(a[3]) == 0.0 and float(a[5]) == 0.0
    dev.reset()
    assert len(dev) == 0 and float(dev.eval().sum()) == 0.0
    # results-dict form: what Trainer.log_results feeds it
    res = [dict(scores=sc[:100], prediction=cl[:100], labels=lb[:100]), dict(scores=sc[:0], prediction=cl[:0], labels=lb[:0]),
           dict(scores=sc[100:], prediction=cl[100:], labels=lb[100:]), dict(hoi_loss=torch.tensor(1.0))]
    dev.append_results(res)
    assert torch.equal(dev.eval(), a)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_training_meter_on_the_device_matches_the_host_meter(seed):
    sc, cl, lb = _meter_case(seed, n=3000, ncls=117)
    host = ev.DetectionAPMeter(117)
    dev = ev.DeviceAPMeter(117, device="cuda")
    for lo in range(0, 3000, 500):
        host.append(sc[lo:lo + 500], cl[lo:lo + 500], lb[lo:lo + 500])
        dev.append(sc[lo:lo + 500].cuda(), cl[lo:lo + 500].cuda(), lb[lo:lo + 500].cuda())
    assert torch.equal(host.eval(), dev.eval())
    with_gt = [int(lb[cl == c].sum()) + 2 for c in range(117)]
    h2 = ev.DetectionAPMeter(117, with_gt); h2.append(sc, cl, lb)
    d2 = ev.DeviceAPMeter(117, with_gt, device="cuda"); d2.append(sc.cuda(), cl.cuda(), lb.cuda())
    assert torch.equal(h2.eval(), d2.eval())
