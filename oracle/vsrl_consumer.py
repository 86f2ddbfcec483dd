"""ORACLE (test infrastructure): the CONSUMER side of the V-COCO export -- how the external `vsrl_eval.VCOCOeval`
(vcoco_evaluation.py:3-10 imports it from a path outside the repository; the package is absent here, so parity is
unpinned at this boundary) reads `vcoco_results.pkl`: per image it collects, from every record,

    agents[i] = [person_box (4) | <action>_agent score for each of the 26 V-COCO actions]
    roles[i, 5 * aid : 5 * aid + 5, j - 1] = record['<action>_<role j>']  = [x1, y1, x2, y2, score]

(`_collect_detections_for_image` of the published V-COCO toolkit), and scores role detection as average precision over
(person, role) matches at IoU >= 0.5 (`_do_role_eval`, scenario 1, VOC-style area under the monotone precision-recall
curve).  Restated from the published algorithm; only what the export's format has to satisfy is kept.
"""
import numpy as np

# The 26 V-COCO actions and their roles (role 0 is always the agent), as the toolkit's annotation files list them.
ACTION_ROLES = [("hold", ["agent", "obj"]), ("stand", ["agent"]), ("sit", ["agent", "instr"]), ("ride", ["agent", "instr"]),
                ("walk", ["agent"]), ("look", ["agent", "obj"]), ("hit", ["agent", "instr", "obj"]),
                ("eat", ["agent", "obj", "instr"]), ("jump", ["agent", "instr"]), ("lay", ["agent", "instr"]),
                ("talk_on_phone", ["agent", "instr"]), ("carry", ["agent", "obj"]), ("throw", ["agent", "obj"]),
                ("catch", ["agent", "obj"]), ("cut", ["agent", "instr", "obj"]), ("run", ["agent"]),
                ("work_on_computer", ["agent", "instr"]), ("ski", ["agent", "instr"]), ("surf", ["agent", "instr"]),
                ("skateboard", ["agent", "instr"]), ("smile", ["agent"]), ("drink", ["agent", "instr"]),
                ("kick", ["agent", "obj"]), ("point", ["agent", "instr"]), ("read", ["agent", "obj"]),
                ("snowboard", ["agent", "instr"])]
NUM_ACTIONS = len(ACTION_ROLES)


def collect_detections_for_image(dets, image_id):
    agents = np.empty((0, 4 + NUM_ACTIONS), dtype=np.float32)
    roles = np.empty((0, 5 * NUM_ACTIONS, 2), dtype=np.float32)
    for det in dets:
        if det["image_id"] != image_id:
            continue
        this_agent = np.zeros((1, 4 + NUM_ACTIONS), dtype=np.float32)
        this_role = np.zeros((1, 5 * NUM_ACTIONS, 2), dtype=np.float32)
        this_agent[0, :4] = det["person_box"]
        for aid, (action, rs) in enumerate(ACTION_ROLES):
            for j, rid in enumerate(rs):
                if rid == "agent":
                    this_agent[0, 4 + aid] = det[action + "_" + rid]
                else:
                    this_role[0, 5 * aid:5 * aid + 5, j - 1] = det[action + "_" + rid]
        agents = np.concatenate((agents, this_agent), axis=0)
        roles = np.concatenate((roles, this_role), axis=0)
    return agents, roles


def _iou(a, b):
    iw = min(a[2], b[2]) - max(a[0], b[0]) + 1.0
    ih = min(a[3], b[3]) - max(a[1], b[1]) + 1.0
    if iw <= 0 or ih <= 0:
        return 0.0
    ua = (a[2] - a[0] + 1.0) * (a[3] - a[1] + 1.0) + (b[2] - b[0] + 1.0) * (b[3] - b[1] + 1.0) - iw * ih
    return iw * ih / ua


def voc_ap(rec, prec):
    rec = np.concatenate(([0.0], rec, [1.0])); prec = np.concatenate(([0.0], prec, [0.0]))
    for i in range(len(prec) - 2, -1, -1):
        prec[i] = max(prec[i], prec[i + 1])
    idx = np.where(rec[1:] != rec[:-1])[0] + 1
    return float(np.sum((rec[idx] - rec[idx - 1]) * prec[idx]))


def role_ap(dets, gt, action, role_j, ovr_thresh=0.5):
    """gt: list of {image_id, person_box, role_box}.  AP of (<action>, role j) detections, scenario 1 (a ground-truth role
    without a box is matched by an empty prediction; not needed by the format check and left out)."""
    aid = [a for a, _ in ACTION_ROLES].index(action)
    cand = []
    for image_id in sorted({g["image_id"] for g in gt} | {d["image_id"] for d in dets}):
        agents, roles = collect_detections_for_image(dets, image_id)
        for i in range(len(agents)):
            r = roles[i, 5 * aid:5 * aid + 5, role_j - 1]
            cand.append((float(r[4]), image_id, agents[i, :4].copy(), r[:4].copy()))
    cand.sort(key=lambda c: -c[0])
    used = set()
    tp, fp = [], []
    for s, image_id, pbox, rbox in cand:
        hit = False
        for gi, g in enumerate(gt):
            if g["image_id"] != image_id or gi in used:
                continue
            if _iou(pbox, g["person_box"]) >= ovr_thresh and _iou(rbox, g["role_box"]) >= ovr_thresh:
                used.add(gi); hit = True
                break
        tp.append(1.0 if hit else 0.0); fp.append(0.0 if hit else 1.0)
    tp, fp = np.cumsum(tp), np.cumsum(fp)
    return voc_ap(tp / max(len(gt), 1), tp / np.maximum(tp + fp, np.finfo(np.float64).eps))
