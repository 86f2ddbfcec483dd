"""The 600-way HOI table the evaluator and the verb tables are built from (skghoi_amd/data/hico_object_to_verb.json,
SURVEY 8f-2): its HOI ORDER must be the reference's (hicodet/hico_text_label.py key order = HICO-DET's official HOI
index, which hicodet/hicodet.py:139-153 turns into object_n_verb_to_interaction).

  * known answers from the published HICO-DET class list (no reference needed): the 20 PASCAL-VOC object classes come
    first, the other 60 alphabetically; HOI 1-10 are the airplane classes board .. no_interaction, HOI 600 is zebra
    no_interaction; 600 classes, 80 objects, 117 verbs, `no_interaction` closes every object's block
  * reference-marked (build container): the committed JSON equals what the reference's file yields, and the object -> verb
    sets agree with the file's own, independently written hico_action_valid_object_list."""
import ast
import json
import os

import pytest

from skghoi_amd import evaluate, synth

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "skghoi_amd", "data",
                    "hico_object_to_verb.json")
# HICO-DET object names in the dataset's alphabetical class index (hico_obj_classes) -- only the ones asserted below
AIRPLANE, BICYCLE, BIRD, PERSON, TV, APPLE, ZEBRA = 0, 9, 10, 49, 75, 1, 79
NO_INTERACTION, BOARD, RIDE, CARRY, HOLD = 57, 4, 76, 8, 36


def _table():
    return json.load(open(DATA))


def test_official_hoi_order_known_answers():
    t = _table()
    pairs = [tuple(p) for p in t["hoi_verb_object"]]
    assert len(pairs) == 600 == len(set(pairs))
    assert pairs[0] == (BOARD, AIRPLANE) and pairs[6] == (RIDE, AIRPLANE) and pairs[9] == (NO_INTERACTION, AIRPLANE)
    assert pairs[10] == (CARRY, BICYCLE) and pairs[11] == (HOLD, BICYCLE)
    assert pairs[599] == (NO_INTERACTION, ZEBRA)
    blocks = []
    for v, o in pairs:
        if not blocks or blocks[-1][0] != o:
            blocks.append([o, []])
        blocks[-1][1].append(v)
    assert len(blocks) == 80                                        # every object's HOIs are contiguous
    order = [b[0] for b in blocks]
    assert order[:3] == [AIRPLANE, BICYCLE, BIRD] and order[15] == PERSON and order[19] == TV       # the 20 VOC classes first
    assert order[20] == APPLE and order[20:] == sorted(order[20:]) and order[-1] == ZEBRA           # then alphabetical
    assert all(b[1][-1] == NO_INTERACTION for b in blocks)          # `no_interaction` closes every block
    assert len(blocks[0][1]) == 10
    o2v = t["object_to_verb"]
    assert len(o2v) == 80 and sum(map(len, o2v)) == 600
    assert max(max(r) for r in o2v) == 116 and o2v == synth.hico_object_to_verb()
    for o, verbs in blocks:
        assert o2v[o] == verbs                                      # per object in HOI order (hicodet.py:176-178)


def test_lut_is_the_inverse_of_the_hoi_list():
    lut = evaluate.hico_object_n_verb_to_interaction()
    pairs = _table()["hoi_verb_object"]
    assert lut.shape == (80, 117) and int((lut >= 0).sum()) == 600
    for i, (v, o) in enumerate(pairs):
        assert int(lut[o, v]) == i                                  # hicodet.py:139-153: lut[obj][verb] = hoi index


@pytest.mark.reference
def test_json_equals_reference_file_and_its_valid_object_list():
    ref = os.path.join(os.environ.get("SKG_REFERENCE_ROOT", "/root/reference"), "hicodet", "hico_text_label.py")
    tables = {}
    for node in ast.parse(open(ref).read()).body:                   # parsed as data, never executed
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name):
            try:
                tables[node.targets[0].id] = ast.literal_eval(node.value)
            except Exception:
                pass
    keys = [list(k) for k in tables["hico_text_label"].keys()]       # (verb, object) in HOI order
    t = _table()
    assert t["hoi_verb_object"] == keys
    valid = tables["hico_action_valid_object_list"]                  # object -> verbs, written out independently
    assert sorted(valid) == list(range(80))
    for o in range(80):
        assert sorted(valid[o]) == sorted(t["object_to_verb"][o]), o
    assert tables["hico_obj_classes"][49] == "person" and tables["hico_action_classes"][57] == "no interaction"
    assert tables["hico_obj_classes"][75] == "tv" and tables["hico_obj_classes"][9] == "bicycle"
