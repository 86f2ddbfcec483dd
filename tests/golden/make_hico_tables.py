"""Build-container-only generator for skghoi_amd/data/hico_object_to_verb.json.

Reads the reference's hicodet/hico_text_label.py AS DATA (ast.literal_eval of the `hico_text_label` dict literal, the
file is never executed) and derives, for each of the 80 HICO-DET object classes, the verb ids (HOI-index order) that form a valid
HOI with it -- the table HICODet.object_to_verb (hicodet/hicodet.py:168-179) hands to GraphHead as
`object_class_to_target_class` (models/adamixer_transH_spatial_r50_models.py:171), plus the (verb, object) pair of each
of the 600 HOI classes in HOI-index order (hicodet/hicodet.py:139-153).
"""
import ast, json, os, sys

REF = os.environ.get("SKG_REFERENCE_ROOT", "/root/reference")
src = open(os.path.join(REF, "hicodet", "hico_text_label.py")).read()
tree = ast.parse(src)
tables = {}
for node in tree.body:
    if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name):
        try:
            tables[node.targets[0].id] = ast.literal_eval(node.value)
        except Exception:
            pass
pairs = list(tables["hico_text_label"].keys())        # (verb, object) in HOI order
assert len(pairs) == 600
o2v = [[] for _ in range(80)]
for v, o in pairs:
    o2v[o].append(v)
# kept in HOI-index order, as hicodet.py:176-178 appends them
out = dict(object_to_verb=o2v, hoi_verb_object=[[int(v), int(o)] for v, o in pairs])
dst = os.path.join(os.path.dirname(__file__), "..", "..", "skghoi_amd", "data", "hico_object_to_verb.json")
json.dump(out, open(dst, "w"), separators=(",", ":"))
print("wrote", os.path.abspath(dst), "verbs per class min/max", min(map(len, o2v)), max(map(len, o2v)))
