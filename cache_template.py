"""V-COCO result record under the import path the reference uses (`from cache_template import CacheTemplate`,
vcoco_evaluation.py), so pickles written by skghoi_amd.evaluate.save_vcoco_pickle unpickle in the reference's
evaluation script and the reference's pickles unpickle here.

Contract (reference cache_template.py:2-15): a mapping pre-filled from keyword arguments; reading an absent
'<action>_agent' key yields the score 0.0, reading any other absent key yields a dummy role entry
[x1, y1, x2, y2, score] = [0, 0, 0.1, 0.1, 0].  Lookups never insert.
"""

_NO_ROLE = (0., 0., .1, .1, 0.)


class CacheTemplate(dict):
    def __init__(self, **fields):
        dict.__init__(self, fields)

    def __missing__(self, key):
        is_agent_score = key.rsplit("_", 1)[-1] == "agent"
        return 0. if is_agent_score else list(_NO_ROLE)
