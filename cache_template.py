"""V-COCO result record, importable as `cache_template.CacheTemplate` exactly like the reference's module of that name
(cache_template.py:2-15): pickles written by skghoi_amd.evaluate.save_vcoco_pickle therefore load in the reference's
vcoco_evaluation.py (`from cache_template import CacheTemplate`) and vice versa."""
from collections import defaultdict


class CacheTemplate(defaultdict):
    """Missing '<action>_agent' keys read as score 0.; missing '<action>_<role>' keys as a tiny box with score 0."""

    def __init__(self, **kwargs):
        super().__init__()
        for k, v in kwargs.items():
            self[k] = v

    def __missing__(self, k):
        if k.split("_")[-1] == "agent":
            return 0.
        return [0., 0., .1, .1, 0.]
