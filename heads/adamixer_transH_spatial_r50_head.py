"""Same module name and location as the reference's heads/adamixer_transH_spatial_r50_head.py: putting <repo>/heads on
sys.path (the reference does `sys.path.append('.../heads')`, models/adamixer_transH_spatial_r50_models.py:24) makes
`from adamixer_transH_spatial_r50_head import InteractionHead, GraphHead` resolve to the MI355X implementation."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from skghoi_amd.adamixer_transH_spatial_r50_head import (  # noqa: E402,F401
    GraphHead, InteractionHead, MessageMBF, MultiBranchFusion, transH_head)
