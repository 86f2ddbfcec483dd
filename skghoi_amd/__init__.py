"""skghoi_amd -- MI355X-native implementation of SKGHOI's interaction-head hot path.

    from skghoi_amd import InteractionHead, GraphHead      # drop-in for heads/adamixer_transH_spatial_r50_head.py

The numerical work lives in skghoi_amd/csrc (HIP, gfx950) behind the C ABI of include/skghoi.h.
"""
__version__ = "0.1.0"

# Process-level HIP runtime settings (hardware queues) are an EXPLICIT call: skghoi_amd.runtime.configure(), before the
# process first touches the GPU (bench.py, the examples and tests/conftest.py make it).  Importing this package changes
# nothing in the environment.


def __getattr__(name):
    if name in ("InteractionHead", "GraphHead", "MultiBranchFusion", "MessageMBF", "transH_head"):
        from . import adamixer_transH_spatial_r50_head as m
        return getattr(m, name)
    raise AttributeError(name)
