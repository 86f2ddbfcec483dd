"""skghoi_amd -- MI355X-native implementation of SKGHOI's interaction-head hot path.

    from skghoi_amd import InteractionHead, GraphHead      # drop-in for heads/adamixer_transH_spatial_r50_head.py

The numerical work lives in skghoi_amd/csrc (HIP, gfx950) behind the C ABI of include/skghoi.h.
"""
import os as _os

__version__ = "0.1.0"

# The HIP runtime opens up to GPU_MAX_HW_QUEUES (default 4) hardware queues per stream-priority class and hands them to
# streams round-robin.  On MI355X / ROCm 7.2 a process that gets to FOUR busy queues of the normal class -- one captured
# or replayed hipGraph is enough, or four live streams -- schedules the training step's dependent kernels ~40 % worse
# from then on (1.40 -> 1.98 ms per batch-4 step, same kernels; tools/_trainleg_probe.py).  Three keeps every
# measured case fast and costs the eval paths nothing (values below three crash the runtime's graph path: do not use).
# Only effective if set before the first HIP call of the process -- i.e. import skghoi_amd (or export it) before
# touching the GPU; an explicit setting of the variable is respected.
if _os.environ.get("WORLD_SIZE", "1") == "1":            # (multi-process runs keep the runtime's default beside RCCL's queues)
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")


def __getattr__(name):
    if name in ("InteractionHead", "GraphHead", "MultiBranchFusion", "MessageMBF", "transH_head"):
        from . import adamixer_transH_spatial_r50_head as m
        return getattr(m, name)
    raise AttributeError(name)
