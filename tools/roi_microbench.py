"""Developer aid (GPU box): MultiScaleRoIAlign throughput on the 20x20 workload (40 boxes / image, 4 FPN levels)."""
import sys
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd import synth
from skghoi_amd.roi_pool import MultiScaleRoIAlign
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W = 800, 1216
feats = {str(i): torch.randn(B, 256, H // s, W // s, device="cuda") for i, s in enumerate((4, 8, 16, 32))}
boxes = [synth.make_image(1000 + i, hw=(H, W))["boxes"].cuda() for i in range(B)]
shapes = [(H, W)] * B
pool = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
for _ in range(3):
    out = pool(feats, boxes, shapes)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
R = 20
for _ in range(R):
    out = pool(feats, boxes, shapes)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / R
wr = out.numel() * 4
# bytes a box touches: its footprint on the chosen level x 256 channels x 4 B (upper bound on compulsory reads)
print("B=%d rois=%d  %.3f ms  -> %.0f images/s; output %.1f MB written -> %.1f GB/s on the write alone" % (
    B, out.shape[0], ms, B / ms * 1e3, wr / 1e6, wr / ms / 1e6))
