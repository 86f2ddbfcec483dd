"""Developer aid: B=256 eager forward followed by single-image forwards through the captured-graph path."""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.dont_write_bytecode = True
from collections import OrderedDict
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench
from skghoi_amd import _capi, transh

if os.environ.get("SKG_SMALL_MODE"):
    _capi.set_tuning(small_mode=int(os.environ["SKG_SMALL_MODE"]))
B = int(os.environ.get("BIG", "256"))
dev = torch.device("cuda", 0)
head = bench.build_head(dev)
dets, pooled, feats, shapes = bench.make_inputs(max(B, 4), 0, dev)
class Pool(torch.nn.Module):
    pooled = None
    def forward(self, f, b, s):
        return self.pooled
head.box_roi_pool = Pool()
with torch.no_grad():
    if B > 8:
        Pool.pooled = pooled
        torch.manual_seed(1)
        r = head(feats, dets, shapes)
        torch.cuda.synchronize()
        print("big forward ok", len(r), flush=True)
    for it, b in enumerate((0, 3, 1, 2, 0)):
        Pool.pooled = pooled[40 * b:40 * (b + 1)]
        f1 = OrderedDict((k, feats["3"][b:b + 1]) for k in "0123")
        torch.manual_seed(1)
        r1 = head(f1, dets[b:b + 1], shapes[b:b + 1])
        torch.cuda.synchronize()
        print("single", it, b, r1[0]["scores"].sum().item(), flush=True)
print("done")
