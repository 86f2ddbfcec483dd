"""Developer aid (GPU box): capture -> replay -> weight change -> (retire, reap) capture -> replay, over and over in ONE
process, single images at test width -- the sequence behind the two hipGraphLaunch crash records of round 5 (DESIGN.md
section 8).  Prints one line per iteration (flushed), so a crash shows where it happened.
usage: graph_churn_probe.py [iterations=40]      env: SKG_G1_ON_SIDE=0|1, SKG_G1_ROOT_NODE=0|1"""
import faulthandler, gc, os, sys
faulthandler.enable()
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.dont_write_bytecode = True
from collections import OrderedDict
import torch
from skghoi_amd import runtime as _rt; _rt.configure()
import cases, gpu_run

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
names = ["tiny", "iter1", "nms", "tiny"]
head = None
for it in range(n):
    case = cases.build_case(names[it % len(names)])
    if head is None or it % 5 == 0:
        head = gpu_run.build_head(case).eval()
        gc.collect()
    else:
        head = gpu_run.build_head(case).eval()
    eng = head.engine()
    eng.small_batch_max, eng.small_batch_buckets, eng.small_capture_after = 8, bool(it & 1), 1
    det = gpu_run.to_cuda(case["detections"]); feats = OrderedDict((k, case["feat3"].cuda()) for k in "0123")
    with torch.no_grad():
        torch.manual_seed(it)
        for _ in range(2):
            head(feats, det, case["shapes"])                 # capture, replay
        head.box_pair_predictor.weight.data.mul_(1.01)       # every plan retired at the next call
        for _ in range(2):
            r = head(feats, det, case["shapes"])             # (reap,) capture, replay
    torch.cuda.synchronize()
    st = eng._small.stats()
    print("iteration %d ok: captures %d hits %d, scores %d" % (it, st["captures"], st["hits"], r[0]["scores"].numel()), flush=True)
print("done: %d iterations, g1_on_side=%s" % (n, os.environ.get("SKG_G1_ON_SIDE", "0")), flush=True)
