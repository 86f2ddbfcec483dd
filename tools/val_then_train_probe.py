"""Developer aid (GPU): does a validation stream (bench.b4_validate) in the same process change the batch-4 bf16 training step?
train -> b4_validate -> train -> gc.collect -> train, ms per step each."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()
import bench
from skghoi_amd import trainer
trainer.limit_host_threads()
dev = torch.device("cuda", 0)


def train(tag):
    el, _, _ = bench.run_train(4, "bf16", 100, 12, dev, 0, 1, False)
    print("%-28s %.3f ms/step" % (tag, el / 100 * 1e3), flush=True)


train("fresh process")
train("again")
if len(sys.argv) > 1 and sys.argv[1] == "b1":
    r = bench.b1_stream(dev)
    print("b1_stream", r["steady_state"]["mean_ms"], flush=True)
else:
    r = bench.b4_validate(dev)
    print("b4_validate", r["steady_state"]["mean_ms"], flush=True)
train("after the eval leg")
gc.collect(); torch.cuda.empty_cache()
train("after gc + empty_cache")
