"""Developer aid (GPU box): N batch-4 training steps, prefetch on or off, for profiling.
usage: train_loop.py [precision=bf16] [prefetch=1] [steps=40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
pf = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
from skghoi_amd import trainer
trainer.limit_host_threads()
dev = torch.device("cuda", 0)
el, losses, _ = bench.run_train(4, prec, steps, 8, dev, 0, 1, False, prefetch=pf)
print("%s prefetch=%s: %.3f ms/step" % (prec, pf, el / steps * 1e3), losses)
