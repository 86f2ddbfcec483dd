"""Developer aid (GPU box): several training runs in ONE process, with allocator events between them: does a run's speed
depend on what the caching allocator holds?   usage: _trainleg_probe.py bf16 gc bf16 empty bf16 ..."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench
from skghoi_amd import trainer
dev = torch.device("cuda", 0)
trainer.limit_host_threads(1)
g = lambda d, k: d.get(k, 0)
for item in sys.argv[1:]:
    if item == "gc":
        gc.collect(); print("gc.collect"); continue
    if item == "empty":
        torch.cuda.empty_cache(); print("empty_cache"); continue
    if item.startswith("eval"):                       # evalB[s]: forwards of B images (s = the engine's two chunk streams)
        B = int(item[4:].rstrip("s"))
        head = bench.build_head(dev)
        dets, pooled, feats, shapes = bench.make_inputs(B, 0, dev)
        head.box_roi_pool = bench.ResidentPool(pooled)
        head.engine().n_streams = 2 if item.endswith("s") else 1
        with torch.no_grad():
            for _ in range(12):
                head(feats, dets, shapes)
        torch.cuda.synchronize()
        print("%s done (plans: %s)" % (item, getattr(head.engine()._small, "plans", None) and len(head.engine()._small.plans)))
        if "keep" not in os.environ.get("SKG_PROBE", ""):
            bench.release_plans(head)
        del head
        continue
    if item == "stream":
        b = bench.b1_stream(dev, n_forwards=512, n_images=128); print("b1_stream", b["plans"]); continue
    st0 = torch.cuda.memory_stats()
    el, losses, inf = bench.run_train(4, item, 60, 12, dev, 0, 1, False)
    st1 = torch.cuda.memory_stats()
    print("%s: %.3f ms/step" % (item, el / 60 * 1e3))
    print("   segments %d -> %d (allocs %d, frees %d), retries %d, reserved %.2f GB, active blocks %d, inactive split %.1f MB" % (
        g(st0, "segment.all.current"), g(st1, "segment.all.current"),
        g(st1, "segment.all.allocated") - g(st0, "segment.all.allocated"), g(st1, "segment.all.freed") - g(st0, "segment.all.freed"),
        g(st1, "num_alloc_retries") - g(st0, "num_alloc_retries"), g(st1, "reserved_bytes.all.current") / 2**30,
        g(st1, "active.all.current"), g(st1, "inactive_split_bytes.all.current") / 2**20), flush=True)
