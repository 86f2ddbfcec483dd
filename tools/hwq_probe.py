"""Developer aid (GPU box): what puts a process into the slow scheduling mode of DESIGN 9 ("hardware queues"), and what
keeps it out.  Each variant runs in a process of its own (the runtime reads GPU_MAX_HW_QUEUES once):

    python tools/hwq_probe.py            # runs every variant as a child, prints one line each
    python tools/hwq_probe.py child ...  # one variant

Sequence of a variant: train bf16 (60 steps) -> 12 single-image eval forwards (captured plan replayed) -> train bf16."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def child(prefetch, keep_plans):
    from skghoi_amd import runtime
    runtime.configure()
    import torch
    import bench
    from skghoi_amd import trainer
    dev = torch.device("cuda", 0)
    trainer.limit_host_threads(1)
    out = dict(runtime=runtime.info(), prefetch=prefetch)

    def train(tag):
        el, _, _ = bench.run_train(4, "bf16", 60, 12, dev, 0, 1, False, prefetch=prefetch)
        out[tag] = round(el / 60 * 1e3, 3)
    train("train_before")
    head = bench.build_head(dev)
    dets, pooled, feats, shapes = bench.make_inputs(1, 0, dev)
    head.box_roi_pool = bench.ResidentPool(pooled)
    with torch.no_grad():
        for _ in range(12):
            head(feats, dets, shapes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            head(feats, dets, shapes)
        torch.cuda.synchronize()
        out["eval_b1_ms"] = round((time.perf_counter() - t0) * 10, 4)
    if not keep_plans:
        bench.release_plans(head)
    train("train_after")
    if not prefetch:
        pass
    print(json.dumps(out))


VARIANTS = [
    ("q4 two-branch graph", dict(SKG_HW_QUEUES="0"), ["1", "0"]),
    ("q4 ONE-branch graph", dict(SKG_HW_QUEUES="0", SKG_SMALL_ONE_BRANCH="1"), ["1", "0"]),
    ("q4 two-branch, train without look-ahead stream", dict(SKG_HW_QUEUES="0"), ["0", "0"]),
    ("q4 one-branch, plans kept alive", dict(SKG_HW_QUEUES="0", SKG_SMALL_ONE_BRANCH="1"), ["1", "1"]),
    ("q3 two-branch graph", dict(SKG_HW_QUEUES="3"), ["1", "0"]),
    ("q3 one-branch graph", dict(SKG_HW_QUEUES="3", SKG_SMALL_ONE_BRANCH="1"), ["1", "0"]),
    ("q2 one-branch graph", dict(SKG_HW_QUEUES="2", SKG_SMALL_ONE_BRANCH="1"), ["1", "0"]),
]

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2] == "1", sys.argv[3] == "1")
        sys.exit(0)
    only = sys.argv[1:]
    for name, env, args in VARIANTS:
        if only and not any(o in name for o in only):
            continue
        e = dict(os.environ, **env)
        e.pop("GPU_MAX_HW_QUEUES", None)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"] + args, env=e, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=240)
            line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
            print("%-52s rc %d  %s" % (name, r.returncode, line[-1] if line else r.stderr.decode()[-300:].replace("\n", " | ")),
                  flush=True)
        except subprocess.TimeoutExpired:
            print("%-52s TIMEOUT" % name, flush=True)
