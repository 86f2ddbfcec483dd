import sys
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import torch, bench
dev = torch.device("cuda", 0)
head = bench.build_head(dev)
dets, pooled, feats, shapes = bench.make_inputs(256, 0, dev)
head.box_roi_pool = bench.ResidentPool(pooled)
with torch.no_grad():
    for _ in range(2): head(feats, dets, shapes)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        head(feats, dets, shapes); torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=14, max_name_column_width=60))
ev = [e for e in prof.events() if "Memcpy" in e.name or "copy_" in e.name]
import collections
c = collections.Counter()
for e in ev:
    st = [s for s in (e.stack or []) if "skghoi_amd" in s or "bench.py" in s]
    c[(e.name, st[0] if st else "?")] += 1
for k, v in c.most_common(25): print(v, k)
