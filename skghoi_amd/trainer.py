"""Thin data-parallel trainer shell for the interaction head.

Mirrors the reference's settings: AdamW with two parameter groups -- interaction head at `lr`, everything else (the
detector backbone/neck when it is fine-tuned) at `lr * 0.1`, weight decay 1e-4
(configures/hicodet/adamixer_transH_spatial_r50_main.py:109-127), LambdaLR that multiplies the rate by `lr_decay`
from epoch `milestone` on (main:128-132), total loss = plain sum of the loss dict (utils.py:221), ValueError on a NaN
HOI loss (utils.py:218-219), DDP with find_unused_parameters=True (utils.py:202-205).  One process per GPU;
`backend="nccl"` is RCCL on ROCm.  The three `n_p` all-reduces of the reference are issued by the head itself
(InteractionHead.distributed=True).
"""
import math

import torch
import torch.distributed as dist
from torch import nn


def limit_host_threads(ranks_on_host: int = 1, cap: int = 4) -> int:
    """Sizes torch's intra-op CPU pool for the training loop: this process's share of the usable cores (cgroup quota
    aware, skghoi_amd.dist.host_cpu_share), at most `cap`.  The step is host-bound and its CPU ops are tiny
    (randperm, small cats): on a 256-core host with a 16-core quota the default 128 threads made it 3-4x slower."""
    from .dist import host_cpu_share
    n = max(1, min(cap, host_cpu_share() // max(1, ranks_on_host)))
    torch.set_num_threads(n)
    return n


def build_optimizer(net: nn.Module, lr: float = 1e-4, weight_decay: float = 1e-4, head_key: str = "interaction_head"):
    """main:109-127: parameters whose name contains `head_key` train at lr, the rest at lr * 0.1.  A bare
    InteractionHead (no wrapper, so no 'interaction_head' in its names) is treated as all-head."""
    named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
    head = [p for n, p in named if head_key in n]
    rest = [p for n, p in named if head_key not in n]
    if not head:
        head, rest = rest, []
    groups = [{"params": head}]
    if rest:
        groups.append({"params": rest, "lr": lr * 0.1})
    # same update rule; on the GPU the multi-tensor "fused" implementation is one launch per step instead of ~10 per
    # parameter group walk (6 ms of host time for the head's 408 tensors)
    on_gpu = bool(named) and all(p.is_cuda for _, p in named)
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, **({"fused": True} if on_gpu else {}))


def build_scheduler(optimizer, milestone: int = 6, lr_decay: float = 0.1):
    """main:128-132."""
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda epoch: 1.0 if epoch < milestone else lr_decay)


def _interaction_heads(module: nn.Module):
    from .adamixer_transH_spatial_r50_head import InteractionHead
    return [m for m in module.modules() if isinstance(m, InteractionHead)]


def wrap_ddp(module: nn.Module, device=None):
    """utils.py:202-205 (pocket's engine wraps the net in DDP with find_unused_parameters=True).  A single process needs
    no gradient hooks: the head's fused step then writes p.grad directly (grad_mode "direct", ~1 ms of autograd
    bookkeeping per step saved); under DDP the gradients go through the autograd engine, whose hooks DDP listens to."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        for h in _interaction_heads(module):
            h.grad_mode = "direct"
        return module
    for h in _interaction_heads(module):
        h.grad_mode = "autograd"
    ids = [device.index] if device is not None and device.type == "cuda" else None
    return nn.parallel.DistributedDataParallel(module, device_ids=ids, find_unused_parameters=True)


def train_step(net, optimizer, *inputs, targets):
    """utils.py:213-229: zero_grad -> forward -> sum of the loss dict -> backward -> step.  Returns the loss dict
    (detached floats) and the per-image results."""
    optimizer.zero_grad(set_to_none=True)
    out = net(*inputs, targets)
    loss_dict = out.pop()
    if torch.isnan(loss_dict["hoi_loss"]):
        raise ValueError(f"The HOI loss is NaN")
    total = sum(loss for loss in loss_dict.values())
    total.backward()
    optimizer.step()
    return {k: float(v.detach()) for k, v in loss_dict.items()}, out
