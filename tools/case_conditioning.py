"""Developer aid (CPU): is a training case WELL CONDITIONED as a gradient fixture?

A gradient fixture made on one CPU is replayed on another (the GPU box's host, the GPU): matrix products there round
differently in the last bit, and a ReLU input that sits within a few ulp of zero switches its unit on or off -- the loss does
not move, but that unit's whole contribution to the gradients of its layer and of everything upstream does (seen on
train_vcoco's first seed: attention_head.fc_2.5.bias 4e-3 away from the fixture on the GPU box, oracle and HIP alike, with
bit-identical losses).  This script perturbs the pooled box features by a few ulp (x (1 +- 3e-7)) and reports the largest
relative gradient change per tensor under the oracle's autograd: a case whose gradients move by more than ~2e-5 has such a
unit and should get another image seed.   usage: case_conditioning.py <case> [<case> ...]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import cases, helpers


def grads_with_gain(name, gain):
    case = cases.build_case(name)
    orig = cases.pooled_for
    cases.pooled_for = lambda c, n: orig(c, n) * gain
    try:
        return helpers.oracle_train_grads(case)[0]
    finally:
        cases.pooled_for = orig


def worst_change(name):
    base = grads_with_gain(name, 1.0)
    worst = (0.0, "")
    for gain in (1.0 + 3e-7, 1.0 - 3e-7):
        g = grads_with_gain(name, np.float32(gain))
        for k, w in base.items():
            if k == "box_pair_head.adjacency.bias":
                continue
            scale = max(float(np.abs(w).max()), 1e-9)
            worst = max(worst, (float(np.abs(g[k] - w).max()) / scale, k))
    return worst


if __name__ == "__main__":
    for name in sys.argv[1:]:
        print(name, "largest relative gradient change under a +-3e-7 input gain: %.2e (%s)" % worst_change(name))
