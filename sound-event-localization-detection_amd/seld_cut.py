"""Cut points of the backward pass, for the data-parallel captured step (seld_graph.py).

The epoch loop upstream (trainer.py:165-179) is one ``loss.backward()``; to put the gradient all-reduce of the layers
whose gradients are final UNDER the rest of the backward pass, the captured iteration is cut into stages at the points a
model marks with ``boundary(x)``:

    stage 0   forward, loss, backward from the loss down to the LAST boundary
    stage k   backward from the k-th boundary (counted from the end) down to the one before it

``boundary`` is the identity unless a ``recording`` is active (only ``seld_graph.GraphedTrainStep`` opens one): then it
returns a detached leaf in place of ``x`` -- autograd stops there -- and remembers the pair; a later stage continues
with ``torch.autograd.backward(x, leaf.grad)``.  Same kernels on the same numbers in the same order as the uncut
backward pass, so the cut changes no result (tests/test_graph_gpu.py, tests/test_ddp_gpu.py)."""
import os

import torch

_active = None
# how fine the cut is: 1 = only the models' primary cut (between the recurrent / attention part and the encoder),
# 2 = also the secondary ones (before the encoder's last block / layer4).  Config.ALLREDUCE_CUT_LEVELS via the trainer;
# SELD_CUT_LEVELS overrides (developer A/B).
levels = int(os.environ.get("SELD_CUT_LEVELS", "2"))


class recording:
    """``with recording() as rec:`` around a forward pass; ``rec.cuts`` = [(outer, leaf), ...] in forward order."""

    def __enter__(self):
        global _active
        self.outer, self.cuts = _active, []
        _active = self.cuts
        return self

    def __exit__(self, *exc):
        global _active
        _active = self.outer
        return False


def boundary(x, level=1):
    if _active is None or level > levels or not torch.is_grad_enabled() or not x.requires_grad:
        return x
    leaf = x.detach().requires_grad_(True)
    _active.append((x, leaf))
    return leaf
