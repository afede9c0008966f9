"""Data-parallel training: one process per GPU, gradients summed over RCCL (xGMI inside a node).

The reference's multi-GPU switch wraps nn.DataParallel (train_e2e_spatial.py:500-501): one process, replicas per
iteration, BatchNorm statistics per replica.  Here every rank owns a full replica and 16 clips; clips are independent
units (the gated shift never crosses a clip, the GRU restarts per clip), so the only exchange per step is

  * two scalars known before the forward pass: the sum of CE class weights and the number of event frames over ALL
    ranks (the loss normalisers, train_e2e_spatial.py:569-572,589-591), so that every rank scales its gradient by the
    global denominators and the SUM of rank gradients is exactly the gradient of the global-batch loss;
  * the gradient all-reduce (sum) of the engine's flat fp32 buffer, issued in four buckets as the backward pass
    completes them (heads, s4, s3, rest) so that it overlaps the remaining backward kernels.

BatchNorm batch statistics stay per rank (as under nn.DataParallel); running statistics are rank-local and rank 0's are
the ones saved.  `backend` is any initialised torch.distributed module/group: "nccl" (= RCCL) on GPUs, "gloo" in the
CPU tests of the bucketing logic.
"""
from __future__ import annotations

import torch


def loss_denominators(label, fg_weight=5.0):
    """[sum of CE class weights, number of event frames] of a label tensor (any shape), as float32 on its device."""
    fg = (label != 0).sum().to(torch.float32)
    n = float(label.numel())
    return torch.stack([(n - fg) + fg_weight * fg, fg])


class BucketReducer:
    """Sums ranges of a flat gradient buffer across ranks, asynchronously, and waits for all of them at the end."""

    def __init__(self, dist, flat_grad):
        self.dist, self.flat, self.pending = dist, flat_grad, []

    def __call__(self, lo, hi):
        if hi > lo:
            self.pending.append(self.dist.all_reduce(self.flat[lo:hi], op=self.dist.ReduceOp.SUM, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []


class DataParallelTrainer:
    def __init__(self, model, dist=None, lr=1e-3, fg_weight=5.0):
        self.model, self.dist, self.lr, self.fg_weight = model, dist, lr, fg_weight
        self.eng = model._model.engine
        if dist is not None:
            dist.broadcast(self.eng.flat, src=0)          # identical replicas
            for b in model._model.buffers():
                if b.dtype.is_floating_point:
                    dist.broadcast(b, src=0)
            self.eng.weights_dirty = True
        self.reducer = BucketReducer(dist, self.eng.flat_grad) if dist is not None else None

    def step(self, frame, label, xy):
        den = None
        if self.dist is not None:
            den = loss_denominators(label, self.fg_weight)
            self.dist.all_reduce(den, op=self.dist.ReduceOp.SUM)
        out = self.model.train_step(frame, label, xy, self.lr, fg_weight=self.fg_weight, ext_denoms=den,
                                    on_ready=self.reducer, after_backward=(lambda eng: self.reducer.wait()) if self.reducer else None)
        return out
