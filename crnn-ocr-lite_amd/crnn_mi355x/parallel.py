"""Data parallelism for the train step (new work: the reference is single-device).

One process per GPU; the batch dimension is sharded (each rank draws its own images); the only exchange per
step is ONE all-reduce of the flat fp32 gradient buffer (13.1 MB for the LSTM variant) through
torch.distributed -- backend "nccl" = RCCL over xGMI on ROCm -- followed by the identical global-norm clip +
Adam on every rank, so the weights stay bit-identical across ranks.  BatchNorm uses per-replica batch
statistics (what Keras' multi_gpu_model would have done); the moving statistics are averaged on demand
(`sync_bn_stats`, e.g. before a checkpoint).
"""
import torch


def allreduce_mean_(flat, dist, world, scale_fn=None):
    """In-place mean of `flat` over all ranks (sum all-reduce, then x 1/world)."""
    if world <= 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if scale_fn is not None:
        scale_fn(flat, 1.0 / world)
    else:
        flat.mul_(1.0 / world)
    return flat


class GradAllReduce:
    """Handed to Engine.train_step: averages engine.grads across ranks.

    overlap=True (default): two asynchronous all-reduces -- `start(view)` is called once the upper layers' gradients
    are final (the collective then runs on RCCL's stream concurrently with the conv-stack backward that the engine
    keeps launching on the compute stream) and once more for the rest; `finish` joins both and applies 1/world.
    overlap=False: one blocking all-reduce of the whole buffer after backward (`__call__`)."""

    def __init__(self, engine, dist, world, overlap=True):
        self.engine, self.dist, self.world, self.overlap = engine, dist, world, overlap
        self._pending = []

    def _scale(self, t, s):
        from .native import check
        from .engine import _ptr, _stream
        check(self.engine.lib.crnn_scale(_ptr(t), t.numel(), float(s), _stream()), "scale")

    def __call__(self, grads):
        allreduce_mean_(grads, self.dist, self.world, self._scale if grads.is_cuda else None)

    def start(self, view):
        if self.world > 1 and view.numel():
            self._pending.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, async_op=True))

    def finish(self, grads):
        for w in self._pending:
            w.wait()            # the compute stream waits for the collective; the host does not block
        self._pending = []
        if self.world > 1:
            if grads.is_cuda:
                self._scale(grads, 1.0 / self.world)
            else:
                grads.mul_(1.0 / self.world)


def sync_bn_stats(engine, dist, world):
    """Average the BatchNorm moving statistics over ranks.  Policy (SURVEY 8e "document the choice"): batch statistics stay
    per replica inside a step (no SyncBN, what Keras' multi_gpu_model would do); the MOVING statistics -- which only the
    inference / validation forward and the saved weight files read -- are averaged over ranks at the end of every epoch
    (before validation and ModelCheckpoint) and at the end of training (before final_weights.h5), so every rank validates
    with, and rank 0 saves, the same numbers.  A collective: every rank must call it."""
    if world > 1:
        for t in (engine.bn_mean, engine.bn_var):
            allreduce_mean_(t, dist, world)


def broadcast_state(engine, dist, world, src=0):
    """Make every rank start from rank `src`'s weights and BatchNorm moving statistics (a freshly built model draws its
    initial weights from OS entropy per process; identical replicas are what makes `all-reduce + same Adam step` keep
    the weights bit-identical afterwards).  A collective: every rank must call it."""
    if world > 1:
        for t in (engine.params, engine.bn_mean, engine.bn_var):
            dist.broadcast(t, src)


def shard(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for this rank.  Every rank gets the SAME number of items (the remainder
    n_items % world is dropped) so that all ranks run the same number of steps -- and therefore issue the same
    number of collectives -- per epoch."""
    per = n_items // world
    return rank * per, rank * per + per
