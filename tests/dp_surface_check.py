"""Run under torch.distributed.run (world 2, `CRNN_DIST_BACKEND` = gloo | nccl): the reference-surface Model in a data-parallel
train_on_batch loop after a RANK-CONDITIONAL weight load (`if rank == 0: model.set_weights(...)`).

The out-of-sync mark set_weights leaves is rank-local; the decision to broadcast is taken collectively (surface.Model._replicas_need_sync:
a MAX all-reduce of the marks), so every rank joins the broadcast -- before round 5 rank 0 alone entered it while rank 1 went on to the gradient
all-reduce (mismatched collectives: a hang or corrupted weights).  Checks: the loop finishes, every rank adopted rank 0's loaded weights for the
first step, the replicas are bit-identical afterwards, and a later step issues no further broadcast (the collective check says "in sync").

Prints "DP_SURFACE OK ..." on rank 0.  Used by tests/test_gpu_cli.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    backend = os.environ.get("CRNN_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    if world > max(torch.cuda.device_count(), 1):
        os.environ["CRNN_FLAGS"] = "1"          # ranks share a GPU: per-step recurrence kernels (see tests/dp_check.py)
    os.environ["CRNN_PRECISION"] = "fp32"
    import utils as U
    from crnn_mi355x import parallel
    from bench import synthetic_batch

    B = 8
    model = U.CRNN(num_classes=38, shape=(40, 32, 1), GRU=False, time_dense_size=32, n_units=64, max_string_len=6).get_model()
    model.compile(loss={"ctc": lambda y_true, y_pred: y_pred}, optimizer=U.optimizers.Adam(lr=1e-3, beta_1=0.5, beta_2=0.999, clipnorm=5))
    T = int(model._engine(B).T)
    x, lab, il, ll = synthetic_batch(B, seed=rank, imgh=40, max_len=6, T=T)
    # every rank trains one step first (fresh replicas: the first step's collective check makes all of them adopt rank 0's initial weights)
    model.train_on_batch(x, lab, il, ll)
    calls = {"n": 0}
    real = parallel.broadcast_state

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    parallel.broadcast_state = counting
    # rank-conditional load: only rank 0's replica changes (and only rank 0's out-of-sync mark is set)
    loaded = None
    if rank == 0:
        ws = model.get_weights()
        rs = np.random.RandomState(5)
        loaded = [w + rs.normal(size=w.shape).astype(w.dtype) * 0.01 for w in ws]
        model.set_weights(loaded)
    model.train_on_batch(x, lab, il, ll)      # must not hang: all ranks broadcast (rank 0's loaded weights), then all-reduce
    assert calls["n"] == 1, "rank %d: %d broadcasts after a rank-conditional set_weights (expected 1 on every rank)" % (rank, calls["n"])
    model.train_on_batch(x, lab, il, ll)      # in sync now: no further broadcast anywhere
    assert calls["n"] == 1, "rank %d: a step of in-sync replicas issued a broadcast" % rank
    eng = model._state["engine"]
    torch.cuda.synchronize()
    parts = [torch.empty_like(eng.params) for _ in range(world)]
    dist.all_gather(parts, eng.params.contiguous())
    for r in range(1, world):
        assert torch.equal(parts[0], parts[r]), "replicas differ between rank 0 and rank %d after the rank-conditional load" % r
    if rank == 0:
        print("DP_SURFACE OK world=%d backend=%s broadcasts=%d" % (world, backend, calls["n"]), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
