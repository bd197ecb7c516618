import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import torch, torch.distributed as dist
world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from bench import synthetic_batch
from crnn_mi355x.engine import Engine
from crnn_mi355x import native
from crnn_mi355x.init import initial_parameters
from crnn_mi355x.optimizers import Adam
from crnn_mi355x.parallel import GradAllReduce, broadcast_state
B, steps = 8, 3
kw = dict(imgh=40, max_len=6, time_dense_size=32, n_units=64, dropout=False, precision="fp32", flags=native.FLAG_RNN_STEP_KERNELS)
eng = Engine(B, **kw)
eng.set_params(initial_parameters(eng.layout, 64, False, seed=100 + rank))
broadcast_state(eng, dist, world)
start = eng.params.clone()
opt = Adam(lr=1e-3, beta_1=0.5, beta_2=0.999, clipnorm=5)
ar = GradAllReduce(eng, dist, world, overlap=os.environ.get("OVL", "1") == "1")
shards = [synthetic_batch(B, seed=r, imgh=40, max_len=6, T=eng.T) for r in range(world)]
x, lab, il, ll = shards[rank]
snaps = []
for it in range(steps):
    eng.train_step(x, lab, il, ll, opt, it, allreduce=ar)
    torch.cuda.synchronize()
    snaps.append((eng.params.clone(), eng.grads.clone(), eng.norm.clone()))
if rank == 0:
    ref = Engine(B, **kw); ref.params.copy_(start)
    opt2 = Adam(lr=1e-3, beta_1=0.5, beta_2=0.999, clipnorm=5)
    for it in range(steps):
        acc = torch.zeros_like(ref.grads)
        for r in range(world):
            xs, ls, ils, lls = shards[r]
            ref.forward(xs, train=True, seed=it); ref.backward(ls, ils, lls, seed=it)
            acc += ref.grads
        ref.grads.copy_(acc); ar._scale(ref.grads, 1.0 / world)
        gref = ref.grads.clone()
        opt2.apply(ref, it); torch.cuda.synchronize()
        p, g, nrm = snaps[it]
        print("step", it, "grad max rel diff", float((g - gref).abs().max() / gref.abs().max()), "norm", nrm.tolist(), ref.norm.tolist(),
              "param max diff", float((p - ref.params).abs().max()), "mean", float((p - ref.params).abs().mean()), flush=True)
dist.barrier(); dist.destroy_process_group()
