"""torchrun debug: which part of the data-parallel step differs from the single-process mean-gradient step at world > 2?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import torch, torch.distributed as dist
world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("CRNN_DIST_BACKEND", "gloo"))
from bench import synthetic_batch
from crnn_mi355x.engine import Engine
from crnn_mi355x.init import initial_parameters
from crnn_mi355x.parallel import GradAllReduce
B = 8
kw = dict(imgh=40, max_len=6, time_dense_size=32, n_units=64, dropout=False, precision="fp32")
eng = Engine(B, **kw)
eng.set_params(initial_parameters(eng.layout, 64, False, seed=100))
x, lab, il, ll = synthetic_batch(B, seed=rank, imgh=40, max_len=6, T=eng.T)
eng.forward(x, train=True, seed=0); eng.backward(lab, il, ll, seed=0)
torch.cuda.synchronize()
g = eng.grads.clone()
parts = [torch.empty_like(g) for _ in range(world)]
dist.all_gather(parts, g)
ref = sum(p.double() for p in parts)
a = g.clone(); dist.all_reduce(a); torch.cuda.synchronize()
print(rank, "blocking all_reduce vs gathered sum: max rel", float(((a.double() - ref).abs().max()) / ref.abs().max()), flush=True)
# the engine's own overlapped path
eng.forward(x, train=True, seed=0)
ar = GradAllReduce(eng, dist, world)
eng.backward_top(lab, il, ll, seed=0)
split = eng.grad_split
ar.start(eng.grads[split:]); eng.backward_bottom(seed=0); ar.start(eng.grads[:split]); ar.finish(eng.grads)
torch.cuda.synchronize()
b = eng.grads.double() * world
print(rank, "overlapped path vs gathered sum: max rel", float((b - ref).abs().max() / ref.abs().max()), "tail", float((b[split:] - ref[split:]).abs().max() / ref.abs().max()),
      "head", float((b[:split] - ref[:split]).abs().max() / ref.abs().max()), flush=True)
dist.barrier(); dist.destroy_process_group()
