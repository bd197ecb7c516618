import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np, torch
import bench
from crnn_mi355x.engine import Engine
from crnn_mi355x.init import initial_parameters
from crnn_mi355x.optimizers import Adam
for flags in (0, 1):
    for B in (64, 256):
        eng = Engine(B, dropout=True, precision="bf16s", flags=flags)
        eng.set_params(initial_parameters(eng.layout, 256, False, seed=1))
        batch = tuple(torch.from_numpy(a if i == 0 else a.astype(np.int32)).cuda() for i, a in enumerate(bench.synthetic_batch(B, 0)))
        opt = Adam(lr=1e-4, beta_1=0.5, beta_2=0.999, clipnorm=5)
        res = []
        for rep in range(6):
            dt, _ = bench.timed_steps(eng, batch, opt, 20, 2 if rep == 0 else 0, it0=rep * 30)
            res.append(round(dt / 20 * 1e3, 2))
        lr = bench.lstm_roofline(eng, iters=5) if flags == 0 else None
        print("flags", flags, "B", B, "ms/step per repetition:", res, "lstm ms:", lr and lr["ms_per_train_step"], flush=True)
        del eng; torch.cuda.empty_cache()
