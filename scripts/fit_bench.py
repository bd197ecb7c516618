#!/usr/bin/env python3
"""Host-inclusive training rate through the reference's own surface: Model.fit_generator fed by a Readf-style generator
that yields host NumPy batches (float64 images as `get_blank_matrices` makes them, the same arrays re-yielded), so the
number includes the float64->float32 conversion, the H->D copy over PCIe, the per-step loss read-back and the callbacks'
Python.  usage: fit_bench.py [--batch 256] [--steps 40] [--precision bf16s]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256); ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--precision", default="bf16s")
args = ap.parse_args()
os.environ["CRNN_PRECISION"] = args.precision
import utils as U
from bench import synthetic_batch

B = args.batch
x, lab, il, ll = synthetic_batch(B, seed=0)
X = x.astype(np.float64)                                   # Readf batches are float64 (utils.py:446-452)
inputs = {"the_input": X, "the_labels": lab.astype(np.int64), "input_length": il.reshape(-1, 1).astype(np.int64),
          "label_length": ll.reshape(-1, 1).astype(np.int64), "source_str": np.array(["x"] * B)}
outputs = {"ctc": np.zeros([B])}

def gen():
    while True:
        np.add(X, 0.0, out=X)                               # the generator owns and rewrites these arrays between yields
        yield inputs, outputs

init_model = U.CRNN(num_classes=38, shape=(100, 32, 1), GRU=False, time_dense_size=128, n_units=256, max_string_len=23)
model = init_model.get_model()
model.compile(loss={"ctc": lambda y_true, y_pred: y_pred}, optimizer=U.optimizers.Adam(lr=1e-4, beta_1=0.5, beta_2=0.999, clipnorm=5))
g = gen()
model.fit_generator(g, steps_per_epoch=5, epochs=1, verbose=0)          # warm-up (engine creation, first launches)
t0 = time.perf_counter()
H = model.fit_generator(g, steps_per_epoch=args.steps, epochs=1, verbose=0)
dt = time.perf_counter() - t0
print(json.dumps({"workload": "Model.fit_generator over host NumPy batches (float64, batch %d), %s" % (B, args.precision),
                  "steps": args.steps, "ms_per_step": round(1e3 * dt / args.steps, 3), "images_per_sec": round(B * args.steps / dt, 1),
                  "loss": round(H.history["loss"][-1], 4)}))
