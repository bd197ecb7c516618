"""-m gpu end-to-end test of the drop-in surface: the reference's train.py / predict.py command lines (same flags)
on a small synthetic image folder -- Readf generator, CRNN(...).get_model(), compile, fit_generator with
ModelCheckpoint + EarlyStoppingIter, artefact files, load_custom_model, init_predictor, predict_generator,
DecodeCTCPred (HIP beam search), edit-distance report and prediction.csv."""
import os
import pickle
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "crnn-ocr-lite_amd")


def _make_dataset(folder, n=48, seed=0):
    from PIL import Image, ImageDraw
    rs = np.random.RandomState(seed)
    alphabet = "abcdefghij0123"
    for i in range(n):
        word = "".join(rs.choice(list(alphabet), size=rs.randint(2, 6)))
        img = Image.new("L", (20 + 12 * len(word), 28), color=235)
        ImageDraw.Draw(img).text((4, 6), word, fill=20)
        img.save(os.path.join(folder, "%d_%s_%d.png" % (i, word, i)))


def test_train_then_predict_cli_roundtrip(tmp_path, capsys):
    sys.path.insert(0, PKG)
    import train as train_cli
    import predict as predict_cli
    data = tmp_path / "data"; out = tmp_path / "out"
    os.makedirs(data); os.makedirs(out)
    _make_dataset(str(data))
    np.random.seed(0)
    train_cli.main(["--path", str(data), "--save_path", str(out), "--model_name", "m1", "--nbepochs", "2", "--norm", "--opt", "adam",
                    "--lr", "0.001", "--batch_size", "8", "--n_units", "64", "--time_dense_size", "32", "--early_stopping", "1000",
                    "--G", "0"])
    mdir = out / "m1"
    for f in ("arguments.txt", "model.json", "model_summary.txt", "checkpoint_weights.h5", "final_weights.h5", "final_model.h5",
              "loss_history.pickle.dat"):
        assert (mdir / f).exists(), f
    hist = pickle.load(open(mdir / "loss_history.pickle.dat", "rb"))
    assert len(hist["loss"]) == 2 and len(hist["val_loss"]) == 2 and np.isfinite(hist["loss"]).all() and np.isfinite(hist["val_loss"]).all()
    assert hist["loss"][1] < hist["loss"][0]
    assert "Total params" in open(mdir / "model_summary.txt").read()
    res = tmp_path / "res"; os.makedirs(res)
    # max_len must match the trained model (the reference reads it from the CLI too, predict.py:65)
    import json
    max_len = json.load(open(mdir / "model.json"))["config"]["crnn"]["max_string_len"]
    predict_cli.main(["--model_path", str(mdir), "--image_path", str(data), "--result_path", str(res), "--validate", "--train_portion", "0.5",
                      "--batch_size", "8", "--max_len", str(max_len), "--G", "0"])
    text = capsys.readouterr().out
    assert "mean edit distance" in text and "predictions decoded" in text
    import pandas as pd
    df = pd.read_csv(res / "prediction.csv", dtype=str)       # digit-only predictions must stay text
    assert len(df) == 24 and set(df.columns) >= {"fname", "prediction"}
    lex = set("0123456789abcdefghijklmnopqrstuvwxyz-")
    for p in df["prediction"]:
        assert (isinstance(p, float) and np.isnan(p)) or (isinstance(p, str) and set(p) <= lex), p     # empty decode -> NaN


def test_drop_in_import_order_fresh_process():
    """A user script does `from utils import *` first (as the reference's train.py does) and never imports torch itself:
    the library must still bind to PyTorch's HIP runtime (crnn_mi355x.native.lib imports torch before dlopen) -- run the
    host-inclusive fit benchmark in a fresh interpreter."""
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, PKG]))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fit_bench.py"), "--batch", "8", "--steps", "3", "--precision", "fp32"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["steps"] == 3 and np.isfinite(res["loss"]) and res["images_per_sec"] > 0


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """The N>1 form of bench.py exactly as the driver launches it (torch.distributed.run, one process per rank): two ranks
    share this box's single GPU and exchange gradients over gloo instead of RCCL (CRNN_DIST_BACKEND) -- everything else
    (staged backward, asynchronous bucketed all-reduce, barrier, max-over-ranks timing, rank-0 JSON line) is the real path."""
    import json
    import subprocess
    env = dict(os.environ, CRNN_DIST_BACKEND="gloo", PYTHONPATH=os.pathsep.join([ROOT, PKG]))
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "16",
           "--no-roofline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                     # rank 0 only
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 32 and res["config"]["parallelism"] == "dp2"
    assert res["scaling"] == "weak" and res["value"] > 0 and np.isfinite(res["config"]["final_loss"])
