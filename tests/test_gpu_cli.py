"""-m gpu end-to-end test of the drop-in surface: the reference's train.py / predict.py command lines (same flags)
on a small synthetic image folder -- Readf generator, CRNN(...).get_model(), compile, fit_generator with
ModelCheckpoint + EarlyStoppingIter, artefact files, load_custom_model, init_predictor, predict_generator,
DecodeCTCPred (HIP beam search), edit-distance report and prediction.csv."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

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
    mj = json.load(open(mdir / "model.json"))       # a Keras-2.2.2 functional-model JSON (crnn_mi355x.keras_json)
    assert mj["class_name"] == "Model" and mj["keras_version"] == "2.2.2" and mj["config"]["output_layers"] == [["ctc", 0, 0]]
    max_len = [l for l in mj["config"]["layers"] if l["name"] == "the_labels"][0]["config"]["batch_input_shape"][1]
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
                         env=env, capture_output=True, text=True, timeout=300)
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
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                     # rank 0 only
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 32 and res["config"]["parallelism"] == "dp2"
    assert res["scaling"] == "weak" and res["value"] > 0 and np.isfinite(res["config"]["final_loss"])
    # the N-rank line proves itself: the process group's own world size, bit-identical replicas after the timed steps, exposed exchange time
    dp = res["data_parallel"]
    assert dp["dist_world_size"] == 2 and dp["ranks_reporting"] == 2 and dp["dist_backend"] == "gloo"
    assert dp["replicas_identical"] is True and dp["param_checksum_max_abs_diff_across_ranks"] == 0.0
    assert np.isfinite(dp["exposed_allreduce_ms"]) and dp["allreduce_bytes_per_step"] > 0
    pf = dp["preflight"]       # two ranks: a + b has one order -- the overlapped exchange must reproduce the blocking one bit for bit
    assert pf["overlapped_schedule_equals_blocking"] is True and pf["max_abs_diff_blocking_vs_overlapped_rel_to_max_gradient"] == 0.0, pf
    assert len(dp["ms_per_step_per_rank"]) == 2


def test_bench_over_rccl_without_enough_gpus_says_so_in_one_line():
    """`bench.py --gpus 2` over RCCL (backend nccl) on a box with one GPU: every rank exits with code 3 before init_process_group and rank 0 prints ONE JSON line naming
    the reason -- not a hang or a stack trace from inside the first collective (round 6)."""
    import json
    import subprocess
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    env = dict(os.environ, CRNN_DIST_BACKEND="nccl", PYTHONPATH=os.pathsep.join([ROOT, PKG]))
    port = 29650 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-roofline", "--no-secondary", "--no-parity", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode != 0 and len(lines) == 1 and "needs 2 GPUs" in json.loads(lines[0])["error"], (out.stdout[-1000:], out.stderr[-1500:])


def _torchrun(script_args, backend, nproc=2, timeout=300):
    import subprocess
    env = dict(os.environ, CRNN_DIST_BACKEND=backend, PYTHONPATH=os.pathsep.join([ROOT, PKG]), HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29900 + os.getpid() % 300 + (7 if backend == "nccl" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_data_parallel_step_equals_single_process_step_gloo():
    """Two ranks (sharing this box's GPU, gradients over gloo): ranks that start from different weights end bit-identical
    (broadcast_state), and equal to one process applying Adam to the mean of the two shard gradients (tests/dp_check.py)."""
    out = _torchrun([os.path.join(ROOT, "tests", "dp_check.py")], "gloo")
    assert out.returncode == 0 and "DP_CHECK OK world=2" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_data_parallel_step_equals_single_process_step_gloo_world4():
    """Four ranks on this box's GPU over gloo: replicas bit-identical, and equal (to fp32 summation order of the four-term all-reduce)
    to one process applying Adam to the mean of the four shard gradients -- half of BASELINE configs[3]'s world size, the largest a
    one-GPU box runs comfortably."""
    out = _torchrun([os.path.join(ROOT, "tests", "dp_check.py")], "gloo", nproc=4, timeout=300)
    if out.returncode != 0:      # keep the whole log where the round's evidence script collects it
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "dp_world4_failure.log"), "w") as f:
                f.write(out.stdout + "\n==== stderr ====\n" + out.stderr)
        except OSError:
            pass
    err = "\n".join(l for l in out.stderr.splitlines() if "Gloo" not in l and "amdgpu.ids" not in l)
    assert out.returncode == 0 and "DP_CHECK OK world=4" in out.stdout, (out.stdout[-1500:], err[-3000:])


def _dp_log(name, out):
    try:      # keep the whole log where the round's evidence script collects it
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", name), "w") as f:
            f.write(out.stdout + "\n==== stderr ====\n" + out.stderr)
    except OSError:
        pass


def test_data_parallel_step_equals_single_process_step_gloo_world8():
    """BASELINE configs[3]'s rank count -- eight ranks, here on this box's one GPU over gloo: shard seeds 0..7, broadcast from rank 0, the two-bucket
    all-reduce split at crnn_grad_split_offset, replicas bit-identical after three steps and equal (to the summation order of the eight-term
    all-reduce) to one process applying Adam to the mean of the eight shard gradients.  The first execution of world size 8 on any backend."""
    out = _torchrun([os.path.join(ROOT, "tests", "dp_check.py")], "gloo", nproc=8, timeout=600)
    if out.returncode != 0:
        _dp_log("dp_world8_failure.log", out)
    err = "\n".join(l for l in out.stderr.splitlines() if "Gloo" not in l and "amdgpu.ids" not in l)
    assert out.returncode == 0 and "DP_CHECK OK world=8" in out.stdout, (out.stdout[-1500:], err[-3000:])


def test_bench_eight_ranks_on_one_gpu_over_gloo():
    """`bench.py --gpus 8 --batch 8` as the driver launches the 8-GPU line (torch.distributed.run, one process per rank), the eight ranks sharing
    this box's GPU over gloo: the JSON line's `data_parallel` self-proof at configs[3]'s world size (world size from the process group, eight
    ranks reporting, bit-identical replicas after the timed steps) and the transport record."""
    import json
    import subprocess
    env = dict(os.environ, CRNN_DIST_BACKEND="gloo", CRNN_FLAGS="1", PYTHONPATH=os.pathsep.join([ROOT, PKG]), OMP_NUM_THREADS="2")
    port = 29300 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--batch", "8",
           "--no-roofline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    if out.returncode != 0:
        _dp_log("bench_world8_failure.log", out)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["config"]["global_batch"] == 64 and res["config"]["parallelism"] == "dp8" and res["scaling"] == "weak"
    dp = res["data_parallel"]
    assert dp["dist_world_size"] == 8 and dp["ranks_reporting"] == 8 and dp["dist_backend"] == "gloo"
    assert dp["replicas_identical"] is True and dp["param_checksum_max_abs_diff_across_ranks"] == 0.0
    # round 6: what makes the first RCCL run unable to fail silently -- the pre-flight comparison of the blocking and the overlapped exchange, every rank's own
    # clock, the GPUs the ranks saw
    pf = dp["preflight"]
    assert pf["overlapped_schedule_equals_blocking"] is True and pf["gradient_checksum_max_abs_diff_across_ranks"] == 0.0, pf
    assert len(dp["ms_per_step_per_rank"]) == 8 and dp["ms_per_step_max_over_ranks"] == max(dp["ms_per_step_per_rank"]) and dp["visible_gpus"] >= 1
    assert abs(dp["ms_per_step_max_over_ranks"] - res["ms_per_step"]) < 0.01 * res["ms_per_step"] + 0.01
    assert dp["transport"]["backend"] == "gloo" and "env" in dp["transport"] and "xgmi_links_reported" in dp["transport"]


def test_rank_conditional_set_weights_then_train_on_batch_does_not_mismatch_collectives():
    """`if rank == 0: model.set_weights(...)` followed by a data-parallel train_on_batch on every rank.  The decision to
    broadcast is collective (tests/dp_surface_check.py): the loop finishes, exactly one broadcast on every rank, replicas bit-identical."""
    out = _torchrun([os.path.join(ROOT, "tests", "dp_surface_check.py")], "gloo", nproc=2, timeout=300)
    if out.returncode != 0:
        _dp_log("dp_surface_failure.log", out)
    assert out.returncode == 0 and "DP_SURFACE OK world=2" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_data_parallel_step_equals_single_process_step_rccl():
    """The same over RCCL (backend "nccl"), one rank per GPU -- needs >= 2 visible GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        why = ("RCCL data-parallel check skipped: torch.cuda.device_count() = %d on this box (HIP_VISIBLE_DEVICES=%r, ROCR_VISIBLE_DEVICES=%r); "
               "BASELINE configs[3] needs an 8-GPU node" % (torch.cuda.device_count(), os.environ.get("HIP_VISIBLE_DEVICES"),
                                                            os.environ.get("ROCR_VISIBLE_DEVICES")))
        try:       # leave the reason where the round's evidence script collects it (profiles/rNN_summary.txt)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "rccl_skip_reason.txt"), "w") as f:
                f.write(why + "\n")
        except OSError:
            pass
        pytest.skip(why)
    out = _torchrun([os.path.join(ROOT, "tests", "dp_check.py")], "nccl")
    assert out.returncode == 0 and "DP_CHECK OK world=2 backend=nccl" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls --gpus 1) must itself start 2 ranks and print
    ONE JSON line with n_gpus 2: over RCCL when 2 GPUs are visible, else both ranks on this GPU over gloo."""
    import json
    import subprocess
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    env = dict(os.environ, CRNN_DIST_BACKEND=backend, PYTHONPATH=os.pathsep.join([ROOT, PKG]))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "16",
                          "--no-roofline"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 32 and res["value"] > 0


def test_train_cli_two_ranks_keep_identical_weights(tmp_path):
    """torchrun train.py (2 ranks, gloo on this GPU): unequal file counts per rank are trimmed to equal shards, early stopping
    monitors the global loss, so both ranks finish (no hang) and rank 0 writes the artefacts."""
    data = tmp_path / "data"; out = tmp_path / "out"
    os.makedirs(data); os.makedirs(out)
    _make_dataset(str(data), n=51)
    res = _torchrun([os.path.join(PKG, "train.py"), "--path", str(data), "--save_path", str(out), "--model_name", "dp", "--nbepochs", "2",
                     "--norm", "--opt", "adam", "--lr", "0.001", "--batch_size", "8", "--n_units", "64", "--time_dense_size", "32",
                     "--early_stopping", "2", "--G", "0"], "gloo")
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    for f in ("model.json", "final_weights.h5", "loss_history.pickle.dat"):
        assert (out / "dp" / f).exists(), f
