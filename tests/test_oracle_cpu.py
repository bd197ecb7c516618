"""CPU tests of the oracle itself: pinned against the reference-generated golden vectors
(tests/golden/, produced by tests/golden/make_golden.py from /root/reference/utils.py) and
cross-checked against an independent torch-autograd mirror.  No GPU needed."""
import itertools
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ops, ctc, model as M
import torch_mirror as TM

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_sampler_matches_reference_golden_bitexact():
    z = np.load(os.path.join(GOLD, "sampler_golden.npz"))
    for k in ("mj", "iam", "small"):
        for tn in ("ident", "pert", "wild"):
            out = ops.sampler_fwd(z[f"{k}_{tn}_img"], z[f"{k}_{tn}_theta"])
            assert out.dtype == np.float32
            np.testing.assert_array_equal(out, z[f"{k}_{tn}_out"])


def test_param_counts_match_model_summary():
    # models/*/model_summary.txt:156-158 (GRU variant as shipped) and SURVEY 8a (LSTM variant)
    assert M.Config(gru=True).n_trainable() == 2823089
    assert M.Config(gru=False).n_trainable() == 3282865
    bn_stats = 2 * sum(c for _, c in M.Config().bn_shapes())
    assert bn_stats == 7938
    assert M.Config(gru=True).n_trainable() + bn_stats == 2831027
    assert M.Config(imgh=200).stn_flat == 1760 and M.Config(imgh=200).T == 102


def _tiny(gru):
    cfg = M.Config(imgh=36, imgw=32, num_classes=7, max_len=4, time_dense_size=12, n_units=8, gru=gru)
    p, bn = M.init_params(cfg, seed=5, dtype=np.float64)
    p = M.randomize_params(cfg, p)
    x, lab, il, ll = M.synthetic_batch(cfg, 3, seed=2, dtype=np.float64)
    rs = np.random.RandomState(9)
    masks = {}
    y, c = M.forward(cfg, p, bn, x, train=True)
    for i in range(1, 8):
        pool = M.BLOCKS[i - 1][1]
        sh = c[f"r{i}"].shape if not pool else ops.maxpool_fwd(c[f"r{i}"], *pool).shape
        masks[f"b{i}"] = (rs.uniform(size=sh) > 0.1).astype(np.float64)
    masks["dense1"] = (rs.uniform(size=c["dense1"].shape) > 0.4).astype(np.float64)
    masks["rnn"] = (rs.uniform(size=c["rnn_out"].shape) > 0.2).astype(np.float64)
    return cfg, p, bn, x, lab, il, ll, masks


@pytest.mark.parametrize("gru", [False, True])
def test_full_model_grads_match_torch_autograd(gru):
    cfg, p, bn, x, lab, il, ll, masks = _tiny(gru)
    loss, loss_b, g, c = M.loss_and_grads(cfg, p, bn, x, lab, il, ll, masks=masks)
    P = {k: TM.t(v, grad=True) for k, v in p.items()}
    yp = TM.forward(cfg, P, TM.t(x), masks=masks)
    np.testing.assert_allclose(yp.detach().numpy(), c["y_pred"], rtol=1e-9, atol=1e-11)
    lb = TM.ctc_cost(yp, lab, il, ll)
    np.testing.assert_allclose(lb.detach().numpy(), loss_b, rtol=1e-8, atol=1e-9)
    lb.mean().backward()
    for k in p:
        ref = P[k].grad.numpy()
        scale = max(1e-12, np.abs(ref).max())
        assert np.abs(g[k] - ref).max() <= 1e-7 * scale + 1e-12, k


def test_ctc_matches_torch_and_handles_repeats_and_short_inputs():
    rs = np.random.RandomState(0)
    B, T, C = 5, 14, 6
    y = ops.softmax_fwd(rs.normal(size=(B, T, C)) * 2)
    labels = np.array([[1, 1, 2, 5, 5], [0, 5, 5, 5, 5], [3, 3, 3, 5, 5], [4, 2, 4, 2, 5], [1, 5, 5, 5, 5]])
    ll = np.array([3, 1, 3, 4, 1])
    il = np.array([12, 12, 7, 9, 1])
    loss, gy = ctc.ctc_loss_and_grad(y, labels, il, ll)
    yt = TM.t(y, grad=True)
    lt = TM.ctc_cost(yt, labels, il, ll)
    np.testing.assert_allclose(loss, lt.detach().numpy(), rtol=1e-9)
    lt.sum().backward()
    np.testing.assert_allclose(gy, yt.grad.numpy(), rtol=1e-7, atol=1e-9)
    assert np.all(gy[:, :2] == 0) and np.all(gy[2, 2 + 7:] == 0)
    # epsilon / re-normalisation is visible at the 1e-3 level (SURVEY A.7 sanity note)
    y50 = ops.softmax_fwd(rs.normal(size=(1, 52, 38)) * 3).astype(np.float32)
    l1, _ = ctc.ctc_loss_and_grad(y50, np.array([[1, 2, 3]]), [50], [3])
    assert np.isfinite(l1).all()


def test_ctc_impossible_label_gives_inf_and_zero_grad():
    y = np.full((1, 5, 4), 0.25)
    loss, gy = ctc.ctc_loss_and_grad(y, np.array([[1, 1, 1]]), [3], [3])  # needs 5 steps, has 3
    assert np.isinf(loss[0]) and np.all(gy == 0)


def _brute_force_best(p):
    T, C = p.shape
    blank = C - 1
    tot = {}
    for path in itertools.product(range(C), repeat=T):
        pr = np.prod([p[t, k] for t, k in enumerate(path)])
        out, prev = [], -1
        for k in path:
            if k != blank and k != prev:
                out.append(k)
            prev = k
        tot[tuple(out)] = tot.get(tuple(out), 0.0) + pr
    return max(tot.items(), key=lambda kv: kv[1])


def test_beam_exhaustive_equals_brute_force():
    rs = np.random.RandomState(3)
    for _ in range(25):
        p = ops.softmax_fwd(rs.normal(size=(1, 5, 4)) * 1.5)
        out, lens, _ = ctc.ctc_beam_decode(p, beam_width=1000, merge_repeated=False)
        best, _ = _brute_force_best(p[0] + 0)  # eps 1e-7 is negligible here
        assert tuple(out[0, :lens[0]]) == best


def _peaked(text, classes, C=38, T=52, hi=0.97):
    seq = []
    prev = None
    for ch in text:
        k = classes[ch]
        if prev == k:
            seq.append(C - 1)
        seq += [k, k]
        prev = k
    seq = seq[:T] + [C - 1] * (T - len(seq))
    p = np.full((T, C), (1 - hi) / (C - 1), dtype=np.float32)
    for t_, k in enumerate(seq):
        p[t_, k] = hi
    return p


def test_beam_known_answers_from_reference_screenshots():
    g = json.load(open(os.path.join(GOLD, "helpers_golden.json")))
    classes = {ch: i for i, ch in enumerate(g["lexicon"])}
    inv = {v: k for k, v in classes.items()}
    for truth, ref_pred in g["beam_known_answers"]:
        p = _peaked(truth, classes)[None]
        out, lens, _ = ctc.ctc_beam_decode(p, beam_width=10, merge_repeated=True)
        assert ctc.labels_to_text(out[0], inv) == ref_pred  # reference artefacts: "cellist"->"celist"
        out2, lens2, _ = ctc.ctc_beam_decode(p, beam_width=10, merge_repeated=False)
        assert ctc.labels_to_text(out2[0], inv) == truth
        gd, gl = ctc.ctc_greedy_decode(p)
        assert ctc.labels_to_text(gd[0], inv) == truth


def test_greedy_ties_and_padding():
    p = np.zeros((1, 4, 3)); p[0, :, :] = [[.5, .5, 0], [.2, .2, .6], [.1, .8, .1], [.1, .8, .1]]
    out, lens = ctc.ctc_greedy_decode(p)
    assert list(out[0]) == [0, 1, -1, -1] and lens[0] == 2


def test_optimizers_one_step():
    rs = np.random.RandomState(0)
    p = {"a": rs.normal(size=(4, 3)), "b": rs.normal(size=5)}
    g = {"a": rs.normal(size=(4, 3)) * 10, "b": rs.normal(size=5) * 10}
    n = np.sqrt((g["a"] ** 2).sum() + (g["b"] ** 2).sum())
    assert n > 5
    p0 = {k: v.copy() for k, v in p.items()}
    M.Adam(lr=1e-2).step(p, g)
    gc = g["a"] * 5 / n
    lr_t = 1e-2 * np.sqrt(1 - .999) / (1 - .5)
    np.testing.assert_allclose(p["a"], p0["a"] - lr_t * (.5 * gc) / (np.sqrt(.001 * gc ** 2) + 1e-7), rtol=1e-12)
    p = {k: v.copy() for k, v in p0.items()}
    M.SGD(lr=1e-2).step(p, g)
    v = -1e-2 * gc
    np.testing.assert_allclose(p["a"], p0["a"] + .9 * v - 1e-2 * gc, rtol=1e-12)


def test_bn_moving_update_formula():
    m, v = ops.bn_moving_update(np.zeros(2), np.ones(2), np.array([1., 2.]), np.array([4., 9.]), 100.0)
    np.testing.assert_allclose(m, [.01, .02])
    np.testing.assert_allclose(v, .99 + .01 * np.array([4., 9.]) * (100 / 99) * (100 / (100 - 1.001)))


def test_oracle_matches_its_committed_snapshot():
    """The oracle checks every GPU parity test; tests/golden/oracle_snapshot.npz (make_oracle_snapshot.py) pins the
    oracle itself: posteriors, losses, gradient norms / leading entries, decodes and an Adam step of two tiny models."""
    sys.path.insert(0, GOLD)
    import make_oracle_snapshot as S
    want = np.load(os.path.join(GOLD, "oracle_snapshot.npz"))
    got = S.snapshot()
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        a, b = np.asarray(got[k]), want[k]
        assert a.shape == b.shape, k
        if b.dtype.kind in "US":
            assert a.tolist() == b.tolist(), k
        elif b.dtype.kind in "iu":
            assert np.array_equal(a, b), k
        else:
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12), (k, float(np.abs(a - b).max()))
