"""torch-CPU restatement of the CRNN graph (Keras 2.2.2 / TF 1.8 semantics) -- TEST / BASELINE INFRASTRUCTURE.

Two consumers, both outside the product: (1) tests/ use it in float64 as an independent autograd cross-check of the hand-written
backward of oracle/ (tests/torch_mirror.py re-exports it); (2) bench.py's `cpu_baseline` leg times `train_step_benchmark` in
float32 on the host cores -- the closest runnable stand-in for the reference's Keras-TF CPU path (train.py:201 with --G 0;
predict.py:88-93 pins 4 intra-/inter-op threads), which cannot run here (no TensorFlow / Keras in the image).  Where torch's
built-in layers have different semantics from Keras 2.2.2 (grid_sample, nn.LSTM/GRU, Adam) the op is written out with elementary
torch ops so that autograd differentiates the *restated* semantics.  Nothing under crnn-ocr-lite_amd/ imports this module."""
import numpy as np
import torch
import torch.nn.functional as F

from . import model as M


def t(a, grad=False):
    x = torch.tensor(np.asarray(a), dtype=torch.float64)
    x.requires_grad_(grad)
    return x


def sampler(image, theta):
    B, H, W, C = image.shape
    xs = torch.linspace(-1, 1, W, dtype=image.dtype)
    ys = torch.linspace(-1, 1, H, dtype=image.dtype)
    yg, xg = torch.meshgrid(ys, xs, indexing="ij")
    G = torch.stack([xg.reshape(-1), yg.reshape(-1), torch.ones(H * W, dtype=image.dtype)], 0)
    S = theta.reshape(B, 2, 3) @ G
    x = 0.5 * (S[:, 0] + 1.0) * W
    y = 0.5 * (S[:, 1] + 1.0) * H
    x0 = torch.trunc(x.detach()).long(); y0 = torch.trunc(y.detach()).long()
    x1 = x0 + 1; y1 = y0 + 1
    x0 = x0.clamp(0, W - 1); x1 = x1.clamp(0, W - 1); y0 = y0.clamp(0, H - 1); y1 = y1.clamp(0, H - 1)
    bi = torch.arange(B)[:, None]
    Pa = image[bi, y0, x0]; Pb = image[bi, y1, x0]; Pc = image[bi, y0, x1]; Pd = image[bi, y1, x1]
    x0f, x1f, y0f, y1f = (a.to(image.dtype) for a in (x0, x1, y0, y1))
    wa = ((x1f - x) * (y1f - y))[..., None]; wb = ((x1f - x) * (y - y0f))[..., None]
    wc = ((x - x0f) * (y1f - y))[..., None]; wd = ((x - x0f) * (y - y0f))[..., None]
    return (wa * Pa + wb * Pb + wc * Pc + wd * Pd).reshape(B, H, W, C)


def conv_nhwc(x, k, b=None, groups=1, padding=0):
    w = k.permute(3, 2, 0, 1)  # HWIO -> OIHW
    y = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=padding, groups=groups)
    return y.permute(0, 2, 3, 1)


def maxpool(x, ph, pw):
    return F.max_pool2d(x.permute(0, 3, 1, 2), (ph, pw)).permute(0, 2, 3, 1)


def bn_train(x, g, b):
    m = x.mean(dim=(0, 1, 2))
    v = ((x - m) ** 2).mean(dim=(0, 1, 2))
    return (x - m) / torch.sqrt(v + 1e-3) * g + b


def hs(z):
    return torch.clamp(0.2 * z + 0.5, 0, 1)


def lstm(x, W, U, b, reverse):
    B, T, _ = x.shape
    u = U.shape[0]
    xW = x @ W + b
    h = torch.zeros(B, u, dtype=x.dtype); c = torch.zeros(B, u, dtype=x.dtype)
    out = [None] * T
    for tt in (range(T - 1, -1, -1) if reverse else range(T)):
        z = xW[:, tt] + h @ U
        i, f, g, o = hs(z[:, :u]), hs(z[:, u:2 * u]), torch.tanh(z[:, 2 * u:3 * u]), hs(z[:, 3 * u:])
        c = f * c + i * g
        h = o * torch.tanh(c)
        out[tt] = h
    return torch.stack(out, 1)


def gru(x, W, U, b, reverse):
    B, T, _ = x.shape
    u = U.shape[0]
    xW = x @ W + b
    h = torch.zeros(B, u, dtype=x.dtype)
    out = [None] * T
    for tt in (range(T - 1, -1, -1) if reverse else range(T)):
        zr = xW[:, tt, :2 * u] + h @ U[:, :2 * u]
        z, r = hs(zr[:, :u]), hs(zr[:, u:])
        hh = torch.tanh(xW[:, tt, 2 * u:] + (r * h) @ U[:, 2 * u:])
        h = z * h + (1 - z) * hh
        out[tt] = h
    return torch.stack(out, 1)


def forward(cfg, P, x, masks=None, stn=True):
    """Training-mode forward with torch ops; P: dict of torch tensors; returns y_pred (B,T,C)."""
    masks = masks or {}
    B = x.shape[0]
    if stn:
        l = maxpool(x, 2, 2)
        l = conv_nhwc(l, P["stn_c1_k"], P["stn_c1_b"])
        l = maxpool(l, 2, 2)
        l = conv_nhwc(l, P["stn_c2_k"], P["stn_c2_b"])
        l = torch.relu(l.reshape(B, -1) @ P["stn_d1_w"] + P["stn_d1_b"])
        theta = l @ P["stn_d2_w"] + P["stn_d2_b"]
        xs = sampler(x, theta)
    else:
        xs = x
    h = F.pad(xs, (0, 0, 2, 2, 2, 2))
    for i, (cout, pool) in enumerate(M.BLOCKS, 1):
        C = h.shape[-1]
        d = conv_nhwc(h, P[f"b{i}_dw"].reshape(3, 3, C, 1).permute(0, 1, 3, 2), None, groups=C, padding=1)
        a = torch.clamp(bn_train(d, P[f"b{i}_bn1_g"], P[f"b{i}_bn1_b"]), 0, 6)
        q = a @ P[f"b{i}_pw"]
        r = torch.clamp(bn_train(q, P[f"b{i}_bn2_g"], P[f"b{i}_bn2_b"]), 0, 6)
        if pool:
            r = maxpool(r, *pool)
        if masks.get(f"b{i}") is not None:
            r = r * t(masks[f"b{i}"]).to(r.dtype) / (1 - M.DROP_BLOCK)
        h = r
    h = h.reshape(B, cfg.T, cfg.feat)
    h = torch.relu(h @ P["dense1_w"] + P["dense1_b"])
    if masks.get("dense1") is not None:
        h = h * t(masks["dense1"]).to(h.dtype) / (1 - M.DROP_DENSE1)
    cell = gru if cfg.gru else lstm
    h = cell(h, P["rnn1f_w"], P["rnn1f_u"], P["rnn1f_b"], False) + cell(h, P["rnn1b_w"], P["rnn1b_u"], P["rnn1b_b"], True)
    h = torch.cat([cell(h, P["rnn2f_w"], P["rnn2f_u"], P["rnn2f_b"], False),
                   cell(h, P["rnn2b_w"], P["rnn2b_u"], P["rnn2b_b"], True)], -1)
    if masks.get("rnn") is not None:
        h = h * t(masks["rnn"]).to(h.dtype) / (1 - M.DROP_RNN)
    return torch.softmax(h @ P["dense2_w"] + P["dense2_b"], -1)


def ctc_cost(y_pred, labels, input_length, label_length):
    """K.ctc_batch_cost on y_pred[:, 2:] via torch's CTC (independent implementation)."""
    y = y_pred[:, 2:, :]
    z = torch.log(y.permute(1, 0, 2) + 1e-7)
    lp = torch.log_softmax(z, -1)
    C = y.shape[-1]
    return F.ctc_loss(lp, torch.tensor(labels), torch.tensor(input_length), torch.tensor(label_length),
                      blank=C - 1, reduction="none", zero_infinity=False)


def train_step_benchmark(batch=64, threads=4, steps=2, warmup=1, imgh=100, seed=0):
    """Seconds per full fp32 train step (forward in training mode, CTC cost, autograd backward, global-norm clip 5, Keras-form Adam)
    of the BiLSTM CRNN on `batch` synthetic 100x32 images with `threads` torch intra-op threads.  Returns (sec_per_step, last_loss)."""
    import time
    prev = torch.get_num_threads()
    torch.set_num_threads(int(threads))
    try:
        cfg = M.Config(imgh=imgh)
        p, bn = M.init_params(cfg, seed=1, dtype=np.float32)
        x, lab, il, ll = M.synthetic_batch(cfg, batch, seed=seed)
        P = {k: torch.tensor(np.asarray(v), dtype=torch.float32, requires_grad=True) for k, v in p.items()}
        xt = torch.tensor(x, dtype=torch.float32)
        m = {k: torch.zeros_like(v) for k, v in P.items()}; vv = {k: torch.zeros_like(v) for k, v in P.items()}
        lr, b1, b2, eps, clip = 1e-4, 0.5, 0.999, 1e-7, 5.0
        t0, loss = None, None
        for it in range(warmup + steps):
            if it == warmup:
                t0 = time.perf_counter()
            y = forward(cfg, P, xt)
            loss = ctc_cost(y, lab, il, ll).mean()
            grads = torch.autograd.grad(loss, list(P.values()))
            with torch.no_grad():
                norm = torch.sqrt(sum((g * g).sum() for g in grads))
                scale = torch.clamp(clip / (norm + 1e-30), max=1.0)
                tt = it + 1
                lr_t = lr * (1.0 - b2 ** tt) ** 0.5 / (1.0 - b1 ** tt)
                for (k, w), g in zip(P.items(), grads):
                    g = g * scale
                    m[k].mul_(b1).add_(g, alpha=1 - b1)
                    vv[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                    w.sub_(lr_t * m[k] / (vv[k].sqrt() + eps))
        return (time.perf_counter() - t0) / steps, float(loss.detach())
    finally:
        torch.set_num_threads(prev)
