"""CPU oracle (TEST INFRASTRUCTURE) -- the whole CRNN graph, its backward, and the optimizers.

Graph: utils.py:58-96 (CRNN.get_model) + utils.py:247-258 (STN) + utils.py:98-103 (CTC);
train-step semantics: train.py:187-192 (Adam beta1=.5 clipnorm 5 / SGD nesterov) -- Keras 2.2.2
optimizer formulas restated from SURVEY A.8.  Parameter names/shapes/order follow Keras'
weight order (models/*/model_summary.txt).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
from . import ops
from . import ctc as ctc_mod

BLOCKS = [  # (Cout, pool) for the 7 depthwise-separable blocks, utils.py:64-70
    (64, None), (128, None), (256, (2, 2)), (256, None), (512, (1, 2)), (512, None), (512, None)]
DROP_BLOCK, DROP_DENSE1, DROP_RNN = 0.1, 0.4, 0.2  # utils.py:56,75,83


class Config:
    def __init__(self, imgh=100, imgw=32, num_classes=38, max_len=23, time_dense_size=128,
                 n_units=256, gru=False):
        self.imgh, self.imgw, self.num_classes = imgh, imgw, num_classes
        self.max_len, self.tds, self.u, self.gru = max_len, time_dense_size, n_units, gru
        self.Hp, self.Wp = imgh + 4, imgw + 4
        self.H1, self.W1 = self.Hp // 2, self.Wp // 2
        self.W2 = self.W1 // 2
        self.T = self.H1
        self.feat = self.W2 * 512
        h = (imgh // 2 - 4) // 2 - 4
        w = (imgw // 2 - 4) // 2 - 4
        self.stn_flat = h * w * 20
        self.ng = 3 if gru else 4

    def param_shapes(self):
        """Trainable tensors in Keras order, BN moving stats listed separately."""
        s = [("stn_c1_k", (5, 5, 1, 20)), ("stn_c1_b", (20,)),
             ("stn_c2_k", (5, 5, 20, 20)), ("stn_c2_b", (20,)),
             ("stn_d1_w", (self.stn_flat, 50)), ("stn_d1_b", (50,)),
             ("stn_d2_w", (50, 6)), ("stn_d2_b", (6,))]
        cin = 1
        for i, (cout, _) in enumerate(BLOCKS, 1):
            s += [(f"b{i}_dw", (3, 3, cin)), (f"b{i}_bn1_g", (cin,)), (f"b{i}_bn1_b", (cin,)),
                  (f"b{i}_pw", (cin, cout)), (f"b{i}_bn2_g", (cout,)), (f"b{i}_bn2_b", (cout,))]
            cin = cout
        s += [("dense1_w", (self.feat, self.tds)), ("dense1_b", (self.tds,))]
        g = self.ng * self.u
        for layer, din in ((1, self.tds), (2, self.u)):
            for d in ("f", "b"):
                s += [(f"rnn{layer}{d}_w", (din, g)), (f"rnn{layer}{d}_u", (self.u, g)),
                      (f"rnn{layer}{d}_b", (g,))]
        s += [("dense2_w", (2 * self.u, self.num_classes)), ("dense2_b", (self.num_classes,))]
        return s

    def bn_shapes(self):
        s = []
        cin = 1
        for i, (cout, _) in enumerate(BLOCKS, 1):
            s += [(f"b{i}_bn1", cin), (f"b{i}_bn2", cout)]
            cin = cout
        return s

    def n_trainable(self):
        return sum(int(np.prod(sh)) for _, sh in self.param_shapes())


# ----------------------------------------------------------------------------------------
# Initialisers (Keras 2.2.2 families; RNG streams cannot match TF -- synthetic)
# ----------------------------------------------------------------------------------------
def _fans(shape):
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def _glorot_uniform(rs, shape, fans=None):
    fi, fo = fans if fans else _fans(shape)
    lim = np.sqrt(6.0 / (fi + fo))
    return rs.uniform(-lim, lim, size=shape)


def _he_normal(rs, shape):
    fi, _ = _fans(shape)
    std = np.sqrt(2.0 / fi)
    v = rs.normal(0, std, size=shape)
    bad = np.abs(v) > 2 * std  # truncated normal: resample beyond 2 sigma
    while bad.any():
        v[bad] = rs.normal(0, std, size=int(bad.sum()))
        bad = np.abs(v) > 2 * std
    return v


def _orthogonal(rs, shape):
    a = rs.normal(0, 1, size=shape)
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    return u if u.shape == tuple(shape) else vt


def init_params(cfg, seed=1, dtype=np.float64, stn_identity=True):
    rs = np.random.RandomState(seed)
    p = {}
    for name, shape in cfg.param_shapes():
        if name.endswith("_dw"):
            v = _glorot_uniform(rs, shape, fans=(9 * shape[2], 9))  # Keras kernel is (3,3,C,1)
        elif name.endswith("bn1_g") or name.endswith("bn2_g"):
            v = np.ones(shape)
        elif name.endswith("bn1_b") or name.endswith("bn2_b"):
            v = np.zeros(shape)
        elif name == "stn_d2_w":
            v = np.zeros(shape) if stn_identity else _glorot_uniform(rs, shape)
        elif name == "stn_d2_b":
            v = np.array([1, 0, 0, 0, 1, 0], dtype=np.float64)  # utils.py:239-245
        elif name.startswith("rnn") and name.endswith("_w"):
            v = _he_normal(rs, shape)
        elif name.startswith("rnn") and name.endswith("_u"):
            v = _orthogonal(rs, shape)
        elif name.startswith("rnn") and name.endswith("_b"):
            v = np.zeros(shape)
            if not cfg.gru:
                v[cfg.u:2 * cfg.u] = 1.0  # unit_forget_bias
        elif name == "dense2_w":
            v = _he_normal(rs, shape)
        elif name.endswith("_b"):
            v = np.zeros(shape)
        else:
            v = _glorot_uniform(rs, shape)
        p[name] = np.ascontiguousarray(v, dtype=dtype)
    bn = {}
    for name, c in cfg.bn_shapes():
        bn[name + "_mean"] = np.zeros(c, dtype=dtype)
        bn[name + "_var"] = np.ones(c, dtype=dtype)
    return p, bn


def randomize_params(cfg, p, seed=3, scale=0.3):
    """Perturb the 'structured' initial values (BN gamma/beta, biases, STN dense_2) so that
    parity tests exercise every gradient path (identity-STN zeroes the locnet gradients)."""
    rs = np.random.RandomState(seed)
    for name in p:
        if name.endswith("_g"):
            p[name] = (1 + scale * rs.uniform(-1, 1, p[name].shape)).astype(p[name].dtype)
        elif name.endswith("_b") and not name.startswith("stn_d2"):
            p[name] = (p[name] + scale * rs.uniform(-1, 1, p[name].shape)).astype(p[name].dtype)
    p["stn_d2_w"] = (0.02 * rs.uniform(-1, 1, p["stn_d2_w"].shape)).astype(p["stn_d2_w"].dtype)
    p["stn_d2_b"] = (p["stn_d2_b"] + 0.05 * rs.uniform(-1, 1, 6)).astype(p["stn_d2_b"].dtype)
    return p


# ----------------------------------------------------------------------------------------
# Forward / backward
# ----------------------------------------------------------------------------------------
def forward(cfg, p, bn, x, train=False, masks=None, stn=True):
    """x (B,imgh,imgw,1) -> softmax (B,T,C).  `masks`: dict of dropout keep-masks
    ('b1'..'b7', 'dense1', 'rnn') or None for no dropout.  Returns (y_pred, cache)."""
    masks = masks or {}
    B = x.shape[0]
    c = {"x": x}
    # ---- STN (utils.py:247-258)
    if stn:
        c["pool1"] = ops.maxpool_fwd(x, 2, 2)
        c["c1"] = ops.conv_valid_fwd(c["pool1"], p["stn_c1_k"], p["stn_c1_b"])
        c["pool2"] = ops.maxpool_fwd(c["c1"], 2, 2)
        c["c2"] = ops.conv_valid_fwd(c["pool2"], p["stn_c2_k"], p["stn_c2_b"])
        c["flat"] = c["c2"].reshape(B, -1)
        c["fc1"] = ops.relu_fwd(ops.dense_fwd(c["flat"], p["stn_d1_w"], p["stn_d1_b"]))
        c["theta"] = ops.dense_fwd(c["fc1"], p["stn_d2_w"], p["stn_d2_b"])
        xs = ops.sampler_fwd(x, c["theta"])
    else:
        xs = x
    c["xs"] = xs
    h = ops.zeropad_fwd(xs, 2)
    # ---- depthwise-separable stack (utils.py:43-56, 64-70)
    c["stats"] = {}
    for i, (cout, pool) in enumerate(BLOCKS, 1):
        c[f"in{i}"] = h
        d = ops.dwconv_fwd(h, p[f"b{i}_dw"])
        if train:
            y, m, v = ops.bn_train_fwd(d, p[f"b{i}_bn1_g"], p[f"b{i}_bn1_b"])
            c["stats"][f"b{i}_bn1"] = (m, v, d.size // d.shape[-1])
        else:
            y = ops.bn_infer_fwd(d, p[f"b{i}_bn1_g"], p[f"b{i}_bn1_b"], bn[f"b{i}_bn1_mean"], bn[f"b{i}_bn1_var"])
        a = ops.relu6_fwd(y)
        q = a @ p[f"b{i}_pw"]
        if train:
            y2, m2, v2 = ops.bn_train_fwd(q, p[f"b{i}_bn2_g"], p[f"b{i}_bn2_b"])
            c["stats"][f"b{i}_bn2"] = (m2, v2, q.size // q.shape[-1])
        else:
            y2 = ops.bn_infer_fwd(q, p[f"b{i}_bn2_g"], p[f"b{i}_bn2_b"], bn[f"b{i}_bn2_mean"], bn[f"b{i}_bn2_var"])
        r = ops.relu6_fwd(y2)
        c[f"d{i}"], c[f"a{i}"], c[f"q{i}"], c[f"r{i}"] = d, a, q, r
        if pool:
            r = ops.maxpool_fwd(r, *pool)
        h = ops.dropout_fwd(r, masks.get(f"b{i}"), DROP_BLOCK)
    c["conv_out"] = h
    # ---- Reshape + dense1 (utils.py:72-75)
    feat = h.reshape(B, cfg.T, cfg.feat)
    c["feat"] = feat
    c["dense1"] = ops.relu_fwd(ops.dense_fwd(feat, p["dense1_w"], p["dense1_b"]))
    h = ops.dropout_fwd(c["dense1"], masks.get("dense1"), DROP_DENSE1)
    c["rnn_in"] = h
    # ---- 2x Bidirectional (utils.py:77-82)
    cell_f = ops.gru_fwd if cfg.gru else ops.lstm_fwd
    hf, c["rnn1f"] = cell_f(h, p["rnn1f_w"], p["rnn1f_u"], p["rnn1f_b"], reverse=False)
    hb, c["rnn1b"] = cell_f(h, p["rnn1b_w"], p["rnn1b_u"], p["rnn1b_b"], reverse=True)
    h = hf + hb
    hf, c["rnn2f"] = cell_f(h, p["rnn2f_w"], p["rnn2f_u"], p["rnn2f_b"], reverse=False)
    hb, c["rnn2b"] = cell_f(h, p["rnn2b_w"], p["rnn2b_u"], p["rnn2b_b"], reverse=True)
    c["rnn_out"] = np.concatenate([hf, hb], axis=-1)
    h = ops.dropout_fwd(c["rnn_out"], masks.get("rnn"), DROP_RNN)
    c["dense2_in"] = h
    # ---- dense2 + softmax (utils.py:85-86)
    c["logits"] = ops.dense_fwd(h, p["dense2_w"], p["dense2_b"])
    y_pred = ops.softmax_fwd(c["logits"])
    c["y_pred"] = y_pred
    return y_pred, c


def backward(cfg, p, c, g_ypred, masks=None, stn=True):
    """Backward of `forward(train=True)`.  g_ypred = dLoss/dy_pred (B,T,C).  Returns grads dict."""
    masks = masks or {}
    g = {}
    B = g_ypred.shape[0]
    gz = ops.softmax_bwd(c["y_pred"], g_ypred)
    gh, g["dense2_w"], g["dense2_b"] = ops.dense_bwd(c["dense2_in"], p["dense2_w"], gz)
    gh = ops.dropout_bwd(gh, masks.get("rnn"), DROP_RNN)
    cell_b = ops.gru_bwd if cfg.gru else ops.lstm_bwd
    u = cfg.u
    dxf, g["rnn2f_w"], g["rnn2f_u"], g["rnn2f_b"] = cell_b(c["rnn2f"], gh[..., :u])
    dxb, g["rnn2b_w"], g["rnn2b_u"], g["rnn2b_b"] = cell_b(c["rnn2b"], gh[..., u:])
    gh = dxf + dxb
    dxf, g["rnn1f_w"], g["rnn1f_u"], g["rnn1f_b"] = cell_b(c["rnn1f"], gh)
    dxb, g["rnn1b_w"], g["rnn1b_u"], g["rnn1b_b"] = cell_b(c["rnn1b"], gh)
    gh = dxf + dxb
    gh = ops.dropout_bwd(gh, masks.get("dense1"), DROP_DENSE1)
    gh = ops.relu_bwd_from_out(c["dense1"], gh)
    gh, g["dense1_w"], g["dense1_b"] = ops.dense_bwd(c["feat"], p["dense1_w"], gh)
    gh = gh.reshape(c["conv_out"].shape)
    for i in range(len(BLOCKS), 0, -1):
        cout, pool = BLOCKS[i - 1]
        gh = ops.dropout_bwd(gh, masks.get(f"b{i}"), DROP_BLOCK)
        if pool:
            gh = ops.maxpool_bwd(c[f"r{i}"], gh, *pool)
        gh = ops.relu6_bwd_from_out(c[f"r{i}"], gh)
        m2, v2, _ = c["stats"][f"b{i}_bn2"]
        gh, g[f"b{i}_bn2_g"], g[f"b{i}_bn2_b"] = ops.bn_train_bwd(c[f"q{i}"], p[f"b{i}_bn2_g"], m2, v2, gh)
        a = c[f"a{i}"]
        g[f"b{i}_pw"] = a.reshape(-1, a.shape[-1]).T @ gh.reshape(-1, gh.shape[-1])
        gh = gh @ p[f"b{i}_pw"].T
        gh = ops.relu6_bwd_from_out(a, gh)
        m1, v1, _ = c["stats"][f"b{i}_bn1"]
        gh, g[f"b{i}_bn1_g"], g[f"b{i}_bn1_b"] = ops.bn_train_bwd(c[f"d{i}"], p[f"b{i}_bn1_g"], m1, v1, gh)
        gh, g[f"b{i}_dw"] = ops.dwconv_bwd(c[f"in{i}"], p[f"b{i}_dw"], gh)
    gxs = ops.zeropad_bwd(gh, 2)
    g["xs"] = gxs
    if stn:
        gt = ops.sampler_bwd(c["x"], c["theta"], gxs)
        g["theta"] = gt
        gh, g["stn_d2_w"], g["stn_d2_b"] = ops.dense_bwd(c["fc1"], p["stn_d2_w"], gt)
        gh = ops.relu_bwd_from_out(c["fc1"], gh)
        gh, g["stn_d1_w"], g["stn_d1_b"] = ops.dense_bwd(c["flat"], p["stn_d1_w"], gh)
        gh = gh.reshape(c["c2"].shape)
        gh, g["stn_c2_k"], g["stn_c2_b"] = ops.conv_valid_bwd(c["pool2"], p["stn_c2_k"], gh)
        gh = ops.maxpool_bwd(c["c1"], gh, 2, 2)
        gh, g["stn_c1_k"], g["stn_c1_b"] = ops.conv_valid_bwd(c["pool1"], p["stn_c1_k"], gh)
    else:
        for n, sh in cfg.param_shapes():
            if n.startswith("stn_"):
                g[n] = np.zeros(sh, dtype=gxs.dtype)
    return g


def loss_and_grads(cfg, p, bn, x, labels, input_length, label_length, masks=None, stn=True):
    """One training-mode forward + CTC + backward.  loss = mean_b ctc_b (train.py:192)."""
    y_pred, c = forward(cfg, p, bn, x, train=True, masks=masks, stn=stn)
    loss_b, gy = ctc_mod.ctc_loss_and_grad(y_pred, labels, input_length, label_length)
    B = x.shape[0]
    grads = backward(cfg, p, c, gy / B, masks=masks, stn=stn)
    return float(loss_b.mean()), loss_b, grads, c


def bn_update(cfg, bn, c):
    for name, _ in cfg.bn_shapes():
        m, v, n = c["stats"][name]
        bn[name + "_mean"], bn[name + "_var"] = ops.bn_moving_update(bn[name + "_mean"], bn[name + "_var"], m, v, float(n))
    return bn


# ----------------------------------------------------------------------------------------
# Optimizers (Keras 2.2.2; train.py:187-190)
# ----------------------------------------------------------------------------------------
def global_norm(grads, names):
    return float(np.sqrt(sum(float((grads[n].astype(np.float64) ** 2).sum()) for n in names)))


def clip_by_global_norm(grads, names, clipnorm):
    n = global_norm(grads, names)
    if clipnorm and n >= clipnorm:
        s = clipnorm / n
        return {k: grads[k] * s for k in names}, n
    return {k: grads[k] for k in names}, n


class Adam:
    def __init__(self, lr=1e-4, beta_1=0.5, beta_2=0.999, epsilon=1e-7, clipnorm=5.0):
        self.lr, self.b1, self.b2, self.eps, self.clipnorm = lr, beta_1, beta_2, epsilon, clipnorm
        self.it = 0
        self.m, self.v = {}, {}

    def step(self, p, grads):
        names = list(p.keys())
        g, _ = clip_by_global_norm(grads, names, self.clipnorm)
        self.it += 1
        t = self.it
        lr_t = self.lr * np.sqrt(1 - self.b2 ** t) / (1 - self.b1 ** t)
        for n in names:
            m = self.b1 * self.m.get(n, 0.0) + (1 - self.b1) * g[n]
            v = self.b2 * self.v.get(n, 0.0) + (1 - self.b2) * g[n] ** 2
            p[n] = p[n] - lr_t * m / (np.sqrt(v) + self.eps)
            self.m[n], self.v[n] = m, v
        return p


class SGD:
    def __init__(self, lr=1e-3, decay=1e-6, momentum=0.9, nesterov=True, clipnorm=5.0):
        self.lr, self.decay, self.mom, self.nesterov, self.clipnorm = lr, decay, momentum, nesterov, clipnorm
        self.it = 0
        self.vel = {}

    def step(self, p, grads):
        names = list(p.keys())
        g, _ = clip_by_global_norm(grads, names, self.clipnorm)
        lr = self.lr / (1.0 + self.decay * self.it)
        self.it += 1
        for n in names:
            v = self.mom * self.vel.get(n, 0.0) - lr * g[n]
            self.vel[n] = v
            p[n] = p[n] + (self.mom * v - lr * g[n] if self.nesterov else v)
        return p


# ----------------------------------------------------------------------------------------
# Synthetic batch
# ----------------------------------------------------------------------------------------
NORM_MEAN, NORM_STD = 118.24236953981779, 36.72835353999682  # utils.py:421


def variable_width_images(rs, B, imgh, imgw, wmin=40):
    """SURVEY 8d C3 (BASELINE configs[2], IAM shape): uint8 noise "text" occupying a random prefix of wmin..imgh rows of the time
    axis (axis 0 after the reference's rotation, utils.py:370), the rest filled with the text's modal grey value the way open_img
    pads a short word (utils.py:372-373,379-400 with the augmentation branch off)."""
    x = np.empty((B, imgh, imgw, 1), dtype=np.uint8)
    for b in range(B):
        w = int(rs.randint(min(wmin, imgh), imgh + 1))
        text = rs.randint(0, 256, (w, imgw)).astype(np.uint8)
        val, counts = np.unique(text, return_counts=True)
        fill = val[np.where(counts == counts.max())[0][0]]
        x[b, :w, :, 0] = text
        x[b, w:, :, 0] = fill
    return x


def synthetic_batch(cfg, B, seed=0, dtype=np.float32, variable_width=False):
    rs = np.random.RandomState(seed)
    raw = variable_width_images(rs, B, cfg.imgh, cfg.imgw) if variable_width else rs.randint(0, 256, (B, cfg.imgh, cfg.imgw, 1))
    x = ((raw.astype(np.float32) - NORM_MEAN) / NORM_STD).astype(dtype)
    blank = cfg.num_classes - 1
    ll = rs.randint(1, cfg.max_len + 1, size=B)
    labels = np.full((B, cfg.max_len), blank, dtype=np.int64)
    for b in range(B):
        labels[b, :ll[b]] = rs.randint(0, blank, size=ll[b])
    il = np.full(B, cfg.T - 2, dtype=np.int64)
    return x, labels, il, ll.astype(np.int64)
