"""CPU oracle (TEST INFRASTRUCTURE, never shipped/measured as the product) -- layer ops.

NumPy restatement of the arithmetic that the reference delegates to Keras 2.2.2 / TF 1.8
for the CRNN-OCR hot path (reference: /root/reference/utils.py).  Parity status: the
third-party arithmetic (Keras/TF) is NOT runnable in this container, so these functions
restate published Keras-2.2.2/TF-1.8 semantics; the pieces of the
reference that ARE runnable (BilinearInterpolation over a NumPy K-shim, pure-Python
helpers) pin the oracle through tests/golden/*.npz.  Everything else is "parity unpinned
by the reference" and is cross-checked against torch-CPU autograd in tests/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

All tensors are NHWC / channels_last (utils.py:60, models/*/model.json "data_format").
Every op has a forward and a hand-written backward; dtype follows the inputs
(float64 for checking, float32 for the timed CPU baseline).
"""
import numpy as np


# ----------------------------------------------------------------------------------------
# MaxPooling2D / MaxPool2D (utils.py:51, 248, 250) -- pool == stride, 'valid'
# ----------------------------------------------------------------------------------------
def maxpool_fwd(x, ph, pw):
    B, H, W, C = x.shape
    Ho, Wo = H // ph, W // pw
    xv = x[:, :Ho * ph, :Wo * pw, :].reshape(B, Ho, ph, Wo, pw, C)
    xv = xv.transpose(0, 1, 3, 2, 4, 5).reshape(B, Ho, Wo, ph * pw, C)
    return xv.max(axis=3)


def maxpool_bwd(x, gy, ph, pw):
    """Gradient goes to the FIRST maximum in window scan order (row-major over the window),
    which is what TF's CPU MaxPoolGrad (strict '<' update) and torch do."""
    B, H, W, C = x.shape
    Ho, Wo = H // ph, W // pw
    xv = x[:, :Ho * ph, :Wo * pw, :].reshape(B, Ho, ph, Wo, pw, C)
    xv = xv.transpose(0, 1, 3, 2, 4, 5).reshape(B, Ho, Wo, ph * pw, C)
    arg = xv.argmax(axis=3)  # first occurrence
    g = np.zeros_like(xv)
    np.put_along_axis(g, arg[:, :, :, None, :], gy[:, :, :, None, :], axis=3)
    g = g.reshape(B, Ho, Wo, ph, pw, C).transpose(0, 1, 3, 2, 4, 5).reshape(B, Ho * ph, Wo * pw, C)
    gx = np.zeros_like(x)
    gx[:, :Ho * ph, :Wo * pw, :] = g
    return gx


# ----------------------------------------------------------------------------------------
# Conv2D 5x5 'valid', bias, linear (STN locnet, utils.py:249,251)
# ----------------------------------------------------------------------------------------
def im2col(x, kh, kw):
    """(B,H,W,C) -> (B*Ho*Wo, kh*kw*C); column index = (i*kw + j)*C + c (HWIO kernel order)."""
    B, H, W, C = x.shape
    Ho, Wo = H - kh + 1, W - kw + 1
    cols = np.empty((B, Ho, Wo, kh, kw, C), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i, j, :] = x[:, i:i + Ho, j:j + Wo, :]
    return cols.reshape(B * Ho * Wo, kh * kw * C)


def col2im(dcols, xshape, kh, kw):
    B, H, W, C = xshape
    Ho, Wo = H - kh + 1, W - kw + 1
    dc = dcols.reshape(B, Ho, Wo, kh, kw, C)
    dx = np.zeros(xshape, dtype=dcols.dtype)
    for i in range(kh):
        for j in range(kw):
            dx[:, i:i + Ho, j:j + Wo, :] += dc[:, :, :, i, j, :]
    return dx


def conv_valid_fwd(x, k, b):
    kh, kw, ci, co = k.shape
    B, H, W, _ = x.shape
    y = im2col(x, kh, kw) @ k.reshape(kh * kw * ci, co) + b
    return y.reshape(B, H - kh + 1, W - kw + 1, co)


def conv_valid_bwd(x, k, gy):
    kh, kw, ci, co = k.shape
    g2 = gy.reshape(-1, co)
    cols = im2col(x, kh, kw)
    dk = (cols.T @ g2).reshape(k.shape)
    db = g2.sum(axis=0)
    dx = col2im(g2 @ k.reshape(kh * kw * ci, co).T, x.shape, kh, kw)
    return dx, dk, db


# ----------------------------------------------------------------------------------------
# BilinearInterpolation (utils.py:140-232), SURVEY A.2 -- exact quirks preserved
# ----------------------------------------------------------------------------------------
def _regular_grid(H, W, dtype):
    xs = np.linspace(-1.0, 1.0, W).astype(dtype)
    ys = np.linspace(-1.0, 1.0, H).astype(dtype)
    xg, yg = np.meshgrid(xs, ys)  # x fastest (utils.py:209-213)
    return np.stack([xg.ravel(), yg.ravel(), np.ones(H * W, dtype=dtype)], 0)  # (3, H*W)


def sampler_fwd(image, theta, out_hw=None):
    """image (B,H,W,C), theta (B,6) -> (B,Ho,Wo,C).  utils.py:222-232 then 140-205.
    x = .5(x+1)*W (not W-1); x0 = trunc toward zero; corners clipped BEFORE the area
    weights are formed (so border points extrapolate); add order ((a+b)+c)+d."""
    B, H, W, C = image.shape
    Ho, Wo = out_hw if out_hw is not None else (H, W)
    dt = image.dtype
    G = _regular_grid(Ho, Wo, dt)
    S = theta.reshape(B, 2, 3).astype(dt) @ G  # (B,2,Ho*Wo)
    x = (dt.type(0.5) * (S[:, 0, :] + dt.type(1.0))) * dt.type(W)
    y = (dt.type(0.5) * (S[:, 1, :] + dt.type(1.0))) * dt.type(H)
    x0 = np.trunc(x).astype(np.int64)
    y0 = np.trunc(y).astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0 = np.clip(x0, 0, W - 1); x1 = np.clip(x1, 0, W - 1)
    y0 = np.clip(y0, 0, H - 1); y1 = np.clip(y1, 0, H - 1)
    bidx = np.arange(B)[:, None]
    Pa = image[bidx, y0, x0]; Pb = image[bidx, y1, x0]
    Pc = image[bidx, y0, x1]; Pd = image[bidx, y1, x1]  # (B,N,C)
    x0f, x1f, y0f, y1f = (a.astype(dt) for a in (x0, x1, y0, y1))
    wa = ((x1f - x) * (y1f - y))[..., None]
    wb = ((x1f - x) * (y - y0f))[..., None]
    wc = ((x - x0f) * (y1f - y))[..., None]
    wd = ((x - x0f) * (y - y0f))[..., None]
    out = ((wa * Pa + wb * Pb) + wc * Pc) + wd * Pd
    return out.reshape(B, Ho, Wo, C)


def sampler_bwd(image, theta, gout):
    """d(loss)/d(theta) only (the image is data, utils.py:62).  x0..y1 are constants."""
    B, H, W, C = image.shape
    Ho, Wo = gout.shape[1:3]
    dt = image.dtype
    G = _regular_grid(Ho, Wo, dt)
    S = theta.reshape(B, 2, 3).astype(dt) @ G
    x = (dt.type(0.5) * (S[:, 0, :] + dt.type(1.0))) * dt.type(W)
    y = (dt.type(0.5) * (S[:, 1, :] + dt.type(1.0))) * dt.type(H)
    x0 = np.trunc(x).astype(np.int64); y0 = np.trunc(y).astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0 = np.clip(x0, 0, W - 1); x1 = np.clip(x1, 0, W - 1)
    y0 = np.clip(y0, 0, H - 1); y1 = np.clip(y1, 0, H - 1)
    bidx = np.arange(B)[:, None]
    Pa = image[bidx, y0, x0]; Pb = image[bidx, y1, x0]
    Pc = image[bidx, y0, x1]; Pd = image[bidx, y1, x1]
    x0f, x1f, y0f, y1f = (a.astype(dt)[..., None] for a in (x0, x1, y0, y1))
    xe, ye = x[..., None], y[..., None]
    g = gout.reshape(B, Ho * Wo, C)
    dx = (g * (-(y1f - ye) * Pa - (ye - y0f) * Pb + (y1f - ye) * Pc + (ye - y0f) * Pd)).sum(-1)
    dy = (g * (-(x1f - xe) * Pa + (x1f - xe) * Pb - (xe - x0f) * Pc + (xe - x0f) * Pd)).sum(-1)
    dS = np.stack([dx * dt.type(0.5 * W), dy * dt.type(0.5 * H)], 1)  # (B,2,N)
    return (dS @ G.T).reshape(B, 6)


# ----------------------------------------------------------------------------------------
# ZeroPadding2D((2,2)) (utils.py:63)
# ----------------------------------------------------------------------------------------
def zeropad_fwd(x, p=2):
    return np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)))


def zeropad_bwd(g, p=2):
    return g[:, p:-p, p:-p, :]


# ----------------------------------------------------------------------------------------
# DepthwiseConv2D 3x3 'same', stride 1, multiplier 1, no bias (utils.py:44)
# kernel (3,3,C)  [Keras stores (3,3,C,1)]
# ----------------------------------------------------------------------------------------
def dwconv_fwd(x, k):
    B, H, W, C = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    y = np.zeros_like(x)
    for i in range(3):
        for j in range(3):
            y += xp[:, i:i + H, j:j + W, :] * k[i, j]
    return y


def dwconv_bwd(x, k, gy):
    B, H, W, C = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    gp = np.pad(gy, ((0, 0), (1, 1), (1, 1), (0, 0)))
    dx = np.zeros_like(x)
    dk = np.zeros_like(k)
    for i in range(3):
        for j in range(3):
            dk[i, j] = (xp[:, i:i + H, j:j + W, :] * gy).sum(axis=(0, 1, 2))
            dx += gp[:, 2 - i:2 - i + H, 2 - j:2 - j + W, :] * k[i, j]
    return dx, dk


# ----------------------------------------------------------------------------------------
# BatchNormalization(axis=-1, momentum=.99, epsilon=1e-3) (utils.py:45,48), SURVEY A.4
# ----------------------------------------------------------------------------------------
BN_EPS = 1e-3
BN_MOMENTUM = 0.99


def bn_train_fwd(x, gamma, beta, eps=BN_EPS):
    """x (..., C): statistics over all leading axes; biased variance in the normaliser."""
    x2 = x.reshape(-1, x.shape[-1])
    mean = x2.mean(axis=0)
    var = ((x2 - mean) ** 2).mean(axis=0)
    inv = 1.0 / np.sqrt(var + x.dtype.type(eps))
    y = (x - mean) * (inv * gamma) + beta
    return y, mean, var


def bn_train_bwd(x, gamma, mean, var, gy, eps=BN_EPS):
    C = x.shape[-1]
    x2 = x.reshape(-1, C); g2 = gy.reshape(-1, C)
    n = x2.shape[0]
    inv = 1.0 / np.sqrt(var + x.dtype.type(eps))
    xhat = (x2 - mean) * inv
    dgamma = (g2 * xhat).sum(axis=0)
    dbeta = g2.sum(axis=0)
    dx = (gamma * inv) * (g2 - dbeta / n - xhat * (dgamma / n))
    return dx.reshape(x.shape), dgamma, dbeta


def bn_infer_fwd(x, gamma, beta, mmean, mvar, eps=BN_EPS):
    return (x - mmean) * (gamma / np.sqrt(mvar + x.dtype.type(eps))) + beta


def bn_moving_update(mmean, mvar, mean, var, n, momentum=BN_MOMENTUM, eps=BN_EPS):
    """Keras 2.2.2 + TF fused batch norm (recalled, SURVEY A.4 /!\\): TF returns the Bessel-
    corrected batch variance, Keras multiplies again by n/(n-(1+eps))."""
    vhat = var * (n / (n - 1.0)) * (n / (n - (1.0 + eps)))
    return (momentum * mmean + (1 - momentum) * mean, momentum * mvar + (1 - momentum) * vhat)


# ----------------------------------------------------------------------------------------
# ReLU(6.) (utils.py:46,49), relu (utils.py:74,254)
# ----------------------------------------------------------------------------------------
def relu6_fwd(x):
    return np.minimum(np.maximum(x, 0), 6)


def relu6_bwd_from_out(y, gy):
    return gy * ((y > 0) & (y < 6))


def relu_fwd(x):
    return np.maximum(x, 0)


def relu_bwd_from_out(y, gy):
    return gy * (y > 0)


# ----------------------------------------------------------------------------------------
# Dropout (inverted; training only) with an injected keep-mask (1 = keep)
# ----------------------------------------------------------------------------------------
def dropout_fwd(x, mask, rate):
    if mask is None:
        return x
    return x * mask * x.dtype.type(1.0 / (1.0 - rate))


dropout_bwd = dropout_fwd


# ----------------------------------------------------------------------------------------
# Dense (utils.py:74,85,253,256)
# ----------------------------------------------------------------------------------------
def dense_fwd(x, W, b):
    return x @ W + b


def dense_bwd(x, W, gy):
    x2 = x.reshape(-1, x.shape[-1]); g2 = gy.reshape(-1, gy.shape[-1])
    return (gy @ W.T), x2.T @ g2, g2.sum(axis=0)


# ----------------------------------------------------------------------------------------
# Recurrent cells (Keras 2.2.2, hard_sigmoid / tanh), SURVEY A.5.  x is (B,T,in).
# ----------------------------------------------------------------------------------------
def hard_sigmoid(z):
    return np.clip(z * 0.2 + 0.5, 0, 1)


def _hs_grad_from_out(a):
    return 0.2 * ((a > 0) & (a < 1))


def lstm_fwd(x, W, U, b, reverse=False):
    """Gate order i,f,c,o.  Returns h sequence (B,T,u) in ORIGINAL time order (the backward
    copy of Bidirectional consumes reversed time and its output is reversed back) + cache."""
    B, T, _ = x.shape
    u = U.shape[0]
    xW = x @ W + b
    h = np.zeros((B, u), dtype=x.dtype); c = np.zeros((B, u), dtype=x.dtype)
    H = np.zeros((B, T, u), dtype=x.dtype); Cs = np.zeros_like(H)
    Gt = np.zeros((B, T, 4 * u), dtype=x.dtype)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        z = xW[:, t] + h @ U
        i = hard_sigmoid(z[:, :u]); f = hard_sigmoid(z[:, u:2 * u])
        g = np.tanh(z[:, 2 * u:3 * u]); o = hard_sigmoid(z[:, 3 * u:])
        c = f * c + i * g
        h = o * np.tanh(c)
        H[:, t] = h; Cs[:, t] = c
        Gt[:, t] = np.concatenate([i, f, g, o], axis=1)
    return H, (x, W, U, H, Cs, Gt, reverse)


def lstm_bwd(cache, gH):
    x, W, U, H, Cs, Gt, reverse = cache
    B, T, _ = x.shape
    u = U.shape[0]
    dZ = np.zeros((B, T, 4 * u), dtype=x.dtype)
    dh_rec = np.zeros((B, u), dtype=x.dtype); dc = np.zeros((B, u), dtype=x.dtype)
    order = list(range(T - 1, -1, -1)) if reverse else list(range(T))
    for idx in range(T - 1, -1, -1):
        t = order[idx]
        tp = order[idx - 1] if idx > 0 else None
        i = Gt[:, t, :u]; f = Gt[:, t, u:2 * u]; g = Gt[:, t, 2 * u:3 * u]; o = Gt[:, t, 3 * u:]
        cprev = Cs[:, tp] if tp is not None else np.zeros((B, u), dtype=x.dtype)
        dh = gH[:, t] + dh_rec
        tc = np.tanh(Cs[:, t])
        do = dh * tc
        dct = dh * o * (1 - tc * tc) + dc
        dz = np.concatenate([dct * g * _hs_grad_from_out(i), dct * cprev * _hs_grad_from_out(f),
                             dct * i * (1 - g * g), do * _hs_grad_from_out(o)], axis=1)
        dZ[:, t] = dz
        dc = dct * f
        dh_rec = dz @ U.T
    Hprev = np.zeros_like(H)
    for idx in range(1, T):
        Hprev[:, order[idx]] = H[:, order[idx - 1]]
    dz2 = dZ.reshape(B * T, 4 * u)
    dW = x.reshape(B * T, -1).T @ dz2
    dU = Hprev.reshape(B * T, u).T @ dz2
    db = dz2.sum(axis=0)
    dx = dZ @ W.T
    return dx, dW, dU, db


def gru_fwd(x, W, U, b, reverse=False):
    """Gate order z,r,h; reset_after=False: hh = tanh(xWh + (r*h)Uh + bh)."""
    B, T, _ = x.shape
    u = U.shape[0]
    xW = x @ W + b
    h = np.zeros((B, u), dtype=x.dtype)
    H = np.zeros((B, T, u), dtype=x.dtype)
    Gt = np.zeros((B, T, 3 * u), dtype=x.dtype)  # z, r, hh
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        zr = xW[:, t, :2 * u] + h @ U[:, :2 * u]
        z = hard_sigmoid(zr[:, :u]); r = hard_sigmoid(zr[:, u:])
        hh = np.tanh(xW[:, t, 2 * u:] + (r * h) @ U[:, 2 * u:])
        h = z * h + (1 - z) * hh
        H[:, t] = h
        Gt[:, t] = np.concatenate([z, r, hh], axis=1)
    return H, (x, W, U, H, Gt, reverse)


def gru_bwd(cache, gH):
    x, W, U, H, Gt, reverse = cache
    B, T, _ = x.shape
    u = U.shape[0]
    dZ = np.zeros((B, T, 3 * u), dtype=x.dtype)
    RH = np.zeros((B, T, u), dtype=x.dtype)
    dh_rec = np.zeros((B, u), dtype=x.dtype)
    order = list(range(T - 1, -1, -1)) if reverse else list(range(T))
    for idx in range(T - 1, -1, -1):
        t = order[idx]
        hprev = H[:, order[idx - 1]] if idx > 0 else np.zeros((B, u), dtype=x.dtype)
        z = Gt[:, t, :u]; r = Gt[:, t, u:2 * u]; hh = Gt[:, t, 2 * u:]
        dh = gH[:, t] + dh_rec
        dz = dh * (hprev - hh) * _hs_grad_from_out(z)
        dhh = dh * (1 - z) * (1 - hh * hh)
        drh = dhh @ U[:, 2 * u:].T
        dr = drh * hprev * _hs_grad_from_out(r)
        dZ[:, t] = np.concatenate([dz, dr, dhh], axis=1)
        RH[:, t] = r * hprev
        dh_rec = dh * z + drh * r + np.concatenate([dz, dr], axis=1) @ U[:, :2 * u].T
    Hprev = np.zeros_like(H)
    for idx in range(1, T):
        Hprev[:, order[idx]] = H[:, order[idx - 1]]
    dz2 = dZ.reshape(B * T, 3 * u)
    dW = x.reshape(B * T, -1).T @ dz2
    dU = np.concatenate([Hprev.reshape(B * T, u).T @ dz2[:, :2 * u],
                         RH.reshape(B * T, u).T @ dz2[:, 2 * u:]], axis=1)
    db = dz2.sum(axis=0)
    dx = dZ @ W.T
    return dx, dW, dU, db


# ----------------------------------------------------------------------------------------
# softmax (utils.py:86)
# ----------------------------------------------------------------------------------------
def softmax_fwd(z):
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def softmax_bwd(p, gp):
    return p * (gp - (gp * p).sum(axis=-1, keepdims=True))
