"""Keras-2.2.2-family initialisers for a freshly built CRNN: glorot_uniform for Conv/Dense,
he_normal (truncated normal, stddev sqrt(2/fan_in)) for the RNN kernels and dense2, orthogonal recurrent
kernels, unit forget bias, BatchNorm gamma=1/beta=0/mean=0/var=1, and the identity affine transform for the
STN's last Dense (utils.py:239-245).  RNG streams cannot match TensorFlow's; only the families do."""
import numpy as np


def _fans(shape):
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def glorot_uniform(rs, shape, fans=None):
    fi, fo = fans if fans else _fans(shape)
    lim = np.sqrt(6.0 / (fi + fo))
    return rs.uniform(-lim, lim, size=shape).astype(np.float32)


def he_normal(rs, shape):
    std = np.sqrt(2.0 / _fans(shape)[0])
    v = rs.normal(0, std, size=shape)
    out = np.abs(v) > 2 * std
    while out.any():
        v[out] = rs.normal(0, std, size=int(out.sum()))
        out = np.abs(v) > 2 * std
    return v.astype(np.float32)


def orthogonal(rs, shape):
    a = rs.normal(0, 1, size=shape)
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    return (u if u.shape == tuple(shape) else vt).astype(np.float32)


def initial_parameters(layout, n_units, gru, seed=None):
    """layout: Engine.layout ({name: (offset, size, dims)}) -> {name: ndarray}."""
    rs = np.random.RandomState(seed)
    p = {}
    for name, (_, _, dims) in layout.items():
        if name.endswith("_dw"):
            v = glorot_uniform(rs, dims, fans=(9 * dims[2], 9))      # Keras kernel shape (3,3,C,1)
        elif name.endswith(("bn1_g", "bn2_g")):
            v = np.ones(dims, np.float32)
        elif name == "stn_d2_w":
            v = np.zeros(dims, np.float32)
        elif name == "stn_d2_b":
            v = np.array([1, 0, 0, 0, 1, 0], np.float32)
        elif name.startswith("rnn") and name.endswith("_w"):
            v = he_normal(rs, dims)
        elif name.startswith("rnn") and name.endswith("_u"):
            v = orthogonal(rs, dims)
        elif name.startswith("rnn") and name.endswith("_b"):
            v = np.zeros(dims, np.float32)
            if not gru:
                v[n_units:2 * n_units] = 1.0
        elif name == "dense2_w":
            v = he_normal(rs, dims)
        elif name.endswith("_b"):
            v = np.zeros(dims, np.float32)
        else:
            v = glorot_uniform(rs, dims)
        p[name] = v
    return p
