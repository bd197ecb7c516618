"""CPU oracle (TEST INFRASTRUCTURE) -- CTC loss, greedy decode, TF-style beam decode.

Restates what the reference reaches through `K.ctc_batch_cost` (utils.py:98-103) and
`K.ctc_decode` (utils.py:347-357) in Keras 2.2.2 / TF 1.8 (un-vendored).
Parity unpinned by reference tests; pinned by the reference's own known-answer decode pairs
(cellist->celist etc.) and cross-checked against torch.nn.functional.ctc_loss.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np

EPS = 1e-7  # K.epsilon()
NEG_INF = -np.inf


def _logsumexp2(a, b):
    if a == NEG_INF:
        return b
    if b == NEG_INF:
        return a
    m = a if a > b else b
    return m + np.log(np.exp(a - m) + np.exp(b - m))


# ----------------------------------------------------------------------------------------
# CTC loss: K.ctc_batch_cost(labels, y_pred[:, 2:, :], input_length, label_length)
# ----------------------------------------------------------------------------------------
def ctc_loss_and_grad(y_pred, labels, input_length, label_length, skip=2):
    """y_pred (B,T,C) softmax output of the model; labels (B,L) ints (blank = C-1 padding);
    input_length/label_length (B,) or (B,1).  Follows utils.py:102-103 then Keras
    ctc_batch_cost: y = y_pred[:, skip:]; logits = log(y + 1e-7) (time-major); TF ctc_loss
    re-applies log-softmax; blank = C-1; ctc_merge_repeated=True.
    Returns (loss (B,), d loss_b / d y_pred (B,T,C))  [per-sample, NOT divided by B]."""
    B, T, C = y_pred.shape
    dt = y_pred.dtype
    blank = C - 1
    input_length = np.asarray(input_length).reshape(-1).astype(np.int64)
    label_length = np.asarray(label_length).reshape(-1).astype(np.int64)
    y = y_pred[:, skip:, :]
    z = np.log(y + dt.type(EPS))
    lsm = z - (z.max(-1, keepdims=True) + np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1, keepdims=True)))
    loss = np.zeros(B, dtype=dt)
    grad = np.zeros_like(y_pred)
    for b in range(B):
        Tb = int(input_length[b]); L = int(label_length[b])
        lab = [int(v) for v in labels[b, :L]]
        ext = [blank]
        for v in lab:
            ext += [v, blank]
        S = len(ext)
        lp = lsm[b]  # (T', C) log-probs
        alpha = np.full((Tb, S), NEG_INF)
        alpha[0, 0] = lp[0, ext[0]]
        if S > 1:
            alpha[0, 1] = lp[0, ext[1]]
        for t in range(1, Tb):
            for s in range(S):
                a = alpha[t - 1, s]
                if s >= 1:
                    a = _logsumexp2(a, alpha[t - 1, s - 1])
                if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                    a = _logsumexp2(a, alpha[t - 1, s - 2])
                alpha[t, s] = a + lp[t, ext[s]] if a != NEG_INF else NEG_INF
        beta = np.full((Tb, S), NEG_INF)
        beta[Tb - 1, S - 1] = lp[Tb - 1, ext[S - 1]]
        if S > 1:
            beta[Tb - 1, S - 2] = lp[Tb - 1, ext[S - 2]]
        for t in range(Tb - 2, -1, -1):
            for s in range(S):
                a = beta[t + 1, s]
                if s + 1 < S:
                    a = _logsumexp2(a, beta[t + 1, s + 1])
                if s + 2 < S and ext[s] != blank and ext[s] != ext[s + 2]:
                    a = _logsumexp2(a, beta[t + 1, s + 2])
                beta[t, s] = a + lp[t, ext[s]] if a != NEG_INF else NEG_INF
        ll = alpha[Tb - 1, S - 1]
        if S > 1:
            ll = _logsumexp2(ll, alpha[Tb - 1, S - 2])
        loss[b] = -ll
        if ll == NEG_INF:
            continue  # no valid path: TF returns inf loss and zero gradient
        # d(-ll)/d z[t,k] = softmax(z)[t,k] - sum_{s: ext[s]=k} exp(alpha+beta - lp - ll)
        gz = np.exp(lp[:Tb]).astype(np.float64)
        for t in range(Tb):
            for s in range(S):
                ab = alpha[t, s] + beta[t, s]
                if ab != NEG_INF:
                    gz[t, ext[s]] -= np.exp(ab - lp[t, ext[s]] - ll)
        # chain through z = log(y + eps)
        grad[b, skip:skip + Tb, :] = (gz / (y[b, :Tb].astype(np.float64) + EPS)).astype(dt)
    return loss, grad


# ----------------------------------------------------------------------------------------
# Greedy decode: K.ctc_decode(greedy=True) -> tf.nn.ctc_greedy_decoder(merge_repeated=True)
# ----------------------------------------------------------------------------------------
def ctc_greedy_decode(y_pred, input_length=None):
    """y_pred (B,T,C).  Returns (dense (B,T) int64 padded with -1, lengths (B,)).
    argmax takes the first index on ties; emit if != blank and != previous argmax."""
    B, T, C = y_pred.shape
    blank = C - 1
    out = np.full((B, T), -1, dtype=np.int64)
    lens = np.zeros(B, dtype=np.int64)
    for b in range(B):
        Tb = T if input_length is None else int(np.asarray(input_length).reshape(-1)[b])
        prev = -1
        n = 0
        am = np.argmax(y_pred[b, :Tb], axis=-1)
        for t in range(Tb):
            k = int(am[t])
            if k != blank and k != prev:
                out[b, n] = k
                n += 1
            prev = k
        lens[b] = n
    return out, lens


# ----------------------------------------------------------------------------------------
# Beam decode: tf.nn.ctc_beam_search_decoder(beam_width, top_paths=1, merge_repeated=True)
# restated from tensorflow/core/util/ctc/ctc_beam_search.h (r1.8) -- SURVEY A.7
# ----------------------------------------------------------------------------------------
class _Node:
    __slots__ = ("label", "parent", "children", "ob", "ol", "ot", "nb", "nl", "nt")

    def __init__(self, label, parent):
        self.label = label; self.parent = parent; self.children = None
        self.ob = self.ol = self.ot = NEG_INF
        self.nb = self.nl = self.nt = NEG_INF

    def active(self):
        return self.nt != NEG_INF


def _beam_one(logits, beam_width, merge_repeated, dtype=np.float32):
    """logits (T,C) = log(p + 1e-7).  Arithmetic in `dtype` (TF uses float32)."""
    T, C = logits.shape
    blank = C - 1
    f = dtype
    lse = lambda a, b: f(_logsumexp2(float(a), float(b)))
    root = _Node(-1, None)
    root.nt = f(0.0); root.nb = f(0.0); root.nl = NEG_INF
    leaves = [root]
    for t in range(T):
        inp = (logits[t] - logits[t].max()).astype(f)
        branches = sorted(leaves, key=lambda n: -n.nt)  # python sort is stable
        leaves = []
        for b in branches:
            b.ob, b.ol, b.ot = b.nb, b.nl, b.nt
        for b in branches:
            if b.parent is not None:
                if b.parent.active():
                    prev = b.parent.ob if b.label == b.parent.label else b.parent.ot
                    b.nl = lse(b.nl, prev)
                b.nl = f(b.nl + inp[b.label]) if b.nl != NEG_INF else NEG_INF
            b.nb = f(b.ot + inp[blank])
            b.nt = lse(b.nb, b.nl)
            leaves.append(b)

        def bottom():
            return min(leaves, key=lambda n: n.nt)

        def cand(total):
            return total > NEG_INF and (len(leaves) < beam_width or total > bottom().nt)

        for b in branches:
            if not cand(b.ot):
                continue
            if b.children is None:
                b.children = [_Node(k, b) for k in range(C - 1)]
            for c in b.children:
                if c.active():
                    continue
                c.nb = NEG_INF
                prev = b.ob if c.label == b.label else b.ot
                c.nl = f(inp[c.label] + prev) if prev != NEG_INF else NEG_INF
                c.nt = c.nl
                if cand(c.nt):
                    if len(leaves) == beam_width:
                        bt = bottom()
                        bt.nb = bt.nl = bt.nt = NEG_INF
                        leaves.remove(bt)
                    leaves.append(c)
                else:
                    c.ob = c.ol = c.ot = NEG_INF
                    c.nb = c.nl = c.nt = NEG_INF
    best = max(leaves, key=lambda n: n.nt)
    seq = []
    prev_label = -1
    c = best
    while c.parent is not None:
        if not merge_repeated or c.label != prev_label:
            seq.append(c.label)
        prev_label = c.label
        c = c.parent
    seq.reverse()
    return seq, float(best.nt)


def ctc_beam_decode(y_pred, beam_width=10, merge_repeated=True, input_length=None):
    """y_pred (B,T,C) softmax.  Returns (dense (B,T) int64 padded -1, lengths, log-scores).
    Keras: log(transpose(y_pred) + 1e-7) -> decoder; top_paths=1 (utils.py:353-354)."""
    B, T, C = y_pred.shape
    out = np.full((B, T), -1, dtype=np.int64)
    lens = np.zeros(B, dtype=np.int64)
    scores = np.zeros(B, dtype=np.float64)
    for b in range(B):
        Tb = T if input_length is None else int(np.asarray(input_length).reshape(-1)[b])
        lg = np.log(y_pred[b, :Tb].astype(np.float32) + np.float32(EPS))
        seq, sc = _beam_one(lg, beam_width, merge_repeated)
        out[b, :len(seq)] = seq
        lens[b] = len(seq)
        scores[b] = sc
    return out, lens, scores


def labels_to_text(labels, inverse_classes):
    """utils.py:338-345: label == len(inverse_classes) (blank) or -1 -> ''."""
    n = len(inverse_classes)
    return "".join("" if (c == n or c == -1) else str(inverse_classes[int(c)]) for c in labels)
