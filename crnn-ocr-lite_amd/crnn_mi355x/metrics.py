"""Edit-distance metrics with the reference's names and return conventions (utils.py:262-298):
levenshtein -> float, edit_distance -> mean over pairs, normalized_edit_distance -> mean of d/len(truth)."""
import numpy as np


def levenshtein(seq1, seq2):
    """Two-row dynamic programme (the reference fills the full matrix; same result, returned as float)."""
    n2 = len(seq2)
    prev = np.arange(n2 + 1, dtype=np.float64)
    for i, a in enumerate(seq1, 1):
        cur = np.empty(n2 + 1, dtype=np.float64)
        cur[0] = i
        for j, b in enumerate(seq2, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (0 if a == b else 1))
        prev = cur
    return float(prev[n2])


def edit_distance(y_pred, y_true):
    total = len(y_true)
    return sum(levenshtein(p, t) / total for p, t in zip(y_pred, y_true)) if total else 0


def normalized_edit_distance(y_pred, y_true):
    total = len(y_true)
    return sum(levenshtein(p, t) / (len(t) * total) for p, t in zip(y_pred, y_true)) if total else 0
