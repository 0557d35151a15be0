"""TEST INFRASTRUCTURE - CPU restatement of the reference's CRF (Viterbi) window decoding.

Follows ``postprocess/helpers.py:398-449`` (``viterbi_decode``) and ``:180-186`` (``logsumexp``)
step by step in numpy f64.  Pinned: ``tests/golden/crf_cases.json`` holds paths produced by the
reference's own function (imported in the build container by ``tests/golden/make_golden_crf.py``)
and the known answers of ``tests/unit/test_viterbi_decode.py:28-80``.  Only tests import this.
"""

from __future__ import annotations

import numpy as np


def logsumexp(x: np.ndarray, axis: int = -1) -> np.ndarray:
    xmax = np.max(x, axis=axis, keepdims=True)
    return xmax.squeeze(axis=axis) + np.log(np.sum(np.exp(x - xmax), axis=axis))


def viterbi_decode(logits, switch_cost: float = 2.0, transition_costs=None) -> np.ndarray:
    z = np.asarray(logits, dtype=np.float64)
    if z.ndim == 1:
        z = z.reshape(1, -1)
    t_len, n_classes = z.shape
    emissions = z - logsumexp(z, axis=-1)[:, None]
    if t_len == 1 or n_classes == 1:
        return np.argmax(emissions, axis=-1)
    if transition_costs is None:
        costs = np.full((n_classes, n_classes), float(switch_cost))
        np.fill_diagonal(costs, 0.0)
    else:
        costs = np.asarray(transition_costs, dtype=np.float64)
    delta = np.empty((t_len, n_classes))
    backptr = np.empty((t_len, n_classes), dtype=np.int64)
    delta[0] = emissions[0]
    for t in range(1, t_len):
        scores = delta[t - 1][:, None] - costs               # [prev, cur]
        backptr[t] = np.argmax(scores, axis=0)
        delta[t] = emissions[t] + scores[backptr[t], np.arange(n_classes)]
    path = np.empty(t_len, dtype=np.int64)
    path[-1] = int(np.argmax(delta[-1]))
    for t in range(t_len - 2, -1, -1):
        path[t] = backptr[t + 1][path[t + 1]]
    return path
