"""CPU: the accuracy oracle (oracle/metrics_oracle.py) against tests/golden/metrics.npz, which the REAL reference class
`utils.metrics.VQAAccuracy` produced (tests/golden/make_golden.py gen_metrics)."""
import os

import numpy as np

from oracle import metrics_oracle as MO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.npz")


def test_accuracy_oracle_matches_reference_running_counters():
    g = np.load(GOLD)
    run = np.zeros(3, dtype=np.int64)
    for i in range(g["running"].shape[0]):
        c, c5, n = MO.accuracy_counts(g[f"logits{i}"], g[f"targets{i}"])
        run += np.array([c, c5, n])
        assert (run == g["running"][i]).all(), (i, run, g["running"][i])
    assert abs(run[0] / run[2] - g["accuracy"][0]) < 1e-12 and abs(run[1] / run[2] - g["accuracy"][1]) < 1e-12


def test_accuracy_oracle_edge_cases():
    x = np.array([[1.0, 3.0, 3.0, 0.0, 3.0, -1.0, 2.0]], dtype=np.float32)
    assert MO.accuracy_counts(x, np.array([1])) == (1, 1, 1)          # first of the tied maxima is the argmax
    assert MO.accuracy_counts(x, np.array([2])) == (0, 1, 1)
    assert MO.accuracy_counts(x, np.array([5])) == (0, 0, 1)          # rank 6
    assert MO.accuracy_counts(x, np.array([9])) == (0, 0, 1)          # target outside the classes never matches
