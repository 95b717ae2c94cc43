"""Product skeleton builder (neural_marionette_amd/skeleton.py, host code) against the trees the
reference's process_affinity_glob produced (tests/golden/g3_trees.npz) and against the oracle."""
import os

import numpy as np
import torch

from neural_marionette_amd.skeleton import build_skeleton, shortest_paths
from oracle import nm_oracle as O


def _fk_order_ok(order, parents):
    seen = set()
    for i, k in enumerate(order):
        assert i == 0 or int(parents[k]) in seen
        seen.add(int(k))


def test_golden_trees(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_trees.npz"))
    for i in range(g["affinity"].shape[0]):
        sk = build_skeleton(g["affinity"][i])
        assert np.array_equal(sk.parents, g["parents"][i]), f"tree {i}"
        assert np.array_equal(sk.A, g["A"][i]), f"tree {i}"
        assert sk.order[0] == g["order"][i][0]
        assert np.array_equal(sk.dist, g["order_values"][i]), f"tree {i}: distances"
        _fk_order_ok(sk.order, sk.parents)


def test_random_trees_match_oracle():
    rng = np.random.default_rng(123)
    for trial in range(60):
        K = 24
        scale = [0.2, 1.0, 3.0][trial % 3]
        p = rng.standard_normal((2, K, K - 1)) * scale
        if trial % 4 == 0:
            p = np.round(p)                      # force plenty of exact ties
        aff = O.affinity_v3(torch.from_numpy(p).float())
        A, order, vals, parents = O.build_tree(aff)
        sk = build_skeleton(aff.numpy())
        assert np.array_equal(sk.parents, parents), trial
        assert np.array_equal(sk.A, A), trial
        assert np.array_equal(sk.dist, vals), trial
        assert np.array_equal(sk.order, order), trial


def test_shortest_paths_small():
    W = np.zeros((4, 4), dtype=np.float32)
    for u, v, w in [(0, 1, 1.0), (1, 2, 1.00001), (0, 2, 3.0)]:
        W[u, v] = W[v, u] = w
    D = shortest_paths(W)
    assert D[0, 2] == np.float32(1.0) + np.float32(1.00001)
    assert D[0, 3] == 1e4 and D[3, 3] == 0.0
