"""Host-side skeleton extraction: affinity (N,K,K,1) -> kinematic tree.

One-shot, K = 24 nodes, float64 — this is host logic in the reference too
(utils/dyna_utils.py:6-171, networkx + numpy) and stays on the host here, without the
networkx dependency: shortest paths come from a dense O(K^2)-per-source Dijkstra whose
distances accumulate from the source outward exactly like a heap-based one (a node's
final distance is min over already-final neighbours of d(v) + w, independent of the
pop order among ties) and in float32, the dtype of the edge weights the reference
hands to its graph library, so every float comparison the reference makes sees
identical numbers.  The two top-k selections use torch.topk on the host like the
reference, so ties resolve identically.

Returned objects mirror what HSVRNNBVH caches (hsvrnn_bvh.py:75-79):
  A        (K,K) float64 tree adjacency
  order    (K,)  int64   evaluation order = priority.indices (root first, parents first)
  dist     (K,)  float64 tree distance from the root in that order = priority.values
  parents  (K,)  int64   parents[root] == root
"""
from __future__ import annotations

from typing import NamedTuple

import numpy as np
import torch

BIG = 1e4


class Skeleton(NamedTuple):
    A: np.ndarray
    order: np.ndarray
    dist: np.ndarray
    parents: np.ndarray


def shortest_paths(W: np.ndarray, big: float = BIG) -> np.ndarray:
    """All-pairs shortest path lengths of the undirected graph whose edge (u,v) exists
    where W[u,v] != 0 and weighs W[u,v]; unreachable pairs get ``big``."""
    K = W.shape[0]
    W = W.astype(np.float32)
    edge = W != 0
    out = np.full((K, K), big, dtype=np.float64)
    for s in range(K):
        d = np.full(K, np.inf, dtype=np.float32)
        d[s] = 0.0
        done = np.zeros(K, dtype=bool)
        for _ in range(K):
            cand = np.where(done, np.inf, d)
            v = int(cand.argmin())
            if not np.isfinite(cand[v]):
                break
            done[v] = True
            nb = edge[v] & ~done
            nd = d[v] + W[v]
            better = nb & (nd < d)
            d[better] = nd[better]
        out[s, done] = d[done]
    return out


def _components(edge: np.ndarray) -> int:
    K = edge.shape[0]
    label = -np.ones(K, dtype=int)
    n = 0
    for s in range(K):
        if label[s] >= 0:
            continue
        stack = [s]
        label[s] = n
        while stack:
            v = stack.pop()
            for u in np.flatnonzero(edge[v]):
                if label[u] < 0:
                    label[u] = n
                    stack.append(int(u))
        n += 1
    return n


def build_skeleton(affinity: np.ndarray, big: float = BIG) -> Skeleton:
    """affinity: (N,K,K,1) or (N,K,K) float array (the detector's get_affinity output)."""
    aff = np.asarray(affinity)
    if aff.ndim == 4:
        aff = aff[..., 0]
    N, K, _ = aff.shape
    infl = aff.max(axis=0)                                            # dyna_utils.py:9
    top = torch.from_numpy(np.ascontiguousarray(infl)).topk(N, dim=-1).indices.numpy()   # :10
    adj = np.zeros((K, K), dtype=np.float32)
    adj[np.arange(K)[:, None], top] = 1
    adj = np.maximum(adj, adj.T)                                      # :13-16

    D = shortest_paths(adj, big)
    if _components(adj != 0) > 1:                                     # :37-67
        total = D.sum(axis=-1)
        root = int(total.argmin())
        rank = np.empty(K)
        rank[total.argsort()] = np.arange(K)
        far = np.flatnonzero(D[root] == big)
        pick = far[0]
        for c in far[1:]:
            if rank[pick] > rank[c]:
                pick = c
        adj[root, pick] = adj[pick, root] = 1
        D = shortest_paths(adj, big)

    # break ties between nodes of equal total distance by nudging shared neighbours' edges (:70-81)
    total = D.sum(axis=-1)
    W = adj.copy()                                                    # float32, like the reference's deepcopy
    nbr = adj != 0
    for k in range(K - 1):
        for kd in range(k + 1, K):
            if total[k] != total[kd]:
                continue
            for n in np.flatnonzero(nbr[k] & nbr[kd]):
                l = kd if infl[n, k] > infl[n, kd] else k
                W[n, l] += 1e-5
                W[l, n] += 1e-5
    D = shortest_paths(np.where(nbr, W, np.float32(0)), big)                                       # :83-97

    total = D.sum(axis=-1)
    root = int(torch.from_numpy(total).topk(K, largest=False).indices[0])   # :101-102
    rank = D[root]
    parents = np.empty(K, dtype=np.int64)
    for k in range(K):                                                # :105-142
        if k == root:
            parents[k] = k
            continue
        mine = np.flatnonzero(adj[k])
        par, gap = -1, -1e3
        for n in mine:
            rd = rank[n] - rank[k]
            if rd < 0 and rd > gap:
                gap, par = rd, n
            elif rd < 0 and rd == gap:
                if infl[k, n] > infl[k, par]:
                    gap, par = rd, n
            elif rd == 0:
                co, co_rank = -1, 1e4
                for m in np.flatnonzero(adj[n]):
                    if m in mine and rank[m] < rank[n] and co_rank > rank[m]:
                        co, co_rank = m, rank[m]
                if co >= 0 and infl[co, n] > infl[co, k]:
                    gap, par = rd, n
        if par < 0:
            par = root
            adj[k, par] = adj[par, k] = 1
        parents[k] = par

    A = np.zeros((K, K), dtype=np.float64)
    kids = np.flatnonzero(parents != np.arange(K))
    A[kids, parents[kids]] = 1
    A[parents[kids], kids] = 1
    D = shortest_paths(np.where(A != 0, W, np.float32(0)), big)                                       # :152-169
    order = np.lexsort((np.arange(K), D[root])).astype(np.int64)      # ascending distance, ties by index
    return Skeleton(A=A, order=order, dist=D[root][order], parents=parents)
